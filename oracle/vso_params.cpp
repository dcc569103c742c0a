// CPU ORACLE (test infrastructure) - defaults of vs::Stabilizer::Parameters
// (/root/reference/include/video/Stabilizer.h:76-175) and the constants that
// src/Stabilizer.cpp hard-codes (:611-619, :647-649).
#include <cstring>
#include "vso.h"

extern "C" void vso_params_default(vs_params_c* p) {
    memset(p, 0, sizeof *p);
    p->struct_size = (int32_t)sizeof *p;
    p->logging = 0;
    p->smoothing_radius = 30;
    p->max_corners = 200;
    p->quality_level = 0.01;
    p->min_distance = 30.0;
    p->block_size = 3;
    p->border_type = VS_BORDER_BLACK;
    p->border_size = 0;
    p->crop_n_zoom = 0;
    p->smoothing_method = VS_SMOOTH_BOX;
    p->horizon_lock = 0;
    p->gaussian_sigma = 2.0;
    p->adaptive_smoothing = 0;
    p->min_smoothing_radius = 5;
    p->max_smoothing_radius = 50;
    p->fade_alpha = 0.1f;
    p->fade_duration = 30;
    p->canvas_scale_factor = 1.5f;
    p->temporal_buffer_size = 30;
    p->canvas_blend_weight = 0.7f;
    p->adaptive_canvas_size = 1;
    p->max_canvas_scale = 2.0f;
    p->min_canvas_scale = 1.2f;
    p->edge_blend_radius = 20;
    p->enable_virtual_canvas = 0;
    p->drone_high_freq_mode = 0;
    p->hf_shake_px = 1.5f;
    p->hf_analysis_max_width = 960;
    p->hf_rot_lp_alpha = 0.2f;
    p->enable_conditional_clahe = 1;
    p->hf_dead_zone_threshold = 2.0f;
    p->hf_freeze_duration = 10;
    p->hf_motion_accumulator_decay = 0.9f;
    p->lk_win_size = 15;
    p->lk_max_level = 2;
    p->lk_max_iters = 20;
    p->lk_epsilon = 0.03;
    p->ransac_max_iters = 500;
    p->ransac_threshold = 5.0;
}
