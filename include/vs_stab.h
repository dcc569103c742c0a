/*
 * vs_stab.h - C ABI of libvideo-stab (MI355X / gfx950 build).
 *
 * This is the drop-in boundary for the per-frame stabilization hot path of
 * OmerMersin/video-stab: everything `vs::Stabilizer::stabilize(frame)` does
 * (reference: include/video/Stabilizer.h:177-198, src/Stabilizer.cpp:258-1172).
 * The reference has no FFI of its own (it is a C++ class over cv::Mat); the
 * binding a maintainer adds is the thin C++ class in include/video/Stabilizer.h
 * of this repo, which forwards to the entry points below.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch/OpenCV types cross this ABI;
 *  - every function returns a vs_status; nothing throws across the ABI;
 *  - images are 8-bit, row-major, with an explicit row stride in BYTES;
 *  - `*_dev` entry points take DEVICE pointers and are asynchronous on the
 *    instance's HIP stream; the host-pointer forms copy in/out and synchronise;
 *  - there is no CPU fallback: if no gfx950 device is usable every compute
 *    entry point fails with VS_ERR_NO_DEVICE.
 */
#ifndef VS_STAB_H
#define VS_STAB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: VS_STAGE_COUNT grew to 9 (VS_STAGE_WARP_TABLES: the arrays of vs_stab_get_stage_times), vs_stab_enable_graph is gone
 *    (round 2), vs_batch_* and vs_dev_copy_rate / vs_dev_memcpy_d2d added.  (The pipelined host call is chosen with
 *    vs_stab_set_host_pipeline - Parameters::hostPipeline of the C++ class - not through vs_params_c, whose layout is unchanged.)
 *    Added since without a layout change: vs_batch_create_params (round 4). */
#define VS_STAB_ABI_VERSION 2

typedef enum vs_status {
    VS_OK = 0,
    VS_ERR_INVALID_ARG = 1,
    VS_ERR_NO_DEVICE = 2,     /* no HIP device / kernels cannot run          */
    VS_ERR_HIP = 3,           /* a HIP runtime call failed (see last_error)   */
    VS_ERR_UNSUPPORTED = 4,   /* parameter combination outside the hot path   */
    VS_ERR_SIZE_CHANGED = 5,  /* frame size differs from the instance's       */
    VS_ERR_CAPACITY = 6       /* a device-side capacity was exceeded          */
} vs_status;

typedef enum vs_pixfmt {
    VS_FMT_BGR8 = 0,          /* interleaved B,G,R  (cv::Mat CV_8UC3)         */
    VS_FMT_NV12 = 1,          /* Y plane (h rows) followed by UV plane (h/2)  */
    VS_FMT_GRAY8 = 2
} vs_pixfmt;

/* Stabilizer.cpp:31-38 mapBorderMode() */
typedef enum vs_border {
    VS_BORDER_BLACK = 0,
    VS_BORDER_REFLECT = 1,
    VS_BORDER_REFLECT_101 = 2,
    VS_BORDER_REPLICATE = 3,
    VS_BORDER_WRAP = 4,
    VS_BORDER_FADE = 5
} vs_border;

/* Stabilizer.cpp:797-823 */
typedef enum vs_smoothing {
    VS_SMOOTH_BOX = 0,
    VS_SMOOTH_GAUSSIAN = 1,
    VS_SMOOTH_KALMAN = 2
} vs_smoothing;

/*
 * Flat POD mirror of vs::Stabilizer::Parameters (Stabilizer.h:76-175).
 * Only the fields the live hot path reads are present;
 * strings became enums.  Fill with vs_params_default() first.
 */
typedef struct vs_params_c {
    int32_t struct_size;          /* = sizeof(vs_params_c), ABI check          */
    int32_t logging;              /* Stabilizer.h:79                           */
    int32_t smoothing_radius;     /* :81  default 30                           */
    int32_t max_corners;          /* :82  default 200                          */
    double  quality_level;        /* :83  default 0.01                         */
    double  min_distance;         /* :84  default 30.0                         */
    int32_t block_size;           /* :85  default 3                            */
    int32_t border_type;          /* :87  vs_border, default BLACK             */
    int32_t border_size;          /* :88  default 0                            */
    int32_t crop_n_zoom;          /* :89  default 0                            */
    int32_t smoothing_method;     /* :92  vs_smoothing, default BOX            */
    int32_t horizon_lock;         /* :95  default 0                            */
    double  gaussian_sigma;       /* :93  default 2.0                          */
    int32_t adaptive_smoothing;   /* :114 default 0                            */
    int32_t min_smoothing_radius; /* :115 default 5                            */
    int32_t max_smoothing_radius; /* :116 default 50                           */
    float   fade_alpha;           /* :128 default 0.1                          */
    int32_t fade_duration;        /* :129 default 30                           */
    int32_t enable_virtual_canvas;/* :154 default 0 (BGR8 streams only)        */
    int32_t drone_high_freq_mode; /* :165 default 0                            */
    float   hf_shake_px;          /* :166 default 1.5                          */
    int32_t hf_analysis_max_width;/* :167 default 960                          */
    float   hf_rot_lp_alpha;      /* :168 default 0.2                          */
    int32_t enable_conditional_clahe; /* :169 default 1                        */
    float   hf_dead_zone_threshold;   /* :172 default 2.0                      */
    int32_t hf_freeze_duration;       /* :173 default 10                       */
    float   hf_motion_accumulator_decay; /* :174 default 0.9                   */
    /* --- extensions: constants hard-coded in Stabilizer.cpp:611-619 --------- */
    int32_t lk_win_size;          /* 15  (Stabilizer_legacy.cpp:218 uses 21)   */
    int32_t lk_max_level;         /* 2   (3 pyramid levels)                    */
    int32_t lk_max_iters;         /* 20                                        */
    double  lk_epsilon;           /* 0.03                                      */
    int32_t ransac_max_iters;     /* 500 (Stabilizer.cpp:649)                  */
    double  ransac_threshold;     /* 5.0                                       */
    /* --- virtual canvas (Stabilizer.h:155-162), read when enable_virtual_canvas != 0; carved out of the
     *     reserved tail, so sizeof(vs_params_c) did not change.  preserveEdgeQuality (:161) is never read
     *     by the reference and has no field. ------------------------------------------------------------ */
    float   canvas_scale_factor;  /* :155 default 1.5                          */
    int32_t temporal_buffer_size; /* :156 default 30   (0..256)                */
    float   canvas_blend_weight;  /* :157 default 0.7  (0..1)                  */
    int32_t adaptive_canvas_size; /* :158 default 1                            */
    float   max_canvas_scale;     /* :159 default 2.0                          */
    float   min_canvas_scale;     /* :160 default 1.2                          */
    int32_t edge_blend_radius;    /* :162 default 20                           */
    int32_t reserved[1];
} vs_params_c;

/* Throughput / health counters ("Metrics"). */
typedef struct vs_counters {
    uint64_t frames_in;
    uint64_t frames_out;
    uint64_t detections;          /* goodFeaturesToTrack runs                  */
    int32_t  last_features;       /* points handed to LK on the last frame     */
    int32_t  last_tracked;        /* status != 0                               */
    int32_t  last_inliers;        /* RANSAC inliers of the chosen model        */
    int32_t  last_candidates;     /* GFTT local maxima above the threshold     */
    int32_t  gftt_overflow;       /* !=0: candidate list was truncated         */
    int32_t  reserved[7];
} vs_counters;

/* Per-frame device results, for parity tests and diagnostics. */
typedef struct vs_debug_frame {
    int32_t n_prev;               /* keypoints handed to LK                    */
    int32_t n_valid;              /* pairs after status compaction             */
    int32_t ransac_best_iter;     /* hypothesis index kept (-1: none)          */
    int32_t ransac_iters_run;     /* final niters                              */
    int32_t n_inliers;
    int32_t detected;             /* 1 if GFTT ran on this frame               */
    int32_t n_detected;
    int32_t box_radius;           /* S1 radius actually used (0 if not box)    */
    int32_t intent;               /* MotionIntent chosen for the output frame  */
    int32_t out_index;            /* index of the frame that was warped, or -1 */
    float   transform[3];         /* (dx,dy,da) measured for this frame        */
    float   smoothed[3];          /* smoothed path at out_index                */
    float   warp_matrix[6];       /* T handed to the warp                      */
    double  model[6];             /* refined 2x3 model (double)                */
} vs_debug_frame;

/* Flat mirror of vs::RollCorrection::Parameters (include/video/RollCorrection.h:16-38). */
typedef struct vs_roll_params_c {
    int32_t struct_size;
    int32_t canny_aperture;          /* 3 (only 3 is supported)                */
    double  scale_factor;            /* 0.25                                   */
    double  canny_threshold_low;     /* 50                                     */
    double  canny_threshold_high;    /* 150                                    */
    float   hough_rho;               /* 1                                      */
    float   hough_theta;             /* pi/180                                 */
    int32_t hough_threshold;         /* 100                                    */
    int32_t reserved0;
    double  angle_filter_min;        /* -10 deg                                */
    double  angle_filter_max;        /* +10 deg                                */
    double  angle_smoothing_alpha;   /* 0.1                                    */
    double  angle_decay;             /* 0.995                                  */
    double  max_angle_change_deg;    /* 0.5                                    */
} vs_roll_params_c;

/* Flat mirror of vs::Enhancer::Parameters (include/video/Enhancer.h:12-43). */
typedef struct vs_enh_params_c {
    int32_t struct_size;
    float   brightness;              /* 0   added to every sample (convertTo beta)  */
    float   contrast;                /* 1   multiplies every sample (convertTo alpha) */
    int32_t enable_white_balance;    /* 0                                      */
    float   wb_strength;             /* 1                                      */
    int32_t enable_vibrance;         /* 0                                      */
    float   vibrance_strength;       /* 0.3                                    */
    int32_t enable_unsharp;          /* 0                                      */
    float   sharpness;               /* 0                                      */
    float   blur_sigma;              /* 1   (kernel 2*round(3*sigma)+1 taps, at most 33) */
    int32_t enable_clahe;            /* 0                                      */
    float   clahe_clip_limit;        /* 2                                      */
    int32_t clahe_tile_grid_size;    /* 8   (1..16)                            */
    int32_t enable_denoise;          /* 0   fastNlMeansDenoisingColored(h, h, 7, 21) */
    float   denoise_strength;        /* 10                                     */
    float   gamma;                   /* 1                                      */
    int32_t use_cuda;                /* 0: stage order of the reference's CPU branch, 1: of its CUDA branch */
    int32_t reserved0;
} vs_enh_params_c;

typedef struct vs_stab vs_stab;   /* opaque instance (one video stream)        */
typedef struct vs_roll vs_roll;   /* opaque roll-correction state              */
typedef struct vs_enh vs_enh;     /* opaque enhancer scratch (tables, stream)  */

/* ---- library ------------------------------------------------------------- */
int          vs_abi_version(void);
/* Short tag of the kernel sources this library was built from (hash of the .hip files): measurements that are kept
 * in files (profiles/warp_traffic.json) name the build they belong to. */
const char*  vs_build_tag(void);
const char*  vs_build_info(void);          /* arch, compiler, feature string  */
int          vs_device_count(void);        /* 0 when no usable GPU            */
void         vs_params_default(vs_params_c* p);
const char*  vs_status_string(int status);

/* ---- vs::Stabilizer (Stabilizer.h:177-198) ------------------------------- */
/* Stabilizer(const Parameters&) - Stabilizer.cpp:50-164 */
int vs_stab_create(const vs_params_c* params, int device, vs_stab** out);
/* ~Stabilizer() - Stabilizer.cpp:216-219 */
void vs_stab_destroy(vs_stab* s);
/* clean() - Stabilizer.cpp:221-256 */
int vs_stab_clean(vs_stab* s);
/*
 * stabilize(frame) - Stabilizer.cpp:258-392.  Host frame in, host frame out.
 * *produced = 1 and `out` filled when a stabilized frame is ready, 0 during
 * the warm-up (the reference returns an empty Mat, :263-265,:384-387).
 * `out` must hold out_h rows of out_stride bytes; query with vs_stab_out_size().
 */
int vs_stab_push(vs_stab* s, const uint8_t* data, int w, int h, size_t stride,
                 int fmt, uint8_t* out, size_t out_stride, int* produced);
/* flush() - Stabilizer.cpp:394-400 */
int vs_stab_flush(vs_stab* s, uint8_t* out, size_t out_stride, int* produced);
/* Device-pointer forms: asynchronous on the instance stream, no host sync.
 * `produced` is decided on the host from the frame count alone (E0). */
int vs_stab_push_dev(vs_stab* s, const void* d_data, int w, int h, size_t stride,
                     int fmt, void* d_out, size_t out_stride, int* produced);
/* n consecutive pushes of one geometry in one call: the j-th result that becomes due goes to d_outs[j]; *produced = how many did */
int vs_stab_push_dev_n(vs_stab* s, const void* const* d_frames, int n, int w, int h, size_t stride, int fmt,
                       void* const* d_outs, size_t out_stride, int* produced);
int vs_stab_flush_dev(vs_stab* s, void* d_out, size_t out_stride, int* produced);
int vs_stab_sync(vs_stab* s);
/* size of the frames stabilize() returns for w x h input (crop/border rules,
 * Stabilizer.cpp:981-990,1108-1127) */
int vs_stab_out_size(const vs_stab* s, int w, int h, int* out_w, int* out_h);
/* size of the frame the LAST successful push/flush produced.  Equals
 * vs_stab_out_size() except for the final frame of a flush when a border pad
 * is configured: the reference returns that frame unpadded (Stabilizer.cpp:
 * 774-780); it is then stored top-left in `out`, the rest zero. */
int vs_stab_last_out_dims(const vs_stab* s, int* w, int* h);
int vs_stab_get_counters(vs_stab* s, vs_counters* out);      /* synchronises  */
int vs_stab_get_debug(vs_stab* s, vs_debug_frame* out);      /* synchronises  */
/* enableVirtualCanvas, state after the last output: {canvas w, canvas h, canvas scale (float bits), empty regions,
 * regions filled from the temporal buffer, temporal index of the last fill or -1, window x, window y}
 * (Stabilizer.cpp:2083-2088, 2232-2241, 2115-2132).  Zeros when the canvas is off. */
int vs_stab_canvas_info(const vs_stab* s, int32_t info[8]);
/* copies the debug arrays of the last push: any pointer may be NULL.
 * prev/curr: n_prev * 2 floats; status: n_prev bytes; inliers: n_valid bytes;
 * detected: n_detected * 2 floats; gray: analysis image (aw*ah bytes). */
int vs_stab_get_debug_arrays(vs_stab* s, float* prev_pts, float* curr_pts,
                             uint8_t* status, uint8_t* inliers,
                             float* detected_pts, uint8_t* gray, int* aw, int* ah);
const char* vs_stab_last_error(const vs_stab* s);
void*       vs_stab_stream(vs_stab* s);    /* hipStream_t of the instance     */
/* Host pipeline for vs_stab_push / vs_stab_flush (replaces the synchronous upload - compute - download of the reference's
 * GPU branch, /root/reference/src/Stabilizer.cpp:1021-1031): a call returns the frame the call BEFORE it computed - one
 * more call of latency than vs::Stabilizer::stabilize(): clamp(smoothingRadius,5,35) empty results instead of one less -
 * while its own frame is uploaded; the analysis and the warp of that frame then run behind the caller's back.  Same
 * frames in the same order; vs_stab_flush first hands out the frame that is still held.  Off by default.  Choose while no
 * frame is queued. */
int vs_stab_set_host_pipeline(vs_stab* s, int enable);
/* Page-locked host memory for frames handed to vs_stab_push / vs_stab_flush and the other host entry points: transfers
 * from and to such buffers are DMA transfers of their own (pageable memory is staged by the runtime, about half the
 * rate). */
int  vs_host_alloc(void** p, size_t bytes);
void vs_host_free(void* p);
/* The same for memory the caller already owns (a cv::Mat's buffer): page-locks it in place / lets it go again.  The memory
 * must be unregistered before it is freed. */
int  vs_host_register(void* p, size_t bytes);
int  vs_host_unregister(void* p);
/* Deferred output for vs_stab_push_dev / vs_stab_flush_dev (batch / file-to-file use): the
 * warps of up to `frames` (1..32) consecutive results are issued as ONE kernel launch, each
 * result into the d_out its push named.  Results are complete after vs_stab_sync(); with
 * frames > 1 every push must be given its own d_out until then.  frames = 1 (default):
 * every push issues its own warp.  The host entry points (vs_stab_push / vs_stab_flush)
 * always deliver their result before returning. */
int vs_stab_set_warp_batch(vs_stab* s, int frames);
/* Batch mode for vs_stab_push_dev / vs_stab_flush_dev: the analysis of `frames` (1..64)
 * consecutive pushes - goodFeaturesToTrack, calcOpticalFlowPyrLK and the RANSAC
 * hypothesis scoring, all latency-bound on one frame - runs as ONE launch per stage over
 * the whole group; the ordered part (hypothesis selection + trajectory append, smoothing)
 * stays per frame, and the warps go out together as with vs_stab_set_warp_batch(frames)
 * (at most 32 frames per warp launch: a batch of 64 is two launches back to back).
 * Results are bit-identical to frames = 1 and complete after vs_stab_sync(); every push must
 * be given its own d_out until then.  Must be chosen before the first frame (or after
 * vs_stab_clean).  BGR8, GRAY8 and NV12 frames; border padding and crop-and-zoom (BGR8
 * only, like everywhere) run batched too; the "fade" border, the virtual canvas and
 * adaptive smoothing keep the per-frame path (each of their outputs depends on the one
 * before it or on a host decision).  Instances of one device share its HIP streams and
 * are to be driven from ONE host thread. */
int vs_stab_set_batch(vs_stab* s, int frames);
/* Zero-copy input for vs_stab_push_dev: the frame is read where the caller put it (decoder
 * surface pool, resident clip) instead of being copied into the instance's queue - the
 * reference aliases the caller's cv::Mat the same way (Stabilizer.cpp:376).  The buffer
 * must stay valid and unchanged until the result of the same push count has been produced
 * (clamp(smoothingRadius,5,35) further pushes plus twice the batch depth: the warps of a
 * batch are issued with the next one) and vs_stab_sync has returned, or the queue has been
 * drained with vs_stab_flush_dev.  Rows may be padded (decoder pitch); all frames in flight
 * share one pitch, which may change only while nothing is queued.  The frame queue must be
 * empty when the mode is switched. */
int vs_stab_set_zero_copy(vs_stab* s, int enable);
/* Decoder hand-off (the step in front of the path: the reference's capture strings end in
 * `nvv4l2decoder ! nvvidconv ! BGR`, src/CamCap.cpp:49-52,66-72).  Hardware decoders export
 * NV12 surfaces as a Y plane and an interleaved UV plane with a common pitch, the UV plane
 * `uv_offset` bytes behind the Y pointer (rocDecode: pitch * aligned surface height; VA-API:
 * offsets[1]) - not as one block of h*3/2 rows.  Sets that offset for the frames given to
 * vs_stab_push_dev (`in`) and for the surfaces it fills (`out`); 0 = contiguous (h * pitch).
 * With zero-copy input the stabilizer then reads decoder surfaces and writes encoder surfaces
 * in place: no repacking blit on either side.  The frame queue must be empty. */
int vs_stab_set_nv12_layout(vs_stab* s, size_t in_uv_offset, size_t out_uv_offset);

/* ---- several streams of one device scheduled together (BASELINE configs[4]: 64 streams = 8 per GPU) ----------------------
 * The reference runs one Stabilizer per stream, each with its own cv::cuda::Stream objects (src/Stabilizer.cpp:102-104); here a
 * GROUP runs one schedule for all its streams: every step, the frames all members have queued go through ONE launch per stage
 * (one argument block per frame, whichever stream it belongs to), the ordered tails through one launch with a workgroup per
 * stream, the warps through launches of 32 frames.  Results are bit-identical to n_streams independent vs_stab instances.
 *   vs_batch_create      n_streams members with the same parameters, in batch mode with `frames_per_step` frames per stream and
 *                        step (1..64; n_streams x frames_per_step frames are analysed together: 8 x 8 fills the device like one
 *                        stream's batch of 64).  Adaptive smoothing, the fade border and the virtual canvas are per-stream
 *                        modes of vs_stab_* (their outputs depend on each other or on a host decision) and are refused here.
 *   vs_batch_create_params  the same with one parameter block per stream (params_per_stream[n_streams]).  The blocks may differ
 *                        in whatever does not shape a launch: smoothing radius and method, horizon lock, the drone filters'
 *                        settings, corner count and quality.  Frame geometry, analysis size (drone mode on or off for all),
 *                        pyramid depth, tracking window, RANSAC iteration count and the border / crop-and-zoom mode are common;
 *                        a step refuses members that disagree on them (VS_ERR_INVALID_ARG).
 * A standalone vs_stab instance in batch mode (vs_stab_set_batch) runs the same schedule as a group of one.
 *   vs_batch_push_dev    one frame per stream (d_frames[i] == NULL: none for stream i this time); device pointers, one geometry,
 *                        pitch and format for all; produced[i] = 1 when the push made an output of stream i due - it is complete
 *                        after vs_batch_sync, in d_outs[i].  A step runs when a member has frames_per_step frames queued.
 *   vs_batch_flush_dev   drains the group, then the next queued frame of every stream (Stabilizer::flush).
 *   vs_batch_stream      the member instance i: for the per-stream getters (vs_stab_get_counters, vs_stab_get_debug, vs_stab_sync,
 *                        ...).  Its frames are pushed through the group only: vs_stab_push* / flush* / clean / set_* on a member
 *                        return VS_ERR_INVALID_ARG, vs_stab_destroy ignores it (the group owns its members).
 * One host thread drives a group (and all instances of a device). */
typedef struct vs_batch vs_batch;
int vs_batch_create(int device, int n_streams, const vs_params_c* params, int frames_per_step, vs_batch** out);
int vs_batch_create_params(int device, int n_streams, const vs_params_c* params_per_stream, int frames_per_step, vs_batch** out);
void vs_batch_destroy(vs_batch* b);
int vs_batch_streams(const vs_batch* b);
vs_stab* vs_batch_stream(vs_batch* b, int i);
int vs_batch_set_zero_copy(vs_batch* b, int enable);
int vs_batch_set_nv12_layout(vs_batch* b, size_t in_uv_offset, size_t out_uv_offset);
int vs_batch_push_dev(vs_batch* b, const void* const* d_frames, int w, int h, size_t stride, int fmt, void* const* d_outs,
                      size_t out_stride, int* produced);
int vs_batch_flush_dev(vs_batch* b, void* const* d_outs, size_t out_stride, int* produced);
int vs_batch_sync(vs_batch* b);
const char* vs_batch_last_error(const vs_batch* b);

/* Per-stage device timing with HIP events recorded on the instance stream
 * (SURVEY.md section 5 "Tracing").  mode 0 = off, 1 = warp stage only,
 * 2 = every stage, 3 = warp stage and the coordinate tables of batched warps.
 * vs_stab_get_stage_times() synchronises, adds the elapsed
 * time of every event pair recorded since the last call into total_ms[stage]
 * / launches[stage] (arrays of VS_STAGE_COUNT) and resets. */
enum {
    VS_STAGE_COPY_IN = 0,   /* frame into the queue ring                       */
    VS_STAGE_GRAY = 1,      /* resize + BGR2GRAY                               */
    VS_STAGE_PYRAMID = 2,   /* pyrDown + Scharr                                */
    VS_STAGE_LK = 3,
    VS_STAGE_RANSAC = 4,    /* compaction + score + select/refine              */
    VS_STAGE_TRAJ = 5,      /* trajectory append + emit                        */
    VS_STAGE_GFTT = 6,
    VS_STAGE_WARP = 7,      /* warpAffine kernel(s) only                       */
    VS_STAGE_WARP_TABLES = 8, /* batch mode: coordinate tables of a batch's warps (queued behind the batch tail) */
    VS_STAGE_COUNT = 9
};
int vs_stab_set_profiling(vs_stab* s, int mode);
int vs_stab_get_stage_times(vs_stab* s, double* total_ms, int64_t* launches);

/* ---- device memory helpers (so callers need no HIP headers) --------------- */
int vs_dev_set_device(int device);          /* device used by the vs_dev_* / vs_op_* calls of this thread */
int vs_dev_malloc(void** d_ptr, size_t bytes);
int vs_dev_free(void* d_ptr);
int vs_dev_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int vs_dev_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int vs_dev_memset(void* d_dst, int value, size_t bytes);
int vs_dev_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes);
int vs_dev_sync(void);
/* Bandwidth yardstick for roofline reports: GB/s (read + written bytes) of a plain device copy of `bytes` bytes - 16 bytes
 * per lane, streaming stores - timed with HIP events around `iters` back-to-back launches that walk through buffers of
 * together more than 512 MB, so no launch finds its input in the Infinity Cache. */
int vs_dev_copy_rate(size_t bytes, int iters, double* gbytes_per_s);
const char* vs_last_error(void);           /* thread-local, op-level calls    */

/* ---- stage operators on device buffers (stream = hipStream_t or NULL) ------
 * Each one is the device counterpart of one OpenCV call made by
 * src/Stabilizer.cpp; the file:line of that call is cited per function.     */

/* cv::warpAffine(src,dst,T,size,INTER_LINEAR,BORDER_CONSTANT) -
 * Stabilizer.cpp:1056-1060.  M = forward 2x3 float matrix as the reference
 * builds it (:902-908).  `batch` frames of identical geometry, frame b at
 * d_src + b*src_frame_bytes with matrix M + 6*b.  cn = 3 (BGR8) or 1. */
int vs_op_warp_affine(const void* d_src, size_t src_stride, size_t src_frame_bytes,
                      void* d_dst, size_t dst_stride, size_t dst_frame_bytes,
                      int w, int h, int cn, const float* M, int batch, void* stream);
/* NV12 surface: Y plane warped with M, interleaved UV plane at half
 * resolution with the translation halved. */
int vs_op_warp_affine_nv12(const void* d_src, size_t src_stride, void* d_dst,
                           size_t dst_stride, int w, int h, const float* M,
                           int batch, size_t src_frame_bytes, size_t dst_frame_bytes,
                           void* stream);
/* std::cos / std::sin / std::atan2 on float as the reference calls them (Stabilizer.cpp:662, 902-908, 1689: the host libm's
 * cosf / sinf / atan2f), evaluated by the DEVICE build of the library's restatement: the sum over i in [start, start + count) of
 * a 64-bit mix of (i, bits of f(argument i)) - fn 0 cosf, 1 sinf, 2 atanf: argument i = the float with bit pattern (uint32_t)i;
 * fn 3 atan2f: pair i of a fixed generator.  tests/test_libm.py compares it with the same sum over the host libm's values. */
int vs_op_libm_checksum(int fn, uint64_t start, uint64_t count, uint64_t* result);
/* cv::resize(INTER_LINEAR) + cv::cvtColor(BGR2GRAY) - Stabilizer.cpp:304-305,
 * 448-450.  fmt BGR8 (resize then gray), GRAY8 / NV12 (luma plane resize). */
int vs_op_resize_gray(const void* d_src, size_t src_stride, int sw, int sh, int fmt,
                      void* d_dst, size_t dst_stride, int dw, int dh, void* stream);
/* cv::pyrDown as used inside calcOpticalFlowPyrLK - Stabilizer.cpp:611 */
int vs_op_pyr_down(const void* d_src, size_t src_stride, int sw, int sh,
                   void* d_dst, size_t dst_stride, void* stream);
/* Scharr derivative image (int16, interleaved dx,dy) of calcOpticalFlowPyrLK */
int vs_op_scharr(const void* d_src, size_t src_stride, int w, int h,
                 void* d_dst /* int16[h][w][2] */, void* stream);
/* One pyramid level as batch mode builds it: the Scharr derivatives of `items` images (w x h, src_frame_bytes apart) and, when
 * d_next != NULL, their pyrDown ((w+1)/2 x (h+1)/2, next_frame_bytes apart) from one staged read of each image. */
int vs_op_pyr_level(const void* d_src, size_t src_stride, size_t src_frame_bytes, int w, int h, void* d_der /* int16[h][w][2] per image */,
                    void* d_next, size_t next_stride, size_t next_frame_bytes, int items, void* stream);
/* cv::calcOpticalFlowPyrLK(prev,next,prevPts,nextPts,status,err,win,maxLevel,
 * TermCriteria(COUNT+EPS,iters,eps)) - Stabilizer.cpp:611-619 */
int vs_op_pyr_lk(const void* d_prev, const void* d_next, size_t stride, int w, int h,
                 const float* d_prev_pts, int n, float* d_next_pts,
                 uint8_t* d_status, float* d_err,
                 int win, int max_level, int max_iters, double eps, void* stream);
/* cv::goodFeaturesToTrack(gray,corners,maxCorners,quality,minDistance,noArray(),
 * blockSize) - Stabilizer.cpp:354-358,740-744.  d_pts: maxCorners*2 floats,
 * d_count: one int32.  d_eig (optional, w*h floats) receives the min-eigen map. */
int vs_op_gftt(const void* d_gray, size_t stride, int w, int h, int max_corners,
               double quality, double min_distance, int block_size,
               float* d_pts, int32_t* d_count, float* d_eig, void* stream);
/* cv::estimateAffinePartial2D(from,to,noArray(),RANSAC,thr,maxIters) -
 * Stabilizer.cpp:647-649.  d_model: 6 doubles (or NaN when no model),
 * d_inliers: n bytes, d_info: int32[4] = {ok, best_iter, iters_run, n_inliers}. */
int vs_op_estimate_affine_partial2d(const float* d_from, const float* d_to, int n,
                                    double thr, int max_iters, double* d_model,
                                    uint8_t* d_inliers, int32_t* d_info, void* stream);

/* ---- roll correction: vs::RollCorrection (RollCorrection.h:12-50, RollCorrection.cpp:16-155) ---- */
/* RollCorrection::Parameters defaults, RollCorrection.h:16-38 */
void vs_roll_params_default(vs_roll_params_c* p);
/* The function-static sSmoothedAngle / sFirstCall (RollCorrection.cpp:13-14) become
 * per-object state. */
int vs_roll_create(const vs_roll_params_c* params, int device, vs_roll** out);
void vs_roll_destroy(vs_roll* r);
const char* vs_roll_last_error(const vs_roll* r);
/* Parameters travel with every autoCorrectRoll call while the smoothed angle persists
 * (RollCorrection.cpp:16-19): replaces the parameters, keeps the state. */
int vs_roll_set_params(vs_roll* r, const vs_roll_params_c* params);
/* cv::Mat RollCorrection::autoCorrectRoll(const cv::Mat&, const Parameters&),
 * RollCorrection.cpp:16-155.  BGR8 in, BGR8 out of the same size; synchronous. */
int vs_roll_correct(vs_roll* r, const uint8_t* data, int w, int h, size_t stride,
                    uint8_t* out, size_t out_stride);
/* Same with frames in HBM; the warp is left in flight on the object's stream
 * (vs_roll_sync to wait). */
int vs_roll_correct_dev(vs_roll* r, const void* d_data, int w, int h, size_t stride,
                        void* d_out, size_t out_stride);
int vs_roll_sync(vs_roll* r);
/* autoCorrectRoll for an NV12 surface in HBM (BASELINE configs[2]: decoder surfaces), ASYNCHRONOUS.  The reference has no NV12
 * path; defined as the BGR operator's geometry applied per plane: the line search (resize x scale_factor, Canny, HoughLines,
 * RollCorrection.cpp:35-119) runs on the luma plane (a gray picture: no cvtColor), the rotation about the picture centre
 * (:141-149, BORDER_REPLICATE) is applied to the luma plane and, with the translation halved, to the half-size interleaved
 * chroma plane.  uv_offset / out_uv_offset: where the chroma plane starts (0 = h * pitch).  The call hands the frame over and
 * returns (it waits only when 128 frames are pending): eight consecutive frames form a batch whose line searches a worker
 * thread (five of them; environment VS_ROLL_WORKERS: 1 .. 8) queues as ONE launch per stage on its own stream; when the batch's results (24 bytes per frame) have arrived the
 * smoothed angle advances - in call order, on the host - and the rotations are queued.  Results are complete after
 * vs_roll_sync (which closes an incomplete batch); surfaces and result buffers must stay untouched until then.
 * vs_roll_get_state (after vs_roll_sync) reports the last frame. */
int vs_roll_correct_nv12_dev(vs_roll* r, const void* d_surface, int w, int h, size_t pitch, size_t uv_offset,
                             void* d_out, size_t out_pitch, size_t out_uv_offset);
/* n surfaces of one layout, in call order (n calls of the above in one) */
int vs_roll_correct_nv12_dev_n(vs_roll* r, const void* const* d_surfaces, void* const* d_outs, int n, int w, int h,
                               size_t pitch, size_t uv_offset, size_t out_pitch, size_t out_uv_offset);
/* smoothed angle (sSmoothedAngle), the angle detected on the last frame, lines found / used */
int vs_roll_get_state(const vs_roll* r, double* smoothed_deg, double* detected_deg,
                      int* n_lines, int* n_used);
/* cv::Canny(gray, edges, low, high, 3, false) - RollCorrection.cpp:54-61 (there cv::cuda) */
int vs_op_canny(const void* d_gray, size_t stride, int w, int h, double low, double high,
                void* d_edges, size_t edges_stride, void* stream);
/* cv::HoughLines(edges, lines, rho, theta, threshold) - RollCorrection.cpp:66-73.
 * d_lines: max_lines (rho,theta) float pairs in OpenCV order (votes descending),
 * d_count: one int32.  At most 8192 peaks are ranked. */
int vs_op_hough_lines(const void* d_edges, size_t stride, int w, int h, float rho, float theta,
                      int threshold, float* d_lines, int max_lines, int32_t* d_count, void* stream);
/* cv::warpAffine(src, dst, M(2x3 double, forward), dsize, INTER_LINEAR, border) -
 * RollCorrection.cpp:146-149 (BORDER_REPLICATE there).  border: VS_BORDER_BLACK | VS_BORDER_REPLICATE */
int vs_op_warp_affine_ex(const void* d_src, size_t src_stride, int sw, int sh, void* d_dst,
                         size_t dst_stride, int dw, int dh, int cn, const double* M, int border,
                         void* stream);

/* ---- auto zoom/crop: vs::AutoZoomCrop (AutoZoomCrop.h:7-17, AutoZoomCrop.cpp:102-283) ---- */
typedef struct vs_azc vs_azc;
int vs_azc_create(int device, vs_azc** out);
void vs_azc_destroy(vs_azc* a);
const char* vs_azc_last_error(const vs_azc* a);
/* cv::Mat AutoZoomCrop::autoZoomCrop(const cv::Mat& corrected, double marginPercent)
 * (marginPercent is ignored by the reference, AutoZoomCrop.cpp:102).  BGR8 (cn 3) or gray
 * (cn 1).  `out` must hold max(w*h, 640*360)*cn bytes and receives packed rows; the
 * result is 640x360, or the unchanged w x h frame on the reference's fall-back paths
 * (no contour :149-152, empty crop :238-249).  Synchronous. */
int vs_azc_apply(vs_azc* a, const uint8_t* data, int w, int h, size_t stride, int cn,
                 uint8_t* out, int* out_w, int* out_h);
/* Same with the frame in HBM; out_stride >= max(w,640)*cn.  The scaled crop is left in
 * flight on the object's stream (vs_azc_sync). */
int vs_azc_apply_dev(vs_azc* a, const void* d_data, int w, int h, size_t stride, int cn,
                     void* d_out, size_t out_stride, int* out_w, int* out_h);
int vs_azc_sync(vs_azc* a);
/* autoZoomCrop for an NV12 surface in HBM, ASYNCHRONOUS (the BGR operator's geometry per plane: content mask from the luma
 * plane, gray > 1; the crop rectangle as it is for the luma plane and halved (x/2, y/2, max(1, w/2), max(1, h/2)) for the
 * half-size interleaved chroma plane; each plane scaled to its share of 640 x 360 by the reference's scale matrix,
 * AutoZoomCrop.cpp:246-270).  d_out receives 640 x 360 (chroma 320 x 180 at out_uv_offset) or, on the fall-back paths, the
 * unchanged w x h surface: out_pitch >= max(w, 640), out_uv_offset >= max(h, 360) * out_pitch.  The call hands the frame over
 * and returns a ticket; eight consecutive frames of one geometry form a batch whose mask kernels are one launch each and
 * whose bit masks reach the host with one copy; the contour logic runs on the object's worker threads (it is host work in the
 * reference too, :141-147), a frame each; the worker that finishes a batch's last contour queues the crop-and-scale of the
 * batch as one launch.  vs_azc_result(ticket) waits for that frame's host part and tells what came out, vs_azc_sync completes
 * the pixels (both close an incomplete batch).  Four batches in flight, twelve worker threads (environment VS_AZC_WORKERS:
 * 1 .. 16), results of the last 1024 tickets kept. */
int vs_azc_apply_nv12_dev(vs_azc* a, const void* d_surface, int w, int h, size_t pitch, size_t uv_offset,
                          void* d_out, size_t out_pitch, size_t out_uv_offset, int64_t* ticket);
int vs_azc_apply_nv12_dev_n(vs_azc* a, const void* const* d_surfaces, void* const* d_outs, int n, int w, int h,
                            size_t pitch, size_t uv_offset, size_t out_pitch, size_t out_uv_offset, int64_t* tickets);
int vs_azc_result(vs_azc* a, int64_t ticket, int* out_w, int* out_h, int32_t* info8);
/* Diagnostics of the asynchronous path: out9 = {frames through the worker threads; seconds, summed over the threads: without a
 * frame to work on, waiting for the masks of the batches on their way (one worker at a time does), in the contour logic, queueing
 * launches and publishing; batches; seconds, summed over the batches: from a batch's launches to the arrival of its masks on the
 * host, from there to its crop-and-scale launch; seconds the caller waited for a free batch slot}. */
int vs_azc_worker_times(vs_azc* a, double* out9);
/* info8 = {n_contours, contour_points, crop_x, crop_y, crop_w, crop_h, iterations, cropped} */
int vs_azc_get_info(const vs_azc* a, int32_t* info8);
/* cvtColor + threshold(gray,1,255,BINARY) + morphologyEx(MORPH_CLOSE, 5x5 ellipse) -
 * AutoZoomCrop.cpp:111-139 (there cv::cuda) */
int vs_op_content_mask(const void* d_src, size_t stride, int w, int h, int cn, void* d_mask,
                       size_t mask_stride, void* stream);
/* Host part of the stage: findContours(EXTERNAL, SIMPLE) -> largest contour -> filled mask ->
 * interior rectangle -> aspect fix (AutoZoomCrop.cpp:141-228) on a HOST mask (the reference
 * also runs this on the CPU, :141-147).  filled_out (optional, w*h bytes) receives the
 * drawContours(FILLED) mask.  Needs no device. */
int vs_azc_crop_from_mask(const uint8_t* mask, int w, int h, size_t stride, int32_t* info8,
                          uint8_t* filled_out);
/* cv::boundingRect of every contour cv::findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) returns, in the order of
 * that vector: the region list of the virtual canvas (Stabilizer.cpp:2232-2241).  HOST mask; xywh receives up to
 * max_boxes rectangles, *n_boxes the number found.  Needs no device. */
int vs_op_external_boxes(const uint8_t* mask, int w, int h, size_t stride, int32_t* xywh, int max_boxes,
                         int32_t* n_boxes);

/* ---- image enhancer: vs::Enhancer (Enhancer.h:10-60, Enhancer.cpp:138-239) ------------ */
/* Enhancer::Parameters defaults, Enhancer.h:12-43 */
void vs_enh_params_default(vs_enh_params_c* p);
/* enhanceImage is a static function without state (Enhancer.cpp:138); the object only owns
 * the device tables, scratch frames and the stream the work is queued on. */
int vs_enh_create(int device, vs_enh** out);
void vs_enh_destroy(vs_enh* e);
const char* vs_enh_last_error(const vs_enh* e);
/* cv::Mat Enhancer::enhanceImage(const cv::Mat& input, const Parameters&), Enhancer.cpp:138-239.
 * BGR8 in, BGR8 out of the same size; synchronous.  Stage order: params->use_cuda = 0 the CPU
 * branch (:142-181: white balance, brightness/contrast, CLAHE, vibrance, unsharp, gamma),
 * 1 the CUDA branch (:183-233: brightness/contrast, unsharp, white balance, vibrance, CLAHE,
 * gamma; fastNlMeansDenoisingColored, :165-169, follows the unsharp mask in both); each stage
 * computes what the CPU OpenCV primitive computes. */
int vs_enh_apply(vs_enh* e, const vs_enh_params_c* params, const uint8_t* data, int w, int h,
                 size_t stride, uint8_t* out, size_t out_stride);
/* Same with frames in HBM (d_out must not alias d_data); left in flight on the object's
 * stream (vs_enh_sync to wait). */
int vs_enh_apply_dev(vs_enh* e, const vs_enh_params_c* params, const void* d_data, int w, int h,
                     size_t stride, void* d_out, size_t out_stride);
/* n frames of one geometry; one launch per pass over all frames when the stage list has no
 * per-frame statistic or multi-pass stage (no white balance / CLAHE / denoise), frame by
 * frame otherwise. */
int vs_enh_apply_batch_dev(vs_enh* e, const vs_enh_params_c* params, const void* const* d_frames,
                           void* const* d_outs, int n, int w, int h, size_t stride,
                           size_t out_stride);
int vs_enh_sync(vs_enh* e);
/* passes over the frame (kernel launches that read it) the last apply needed */
int vs_enh_last_passes(const vs_enh* e);
/* cv::cvtColor on packed 8-bit 3-channel pixels (Enhancer.cpp:43,56,61,68), on the object's stream */
enum vs_cvt_code { VS_CVT_BGR2HSV = 0, VS_CVT_HSV2BGR = 1, VS_CVT_BGR2LAB = 2, VS_CVT_LAB2BGR = 3 };
int vs_enh_cvt_color(vs_enh* e, int code, const void* d_src, void* d_dst, size_t npix);
/* cv::GaussianBlur(src, dst, Size(0,0), sigma) on BGR8 (Enhancer.cpp:160-161), on the object's stream */
int vs_enh_gaussian_blur(vs_enh* e, const void* d_src, size_t stride, int w, int h, double sigma,
                         void* d_dst, size_t dstride);

/* ------------------------------------------------------------------ config layer (host only, no device)
 * The YAML the reference's example mains read through cv::FileStorage and the key -> parameter mapping each of
 * them repeats (examples/config.yaml:1-159; examples/vs.cpp:50-168; examples/vsg.cpp:1007-1112), plus the
 * st_mtime test of their hot reload (vs.cpp:199-200,381-394).  Scalars and `>>` conversions follow OpenCV's YAML
 * reader (see csrc/config.cpp for the rules restated).  Key paths are section names joined by '.',
 * e.g. "stabilizer.smoothing_radius". */
typedef struct vs_config vs_config;

typedef enum vs_config_kind_e {
    VS_CFG_NONE = 0,      /* absent, or an empty value */
    VS_CFG_INT = 1,
    VS_CFG_REAL = 2,
    VS_CFG_STRING = 3,
    VS_CFG_MAP = 4,
    VS_CFG_SEQ = 5
} vs_config_kind_e;

/* VS_ERR_INVALID_ARG when the file cannot be read or is malformed (vs_last_error names the line). */
int vs_config_open(const char* path, vs_config** out);
int vs_config_parse(const char* text, size_t len, vs_config** out);
void vs_config_close(vs_config* c);
int vs_config_kind(const vs_config* c, const char* key_path);
int vs_config_size(const vs_config* c, const char* key_path);      /* entries of a map / sequence, 1 for a scalar */
/* `node >> value` of cv::FileNode: an absent key gives 0 / 0.0 / "", a value of the wrong kind INT_MAX / DBL_MAX /
 * FLT_MAX / "", a real read as int is rounded half to even. */
int vs_config_get_int(const vs_config* c, const char* key_path, int32_t* v);
int vs_config_get_double(const vs_config* c, const char* key_path, double* v);
int vs_config_get_float(const vs_config* c, const char* key_path, float* v);
int vs_config_get_string(const vs_config* c, const char* key_path, char* buf, size_t cap);
int vs_config_seq_get_double(const vs_config* c, const char* key_path, int index, double* v);
/* The sections of config.yaml into the flat parameter structs (struct_size must be set; start from
 * vs_*_params_default).  An absent section leaves *p alone and reports *present = 0 (the mains test
 * `!node.empty()`).  zero_missing = 1 is the reference to the letter: every key the mains read is assigned, an
 * absent one as 0 / "" (that is what `node["k"] >> field` does); 0 keeps the field's value for absent keys. */
int vs_config_read_stab(const vs_config* c, const char* section, int zero_missing, vs_params_c* p, int* present);
int vs_config_read_roll(const vs_config* c, const char* section, int zero_missing, vs_roll_params_c* p, int* present);
int vs_config_read_enh(const vs_config* c, const char* section, int zero_missing, vs_enh_params_c* p, int* present);
int vs_config_mtime(const char* path, int64_t* mtime);

#ifdef __cplusplus
}
#endif
#endif /* VS_STAB_H */
