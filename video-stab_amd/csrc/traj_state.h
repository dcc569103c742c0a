// Device-resident trajectory state of one stream (vs::Stabilizer members
// transforms_, path_, drone-mode state; Stabilizer.h:311-371,418-429).
#ifndef VS_TRAJ_STATE_H
#define VS_TRAJ_STATE_H

#include <stdint.h>

namespace vsd {

constexpr int TRAJ_RING = 256;   // >= 35 (queue) + 50 (drone box radius) + 20, power of two
constexpr int GAUSS_MAX = 63;

struct TrajState {
    int n;                          // transforms_.size() == path_.size()
    float last_path[3];
    float transforms[TRAJ_RING][3];
    float path[TRAJ_RING][3];
    int smoothing_radius;           // params_.smoothingRadius (mutated by adaptSmoothingRadius)
    // incremental Kalman (x0,x1,P00,P01,P10,P11) per component
    float kal[3][6];
    int kal_n[3];
    float kal_last[3];
    // drone mode
    float hfMedian[2];
    float hfRotLP;
    int hfInDeadZone;
    int hfFreezeCounter;
    float hfAccum;
    int hfHistN;
    float hfHist[10][2];
};

struct TrajParams {
    int method;                     // vs_smoothing
    int horizon_lock;
    int drone;
    int adaptive, min_radius, max_radius;
    float hf_shake_px, hf_rot_lp_alpha, hf_dead_zone, hf_decay;
    int hf_freeze_duration;
    int gauss_ksize;
    float gauss_kernel[GAUSS_MAX];
};

}  // namespace vsd
#endif
