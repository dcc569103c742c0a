// Virtual canvas (vs::Stabilizer::applyVirtualCanvasStabilization and helpers, /root/reference/src/Stabilizer.cpp:1130-1134,
// 2066-2443): k_canvas.hip.
#ifndef VS_CANVAS_H
#define VS_CANVAS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vs_stab.h"
#include "traj_state.h"

namespace vsd {

struct Canvas;

Canvas* canvas_new();
void canvas_delete(Canvas* c);      // synchronizes the device before it frees

// One output of a BGR8 stream.  d_frame: the frame that leaves the queue (w x h, row pitch `pitch`); d_t: its correction
// (dx, dy, da), three floats the trajectory kernel wrote earlier on `st`; d_traj: the trajectory state (read once, when the
// canvas scale is chosen).  The result (w x h) goes to d_out.  The call waits for `st` once (the window offset and the
// choice of the temporal frame are host decisions on device values, as in the reference); the output itself is written
// asynchronously on `st`.
int canvas_apply(Canvas* c, const vs_params_c& p, const uint8_t* d_frame, size_t pitch, int w, int h, const float* d_t,
                 const TrajState* d_traj, uint8_t* d_out, size_t out_stride, hipStream_t st);

// {canvas w, h, scale (float bits), regions, regions filled, temporal index of the last fill, window x, y}
void canvas_info(const Canvas* c, int32_t info[8]);

}  // namespace vsd
#endif
