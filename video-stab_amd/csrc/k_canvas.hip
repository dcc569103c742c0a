// Virtual canvas: vs::Stabilizer::applyVirtualCanvasStabilization() and its helpers
// (/root/reference/src/Stabilizer.cpp:1130-1134 call site, :2066-2443).
//
// What the reference computes there (every step on the CPU, whole-image cv::Mat temporaries):
//   * the frame that leaves the queue - not the warped one, whose result is overwritten (:1133) - pasted into the
//     middle of a black canvas of int(cols * scale) x int(rows * scale) pixels (:2169-2212);
//   * "empty" canvas regions = bounding rectangles (area > 100) of the external contours of gray <= 1 (:2224-2241); each
//     is filled from the newest older frame of the temporal buffer that covers more than half of it (:2405-2427):
//     that frame motion-compensated (warpAffine, REFLECT border), cut to the rectangle, stretched back to its size
//     when the cut was clipped (:2316-2347), blended with a weight that falls off towards the rectangle's edges
//     (:2349-2403);
//   * the output = the cols x rows window of the canvas at the integer offset centre - (dx, dy) (:2115-2149).
//
// Here: the canvas is never materialised unless a region is filled (the default scale 1.5 never fills: the one region
// is the whole canvas, and a frame covers 1/2.25 of it).  Without a fill the output is two 2-D copies.  With fills, one
// kernel per region evaluates compensation warp -> cut -> stretch -> blend per canvas pixel, with no intermediate
// image.  When the frame lies strictly inside the canvas the black ring around it is the only external contour and the
// region list is known without looking at the pixels; otherwise the gray <= 1 mask is built on the device as a bit
// plane (64 pixels per word) and its outer borders are followed on the host (azc_contour.cpp), as AutoZoomCrop does.
// The choice of the temporal frame and all rectangles are host decisions on the correction (dx, dy, da), which the
// trajectory kernel produces on the device: the call waits for the stream once per frame to read those 12 bytes.
#include <algorithm>
#include <climits>
#include <cstring>
#include <deque>
#include <new>
#include <vector>

#include "azc_contour.h"
#include "canvas.h"
#include "vs_common.h"

namespace vsd {

namespace {

typedef unsigned long long u64;

struct Rect {
    int x = 0, y = 0, w = 0, h = 0;
    int area() const { return w * h; }
};
Rect inter(const Rect& a, const Rect& b) {              // cv::Rect_::operator&
    Rect r;
    r.x = std::max(a.x, b.x); r.y = std::max(a.y, b.y);
    r.w = std::min(a.x + a.w, b.x + b.w) - r.x;
    r.h = std::min(a.y + a.h, b.y + b.h) - r.y;
    if (r.w <= 0 || r.h <= 0) r = Rect();
    return r;
}

// ---- device ------------------------------------------------------------------------------------------

// gray = (B*3735 + G*19235 + R*9798 + 2^14) >> 15; THRESH_BINARY_INV at 1 keeps gray <= 1
__device__ __forceinline__ unsigned empty_bgr(uint32_t b, uint32_t g, uint32_t r) {
    return b * 3735u + g * 19235u + r * 9798u + (1u << 14) < (2u << 15) ? 1u : 0u;
}

// The gray <= 1 mask of the canvas as a zero-framed BitFrame: a thread = one word (64 pixels of a canvas row); pixels
// outside the pasted frame are black, hence set.  fr = the frame's rectangle on the canvas, (ox, oy) = the frame pixel
// at its corner.
__global__ __launch_bounds__(256) void canvas_bits_kernel(const uint8_t* __restrict__ frame, size_t pitch, int fx, int fy, int fw, int fh,
                                                          int ox, int oy, int cw, int ch, u64* __restrict__ F, int fpitch) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int wpr = (cw + 63) / 64;
    if (k >= wpr || y >= ch) return;
    u64 bits = 0;
    const bool row_in = y >= fy && y < fy + fh;
    const uint8_t* row = frame + (size_t)(y - fy + oy) * pitch;
    for (int i = 0; i < 64; i++) {
        const int x = k * 64 + i;
        if (x >= cw) break;
        unsigned e = 1u;
        if (row_in && x >= fx && x < fx + fw) {
            const uint8_t* q = row + (size_t)(x - fx + ox) * 3;
            e = empty_bgr(q[0], q[1], q[2]);
        }
        bits |= (u64)e << i;
    }
    F[(size_t)(y + 1) * fpitch + 1 + k] = bits;
}

__device__ __forceinline__ int reflect(int p, int len) {        // cv::borderInterpolate(BORDER_REFLECT)
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p - 1;
        else p = len - 1 - (p - len);
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

struct FillArgs {
    uint8_t* canvas; size_t cpitch;
    int rx, ry, rw, rh;                  // the region on the canvas
    const uint8_t* src; size_t spitch; int sw, sh;      // the temporal frame
    double Minv[6];                      // inverse of the compensation matrix (cv::warpAffine inverts on the host)
    int vx, vy, vw, vh;                  // the cut of the compensated frame
    int stretch;                         // the cut is smaller than the region: cv::resize(INTER_LINEAR) back to it
    double scale_x, scale_y;             // cut size / region size, as cv::resize computes them
    float weight;
    int edge_radius;
};

// One pixel (3 channels) of cv::warpAffine(src, M, src.size(), INTER_LINEAR, BORDER_REFLECT) at (X, Y): the fixed-point
// coordinates of imgwarp.cpp (AB_BITS 10, INTER_BITS 5) and the 15-bit bilinear table, whose weights are the exact
// products (32 - fx)(32 - fy) * 32 (the saturated (0, 0) entry gives the same byte); taps through borderInterpolate.
__device__ __forceinline__ void warp_px(const FillArgs& a, int X, int Y, int out[3]) {
    const int adelta = d_round(a.Minv[0] * X * 1024), bdelta = d_round(a.Minv[3] * X * 1024);
    const int X0 = d_round((a.Minv[1] * Y + a.Minv[2]) * 1024) + 16, Y0 = d_round((a.Minv[4] * Y + a.Minv[5]) * 1024) + 16;
    const int Xf = (X0 + adelta) >> 5, Yf = (Y0 + bdelta) >> 5;
    const int sx = sat_s16(Xf >> 5), sy = sat_s16(Yf >> 5), fx = Xf & 31, fy = Yf & 31;
    const int x0 = reflect(sx, a.sw), x1 = reflect(sx + 1, a.sw), y0 = reflect(sy, a.sh), y1 = reflect(sy + 1, a.sh);
    const uint8_t* r0 = a.src + (size_t)y0 * a.spitch;
    const uint8_t* r1 = a.src + (size_t)y1 * a.spitch;
    const int w00 = (32 - fx) * (32 - fy), w01 = fx * (32 - fy), w10 = (32 - fx) * fy, w11 = fx * fy;
    for (int c = 0; c < 3; c++)
        out[c] = (r0[x0 * 3 + c] * w00 + r0[x1 * 3 + c] * w01 + r1[x0 * 3 + c] * w10 + r1[x1 * 3 + c] * w11 + 512) >> 10;
}

// A thread = one pixel of the region.
__global__ __launch_bounds__(256) void canvas_fill_kernel(FillArgs a) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= a.rw || y >= a.rh) return;
    int s[3];
    if (!a.stretch) {
        warp_px(a, a.vx + x, a.vy + y, s);
    } else {
        // cv::resize, INTER_LINEAR, 8-bit: HResizeLinear with 11-bit coefficients, VResizeLinear's
        // ((b0 * (S0 >> 4)) >> 16 + (b1 * (S1 >> 4)) >> 16 + 2) >> 2
        float fx = (float)((x + 0.5) * a.scale_x - 0.5);
        int sx = f_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        const bool last = sx + 1 >= a.vw;              // dx >= xmax: one tap, scaled
        if (sx >= a.vw - 1) { fx = 0.f; sx = a.vw - 1; }
        const int a0 = sat_s16(f_round((1.f - fx) * 2048.f)), a1 = sat_s16(f_round(fx * 2048.f));
        float fy = (float)((y + 0.5) * a.scale_y - 0.5);
        const int sy = f_floor(fy);
        fy -= sy;
        const int b0 = sat_s16(f_round((1.f - fy) * 2048.f)), b1 = sat_s16(f_round(fy * 2048.f));
        const int y0 = sy < 0 ? 0 : (sy < a.vh ? sy : a.vh - 1), y1 = sy + 1 < 0 ? 0 : (sy + 1 < a.vh ? sy + 1 : a.vh - 1);
        int p00[3], p01[3] = {0, 0, 0}, p10[3], p11[3] = {0, 0, 0};
        warp_px(a, a.vx + sx, a.vy + y0, p00);
        warp_px(a, a.vx + sx, a.vy + y1, p10);
        if (!last) {
            warp_px(a, a.vx + sx + 1, a.vy + y0, p01);
            warp_px(a, a.vx + sx + 1, a.vy + y1, p11);
        }
        for (int c = 0; c < 3; c++) {
            const int S0 = last ? p00[c] * 2048 : p00[c] * a0 + p01[c] * a1;
            const int S1 = last ? p10[c] * 2048 : p10[c] * a0 + p11[c] * a1;
            s[c] = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        }
    }
    // seamlessBlend (:2366-2401): alpha = weight * (distance to the region's edge / edgeRadius) inside the edge band
    float alpha = 1.0f * a.weight;
    const float dist = (float)min(min(x, y), min(a.rw - x - 1, a.rh - y - 1));
    if (dist < (float)a.edge_radius) alpha *= dist / (float)a.edge_radius;
    uint8_t* t = a.canvas + (size_t)(a.ry + y) * a.cpitch + (size_t)(a.rx + x) * 3;
    for (int c = 0; c < 3; c++) t[c] = (uint8_t)((1.0f - alpha) * (float)t[c] + alpha * (float)s[c]);
}

// max over the last min(30, n) transforms of sqrt(dx^2 + dy^2) (:2289-2300); out = {max, n as float bits}
__global__ void canvas_motion_kernel(const TrajState* s, float* out) {
    if (threadIdx.x != 0) return;
    const int n = s->n;
    float mx = 0.0f;
    const int recent = n < 30 ? n : 30;
    for (int i = n - recent; i < n; i++) {
        if (i < 0) continue;
        const float a = s->transforms[i & (TRAJ_RING - 1)][0], b = s->transforms[i & (TRAJ_RING - 1)][1];
        mx = fmaxf(mx, sqrtf(a * a + b * b));
    }
    out[0] = mx;
    out[1] = __int_as_float(n);
}

}  // namespace

// ---- host --------------------------------------------------------------------------------------------

struct Canvas {
    struct Slot { uint8_t* d = nullptr; size_t cap = 0; int w = 0, h = 0; float t[3] = {0, 0, 0}; };
    std::vector<Slot> slots;             // the temporal buffer's device frames (packed rows)
    std::deque<int> order;               // temporalFrameBuffer_ as slot numbers, oldest first
    std::vector<int> free_slots;
    bool have_canvas = false;            // !virtualCanvas_.empty()
    int canvas_cols = 0, canvas_rows = 0;
    float scale = 0.f;                   // currentCanvasScale_
    bool scale_init = false;
    int cw = 0, ch = 0;                  // canvasSize_
    float cx = 0.f, cy = 0.f;            // canvasCenter_
    uint8_t* d_canvas = nullptr; size_t canvas_cap = 0;
    u64* d_bits = nullptr; size_t bits_cap = 0; int bits_w = 0, bits_h = 0;
    u64* h_bits = nullptr; size_t h_bits_cap = 0;
    float* d_motion = nullptr;
    float* h_pin = nullptr;              // pinned: [0..2] the correction, [4..5] canvas_motion_kernel's result
    CropScratch scratch;
    std::vector<Box> boxes;
    int32_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

Canvas* canvas_new() { return new (std::nothrow) Canvas(); }

void canvas_delete(Canvas* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    for (Canvas::Slot& s : c->slots) if (s.d) (void)hipFree(s.d);
    if (c->d_canvas) (void)hipFree(c->d_canvas);
    if (c->d_bits) (void)hipFree(c->d_bits);
    if (c->h_bits) (void)hipHostFree(c->h_bits);
    if (c->d_motion) (void)hipFree(c->d_motion);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    delete c;
}

void canvas_info(const Canvas* c, int32_t info[8]) {
    if (c) memcpy(info, c->info, sizeof c->info);
    else memset(info, 0, 8 * sizeof(int32_t));
}

int canvas_apply(Canvas* c, const vs_params_c& p, const uint8_t* d_frame, size_t pitch, int w, int h, const float* d_t,
                 const TrajState* d_traj, uint8_t* d_out, size_t out_stride, hipStream_t st) {
    if (!c || !d_frame || !d_t || !d_out || w < 1 || h < 1) { set_last_error("canvas: invalid argument"); return VS_ERR_INVALID_ARG; }
    const size_t row = (size_t)w * 3;
    if (!c->h_pin) VS_HIP_TRY(hipHostMalloc((void**)&c->h_pin, 8 * sizeof(float)));
    if (!c->d_motion) VS_HIP_TRY(hipMalloc((void**)&c->d_motion, 2 * sizeof(float)));
    if (!c->scale_init) { c->scale = p.canvas_scale_factor; c->scale_init = true; }          // :203

    // updateTemporalFrameBuffer (:2151-2167): the frame joins the buffer before it is looked at
    const int keep = p.temporal_buffer_size;
    int cur_slot = -1;
    if (keep > 0) {
        if ((int)c->order.size() >= keep) {                       // the oldest leaves; nothing on the stream still reads it
            c->free_slots.push_back(c->order.front());
            c->order.pop_front();
        }
        if (c->free_slots.empty()) { c->slots.emplace_back(); c->free_slots.push_back((int)c->slots.size() - 1); }
        cur_slot = c->free_slots.back();
        c->free_slots.pop_back();
        Canvas::Slot& s = c->slots[cur_slot];
        const size_t need = row * h;
        if (s.cap < need) {
            if (s.d) { VS_HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(s.d); s.d = nullptr; s.cap = 0; }
            VS_HIP_TRY(hipMalloc((void**)&s.d, need));
            s.cap = need;
        }
        s.w = w; s.h = h;
        VS_HIP_TRY(hipMemcpy2DAsync(s.d, row, d_frame, pitch, row, h, hipMemcpyDeviceToDevice, st));
        c->order.push_back(cur_slot);
    }
    const bool reinit = !c->have_canvas || c->canvas_cols != static_cast<int>(w * c->scale) || c->canvas_rows != static_cast<int>(h * c->scale);
    if (reinit && p.adaptive_canvas_size) {
        hipLaunchKernelGGL(canvas_motion_kernel, dim3(1), dim3(64), 0, st, d_traj, c->d_motion);
        VS_HIP_TRY(hipGetLastError());
        VS_HIP_TRY(hipMemcpyAsync(c->h_pin + 4, c->d_motion, 2 * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    VS_HIP_TRY(hipMemcpyAsync(c->h_pin, d_t, 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    VS_HIP_TRY(hipStreamSynchronize(st));
    const float t[3] = {c->h_pin[0], c->h_pin[1], c->h_pin[2]};
    if (cur_slot >= 0) memcpy(c->slots[cur_slot].t, t, sizeof t);

    // applyVirtualCanvasStabilization (:2066-2149)
    if (reinit) {
        int n_tr = 0;
        memcpy(&n_tr, &c->h_pin[5], sizeof n_tr);
        if (p.adaptive_canvas_size && n_tr > 0) {                 // calculateOptimalCanvasSize :2281-2314
            const float maxMotion = c->h_pin[4];
            const float motionFactor = std::max(1.0f, maxMotion / 50.0f);
            float optimalScale = p.canvas_scale_factor + (motionFactor - 1.0f) * 0.5f;
            optimalScale = std::max(p.min_canvas_scale, std::min(p.max_canvas_scale, optimalScale));
            c->scale = optimalScale;
        } else {
            c->scale = p.canvas_scale_factor;
        }
        const float fw = w * c->scale, fh = h * c->scale;
        if (!(fw >= 1.0f && fw <= 65535.0f && fh >= 1.0f && fh <= 32767.0f)) {
            set_last_error("virtual canvas: canvas size out of range (1..65535 x 1..32767)");
            return VS_ERR_UNSUPPORTED;
        }
        c->cw = static_cast<int>(fw); c->ch = static_cast<int>(fh);
        c->cx = c->cw / 2.0f; c->cy = c->ch / 2.0f;
    }
    c->have_canvas = true; c->canvas_cols = c->cw; c->canvas_rows = c->ch;
    const int cw = c->cw, ch = c->ch;
    // createVirtualCanvas (:2169-2212)
    const Rect frameRect{static_cast<int>(c->cx - w / 2.0f), static_cast<int>(c->cy - h / 2.0f), w, h};
    const Rect validRect = inter(frameRect, Rect{0, 0, cw, ch});
    const Rect srcRect{validRect.x - frameRect.x, validRect.y - frameRect.y, validRect.w, validRect.h};

    // blendTemporalRegions (:2214-2279)
    struct Fill { Rect region; int slot; float rel[3]; float weight; };
    std::vector<Fill> fills;
    c->info[3] = c->info[4] = 0; c->info[5] = -1;
    const size_t nbuf = c->order.size();
    if (nbuf >= 2) {
        c->boxes.clear();
        const bool ringed = validRect.w == w && validRect.h == h && validRect.x > 0 && validRect.y > 0 &&
                            validRect.x + w < cw && validRect.y + h < ch;
        if (ringed) {
            // the black ring around the frame is one 8-connected component that encloses everything else
            c->boxes.push_back(Box{0, 0, cw - 1, ch - 1});
        } else {
            const int fpitch = BitFrame::pitch_for(cw);
            const size_t words = BitFrame::words_for(cw, ch);
            if (c->bits_cap < words) {
                if (c->d_bits) { (void)hipFree(c->d_bits); c->d_bits = nullptr; c->bits_cap = 0; }
                VS_HIP_TRY(hipMalloc((void**)&c->d_bits, words * sizeof(u64)));
                c->bits_cap = words; c->bits_w = 0;
            }
            if (c->h_bits_cap < words) {
                if (c->h_bits) { (void)hipHostFree(c->h_bits); c->h_bits = nullptr; c->h_bits_cap = 0; }
                VS_HIP_TRY(hipHostMalloc((void**)&c->h_bits, words * sizeof(u64)));
                c->h_bits_cap = words;
            }
            if (c->bits_w != cw || c->bits_h != ch) {              // the zero frame around the plane, once per geometry
                VS_HIP_TRY(hipMemsetAsync(c->d_bits, 0, words * sizeof(u64), st));
                c->bits_w = cw; c->bits_h = ch;
            }
            const int wpr = (cw + 63) / 64;
            hipLaunchKernelGGL(canvas_bits_kernel, dim3((wpr + 255) / 256, ch), dim3(256), 0, st, d_frame, pitch, validRect.x, validRect.y,
                               validRect.w, validRect.h, srcRect.x, srcRect.y, cw, ch, c->d_bits, fpitch);
            VS_HIP_TRY(hipGetLastError());
            VS_HIP_TRY(hipMemcpyAsync(c->h_bits, c->d_bits, words * sizeof(u64), hipMemcpyDeviceToHost, st));
            VS_HIP_TRY(hipStreamSynchronize(st));
            BitFrame bf;
            bf.w = cw; bf.h = ch; bf.pitch = fpitch; bf.F = (const uint64_t*)c->h_bits;
            external_boxes(bf, c->scratch, c->boxes);
        }
        for (const Box& b : c->boxes) {
            const Rect region{b.x0, b.y0, b.x1 - b.x0 + 1, b.y1 - b.y0 + 1};
            if (!(region.area() > 100)) continue;
            c->info[3]++;
            int best = -1;
            float bestWeight = 0.0f, bestRel[3] = {0, 0, 0};
            for (size_t i = 0; i + 1 < nbuf; i++) {
                const Canvas::Slot& s = c->slots[c->order[i]];
                const float rel[3] = {t[0] - s.t[0], t[1] - s.t[1], t[2] - s.t[2]};
                // isRegionAvailable :2405-2427
                const Rect moved{region.x + static_cast<int>(rel[0]), region.y + static_cast<int>(rel[1]), region.w, region.h};
                const Rect in = inter(moved, Rect{0, 0, s.w, s.h});
                const float coverage = static_cast<float>(in.area()) / static_cast<float>(region.area());
                if (!(coverage > 0.5f)) continue;
                float temporalWeight = static_cast<float>(i + 1) / nbuf;
                temporalWeight *= p.canvas_blend_weight;
                if (temporalWeight > bestWeight) { best = (int)i; bestWeight = temporalWeight; memcpy(bestRel, rel, sizeof rel); }
            }
            if (best >= 0 && bestWeight > 0.0f) {
                Fill f;
                f.region = region; f.slot = c->order[best]; memcpy(f.rel, bestRel, sizeof bestRel); f.weight = bestWeight;
                fills.push_back(f);
                c->info[4]++; c->info[5] = best;
            }
        }
    }

    // the output window (:2115-2149)
    const float ox = c->cx - w / 2.0f - t[0], oy = c->cy - h / 2.0f - t[1];
    Rect ex{std::max(0, static_cast<int>(ox)), std::max(0, static_cast<int>(oy)), w, h};
    ex.x = std::min(ex.x, cw - ex.w);
    ex.y = std::min(ex.y, ch - ex.h);
    ex.w = std::min(ex.w, cw - ex.x);
    ex.h = std::min(ex.h, ch - ex.y);
    c->info[0] = cw; c->info[1] = ch; memcpy(&c->info[2], &c->scale, 4); c->info[6] = ex.x; c->info[7] = ex.y;
    const bool window_ok = ex.w > 0 && ex.h > 0 && ex.x >= 0 && ex.y >= 0 && ex.x + ex.w <= cw && ex.y + ex.h <= ch;
    if (!window_ok) {                                               // :2148 the frame as it came
        VS_HIP_TRY(hipMemcpy2DAsync(d_out, out_stride, d_frame, pitch, row, h, hipMemcpyDeviceToDevice, st));
        return VS_OK;
    }
    const bool pasted = validRect.w > 0 && validRect.h > 0 && srcRect.x >= 0 && srcRect.y >= 0 && srcRect.x + srcRect.w <= w &&
                        srcRect.y + srcRect.h <= h;
    if (fills.empty()) {
        // no canvas needed: black, then the part of the pasted frame the window sees
        const Rect seen = pasted ? inter(ex, validRect) : Rect();
        if (seen.w != w || seen.h != h) VS_HIP_TRY(hipMemset2DAsync(d_out, out_stride, 0, row, h, st));
        if (seen.w > 0 && seen.h > 0)
            VS_HIP_TRY(hipMemcpy2DAsync(d_out + (size_t)(seen.y - ex.y) * out_stride + (size_t)(seen.x - ex.x) * 3, out_stride,
                                        d_frame + (size_t)(seen.y - validRect.y + srcRect.y) * pitch + (size_t)(seen.x - validRect.x + srcRect.x) * 3,
                                        pitch, (size_t)seen.w * 3, seen.h, hipMemcpyDeviceToDevice, st));
        return VS_OK;
    }
    const size_t cpitch = (size_t)cw * 3, cbytes = cpitch * ch;
    if (c->canvas_cap < cbytes) {
        if (c->d_canvas) { (void)hipFree(c->d_canvas); c->d_canvas = nullptr; c->canvas_cap = 0; }
        VS_HIP_TRY(hipMalloc((void**)&c->d_canvas, cbytes));
        c->canvas_cap = cbytes;
    }
    VS_HIP_TRY(hipMemsetAsync(c->d_canvas, 0, cbytes, st));
    if (pasted)
        VS_HIP_TRY(hipMemcpy2DAsync(c->d_canvas + (size_t)validRect.y * cpitch + (size_t)validRect.x * 3, cpitch,
                                    d_frame + (size_t)srcRect.y * pitch + (size_t)srcRect.x * 3, pitch, (size_t)validRect.w * 3, validRect.h,
                                    hipMemcpyDeviceToDevice, st));
    for (const Fill& f : fills) {
        const Canvas::Slot& s = c->slots[f.slot];
        FillArgs a;
        a.canvas = c->d_canvas; a.cpitch = cpitch;
        a.rx = f.region.x; a.ry = f.region.y; a.rw = f.region.w; a.rh = f.region.h;
        a.src = s.d; a.spitch = (size_t)s.w * 3; a.sw = s.w; a.sh = s.h;
        // applyMotionCompensation :2429-2443
        const float dx = -f.rel[0], dy = -f.rel[1], da = -f.rel[2];
        const float M[6] = {std::cos(da), -std::sin(da), dx, std::sin(da), std::cos(da), dy};
        warp_invert(M, a.Minv);
        // extractTemporalRegion :2316-2347
        const Rect moved{f.region.x + static_cast<int>(f.rel[0]), f.region.y + static_cast<int>(f.rel[1]), f.region.w, f.region.h};
        const Rect valid = inter(moved, Rect{0, 0, s.w, s.h});
        a.vx = valid.x; a.vy = valid.y; a.vw = valid.w; a.vh = valid.h;
        a.stretch = (valid.w != f.region.w || valid.h != f.region.h) ? 1 : 0;
        const double inv_scale_x = (double)f.region.w / valid.w, inv_scale_y = (double)f.region.h / valid.h;
        a.scale_x = 1. / inv_scale_x; a.scale_y = 1. / inv_scale_y;
        a.weight = f.weight;
        a.edge_radius = std::min(p.edge_blend_radius, std::min(f.region.w, f.region.h) / 4);
        hipLaunchKernelGGL(canvas_fill_kernel, dim3((a.rw + 255) / 256, a.rh), dim3(256), 0, st, a);
        VS_HIP_TRY(hipGetLastError());
    }
    VS_HIP_TRY(hipMemcpy2DAsync(d_out, out_stride, c->d_canvas + (size_t)ex.y * cpitch + (size_t)ex.x * 3, cpitch, row, h,
                                hipMemcpyDeviceToDevice, st));
    return VS_OK;
}

}  // namespace vsd
