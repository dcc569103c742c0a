// CPU ORACLE (test infrastructure) - the "virtual canvas" branch of vs::Stabilizer restated,
// /root/reference/src/Stabilizer.cpp:1130-1134 (call site), :2066-2443 (the functions).  PARITY UNPINNED (see vso.h).
//
// What the reference does there, as written:
//   * the frame that leaves the queue (NOT the warped one: the warp's result is overwritten, :1133) is pasted
//     into the middle of a black canvas of int(cols * scale) x int(rows * scale) pixels (:2169-2212);
//   * the "empty" parts of the canvas - gray <= 1, so dark picture content counts too - are found as the bounding
//     rectangles of the external contours of that mask (:2224-2241); for every rectangle of more than 100 pixels
//     the newest older frame of the temporal buffer that "covers" more than half of it (:2405-2427, canvas
//     coordinates compared with frame coordinates, as written) is motion-compensated with a REFLECT border, cut
//     to the rectangle, stretched to its size if it was clipped (:2320-2347) and blended in with a weight that
//     falls off towards the rectangle's edges (:2349-2403);
//   * the result is the cols x rows window of the canvas at the integer offset (centre - t) (:2115-2163).
// The canvas is rebuilt from zeros for every frame, so the only state is the temporal buffer, the scale chosen at
// the first call (adaptive: from the largest of the last 30 transforms, :2281-2314) and the canvas size.  None of
// it is reset by Stabilizer::clean() (:221-256).  canvasBlendMask_ (:2094-2106) is computed and never read, and
// is not restated.  Floating-point expressions keep the reference's types and order (float, no contraction).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <deque>

#include "vso_internal.h"

namespace vso {

namespace {

struct Rect {
    int x = 0, y = 0, w = 0, h = 0;
    int area() const { return w * h; }
};
Rect operator&(const Rect& a, const Rect& b) {          // cv::Rect_::operator&
    Rect r;
    r.x = std::max(a.x, b.x); r.y = std::max(a.y, b.y);
    r.w = std::min(a.x + a.w, b.x + b.w) - r.x;
    r.h = std::min(a.y + a.h, b.y + b.h) - r.y;
    if (r.w <= 0 || r.h <= 0) r = Rect();
    return r;
}

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;          // BGR, packed rows
    bool empty() const { return d.empty(); }
    size_t stride() const { return (size_t)w * 3; }
};

}  // namespace

struct CanvasState {
    std::deque<Image> temporalFrames;                   // temporalFrameBuffer_
    std::deque<std::array<float, 3>> temporalTransforms;    // temporalTransformBuffer_
    bool haveCanvas = false;                            // !virtualCanvas_.empty()
    int canvasCols = 0, canvasRows = 0;                 // virtualCanvas_.cols / rows
    float scale = 0.f;                                  // currentCanvasScale_
    bool scaleInit = false;
    int cw = 0, ch = 0;                                 // canvasSize_
    float cx = 0.f, cy = 0.f;                           // canvasCenter_
    int32_t info[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

CanvasState* canvas_new() { return new CanvasState(); }
void canvas_delete(CanvasState* c) { delete c; }
void canvas_info(const CanvasState* c, int32_t info[8]) { memcpy(info, c->info, sizeof c->info); }

// :2281-2314
static float optimal_canvas_scale(const vs_params_c& p, const std::vector<float>& transforms) {
    const int n = (int)(transforms.size() / 3);
    if (n == 0) return p.canvas_scale_factor;
    float maxMotion = 0.0f;
    const int recentFrames = std::min(30, n);
    for (int i = n - recentFrames; i < n; i++) {
        if (i < 0) continue;
        const float mx = transforms[3 * i], my = transforms[3 * i + 1];
        const float magnitude = std::sqrt(mx * mx + my * my);
        maxMotion = std::max(maxMotion, magnitude);
    }
    const float motionFactor = std::max(1.0f, maxMotion / 50.0f);
    float optimalScale = p.canvas_scale_factor + (motionFactor - 1.0f) * 0.5f;
    optimalScale = std::max(p.min_canvas_scale, std::min(p.max_canvas_scale, optimalScale));
    return optimalScale;
}

// :2169-2212
static void create_canvas(const CanvasState& c, const uint8_t* frame, int w, int h, size_t stride, Image& canvas) {
    canvas.w = c.cw; canvas.h = c.ch;
    canvas.d.assign((size_t)c.cw * c.ch * 3, 0);
    const float fx = c.cx - w / 2.0f, fy = c.cy - h / 2.0f;
    const Rect frameRect{static_cast<int>(fx), static_cast<int>(fy), w, h};
    const Rect validRect = frameRect & Rect{0, 0, c.cw, c.ch};
    if (validRect.w > 0 && validRect.h > 0) {
        const Rect srcRect{validRect.x - frameRect.x, validRect.y - frameRect.y, validRect.w, validRect.h};
        if (srcRect.x >= 0 && srcRect.y >= 0 && srcRect.x + srcRect.w <= w && srcRect.y + srcRect.h <= h)
            for (int y = 0; y < validRect.h; y++)
                memcpy(&canvas.d[((size_t)(validRect.y + y) * c.cw + validRect.x) * 3],
                       frame + (size_t)(srcRect.y + y) * stride + (size_t)srcRect.x * 3, (size_t)validRect.w * 3);
    }
}

// :2405-2427
static bool region_available(const Rect& region, const float rel[3], const Image& frame) {
    const Rect moved{region.x + static_cast<int>(rel[0]), region.y + static_cast<int>(rel[1]), region.w, region.h};
    const Rect inter = moved & Rect{0, 0, frame.w, frame.h};
    const float coverage = static_cast<float>(inter.area()) / static_cast<float>(region.area());
    return coverage > 0.5f;
}

// :2316-2347 with applyMotionCompensation :2429-2443
static Image extract_temporal_region(const Image& frame, const Rect& region, const float rel[3]) {
    const float dx = -rel[0], dy = -rel[1], da = -rel[2];
    const float M[6] = {libm_cosf(da), -libm_sinf(da), dx, libm_sinf(da), libm_cosf(da), dy};
    double Md[6];
    for (int i = 0; i < 6; i++) Md[i] = (double)M[i];
    Image comp;
    comp.w = frame.w; comp.h = frame.h; comp.d.resize(frame.d.size());
    warp_affine_d(frame.d.data(), frame.w, frame.h, frame.stride(), 3, comp.d.data(), frame.w, frame.h, frame.stride(), Md,
                  VS_BORDER_REFLECT, g_threads);
    const Rect moved{region.x + static_cast<int>(rel[0]), region.y + static_cast<int>(rel[1]), region.w, region.h};
    const Rect valid = moved & Rect{0, 0, frame.w, frame.h};
    Image out;
    if (valid.w <= 0 || valid.h <= 0) return out;
    Image cut;
    cut.w = valid.w; cut.h = valid.h; cut.d.resize((size_t)valid.w * valid.h * 3);
    for (int y = 0; y < valid.h; y++)
        memcpy(&cut.d[(size_t)y * valid.w * 3], &comp.d[((size_t)(valid.y + y) * frame.w + valid.x) * 3], (size_t)valid.w * 3);
    if (cut.w == region.w && cut.h == region.h) return cut;
    out.w = region.w; out.h = region.h; out.d.resize((size_t)region.w * region.h * 3);
    resize_linear_u8(cut.d.data(), cut.w, cut.h, cut.stride(), 3, out.d.data(), out.w, out.h, out.stride());
    return out;
}

// :2349-2403 (the region is a bounding rectangle inside the canvas and the source has its size: neither the clip of
// the region nor the second resize does anything)
static void seamless_blend(Image& target, const Image& source, const Rect& region, float weight, int edge_blend_radius) {
    const int edgeRadius = std::min(edge_blend_radius, std::min(region.w, region.h) / 4);
    for (int y = 0; y < region.h; y++)
        for (int x = 0; x < region.w; x++) {
            float alpha = 1.0f * weight;
            const float distFromEdge = (float)std::min(std::min(x, y), std::min(region.w - x - 1, region.h - y - 1));
            if (distFromEdge < edgeRadius) {
                const float edgeWeight = distFromEdge / edgeRadius;
                alpha *= edgeWeight;
            }
            uint8_t* t = &target.d[((size_t)(region.y + y) * target.w + region.x + x) * 3];
            const uint8_t* s = &source.d[((size_t)y * source.w + x) * 3];
            for (int c = 0; c < 3; c++) t[c] = static_cast<uint8_t>((1.0f - alpha) * t[c] + alpha * s[c]);
        }
}

// :2214-2279
static void blend_temporal_regions(CanvasState& c, const vs_params_c& p, Image& result, const float t[3]) {
    c.info[3] = c.info[4] = 0; c.info[5] = -1;
    const size_t nbuf = c.temporalFrames.size();
    if (nbuf < 2) return;
    std::vector<uint8_t> gray((size_t)result.w * result.h);
    bgr2gray(result.d.data(), result.w, result.h, result.stride(), gray.data(), result.w);
    for (uint8_t& g : gray) g = g > 1 ? 0 : 255;                        // THRESH_BINARY_INV, thresh 1
    std::vector<std::vector<Pt>> contours;
    find_contours_external(gray.data(), result.w, result.h, result.w, contours);
    std::vector<Rect> emptyRegions;
    for (const auto& contour : contours) {
        int x0 = INT_MAX, y0 = INT_MAX, x1 = INT_MIN, y1 = INT_MIN;     // cv::boundingRect of integer points
        for (const Pt& q : contour) { x0 = std::min(x0, q.x); y0 = std::min(y0, q.y); x1 = std::max(x1, q.x); y1 = std::max(y1, q.y); }
        const Rect br{x0, y0, x1 - x0 + 1, y1 - y0 + 1};
        if (br.area() > 100) emptyRegions.push_back(br);
    }
    c.info[3] = (int)emptyRegions.size();
    for (const Rect& region : emptyRegions) {
        int best = -1;
        float bestWeight = 0.0f;
        float bestRel[3] = {0, 0, 0};
        for (size_t i = 0; i < nbuf - 1; i++) {
            const float rel[3] = {t[0] - c.temporalTransforms[i][0], t[1] - c.temporalTransforms[i][1], t[2] - c.temporalTransforms[i][2]};
            if (!region_available(region, rel, c.temporalFrames[i])) continue;
            // (an available region always yields a non-empty cut: the same intersection decides both)
            float temporalWeight = static_cast<float>(i + 1) / nbuf;
            temporalWeight *= p.canvas_blend_weight;
            if (temporalWeight > bestWeight) { best = (int)i; bestWeight = temporalWeight; memcpy(bestRel, rel, sizeof rel); }
        }
        if (best >= 0 && bestWeight > 0.0f) {
            const Image fill = extract_temporal_region(c.temporalFrames[best], region, bestRel);
            seamless_blend(result, fill, region, bestWeight, p.edge_blend_radius);
            c.info[4]++; c.info[5] = best;
        }
    }
}

void canvas_apply(CanvasState* cs, const vs_params_c& p, const uint8_t* frame, int w, int h, size_t stride, const float t[3],
                  const std::vector<float>& transforms, uint8_t* out, size_t out_stride) {
    CanvasState& c = *cs;
    if (!c.scaleInit) { c.scale = p.canvas_scale_factor; c.scaleInit = true; }      // :203
    // updateTemporalFrameBuffer :2151-2167
    Image f;
    f.w = w; f.h = h; f.d.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++) memcpy(&f.d[(size_t)y * w * 3], frame + (size_t)y * stride, (size_t)w * 3);
    c.temporalFrames.push_back(std::move(f));
    c.temporalTransforms.push_back({t[0], t[1], t[2]});
    while (c.temporalFrames.size() > static_cast<size_t>(p.temporal_buffer_size)) {
        c.temporalFrames.pop_front();
        c.temporalTransforms.pop_front();
    }
    // applyVirtualCanvasStabilization :2066-2149
    if (!c.haveCanvas || c.canvasCols != static_cast<int>(w * c.scale) || c.canvasRows != static_cast<int>(h * c.scale)) {
        c.scale = (p.adaptive_canvas_size && !transforms.empty()) ? optimal_canvas_scale(p, transforms) : p.canvas_scale_factor;
        c.cw = static_cast<int>(w * c.scale); c.ch = static_cast<int>(h * c.scale);
        c.cx = c.cw / 2.0f; c.cy = c.ch / 2.0f;
    }
    Image canvas;
    create_canvas(c, frame, w, h, stride, canvas);
    c.haveCanvas = !canvas.d.empty(); c.canvasCols = c.cw; c.canvasRows = c.ch;
    blend_temporal_regions(c, p, canvas, t);
    const float ox = c.cx - w / 2.0f - t[0], oy = c.cy - h / 2.0f - t[1];
    Rect ex{std::max(0, static_cast<int>(ox)), std::max(0, static_cast<int>(oy)), w, h};
    ex.x = std::min(ex.x, canvas.w - ex.w);
    ex.y = std::min(ex.y, canvas.h - ex.h);
    ex.w = std::min(ex.w, canvas.w - ex.x);
    ex.h = std::min(ex.h, canvas.h - ex.y);
    c.info[0] = c.cw; c.info[1] = c.ch; memcpy(&c.info[2], &c.scale, 4); c.info[6] = ex.x; c.info[7] = ex.y;
    if (ex.w > 0 && ex.h > 0 && ex.x >= 0 && ex.y >= 0 && ex.x + ex.w <= canvas.w && ex.y + ex.h <= canvas.h) {
        // (the window has the frame's size whenever it passes this test: the LANCZOS4 resize of :2141-2143 never runs)
        for (int y = 0; y < h; y++)
            memcpy(out + (size_t)y * out_stride, &canvas.d[((size_t)(ex.y + y) * canvas.w + ex.x) * 3], (size_t)w * 3);
        return;
    }
    for (int y = 0; y < h; y++) memcpy(out + (size_t)y * out_stride, frame + (size_t)y * stride, (size_t)w * 3);   // :2148
}

}  // namespace vso
