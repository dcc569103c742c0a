// CPU ORACLE (test infrastructure) - vs::AutoZoomCrop::autoZoomCrop restated,
// /root/reference/src/AutoZoomCrop.cpp:10-283.
//
// The reference runs gray/threshold/morphology/warp on cv::cuda and findContours /
// drawContours on the CPU.  As for the other stages the parity target is the CPU OpenCV
// 4.11 definition of each primitive (the CPU-only ancestor spare/"AutoZoomCrop copy.cpp"
// :124-215 uses exactly those):
//   cvtColor BGR2GRAY, threshold(>1, BINARY), morphologyEx(MORPH_CLOSE, 5x5 MORPH_ELLIPSE,
//   default border = outside pixels ignored), findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
//   [Suzuki-Abe border following as in cvFindNextContour / icvFetchContour, restated from the
//   published algorithm], drawContours(FILLED), warpAffine(INTER_LINEAR, BORDER_CONSTANT).
// Dead code of the reference (the inverse mask of :116-125, never read) is not restated.
// PARITY UNPINNED (see vso.h).
#include <algorithm>
#include <cstring>

#include "vso_internal.h"

namespace vso {

// getStructuringElement(MORPH_ELLIPSE, Size(5,5)): rows 1..3 full, rows 0 and 4 centre only
static const uint8_t ELLIPSE5[5][5] = {
    {0, 0, 1, 0, 0}, {1, 1, 1, 1, 1}, {1, 1, 1, 1, 1}, {1, 1, 1, 1, 1}, {0, 0, 1, 0, 0}};

static void morph5(const uint8_t* src, int w, int h, uint8_t* dst, bool dilate) {
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = dilate ? 0 : 255;
            for (int ky = -2; ky <= 2; ky++)
                for (int kx = -2; kx <= 2; kx++) {
                    if (!ELLIPSE5[ky + 2][kx + 2]) continue;
                    const int yy = y + ky, xx = x + kx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;   // morphologyDefaultBorderValue
                    const int p = src[(size_t)yy * w + xx];
                    v = dilate ? std::max(v, p) : std::min(v, p);
                }
            dst[(size_t)y * w + x] = (uint8_t)v;
        }
}

// AutoZoomCrop.cpp:111-139 (content mask branch): gray -> (>1 ? 255 : 0) -> close
void content_mask(const uint8_t* src, int w, int h, size_t stride, int cn, uint8_t* mask) {
    std::vector<uint8_t> gray((size_t)w * h), t((size_t)w * h);
    if (cn == 3) bgr2gray(src, w, h, stride, gray.data(), w);
    else for (int y = 0; y < h; y++) memcpy(&gray[(size_t)y * w], src + (size_t)y * stride, w);
    for (size_t i = 0; i < gray.size(); i++) gray[i] = gray[i] > 1 ? 255 : 0;
    morph5(gray.data(), w, h, t.data(), true);
    morph5(t.data(), w, h, mask, false);
}

// cv::findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
void find_contours_external(const uint8_t* mask, int w, int h, size_t stride, std::vector<std::vector<Pt>>& out) {
    out.clear();
    const int W = w + 2, H = h + 2;
    std::vector<int8_t> img((size_t)W * H, 0);          // binary image inside a zero frame
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * W + x + 1] = mask[(size_t)y * stride + x] ? 1 : 0;
    // chain code deltas, counter-clockwise from "right" (y down): 0:E 1:NE 2:N 3:NW 4:W 5:SW 6:S 7:SE
    const int dxs[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dys[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int delta[16];
    for (int i = 0; i < 16; i++) delta[i] = dys[i & 7] * W + dxs[i & 7];
    const int8_t NBD = 2, NBD_RIGHT = (int8_t)(2 | -128);
    for (int y = 1; y <= h; y++) {
        int8_t* row = &img[(size_t)y * W];
        int prev = 0, lnbd_x = 0;
        for (int x = 1; x <= w; x++) {
            int p = row[x];
            if (p == prev) continue;
            bool trace = false;
            if (prev == 0 && p == 1) {
                trace = !(row[lnbd_x] > 0);            // RETR_EXTERNAL: not inside an already traced border
            }                                           // hole borders (p==0 && prev>=1) are never followed
            if (trace) {
                std::vector<Pt> c;
                const int i0 = y * W + x;
                int s_end = 4, s = 4, i1 = 0;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + delta[s];
                } while (img[i1] == 0 && s != s_end);
                if (s == s_end) {                       // isolated pixel
                    img[i0] = NBD_RIGHT;
                    c.push_back({x - 1, y - 1});
                } else {
                    int i3 = i0, prev_s = s ^ 4;
                    Pt pt{x - 1, y - 1};
                    for (;;) {
                        s_end = s;
                        int i4;
                        for (;;) {
                            i4 = i3 + delta[++s];
                            if (img[i4] != 0) break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) img[i3] = NBD_RIGHT;     // right neighbour examined and 0
                        else if (img[i3] == 1) img[i3] = NBD;
                        if (s != prev_s) { c.push_back(pt); prev_s = s; }                // CHAIN_APPROX_SIMPLE
                        pt.x += dxs[s]; pt.y += dys[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                out.push_back(std::move(c));
                prev = row[x];
                lnbd_x = x;
            } else {
                prev = p;
                if (prev & -2) lnbd_x = x;
            }
        }
    }
    // cvInsertNodeIntoTree() puts every finished contour at the head of its parent's child list: the vector
    // cv::findContours returns lists the contours last found first (bottom of the image to top)
    std::reverse(out.begin(), out.end());
}

// cv::drawContours(mask, {c}, 0, 255, FILLED): even-odd scanline fill of the polygon through the
// points plus its outline.  All edges of a CHAIN_APPROX_SIMPLE contour are horizontal, vertical
// or exact diagonals, so every crossing is an integer and no sub-pixel rule is involved.
void fill_contour(const std::vector<Pt>& c, int w, int h, uint8_t* mask) {
    memset(mask, 0, (size_t)w * h);
    const int n = (int)c.size();
    if (n == 0) return;
    int ymin = c[0].y, ymax = c[0].y;
    for (const Pt& p : c) { ymin = std::min(ymin, p.y); ymax = std::max(ymax, p.y); }
    std::vector<int> xs;
    for (int y = ymin; y <= ymax; y++) {
        xs.clear();
        for (int i = 0; i < n; i++) {
            const Pt a = c[i], b = c[(i + 1) % n];
            if (a.y == b.y) continue;
            const int lo = std::min(a.y, b.y), hi = std::max(a.y, b.y);
            if (y < lo || y >= hi) continue;
            xs.push_back(a.x + (y - a.y) * (b.x - a.x) / (b.y - a.y));
        }
        std::sort(xs.begin(), xs.end());
        for (size_t k = 0; k + 1 < xs.size(); k += 2)
            for (int x = xs[k]; x <= xs[k + 1]; x++) mask[(size_t)y * w + x] = 255;
    }
    for (int i = 0; i < n; i++) {                       // outline
        Pt a = c[i];
        const Pt b = c[(i + 1) % n];
        const int sx = (b.x > a.x) - (b.x < a.x), sy = (b.y > a.y) - (b.y < a.y);
        for (;;) {
            mask[(size_t)a.y * w + a.x] = 255;
            if (a.x == b.x && a.y == b.y) break;
            a.x += sx; a.y += sy;
        }
    }
    (void)h;
}

// checkInteriorExterior, AutoZoomCrop.cpp:10-80.  Returns true when no border pixel of the
// rectangle is outside the mask.  Degenerate rectangles (zero width or height), for which the
// reference reads outside its sub-matrix, are reported as finished.
static bool check_interior_exterior(const uint8_t* mask, int mw, int rx, int ry, int rw, int rh, int& top,
                                    int& bottom, int& left, int& right) {
    if (rw <= 0 || rh <= 0) return true;
    bool ok = true;
    unsigned cTop = 0, cBottom = 0, cLeft = 0, cRight = 0;
    auto at = [&](int y, int x) { return mask[(size_t)(ry + y) * mw + rx + x]; };
    for (int x = 0; x < rw; x++) if (at(0, x) == 0) { ok = false; ++cTop; }
    for (int x = 0; x < rw; x++) if (at(rh - 1, x) == 0) { ok = false; ++cBottom; }
    for (int y = 0; y < rh; y++) if (at(y, 0) == 0) { ok = false; ++cLeft; }
    for (int y = 0; y < rh; y++) if (at(y, rw - 1) == 0) { ok = false; ++cRight; }
    if (cTop > cBottom) { if (cTop > cLeft && cTop > cRight) top = 1; }
    else if (cBottom > cLeft && cBottom > cRight) bottom = 1;
    if (cLeft >= cRight) { if (cLeft >= cBottom && cLeft >= cTop) left = 1; }
    else if (cRight >= cTop && cRight >= cBottom) right = 1;
    return ok;
}

// AutoZoomCrop.cpp:141-228: contours -> largest by point count -> interior rectangle -> aspect fix
// info: {n_contours, contour_points, x, y, w, h, iterations, valid}
void azc_crop_rect(const uint8_t* cmask, int w, int h, int32_t info[8]) {
    for (int i = 0; i < 8; i++) info[i] = 0;
    std::vector<std::vector<Pt>> contours;
    find_contours_external(cmask, w, h, w, contours);
    info[0] = (int)contours.size();
    if (contours.empty()) return;                                             // :149-152
    size_t maxSize = 0, id = 0;
    for (size_t i = 0; i < contours.size(); i++)                               // :155-164
        if (contours[i].size() > maxSize) { maxSize = contours[i].size(); id = i; }
    info[1] = (int)maxSize;
    std::vector<uint8_t> filled((size_t)w * h);
    fill_contour(contours[id], w, h, filled.data());                          // :167-168
    std::vector<int> sx, sy;
    for (const Pt& p : contours[id]) { sx.push_back(p.x); sy.push_back(p.y); }
    std::sort(sx.begin(), sx.end());                                           // :171-175 (only .x / .y are read)
    std::sort(sy.begin(), sy.end());
    unsigned minX = 0, maxX = (unsigned)sx.size() - 1, minY = 0, maxY = (unsigned)sy.size() - 1;
    int bx = 0, by = 0, bw = 0, bh = 0, iters = 0;
    const double ar = (double)w / h;                                           // :186
    while (minX < maxX && minY < maxY) {                                       // :189-205
        bx = sx[minX]; by = sy[minY]; bw = sx[maxX] - bx; bh = sy[maxY] - by;
        ++iters;
        int t = 0, b = 0, l = 0, r = 0;
        if (check_interior_exterior(filled.data(), w, bx, by, bw, bh, t, b, l, r)) break;
        if (l) ++minX;
        if (r) --maxX;
        if (t) ++minY;
        if (b) --maxY;
    }
    const int newW = (int)(bh * ar);                                           // :208
    const int cx = bx + bw / 2;                                                // :211-213
    bw = newW;
    bx = cx - newW / 2;
    if (bx < 0) bx = 0;                                                        // :216-218
    if (bx + bw > w) bx = w - bw;
    // cv::Rect &= image (:221, :226)
    const int x1 = std::max(bx, 0), y1 = std::max(by, 0);
    const int x2 = std::min(bx + bw, w), y2 = std::min(by + bh, h);
    info[6] = iters;
    if (x2 - x1 <= 0 || y2 - y1 <= 0) return;
    info[2] = x1; info[3] = y1; info[4] = x2 - x1; info[5] = y2 - y1; info[7] = 1;
}

}  // namespace vso

using namespace vso;

extern "C" {

void vso_content_mask(const uint8_t* src, int w, int h, size_t stride, int cn, uint8_t* mask) {
    content_mask(src, w, h, stride, cn, mask);
}

// flattened contours: counts[i] points each, xy pairs consecutively.  Returns the number of contours,
// or -1 when a capacity is too small.
int vso_find_contours(const uint8_t* mask, int w, int h, size_t stride, int32_t* counts, int max_contours, int32_t* xy,
                      int max_points) {
    std::vector<std::vector<Pt>> cs;
    find_contours_external(mask, w, h, stride, cs);
    if ((int)cs.size() > max_contours) return -1;
    int k = 0;
    for (size_t i = 0; i < cs.size(); i++) {
        counts[i] = (int)cs[i].size();
        for (const Pt& p : cs[i]) {
            if (k >= max_points) return -1;
            xy[2 * k] = p.x; xy[2 * k + 1] = p.y; k++;
        }
    }
    return (int)cs.size();
}

void vso_fill_contour(const int32_t* xy, int n, int w, int h, uint8_t* mask) {
    std::vector<Pt> c(n);
    for (int i = 0; i < n; i++) c[i] = {xy[2 * i], xy[2 * i + 1]};
    fill_contour(c, w, h, mask);
}

void vso_azc_crop_rect(const uint8_t* content_mask, int w, int h, int32_t* info) { azc_crop_rect(content_mask, w, h, info); }

// autoZoomCrop, AutoZoomCrop.cpp:102-283.  out must hold max(w*h, 640*360)*cn bytes; the result is
// 640x360 (or the unchanged input on the fallback paths :149-152, :238-249); returns 1 when cropped.
int vso_azc_apply(const uint8_t* src, int w, int h, size_t stride, int cn, uint8_t* out, int32_t* out_w, int32_t* out_h,
                  int32_t* info) {
    int32_t inf[8];
    std::vector<uint8_t> cm((size_t)w * h);
    content_mask(src, w, h, stride, cn, cm.data());
    azc_crop_rect(cm.data(), w, h, inf);
    if (info) memcpy(info, inf, sizeof inf);
    if (!inf[7]) {
        for (int y = 0; y < h; y++) memcpy(out + (size_t)y * w * cn, src + (size_t)y * stride, (size_t)w * cn);
        *out_w = w; *out_h = h;
        return 0;
    }
    const double scaleX = 640.0 / inf[4], scaleY = 360.0 / inf[5];               // :251-252
    const double M[6] = {(double)(float)scaleX, 0, 0, 0, (double)(float)scaleY, 0};   // :261-262 (CV_32F matrix)
    warp_affine_d(src + (size_t)inf[3] * stride + (size_t)inf[2] * cn, inf[4], inf[5], stride, cn, out, 640, 360,
                  (size_t)640 * cn, M, VS_BORDER_BLACK, g_threads);                 // :270
    *out_w = 640; *out_h = 360;
    return 1;
}

// autoZoomCrop on an NV12 surface (luma plane of w x h, interleaved chroma plane of w/2 x h/2 pairs `uv_offset` bytes behind it,
// one row pitch).  The reference has no NV12 path (its cvtColor(BGR2GRAY) throws on one channel): DEFINED as the BGR operator's
// geometry applied per plane - content mask from the luma plane (gray > 1, the threshold of :121-127 on a picture that is gray
// already), the crop rectangle (x, y, w, h) for the luma plane and (x/2, y/2, max(1, w/2), max(1, h/2)) for the chroma plane,
// each plane scaled to its share of 640 x 360 (320 x 180 pairs) by the reference's scale matrix (:251-270, a CV_32F matrix).
// out: luma rows of out_stride bytes, the chroma plane at out_uv_offset.  Returns 1 when cropped (else the surface is copied).
int vso_azc_apply_nv12(const uint8_t* src, int w, int h, size_t stride, size_t uv_offset, uint8_t* out, size_t out_stride, size_t out_uv_offset,
                       int32_t* out_w, int32_t* out_h, int32_t* info) {
    int32_t inf[8];
    std::vector<uint8_t> cm((size_t)w * h);
    content_mask(src, w, h, stride, 1, cm.data());
    azc_crop_rect(cm.data(), w, h, inf);
    if (info) memcpy(info, inf, sizeof inf);
    if (!inf[7]) {
        for (int y = 0; y < h; y++) memcpy(out + (size_t)y * out_stride, src + (size_t)y * stride, (size_t)w);
        for (int y = 0; y < h / 2; y++) memcpy(out + out_uv_offset + (size_t)y * out_stride, src + uv_offset + (size_t)y * stride, (size_t)w);
        *out_w = w; *out_h = h;
        return 0;
    }
    const int ux = inf[2] / 2, uy = inf[3] / 2, uw = std::max(1, inf[4] / 2), uh = std::max(1, inf[5] / 2);
    const double My[6] = {(double)(float)(640.0 / inf[4]), 0, 0, 0, (double)(float)(360.0 / inf[5]), 0};
    const double Mu[6] = {(double)(float)(320.0 / uw), 0, 0, 0, (double)(float)(180.0 / uh), 0};
    warp_affine_d(src + (size_t)inf[3] * stride + (size_t)inf[2], inf[4], inf[5], stride, 1, out, 640, 360, out_stride, My, VS_BORDER_BLACK, g_threads);
    warp_affine_d(src + uv_offset + (size_t)uy * stride + (size_t)ux * 2, uw, uh, stride, 2, out + out_uv_offset, 320, 180, out_stride, Mu,
                  VS_BORDER_BLACK, g_threads);
    *out_w = 640; *out_h = 360;
    return 1;
}

}  // extern "C"
