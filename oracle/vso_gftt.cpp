// CPU ORACLE (test infrastructure) - cv::goodFeaturesToTrack restated
// (reference call sites: src/Stabilizer.cpp:355-357 and :740-744).
//
// Definition choices where OpenCV's bits depend on the build's SIMD/FMA:
//   Dx = fl(fl(fl(r0+r2)*f1) + fl(r1*f0)),  r_k = S[y+k-1][x+1]-S[y+k-1][x-1]
//   Dy = fl(t2 - t0), t_k = fl(fl(c*f0) + fl((a+b)*f1)) on row y+k-1
//   with f1 = (float)(1/(4*blockSize*255)), f0 = 2*f1, no FMA contraction;
//   box sums in double (exact for these magnitudes, hence order-free);
//   lambda_min = fl(fl(a+c) - sqrtf(fl(fl((a-c)*(a-c)) + fl(b*b)))).
// Borders: BORDER_REFLECT_101 (cornerMinEigenVal's BORDER_DEFAULT).
#include "vso_internal.h"

#include <algorithm>

namespace vso {

static void min_eigen(const uint8_t* g, int w, int h, size_t stride, int bs, float* eig) {
    double scale = (double)(1 << 2) * bs * 255.0;
    scale = 1.0 / scale;
    const float f1 = (float)scale, f0 = 2.f * f1;
    auto R = [&](int p, int len) { return border_interpolate(p, len, VS_BORDER_REFLECT_101); };
    auto px = [&](int y, int x) -> int { return g[(size_t)R(y, h) * stride + R(x, w)]; };
    std::vector<float> cov((size_t)w * h * 3);
    // horizontal intermediates per source row (reflect applies to the source image)
    std::vector<float> rdx((size_t)w * h), tdy((size_t)w * h);
    parallel_rows(h, [&](int ya, int yb) {
    for (int y = ya; y < yb; y++)
        for (int x = 0; x < w; x++) {
            int a = px(y, x - 1), c = px(y, x), b = px(y, x + 1);
            rdx[(size_t)y * w + x] = (float)(b - a);
            tdy[(size_t)y * w + x] = (float)c * f0 + (float)(a + b) * f1;
        }
    });
    parallel_rows(h, [&](int ya, int yb) {
    for (int y = ya; y < yb; y++) {
        int y0 = R(y - 1, h), y2 = R(y + 1, h);
        for (int x = 0; x < w; x++) {
            float r0 = rdx[(size_t)y0 * w + x], r1 = rdx[(size_t)y * w + x], r2 = rdx[(size_t)y2 * w + x];
            float dx = (r0 + r2) * f1 + r1 * f0;
            float dy = tdy[(size_t)y2 * w + x] - tdy[(size_t)y0 * w + x];
            float* cv = &cov[((size_t)y * w + x) * 3];
            cv[0] = dx * dx;
            cv[1] = dx * dy;
            cv[2] = dy * dy;
        }
    }
    });
    int anchor = bs / 2;
    parallel_rows(h, [&](int ya, int yb) {
    for (int y = ya; y < yb; y++)
        for (int x = 0; x < w; x++) {
            double s[3] = {0, 0, 0};
            for (int j = 0; j < bs; j++) {
                int yy = R(y - anchor + j, h);
                double rs[3] = {0, 0, 0};
                for (int i = 0; i < bs; i++) {
                    int xx = R(x - anchor + i, w);
                    const float* cv = &cov[((size_t)yy * w + xx) * 3];
                    rs[0] += (double)cv[0];
                    rs[1] += (double)cv[1];
                    rs[2] += (double)cv[2];
                }
                s[0] += rs[0]; s[1] += rs[1]; s[2] += rs[2];
            }
            float a = (float)s[0] * 0.5f, b = (float)s[1], c = (float)s[2] * 0.5f;
            float d = (a - c) * (a - c) + b * b;
            eig[(size_t)y * w + x] = (a + c) - std::sqrt(d);
        }
    });
}

int gftt(const uint8_t* gray, int w, int h, size_t stride, int max_corners, double quality,
         double min_distance, int block_size, std::vector<float>& pts, int* n_candidates) {
    pts.clear();
    if (n_candidates) *n_candidates = 0;
    if (w < 3 || h < 3) return 0;
    std::vector<float> eig((size_t)w * h);
    min_eigen(gray, w, h, stride, block_size, eig.data());
    float mx = eig[0];
    for (size_t i = 1; i < eig.size(); i++) mx = std::max(mx, eig[i]);
    double maxVal = (double)mx;
    float thr = (float)(maxVal * quality);
    for (auto& v : eig) v = v > thr ? v : 0.f;  // THRESH_TOZERO
    // dilate 3x3 (border pixels ignored = -inf) and collect local maxima
    std::vector<int> cand;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            float val = eig[(size_t)y * w + x];
            if (val == 0) continue;
            float m = val;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) m = std::max(m, eig[(size_t)(y + j) * w + x + i]);
            if (val == m) cand.push_back(y * w + x);
        }
    if (n_candidates) *n_candidates = (int)cand.size();
    if (cand.empty()) return 0;
    // greaterThanPtr: value descending, ties -> higher address first
    std::sort(cand.begin(), cand.end(), [&](int a, int b) {
        float va = eig[a], vb = eig[b];
        return va > vb ? true : va < vb ? false : a > b;
    });
    int ncorners = 0;
    if (min_distance >= 1) {
        const int cell = cv_round(min_distance);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<std::vector<float>> grid((size_t)gw * gh);
        float md2 = (float)(min_distance * min_distance);
        // OpenCV squares the double and compares the float dx*dx+dy*dy against it
        double md2d = min_distance * min_distance;
        (void)md2;
        for (int idx : cand) {
            int y = idx / w, x = idx - y * w;
            bool good = true;
            int xc = x / cell, yc = y / cell;
            int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1);
            int x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++) {
                    auto& m = grid[(size_t)yy * gw + xx];
                    for (size_t j = 0; j < m.size(); j += 2) {
                        float dx = x - m[j], dy = y - m[j + 1];
                        if ((double)(dx * dx + dy * dy) < md2d) { good = false; break; }
                    }
                }
            if (good) {
                auto& m = grid[(size_t)yc * gw + xc];
                m.push_back((float)x);
                m.push_back((float)y);
                pts.push_back((float)x);
                pts.push_back((float)y);
                ++ncorners;
                if (max_corners > 0 && ncorners == max_corners) break;
            }
        }
    } else {
        for (int idx : cand) {
            int y = idx / w, x = idx - y * w;
            pts.push_back((float)x);
            pts.push_back((float)y);
            ++ncorners;
            if (max_corners > 0 && ncorners == max_corners) break;
        }
    }
    return ncorners;
}

}  // namespace vso

extern "C" {
void vso_min_eigen(const uint8_t* gray, int w, int h, size_t stride, int block_size, float* eig) {
    vso::min_eigen(gray, w, h, stride, block_size, eig);
}
int vso_gftt(const uint8_t* gray, int w, int h, size_t stride, int max_corners, double quality,
             double min_distance, int block_size, float* out_pts, int* n_candidates) {
    std::vector<float> pts;
    int n = vso::gftt(gray, w, h, stride, max_corners, quality, min_distance, block_size, pts,
                      n_candidates);
    for (size_t i = 0; i < pts.size(); i++) out_pts[i] = pts[i];
    return n;
}
}
