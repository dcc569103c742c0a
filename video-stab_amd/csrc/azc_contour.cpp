// Host side of AutoZoomCrop (/root/reference/src/AutoZoomCrop.cpp:141-228): external contours of the content mask,
// the largest by point count, its filled interior, the shrink loop towards an inscribed rectangle and the aspect
// fix.  The reference runs this part on the CPU too (it downloads the mask for cv::findContours); it is product
// code, written independently of the oracle (bit planes and row spans where the oracle uses a labelled byte image
// and a filled image).  Plain C++: also built with sanitizers by scratch/fuzz/azc_fuzz.cpp.
#include "azc_contour.h"

#include <algorithm>
#include <climits>
#include <cstring>
#include <string>

#include "../../include/vs_stab.h"

namespace vsd {

void set_last_error(const std::string& msg);      // vs_api.cpp

// Follows the outer borders of the mask the way cv::findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
// finds them and keeps the one with the most points.  cv::findContours lists the contours last found first and
// AutoZoomCrop.cpp:155-164 keeps the first of equals in that list: of equals the one found LAST wins here.
// Border pixels already followed carry one of two marks (bit planes ML / MR beside the mask): MR where the
// border passed with the outside to its east, ML elsewhere.  Scanning a row, an unmarked mask pixel with
// background to its west starts a new outer border unless the last mark before it on the row is an ML (then the
// scan is inside a component already followed: a hole, or something within a hole).  Returns the number of
// contours.
static int trace_largest(const BitFrame& bf, CropScratch& S, std::vector<Box>* boxes = nullptr) {
    const int h = bf.h, P = bf.pitch;
    const uint64_t* F = bf.F;
    // the mark planes start from zero: only the words the last call marked need wiping
    const size_t n_words = (size_t)P * (h + 2);
    if (S.ml.size() != n_words) { S.ml.assign(n_words, 0); S.mr.assign(n_words, 0); }
    else for (size_t i : S.touched) S.ml[i] = S.mr[i] = 0;
    S.touched.clear();
    uint64_t* ML = S.ml.data();
    uint64_t* MR = S.mr.data();
    // 8 directions counter-clockwise from east, image y pointing down (twice, so that a turn needs no wrap)
    static const int DX[16] = {1, 1, 0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1};
    static const int DY[16] = {0, -1, -1, -1, 0, 1, 1, 1, 0, -1, -1, -1, 0, 1, 1, 1};
    S.best.clear(); S.chain_best.clear();
    int n_contours = 0;

    // A pixel as one number: pos = (y + 1) * 64 P + x + 64, so that pos >> 6 is its word and pos & 63 its bit, and a
    // step in direction d adds STEP[d].
    const long long P64 = (long long)P * 64;
    long long STEP[16];
    for (int d = 0; d < 16; d++) STEP[d] = DY[d] * P64 + DX[d];
    auto fg_at = [F](long long pos) { return (F[pos >> 6] >> (pos & 63)) & 1; };

    auto follow = [&](int sx, int sy) {
        S.cur.clear(); S.chain_cur.clear();
        auto mark_right = [&](long long pos) {
            const size_t i = (size_t)(pos >> 6); const uint64_t bit = 1ull << (pos & 63);
            MR[i] |= bit; ML[i] &= ~bit;
            S.touched.push_back(i);
        };
        auto mark_left_if_new = [&](long long pos) {
            const size_t i = (size_t)(pos >> 6); const uint64_t bit = 1ull << (pos & 63);
            if (!((ML[i] | MR[i]) & bit)) { ML[i] |= bit; S.touched.push_back(i); }
        };
        const long long start = (sy + 1) * P64 + sx + 64;
        // first neighbour clockwise from west
        int dir = 4;
        bool alone = true;
        for (int k = 0; k < 7; k++) {
            dir = (dir + 7) & 7;
            if (fg_at(start + STEP[dir])) { alone = false; break; }
        }
        if (alone) {
            mark_right(start);
            S.cur.push_back({sx, sy});
            S.chain_cur.push_back((sy << 16) | sx);
            return;
        }
        const long long second = start + STEP[dir];
        long long at = start;
        int ax = sx, ay = sy, last_dir = dir ^ 4;
        while (true) {
            const int from = dir;
            long long next;
            do { next = at + STEP[++dir]; } while (!fg_at(next));
            const bool passed_east = (unsigned)((dir & 7) - 1) < (unsigned)from;
            if (passed_east) mark_right(at);
            else mark_left_if_new(at);
            S.chain_cur.push_back((ay << 16) | ax);
            if ((dir & 7) != last_dir) { S.cur.push_back({ax, ay}); last_dir = dir & 7; }
            if (next == start && at == second) break;
            ax += DX[dir]; ay += DY[dir];
            at = next;
            dir = (dir + 4) & 7;
        }
    };

    const int wpr = P - 2;
    S.begins.assign((size_t)wpr + 4, 0);           // (+ 4: the scan below looks at four words at a time)
    uint64_t* B = S.begins.data();
    for (int y = 0; y < h; y++) {
        const size_t r = (size_t)(y + 1) * P + 1;
        // The mask pixels with background to their west, for the whole row first: one streaming pass over the mask words without
        // dependencies between them (the compiler vectorises it), and most rows of a dark picture end here.
        uint64_t any = 0;
        {
            const uint64_t* Fr = F + r;
            for (int k = 0; k < wpr; k++) { B[k] = Fr[k] & ~((Fr[k] << 1) | (Fr[k - 1] >> 63)); any |= B[k]; }
        }
        if (!any) continue;
        // kind of the last mark on this row before the scan position (0 none, 1 ML, 2 MR), brought up to date only
        // when a candidate asks for it: most rows have none, and then the mark planes are not read at all
        int last_mark = 0, marks_upto = 0;           // words [0, marks_upto) are accounted for in last_mark
        for (int k4 = 0; k4 < wpr; k4 += 4) {
            if (!(B[k4] | B[k4 + 1] | B[k4 + 2] | B[k4 + 3])) continue;
            for (int k = k4; k < k4 + 4; k++) {
                const uint64_t begins = B[k];
                if (begins == 0) continue;
                uint64_t cand = begins & ~(ML[r + k] | MR[r + k]);
                while (cand) {
                    for (; marks_upto < k; marks_upto++) {
                        const uint64_t ml = ML[r + marks_upto], mr = MR[r + marks_upto];
                        if (ml | mr) last_mark = ml > mr ? 1 : 2;     // the higher bit is the later pixel
                    }
                    const int b = __builtin_ctzll(cand);
                    const uint64_t below = (1ull << b) - 1;
                    const uint64_t ml = ML[r + k] & below, mr = MR[r + k] & below;
                    int kind = last_mark;
                    if (ml | mr) kind = ml > mr ? 1 : 2;
                    if (kind != 1) {
                        ++n_contours;
                        follow(k * 64 + b, y);
                        if (boxes) {
                            Box bx{INT_MAX, INT_MAX, INT_MIN, INT_MIN};
                            for (const int c : S.chain_cur) {
                                const int cx = c & 0xFFFF, cy = c >> 16;
                                bx.x0 = std::min(bx.x0, cx); bx.x1 = std::max(bx.x1, cx);
                                bx.y0 = std::min(bx.y0, cy); bx.y1 = std::max(bx.y1, cy);
                            }
                            boxes->push_back(bx);
                        }
                        if (S.cur.size() >= S.best.size()) { S.best.swap(S.cur); S.chain_best.swap(S.chain_cur); }
                    }
                    // the planes may have changed under the scan: look again at what is left of the word
                    const uint64_t above = b == 63 ? 0 : ~((2ull << b) - 1);
                    cand = begins & ~(ML[r + k] | MR[r + k]) & above;
                }
            }
        }
    }
    return n_contours;
}

// Bounding boxes of the outer borders, in the order of the vector cv::findContours(RETR_EXTERNAL) returns (every contour
// goes to the head of the list when it is finished: last found first).
void external_boxes(const BitFrame& bf, CropScratch& S, std::vector<Box>& out) {
    out.clear();
    trace_largest(bf, S, &out);
    std::reverse(out.begin(), out.end());
}

// 0 / non-zero bytes -> BitFrame planes in S.bits (the host-only entry point; the device hands over bits)
BitFrame pack_mask(const uint8_t* mask, int w, int h, size_t stride, CropScratch& S) {
    BitFrame bf;
    bf.w = w; bf.h = h; bf.pitch = BitFrame::pitch_for(w);
    S.bits.assign(BitFrame::words_for(w, h), 0);
    for (int y = 0; y < h; y++) {
        const uint8_t* m = mask + (size_t)y * stride;
        uint64_t* d = &S.bits[(size_t)(y + 1) * bf.pitch + 1];
        for (int x = 0; x < w; x++) d[x >> 6] |= (uint64_t)(m[x] != 0) << (x & 63);   // host-only entry: not a hot path
    }
    bf.F = S.bits.data();
    return bf;
}

// The filled contour (cv::drawContours FILLED) as row spans: interior by the even-odd rule on the border chain
// (each chain step that changes row is one crossing of the upper of its two rows) plus the border pixels
// themselves.  Nothing of the size of the image is written: the shrink loop below only asks how many pixels
// of a row / column segment are NOT filled.
static void fill_spans(int w, int h, CropScratch& S, std::vector<uint8_t>* dump) {
    typedef CropScratch::Span Span;
    S.rows.resize(h); S.spans.resize(h);
    for (int y = 0; y < h; y++) { S.rows[y].clear(); S.spans[y].clear(); }
    const size_t n = S.chain_best.size();
    int run_y = -1;
    Span* run = nullptr;                 // the span the chain is drawing on row run_y: a border running along a row is one span
    for (size_t i = 0; i < n; i++) {
        const int a = S.chain_best[i], b = S.chain_best[i + 1 < n ? i + 1 : 0];
        const int ay = a >> 16, ax = a & 0xFFFF, by = b >> 16, bx = b & 0xFFFF;
        if (ay == run_y && ax >= run->a - 1 && ax <= run->b + 1) {
            run->a = std::min(run->a, ax); run->b = std::max(run->b, ax);
        } else {
            S.spans[ay].push_back(Span{ax, ax});
            run = &S.spans[ay].back(); run_y = ay;
        }
        if (ay == by) continue;
        if (ay < by) S.rows[ay].push_back(ax); else S.rows[by].push_back(bx);
    }
    for (int y = 0; y < h; y++) {
        std::vector<int>& r = S.rows[y];
        std::vector<Span>& sp = S.spans[y];
        if (sp.empty()) continue;
        std::sort(r.begin(), r.end());
        for (size_t k = 0; k + 1 < r.size(); k += 2)
            if (r[k + 1] > r[k]) sp.push_back(Span{r[k], r[k + 1]});
        std::sort(sp.begin(), sp.end(), [](const Span& p, const Span& q) { return p.a < q.a; });
        size_t m = 0;
        for (size_t k = 1; k < sp.size(); k++) {
            if (sp[k].a <= sp[m].b + 1) sp[m].b = std::max(sp[m].b, sp[k].b);
            else sp[++m] = sp[k];
        }
        sp.resize(m + 1);
    }
    if (dump) {
        dump->assign((size_t)w * h, 0);
        for (int y = 0; y < h; y++)
            for (const Span& q : S.spans[y]) memset(&(*dump)[(size_t)y * w + q.a], 255, (size_t)(q.b - q.a + 1));
    }
}

// info = {n_contours, contour_points, x, y, w, h, iterations, valid}
void crop_from_mask(const BitFrame& bf, CropScratch& S, int32_t info[8], std::vector<uint8_t>* filled_dump) {
    const int w = bf.w, h = bf.h;
    for (int i = 0; i < 8; i++) info[i] = 0;
    info[0] = trace_largest(bf, S);
    if (info[0] == 0) return;
    info[1] = (int)S.best.size();
    fill_spans(w, h, S, filled_dump);
    S.sx.clear(); S.sy.clear();
    for (const P2& p : S.best) { S.sx.push_back(p.x); S.sy.push_back(p.y); }
    std::sort(S.sx.begin(), S.sx.end());
    std::sort(S.sy.begin(), S.sy.end());
    auto row_zeros = [&](int y, int xa, int xb) {   // zeros in row y, columns [xa, xb)
        int filled = 0;
        for (const CropScratch::Span& q : S.spans[y]) {
            const int lo = std::max(q.a, xa), hi = std::min(q.b, xb - 1);
            if (hi >= lo) filled += hi - lo + 1;
        }
        return (xb - xa) - filled;
    };
    // Column queries dominate the loop (two per round, over the height of the rectangle), so rows of one span, the
    // usual kind, are kept as two flat arrays the compiler can vectorise over; the others are fixed up one by one.
    S.one_a.assign(h, INT_MAX); S.one_b.assign(h, INT_MIN); S.multi.clear();
    for (int y = 0; y < h; y++) {
        const std::vector<CropScratch::Span>& sp = S.spans[y];
        if (sp.size() == 1) { S.one_a[y] = sp[0].a; S.one_b[y] = sp[0].b; }
        else if (sp.size() > 1) S.multi.push_back(y);
    }
    auto col_zeros = [&](int x, int ya, int yb) {   // zeros in column x, rows [ya, yb)
        const int* A = S.one_a.data();
        const int* B = S.one_b.data();
        int zeros = 0;
        for (int y = ya; y < yb; y++) zeros += (x < A[y]) | (x > B[y]);
        for (int y : S.multi) {
            if (y < ya || y >= yb) continue;
            for (const CropScratch::Span& q : S.spans[y]) if (q.a <= x && x <= q.b) { --zeros; break; }
        }
        return zeros;
    };
    // The rectangle only shrinks: while a side keeps its column, its count changes by the rows that left the range.
    struct ColCache { int x = -1, ya = 0, yb = 0, zeros = 0; } lcache, rcache;
    auto col_cached = [&](ColCache& c, int x, int ya, int yb) {
        if (c.x == x && ya >= c.ya && yb <= c.yb && ya <= yb) c.zeros -= col_zeros(x, c.ya, ya) + col_zeros(x, yb, c.yb);
        else c.zeros = col_zeros(x, ya, yb);
        c.x = x; c.ya = ya; c.yb = yb;
        return c.zeros;
    };
    size_t lo_x = 0, hi_x = S.sx.size() - 1, lo_y = 0, hi_y = S.sy.size() - 1;
    int bx = 0, by = 0, bw = 0, bh = 0, iters = 0;
    while (lo_x < hi_x && lo_y < hi_y) {
        bx = S.sx[lo_x]; by = S.sy[lo_y]; bw = S.sx[hi_x] - bx; bh = S.sy[hi_y] - by;
        ++iters;
        if (bw <= 0 || bh <= 0) break;              // degenerate rectangle: nothing to test
        const int top = row_zeros(by, bx, bx + bw), bottom = row_zeros(by + bh - 1, bx, bx + bw);
        const int left = col_cached(lcache, bx, by, by + bh), right = col_cached(rcache, bx + bw - 1, by, by + bh);
        if (!(top | bottom | left | right)) break;
        // which side gives way (AutoZoomCrop.cpp:57-77)
        bool mv_top = false, mv_bottom = false, mv_left = false, mv_right = false;
        if (top > bottom) mv_top = top > left && top > right;
        else mv_bottom = bottom > left && bottom > right;
        if (left >= right) mv_left = left >= bottom && left >= top;
        else mv_right = right >= top && right >= bottom;
        if (mv_left) ++lo_x;
        if (mv_right) --hi_x;
        if (mv_top) ++lo_y;
        if (mv_bottom) --hi_y;
    }
    const double ar = (double)w / h;
    const int new_w = (int)(bh * ar);
    const int centre = bx + bw / 2;
    bw = new_w;
    bx = centre - new_w / 2;
    if (bx < 0) bx = 0;
    if (bx + bw > w) bx = w - bw;
    const int x1 = std::max(bx, 0), y1 = std::max(by, 0), x2 = std::min(bx + bw, w), y2 = std::min(by + bh, h);
    info[6] = iters;
    if (x2 <= x1 || y2 <= y1) return;
    info[2] = x1; info[3] = y1; info[4] = x2 - x1; info[5] = y2 - y1; info[7] = 1;
}

}  // namespace vsd

using namespace vsd;

extern "C" {

// Host logic only (no device needed): :141-228 on a host mask.
int vs_azc_crop_from_mask(const uint8_t* mask, int w, int h, size_t stride, int32_t* info, uint8_t* filled_out) {
    if (!mask || !info || w <= 0 || h <= 0 || stride < (size_t)w) return VS_ERR_INVALID_ARG;
    if (w > 65535 || h > 32767) { set_last_error("auto zoom/crop: image too large"); return VS_ERR_INVALID_ARG; }
    static thread_local CropScratch S;       // kept between calls, as the vs_azc object keeps its own
    std::vector<uint8_t> filled;
    crop_from_mask(pack_mask(mask, w, h, stride, S), S, info, filled_out ? &filled : nullptr);
    if (filled_out) {
        if (filled.empty()) memset(filled_out, 0, (size_t)w * h);
        else memcpy(filled_out, filled.data(), (size_t)w * h);
    }
    return VS_OK;
}

// Host logic only: cv::boundingRect of the contours cv::findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) returns, in
// the order of that vector (the virtual canvas's region list, Stabilizer.cpp:2232-2241).
int vs_op_external_boxes(const uint8_t* mask, int w, int h, size_t stride, int32_t* xywh, int max_boxes, int32_t* n_boxes) {
    if (!mask || !xywh || !n_boxes || max_boxes < 0 || w <= 0 || h <= 0 || stride < (size_t)w) return VS_ERR_INVALID_ARG;
    if (w > 65535 || h > 32767) { set_last_error("external boxes: image too large"); return VS_ERR_INVALID_ARG; }
    static thread_local CropScratch S;
    static thread_local std::vector<Box> boxes;
    external_boxes(pack_mask(mask, w, h, stride, S), S, boxes);
    *n_boxes = (int32_t)boxes.size();
    for (int i = 0; i < (int)boxes.size() && i < max_boxes; i++) {
        xywh[4 * i] = boxes[i].x0; xywh[4 * i + 1] = boxes[i].y0;
        xywh[4 * i + 2] = boxes[i].x1 - boxes[i].x0 + 1; xywh[4 * i + 3] = boxes[i].y1 - boxes[i].y0 + 1;
    }
    return VS_OK;
}

}  // extern "C"
