// Host side of AutoZoomCrop (AutoZoomCrop.cpp:141-228): the contour logic the reference runs on the CPU after
// downloading the mask.  Plain C++ (no device code): k_azc.hip feeds it the mask as a BitFrame.
#ifndef VS_AZC_CONTOUR_H
#define VS_AZC_CONTOUR_H

#include <cstddef>
#include <cstdint>
#include <vector>

namespace vsd {

struct P2 { int x, y; };
struct Box { int x0, y0, x1, y1; };     // inclusive corners

// Scratch kept between frames so that the per-frame host work allocates nothing.
struct CropScratch {
    std::vector<uint64_t> bits;       // the mask as a BitFrame when it arrives as bytes
    std::vector<uint64_t> ml, mr;     // marks on followed border pixels (same layout)
    std::vector<size_t> touched;      // words of ml / mr that hold marks
    std::vector<uint64_t> begins;     // one row's mask pixels with background to their west
    std::vector<P2> best, cur;        // SIMPLE points of the largest / current contour
    std::vector<int> chain_best, chain_cur;   // every border pixel (y << 16 | x) of the same
    struct Span { int a, b; };                // filled pixels a..b (inclusive) of one row
    std::vector<std::vector<Span>> spans;     // the filled contour, row by row (merged, ascending)
    std::vector<int> sx, sy;
    std::vector<int> one_a, one_b, multi;     // rows of exactly one span: its ends; rows of several: their numbers
    std::vector<std::vector<int>> rows;       // crossings per row
};

// The mask as the host sees it: one bit per pixel, 64 pixels per word, inside a frame of zero words (one word left
// and right of every row, one row above and below), so that a neighbour test never leaves the buffer.  The device
// writes it in this form (close5_bits_kernel): 1 MB instead of 8 MB over PCIe at 4K, and the scan for contour
// starts looks at 64 pixels per step.
struct BitFrame {
    int w = 0, h = 0, pitch = 0;      // pitch in words = ceil(w / 64) + 2
    const uint64_t* F = nullptr;
    static int pitch_for(int w) { return (w + 63) / 64 + 2; }
    static size_t words_for(int w, int h) { return (size_t)pitch_for(w) * (h + 2); }
};

// 0 / non-zero bytes -> BitFrame planes in S.bits (the host-only entry point; the device hands over bits)
BitFrame pack_mask(const uint8_t* mask, int w, int h, size_t stride, CropScratch& S);

// info = {n_contours, contour_points, x, y, w, h, iterations, valid}; filled_dump (optional) receives the filled
// largest contour as w*h bytes.  w <= 65535, h <= 32767.
void crop_from_mask(const BitFrame& bf, CropScratch& S, int32_t info[8], std::vector<uint8_t>* filled_dump);

// cv::boundingRect of every contour cv::findContours(mask, RETR_EXTERNAL, ...) returns, in the order of that vector
// (virtual canvas, Stabilizer.cpp:2232-2241).
void external_boxes(const BitFrame& bf, CropScratch& S, std::vector<Box>& out);

}  // namespace vsd

#endif
