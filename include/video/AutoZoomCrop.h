// vs::AutoZoomCrop for MI355X - source-compatible with the reference's
// include/video/AutoZoomCrop.h:7-17.  The content mask and the final crop+scale run on the
// GPU, the contour logic on the host (include/vs_stab.h, vs_azc_*).
#ifndef VIDEO_AUTO_ZOOM_CROP_HPP
#define VIDEO_AUTO_ZOOM_CROP_HPP

#include <opencv2/opencv.hpp>

namespace vs {

class AutoZoomCrop {
public:
    /// Crops away the black corners left by a rotation and scales the result to 640x360
    /// (the reference's fixed output size, AutoZoomCrop.cpp:246-261).  marginPercent is
    /// accepted and ignored, as in the reference (AutoZoomCrop.cpp:102).
    static cv::Mat autoZoomCrop(const cv::Mat& corrected, double marginPercent = 0.05);
};

}  // namespace vs

#endif
