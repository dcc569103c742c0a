// vs::RollCorrection for MI355X - source-compatible with the reference's
// include/video/RollCorrection.h:12-50 (same Parameters fields and defaults, same static
// entry point), as used by examples/roll-correction-file.cpp and examples/vs.cpp.
// Canny, HoughLines and the rotation run on the GPU through include/vs_stab.h (vs_roll_*).
#ifndef VIDEO_ROLL_CORRECTION_HPP
#define VIDEO_ROLL_CORRECTION_HPP

#include <opencv2/opencv.hpp>

namespace vs {

class RollCorrection {
public:
    struct Parameters {
        double scaleFactor = 0.25;            ///< analysis downscale (0..1]

        double cannyThresholdLow = 50.0;
        double cannyThresholdHigh = 150.0;
        int cannyAperture = 3;                ///< only 3 is supported

        float houghRho = 1.0f;
        float houghTheta = static_cast<float>(CV_PI / 180.0f);
        int houghThreshold = 100;

        double angleFilterMin = -10.0;        ///< degrees around horizontal
        double angleFilterMax = 10.0;

        double angleSmoothingAlpha = 0.1;
        double angleDecay = 0.995;
        double maxAngleChangeDeg = 0.5;       ///< 0: no clamp
    };

    /// Detects the horizon tilt and returns the frame rotated by the smoothed angle.
    /// The smoothed angle lives in process-wide state, as in the reference
    /// (RollCorrection.cpp:13-14); calls are serialised by a mutex.
    static cv::Mat autoCorrectRoll(const cv::Mat& input, const Parameters& params);
    /// Default parameters: the call form of examples/roll-correction-file.cpp:61.
    static cv::Mat autoCorrectRoll(const cv::Mat& input);
};

}  // namespace vs

#endif
