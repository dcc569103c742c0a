// vs::DeepStreamTracker - link-surface stand-in.  The reference's tracker (include/video/DeepStreamTracker.h:19-143,
// src/DeepStreamTracker.cpp) is an NVIDIA DeepStream / TensorRT GStreamer pipeline: nothing of it exists on an MI355X host
// and object tracking is not part of the stabilization path.  This header keeps the public
// types and calls the mains use (examples/vs.cpp:241,567-584) so that they compile and run with the tracker reporting
// "unavailable": initialize() fails with a message, processFrame() finds nothing, drawDetections() hands the frame back.
#ifndef VIDEO_DEEPSTREAM_TRACKER_H
#define VIDEO_DEEPSTREAM_TRACKER_H

#include <opencv2/opencv.hpp>
#include <string>
#include <vector>

namespace vs {

class DeepStreamTracker {
public:
    struct Parameters {
        std::string modelEngine;             ///< accepted, unused
        std::string modelConfigFile;         ///< accepted, unused
        std::string trackerConfigFile;       ///< accepted, unused
        int processingWidth = 640;
        int processingHeight = 384;
        int batchSize = 1;
        bool enableLowLatency = true;
        bool debugMode = false;
        bool saveDetectionImages = false;
        std::string saveImagePath = "/tmp/detections/";
        float confidenceThreshold = 0.5f;
        int gpuId = 0;
        int maxTrackedObjects = 100;
    };

    struct Detection {
        int classId = 0;
        float confidence = 0.f;
        cv::Rect bbox;
        int trackId = -1;
        std::string label;
    };

    DeepStreamTracker();                                             ///< default Parameters
    DeepStreamTracker(const Parameters& params);
    ~DeepStreamTracker();

    bool initialize();                                               ///< always false here; see getLastError()
    std::vector<Detection> processFrame(const cv::Mat& frame);      ///< no detections
    cv::Mat drawDetections(const cv::Mat& frame, const std::vector<Detection>& detections, int selX = -1, int selY = -1);
    int pickIdAt(int x, int y) const;                                ///< -1: nothing under the point
    void release();
    std::string getLastError() const;

private:
    Parameters params_;
    std::string lastErrorMessage_;
};

}  // namespace vs

#endif
