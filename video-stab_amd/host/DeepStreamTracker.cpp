// vs::DeepStreamTracker stand-in (include/video/DeepStreamTracker.h): no detector is built into this library.
#include "video/DeepStreamTracker.h"

namespace vs {

DeepStreamTracker::DeepStreamTracker() = default;
DeepStreamTracker::DeepStreamTracker(const Parameters& params) : params_(params) {}
DeepStreamTracker::~DeepStreamTracker() { release(); }

bool DeepStreamTracker::initialize() {
    lastErrorMessage_ = "DeepStreamTracker: NVIDIA DeepStream is not available in this build (MI355X); tracking is disabled";
    return false;
}

std::vector<DeepStreamTracker::Detection> DeepStreamTracker::processFrame(const cv::Mat&) { return {}; }

cv::Mat DeepStreamTracker::drawDetections(const cv::Mat& frame, const std::vector<Detection>&, int, int) {
    return frame.empty() ? frame : frame.clone();     // the reference draws on a copy (DeepStreamTracker.cpp:139-146)
}

int DeepStreamTracker::pickIdAt(int, int) const { return -1; }
void DeepStreamTracker::release() {}
std::string DeepStreamTracker::getLastError() const { return lastErrorMessage_; }

}  // namespace vs
