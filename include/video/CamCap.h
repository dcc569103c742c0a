// vs::CamCap - frame source of the reference's mains, source-compatible with /root/reference/include/video/CamCap.h:20-75
// (same Parameters, same public calls).  Host code only: a cv::VideoCapture plus an optional reader thread with a bounded
// frame queue.  What differs from the reference (src/CamCap.cpp:35-72,177-190): an "rtsp://..." or "*.mp4" source is
// handed to cv::VideoCapture as it is instead of being wrapped in a Jetson GStreamer string
// (`nvv4l2decoder ! nvvidconv ! ...`): those elements do not exist on an MI355X host.  Applications that decode on the
// GPU hand NV12 surfaces to vs_stab_push_dev instead (include/vs_stab.h).
#ifndef VIDEO_CAMCAP_H
#define VIDEO_CAMCAP_H

#include <opencv2/opencv.hpp>
#include <memory>
#include <string>

namespace vs {

class CamCap {
public:
    struct Parameters {
        std::string source = "0";       ///< camera index ("0"), file path or URL
        bool streamMode = false;        ///< accepted, unused (as in the reference)
        int backend = 0;                ///< cv::VideoCapture API preference (cv::CAP_ANY)
        std::string colorspace;         ///< "BGR2GRAY" | "BGR2HSV" | "BGR2YUV"; anything else is ignored
        std::string codec = "h265";     ///< accepted, unused here (the reference picks its GStreamer parser with it)
        bool logging = false;
        int timeDelay = 0;              ///< seconds to wait before the first read
        bool threadedQueueMode = true;  ///< a reader thread fills a queue; read() pops it
        int queueSize = 5;              ///< frames the queue holds before the reader blocks
        int threadTimeout = 500;        ///< ms read() waits for a frame; <= 0: no limit
    };

    /// Opens the source and reads one frame; throws std::runtime_error when either fails (CamCap.cpp:75-77,112-114).
    explicit CamCap(const Parameters& params);
    ~CamCap();
    CamCap(const CamCap&) = delete;
    CamCap& operator=(const CamCap&) = delete;

    void start();            ///< starts the reader thread (threaded mode; otherwise nothing)
    cv::Mat read();          ///< next frame; empty on time-out, at the end of the source and after stop()
    void stop();             ///< joins the reader, releases the capture, drops queued frames
    bool isHealthy() const;  ///< capture open and reader running

    double getFrameRate() const;
    double getWidth() const;
    double getHeight() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace vs

#endif
