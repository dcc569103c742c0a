// RTSPServer - link-surface stand-in for /root/reference/include/video/RTSPServer.h:10-45 (a gst-rtsp-server wrapper:
// appsrc -> encoder -> RTSP mount).  examples/vs.cpp includes the header but never makes an RTSPServer (it pushes into
// its own GStreamer pipeline), so all that is kept is the public surface, without the GStreamer headers the
// reference's declaration pulls in: startServer() reports that no server was built in, pushFrame() drops the frame.
#ifndef VIDEO_RTSP_SERVER_H
#define VIDEO_RTSP_SERVER_H

#include <opencv2/opencv.hpp>
#include <string>

class RTSPServer {
public:
    RTSPServer();
    ~RTSPServer();

    bool startServer(int port, const std::string& mountPoint, int width = 1920, int height = 1080, int fps = 30);
    void pushFrame(const cv::Mat& frame);
    bool isReady() const;

private:
    int frameWidth = 0, frameHeight = 0, framerate = 0;
};

#endif
