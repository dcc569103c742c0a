// RTSPServer stand-in (include/video/RTSPServer.h): no gst-rtsp-server in this build.
#include "video/RTSPServer.h"

#include <iostream>

RTSPServer::RTSPServer() = default;
RTSPServer::~RTSPServer() = default;

bool RTSPServer::startServer(int port, const std::string& mountPoint, int width, int height, int fps) {
    frameWidth = width; frameHeight = height; framerate = fps;
    std::cerr << "[RTSPServer] not built in: rtsp://0.0.0.0:" << port << mountPoint << " is not served" << std::endl;
    return false;
}

void RTSPServer::pushFrame(const cv::Mat&) {}
bool RTSPServer::isReady() const { return false; }
