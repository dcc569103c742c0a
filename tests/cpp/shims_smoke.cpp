// Drives the link-surface classes of examples/vs.cpp (CamCap, TcpReciever, DeepStreamTracker, RTSPServer) the way that
// main does (vs.cpp:232-254,355-364,496,567-584,746-759).  argv[1] picks the scenario; output is read by
// tests/test_host_shims.py.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include "video/CamCap.h"
#include "video/DeepStreamTracker.h"
#include "video/RTSPServer.h"
#include "video/TcpReciever.h"

static int camcap(const std::string &source, bool threaded, const std::string &colorspace, int want) {
    vs::CamCap::Parameters camParams;
    camParams.source = source;
    camParams.codec = "h264";
    camParams.threadedQueueMode = threaded;
    camParams.colorspace = colorspace;
    camParams.threadTimeout = 3000;
    std::unique_ptr<vs::CamCap> cam;
    try {
        cam = std::make_unique<vs::CamCap>(camParams);
    } catch (const std::exception &e) {
        std::printf("THROW %s\n", e.what());
        return 0;
    }
    std::printf("PROPS %.0f %.0f %.0f healthy=%d\n", cam->getWidth(), cam->getHeight(), cam->getFrameRate(), (int)cam->isHealthy());
    cam->start();
    std::printf("STARTED healthy=%d\n", (int)cam->isHealthy());
    for (int k = 0; k < want; k++) {
        cv::Mat frame = cam->read();
        if (frame.empty()) { std::printf("EMPTY\n"); break; }
        std::printf("FRAME %d %d %d %d\n", frame.rows, frame.cols, frame.channels(), (int)frame.data[0]);
    }
    cam->stop();
    std::printf("STOPPED healthy=%d empty_after_stop=%d\n", (int)cam->isHealthy(), (int)cam->read().empty());
    return 0;
}

static int tcp() {
    vs::TcpReciever tcp(0);
    if (!tcp.start()) { std::printf("START FAILED\n"); return 1; }
    std::printf("PORT %d\n", (int)tcp.port());
    std::fflush(stdout);
    int x = -1, y = -1;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20)) {
        if (tcp.tryGetLatest(x, y)) {
            std::printf("GOT %d %d\n", x, y);
            std::fflush(stdout);
            if (x == 9999) break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    tcp.stop();
    std::printf("AGAIN %d\n", (int)(tcp.start() && tcp.port() != 0));       // a stopped receiver can be started again
    tcp.stop();
    return 0;
}

static int tracker() {
    vs::DeepStreamTracker::Parameters trackerParams;
    trackerParams.processingWidth = 320;
    auto tracker = std::make_unique<vs::DeepStreamTracker>(trackerParams);
    const bool ok = tracker->initialize();
    cv::Mat frame(48, 64, CV_8UC3);
    std::memset(frame.data, 7, frame.step * 48);
    auto detections = tracker->processFrame(frame);
    cv::Mat drawn = tracker->drawDetections(frame, detections, 10, 10);
    cv::Mat plain = tracker->drawDetections(frame, detections);
    std::printf("TRACKER init=%d detections=%zu same_size=%d copy=%d pick=%d err=%s\n", (int)ok, detections.size(),
                (int)(drawn.rows == 48 && drawn.cols == 64 && plain.data[5] == 7), (int)(drawn.data != frame.data),
                tracker->pickIdAt(3, 3), tracker->getLastError().c_str());
    RTSPServer rtsp;
    const bool serving = rtsp.startServer(8554, "/test");
    rtsp.pushFrame(frame);
    std::printf("RTSP serving=%d ready=%d\n", (int)serving, (int)rtsp.isReady());
    return 0;
}

int main(int argc, char **argv) {
    const std::string what = argc > 1 ? argv[1] : "";
    if (what == "camcap") return camcap(argv[2], std::strcmp(argv[3], "threaded") == 0, argv[4], std::atoi(argv[5]));
    if (what == "tcp") return tcp();
    if (what == "tracker") return tracker();
    return 2;
}
