// vs::CamCap (include/video/CamCap.h): behaviour of /root/reference/src/CamCap.cpp:15-386, host code only.
//   constructor   open (all-digit source = camera index), optional colour conversion code, frame rate (< 1 -> 0),
//                 warm-up sleep, one frame read as a proof of life and queued (:15-134)
//   reader        one thread: read -> convert -> push, blocking while the queue is full; five failed reads in a row
//                 -> release, wait a second, open again; an empty Mat marks the end when it leaves (:154-256)
//   read()        threaded: wait up to threadTimeout ms for a queued frame; direct mode: read + convert (:258-319)
//   stop()        join, release, drop what is queued; start() after stop() re-opens through the reader's
//                 failure path (examples/vs.cpp:355-364 relies on that) (:321-347)
#include "video/CamCap.h"

#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace vs {

namespace {

constexpr int kFailuresBeforeReopen = 5;
constexpr auto kReopenPause = std::chrono::milliseconds(1000);
constexpr auto kRetryPause = std::chrono::milliseconds(50);

int colour_code(const std::string& name) {
    if (name == "BGR2GRAY") return cv::COLOR_BGR2GRAY;
    if (name == "BGR2HSV") return cv::COLOR_BGR2HSV;
    if (name == "BGR2YUV") return cv::COLOR_BGR2YUV;
    return -1;
}

bool all_digits(const std::string& s) {
    if (s.empty()) return false;
    for (unsigned char c : s)
        if (!std::isdigit(c)) return false;
    return true;
}

}  // namespace

struct CamCap::Impl {
    Parameters p;
    cv::VideoCapture cap;
    int code = -1;
    double fps = 0.0;

    std::thread reader;
    std::atomic<bool> quit{false};
    std::atomic<bool> running{false};
    std::mutex m;
    std::condition_variable cv_;
    std::deque<cv::Mat> q;

    bool open_source() {
        if (all_digits(p.source)) return cap.open(std::stoi(p.source), p.backend);
        return cap.open(p.source, p.backend);
    }
    void convert(cv::Mat& f) const {
        if (code != -1) cv::cvtColor(f, f, code);
    }
    void log(const char* what) const {
        if (p.logging) std::cout << "[CamCap] " << what << std::endl;
    }

    void reader_loop() {
        int failed = 0;
        while (!quit) {
            cv::Mat f;
            if (!cap.isOpened() || !cap.read(f) || f.empty()) {
                failed++;
                log("read failed");
                if (failed >= kFailuresBeforeReopen) {
                    log("re-opening the source");
                    cap.release();
                    std::this_thread::sleep_for(kReopenPause);
                    if (!quit && open_source()) { failed = 0; continue; }
                }
                std::this_thread::sleep_for(kRetryPause);
                continue;
            }
            failed = 0;
            convert(f);
            std::unique_lock<std::mutex> lk(m);
            cv_.wait(lk, [&] { return quit || q.size() < (size_t)p.queueSize; });
            if (quit) break;
            q.push_back(f);
            cv_.notify_all();
        }
        {
            std::lock_guard<std::mutex> lk(m);
            q.push_back(cv::Mat());          // end marker: a waiting read() returns empty
            cv_.notify_all();
        }
        running = false;
        log("reader thread left");
    }
};

CamCap::CamCap(const Parameters& params) : impl_(new Impl) {
    Impl& s = *impl_;
    s.p = params;
    if (s.p.logging) std::cout << "[CamCap] source: " << s.p.source << std::endl;
    if (!s.open_source()) throw std::runtime_error("[CamCap] Failed to open source: " + s.p.source);
    if (!s.p.colorspace.empty()) {
        s.code = colour_code(s.p.colorspace);
        if (s.code == -1 && s.p.logging) std::cerr << "[CamCap] Warning: Invalid colorspace " << s.p.colorspace << " ignored." << std::endl;
    }
    s.fps = s.cap.get(cv::CAP_PROP_FPS);
    if (!(s.fps >= 1.0)) s.fps = 0.0;
    if (s.p.timeDelay > 0) std::this_thread::sleep_for(std::chrono::seconds(s.p.timeDelay));
    cv::Mat first;
    if (!s.cap.read(first) || first.empty()) throw std::runtime_error("[CamCap] Failed to read initial frame!");
    s.convert(first);
    if (s.p.threadedQueueMode) s.q.push_back(first.clone());
}

CamCap::~CamCap() { stop(); }

void CamCap::start() {
    Impl& s = *impl_;
    if (!s.p.threadedQueueMode || s.running) return;
    if (s.reader.joinable()) s.reader.join();      // a reader that left on its own
    s.quit = false;
    s.running = true;
    s.reader = std::thread([&s] { s.reader_loop(); });
    s.log("reader thread started");
}

cv::Mat CamCap::read() {
    Impl& s = *impl_;
    if (!s.p.threadedQueueMode) {
        cv::Mat f;
        if (!s.cap.isOpened() || !s.cap.read(f) || f.empty()) return cv::Mat();
        s.convert(f);
        return f;
    }
    std::unique_lock<std::mutex> lk(s.m);
    auto ready = [&] { return !s.q.empty() || s.quit.load(); };
    if (s.p.threadTimeout <= 0) {
        s.cv_.wait(lk, ready);
    } else if (!s.cv_.wait_for(lk, std::chrono::milliseconds(s.p.threadTimeout), ready)) {
        if (s.p.logging) std::cerr << "[CamCap] Timed out waiting for frame in queue." << std::endl;
        return cv::Mat();
    }
    if (s.q.empty()) return cv::Mat();
    cv::Mat f = s.q.front();
    s.q.pop_front();
    s.cv_.notify_all();
    return f;
}

void CamCap::stop() {
    Impl& s = *impl_;
    {
        std::lock_guard<std::mutex> lk(s.m);
        s.quit = true;
        s.cv_.notify_all();
    }
    if (s.reader.joinable()) s.reader.join();
    s.running = false;
    if (s.cap.isOpened()) s.cap.release();
    std::lock_guard<std::mutex> lk(s.m);
    s.q.clear();
}

bool CamCap::isHealthy() const { return impl_->cap.isOpened() && impl_->running; }
double CamCap::getFrameRate() const { return impl_->fps; }
double CamCap::getWidth() const { return impl_->cap.get(cv::CAP_PROP_FRAME_WIDTH); }
double CamCap::getHeight() const { return impl_->cap.get(cv::CAP_PROP_FRAME_HEIGHT); }

}  // namespace vs
