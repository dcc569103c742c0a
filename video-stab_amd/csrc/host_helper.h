// One helper thread per stabilizer for the host entry points: a copy from or to PAGEABLE memory keeps the calling thread
// inside hipMemcpy for its whole length (the runtime stages it through its own bounce buffers on that thread), so the
// download of the previous result and the upload of this call's frame only overlap if they are issued from two threads
// (stabilizer.cpp, push_host_pipelined).  Host code; the thread sleeps between streams and polls while frames keep coming.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace vsd {

class HostHelper {
public:
    HostHelper();
    ~HostHelper();                          // joins
    HostHelper(const HostHelper&) = delete;
    HostHelper& operator=(const HostHelper&) = delete;

    void start(std::function<int()> job);   // one job at a time: wait() before the next start()
    int wait();                             // the job's return value

private:
    void loop();
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    std::atomic<int> state_{0};             // 0 idle, 1 job posted, 2 job done, 3 leave
    std::atomic<bool> asleep_{false};
    int result_ = 0;
    int spin_us_;
};

bool host_ptr_page_locked(const void* p);   // hipHostMalloc / hipHostRegister memory (what the DMA engine reads directly)

}  // namespace vsd
