// See host_helper.h.
#include "host_helper.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>

namespace vsd {

HostHelper::HostHelper() {
    // A sleeping thread takes 50 - 150 us to wake - most of the copy it is meant to hide.  The helper therefore polls for
    // VS_STAB_HELPER_SPIN_US (default 2000) after its last job before it sleeps: awake while a stream runs (a call every
    // 0.2 - 0.5 ms), asleep otherwise.
    const char* e = std::getenv("VS_STAB_HELPER_SPIN_US");
    spin_us_ = e && *e ? std::max(0, std::min(1000000, std::atoi(e))) : 2000;
    th_ = std::thread([this] { loop(); });
}

HostHelper::~HostHelper() {
    (void)wait();       // a job still posted or running would overwrite the "leave" state with "done" and the join below would never return
    {
        std::lock_guard<std::mutex> lk(m_);
        state_.store(3, std::memory_order_release);
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
}

void HostHelper::loop() {
    for (;;) {
        const auto t0 = std::chrono::steady_clock::now();
        int st;
        while ((st = state_.load(std::memory_order_acquire)) != 1 && st != 3) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) {
                std::unique_lock<std::mutex> lk(m_);
                asleep_.store(true, std::memory_order_seq_cst);
                cv_.wait(lk, [this] { const int s = state_.load(std::memory_order_acquire); return s == 1 || s == 3; });
                asleep_.store(false, std::memory_order_seq_cst);
                continue;
            }
            __builtin_ia32_pause();
        }
        if (st == 3) return;
        result_ = job_();
        state_.store(2, std::memory_order_release);
    }
}

void HostHelper::start(std::function<int()> job) {
    job_ = std::move(job);
    {
        std::lock_guard<std::mutex> lk(m_);     // (with the lock: a helper about to sleep sees the job or gets the notify)
        state_.store(1, std::memory_order_seq_cst);
    }
    if (asleep_.load(std::memory_order_seq_cst)) cv_.notify_one();
}

int HostHelper::wait() {
    if (state_.load(std::memory_order_acquire) == 0) return 0;
    while (state_.load(std::memory_order_acquire) != 2) std::this_thread::yield();
    state_.store(0, std::memory_order_relaxed);
    return result_;
}

bool host_ptr_page_locked(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();    // "not a HIP pointer" is the answer, not an error to keep
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

}  // namespace vsd
