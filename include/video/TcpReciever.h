// vs::TcpReciever - listens on a TCP port for "x y\n" text lines (the click coordinates the reference's mains feed to the
// object tracker, examples/vs.cpp:242-243,570) and keeps the most recent pair.  Source-compatible with
// /root/reference/include/video/TcpReciever.h:9-31 (the class name keeps the reference's spelling).
#ifndef VIDEO_TCP_RECIEVER_H
#define VIDEO_TCP_RECIEVER_H

#include <atomic>
#include <cstdint>
#include <thread>

namespace vs {

class TcpReciever {
public:
    explicit TcpReciever(uint16_t port);     ///< port 0: the system picks one (see port())
    ~TcpReciever();
    TcpReciever(const TcpReciever&) = delete;
    TcpReciever& operator=(const TcpReciever&) = delete;

    bool start();    ///< binds, listens and starts the listener thread; false when the port cannot be bound
    void stop();     ///< closes the listener and joins the thread

    /// The most recent pair that arrived since the last call; a pair is handed out once.  Pairs with a negative
    /// coordinate count as "nothing" (TcpReciever.cpp:63-71).
    bool tryGetLatest(int& outX, int& outY);

    uint16_t port() const { return port_; }  ///< the bound port (after start())

private:
    void listenLoop();

    uint16_t port_;
    int listenFd_{-1};
    int wakeFd_[2]{-1, -1};                   // stop() writes, the listener polls: no blocking call is left to interrupt
    std::thread thread_;
    std::atomic<bool> running_{false};
    std::atomic<uint64_t> latest_{~0ull};     // (x, y) as one word: a reader never sees x of one line with y of another
};

}  // namespace vs

#endif
