// vs::TcpReciever (include/video/TcpReciever.h): one listener thread, one client at a time, text lines "x y\n"
// (/root/reference/src/TcpReciever.cpp:21-105).  poll() over the sockets and a wake pipe, so stop() never has to
// interrupt a blocking accept()/recv().
#include "video/TcpReciever.h"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cstdio>
#include <string>

namespace vs {

namespace {

uint64_t pack(int x, int y) { return ((uint64_t)(uint32_t)x << 32) | (uint32_t)y; }
constexpr uint64_t kNothing = ~0ull;      // (-1, -1)

void close_fd(int& fd) {
    if (fd >= 0) ::close(fd);
    fd = -1;
}

}  // namespace

TcpReciever::TcpReciever(uint16_t port) : port_(port) {}
TcpReciever::~TcpReciever() { stop(); }

bool TcpReciever::start() {
    if (running_) return true;
    listenFd_ = ::socket(AF_INET, SOCK_STREAM, 0);
    if (listenFd_ < 0) return false;
    const int on = 1;
    ::setsockopt(listenFd_, SOL_SOCKET, SO_REUSEADDR, &on, sizeof(on));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
    addr.sin_port = htons(port_);
    socklen_t len = sizeof(addr);
    if (::bind(listenFd_, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) < 0 || ::listen(listenFd_, 4) < 0 ||
        ::getsockname(listenFd_, reinterpret_cast<sockaddr*>(&addr), &len) < 0 || ::pipe(wakeFd_) < 0) {
        close_fd(listenFd_);
        return false;
    }
    port_ = ntohs(addr.sin_port);
    running_ = true;
    thread_ = std::thread(&TcpReciever::listenLoop, this);
    return true;
}

void TcpReciever::stop() {
    if (!running_.exchange(false)) return;
    const char c = 0;
    if (::write(wakeFd_[1], &c, 1) < 0) { /* the listener also leaves when its sockets close */ }
    if (thread_.joinable()) thread_.join();
    close_fd(listenFd_);
    close_fd(wakeFd_[0]);
    close_fd(wakeFd_[1]);
}

bool TcpReciever::tryGetLatest(int& outX, int& outY) {
    const uint64_t v = latest_.exchange(kNothing);
    const int x = (int)(uint32_t)(v >> 32), y = (int)(uint32_t)v;
    if (x < 0 || y < 0) return false;
    outX = x;
    outY = y;
    return true;
}

void TcpReciever::listenLoop() {
    int client = -1;
    std::string pending;
    while (running_) {
        pollfd fds[2] = {{wakeFd_[0], POLLIN, 0}, {client >= 0 ? client : listenFd_, POLLIN, 0}};
        if (::poll(fds, 2, -1) < 0) continue;
        if (fds[0].revents) break;
        if (!fds[1].revents) continue;
        if (client < 0) {
            client = ::accept(listenFd_, nullptr, nullptr);
            pending.clear();
            continue;
        }
        char buf[256];
        const ssize_t n = ::recv(client, buf, sizeof(buf), 0);
        if (n <= 0) { close_fd(client); continue; }       // the client left: wait for the next one
        pending.append(buf, (size_t)n);
        for (size_t nl; (nl = pending.find('\n')) != std::string::npos; pending.erase(0, nl + 1)) {
            int x, y;
            if (std::sscanf(pending.substr(0, nl).c_str(), "%d %d", &x, &y) == 2) latest_ = pack(x, y);
        }
        if (pending.size() > 4096) pending.clear();       // no line end in sight: not our protocol
    }
    close_fd(client);
}

}  // namespace vs
