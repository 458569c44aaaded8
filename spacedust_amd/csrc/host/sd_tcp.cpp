// Host-side rendezvous between the ranks of one job over TCP (MASTER_ADDR / MASTER_PORT as a one-process-per-GPU launcher
// sets them): the 128-byte RCCL unique id travels this way from rank 0 to the others (sd_tcp_bcast), and -- only when the
// ranks share a device, which RCCL refuses (test rigs with one GPU) -- the result records themselves (sd_tcp_gather).
// The reference's counterpart is MPI (M/src/commons/MMseqsMPI.cpp); one root, star topology, one connection per peer and call.
#include "spacedust_gpu.h"

#include <algorithm>
#include <cstdio>

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdlib>
#include <ctime>
#include <new>
#include <sys/time.h>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

namespace {

bool sendAll(int fd, const void *buf, uint64_t n) {
    const char *p = (const char *) buf;
    while (n) {
        const ssize_t w = ::send(fd, p, (size_t) std::min<uint64_t>(n, 1u << 24), MSG_NOSIGNAL);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) return false;
        p += w;
        n -= (uint64_t) w;
    }
    return true;
}

bool recvAll(int fd, void *buf, uint64_t n) {
    char *p = (char *) buf;
    while (n) {
        const ssize_t r = ::recv(fd, p, (size_t) std::min<uint64_t>(n, 1u << 24), 0);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) return false;
        p += r;
        n -= (uint64_t) r;
    }
    return true;
}

// root: a listening socket on port; peers: a connection to addr:port (retried while the root is not up yet)
int listenOn(int port) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    int one = 1;
    ::setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a;
    memset(&a, 0, sizeof(a));
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_ANY);
    a.sin_port = htons((uint16_t) port);
    for (int tries = 0; tries < 300; tries++) {   // the previous call's socket may still be closing
        if (::bind(fd, (sockaddr *) &a, sizeof(a)) == 0 && ::listen(fd, 128) == 0) return fd;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    ::close(fd);
    return -1;
}

int connectTo(const char *addr, int port, int timeoutSec) {
    addrinfo hints, *res = nullptr;
    memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    char portStr[16];
    snprintf(portStr, sizeof(portStr), "%d", port);
    if (::getaddrinfo(addr && *addr ? addr : "127.0.0.1", portStr, &hints, &res) != 0 || !res) return -1;
    int fd = -1;
    for (int tries = 0; tries < timeoutSec * 10; tries++) {
        fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) break;
        if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
            int one = 1;
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            ::freeaddrinfo(res);
            return fd;
        }
        ::close(fd);
        fd = -1;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    ::freeaddrinfo(res);
    return -1;
}

struct Hello {
    uint32_t magic, rank;
};
constexpr uint32_t MAGIC = 0x53444750u;   // "SDGP"

}  // namespace

struct sd_tcp {
    int nRanks = 1, rank = 0;
    std::vector<int> peer;   // root: the socket of every rank (peer[0] unused); others: peer[0] = the root
};

extern "C" {

// every rank connects to rank 0 once; the calls below reuse the connections (so they are matched in program order)
int sd_tcp_connect(const char *addr, int port, int nRanks, int rank, sd_tcp **out) {
    if (!out || nRanks < 1 || rank < 0 || rank >= nRanks) return SD_EINVAL;
    sd_tcp *t = new sd_tcp();
    t->nRanks = nRanks;
    t->rank = rank;
    if (nRanks > 1 && rank == 0) {
        t->peer.assign((size_t) nRanks, -1);
        const int ls = listenOn(port);
        if (ls < 0) {
            delete t;
            return SD_EHIP;
        }
        // Every rank must say hello within the deadline (SD_TCP_TIMEOUT seconds, default 600: a peer that died before connecting
        // must not park rank 0 forever).  A connection that does not introduce itself properly -- a port scanner, another job on
        // this port -- is closed and ignored; it does not take the job down.
        bool ok = true;
        const int deadlineS = getenv("SD_TCP_TIMEOUT") ? atoi(getenv("SD_TCP_TIMEOUT")) : 600;
        const time_t tEnd = time(nullptr) + (deadlineS > 0 ? deadlineS : 600);
        int have = 1;
        while (have < nRanks && ok) {
            const time_t nowT = time(nullptr);
            if (nowT >= tEnd) {
                ok = false;
                break;
            }
            pollfd pf;
            pf.fd = ls;
            pf.events = POLLIN;
            pf.revents = 0;
            const int pr = ::poll(&pf, 1, (int) std::min<long>(1000L * (long) (tEnd - nowT), 5000L));
            if (pr < 0 && errno != EINTR) ok = false;
            if (pr <= 0) continue;
            const int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) continue;
            timeval tv;
            tv.tv_sec = 10;
            tv.tv_usec = 0;
            ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));   // for the hello only
            Hello h;
            if (!recvAll(fd, &h, sizeof(h)) || h.magic != MAGIC || h.rank == 0 || h.rank >= (uint32_t) nRanks || t->peer[h.rank] >= 0) {
                ::close(fd);   // not one of ours (or a duplicate): ignored
                continue;
            }
            tv.tv_sec = 0;
            ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));   // payload receives block (a rank may compute for a long time)
            int one = 1;
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            t->peer[h.rank] = fd;
            have++;
        }
        ::close(ls);
        if (!ok) {
            for (int fd : t->peer)
                if (fd >= 0) ::close(fd);
            delete t;
            return SD_EHIP;
        }
    } else if (nRanks > 1) {
        const int fd = connectTo(addr, port, 300);
        Hello h = {MAGIC, (uint32_t) rank};
        if (fd < 0 || !sendAll(fd, &h, sizeof(h))) {
            if (fd >= 0) ::close(fd);
            delete t;
            return SD_EHIP;
        }
        t->peer.assign(1, fd);
    }
    *out = t;
    return SD_OK;
}

void sd_tcp_close(sd_tcp *t) {
    if (!t) return;
    for (int fd : t->peer)
        if (fd >= 0) ::close(fd);
    delete t;
}

// buf[bytes] of rank 0 -> every rank
int sd_tcp_bcast(sd_tcp *t, void *buf, uint64_t bytes) {
    if (!t || (bytes && !buf)) return SD_EINVAL;
    if (t->nRanks == 1) return SD_OK;
    if (t->rank == 0) {
        for (int r = 1; r < t->nRanks; r++)
            if (!sendAll(t->peer[(size_t) r], buf, bytes)) return SD_EHIP;
        return SD_OK;
    }
    return recvAll(t->peer[0], buf, bytes) ? SD_OK : SD_EHIP;
}

// gatherv of byte records to rank 0: sizes[nRanks] and the records concatenated in rank order on the root (outCap too small:
// SD_ENOMEM with *outBytes = the size needed, on the root; the records are consumed either way -- size them with a gather of
// the byte counts first)
int sd_tcp_gather(sd_tcp *t, const void *local, uint64_t nBytes, uint64_t *sizes, void *outOnRoot, uint64_t outCap, uint64_t *outBytes) {
    if (!t || (nBytes && !local)) return SD_EINVAL;
    if (t->rank != 0) {
        uint32_t ack = 0;
        const bool ok = sendAll(t->peer[0], &nBytes, sizeof(nBytes)) && sendAll(t->peer[0], local, nBytes) && recvAll(t->peer[0], &ack, sizeof(ack));
        return ok && ack == MAGIC ? SD_OK : SD_EHIP;
    }
    std::vector<std::vector<char> > parts((size_t) t->nRanks);
    parts[0].assign((const char *) local, (const char *) local + nBytes);
    for (int r = 1; r < t->nRanks; r++) {
        uint64_t n = 0;
        const uint32_t ack = MAGIC;
        if (!recvAll(t->peer[(size_t) r], &n, sizeof(n))) return SD_EHIP;
        if (n > (1ull << 36)) return SD_EINVAL;   // 64 GiB of result records from one rank: a corrupt length, not a result
        try {
            parts[(size_t) r].resize(n);
        } catch (const std::bad_alloc &) {
            return SD_ENOMEM;
        }
        if (!recvAll(t->peer[(size_t) r], parts[(size_t) r].data(), n) || !sendAll(t->peer[(size_t) r], &ack, sizeof(ack))) return SD_EHIP;
    }
    uint64_t total = 0;
    for (int r = 0; r < t->nRanks; r++) {
        if (sizes) sizes[r] = parts[(size_t) r].size();
        total += parts[(size_t) r].size();
    }
    if (outBytes) *outBytes = total;
    if (total > outCap) return SD_ENOMEM;
    uint64_t off = 0;
    for (int r = 0; r < t->nRanks; r++) {
        if (!parts[(size_t) r].empty()) memcpy((char *) outOnRoot + off, parts[(size_t) r].data(), parts[(size_t) r].size());
        off += parts[(size_t) r].size();
    }
    return SD_OK;
}

}  // extern "C"
