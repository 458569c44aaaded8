// Multi-GPU seam of the C ABI: one process per GPU, query genome sets dealt to the ranks, the target
// index replicated, no collective on the data path; the only exchange is the final gather of the per-rank result
// records to rank `root` over RCCL (xGMI).  The reference's counterpart is the MPI master merging per-rank result files
// through the shared file system (M/src/prefiltering/Prefiltering.cpp:630-658).
//
// RCCL is opened with dlopen on the first sd_comm_* call (the librccl next to the HIP runtime the process runs on), so a
// single-GPU run never loads it.
#include "sd_common.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace {

// the few RCCL entry points used (signatures of rccl.h; ncclComm_t / ncclUniqueId are opaque here)
typedef struct { char internal[128]; } RcclUniqueId;
typedef void *RcclComm;
typedef int (*FnGetUniqueId)(RcclUniqueId *);
typedef int (*FnCommInitRank)(RcclComm *, int, RcclUniqueId, int);
typedef int (*FnCommDestroy)(RcclComm);
typedef int (*FnAllGather)(const void *, void *, size_t, int /*ncclDataType_t*/, RcclComm, hipStream_t);
typedef int (*FnSend)(const void *, size_t, int, int, RcclComm, hipStream_t);
typedef int (*FnRecv)(void *, size_t, int, int, RcclComm, hipStream_t);
typedef int (*FnGroup)(void);
typedef const char *(*FnErr)(int);
enum { RCCL_CHAR = 0, RCCL_UINT64 = 5 };   // ncclInt8 = 0, ncclUint64 = 5

struct Rccl {
    void *lib = nullptr;
    FnGetUniqueId getUniqueId = nullptr;
    FnCommInitRank commInitRank = nullptr;
    FnCommDestroy commDestroy = nullptr;
    FnAllGather allGather = nullptr;
    FnSend send = nullptr;
    FnRecv recv = nullptr;
    FnGroup groupStart = nullptr, groupEnd = nullptr;
    FnErr errString = nullptr;
    std::string err;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // The RCCL that belongs to the HIP runtime this process runs on: the one next to the loaded libamdhip64 (a process that
        // imported PyTorch first runs on PyTorch's bundled runtime and RCCL, any other on /opt/rocm's) -- an RCCL built for
        // another runtime fails in ncclCommInitRank.  By path, so that a differently-placed copy already in the process does
        // not answer for it.
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr((void *) &hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                names.push_back(dir + "/librccl.so.1");
                names.push_back(dir + "/librccl.so");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const std::string &n : names) {
            r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            r.err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "");
            return;
        }
        r.getUniqueId = (FnGetUniqueId) dlsym(r.lib, "ncclGetUniqueId");
        r.commInitRank = (FnCommInitRank) dlsym(r.lib, "ncclCommInitRank");
        r.commDestroy = (FnCommDestroy) dlsym(r.lib, "ncclCommDestroy");
        r.allGather = (FnAllGather) dlsym(r.lib, "ncclAllGather");
        r.send = (FnSend) dlsym(r.lib, "ncclSend");
        r.recv = (FnRecv) dlsym(r.lib, "ncclRecv");
        r.groupStart = (FnGroup) dlsym(r.lib, "ncclGroupStart");
        r.groupEnd = (FnGroup) dlsym(r.lib, "ncclGroupEnd");
        r.errString = (FnErr) dlsym(r.lib, "ncclGetErrorString");
        if (!r.getUniqueId || !r.commInitRank || !r.commDestroy || !r.allGather || !r.send || !r.recv || !r.groupStart || !r.groupEnd)
            r.err = "librccl lacks an expected entry point";
    });
    return &r;
}

}  // namespace

struct sd_comm {
    RcclComm comm = nullptr;
    int nRanks = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // staging kept between calls (grow-only): [0] this rank's records, [1] the gathered records (root) -- a pinned host buffer a caller
    // can build its records in / read the gathered ones from (sd_comm_host_buffer), and the device buffer the exchange runs on
    void *hostBuf[2] = {nullptr, nullptr};
    uint64_t hostCap[2] = {0, 0};
    char *devBuf[2] = {nullptr, nullptr};
    uint64_t devCap[2] = {0, 0};
    uint64_t *dSizes = nullptr;   // [nRanks + 1]: the size / status exchange of a gather (kept: hipFree waits for the whole device)
    hipEvent_t ev = nullptr;
    uint64_t *hWords = nullptr;   // pinned [nRanks + 1]: the host side of the size / status exchange
    // waits for the communicator's stream asleep: hipStreamSynchronize spins (thread CPU time = wall time on this runtime), and a gather
    // that runs round by round behind the search waits for its peers while the rank's host stages need its two or three cores
    hipError_t wait() {
        if (!ev) return hipStreamSynchronize(stream);
        const hipError_t e = hipEventRecord(ev, stream);
        return e == hipSuccess ? sdEventWait(ev) : e;
    }
    int ensureDev(int which, uint64_t bytes) {
        if (devCap[which] >= bytes) return SD_OK;
        if (devBuf[which]) (void) hipFree(devBuf[which]);
        devBuf[which] = nullptr;
        devCap[which] = 0;
        if (hipMalloc((void **) &devBuf[which], bytes) != hipSuccess) {
            (void) hipGetLastError();
            return SD_ENOMEM;
        }
        devCap[which] = bytes;
        return SD_OK;
    }
};

extern "C" {

int sd_comm_unique_id(char *out128) {
    if (!out128) return SD_EINVAL;
    Rccl *R = rccl();
    if (!R->err.empty()) return SD_EUNSUPPORTED;
    RcclUniqueId id;
    if (R->getUniqueId(&id) != 0) return SD_EHIP;
    memcpy(out128, id.internal, 128);
    return SD_OK;
}

int sd_comm_init(int device, int nRanks, int rank, const char *uniqueId128, sd_comm **out) {
    if (!out || !uniqueId128 || nRanks < 1 || rank < 0 || rank >= nRanks) return SD_EINVAL;
    Rccl *R = rccl();
    if (!R->err.empty()) {
        fprintf(stderr, "sd_comm_init: %s\n", R->err.c_str());
        return SD_EUNSUPPORTED;
    }
    if (hipSetDevice(device) != hipSuccess) return SD_ENODEVICE;
    sd_comm *c = new sd_comm();
    c->nRanks = nRanks;
    c->rank = rank;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SD_EHIP;
    }
    if (hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) {
        (void) hipGetLastError();
        c->ev = nullptr;   // (hipStreamSynchronize then)
    }
    RcclUniqueId id;
    memcpy(id.internal, uniqueId128, 128);
    const int rc = R->commInitRank(&c->comm, nRanks, id, rank);
    if (rc != 0) {
        fprintf(stderr, "sd_comm_init: ncclCommInitRank failed: %s\n", R->errString ? R->errString(rc) : "?");
        (void) hipStreamDestroy(c->stream);
        delete c;
        return SD_EHIP;
    }
    *out = c;
    return SD_OK;
}

// A pinned host buffer owned by the communicator (freed by sd_comm_destroy; a larger request replaces it, so take the pointer again
// after every call): which = 0 for this rank's records, 1 for the gathered records on the root.  A rank that builds its records in
// buffer 0 and a root that receives into buffer 1 move them between host and device at the bus rate with no staging copy -- pageable
// memory goes through the runtime's bounce buffers (a 32-GB gather of eight ranks: ~3 s on the root) and a fresh 32-GB array has to be
// touched first.  The matching device buffer is allocated with it, so a gather inside a timed region allocates nothing.
int sd_comm_host_buffer(sd_comm *c, int which, uint64_t bytes, void **ptr) {
    if (!c || which < 0 || which > 1 || !ptr) return SD_EINVAL;
    if (hipSetDevice(c->device) != hipSuccess) return SD_ENODEVICE;
    if (c->hostCap[which] < bytes) {
        if (c->hostBuf[which]) (void) hipHostFree(c->hostBuf[which]);
        c->hostBuf[which] = nullptr;
        c->hostCap[which] = 0;
        if (bytes && hipHostMalloc(&c->hostBuf[which], bytes, hipHostMallocDefault) != hipSuccess) {
            (void) hipGetLastError();
            return SD_ENOMEM;
        }
        c->hostCap[which] = bytes;
    }
    if (bytes) {
        const int rc = c->ensureDev(which, bytes);
        if (rc != SD_OK) return rc;
    }
    *ptr = c->hostBuf[which];
    return SD_OK;
}

void sd_comm_destroy(sd_comm *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    for (int w = 0; w < 2; w++) {
        if (c->hostBuf[w]) (void) hipHostFree(c->hostBuf[w]);
        if (c->devBuf[w]) (void) hipFree(c->devBuf[w]);
    }
    if (c->dSizes) (void) hipFree(c->dSizes);
    if (c->ev) (void) hipEventDestroy(c->ev);
    if (c->hWords) (void) hipHostFree(c->hWords);
    if (c->comm) rccl()->commDestroy(c->comm);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

const char *sd_comm_last_error(sd_comm *c) { return c ? c->err.c_str() : ""; }

// gatherv of byte records to `root`: all_gather of the sizes, then grouped send / recv of the payloads
int sd_gather_results(sd_comm *c, const void *local, uint64_t nBytes, int root, uint64_t *sizes, void *outOnRoot, uint64_t outCap,
                      uint64_t *outBytes) {
    if (!c || (nBytes && !local) || !sizes || root < 0 || root >= c->nRanks) return SD_EINVAL;
    Rccl *R = rccl();
    auto fail = [&](const char *what, int rc) {
        c->err = std::string(what) + ": " + (R->errString ? R->errString(rc) : "RCCL error");
        return SD_EHIP;
    };
    if (hipSetDevice(c->device) != hipSuccess) return SD_ENODEVICE;
    uint64_t *dSizes = nullptr;
    char *dLocal = nullptr, *dAll = nullptr;   // (the communicator's grow-only staging buffers: nothing here is freed per call)
    int status = SD_OK;
    do {
        if (!c->dSizes && hipMalloc((void **) &c->dSizes, sizeof(uint64_t) * ((size_t) c->nRanks + 1)) != hipSuccess) { status = SD_ENOMEM; break; }
        dSizes = c->dSizes;
        // the few words of the size / status exchange go through pinned host words of the communicator: a copy between the device and
        // pageable memory -- a stack variable, the caller's array -- waits for the stream inside the call, spinning
        if (!c->hWords && hipHostMalloc((void **) &c->hWords, sizeof(uint64_t) * ((size_t) c->nRanks + 1), hipHostMallocDefault) != hipSuccess) {
            (void) hipGetLastError();
            c->hWords = nullptr;
            status = SD_ENOMEM;
            break;
        }
        uint64_t *hW = c->hWords;
        hW[c->nRanks] = nBytes;
        if (hipMemcpyAsync(dSizes + c->nRanks, hW + c->nRanks, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) { status = SD_EHIP; break; }
        int rc = R->allGather(dSizes + c->nRanks, dSizes, 1, RCCL_UINT64, c->comm, c->stream);
        if (rc != 0) { status = fail("ncclAllGather", rc); break; }
        if (hipMemcpyAsync(hW, dSizes, sizeof(uint64_t) * c->nRanks, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { status = SD_EHIP; break; }
        if (c->wait() != hipSuccess) { status = SD_EHIP; break; }
        memcpy(sizes, hW, sizeof(uint64_t) * c->nRanks);
        uint64_t total = 0;
        for (int r = 0; r < c->nRanks; r++) total += sizes[r];
        if (outBytes) *outBytes = total;
        // local staging; whatever fails here is agreed on by all ranks before anyone enters the payload exchange -- a rank that
        // left early would leave the others waiting in their send / recv
        int localStatus = SD_OK;
        bool capacity = false;   // the root's buffer is too small: every rank returns SD_ENOMEM (a size probe), nothing is exchanged
        if (c->rank == root && total > outCap) {
            localStatus = SD_ENOMEM;
            capacity = true;
        }
        if (localStatus == SD_OK && nBytes) {
            if (c->ensureDev(0, nBytes) != SD_OK) localStatus = SD_ENOMEM;
            else {
                dLocal = c->devBuf[0];
                if (hipMemcpyAsync(dLocal, local, nBytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) localStatus = SD_EHIP;
            }
        }
        if (localStatus == SD_OK && c->rank == root && total) {
            if (c->ensureDev(1, total) != SD_OK) localStatus = SD_ENOMEM;
            else dAll = c->devBuf[1];
        }
        (void) hipGetLastError();
        uint64_t mine = (uint64_t) (localStatus == SD_OK ? 0 : capacity ? 2 : 1);
        std::vector<uint64_t> all((size_t) c->nRanks, 0);
        hW[c->nRanks] = mine;
        if (hipMemcpyAsync(dSizes + c->nRanks, hW + c->nRanks, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) { status = SD_EHIP; break; }
        rc = R->allGather(dSizes + c->nRanks, dSizes, 1, RCCL_UINT64, c->comm, c->stream);
        if (rc != 0) { status = fail("ncclAllGather (status)", rc); break; }
        if (hipMemcpyAsync(hW, dSizes, sizeof(uint64_t) * c->nRanks, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { status = SD_EHIP; break; }
        if (c->wait() != hipSuccess) { status = SD_EHIP; break; }
        memcpy(all.data(), hW, sizeof(uint64_t) * c->nRanks);
        int firstBad = -1;
        bool realFailure = false;   // some rank could not stage its records (as opposed to the root's capacity probe)
        for (int r = 0; r < c->nRanks; r++) {
            if (all[r] && firstBad < 0) firstBad = r;
            if (all[r] == 1) realFailure = true;
        }
        if (firstBad >= 0) {   // nobody exchanges anything
            // every rank returns the SAME code: SD_ENOMEM only for the capacity probe (callers repeat the collective with a larger
            // buffer -- all of them or none), SD_EHIP when a rank's own staging failed
            status = realFailure ? SD_EHIP : SD_ENOMEM;
            c->err = "sd_gather_results: rank " + std::to_string(firstBad) +
                     (firstBad == root ? " (root) could not take the gathered records (output capacity or device memory)"
                                       : " could not stage its records");
            break;
        }
        rc = R->groupStart();
        if (rc != 0) { status = fail("ncclGroupStart", rc); break; }
        if (c->rank == root) {
            uint64_t off = 0;
            for (int r = 0; r < c->nRanks; r++) {
                if (sizes[r] && r != root) R->recv(dAll + off, sizes[r], RCCL_CHAR, r, c->comm, c->stream);
                off += sizes[r];
            }
        } else if (nBytes) {
            R->send(dLocal, nBytes, RCCL_CHAR, root, c->comm, c->stream);
        }
        rc = R->groupEnd();
        if (rc != 0) { status = fail("ncclGroupEnd", rc); break; }
        if (c->rank == root && dAll) {
            uint64_t off = 0;
            for (int r = 0; r < root; r++) off += sizes[r];
            if (nBytes && hipMemcpyAsync(dAll + off, dLocal, nBytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) status = SD_EHIP;
            if (status == SD_OK && outOnRoot && hipMemcpyAsync(outOnRoot, dAll, total, hipMemcpyDeviceToHost, c->stream) != hipSuccess) status = SD_EHIP;
        }
        if (c->wait() != hipSuccess && status == SD_OK) status = SD_EHIP;
    } while (false);
    return status;
}

// ---- the gather, round by round, behind a running search -------------------------------------------------------------------
// One blob per rank at the end puts N x (records of all steps) through the root's PCIe link behind the last kernel.  Here the ranges
// of a stream are grouped into rounds (a bench step, a batch of query sets); the records of a range arrive through
// sd_gather_stream_sink -- sd_search's records sink, called on the stream's finalising thread when the range is done -- and are
// appended to the communicator's pinned send buffer; when the last range of a round has arrived, a worker thread runs
// sd_gather_results for that round on the communicator's own HIP stream while the search goes on with the next ranges.  Every rank
// runs exactly nRounds gathers in round order (a round without ranges on this rank sends 0 bytes), so the collectives match.  On
// the root the rounds land back to back in outOnRoot: round 0's records of rank 0 .. N-1, round 1's, ...
struct sd_gather_stream {
    sd_comm *c = nullptr;   // the RCCL communicator, or
    sd_tcp *tcp = nullptr;  // the TCP rendezvous (ranks that share a device: root 0)
    int root = 0, nRanks = 1, rank = 0;
    uint32_t nRanges = 0, nRounds = 0;
    std::vector<uint32_t> rangesThrough;   // [nRounds]: ranges of rounds 0 .. r
    char *out = nullptr;                   // the caller's buffer on the root, or (ownOut) owned.data()
    uint64_t outCap = 0, outUsed = 0;
    bool ownOut = false;                   // no buffer given: the root's buffer grows round by round
    std::vector<char> owned;
    std::vector<char> overflow;            // a round that does not fit the pinned send buffer is staged here (pageable: slower, never wrong)
    uint64_t sendUsed = 0;                 // bytes of the send buffer handed out so far
    uint64_t roundBase = 0;                // send-buffer offset of the current round's first byte
    bool roundInOverflow = false;
    struct Round { uint64_t off, bytes; bool inOverflow; std::vector<char> own; };
    std::vector<Round> rounds;             // filled as rounds complete
    uint32_t nSunk = 0, nReady = 0;
    std::vector<uint64_t> roundOff, sizes;
    int status = SD_OK;
    bool closing = false;
    std::mutex mu;
    std::condition_variable cv;
    std::thread worker;
    char *sendBuf() const { return c ? (char *) c->hostBuf[0] : nullptr; }
    uint64_t sendCap() const { return c ? c->hostCap[0] : 0; }
    bool grow(uint64_t need) {   // root, ownOut: room for `need` more bytes behind outUsed
        if (outUsed + need <= owned.size()) return true;
        try {
            owned.resize(std::max<uint64_t>(outUsed + need, owned.size() + owned.size() / 2));
        } catch (const std::bad_alloc &) {
            return false;
        }
        out = owned.data();
        outCap = owned.size();
        return true;
    }
    // one round over RCCL: when the root's room is too small every rank learns the size (SD_ENOMEM is agreed on before any payload
    // moves); with a buffer of its own the root grows it and every rank repeats the collective
    int roundRccl(const char *src, uint64_t bytes, uint64_t *szs, uint64_t *total) {
        int rc = sd_gather_results(c, src, bytes, root, szs, rank == root ? out + outUsed : nullptr, rank == root ? outCap - outUsed : 0, total);
        if (rc == SD_ENOMEM && ownOut) {
            const bool ok = rank != root || grow(*total);
            rc = sd_gather_results(c, src, bytes, root, szs, rank == root && ok ? out + outUsed : nullptr, rank == root && ok ? outCap - outUsed : 0, total);
        }
        return rc;
    }
    // one round over the TCP rendezvous: sd_tcp_gather consumes the records whatever the root's room, so a root with a buffer of its own
    // gathers the byte counts first
    int roundTcp(const char *src, uint64_t bytes, uint64_t *szs, uint64_t *total) {
        bool room = true;
        if (ownOut) {
            std::vector<uint64_t> all((size_t) nRanks, 0);
            uint64_t got = 0;
            const int rc = sd_tcp_gather(tcp, &bytes, sizeof(bytes), nullptr, all.data(), all.size() * sizeof(uint64_t), &got);
            if (rc != SD_OK) return rc;
            if (rank == 0) {
                uint64_t need = 0;
                for (uint64_t v : all) need += v;
                room = grow(need);   // (without room the payload round still runs: the peers are already sending)
            }
        }
        const int rc = sd_tcp_gather(tcp, src, bytes, szs, rank == 0 && room ? out + outUsed : nullptr, rank == 0 && room ? outCap - outUsed : 0, total);
        return room || rc == SD_EHIP ? rc : SD_ENOMEM;
    }
    void run() {
        for (uint32_t r = 0; r < nRounds; r++) {
            Round rd;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return nReady > r || closing; });
                if (nReady > r) rd = std::move(rounds[r]);
                else rd = Round{0, 0, false, {}};   // closed early: the ranks still run the same number of collectives
            }
            roundOff[r] = outUsed;
            // RCCL: every rank saw the same code in the same round, nobody calls the collective again.  TCP: only the root knows; it keeps
            // consuming what the others send (with no room: the records are dropped, the first code stands)
            if (status != SD_OK && c) continue;
            const char *src = rd.bytes == 0 ? nullptr : rd.inOverflow ? rd.own.data() : sendBuf() + rd.off;
            uint64_t total = 0;
            uint64_t *szs = sizes.data() + (size_t) r * nRanks;
            int rc;
            if (c) rc = roundRccl(src, rd.bytes, szs, &total);
            else if (status != SD_OK) rc = sd_tcp_gather(tcp, src, rd.bytes, nullptr, nullptr, 0, &total) == SD_EHIP ? SD_EHIP : status;
            else rc = roundTcp(src, rd.bytes, szs, &total);
            if (rc != SD_OK) {
                std::lock_guard<std::mutex> lk(mu);
                if (status == SD_OK) status = rc;
                continue;
            }
            if (rank == root) outUsed += total;
        }
        roundOff[nRounds] = outUsed;
        // the closing round: a rank whose search failed closes its stream before all its ranges have arrived -- its missing rounds went
        // out empty so that nobody waits, and this word tells the root that what it holds is not everything
        uint64_t shortBy;
        {
            std::lock_guard<std::mutex> lk(mu);
            shortBy = nSunk < nRanges ? (uint64_t) (nRanges - nSunk) : 0;
        }
        std::vector<uint64_t> flags((size_t) nRanks, 0), szs((size_t) nRanks, 0);
        uint64_t total = 0;
        int rc = SD_OK;
        if (c) {
            if (status == SD_OK) rc = sd_gather_results(c, &shortBy, sizeof(shortBy), root, szs.data(), rank == root ? flags.data() : nullptr,
                                                        rank == root ? flags.size() * sizeof(uint64_t) : 0, &total);
        } else {
            rc = sd_tcp_gather(tcp, &shortBy, sizeof(shortBy), nullptr, flags.data(), flags.size() * sizeof(uint64_t), &total);
        }
        std::lock_guard<std::mutex> lk(mu);
        if (rc != SD_OK && status == SD_OK) status = rc;
        if (status == SD_OK && rank == root)
            for (int r = 0; r < nRanks; r++)
                if (flags[(size_t) r]) {
                    status = SD_EMISMATCH;
                    incomplete = r;
                    break;
                }
    }
    int incomplete = -1;   // root: the first rank that delivered fewer ranges than it announced (SD_EMISMATCH from _wait / _end)
};

namespace {
int gatherStreamBegin(sd_comm *c, sd_tcp *tcp, int nRanks, int rank, int root, uint32_t nRanges, const uint32_t *roundOfRange, uint32_t nRounds,
                      void *outOnRoot, uint64_t outCap, int ownBuffer, sd_gather_stream **out) {
    if (!out || root < 0 || root >= nRanks || rank < 0 || rank >= nRanks || (nRanges && !roundOfRange) || nRounds == 0) return SD_EINVAL;
    for (uint32_t i = 0; i < nRanges; i++)
        if (roundOfRange[i] >= nRounds || (i && roundOfRange[i] < roundOfRange[i - 1])) return SD_EINVAL;
    sd_gather_stream *g = new sd_gather_stream();
    g->c = c;
    g->tcp = tcp;
    g->nRanks = nRanks;
    g->rank = rank;
    g->root = root;
    g->nRanges = nRanges;
    g->nRounds = nRounds;
    g->rangesThrough.assign(nRounds, 0);
    for (uint32_t i = 0; i < nRanges; i++) g->rangesThrough[roundOfRange[i]]++;
    for (uint32_t r = 1; r < nRounds; r++) g->rangesThrough[r] += g->rangesThrough[r - 1];
    g->ownOut = ownBuffer != 0;   // (the same on every rank: it decides the rounds' protocol)
    g->out = g->ownOut ? nullptr : (char *) outOnRoot;
    g->outCap = (rank == root && outOnRoot && !g->ownOut) ? outCap : 0;
    g->rounds.resize(nRounds);
    g->roundOff.assign((size_t) nRounds + 1, 0);
    g->sizes.assign((size_t) nRounds * nRanks, 0);
    // rounds without ranges on this rank are ready at once
    while (g->nReady < nRounds && g->rangesThrough[g->nReady] == 0) {
        g->rounds[g->nReady] = sd_gather_stream::Round{0, 0, false, {}};
        g->nReady++;
    }
    g->worker = std::thread([g] { g->run(); });
    *out = g;
    return SD_OK;
}
}  // namespace

int sd_gather_stream_begin(sd_comm *c, int root, uint32_t nRanges, const uint32_t *roundOfRange, uint32_t nRounds, void *outOnRoot,
                           uint64_t outCap, int ownBuffer, sd_gather_stream **out) {
    if (!c) return SD_EINVAL;
    return gatherStreamBegin(c, nullptr, c->nRanks, c->rank, root, nRanges, roundOfRange, nRounds, outOnRoot, outCap, ownBuffer, out);
}

int sd_gather_stream_begin_tcp(sd_tcp *t, int nRanks, int rank, uint32_t nRanges, const uint32_t *roundOfRange, uint32_t nRounds, void *outOnRoot,
                               uint64_t outCap, int ownBuffer, sd_gather_stream **out) {
    if (!t) return SD_EINVAL;
    return gatherStreamBegin(nullptr, t, nRanks, rank, 0, nRanges, roundOfRange, nRounds, outOnRoot, outCap, ownBuffer, out);
}

void sd_gather_stream_sink(void *gatherStream, uint32_t range, const void *records, uint64_t bytes) {
    sd_gather_stream *g = (sd_gather_stream *) gatherStream;
    (void) range;   // (ranges arrive in order: the stream finalises them in order)
    if (!g) return;
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->nSunk >= g->nRanges || g->closing) return;
    // where the current round's bytes go: the pinned send buffer while the round fits, the round's own vector otherwise
    if (!g->roundInOverflow && g->sendUsed + bytes > g->sendCap()) {
        if (g->sendUsed > g->roundBase) g->overflow.assign((const char *) g->sendBuf() + g->roundBase, (const char *) g->sendBuf() + g->sendUsed);
        else g->overflow.clear();
        g->roundInOverflow = true;
    }
    if (bytes) {
        if (g->roundInOverflow) g->overflow.insert(g->overflow.end(), (const char *) records, (const char *) records + bytes);
        else memcpy(g->sendBuf() + g->sendUsed, records, bytes);
    }
    if (!g->roundInOverflow) g->sendUsed += bytes;
    g->nSunk++;
    bool woke = false;
    while (g->nReady < g->nRounds && g->rangesThrough[g->nReady] <= g->nSunk) {   // the round (and any empty rounds behind it) is complete
        sd_gather_stream::Round &rd = g->rounds[g->nReady];
        if (g->roundInOverflow) {
            rd = sd_gather_stream::Round{0, (uint64_t) g->overflow.size(), true, {}};
            rd.own.swap(g->overflow);
            g->roundInOverflow = false;
            g->sendUsed = g->roundBase;
        } else {
            rd = sd_gather_stream::Round{g->roundBase, g->sendUsed - g->roundBase, false, {}};
        }
        g->roundBase = g->sendUsed;
        g->nReady++;
        woke = true;
    }
    lk.unlock();
    if (woke) g->cv.notify_all();
}

int sd_gather_stream_wait(sd_gather_stream *g, uint64_t *roundOffsets, uint64_t *sizes, uint64_t *totalOnRoot, const void **dataOnRoot) {
    if (!g) return SD_EINVAL;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->closing = true;   // ranges that never arrived (a failed stream) count as empty: the collectives still match
    }
    g->cv.notify_all();
    if (g->worker.joinable()) g->worker.join();
    if (roundOffsets) memcpy(roundOffsets, g->roundOff.data(), g->roundOff.size() * sizeof(uint64_t));
    if (sizes) memcpy(sizes, g->sizes.data(), g->sizes.size() * sizeof(uint64_t));
    if (totalOnRoot) *totalOnRoot = g->outUsed;
    if (dataOnRoot) *dataOnRoot = g->rank == g->root ? g->out : nullptr;
    return g->status;
}

void sd_gather_stream_destroy(sd_gather_stream *g) {
    if (!g) return;
    (void) sd_gather_stream_wait(g, nullptr, nullptr, nullptr, nullptr);
    delete g;
}

int sd_gather_stream_end(sd_gather_stream *g, uint64_t *roundOffsets, uint64_t *sizes, uint64_t *totalOnRoot) {
    if (!g) return SD_EINVAL;
    const int rc = sd_gather_stream_wait(g, roundOffsets, sizes, totalOnRoot, nullptr);
    delete g;
    return rc;
}

// Whole query genome sets -> ranks, greedy by residue count (largest first; ties: lower set index, lower rank), so that a
// set's hits stay on one rank through besthitbyset -> combinehits -> clusterhits, which group by query set.
// Deterministic: every rank computes the same assignment and keeps its own part (ascending set index).
int sd_shard_query_sets(const uint64_t *setResidues, uint32_t nSets, uint32_t world, uint32_t rank, uint32_t *mine, uint32_t *nMine) {
    if ((nSets && !setResidues) || !mine || !nMine || world == 0 || rank >= world) return SD_EINVAL;
    std::vector<uint32_t> order(nSets);
    for (uint32_t i = 0; i < nSets; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return setResidues[a] > setResidues[b]; });
    std::vector<uint64_t> load(world, 0);
    uint32_t n = 0;
    for (uint32_t x = 0; x < nSets; x++) {
        uint32_t best = 0;
        for (uint32_t r = 1; r < world; r++)
            if (load[r] < load[best]) best = r;
        load[best] += setResidues[order[x]];
        if (best == rank) mine[n++] = order[x];
    }
    std::sort(mine, mine + n);
    *nMine = n;
    return SD_OK;
}

}  // extern "C"
