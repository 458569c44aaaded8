// Internal declarations shared by the HIP translation units of libsdgpu.so.
#ifndef SD_COMMON_H
#define SD_COMMON_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <ctime>
#include <sys/prctl.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "spacedust_gpu.h"

struct sd_profile_entry {
    double ms = 0.0;
    uint64_t launches = 0;
};

struct sd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string lastError;
    bool profiling = false;
    std::map<std::string, sd_profile_entry> profile;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    bool biasTablesUploaded = false;   // sd_comp_bias_batch: correction tables resident in this context's workspace
    hipEvent_t evSync = nullptr;   // blocking-sync event: host threads sleep while they wait for the stream (sdStreamSync)
    uint64_t cellsFwd = 0, cellsRev = 0, cellsTb = 0;
    bool cigarPool = false;   // sd_sw_set_cigar_pool: the alignment calls return run-length text instead of backtrace letters
    uint64_t d2hRecordBytes = 0, d2hPoolBytes = 0;   // what the alignment calls brought back (sd_sw_download_bytes)
    hipDeviceProp_t prop;
    // grow-only device workspace: hipMalloc/hipFree per call cost far more than the kernels they serve
    struct WsEntry { void *p = nullptr; size_t bytes = 0; };
    std::map<std::string, WsEntry> ws;
    std::map<std::string, WsEntry> pinned;
    // kernel timing without stalling the launching thread: event pairs are recorded around a launch and read back later
    // (sdProfDrain), at a point where the stream is waited for anyway
    struct ProfPending { std::string name; hipEvent_t a = nullptr, b = nullptr; };
    std::vector<ProfPending> profPending;
    std::vector<hipEvent_t> evPool;
    // device buffers of destroyed sequence sets, kept for the next one (a pipeline makes and drops a query set per chunk; hipFree
    // waits for every stream of the device, i.e. for the other lanes' queued work)
    struct PoolBuf { void *p = nullptr; size_t bytes = 0; };
    std::vector<PoolBuf> pool;
    std::mutex poolMutex;
    // pinned bounce area of the small device -> host reads of a call (sdD2H): blocks are kept, a wait delivers and rewinds
    struct BounceBlock { uint8_t *p = nullptr; size_t cap = 0; };
    struct BounceItem { void *dst = nullptr; const uint8_t *src = nullptr; size_t bytes = 0; };
    std::vector<BounceBlock> bounce;
    std::vector<BounceItem> bounceItems;
    size_t bounceCur = 0, bounceUsed = 0;
};

// a device buffer of at least `bytes` from the context's pool (best fit), else a fresh one with a quarter of slack
inline hipError_t poolGet(sd_ctx *ctx, size_t bytes, void **out, size_t *got) {
    {
        std::lock_guard<std::mutex> lock(ctx->poolMutex);
        int best = -1;
        for (size_t i = 0; i < ctx->pool.size(); i++)
            if (ctx->pool[i].bytes >= bytes && (best < 0 || ctx->pool[i].bytes < ctx->pool[(size_t) best].bytes)) best = (int) i;
        if (best >= 0 && ctx->pool[(size_t) best].bytes <= 4 * bytes + (1u << 20)) {
            *out = ctx->pool[(size_t) best].p;
            *got = ctx->pool[(size_t) best].bytes;
            ctx->pool.erase(ctx->pool.begin() + best);
            return hipSuccess;
        }
    }
    const size_t grow = bytes + bytes / 4 + 256;
    const hipError_t e = hipMalloc(out, grow);
    *got = e == hipSuccess ? grow : 0;
    return e;
}

// hands a buffer back.  The pool keeps at most 24 buffers and at most POOL_BYTES_MAX bytes: beyond either the OLDEST goes (the
// pool is in hand-back order) -- the buffers a pipeline lane recycles per chunk are handed back and taken again all the time and
// so stay young, while the target-sized buffers of a destroyed set that nothing asks for again age out instead of holding HBM
// until the context is destroyed.  A single buffer larger than the budget is freed at once.
constexpr size_t POOL_BYTES_MAX = (size_t) 2 << 30;
inline size_t poolBytes(sd_ctx *ctx) {
    std::lock_guard<std::mutex> lock(ctx->poolMutex);
    size_t sum = 0;
    for (const sd_ctx::PoolBuf &b : ctx->pool) sum += b.bytes;
    return sum;
}
inline void poolPut(sd_ctx *ctx, void *p, size_t bytes) {
    if (!p) return;
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lock(ctx->poolMutex);
        if (bytes > POOL_BYTES_MAX) {
            drop.push_back(p);
        } else {
            sd_ctx::PoolBuf b;
            b.p = p;
            b.bytes = bytes;
            ctx->pool.push_back(b);
            size_t sum = 0;
            for (const sd_ctx::PoolBuf &x : ctx->pool) sum += x.bytes;
            while (ctx->pool.size() > 24 || sum > POOL_BYTES_MAX) {
                drop.push_back(ctx->pool.front().p);
                sum -= ctx->pool.front().bytes;
                ctx->pool.erase(ctx->pool.begin());
            }
        }
    }
    for (void *d : drop) (void) hipFree(d);
}

inline hipEvent_t sdProfEvent(sd_ctx *ctx) {
    if (!ctx->evPool.empty()) {
        hipEvent_t e = ctx->evPool.back();
        ctx->evPool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void) hipEventCreate(&e);
    return e;
}

// Fold the finished event pairs into the profile (a stream completes its events in order: the first unfinished pair ends
// the walk); wait = true blocks until every pair has finished.
inline void sdProfDrain(sd_ctx *ctx, bool wait) {
    size_t done = 0;
    for (; done < ctx->profPending.size(); done++) {
        sd_ctx::ProfPending &pp = ctx->profPending[done];
        if (wait) (void) hipEventSynchronize(pp.b);
        else if (hipEventQuery(pp.b) != hipSuccess) break;
        float ms = 0;
        if (hipEventElapsedTime(&ms, pp.a, pp.b) == hipSuccess) {
            sd_profile_entry &e = ctx->profile[pp.name];
            e.ms += ms;
            e.launches += 1;
        }
        ctx->evPool.push_back(pp.a);
        ctx->evPool.push_back(pp.b);
    }
    if (done) ctx->profPending.erase(ctx->profPending.begin(), ctx->profPending.begin() + (long) done);
}

// Wait for the context's stream without burning a core.  hipStreamSynchronize -- and hipEventSynchronize even on a
// hipEventBlockingSync event (measured, ROCm 7.2: thread CPU time = wall time) -- busy-wait; a pipeline keeps two such
// threads per GPU waiting most of the time, and on a box whose CPU quota is a couple of cores per GPU that starves the
// host stages (OpenMP teams).  So: record an event, poll it, and sleep ~20 us between polls after a short spin.
inline hipError_t sdEventWait(hipEvent_t ev) {
    for (int i = 0; i < 64; i++) {   // a few microseconds: results that are (almost) ready
        hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
    }
    static thread_local bool slackSet = false;
    if (!slackSet) {   // 1 us of timer slack instead of the default 50 us for this thread's short sleeps
        prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);
        slackSet = true;
    }
    // 20 us between polls for short waits, backing off to 100 us: a lane thread spends most of its life here, and every poll is
    // a system call plus a runtime query (at 20 us throughout, four waiting lanes cost most of a core)
    for (int n = 0;; n++) {
        timespec ts = {0, n < 8 ? 20000 : (n < 24 ? 50000 : 100000)};
        nanosleep(&ts, nullptr);
        hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
    }
}
struct sd_ctx;
inline void sdProfDrain(sd_ctx *ctx, bool wait);
inline hipError_t sdStreamSyncRaw(sd_ctx *ctx);
inline hipError_t sdStreamSync(sd_ctx *ctx) {
    const hipError_t e = sdStreamSyncRaw(ctx);
    sdProfDrain(ctx, false);   // everything recorded before this point has finished
    return e;
}

// Device -> host read of a call's control values (counts, flags, totals, per-query tables) on the context's stream.
// hipMemcpyAsync into pageable memory -- a stack variable, a std::vector, the caller's array -- is not asynchronous: the runtime
// waits inside the call for everything queued on the stream before it, in a busy loop, so a lane thread spent its kernels' whole
// run time there at full CPU (measured: 0.5 core-seconds per lane and 0.6-s step, whatever sdEventWait's back-off).  sdD2H copies
// into a pinned bounce block of the context instead, which returns at once, and notes the destination; the next sdStreamSync --
// the sleeping wait -- moves the bytes to where they belong.  The destination must stay valid until that wait.
inline hipError_t sdD2H(sd_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    const size_t need = (bytes + 63) & ~(size_t) 63;
    for (;;) {
        if (ctx->bounceCur < ctx->bounce.size()) {
            if (ctx->bounceUsed + need <= ctx->bounce[ctx->bounceCur].cap) break;
            ctx->bounceCur++;
            ctx->bounceUsed = 0;
            continue;
        }
        sd_ctx::BounceBlock b;
        b.cap = std::max<size_t>(need, (size_t) 1 << 20);
        const hipError_t e = hipHostMalloc((void **) &b.p, b.cap, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        ctx->bounce.push_back(b);
    }
    uint8_t *at = ctx->bounce[ctx->bounceCur].p + ctx->bounceUsed;
    const hipError_t e = hipMemcpyAsync(at, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) return e;
    sd_ctx::BounceItem it;
    it.dst = dst;
    it.src = at;
    it.bytes = bytes;
    ctx->bounceItems.push_back(it);
    ctx->bounceUsed += need;
    return hipSuccess;
}

// at the entry of a call: reads a failed call left pending point at its dead stack frame -- drop them (their copies into the bounce area
// are harmless, and the stream keeps later copies into the same bytes behind them)
inline void sdD2HReset(sd_ctx *ctx) {
    ctx->bounceItems.clear();
    ctx->bounceCur = 0;
    ctx->bounceUsed = 0;
}

inline hipError_t sdStreamSyncRaw(sd_ctx *ctx) {
    hipError_t e;
    if (ctx->evSync == nullptr) {
        e = hipStreamSynchronize(ctx->stream);
    } else {
        e = hipEventRecord(ctx->evSync, ctx->stream);
        if (e == hipSuccess) e = sdEventWait(ctx->evSync);
    }
    if (!ctx->bounceItems.empty()) {
        if (e == hipSuccess)
            for (const sd_ctx::BounceItem &it : ctx->bounceItems) memcpy(it.dst, it.src, it.bytes);
        ctx->bounceItems.clear();
        ctx->bounceCur = 0;
        ctx->bounceUsed = 0;
    }
    return e;
}

// persistent device buffer `key` of at least count elements (contents undefined after growth)
template <typename T>
hipError_t wsGet(sd_ctx *ctx, const char *key, size_t count, T **out) {
    sd_ctx::WsEntry &e = ctx->ws[key];
    const size_t need = std::max<size_t>(count, 1) * sizeof(T);
    if (e.bytes < need) {
        static const bool dbgWs = getenv("SD_DEBUG_WS") != nullptr;
        if (dbgWs) fprintf(stderr, "[ws] t=%.3f %s: %zu -> %zu bytes%s\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), key, e.bytes, need, e.p ? " (hipFree: waits for the device)" : "");
        if (e.p) (void) hipFree(e.p);
        e.p = nullptr;
        e.bytes = 0;
        // slack against regrowth (a hipFree waits for the device): a quarter, an eighth for the buffers of a gigabyte and more
        const size_t grow = need + (need >= (1ull << 30) ? need / 8 : need / 4) + 256;
        hipError_t err = hipMalloc(&e.p, grow);
        if (err != hipSuccess) return err;
        e.bytes = grow;
    }
    *out = (T *) e.p;
    return hipSuccess;
}

// persistent pinned host staging buffer
template <typename T>
hipError_t pinGet(sd_ctx *ctx, const char *key, size_t count, T **out) {
    sd_ctx::WsEntry &e = ctx->pinned[key];
    const size_t need = std::max<size_t>(count, 1) * sizeof(T);
    if (e.bytes < need) {
        static const bool dbgWs = getenv("SD_DEBUG_WS") != nullptr;
        if (dbgWs) fprintf(stderr, "[ws] t=%.3f pinned %s: %zu -> %zu bytes%s\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(), key, e.bytes, need, e.p ? " (hipHostFree)" : "");
        if (e.p) (void) hipHostFree(e.p);
        e.p = nullptr;
        e.bytes = 0;
        const size_t grow = need + need / 4 + 256;
        hipError_t err = hipHostMalloc(&e.p, grow, hipHostMallocDefault);
        if (err != hipSuccess) return err;
        e.bytes = grow;
    }
    *out = (T *) e.p;
    return hipSuccess;
}

struct sd_seqset {
    sd_ctx *ctx = nullptr;
    uint32_t n = 0;
    uint64_t total = 0;
    uint8_t *dRes = nullptr;      // residues
    int8_t *dBias = nullptr;      // SW composition bias (int8 per residue)
    int8_t *dProf = nullptr;      // profile sets only: int8 [position][21] alignment profile (replaces matrix row + bias)
    std::vector<int8_t> hProf;        // profile sets only: host copy of the alignment profile (scoreIdentical)
    std::vector<int32_t> hProfBias;   // profile sets only: per profile |min(0, min score)| (ssw_init, StripedSmithWaterman.cpp:1276-1287)
    uint64_t *dOff = nullptr;     // offsets n+1
    size_t bRes = 0, bBias = 0, bProf = 0, bOff = 0;   // sizes of the device buffers (they come from and return to the context's pool)
    std::vector<uint64_t> hOff;   // host copy of the offsets
    std::vector<int32_t> hMinBias;// per sequence min(0, min cb8)
    int32_t maxEntryAdd = 0;      // largest composition bias of the set (sequences) / largest profile entry (profile sets)
    std::vector<uint8_t> hRes;    // host copy (traceback identity count, scoreIdentical)
    std::vector<int8_t> hBias;
    uint32_t *dGroupOf = nullptr, *dGroupKey = nullptr;   // sd_seqset_set_groups: target set and DB key of every sequence
    size_t bGroupOf = 0, bGroupKey = 0;
    uint32_t nGroups = 0;
};

// the target side of the prefilter, resident in HBM (built by sd_target_create* from host arrays or by sd_target_build on the device)
struct sd_target {
    sd_ctx *ctx = nullptr;
    int k = 6;
    uint32_t nSeq = 0;
    uint64_t nEntries = 0;
    uint64_t tableSize = 0;
    uint32_t *dOffsets = nullptr;
    uint64_t *dBlockBase = nullptr;  // wide indexes (>= 2^32 entries): list i starts at dBlockBase[i >> 16] + dOffsets[i]
    uint32_t *dEntrySeq = nullptr;   // upload staging only (freed after the interleaved copy is built)
    uint16_t *dEntryPos = nullptr;
    uint2 *dEntries = nullptr;       // (seqId, position) per index entry, 8 B: one sector per short list instead of two
    uint8_t *dMasked = nullptr;
    uint64_t *dSeqOff = nullptr;
    int16_t *dExt3Score = nullptr;
    // countGE(row, cutoff) of a 3-mer row as ONE table read: dExt3Cum[row * 256 + (cutoff - ext3Lo)] = entries of the row with a score >= cutoff
    // (sdBuildExt3Cum; null when the scores span more than 255 values: the kernels then search the sorted row)
    uint16_t *dExt3Cum = nullptr;
    int ext3Lo = 0;
    uint16_t *dExt3Index = nullptr;
    int16_t *dExt2Score = nullptr;
    uint16_t *dExt2Index = nullptr;
    std::vector<uint64_t> hSeqOff;
};

int sdFail(sd_ctx *ctx, int code, const char *fmt, ...);
int sdBuildExt3Cum(sd_ctx *ctx, sd_target *t);   // sd_prefilter.hip; after dExt3Score is resident

// (a failed call returns from here with reads of sdD2H possibly still queued: their destinations -- stack variables, local
// vectors of the returning function -- die with this return, so the queue is dropped before any later wait could deliver them)
#define SD_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            if ((ctx) != nullptr) sdD2HReset(ctx);                                                 \
            return sdFail((ctx), SD_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        }                                                                                          \
    } while (0)

// RAII device buffer
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void) hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc((void **) &p, count * sizeof(T));
    }
};

// profiling helpers: bracket a launch with events on ctx->stream
struct ProfScope {
    sd_ctx *ctx;
    const char *name;
    hipEvent_t a = nullptr;
    ProfScope(sd_ctx *c, const char *n) : ctx(c), name(n) {
        if (ctx->profiling) {
            a = sdProfEvent(ctx);
            (void) hipEventRecord(a, ctx->stream);
        }
    }
    ~ProfScope() {
        if (a) {
            sd_ctx::ProfPending pp;
            pp.name = name;
            pp.a = a;
            pp.b = sdProfEvent(ctx);
            (void) hipEventRecord(pp.b, ctx->stream);
            ctx->profPending.push_back(pp);   // read back by sdProfDrain at the next stream wait: the launching thread does not stall
        }
    }
};


// wall-clock phases on the host side of a call (reported next to the kernel times as "host:<name>")
#include <chrono>
struct HostScope {
    sd_ctx *ctx;
    std::string name;
    std::chrono::steady_clock::time_point t0;
    HostScope(sd_ctx *c, const char *n) : ctx(c), name(std::string("host:") + n), t0(std::chrono::steady_clock::now()) {}
    ~HostScope() {
        if (!ctx->profiling) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        sd_profile_entry &e = ctx->profile[name];
        e.ms += ms;
        e.launches += 1;
    }
};

#endif
