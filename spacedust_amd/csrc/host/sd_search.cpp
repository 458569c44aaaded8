// sd_search: the C++ host driver of the streaming clustersearch pipeline (include/spacedust_gpu.h, "the whole search in
// one object").  The reference runs the same work as separate processes with DBs in between
// (R/data/clustersearch.sh:110-146: search = prefilter + align, prefixid, besthitbyset, mergeresultsbyset, combinehits,
// clusterhits); here the stages of consecutive query chunks overlap on stage threads and two HIP streams:
//     bias(i+2)  |  prefilter + pair list (i+1)  |  alignments (i)  |  aggregation (i-1)
// Every stage is a call of the C ABI (sd_prefilter_batch, sd_sw_align_batch_compact, sd_agg_*, sd_clusterhits_batch);
// nothing is computed here.
#include "sd_host.h"
#include "spacedust_gpu.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <pthread.h>
#include <vector>

namespace {

double nowSec() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// CPU time of the calling thread: what a stage thread itself burns (staging copies, pair lists, record building, launch
// overhead); OpenMP workers of the host stages it calls are not in it
double threadCpuSec() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

// one thread executing submitted jobs in order (a stage of the pipeline)
class StageThread {
public:
    // name: what /proc/<pid>/task/<tid>/comm shows (at most 15 characters); threads a stage thread creates -- the OpenMP team of the
    // host stages it calls -- inherit it, so a per-name CPU table (bench.py: host_cpu_threads) is a per-stage table
    explicit StageThread(const char *name = "sd-stage") : stop_(false), name_(name), th_([this] { run(); }) {}
    ~StageThread() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    template <class F>
    auto submit(F f) -> std::future<decltype(f())> {
        typedef decltype(f()) R;
        std::shared_ptr<std::packaged_task<R()> > task(new std::packaged_task<R()>(std::move(f)));
        std::future<R> fut = task->get_future();
        {
            std::lock_guard<std::mutex> l(m_);
            q_.push_back([task] { (*task)(); });
        }
        cv_.notify_one();
        return fut;
    }

private:
    void run() {
        pthread_setname_np(pthread_self(), name_.c_str());
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()> > q_;
    bool stop_;
    std::string name_;
    std::thread th_;
};

int hostCpus() {
    long t = 0;
    std::ifstream f("/sys/fs/cgroup/cpu.max");
    std::string q, p;
    if (f >> q >> p && q != "max") {
        const double quota = strtod(q.c_str(), nullptr), period = strtod(p.c_str(), nullptr);
        if (quota > 0 && period > 0) t = (long) (quota / period + 0.5);
    }
    const unsigned hw = std::thread::hardware_concurrency();
    if (t <= 0 || (hw > 0 && t > (long) hw)) t = hw > 0 ? (long) hw : 1;
    if (const char *e = getenv("SD_CPUS")) t = std::max(1, atoi(e));
    return (int) std::max(1L, t);
}

struct BiasOut {
    uint32_t c0 = 0, c1 = 0;
    std::vector<uint64_t> off;
    std::vector<int8_t> sw, dg;
    std::vector<int16_t> km;
    double seconds = 0, cpu = 0;
    int rc = SD_OK;
};

struct PfOut {
    std::unique_ptr<BiasOut> bias;
    std::vector<sd_hit> hits;
    std::vector<uint32_t> counts;
    std::vector<uint64_t> stats;   // 4 per query
    std::vector<uint32_t> pairQ, pairT;
    std::vector<uint16_t> pairDiag;   // the prefilter's diagonal of every pair (sd_sw_align_batch_compact_diag)
    uint64_t nPairs = 0, notComputed = 0;
    double tPrefilter = 0, tPairs = 0, cpu = 0;
    int rc = SD_OK;
    std::string err;
};

enum { T_INDEX, T_UPLOAD, T_BIAS, T_PREFILTER, T_PAIRS, T_SEQSET, T_ALIGN, T_AGG_WAIT, T_AGG_BUSY, T_CLUSTERHITS, T_PF_WAIT, T_TOTAL,
       T_CPU_BIAS, T_CPU_PF, T_CPU_ALIGN, T_CPU_AGG_MAIN,   // thread CPU seconds of the stage threads (bias | prefilter lanes | alignment lanes | aggregation + the driving thread)
       T_N = 16 };
enum { S_KMERS, S_INDEX_HITS, S_DIAGONALS, S_DIAG_LEN, S_PREF_HITS, S_PAIRS, S_CELLS_FWD, S_CELLS_REV, S_CELLS_TB, S_ENTRIES, S_MASKED, S_K,
       S_KMER_THR, S_BIN, S_NOT_COMPUTED, S_EVAL_PUSHDOWN /* 1: the alignments ran with combinehits' E-value bound as their gate */, S_N = 16 };

}  // namespace

struct sd_search_result {
    sd_agg *agg = nullptr;
    uint64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<uint64_t> entryOff;
    std::vector<uint32_t> entryQ, entryT, hitQ, hitT, clusterOf, rank, nClusters, cSize;
    std::vector<double> pval, pCO, pMH;
    std::vector<char> records;   // the range's cluster records, built inside the stream (sd_search_set_want_records)
    bool haveRecords = false;
    ~sd_search_result() {
        if (agg) sd_agg_destroy(agg);
    }
};

struct sd_search {
    sd_search_params par;
    sd_setdb T;
    int device = 0;
    sd_host *host = nullptr;
    sd_ctx *ctxPf = nullptr, *ctxAl = nullptr, *ctxBias = nullptr;
    sd_ctx *ctxAl2 = nullptr;   // second alignment lane: consecutive chunks are aligned concurrently, each on its own stream
    sd_ctx *ctxCh = nullptr;    // clusterhits (the main thread finalises ranges while both lanes may be busy)
    sd_ctx *ctxPf2 = nullptr;   // second prefilter lane (same target index, its own workspace and stream)
    sd_ctx *ctxPfMore[2] = {nullptr, nullptr}, *ctxAlMore[2] = {nullptr, nullptr};   // lanes 3 and 4 (SD_PF_LANES / SD_ALIGN_LANES up to 4)
    int alignLanes = 2, pfLanes = 2;
    sd_host_index *index = nullptr;
    sd_target *target = nullptr;
    sd_seqset *tSeqs = nullptr;
    int k = 6, kmerThr = 0;
    sd_prefilter_params pfPar;
    sd_sw_params swPar;
    sd_sw_params swParRun;   // swPar with the E-value gate of the running stream (sd_search_stream: predicate pushdown)
    bool targetHasGroups = false;   // sd_seqset_set_groups was applied to tSeqs
    bool bestOnDevice = false;      // the running stream lets the device keep only besthitbyset's candidates
    bool cigarOnDevice = false;     // ... and returns run-length text instead of backtrace letters
    bool wantRecords = false;       // sd_search_set_want_records
    sd_ch_params chPar;
    std::vector<int32_t> tLen;
    std::vector<uint32_t> tSetSize;
    std::vector<double> lgamma;
    uint64_t stats[S_N];
    double seconds[T_N];
    std::string err;
    sd_pref_sink prefSink = nullptr;
    sd_aln_sink alnSink = nullptr;
    sd_records_sink recordsSink = nullptr;   // sd_search_set_records_sink
    void *recordsSinkUser = nullptr;
    void *sinkUser = nullptr;
    // alignment result buffers in a ring: up to two chunks being aligned (one per lane) while the aggregation reads the
    // chunk before them
    struct AlnBuf {
        std::vector<sd_sw_result> res;
        std::vector<uint32_t> idx, pq, pt;
        std::vector<uint8_t> ident;
        std::vector<char> pool;
    } buf[8];   // one per alignment lane + the chunks the aggregation may still read
    int flip = 0;

    ~sd_search() {
        if (tSeqs) sd_seqset_destroy(tSeqs);
        if (target) sd_target_destroy(target);
        if (index) sd_host_index_destroy(index);
        if (ctxBias) sd_ctx_destroy(ctxBias);
        for (int x = 0; x < 2; x++) {
            if (ctxPfMore[x]) sd_ctx_destroy(ctxPfMore[x]);
            if (ctxAlMore[x]) sd_ctx_destroy(ctxAlMore[x]);
        }
        if (ctxPf2) sd_ctx_destroy(ctxPf2);
        if (ctxCh) sd_ctx_destroy(ctxCh);
        if (ctxAl2) sd_ctx_destroy(ctxAl2);
        if (ctxAl) sd_ctx_destroy(ctxAl);
        if (ctxPf) sd_ctx_destroy(ctxPf);
        if (host) sd_host_destroy(host);
    }
    int fail(int rc, const std::string &what, sd_ctx *ctx = nullptr) {
        err = what + " failed (" + std::to_string(rc) + ")";
        if (ctx) err += std::string(": ") + sd_last_error(ctx);
        return rc;
    }
};

extern "C" {

void sd_search_default_params(sd_search_params *p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->sensitivity = 5.7f;   // R/src/workflow/clustersearch.cpp:9-37
    p->kmerSize = 0;
    p->maxSeqs = 300;
    p->minDiagScore = 15;
    p->binSize = 0;
    p->mask = 1;
    p->maskProb = 0.9;
    p->compBiasCorr = 1;
    p->evalThr = 10.0;
    p->covMode = 2;
    p->covThr = 0.8f;
    p->alnLenThr = 30;
    p->maxGeneGap = 3;
    p->clusterSize = 2;
    p->alpha = 1.0;
    p->pCluThr = 0.01f;
    p->pMHThr = 0.01f;
    p->filterSelfMatch = 0;
    p->profileQueries = 0;
    p->chunkQueries = 0;   // by the size of the target set (sd_search_stream)
    p->deviceBias = -1;
    p->threads = 0;
    p->alignPriority = 1;
}

int sd_search_create(int device, const sd_search_params *par, const sd_setdb *target, sd_search **out) {
    return sd_search_create_indexed(device, par, target, nullptr, out);
}

int sd_search_create_indexed(int device, const sd_search_params *par, const sd_setdb *target, const sd_index_view *view, sd_search **out) {
    if (!par || !target || !out || !target->residues || !target->offsets) return SD_EINVAL;
    if (view && (!view->kmerOffsets || !view->entrySeq || !view->entryPos || !view->maskedResidues)) return SD_EINVAL;
    std::unique_ptr<sd_search> s(new sd_search());
    s->par = *par;
    s->T = *target;
    s->device = device;
    memset(s->stats, 0, sizeof(s->stats));
    memset(s->seconds, 0, sizeof(s->seconds));
    const int cpus = hostCpus();
    const int threads = par->threads > 0 ? par->threads : cpus;
    int rc = sd_host_create(threads, &s->host);
    if (rc != SD_OK) return rc;
    const int pfPrio = getenv("SD_PF_PRIO") ? atoi(getenv("SD_PF_PRIO")) : 0;   // stream priority of the prefilter lanes (-1 highest; measured: no effect on the throughput)
    rc = sd_ctx_create_prio(device, pfPrio, &s->ctxPf);
    if (rc != SD_OK) return rc;
    // CUs the alignment streams leave alone (see sd_ctx_create_masked)
    const int alReserve = getenv("SD_ALIGN_RESERVE_CUS") ? atoi(getenv("SD_ALIGN_RESERVE_CUS")) : 0;
    rc = sd_ctx_create_masked(device, par->alignPriority, alReserve, &s->ctxAl);
    if (rc != SD_OK) return rc;
    // (two lanes per stage whatever the CPU quota: a lane thread sleeps while its kernels run -- 0.04 - 0.08 core-seconds per step
    // since the small reads go through sdD2H; before that a second lane cost 0.7 and ranks with fewer than 4 cores ran one)
    if (const char *e = getenv("SD_ALIGN_LANES")) s->alignLanes = std::max(1, std::min(4, atoi(e)));
    if (s->alignLanes > 1) {
        rc = sd_ctx_create_masked(device, par->alignPriority, alReserve, &s->ctxAl2);
        if (rc != SD_OK) return rc;
    }
    for (int x = 2; x < s->alignLanes; x++) {
        rc = sd_ctx_create_masked(device, par->alignPriority, alReserve, &s->ctxAlMore[x - 2]);
        if (rc != SD_OK) return rc;
    }
    rc = sd_ctx_create(device, &s->ctxCh);
    if (rc != SD_OK) return rc;
    // prefilter lanes: SD_PF_LANES fixes the number; otherwise four contexts are made (a context is a stream and an empty workspace)
    // and the number in use is decided once the target is resident (below)
    const bool pfLanesFixed = getenv("SD_PF_LANES") != nullptr;
    if (const char *e = getenv("SD_PF_LANES")) s->pfLanes = std::max(1, std::min(4, atoi(e)));
    else s->pfLanes = 4;
    if (s->pfLanes > 1) {
        rc = sd_ctx_create_prio(device, pfPrio, &s->ctxPf2);
        if (rc != SD_OK) return rc;
    }
    for (int x = 2; x < s->pfLanes; x++) {
        rc = sd_ctx_create_prio(device, pfPrio, &s->ctxPfMore[x - 2]);
        if (rc != SD_OK) return rc;
    }
    // composition bias: on the device unless the caller asks for the host stage (deviceBias == 0).  The two kernels cost 0.13 ms per
    // chunk of 10 000 queries; the host stage the same chunk 0.4 core-seconds of an OpenMP team (bit-identical by construction,
    // tests/test_gpu_pipeline.py::test_device_composition_bias_equals_host)
    const bool devBias = par->deviceBias != 0;
    if (devBias && par->compBiasCorr && !par->profileQueries) {
        rc = sd_ctx_create(device, &s->ctxBias);
        if (rc != SD_OK) return rc;
    }
    const uint64_t tRes = target->offsets[target->n];
    s->k = par->kmerSize ? par->kmerSize : sd_host_auto_kmer_size(tRes);
    if (s->k != 6 && s->k != 7) return SD_EUNSUPPORTED;
    s->kmerThr = par->profileQueries ? sd_host_profile_kmer_threshold(par->sensitivity, s->k) : sd_host_kmer_threshold(par->sensitivity, s->k);
    const int indexThr = par->profileQueries ? 0 : s->kmerThr;   // profile searches index every k-mer (Prefiltering.cpp:525-527)
    if (view && (view->kmerSize != s->k || view->kmerThr != indexThr)) return SD_EINVAL;
    double t0 = nowSec();
    uint64_t nEntries = 0, masked = 0;
    const uint32_t *kOff = nullptr, *eSeq = nullptr;
    const uint16_t *ePos = nullptr;
    const uint8_t *mres = nullptr;
    const uint64_t *kBase = nullptr;   // block bases of a wide index (>= 2^32 entries)
    if (view) {
        kBase = view->kmerBlockBase;
        kOff = view->kmerOffsets;
        eSeq = view->entrySeq;
        ePos = view->entryPos;
        mres = view->maskedResidues;
        nEntries = view->nEntries;
        masked = view->nMaskedResidues;
    } else if (!getenv("SD_INDEX_HOST")) {
        // the index is built where it is used: masking, k-mer lists and list starts on the device (sd_target_build)
        const int16_t *s2, *s3;
        const uint16_t *i2, *i3;
        uint32_t z2, z3;
        sd_host_ext_matrix(s->host, 2, &s2, &i2, &z2);
        sd_host_ext_matrix(s->host, 3, &s3, &i3, &z3);
        double ratios[21 * 21];
        int8_t self[21];
        sd_host_index_tables(s->host, ratios, self);
        uint64_t st[4] = {0, 0, 0, 0};
        rc = sd_target_build(s->ctxPf, s->k, indexThr, par->mask ? 1 : 0, par->maskProb, target->residues, target->offsets, target->n, ratios,
                             self, s2, i2, s3, i3, &s->target, st);
        if (rc != SD_OK) return rc;
        nEntries = st[0];
        masked = st[1];
        kOff = nullptr;
    } else {
        rc = sd_host_index_build(s->host, target->residues, target->offsets, target->n, s->k, indexThr, par->mask ? 1 : 0, par->maskProb,
                                 &s->index);
        if (rc != SD_OK) return rc;
        uint64_t tableSize = 0;
        sd_host_index_info(s->index, &tableSize, &nEntries, &masked);
        sd_host_index_arrays(s->index, &kOff, &eSeq, &ePos, &mres);
        sd_host_index_block_base(s->index, &kBase, nullptr);
    }
    s->seconds[T_INDEX] = nowSec() - t0;
    s->stats[S_ENTRIES] = nEntries;
    s->stats[S_MASKED] = masked;
    s->stats[S_K] = (uint64_t) s->k;
    s->stats[S_KMER_THR] = (uint64_t) s->kmerThr;
    t0 = nowSec();
    if (!s->target) {
        const int16_t *s2, *s3;
        const uint16_t *i2, *i3;
        uint32_t z2, z3;
        sd_host_ext_matrix(s->host, 2, &s2, &i2, &z2);
        sd_host_ext_matrix(s->host, 3, &s3, &i3, &z3);
        rc = sd_target_create_wide(s->ctxPf, s->k, kOff, kBase, eSeq, ePos, nEntries, mres, target->offsets, target->n, s2, i2, s3, i3,
                                   &s->target);
        if (rc != SD_OK) return rc;
    }
    if (s->index) {   // the host copy is not needed once the target is resident
        sd_host_index_destroy(s->index);
        s->index = nullptr;
    }
    rc = sd_seqset_create(s->ctxAl, target->residues, target->offsets, target->n, nullptr, &s->tSeqs);
    if (rc != SD_OK) return rc;
    if (target->setId && target->nSets > 0) {   // besthitbyset on the device needs the set (and DB key) of every target beside the sequences
        rc = sd_seqset_set_groups(s->tSeqs, target->setId, target->nSets, target->keys);
        if (rc != SD_OK) return rc;
        s->targetHasGroups = true;
    }
    s->seconds[T_UPLOAD] = nowSec() - t0;
    {   // a target that fills most of the device (10 000 proteomes: 84 GB of index + sequences) leaves room for one prefilter
        // and one alignment workspace, not two of each
        uint64_t freeB = 0, totalB = 0;
        const bool tight = sd_device_memory(s->ctxPf, &freeB, &totalB) == SD_OK && totalB > 0 && freeB < totalB / 4 * 3;
        if (tight) s->pfLanes = s->alignLanes = 1;
        // Proteome-scale target sets (10^6 sequences and more): a prefilter call is a chain of ~17 sub-batches of ~700 queries with a
        // dozen host synchronisation points each, and two lanes leave the stage idle 28 % of the time while it is what bounds a step
        // (measured at 1 000 proteomes, 12 steps: 2 lanes 2 200 - 2 220, 3 lanes 2 308, 4 lanes 2 099 genome-pairs/s).  Smaller targets:
        // two (100 proteomes: 2 x 2 and 3 x 2 lanes within noise, the third workspace is not worth its 9 GB)
        else if (!pfLanesFixed) s->pfLanes = target->n >= 1000000u ? 4 : 2;   // (four lanes with the chunks of 2 500 queries such a target gets: 2 628 vs 2 575 genome-pairs/s with three)
    }
    memset(&s->pfPar, 0, sizeof(s->pfPar));
    s->pfPar.kmerSize = s->k;
    s->pfPar.kmerThr = s->kmerThr;
    s->pfPar.maxHitsPerQuery = (int32_t) std::max<int64_t>(1, std::min<int64_t>(par->maxSeqs, target->n));
    s->pfPar.minDiagScore = par->minDiagScore;
    s->pfPar.binSize = par->binSize ? par->binSize : sd_host_bin_size(target->n, 0);
    s->pfPar.covMode = par->covMode;
    s->pfPar.covThr = (par->covMode == 0 || par->covMode == 2 || par->covMode == 5) ? par->covThr : 0.0f;
    sd_host_matrix(s->host, 2, s->pfPar.ungappedMatrix, nullptr, nullptr);
    s->stats[S_BIN] = s->pfPar.binSize;
    memset(&s->swPar, 0, sizeof(s->swPar));
    s->swPar.gapOpen = 11;
    s->swPar.gapExtend = 1;
    sd_host_matrix(s->host, 0, s->swPar.matrix, nullptr, nullptr);
    s->swPar.covMode = par->covMode;
    s->swPar.covThr = par->covThr;
    s->swPar.evalThr = par->evalThr;
    s->swPar.swMode = 2;
    s->swPar.dbResidues = tRes;
    s->chPar.maxGeneGap = par->maxGeneGap;
    s->chPar.clusterSize = par->clusterSize;
    s->chPar.alpha = par->alpha;
    s->chPar.pCluThr = par->pCluThr;
    s->chPar.pMHThr = par->pMHThr;
    s->tLen.resize(target->n);
    for (uint32_t i = 0; i < target->n; i++) s->tLen[i] = (int32_t) (target->offsets[i + 1] - target->offsets[i]);
    s->tSetSize.assign(target->nSets, 0);
    if (target->setId)
        for (uint32_t i = 0; i < target->n; i++)
            if (target->setId[i] < target->nSets) s->tSetSize[target->setId[i]]++;
    *out = s.release();
    return SD_OK;
}

void sd_search_destroy(sd_search *s) { delete s; }
const char *sd_search_last_error(sd_search *s) { return s ? s->err.c_str() : ""; }
const sd_target *sd_search_target(sd_search *s) { return s ? s->target : nullptr; }

sd_ctx *sd_search_ctx(sd_search *s, int which) {
    if (!s) return nullptr;
    switch (which) {
        case 0: return s->ctxPf;
        case 1: return s->ctxAl;
        case 2: return s->ctxBias;
        case 3: return s->ctxAl2;
        case 4: return s->ctxCh;
        case 5: return s->ctxPf2;
        case 6: return s->ctxPfMore[0];
        case 7: return s->ctxPfMore[1];
        case 8: return s->ctxAlMore[0];
        case 9: return s->ctxAlMore[1];
        default: return nullptr;
    }
}

int sd_search_set_sinks(sd_search *s, sd_pref_sink pref, sd_aln_sink aln, void *user) {
    if (!s) return SD_EINVAL;
    s->prefSink = pref;
    s->alnSink = aln;
    s->sinkUser = user;
    return SD_OK;
}

int sd_search_set_chunk_queries(sd_search *s, int32_t chunkQueries) {
    if (!s || chunkQueries < 0) return SD_EINVAL;   // 0: the library's choice (sd_search_stream)
    s->par.chunkQueries = chunkQueries;
    return SD_OK;
}

// on != 0: sd_search_stream builds every range's cluster records when the range is finalised -- inside the pipeline, on the thread that is
// otherwise waiting for the lanes -- so that sd_search_result_records is a copy.  (A rank of a multi-GPU run needs them all for the
// final gather; built behind the stream they were seconds of host time after the last kernel.)
int sd_search_set_records_sink(sd_search *s, sd_records_sink fn, void *user) {
    if (!s) return SD_EINVAL;
    s->recordsSink = fn;
    s->recordsSinkUser = user;
    return SD_OK;
}

int sd_search_set_want_records(sd_search *s, int on) {
    if (!s) return SD_EINVAL;
    s->wantRecords = on != 0;
    return SD_OK;
}

int sd_search_stats(sd_search *s, uint64_t *stats, double *seconds) {
    if (!s) return SD_EINVAL;
    if (stats) memcpy(stats, s->stats, sizeof(s->stats));
    if (seconds) memcpy(seconds, s->seconds, sizeof(s->seconds));
    return SD_OK;
}

int sd_search_download_bytes(sd_search *s, uint64_t *recordBytes, uint64_t *poolBytes) {
    if (!s) return SD_EINVAL;
    uint64_t rec = 0, pool = 0;
    sd_ctx *al[4] = {s->ctxAl, s->ctxAl2, s->ctxAlMore[0], s->ctxAlMore[1]};
    for (sd_ctx *c : al) {
        uint64_t a = 0, b = 0;
        if (c && sd_sw_download_bytes(c, &a, &b) == SD_OK) {
            rec += a;
            pool += b;
        }
    }
    if (recordBytes) *recordBytes = rec;
    if (poolBytes) *poolBytes = pool;
    return SD_OK;
}

int sd_search_stream(sd_search *s, const sd_setdb *Q, int sameDb, uint32_t nRanges, const uint32_t *rangeBegin,
                     const uint32_t *rangeEnd, sd_search_result **results) {
    if (!s || !Q || !results || (nRanges && (!rangeBegin || !rangeEnd))) return SD_EINVAL;
    const bool profile = Q->alnProfile != nullptr;
    if (profile != (s->par.profileQueries != 0)) return s->fail(SD_EINVAL, "sd_search_stream: query DB type does not match profileQueries");
    if (profile && (!Q->sortedScore || !Q->sortedIndex)) return SD_EINVAL;
    const double tAll = nowSec();
    const sd_setdb &T = s->T;
    std::vector<int32_t> qLen(Q->n);
    for (uint32_t i = 0; i < Q->n; i++) qLen[i] = (int32_t) (Q->offsets[i + 1] - Q->offsets[i]);
    // without set membership on both sides (a plain `search`) the stream stops after the alignments: the sinks get the
    // prefilter rows and alignment records, nothing is aggregated or clustered
    const bool aggregate = Q->setId && Q->posInSet && Q->strand && T.setId && T.posInSet && T.strand;
    // Predicate pushdown of the fused workflow.  combinehits keeps a best hit only if log P(E) < log(10e-7) after its %.3E text round
    // trips (combinehits.cpp:101-113), which no alignment with E >= 1.1e-6 survives (sd_agg_add has the arithmetic).  When nothing
    // but the cluster records leaves this call -- no alignment sink, so no alignment DB is written -- the alignments therefore run
    // with 1.2e-6 as their E-value gate instead of the user's -e (10 for clustersearch): a pair above it stops after the score
    // pass, exactly as the reference stops a pair above -e (StripedSmithWaterman.cpp:389-398), and the reverse pass, the traceback,
    // the record and its backtrace are spent on the ~20 % of the pairs that can reach the result.  The cluster-hit TSV is unchanged
    // (every hit that passes combinehits has its full record; asserted by the md5s of tests/test_gpu_cli.py and, at the measured
    // size, by bench.py's back-half parity leg against the reference run with -e 10); the `accepted` counter then counts what
    // passed the tighter gate.  SD_EVAL_PUSHDOWN=0 runs every pair to the user's -e.
    const bool pushdown = aggregate && !s->alnSink && !(getenv("SD_EVAL_PUSHDOWN") && atoi(getenv("SD_EVAL_PUSHDOWN")) == 0);
    s->swParRun = s->swPar;
    if (pushdown) s->swParRun.evalThr = std::min(s->swPar.evalThr, 1.2e-6);
    s->stats[S_EVAL_PUSHDOWN] = pushdown ? 1 : 0;
    // The same streams also leave besthitbyset's choice to the device: of a query's accepted alignments only the first per target
    // set (Matcher::compareHits order) and the identity pair come back (sd_sw_align_batch_best_by_group); the aggregation's own
    // selection over those gives what it gave over all records.  SD_BEST_ON_DEVICE=0: every accepted record comes back.
    s->bestOnDevice = pushdown && s->targetHasGroups && !(getenv("SD_BEST_ON_DEVICE") && atoi(getenv("SD_BEST_ON_DEVICE")) == 0);
    // ... and Matcher::compressAlignment: with the aggregation as the only consumer of the backtraces (no alignment sink), the lanes
    // return every backtrace as its run-length text (sd_sw_set_cigar_pool) and sd_agg_add copies it.  SD_CIGAR_ON_DEVICE=0: letters.
    s->cigarOnDevice = aggregate && !s->alnSink && !(getenv("SD_CIGAR_ON_DEVICE") && atoi(getenv("SD_CIGAR_ON_DEVICE")) == 0);
    struct CigarMode {   // (the lane contexts are handed out by sd_search_context: the mode ends with the stream)
        sd_ctx *al[4];
        explicit CigarMode(sd_search *s_, bool on) : al{s_->ctxAl, s_->ctxAl2, s_->ctxAlMore[0], s_->ctxAlMore[1]} {
            for (sd_ctx *c : al)
                if (c) sd_sw_set_cigar_pool(c, on ? 1 : 0);
        }
        ~CigarMode() {
            for (sd_ctx *c : al)
                if (c) sd_sw_set_cigar_pool(c, 0);
        }
    } cigarMode(s, s->cigarOnDevice);
    std::vector<uint32_t> qSetSize(Q->nSets, 0);
    if (aggregate)
        for (uint32_t i = 0; i < Q->n; i++)
            if (Q->setId[i] < Q->nSets) qSetSize[Q->setId[i]]++;
    // lgamma table for clusterhits: set sizes and gene positions bound the indices (ClusterHits.cpp:259-271)
    if (aggregate) {
        uint32_t m = 0;
        for (uint32_t v : qSetSize) m = std::max(m, v);
        for (uint32_t v : s->tSetSize) m = std::max(m, v);
        for (uint32_t i = 0; i < Q->n; i++) m = std::max(m, Q->posInSet[i]);
        for (uint32_t i = 0; i < T.n; i++) m = std::max(m, T.posInSet[i]);
        if (s->lgamma.size() < (size_t) m + 8) {
            s->lgamma.resize((size_t) m + 8);
            sd_host_lgamma_table(s->lgamma.data(), (uint32_t) s->lgamma.size());
        }
    }
    std::vector<std::unique_ptr<sd_search_result> > res(nRanges);
    for (uint32_t r = 0; r < nRanges; r++) {
        res[r].reset(new sd_search_result());
        if (!aggregate) continue;
        int rc = sd_agg_create(Q->setId, qLen.data(), Q->n, T.setId, s->tLen.data(), T.n, Q->nSets, T.nSets, s->par.evalThr, s->par.covMode,
                               s->par.covThr, s->par.alnLenThr, s->par.filterSelfMatch, &res[r]->agg);
        if (rc != SD_OK) return s->fail(rc, "sd_agg_create");
        if (Q->keys || T.keys) sd_agg_set_keys(res[r]->agg, Q->keys, T.keys);
        if (s->cigarOnDevice) sd_agg_set_pool_form(res[r]->agg, 1);
    }
    // chunks: whole query proteins; the very first chunk of a stream is a quarter of the others (its prefilter is the one
    // stage nothing overlaps with, so the alignment thread starts that much earlier)
    struct Chunk {
        uint32_t range, c0, c1;
    };
    std::vector<Chunk> chunks;
    // Chunk size.  What a chunk costs follows its index hits, not its queries: against 3e5 target sequences (100 proteomes) a query has
    // 1.5e5 hits and a prefilter sub-batch is 4 096 queries, against 3e6 sequences it has 1.5e6 and a sub-batch (bounded by 2^30 hits) is
    // 700 queries -- there chunks of 10 000 queries keep three prefilter lanes and two alignment lanes busy with very uneven pieces
    // (a 12 000-query range is 10 000 + 2 000).  Measured on one box, 1 000 proteomes: 10 000 -> 2 200 - 2 290, 6 000 -> 2 400 - 2 500,
    // 4 000 -> 2 530 - 2 550, 2 000 -> 2 570 - 2 590 genome-pairs/s; 100 proteomes: 10 000 -> 1 935, 6 000 -> 1 890, 3 000 -> 1 685
    // (profiles/r05_experiments.txt item 14).  A range is cut into EQUAL chunks of at most that size.
    const uint32_t chunkQ = (uint32_t) (s->par.chunkQueries > 0 ? s->par.chunkQueries : (T.n >= 1000000u ? 2500 : 10000));
    for (uint32_t r = 0; r < nRanges; r++) {
        if (rangeEnd[r] > Q->n || rangeBegin[r] > rangeEnd[r]) return s->fail(SD_EINVAL, "sd_search_stream: bad range");
        uint32_t c0 = rangeBegin[r];
        if (chunks.empty() && c0 < rangeEnd[r]) {   // the stream's first chunk (see above)
            Chunk c;
            c.range = r;
            c.c0 = c0;
            c.c1 = (uint32_t) std::min<uint64_t>(rangeEnd[r], (uint64_t) c0 + std::max<uint32_t>(1, std::min(chunkQ, std::max<uint32_t>(1000, chunkQ / 4))));
            chunks.push_back(c);
            c0 = c.c1;
        }
        const uint32_t left = rangeEnd[r] - c0;
        const uint32_t parts = (left + chunkQ - 1) / chunkQ;
        for (uint32_t x = 0; x < parts; x++) {   // equal parts: the first left % parts of them one query longer
            Chunk c;
            c.range = r;
            c.c0 = c0;
            c.c1 = c0 + left / parts + (x < left % parts ? 1u : 0u);
            chunks.push_back(c);
            c0 = c.c1;
        }
    }
    std::vector<int64_t> lastChunkOf(nRanges, -1);
    for (size_t x = 0; x < chunks.size(); x++) lastChunkOf[chunks[x].range] = (int64_t) x;

    StageThread biasStage{"sd-bias"}, pfStage{"sd-pf0"}, aggStage{"sd-agg"};
    double *tm = s->seconds;

    auto biasJob = [s, Q, profile](uint32_t c0, uint32_t c1) {
        std::unique_ptr<BiasOut> o(new BiasOut());
        const double cpu0 = threadCpuSec();
        o->c0 = c0;
        o->c1 = c1;
        const uint32_t nq = c1 - c0;
        const uint64_t r0 = Q->offsets[c0], r1 = Q->offsets[c1];
        o->off.resize((size_t) nq + 1);
        for (uint32_t i = 0; i <= nq; i++) o->off[i] = Q->offsets[c0 + i] - r0;
        if (profile) {   // no composition bias for profile queries (QueryMatcher.cpp:93-99, ssw_init :1229-1240)
            o->cpu = threadCpuSec() - cpu0;
            return o;
        }
        const double t0 = nowSec();
        o->sw.assign(r1 - r0 + 1, 0);
        o->dg.assign(r1 - r0 + 1, 0);
        o->km.assign(r1 - r0 + 1, 0);
        if (s->par.compBiasCorr) {
            if (s->ctxBias)
                o->rc = sd_comp_bias_batch(s->ctxBias, s->host, Q->residues + r0, o->off.data(), nq, s->k, o->sw.data(), o->dg.data(), o->km.data());
            else
                o->rc = sd_host_comp_bias(s->host, Q->residues + r0, o->off.data(), nq, s->k, o->sw.data(), o->dg.data(), o->km.data());
        }
        o->seconds = nowSec() - t0;
        o->cpu = threadCpuSec() - cpu0;
        return o;
    };
    typedef std::future<std::unique_ptr<BiasOut> > BiasFut;
    auto pfJob = [s, Q, profile, sameDb](std::shared_ptr<BiasFut> bf, sd_ctx *pfCtx) {
        std::unique_ptr<PfOut> o(new PfOut());
        o->bias = bf->get();
        const double cpu0 = threadCpuSec();
        BiasOut &b = *o->bias;
        if (b.rc != SD_OK) {
            o->rc = b.rc;
            o->err = "composition bias";
            return o;
        }
        const uint32_t nq = b.c1 - b.c0;
        const uint64_t r0 = Q->offsets[b.c0];
        std::vector<uint32_t> ident(nq);
        for (uint32_t i = 0; i < nq; i++) ident[i] = sameDb ? b.c0 + i : UINT32_MAX;
        const uint32_t W = (uint32_t) s->pfPar.maxHitsPerQuery;
        o->hits.resize((size_t) nq * W);
        o->counts.assign(nq, 0);
        o->stats.assign((size_t) nq * 4, 0);
        double t0 = nowSec();
        if (profile)
            o->rc = sd_prefilter_profile_batch(pfCtx, s->target, &s->pfPar, nq, Q->residues + r0, b.off.data(), Q->sortedScore + r0 * 20,
                                               Q->sortedIndex + r0 * 20, Q->alnProfile + r0 * 21, ident.data(), o->hits.data(),
                                               o->counts.data(), o->stats.data());
        else
            o->rc = sd_prefilter_batch(pfCtx, s->target, &s->pfPar, nq, Q->residues + r0, b.off.data(), b.km.data(), b.dg.data(),
                                       ident.data(), o->hits.data(), o->counts.data(), o->stats.data());
        o->tPrefilter = nowSec() - t0;
        if (o->rc != SD_OK) {
            o->err = std::string("sd_prefilter_batch: ") + sd_last_error(pfCtx);
            return o;
        }
        for (uint32_t i = 0; i < nq; i++)
            if (o->counts[i] == UINT32_MAX) {   // per-query error slot of sd_prefilter_batch: counted, reported, never silent
                o->counts[i] = 0;
                if (!o->notComputed) o->err = sd_last_error(pfCtx);
                o->notComputed++;
            }
        // pair list in prefilter order (Alignment.cpp:346-379); Alignment::run's coverage pre-check (:370-373) is the test
        // the prefilter already applied
        t0 = nowSec();
        o->nPairs = sd_host_pair_list(o->hits.data(), o->counts.data(), nq, W, nullptr, nullptr);
        o->pairQ.resize(std::max<uint64_t>(o->nPairs, 1));
        o->pairT.resize(std::max<uint64_t>(o->nPairs, 1));
        if (o->nPairs) sd_host_pair_list(o->hits.data(), o->counts.data(), nq, W, o->pairQ.data(), o->pairT.data());
        o->pairDiag.resize(std::max<uint64_t>(o->nPairs, 1));
        {   // same order as the pair list: query-major, a query's hits in prefilter order
            uint64_t w = 0;
            for (uint32_t q = 0; q < nq; q++) {
                const sd_hit *row = o->hits.data() + (size_t) q * W;
                // 0x8000 = no hint: only a pair whose ungapped prefilter score is already near the byte range can have a
                // diagonal that saturates the byte kernel, the others need not be walked
                for (uint32_t x = 0; x < o->counts[q]; x++) o->pairDiag[w++] = row[x].score >= 200 ? row[x].diagonal : (uint16_t) 0x8000;
            }
        }
        o->tPairs = nowSec() - t0;
        o->cpu = threadCpuSec() - cpu0;
        return o;
    };
    typedef std::future<std::unique_ptr<PfOut> > PfFut;
    std::vector<std::shared_ptr<BiasFut> > biasFut(chunks.size());
    auto submitBias = [&](size_t x) {
        if (x < chunks.size() && !biasFut[x]) {
            const uint32_t c0 = chunks[x].c0, c1 = chunks[x].c1;
            biasFut[x].reset(new BiasFut(biasStage.submit([biasJob, c0, c1] { return biasJob(c0, c1); })));
        }
    };
    // prefilter lanes: chunk x runs on lane x % lanes (its own context, workspace and stream; the target index is shared)
    const int pfLanes = s->pfLanes;
    std::unique_ptr<StageThread> pfLaneMore[3];
    for (int l = 1; l < pfLanes; l++) pfLaneMore[l - 1].reset(new StageThread(("sd-pf" + std::to_string(l)).c_str()));
    sd_ctx *pfCtxOf[4] = {s->ctxPf, s->ctxPf2, s->ctxPfMore[0], s->ctxPfMore[1]};
    auto submitPf = [&](size_t x) {
        submitBias(x);
        std::shared_ptr<BiasFut> bf = biasFut[x];
        sd_ctx *pfCtx = pfCtxOf[x % (size_t) pfLanes];
        StageThread &st = (x % (size_t) pfLanes) ? *pfLaneMore[x % (size_t) pfLanes - 1] : pfStage;
        PfFut f = st.submit([pfJob, bf, pfCtx] { return pfJob(bf, pfCtx); });
        for (int a = 1; a <= pfLanes; a++) submitBias(x + (size_t) a);
        return f;
    };

    int status = SD_OK;
    StageThread finStage{"sd-fin"};   // finalises ranges (see `finalize` below)
    std::mutex finMu;                 // the stage clocks both this thread and the finalising thread add to
    std::future<std::pair<int, double> > pending;
    std::unique_ptr<double> aggCpu(new double(0.0));   // thread CPU seconds of the aggregation jobs
    double *aggCpuP = aggCpu.get();
    const double mainCpu0 = threadCpuSec();
    bool havePending = false;
    size_t pendingChunk = 0;   // chunk whose aggregation job `pending` is
    auto waitPending = [&]() {
        if (!havePending) return;
        const double t0 = nowSec();
        const std::pair<int, double> r = pending.get();
        {
            std::lock_guard<std::mutex> l(finMu);
            tm[T_AGG_WAIT] += nowSec() - t0;
        }
        tm[T_AGG_BUSY] += r.second;
        havePending = false;
        if (r.first != SD_OK && status == SD_OK) status = s->fail(r.first, "sd_agg_add");
    };

    // A range is finalised -- sd_agg_finish, clusterhits on the device, the cluster records -- on a stage thread of its own: on the driving
    // thread (where it ran until round 6) the 0.1 - 0.2 s of record building per range were time in which no prefilter result was collected
    // and no alignment handed over (measured with the records built at N = 1 as well: 1 445 -> 1 526 ms per 1 000-proteome step).  A job
    // touches its own range's result, the clusterhits context and, under finMu, the stage clocks; errors come back as text.
    auto finalize = [&](uint32_t r, std::string *errOut) -> int {
        auto failF = [&](int rc_, const std::string &what, sd_ctx *ctx_ = nullptr) {
            *errOut = what + " failed (" + std::to_string(rc_) + ")";
            if (ctx_) *errOut += std::string(": ") + sd_last_error(ctx_);
            return rc_;
        };
        double tAggWait = 0, tCh = 0;
        struct Clocks {
            std::mutex &m;
            double *tm;
            double &a, &c;
            ~Clocks() {
                std::lock_guard<std::mutex> l(m);
                tm[T_AGG_WAIT] += a;
                tm[T_CLUSTERHITS] += c;
            }
        } clocks{finMu, tm, tAggWait, tCh};
        sd_search_result &R = *res[r];
        if (!aggregate) return SD_OK;
        double t0 = nowSec();
        uint64_t ne = 0, nh = 0;
        int rc = sd_agg_finish(R.agg, &ne, &nh);
        if (rc != SD_OK) return failF(rc, "sd_agg_finish");
        R.entryOff.assign(ne + 1, 0);
        R.entryQ.assign(std::max<uint64_t>(ne, 1), 0);
        R.entryT.assign(std::max<uint64_t>(ne, 1), 0);
        R.hitQ.assign(std::max<uint64_t>(nh, 1), 0);
        R.hitT.assign(std::max<uint64_t>(nh, 1), 0);
        R.pval.assign(std::max<uint64_t>(nh, 1), 0.0);
        rc = sd_agg_get(R.agg, R.entryOff.data(), R.entryQ.data(), R.entryT.data(), R.hitQ.data(), R.hitT.data(), R.pval.data());
        if (rc != SD_OK) return failF(rc, "sd_agg_get");
        R.entryQ.resize(ne);
        R.entryT.resize(ne);
        R.hitQ.resize(nh);
        R.hitT.resize(nh);
        R.pval.resize(nh);
        tAggWait += nowSec() - t0;
        t0 = nowSec();
        R.clusterOf.assign(nh, UINT32_MAX);
        R.rank.assign(nh, 0);
        R.nClusters.assign(ne, 0);
        R.cSize.assign(nh, 0);
        R.pCO.assign(nh, 0.0);
        R.pMH.assign(nh, 0.0);
        uint64_t nClu = 0, nCluHits = 0;
        if (nh > 0) {
            std::vector<uint32_t> qp(nh), tp(nh), nq(ne);
            std::vector<uint8_t> sd(nh);
            for (uint64_t h = 0; h < nh; h++) {
                qp[h] = Q->posInSet[R.hitQ[h]];
                tp[h] = T.posInSet[R.hitT[h]];
                sd[h] = (uint8_t) (Q->strand[R.hitQ[h]] | (T.strand[R.hitT[h]] << 1));
            }
            for (uint64_t e = 0; e < ne; e++) nq[e] = qSetSize[R.entryQ[e]];
            rc = sd_clusterhits_batch(s->ctxCh, &s->chPar, (uint32_t) ne, R.entryOff.data(), qp.data(), tp.data(), sd.data(), R.pval.data(),
                                      nq.data(), s->lgamma.data(), (uint32_t) s->lgamma.size(), R.clusterOf.data(), R.rank.data(),
                                      R.nClusters.data(), R.pCO.data(), R.pMH.data(), R.cSize.data());
            if (rc != SD_OK) return failF(rc, "sd_clusterhits_batch", s->ctxCh);
            for (uint64_t e = 0; e < ne; e++) nClu += R.nClusters[e];
            for (uint64_t h = 0; h < nh; h++) nCluHits += R.clusterOf[h] != UINT32_MAX;
        }
        tCh += nowSec() - t0;
        uint64_t na = 0, nacc = 0;
        sd_agg_stats(R.agg, &na, &nacc);
        if (s->wantRecords) {
            static const uint32_t zero32 = 0;
            static const double zeroD = 0.0;
            const bool empty = R.hitQ.empty();
            uint64_t need = 0;
            for (int pass = 0; pass < 2; pass++) {   // size, then the records
                rc = sd_agg_records(R.agg, empty ? &zero32 : R.clusterOf.data(), empty ? &zero32 : R.rank.data(),
                                    R.nClusters.empty() ? &zero32 : R.nClusters.data(), empty ? &zeroD : R.pCO.data(), empty ? &zeroD : R.pMH.data(),
                                    empty ? &zero32 : R.cSize.data(), pass ? R.records.data() : nullptr, need, &need);
                if (rc != SD_OK) return failF(rc, "sd_agg_records");
                if (!pass) R.records.resize(need);
            }
            R.haveRecords = true;
            if (s->recordsSink) s->recordsSink(s->recordsSinkUser, r, R.records.data(), (uint64_t) R.records.size());
        }
        R.counts[0] = ne;
        R.counts[1] = nh;
        R.counts[2] = nClu;
        R.counts[3] = nCluHits;
        R.counts[4] = na;
        R.counts[5] = nacc;
        R.counts[7] = pushdown ? 1 : 0;   // which gate counts[5] refers to: 1 = combinehits' E-value bound (predicate pushdown), 0 = the user's -e
        return SD_OK;
    };

    struct FinJob {
        std::future<int> fut;
        std::shared_ptr<std::string> err;
    };
    std::vector<FinJob> finJobs;
    static const bool finInline = getenv("SD_FIN_INLINE") && atoi(getenv("SD_FIN_INLINE")) != 0;   // A/B: finalise on the driving thread
    auto submitFinalize = [&](uint32_t r) {
        FinJob j;
        j.err.reset(new std::string());
        std::shared_ptr<std::string> e = j.err;
        if (finInline) {
            std::promise<int> pr;
            j.fut = pr.get_future();
            pr.set_value(finalize(r, e.get()));
        } else {
            j.fut = finStage.submit([&finalize, r, e] { return finalize(r, e.get()); });
        }
        finJobs.push_back(std::move(j));
    };
    std::vector<std::pair<uint32_t, size_t> > toFinalize;   // (range, chunk whose aggregation must have finished)
    std::vector<char> finalized(nRanges, 0);
    std::vector<uint64_t> prefHitsOfRange(nRanges, 0), pairsOfRange(nRanges, 0);

    // ---- alignment lanes: chunk ci is aligned by lane ci % lanes on that lane's context / stream, so that the host-side
    // gaps of one call (sorts, scans, downloads) are filled by the other call's kernels; results are retired in chunk order
    struct AlOut {
        int rc = SD_OK;
        std::string err;
        std::unique_ptr<PfOut> d;
        size_t ci = 0;
        uint32_t n = 0, nOut = 0;
        sd_search::AlnBuf *B = nullptr;
        uint64_t f = 0, rv = 0, tb = 0;
        double tSeqset = 0, tAlign = 0, cpu = 0;
    };
    const int lanes = s->alignLanes;
    std::unique_ptr<StageThread> alStage[4];
    for (int l = 0; l < lanes; l++) alStage[l].reset(new StageThread(("sd-al" + std::to_string(l)).c_str()));
    sd_ctx *laneCtx[4] = {s->ctxAl, s->ctxAl2, s->ctxAlMore[0], s->ctxAlMore[1]};
    const std::vector<int32_t> *qLenP = &qLen;
    auto alignJob = [s, Q, profile, sameDb, qLenP](std::shared_ptr<std::unique_ptr<PfOut> > dp, sd_ctx *ctx, sd_search::AlnBuf *Bp, size_t ci) {
        std::unique_ptr<AlOut> o(new AlOut());
        const double cpu0 = threadCpuSec();
        o->d = std::move(*dp);
        o->ci = ci;
        o->B = Bp;
        PfOut *d = o->d.get();
        if (d->nPairs == 0) return o;
        const uint32_t c0 = d->bias->c0, nq = d->bias->c1 - c0;
        const uint64_t r0 = Q->offsets[c0];
        double t0 = nowSec();
        sd_seqset *qset = nullptr;
        int rc;
        if (profile)
            rc = sd_profileset_create(ctx, Q->residues + r0, d->bias->off.data(), nq, Q->alnProfile + r0 * 21, &qset);
        else
            rc = sd_seqset_create(ctx, Q->residues + r0, d->bias->off.data(), nq, d->bias->sw.data(), &qset);
        if (rc != SD_OK) {
            o->rc = rc;
            o->err = std::string("sd_seqset_create: ") + sd_last_error(ctx);
            return o;
        }
        o->tSeqset = nowSec() - t0;
        t0 = nowSec();
        const uint32_t n = (uint32_t) d->nPairs;
        o->n = n;
        std::vector<uint8_t> identAll(n, 0);
        if (sameDb)
            for (uint32_t i = 0; i < n; i++) identAll[i] = (d->pairQ[i] + c0 == d->pairT[i]) ? 1 : 0;
        sd_search::AlnBuf &B = *Bp;
        if (B.res.size() < n) {
            B.res.resize((size_t) (1.25 * n) + 16);
            B.idx.resize(B.res.size());
        }
        uint64_t cap = std::max<uint64_t>(std::max<uint64_t>(1u << 20, 96ull * n), B.pool.size());
        bool exact = false;
        uint32_t nOut = 0;
        for (;;) {
            if (B.pool.size() < cap) B.pool.resize(cap);
            uint64_t used = 0;
            // only the reportable pairs come back (identity pairs + pairs past every gate, ~10 %): everything else
            // fails Alignment::checkCriteria and would be skipped by the aggregation anyway
            if (s->bestOnDevice)
                rc = sd_sw_align_batch_best_by_group(ctx, &s->swParRun, qset, s->tSeqs, n, d->pairQ.data(), d->pairT.data(), d->pairDiag.data(),
                                                     identAll.data(), 0.0f /* sd_agg: no --min-seq-id on this path */, s->par.alnLenThr, B.idx.data(), B.res.data(), &nOut,
                                                     B.pool.data(), B.pool.size(), &used);
            else
            rc = sd_sw_align_batch_compact_diag(ctx, &s->swParRun, qset, s->tSeqs, n, d->pairQ.data(), d->pairT.data(), d->pairDiag.data(),
                                                identAll.data(), B.idx.data(), B.res.data(), &nOut, B.pool.data(), B.pool.size(), &used);
            if (rc == SD_ENOMEM && !exact) {   // the backtrace pool has to grow: repeat with the exact bound
                uint64_t need = 64;
                for (uint32_t i = 0; i < n; i++) need += (uint64_t) (*qLenP)[c0 + d->pairQ[i]] + (uint64_t) s->tLen[d->pairT[i]];
                cap = s->cigarOnDevice ? 2 * need : need;   // (run-length text: at most two characters per letter)
                exact = true;
                continue;
            }
            break;
        }
        sd_seqset_destroy(qset);
        if (rc != SD_OK) {
            o->rc = rc;
            o->err = std::string("sd_sw_align_batch_compact: ") + sd_last_error(ctx);
            return o;
        }
        o->nOut = nOut;
        B.pq.resize(std::max<uint32_t>(nOut, 1));
        B.pt.resize(std::max<uint32_t>(nOut, 1));
        B.ident.resize(std::max<uint32_t>(nOut, 1));
        for (uint32_t x = 0; x < nOut; x++) {
            B.pq[x] = d->pairQ[B.idx[x]];
            B.pt[x] = d->pairT[B.idx[x]];
            B.ident[x] = identAll[B.idx[x]];
        }
        o->tAlign = nowSec() - t0;
        sd_sw_last_cells(ctx, &o->f, &o->rv, &o->tb);
        o->cpu = threadCpuSec() - cpu0;
        return o;
    };
    typedef std::future<std::unique_ptr<AlOut> > AlFut;
    std::deque<AlFut> inflight;

    // a finished chunk, in chunk order: counters, then its records go to the aggregation stage
    auto retire = [&](std::unique_ptr<AlOut> a) {
        if (a->rc != SD_OK) {
            if (status == SD_OK) status = s->fail(a->rc, a->err);
            return;
        }
        if (status != SD_OK) return;
        const size_t ci = a->ci;
        const uint32_t r = chunks[ci].range;
        const uint32_t c0 = a->d->bias->c0, nq = a->d->bias->c1 - c0;
        tm[T_CPU_ALIGN] += a->cpu;
        if (a->n > 0) {
            tm[T_SEQSET] += a->tSeqset;
            tm[T_ALIGN] += a->tAlign;
            s->stats[S_CELLS_FWD] += a->f;
            s->stats[S_CELLS_REV] += a->rv;
            s->stats[S_CELLS_TB] += a->tb;
            s->stats[S_PAIRS] += a->n;
            pairsOfRange[r] += a->n;
            waitPending();
            if (status != SD_OK) return;
            sd_agg *agg = res[r]->agg;
            sd_search::AlnBuf *bp = a->B;
            sd_search *sp = s;
            const uint32_t nOut = a->nOut;
            pendingChunk = ci;
            pending = aggStage.submit([agg, bp, nOut, c0, nq, sp, aggCpuP]() {
                const double t1 = nowSec();
                const double cpu1 = threadCpuSec();
                struct AddCpu {   // the aggregation jobs run one at a time: a plain accumulator
                    double *acc, c0;
                    ~AddCpu() { *acc += threadCpuSec() - c0; }
                } addCpu{aggCpuP, cpu1};
                if (sp->alnSink)
                    sp->alnSink(sp->sinkUser, c0, nq, nOut, bp->pq.data(), bp->pt.data(), bp->res.data(), bp->ident.data(), bp->pool.data());
                const int rc2 = agg ? sd_agg_add(agg, nOut, c0, bp->pq.data(), bp->pt.data(), bp->res.data(), bp->ident.data(), bp->pool.data())
                                    : SD_OK;
                return std::make_pair(rc2, nowSec() - t1);
            });
            havePending = true;
        } else if (s->alnSink) {
            waitPending();
            s->alnSink(s->sinkUser, c0, nq, 0, nullptr, nullptr, nullptr, nullptr, nullptr);
        }
        if (lastChunkOf[r] == (int64_t) ci) toFinalize.push_back(std::make_pair(r, ci));
        // ranges whose last aggregation job has finished meanwhile
        while (!toFinalize.empty() && status == SD_OK) {
            // the aggregation job in flight may belong to the range at the front (its last chunk with pairs, when later chunks of
            // the range had none): the range is finalised only once that job has been collected -- never beside it
            // (jobs run one at a time, each submitted after the previous was collected: a job of a later chunk in flight means
            // every job of the front range is done)
            const bool mine = havePending && pendingChunk <= toFinalize.front().second;
            if (mine) {
                if (pending.wait_for(std::chrono::seconds(0)) != std::future_status::ready) break;   // come back for it later
                waitPending();
                if (status != SD_OK) break;
            }
            const uint32_t fr = toFinalize.front().first;
            toFinalize.erase(toFinalize.begin());
            submitFinalize(fr);   // (every aggregation job of the range has been collected: the range's sd_agg is the job's alone now)
            finalized[fr] = 1;
        }
    };

    // prefilter jobs submitted and not yet collected, in chunk order.  A lane's thread runs its jobs one after the other; with ONE job
    // per lane a lane that finishes chunk x + 1 before chunk x is collected (chunks are collected in order, and this thread is also the
    // one that retires alignments and runs clusterhits) stands idle until then -- and the prefilter is the stage a step waits for.  With
    // SD_PF_DEPTH jobs per lane it goes on with its next chunk; a finished chunk's rows wait in host memory.  Measured (round 5, 1 000
    // proteomes, three and four lanes, depth 1 / 2 / 3): this thread's wait for the prefilter 10.4 -> 7 s per 12 steps, the throughput the
    // same (2 575 / 2 560 / 2 550 and 2 628 / 2 620 / 2 593) -- the device, not the lanes' idle time, bounds a step: one job per lane stays.
    std::deque<PfFut> pfQueue;
    size_t pfSubmitted = 0;
    static const int pfDepth = getenv("SD_PF_DEPTH") ? std::max(1, std::min(4, atoi(getenv("SD_PF_DEPTH")))) : 1;
    auto pumpPf = [&]() {
        while (pfSubmitted < chunks.size() && (int) pfQueue.size() < pfLanes * pfDepth) pfQueue.push_back(submitPf(pfSubmitted++));
    };
    pumpPf();
    for (size_t ci = 0; ci < chunks.size() && status == SD_OK; ci++) {
        const uint32_t r = chunks[ci].range;
        double t0 = nowSec();
        std::unique_ptr<PfOut> d = pfQueue.front().get();
        pfQueue.pop_front();
        tm[T_PF_WAIT] += nowSec() - t0;
        pumpPf();
        if (d->rc != SD_OK) {
            status = s->fail(d->rc, d->err);
            break;
        }
        tm[T_BIAS] += d->bias->seconds;
        tm[T_CPU_BIAS] += d->bias->cpu;
        tm[T_CPU_PF] += d->cpu;
        tm[T_PREFILTER] += d->tPrefilter;
        tm[T_PAIRS] += d->tPairs;
        const uint32_t c0 = d->bias->c0, c1 = d->bias->c1, nq = c1 - c0;
        for (uint32_t i = 0; i < nq; i++) {
            s->stats[S_KMERS] += d->stats[(size_t) i * 4];
            s->stats[S_INDEX_HITS] += d->stats[(size_t) i * 4 + 1];
            s->stats[S_DIAGONALS] += d->stats[(size_t) i * 4 + 2];
            s->stats[S_DIAG_LEN] += d->stats[(size_t) i * 4 + 3];
        }
        s->stats[S_PREF_HITS] += d->nPairs;
        if (d->notComputed) {
            s->stats[S_NOT_COMPUTED] += d->notComputed;
            s->err = std::to_string(s->stats[S_NOT_COMPUTED]) + " queries were not computed: " + d->err;
        }
        prefHitsOfRange[r] += d->nPairs;
        // the prefilter sink sees the chunks in order (the lanes finish in any order)
        if (s->prefSink) s->prefSink(s->sinkUser, c0, nq, d->hits.data(), d->counts.data(), (uint32_t) s->pfPar.maxHitsPerQuery);
        // at most one chunk per lane in flight; the oldest is retired (in chunk order) before its lane takes the next
        while ((int) inflight.size() >= lanes && status == SD_OK) {
            std::unique_ptr<AlOut> a = inflight.front().get();
            inflight.pop_front();
            retire(std::move(a));
        }
        if (status != SD_OK) break;
        // the buffer of chunk ci - 8 is free (at most four chunks in the lanes, one in the aggregation)
        s->flip = (s->flip + 1) & 7;
        sd_search::AlnBuf *Bp = &s->buf[s->flip];
        std::shared_ptr<std::unique_ptr<PfOut> > dp(new std::unique_ptr<PfOut>(std::move(d)));
        sd_ctx *ctx = laneCtx[ci % (size_t) lanes];
        inflight.push_back(alStage[ci % (size_t) lanes]->submit([alignJob, dp, ctx, Bp, ci] { return alignJob(dp, ctx, Bp, ci); }));
    }
    while (!inflight.empty()) {   // also on errors: the lanes still hold jobs that reference this frame
        std::unique_ptr<AlOut> a = inflight.front().get();
        inflight.pop_front();
        retire(std::move(a));
    }
    // drain: the stage threads may still hold jobs that reference this frame
    for (size_t x = 0; x < pfQueue.size(); x++)
        if (pfQueue[x].valid()) pfQueue[x].wait();
    for (size_t x = 0; x < biasFut.size(); x++)
        if (biasFut[x] && biasFut[x]->valid()) biasFut[x]->wait();
    waitPending();
    if (status == SD_OK)
        for (uint32_t r = 0; r < nRanges; r++)
            if (!finalized[r]) submitFinalize(r);
    for (FinJob &j : finJobs) {   // (also on errors: the jobs reference this frame)
        const int rcF = j.fut.get();
        if (rcF != SD_OK && status == SD_OK) {
            s->err = *j.err;
            status = rcF;
        }
    }
    {
        std::lock_guard<std::mutex> l(finMu);
        tm[T_TOTAL] = nowSec() - tAll;
    }
    tm[T_CPU_AGG_MAIN] += *aggCpu + (threadCpuSec() - mainCpu0);
    if (status != SD_OK) return status;
    for (uint32_t r = 0; r < nRanges; r++) {
        res[r]->counts[6] = prefHitsOfRange[r];
        res[r]->counts[4] = pairsOfRange[r];
        results[r] = res[r].release();
    }
    return SD_OK;
}

int sd_search_result_counts(sd_search_result *r, uint64_t *counts) {
    if (!r || !counts) return SD_EINVAL;
    memcpy(counts, r->counts, sizeof(r->counts));
    return SD_OK;
}

int sd_search_result_arrays(sd_search_result *r, uint64_t *entryOff, uint32_t *entryQSet, uint32_t *entryTSet, uint32_t *hitQ,
                            uint32_t *hitT, double *pval, uint32_t *clusterOfHit, uint32_t *rankInCluster, uint32_t *nClusters,
                            double *pCO, double *pMH, uint32_t *clusterSize) {
    if (!r) return SD_EINVAL;
#define SD_COPY(dst, v) \
    if (dst && !(v).empty()) memcpy(dst, (v).data(), (v).size() * sizeof((v)[0]))
    SD_COPY(entryOff, r->entryOff);
    SD_COPY(entryQSet, r->entryQ);
    SD_COPY(entryTSet, r->entryT);
    SD_COPY(hitQ, r->hitQ);
    SD_COPY(hitT, r->hitT);
    SD_COPY(pval, r->pval);
    SD_COPY(clusterOfHit, r->clusterOf);
    SD_COPY(rankInCluster, r->rank);
    SD_COPY(nClusters, r->nClusters);
    SD_COPY(pCO, r->pCO);
    SD_COPY(pMH, r->pMH);
    SD_COPY(clusterSize, r->cSize);
#undef SD_COPY
    return SD_OK;
}

int sd_search_result_records(sd_search_result *r, void *out, uint64_t cap, uint64_t *bytes) {
    if (!r || !r->agg || !bytes) return SD_EINVAL;
    if (r->haveRecords) {   // built inside the stream
        *bytes = r->records.size();
        if (!out) return SD_OK;
        if (cap < r->records.size()) return SD_ENOMEM;
        if (!r->records.empty()) memcpy(out, r->records.data(), r->records.size());
        return SD_OK;
    }
    static const uint32_t zero32 = 0;
    static const double zeroD = 0.0;
    const bool empty = r->hitQ.empty();
    return sd_agg_records(r->agg, empty ? &zero32 : r->clusterOf.data(), empty ? &zero32 : r->rank.data(),
                          r->nClusters.empty() ? &zero32 : r->nClusters.data(), empty ? &zeroD : r->pCO.data(),
                          empty ? &zeroD : r->pMH.data(), empty ? &zero32 : r->cSize.data(), out, cap, bytes);
}

int sd_search_result_write_tsv(sd_search_result *r, const char *path, const char *qNames, const uint64_t *qNameOff,
                               const char *tNames, const uint64_t *tNameOff, const char *qSources, const uint64_t *qSourceOff,
                               const char *tSources, const uint64_t *tSourceOff, int canonical, int append,
                               uint64_t firstClusterKey, uint64_t *nClusterLines, uint64_t *nHitLines) {
    if (!r || !path || !r->agg) return SD_EINVAL;
    static const uint32_t zero32 = 0;
    static const double zeroD = 0.0;
    const bool empty = r->hitQ.empty();
    return sd_agg_write_tsv_from(r->agg, path, append, firstClusterKey, empty ? &zero32 : r->clusterOf.data(),
                                 empty ? &zero32 : r->rank.data(), r->nClusters.empty() ? &zero32 : r->nClusters.data(),
                                 empty ? &zeroD : r->pCO.data(), empty ? &zeroD : r->pMH.data(), empty ? &zero32 : r->cSize.data(), qNames,
                                 qNameOff, tNames, tNameOff, qSources, qSourceOff, tSources, tSourceOff, canonical, nClusterLines,
                                 nHitLines);
}

void sd_search_result_destroy(sd_search_result *r) { delete r; }

}  // extern "C"
