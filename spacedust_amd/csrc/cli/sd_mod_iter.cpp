// `clustersearch --num-iterations N` without DB files between the modules of an iteration (BASELINE configs[3]).
//
// The reference runs an iterative profile search as a chain of processes over on-disk DBs (M/data/workflow/blastpgp.sh:62-133:
// prefilter -> [subtractdbs] -> align -> [mergedbs] -> result2profile per iteration, Search.cpp:476-518 builds their parameters),
// then R/data/clustersearch.sh:121-151 on the merged alignment DB.  Every step of that chain is per QUERY: a query's rows,
// alignments and profile of iteration k depend on its own results of iteration k - 1 and on nothing else.  So here a chunk of queries
// (one query proteome by default) is taken through ALL iterations by one worker, in memory --
//     prefilter rows (sd_prefilter_batch / sd_prefilter_profile_batch)  ->  minus the targets aligned already (subtractdbs)
//     ->  alignChunkCore (the body of `align`: sd_sw_align_batch*, criteria, compareHits order, --realign in iteration 0)
//     ->  the accepted records appended to the query's list (mergedbs)  ->  sd_r2p_batch_device  ->  sd_host_map_profiles
// -- and several workers run side by side, each on its own context / stream: the host stages of one chunk (MSA assembly and the
// diversity filter of result2profile, criteria and sorts) overlap the kernels of the others, and the serial chains of the weight
// kernel (csrc/hip/sd_r2p.hip) share the device with prefilter and alignment kernels instead of owning it.  A finished chunk's
// records go to ONE aggregation object in chunk order (sd_agg_add: prefixid -> besthitbyset -> mergeresultsbyset -> combinehits),
// then sd_clusterhits_batch and the TSV from the cluster records, as in the single-pass clustersearch.
//
// What the text hand-offs of the module chain do to a value is kept: an E-value is compared after its "%.3E" round trip
// (sd_host_quantise_3e) where result2profile (result2profile.cpp:197-204) and subtractdbs (subtractdbs.cpp:60-90) parse it from
// an alignment line.  `--keep-tmp 1` (the per-iteration DBs are wanted) and SD_ITER_FILES=1 run the module chain instead;
// tools/iter3_scale.py checks that chain's DBs against the reference classes and this path's TSV against that chain's.
#include "sd_align_core.h"
#include "sd_cli.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <omp.h>
#include <string>
#include <sys/resource.h>
#include <thread>
#include <vector>

namespace sdcli {

namespace {

double nowS() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// CPU seconds of the whole process so far (all threads): per stage this is only meaningful with one worker (SD_ITER_WORKERS=1)
double cpuS() {
    rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    return (double) ru.ru_utime.tv_sec + 1e-6 * (double) ru.ru_utime.tv_usec + (double) ru.ru_stime.tv_sec + 1e-6 * (double) ru.ru_stime.tv_usec;
}

struct Rec {   // an accepted alignment of a query; r.btOffset points into the chunk's pool
    uint32_t tId;
    uint8_t ident;
    sd_sw_result r;
};

enum { IT_PREF, IT_ALIGN, IT_R2P, IT_N };

// what a finished chunk hands to the aggregation: the final records of its queries, query-major (pairQ = index inside the chunk)
struct ChunkOut {
    uint32_t g0 = 0, nq = 0;
    std::vector<uint32_t> pq, pt;
    std::vector<sd_sw_result> res;
    std::vector<uint8_t> ident;
    std::string pool;
    uint64_t notComputed = 0, prefHits = 0, aligned = 0, accepted = 0;
    uint64_t pfStats[5] = {0, 0, 0, 0, 0};   // similar k-mers, index hits, diagonals, diagonal length (sd_prefilter_batch's stats), query residues
    int rc = SD_OK;
    std::string err;
    bool done = false;
    double sec[8][IT_N];   // seconds per iteration and stage
    double cpu[8][IT_N];   // process CPU seconds in the same intervals
    uint64_t cnt[8][4];    // per iteration: prefilter rows, rows left after the subtraction, pairs aligned, pairs accepted
};

struct Shared {
    const SeqDb *qdb = nullptr, *tdb = nullptr;
    bool sameDb = false;
    int numIt = 1, device = 0;
    sd_target *seqTarget = nullptr, *profTarget = nullptr;
    sd_seqset *tset = nullptr;
    PrefSetup pfSeq, pfProf;
    AlignSetup alFirst, alMid, alLast;   // iteration 0 (--realign 1, the profile E-value) | the middle ones | the last (the user's -e)
    const R2pSetup *r2p = nullptr;
    double subtractEval = 0.001;         // subtractdbs' bound (min of -e and --e-profile)
    bool profile = false;                // SD_ITER_PROFILE: per-kernel event times of every worker's context
};

struct Worker {
    sd_ctx *ctx = nullptr;
    sd_host *host = nullptr;
    sd_r2p *r2p = nullptr;
    int threads = 1;
    AlignChunk C;
    std::vector<uint32_t> mark;   // target id -> query stamp (subtractdbs)
    uint32_t stamp = 0;
    ~Worker() {
        if (r2p) sd_r2p_destroy(r2p);
        if (ctx) sd_ctx_destroy(ctx);
        if (host) sd_host_destroy(host);
    }
};

// the value an alignment line's E-value column parses to
double quantE(double v) {
    char text[32];
    double back = v;
    sd_host_quantise_3e(v, text, &back);
    return back;
}

int failChunk(ChunkOut &o, int rc, const std::string &what, sd_ctx *ctx) {
    o.rc = rc;
    o.err = what + " failed (" + std::to_string(rc) + ")";
    if (ctx) o.err += std::string(": ") + sd_last_error(ctx);
    return rc;
}

// one chunk of queries [g0, g1) through all iterations
int processChunk(const Shared &S, Worker &W, uint32_t g0, uint32_t g1, ChunkOut &out) {
    const SeqDb &qdb = *S.qdb, &tdb = *S.tdb;
    const uint32_t nq = g1 - g0;
    out.g0 = g0;
    out.nq = nq;
    memset(out.sec, 0, sizeof(out.sec));
    memset(out.cpu, 0, sizeof(out.cpu));
    memset(out.cnt, 0, sizeof(out.cnt));
    std::vector<std::vector<Rec> > acc(nq);
    std::string &pool = out.pool;
    pool.clear();
    std::unique_ptr<SeqDb> pdb;   // the chunk's profiles of the previous iteration (local ids)
    std::vector<sd_hit> hits;
    std::vector<uint32_t> counts, identId;
    std::vector<uint64_t> off((size_t) nq + 1), pfSt((size_t) nq * 4);
    std::vector<int8_t> diagBias;
    std::vector<int16_t> kmerBias;
    if (W.mark.size() != tdb.n) W.mark.assign(tdb.n, 0);
    AlignChunk &C = W.C;
    for (int step = 0; step < S.numIt; step++) {
        const bool last = step == S.numIt - 1;
        const bool prof = step > 0;
        const SeqDb &qd = prof ? *pdb : qdb;
        const uint32_t base = prof ? 0 : g0;
        const bool sameDbStep = S.sameDb && !prof;   // (a profile DB is never "the same DB" as the target: blastpgp.sh hands the modules its path)
        const PrefSetup &PS = prof ? S.pfProf : S.pfSeq;
        // ---- prefilter (the loop body of prefilterModule)
        double t0 = nowS(), c0 = cpuS();
        const uint64_t r0 = qd.offsets[base], r1 = qd.offsets[base + nq];
        for (uint32_t i = 0; i <= nq; i++) off[i] = qd.offsets[base + i] - r0;
        identId.resize(nq);
        for (uint32_t i = 0; i < nq; i++) {
            uint32_t id = UINT32_MAX;
            if (sameDbStep) id = g0 + i;
            else if (PS.includeIdentity) {
                const size_t t = tdb.rd.idOfKey(qdb.keys[g0 + i]);
                if (t != SIZE_MAX) id = (uint32_t) t;
            }
            identId[i] = id;
        }
        const uint32_t Wd = (uint32_t) PS.par.maxHitsPerQuery;
        hits.resize((size_t) nq * Wd);
        counts.assign(nq, 0);
        int rc;
        if (prof) {
            rc = sd_prefilter_profile_batch(W.ctx, S.profTarget, &PS.par, nq, qd.residues.data() + r0, off.data(), qd.sortedScore.data() + r0 * 20,
                                            qd.sortedIndex.data() + r0 * 20, qd.alnProfile.data() + r0 * 21, identId.data(), hits.data(), counts.data(),
                                            pfSt.data());
        } else {
            diagBias.assign(r1 - r0 + 1, 0);
            kmerBias.assign(r1 - r0 + 1, 0);
            if (PS.compBias) sd_host_comp_bias(W.host, qd.residues.data() + r0, off.data(), nq, PS.k, nullptr, diagBias.data(), kmerBias.data());
            rc = sd_prefilter_batch(W.ctx, S.seqTarget, &PS.par, nq, qd.residues.data() + r0, off.data(), kmerBias.data(), diagBias.data(),
                                    identId.data(), hits.data(), counts.data(), pfSt.data());
        }
        if (rc != SD_OK) return failChunk(out, rc, prof ? "sd_prefilter_profile_batch" : "sd_prefilter_batch", W.ctx);
        for (uint32_t i = 0; i < nq; i++)
            if (counts[i] == UINT32_MAX) {   // per-query error slot: not computed (reported, never silent)
                if (!out.notComputed) out.err = sd_last_error(W.ctx);
                out.notComputed++;
                counts[i] = 0;
            }
        for (uint32_t i = 0; i < nq; i++)
            for (int k = 0; k < 4; k++) out.pfStats[k] += pfSt[(size_t) i * 4 + k];
        out.pfStats[4] += r1 - r0;
        out.sec[std::min(step, 7)][IT_PREF] += nowS() - t0;
        out.cpu[std::min(step, 7)][IT_PREF] += cpuS() - c0;
        // ---- the pairs: a query's rows in prefilter order, minus the targets it has aligned already (subtractdbs.cpp:60-90:
        // a target listed with E <= the profile E-value in the alignments so far)
        t0 = nowS();
        c0 = cpuS();
        const AlignSetup &AS = step == 0 ? S.alFirst : (last ? S.alLast : S.alMid);
        C.localQ.resize(nq);
        for (uint32_t i = 0; i < nq; i++) C.localQ[i] = base + i;
        C.pq.clear();
        C.pt.clear();
        C.ident.clear();
        for (uint32_t q = 0; q < nq; q++) {
            if (prof) {
                if (++W.stamp == 0) {   // the stamps wrapped: start over
                    std::fill(W.mark.begin(), W.mark.end(), 0u);
                    W.stamp = 1;
                }
                for (const Rec &e : acc[q])
                    if (quantE(e.r.evalue) <= S.subtractEval) W.mark[e.tId] = W.stamp;
            }
            const uint32_t qKey = qdb.keys[g0 + q];
            const float qL = (float) qd.lens[base + q];
            const sd_hit *row = hits.data() + (size_t) q * Wd;
            out.cnt[std::min(step, 7)][0] += counts[q];
            for (uint32_t x = 0; x < counts[q]; x++) {
                const uint32_t tId = row[x].seqId;
                if (prof && W.mark[tId] == W.stamp) continue;
                out.prefHits++;
                out.cnt[std::min(step, 7)][1]++;
                // Util::canBeCovered pre-check (Alignment.cpp:370-373): a rejected pair, never aligned
                const bool can = sd_host_can_be_covered(AS.canCovThr, AS.covMode, qL, (float) tdb.lens[tId]) != 0;
                C.pq.push_back(q);
                C.pt.push_back(tId);
                C.ident.push_back(!can ? 2 : ((qKey == tdb.keys[tId] && (AS.includeIdentity || sameDbStep)) ? 1 : 0));
            }
        }
        // ---- align (the chunk body of alignModule) and the accepted records, appended to each query's list (mergedbs)
        const char *what = "";
        rc = alignChunkCore(W.ctx, W.host, AS, qd, tdb, S.tset, C, nullptr, &what);
        if (rc != SD_OK) return failChunk(out, rc, what, W.ctx);
        out.aligned += C.aligned;
        out.cnt[std::min(step, 7)][2] += C.aligned;
        out.cnt[std::min(step, 7)][3] += C.accepted;
        {
            uint64_t w = 0;
            for (uint32_t q = 0; q < nq; q++)
                for (uint32_t x = 0; x < (*C.outCounts)[q]; x++, w++) {
                    const uint32_t i = (*C.outOrder)[w];
                    Rec e;
                    e.tId = (*C.outT)[i];
                    e.ident = (*C.outIdent)[i];
                    e.r = (*C.outRecs)[i];
                    const uint64_t at = pool.size();
                    if (e.r.btLen > 0) pool.append(C.outPool->data() + e.r.btOffset, (size_t) e.r.btLen);
                    e.r.btOffset = at;
                    acc[q].push_back(e);
                    out.accepted++;
                }
        }
        if (const char *dq = getenv("SD_ITER_DEBUG_Q")) {   // the records of one query after this iteration (debugging aid)
            for (const char *tok = dq; tok && *tok; tok = strchr(tok, ',') ? strchr(tok, ',') + 1 : nullptr) {
            const long want = atol(tok);
            if (want >= (long) g0 && want < (long) g1) {
                const uint32_t q = (uint32_t) (want - g0);
                FILE *df = getenv("SD_ITER_DEBUG_FILE") ? fopen(getenv("SD_ITER_DEBUG_FILE"), "a") : stderr;
                if (!df) df = stderr;
                fprintf(df, "[iter debug] query %ld after iteration %d: %zu records (rows %u)\n", want, step, acc[q].size(), counts[q]);
                for (const Rec &e : acc[q])
                    fprintf(df, "[iter debug]   t %u key %u ident %d score %d E %.4g q %d-%d t %d-%d bt %d\n", e.tId, tdb.keys[e.tId], (int) e.ident, e.r.score,
                            e.r.evalue, e.r.qStart, e.r.qEnd, e.r.tStart, e.r.tEnd, e.r.btLen);
                if (df != stderr) fclose(df);
            }
            }
        }
        out.sec[std::min(step, 7)][IT_ALIGN] += nowS() - t0;
        out.cpu[std::min(step, 7)][IT_ALIGN] += cpuS() - c0;
        if (last) break;
        // ---- result2profile (result2profile.cpp:150-282): the query's alignments with E < the profile E-value, the query itself left
        // out when query and target DB are the same one
        t0 = nowS();
        c0 = cpuS();
        std::vector<uint8_t> qLetters(qd.residues.begin() + r0, qd.residues.begin() + r1);
        std::vector<uint64_t> edgeOff(1, 0), btOff(1, 0);
        std::vector<uint32_t> edgeT;
        std::vector<int32_t> eQ, eT;
        std::string btPool;
        for (uint32_t q = 0; q < nq; q++) {
            const uint32_t qKey = qdb.keys[g0 + q];
            for (const Rec &e : acc[q]) {
                if (sameDbStep && tdb.keys[e.tId] == qKey) continue;
                if (!(quantE(e.r.evalue) < S.r2p->evalProfile)) continue;
                edgeT.push_back(e.tId);
                eQ.push_back(e.r.qStart);
                eT.push_back(e.r.tStart);
                if (e.r.btLen > 0) btPool.append(pool.data() + e.r.btOffset, (size_t) e.r.btLen);
                else btPool.push_back('M');   // (Matcher::uncompressAlignment of "0M": one letter)
                btOff.push_back(btPool.size());
            }
            edgeOff.push_back(edgeT.size());
        }
        std::vector<char> profiles((r1 - r0) * 25 + 1, 0);
        btPool.push_back(' ');
        qLetters.push_back(0);
        edgeT.push_back(0);
        eQ.push_back(0);
        eT.push_back(0);
        rc = sd_r2p_batch_device(W.ctx, W.r2p, &S.r2p->par, nq, qLetters.data(), off.data(), edgeOff.data(), edgeT.data(), eQ.data(), eT.data(),
                                 btPool.data(), btOff.data(), tdb.residues.data(), tdb.offsets.data(), profiles.data(), nullptr);
        if (rc != SD_OK) return failChunk(out, rc, "sd_r2p_batch_device", W.ctx);
        // the profile DB entry of every query as the next iteration's query (Sequence::mapProfile)
        std::unique_ptr<SeqDb> nx(new SeqDb());
        nx->n = nq;
        nx->profile = true;
        nx->keys.assign(qdb.keys.begin() + g0, qdb.keys.begin() + g1);
        const uint64_t total = r1 - r0;
        std::vector<uint64_t> byteOff((size_t) nq + 1);
        for (uint32_t i = 0; i <= nq; i++) byteOff[i] = off[i] * 25;
        nx->residues.resize(total + 1);
        nx->consensus.resize(total + 1);
        nx->alnProfile.resize((total + 1) * 21);
        nx->sortedScore.resize((total + 1) * 20);
        nx->sortedIndex.resize((total + 1) * 20);
        nx->offsets.assign((size_t) nq + 1, 0);
        rc = sd_host_map_profiles(profiles.data(), byteOff.data(), nq, nx->residues.data(), nx->consensus.data(), nx->alnProfile.data(),
                                  nx->sortedScore.data(), nx->sortedIndex.data(), nx->offsets.data());
        if (rc != SD_OK) return failChunk(out, rc, "sd_host_map_profiles", nullptr);
        nx->lens.resize(nq);
        for (uint32_t i = 0; i < nq; i++) nx->lens[i] = (int32_t) (nx->offsets[i + 1] - nx->offsets[i]);
        pdb = std::move(nx);
        out.sec[std::min(step, 7)][IT_R2P] += nowS() - t0;
        out.cpu[std::min(step, 7)][IT_R2P] += cpuS() - c0;
    }
    // the chunk's final records, query-major in list order (the merged alignment DB's lines)
    size_t n = 0;
    for (uint32_t q = 0; q < nq; q++) n += acc[q].size();
    out.pq.resize(std::max<size_t>(n, 1));
    out.pt.resize(std::max<size_t>(n, 1));
    out.res.resize(std::max<size_t>(n, 1));
    out.ident.resize(std::max<size_t>(n, 1));
    size_t w = 0;
    for (uint32_t q = 0; q < nq; q++)
        for (const Rec &e : acc[q]) {
            out.pq[w] = q;
            out.pt[w] = e.tId;
            out.res[w] = e.r;
            out.ident[w] = e.ident;
            w++;
        }
    out.pq.resize(n);
    out.pt.resize(n);
    out.res.resize(n);
    out.ident.resize(n);
    return SD_OK;
}

Args subArgs(const char *name, const std::vector<std::string> &flags, std::string *err) {
    std::vector<const char *> argv;
    for (const std::string &s : flags) argv.push_back(s.c_str());
    Args sub;
    sub.module = name;
    sub.parse((int) argv.size(), argv.data(), err);
    return sub;
}

}  // namespace

// a: the clustersearch command line with the workflow's defaults filled in; pref / aln / prof: the parameter strings iterativeSearch
// builds for the modules (Search.cpp:476-518).  Writes the cluster-hit TSV.
int iterativeClusterSearchInMemory(const Args &a, const std::string &Q, const std::string &T, const std::string &tsvPath,
                                   const std::vector<std::string> &prefFlags, const std::vector<std::string> &alnFlags,
                                   const std::vector<std::string> &profFlags, const std::string &eUser, const std::string &eProfile) {
    const int numIt = (int) a.integer("--num-iterations", 1);
    const int threads = threadsOf(a);
    const int device = a.has("--device") ? (int) a.integer("--device", 0) : (getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0);
    const double tStart = nowS();
    struct HostG {
        sd_host *h = nullptr;
        ~HostG() { if (h) sd_host_destroy(h); }
    } host;
    if (sd_host_create(threads, &host.h) != SD_OK) return fail("sd_host_create failed");
    std::string err;
    const bool sameDb = Q == T;
    std::unique_ptr<SeqDb> tdb(new SeqDb()), qdbOwn;
    if (!tdb->load(T, host.h, &err)) return fail(err);
    if (tdb->profile) return fail("profile target databases are not supported on this path");
    SeqDb *qdb = tdb.get();
    if (!sameDb) {
        qdbOwn.reset(new SeqDb());
        if (!qdbOwn->load(Q, host.h, &err)) return fail(err);
        qdb = qdbOwn.get();
    }
    if (qdb->profile) return fail("--num-iterations starts from a sequence query DB");
    SetInfo qs, tsOwn;
    if (!qs.load(Q, true, &err)) return fail(err);
    if (!sameDb && !tsOwn.load(T, true, &err)) return fail(err);
    const SetInfo *tsP = sameDb ? &qs : &tsOwn;
    SetDbArrays tv, qv;
    tv.fill(*tdb, tsP);
    qv.fill(*qdb, &qs);
    info(a, "Query database size: %u type: Aminoacid\nTarget database size: %u type: Aminoacid\n", qdb->n, tdb->n);
    const double tLoaded = nowS();

    // ---- the modules' parameters, derived by the modules' own setup functions from the modules' own parameter strings
    Shared S;
    S.qdb = qdb;
    S.tdb = tdb.get();
    S.sameDb = sameDb;
    S.numIt = numIt;
    S.device = device;
    S.profile = getenv("SD_ITER_PROFILE") != nullptr;
    Args prefA = subArgs("prefilter", prefFlags, &err);
    if (int rc = prefilterSetupFromArgs(prefA, host.h, *tdb, false, S.pfSeq)) return rc;
    if (int rc = prefilterSetupFromArgs(prefA, host.h, *tdb, true, S.pfProf)) return rc;
    auto alnArgs = [&](const std::string &e, bool realign) {
        std::vector<std::string> f = alnFlags;
        f.push_back("-e");
        f.push_back(e);
        f.push_back("--realign");
        f.push_back(realign ? "1" : "0");
        return subArgs("align", f, &err);
    };
    if (int rc = alignSetupFromArgs(alnArgs(eProfile, true), host.h, tdb->totalResidues(), S.alFirst)) return rc;
    if (int rc = alignSetupFromArgs(alnArgs(eProfile, false), host.h, tdb->totalResidues(), S.alMid)) return rc;
    if (int rc = alignSetupFromArgs(alnArgs(eUser, false), host.h, tdb->totalResidues(), S.alLast)) return rc;
    R2pSetup RS;
    if (int rc = r2pSetupFromArgs(subArgs("result2profile", profFlags, &err), RS)) return rc;
    S.r2p = &RS;
    {
        const double eU = strtod(eUser.c_str(), nullptr), eP = strtod(eProfile.c_str(), nullptr);
        S.subtractEval = eU < eP ? eU : eP;   // subtractdbs.cpp:20-21
    }

    // ---- workers: a context, a host object and a result2profile object each
    int nWorkers = getenv("SD_ITER_WORKERS") ? atoi(getenv("SD_ITER_WORKERS")) : 4;   // (1 000 target proteomes, 20 query proteomes, one box: 3 workers 491 - 496, 4 workers 501 - 504, 5 workers 454 genome-pairs/s; profiles/r06l_iter3_ab.txt)
    nWorkers = std::max(1, std::min(nWorkers, 8));
    uint32_t chunkQ = (uint32_t) std::max<long long>(1, a.integer("--iter-chunk-queries", getenv("SD_ITER_CHUNK") ? atoi(getenv("SD_ITER_CHUNK")) : 1500));
    // the host stages of a chunk run on an OpenMP team of its worker; the teams overlap each other's device phases, so together they
    // may ask for more threads than there are CPUs
    const int perWorker = std::max(1, std::min(threads, (threads * 3 / 2 + nWorkers - 1) / nWorkers));
    std::vector<std::unique_ptr<Worker> > workers;
    for (int w = 0; w < nWorkers; w++) {
        workers.emplace_back(new Worker());
        Worker &W = *workers.back();
        W.threads = perWorker;
        int rc = sd_ctx_create(device, &W.ctx);
        if (rc != SD_OK)
            return fail("no usable HIP device (sd_ctx_create returned " + std::to_string(rc) + "); this path has no CPU fallback");
        if (sd_host_create(perWorker, &W.host) != SD_OK) return fail("sd_host_create failed");
        if (sd_r2p_create(&W.r2p) != SD_OK) return fail("sd_r2p_create failed");
        if (S.profile) sd_profile_enable(W.ctx, 1);
    }
    // ---- the target: both indexes (the sequence search's k-mer threshold; every k-mer for the profile searches,
    // Prefiltering.cpp:525-527) and the sequences, resident for the whole run
    uint64_t indexEntries = 0;
    struct TargetG {
        sd_target *a = nullptr, *b = nullptr;
        sd_seqset *s = nullptr;
        ~TargetG() {
            if (s) sd_seqset_destroy(s);
            if (b) sd_target_destroy(b);
            if (a) sd_target_destroy(a);
        }
    } tg;
    {
        const int16_t *s2, *s3;
        const uint16_t *i2, *i3;
        uint32_t z2, z3;
        sd_host_ext_matrix(host.h, 2, &s2, &i2, &z2);
        sd_host_ext_matrix(host.h, 3, &s3, &i3, &z3);
        double ratios[21 * 21];
        int8_t self[21];
        sd_host_index_tables(host.h, ratios, self);
        sd_ctx *c0 = workers[0]->ctx;
        uint64_t st[4] = {0, 0, 0, 0};
        int rc = sd_target_build(c0, S.pfSeq.k, S.pfSeq.indexThr, S.pfSeq.mask ? 1 : 0, S.pfSeq.maskProb, tdb->residues.data(), tdb->offsets.data(),
                                 tdb->n, ratios, self, s2, i2, s3, i3, &tg.a, st);
        if (rc != SD_OK) return failCtx(c0, rc, "sd_target_build");
        indexEntries = st[0];
        info(a, "Index table k-mer threshold: %d at k-mer size %d\nIndex statistics\nEntries:          %llu\n", S.pfSeq.kmerThr, S.pfSeq.k,
             (unsigned long long) st[0]);
        if (numIt > 1) {
            rc = sd_target_build(c0, S.pfProf.k, S.pfProf.indexThr, S.pfProf.mask ? 1 : 0, S.pfProf.maskProb, tdb->residues.data(), tdb->offsets.data(),
                                 tdb->n, ratios, self, s2, i2, s3, i3, &tg.b, st);
            if (rc != SD_OK) return failCtx(c0, rc, "sd_target_build (profile searches)");
            indexEntries = std::max<uint64_t>(indexEntries, st[0]);
            info(a, "Index table k-mer threshold: %d at k-mer size %d (profile iterations: every k-mer indexed)\nEntries:          %llu\n",
                 S.pfProf.kmerThr, S.pfProf.k, (unsigned long long) st[0]);
        }
        rc = sd_seqset_create(c0, tdb->residues.data(), tdb->offsets.data(), tdb->n, nullptr, &tg.s);
        if (rc != SD_OK) return failCtx(c0, rc, "sd_seqset_create(targets)");
    }
    S.seqTarget = tg.a;
    S.profTarget = tg.b;
    S.tset = tg.s;
    const double tResident = nowS();
    // ---- how many of the workers the device's free memory allows.  A worker's largest workspace is the prefilter's hit stream: the
    // profile iterations run the lookup path, whose sub-batches hold up to 2^30 hits at ~26 B each (a sub-batch of this chunk size may
    // hold fewer: similar k-mers per query x the index's mean list length), next to the k-mer streams and the alignment buffers.
    {
        uint64_t freeB = 0, totalB = 0;
        if (sd_device_memory(workers[0]->ctx, &freeB, &totalB) == SD_OK && totalB > 0) {
            const double meanList = (double) indexEntries / 64000000.0;   // (k = 6: 6.4 * 10^7 k-mers)
            const double hitsPerSubBatch = std::min((double) (1ull << 30), (double) chunkQ * 1.3e5 * std::max(meanList, 1.0));
            const double perWorker = 26.0 * hitsPerSubBatch * 1.3 + 6e9;
            const int fit = (int) std::max(1.0, std::floor(((double) freeB - 4e9) / perWorker));
            if (fit < nWorkers) {
                info(a, "%d of %d workers: %.0f GB of device memory free, ~%.0f GB per worker\n", fit, nWorkers, (double) freeB / 1e9, perWorker / 1e9);
                nWorkers = fit;
            }
        }
    }

    // ---- groups and chunks.  A group is a query set (sets are contiguous id ranges in a createsetdb DB; a DB that is not laid out that
    // way is one group): it has its own aggregation, finalised -- clusterhits, cluster records, its part of the TSV -- as soon as its
    // last chunk has been added, on a thread of its own, while the workers are busy with the next sets.  Chunks are equal parts of a
    // group of at most chunkQ queries.
    struct Group {
        uint32_t b = 0, e = 0;
        size_t firstChunk = 0, nChunks = 0;
        sd_agg *agg = nullptr;
    };
    std::vector<Group> groups;
    {
        bool contiguous = true;
        for (uint32_t i = 1; i < qdb->n && contiguous; i++) contiguous = qv.setId[i] >= qv.setId[i - 1];
        if (contiguous && qdb->n) {
            uint32_t b0 = 0;
            for (uint32_t i = 1; i <= qdb->n; i++)
                if (i == qdb->n || qv.setId[i] != qv.setId[b0]) {
                    Group g;
                    g.b = b0;
                    g.e = i;
                    groups.push_back(g);
                    b0 = i;
                }
        } else if (qdb->n) {
            Group g;
            g.e = qdb->n;
            groups.push_back(g);
        }
    }
    std::vector<std::pair<uint32_t, uint32_t> > chunks;
    std::vector<size_t> groupOfChunk;
    for (size_t gi = 0; gi < groups.size(); gi++) {
        Group &g = groups[gi];
        const uint32_t n = g.e - g.b;
        const uint32_t parts = (n + chunkQ - 1) / chunkQ;
        g.firstChunk = chunks.size();
        g.nChunks = parts;
        uint32_t c0 = g.b;
        for (uint32_t x = 0; x < parts; x++) {
            const uint32_t c1 = c0 + n / parts + (x < n % parts ? 1u : 0u);
            chunks.push_back(std::make_pair(c0, c1));
            groupOfChunk.push_back(gi);
            c0 = c1;
        }
    }
    std::vector<std::unique_ptr<ChunkOut> > outs(chunks.size());
    for (auto &o : outs) o.reset(new ChunkOut());
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    // at most nWorkers + 2 finished chunks wait for the aggregation (their records are the memory this path holds)
    size_t added = 0;
    auto work = [&](int wi) {
        Worker &W = *workers[(size_t) wi];
        omp_set_num_threads(W.threads);
        for (;;) {
            const size_t x = next.fetch_add(1);
            if (x >= chunks.size() || failed.load()) break;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return x < added + (size_t) nWorkers + 2 || failed.load(); });
            }
            ChunkOut &o = *outs[x];
            const int rc = processChunk(S, W, chunks[x].first, chunks[x].second, o);
            if (rc != SD_OK) failed.store(1);
            {
                std::lock_guard<std::mutex> lk(mu);
                o.done = true;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> threadsV;
    for (int w = 0; w < nWorkers; w++) threadsV.emplace_back(work, w);
    struct Join {
        std::vector<std::thread> &v;
        std::atomic<int> &failed;
        std::condition_variable &cv;
        ~Join() {
            failed.store(1);
            cv.notify_all();
            for (std::thread &t : v)
                if (t.joinable()) t.join();
        }
    };

    // ---- what the finalising thread needs, prepared while the workers run their first chunks: set sizes, the logGamma table of
    // clusterhits (ClusterHits.cpp:259-271), the name tables of the TSV, a context of its own
    std::vector<int32_t> qLen(qdb->lens.begin(), qdb->lens.end()), tLen(tdb->lens.begin(), tdb->lens.end());
    const double evalThr = strtod(eUser.c_str(), nullptr);
    const int covMode = (int) a.integer("--cov-mode", 0);
    const float covThr = (float) a.real("-c", 0.0);
    const int alnLenThr = (int) a.integer("--min-aln-len", 0);
    const int filterSelf = a.flag("--filter-self-match", false) ? 1 : 0;
    sd_ch_params ch;
    memset(&ch, 0, sizeof(ch));
    ch.maxGeneGap = (uint32_t) a.integer("--max-gene-gap", 3);
    ch.clusterSize = (uint32_t) a.integer("--cluster-size", 2);
    ch.alpha = a.real("--alpha", 1.0);
    ch.pCluThr = (float) a.real("--cluster-pval", 0.01);
    ch.pMHThr = (float) a.real("--multihit-pval", 0.01);
    std::string failure;
    uint64_t notComputed = 0, prefHits = 0, aligned = 0, accepted = 0, pfStats[5] = {0, 0, 0, 0, 0};
    double sec[8][IT_N], cpu[8][IT_N];
    uint64_t cnt[8][4];
    memset(cnt, 0, sizeof(cnt));
    memset(sec, 0, sizeof(sec));
    memset(cpu, 0, sizeof(cpu));
    double tAgg = 0, tFinalize = 0;
    uint64_t nEntriesAll = 0, nCluAll = 0, nCluLines = 0, nHitLines = 0;
    double tSearched = 0;
    struct CtxG {
        sd_ctx *c = nullptr;
        ~CtxG() { if (c) sd_ctx_destroy(c); }
    } finCtx;
    {
        Join join{threadsV, failed, cv};
        if (sd_ctx_create(device, &finCtx.c) != SD_OK) return fail("sd_ctx_create failed");
        std::vector<uint32_t> qSetSize(qs.nSets, 0), tSetSize(tsP->nSets, 0);
        std::vector<double> lgamma;
        {
            for (uint32_t i = 0; i < qdb->n; i++)
                if (qv.setId[i] < qs.nSets) qSetSize[qv.setId[i]]++;
            uint32_t m = 0;
            for (uint32_t i = 0; i < tdb->n; i++) {
                if (tv.setId[i] < tsP->nSets) tSetSize[tv.setId[i]]++;
                m = std::max(m, tv.pos[i]);
            }
            for (uint32_t i = 0; i < qdb->n; i++) m = std::max(m, qv.pos[i]);
            for (uint32_t v : qSetSize) m = std::max(m, v);
            for (uint32_t v : tSetSize) m = std::max(m, v);
            lgamma.resize((size_t) m + 8);
            sd_host_lgamma_table(lgamma.data(), (uint32_t) lgamma.size());
        }
        std::string qn, tn, qsrc, tsrc;
        std::vector<uint64_t> qno, tno, qso, tso;
        {
            std::vector<std::string> names(qdb->n);
            for (uint32_t i = 0; i < qdb->n; i++) names[i] = qdb->keys[i] < qs.nameOfKey.size() ? qs.nameOfKey[qdb->keys[i]] : std::string();
            packNames(names, qn, qno);
            names.assign(tdb->n, std::string());
            for (uint32_t i = 0; i < tdb->n; i++) names[i] = tdb->keys[i] < tsP->nameOfKey.size() ? tsP->nameOfKey[tdb->keys[i]] : std::string();
            packNames(names, tn, tno);
            packNames(qs.sourceOfSet, qsrc, qso);
            packNames(tsP->sourceOfSet, tsrc, tso);
        }
        {   // the TSV exists (and is empty) whatever follows
            uint64_t z0 = 0, z1 = 0;
            if (sd_records_write_tsv(nullptr, 0, tsvPath.c_str(), 0, 0, qn.data(), qno.data(), tn.data(), tno.data(), qsrc.data(), qso.data(), tsrc.data(),
                                     tso.data(), 0, &z0, &z1) != SD_OK)
                return fail("cannot write " + tsvPath);
        }
        // besthitbyset ... combinehits are done when a group's last chunk is in its sd_agg; clusterhits and the group's part of the TSV
        // (R/data/clustersearch.sh:141-151), groups in order (cluster keys run through the file)
        std::string finError;
        auto finalizeGroup = [&](sd_agg *agg) -> int {
            const double t0 = nowS();
            uint64_t ne = 0, nh = 0;
            int rc = sd_agg_finish(agg, &ne, &nh);
            if (rc != SD_OK) {
                finError = "sd_agg_finish failed (" + std::to_string(rc) + ")";
                return rc;
            }
            std::vector<uint64_t> entryOff(ne + 1, 0);
            std::vector<uint32_t> entryQ(std::max<uint64_t>(ne, 1)), entryT(std::max<uint64_t>(ne, 1)), hitQ(std::max<uint64_t>(nh, 1)),
                hitT(std::max<uint64_t>(nh, 1));
            std::vector<double> pval(std::max<uint64_t>(nh, 1));
            rc = sd_agg_get(agg, entryOff.data(), entryQ.data(), entryT.data(), hitQ.data(), hitT.data(), pval.data());
            if (rc != SD_OK) {
                finError = "sd_agg_get failed (" + std::to_string(rc) + ")";
                return rc;
            }
            if (const char *dq = getenv("SD_ITER_DEBUG_Q")) {   // (debugging aid: what the aggregation kept of some queries)
                for (const char *tok = dq; tok && *tok; tok = strchr(tok, ',') ? strchr(tok, ',') + 1 : nullptr) {
                    const uint32_t want = (uint32_t) atol(tok);
                    FILE *df = getenv("SD_ITER_DEBUG_FILE") ? fopen(getenv("SD_ITER_DEBUG_FILE"), "a") : stderr;
                    if (!df) df = stderr;
                    for (uint64_t e = 0; e < ne; e++)
                        for (uint64_t h = entryOff[e]; h < entryOff[e + 1]; h++)
                            if (hitQ[h] == want)
                                fprintf(df, "[iter debug] aggregated: entry (%u, %u) hit %llu of %llu: q %u t %u (key %u) pval %.4g\n", entryQ[e], entryT[e],
                                        (unsigned long long) (h - entryOff[e]), (unsigned long long) (entryOff[e + 1] - entryOff[e]), hitQ[h], hitT[h],
                                        tdb->keys[hitT[h]], pval[h]);
                    if (df != stderr) fclose(df);
                }
            }
            std::vector<uint32_t> clusterOf(std::max<uint64_t>(nh, 1), UINT32_MAX), rank(std::max<uint64_t>(nh, 1), 0),
                nClusters(std::max<uint64_t>(ne, 1), 0), cSize(std::max<uint64_t>(nh, 1), 0);
            std::vector<double> pCO(std::max<uint64_t>(nh, 1), 0.0), pMH(std::max<uint64_t>(nh, 1), 0.0);
            if (nh > 0) {
                std::vector<uint32_t> qp(nh), tp(nh), nqOf(ne);
                std::vector<uint8_t> sd(nh);
                for (uint64_t h = 0; h < nh; h++) {
                    qp[h] = qv.pos[hitQ[h]];
                    tp[h] = tv.pos[hitT[h]];
                    sd[h] = (uint8_t) (qv.strand[hitQ[h]] | (tv.strand[hitT[h]] << 1));
                }
                for (uint64_t e = 0; e < ne; e++) nqOf[e] = qSetSize[entryQ[e]];
                rc = sd_clusterhits_batch(finCtx.c, &ch, (uint32_t) ne, entryOff.data(), qp.data(), tp.data(), sd.data(), pval.data(), nqOf.data(),
                                          lgamma.data(), (uint32_t) lgamma.size(), clusterOf.data(), rank.data(), nClusters.data(), pCO.data(),
                                          pMH.data(), cSize.data());
                if (rc != SD_OK) {
                    finError = std::string("sd_clusterhits_batch failed (") + std::to_string(rc) + "): " + sd_last_error(finCtx.c);
                    return rc;
                }
                for (uint64_t e = 0; e < ne; e++) nCluAll += nClusters[e];
            }
            nEntriesAll += ne;
            std::vector<char> rec;
            uint64_t need = 0;
            rc = sd_agg_records(agg, clusterOf.data(), rank.data(), nClusters.data(), pCO.data(), pMH.data(), cSize.data(), nullptr, 0, &need);
            if (rc == SD_OK) {
                rec.resize(need);
                rc = sd_agg_records(agg, clusterOf.data(), rank.data(), nClusters.data(), pCO.data(), pMH.data(), cSize.data(), rec.data(), need, &need);
            }
            if (rc != SD_OK) {
                finError = "sd_agg_records failed (" + std::to_string(rc) + ")";
                return rc;
            }
            uint64_t nc = 0, nhl = 0;
            rc = sd_records_write_tsv(rec.data(), rec.size(), tsvPath.c_str(), 1, nCluLines, qn.data(), qno.data(), tn.data(), tno.data(), qsrc.data(),
                                      qso.data(), tsrc.data(), tso.data(), 0, &nc, &nhl);
            if (rc != SD_OK) {
                finError = "sd_records_write_tsv failed (" + std::to_string(rc) + ")";
                return rc;
            }
            nCluLines += nc;
            nHitLines += nhl;
            tFinalize += nowS() - t0;
            return SD_OK;
        };
        // the finalising thread: groups in order
        std::mutex fmu;
        std::condition_variable fcv;
        std::vector<sd_agg *> finQueue;
        bool finClose = false;
        int finStatus = SD_OK;
        std::thread finThread([&] {
            omp_set_num_threads(std::max(2, threads / 4));
            size_t at = 0;
            for (;;) {
                sd_agg *agg = nullptr;
                {
                    std::unique_lock<std::mutex> lk(fmu);
                    fcv.wait(lk, [&] { return at < finQueue.size() || finClose; });
                    if (at >= finQueue.size()) return;
                    agg = finQueue[at++];
                }
                if (finStatus == SD_OK) {
                    const int rcF = finalizeGroup(agg);
                    if (rcF != SD_OK) {
                        finStatus = rcF;
                        failed.store(1);
                        cv.notify_all();
                    }
                }
                sd_agg_destroy(agg);
            }
        });
        struct FinJoin {
            std::thread &t;
            std::mutex &m;
            std::condition_variable &c;
            bool &close;
            ~FinJoin() {
                {
                    std::lock_guard<std::mutex> lk(m);
                    close = true;
                }
                c.notify_all();
                if (t.joinable()) t.join();
            }
        };
        {
            FinJoin finJoin{finThread, fmu, fcv, finClose};
            // ---- the aggregation takes the chunks in order while the workers go on
            for (size_t x = 0; x < chunks.size(); x++) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return outs[x]->done || failed.load(); });
                    if (!outs[x]->done) break;   // a worker (or the finalising thread) failed
                }
                ChunkOut &o = *outs[x];
                if (o.rc != SD_OK) {
                    failure = o.err;
                    break;
                }
                Group &g = groups[groupOfChunk[x]];
                const double t0 = nowS();
                if (!g.agg) {
                    int rc = sd_agg_create(qv.setId.data(), qLen.data(), qdb->n, tv.setId.data(), tLen.data(), tdb->n, qs.nSets, tsP->nSets, evalThr, covMode,
                                           covThr, alnLenThr, filterSelf, &g.agg);
                    if (rc != SD_OK) {
                        failure = "sd_agg_create failed (" + std::to_string(rc) + ")";
                        break;
                    }
                    sd_agg_set_keys(g.agg, qdb->keys.data(), tdb->keys.data());
                    sd_agg_set_list_order(g.agg, 1);   // a query's records are the lines of the merged alignment DB: iteration after iteration, each list sorted
                }
                if (!o.pq.empty()) {
                    const int rc = sd_agg_add(g.agg, (uint32_t) o.pq.size(), o.g0, o.pq.data(), o.pt.data(), o.res.data(), o.ident.data(), o.pool.data());
                    if (rc != SD_OK) {
                        failure = "sd_agg_add failed (" + std::to_string(rc) + ")";
                        break;
                    }
                }
                tAgg += nowS() - t0;
                notComputed += o.notComputed;
                if (o.notComputed && notComputed == o.notComputed) fprintf(stderr, "sdgpu clustersearch: %s\n", o.err.c_str());
                prefHits += o.prefHits;
                aligned += o.aligned;
                accepted += o.accepted;
                for (int k = 0; k < 5; k++) pfStats[k] += o.pfStats[k];
                for (int s = 0; s < 8; s++)
                    for (int k = 0; k < IT_N; k++) {
                        sec[s][k] += o.sec[s][k];
                        cpu[s][k] += o.cpu[s][k];
                    }
                for (int s = 0; s < 8; s++)
                    for (int k = 0; k < 4; k++) cnt[s][k] += o.cnt[s][k];
                outs[x].reset(new ChunkOut());   // the records are in the aggregation now
                if (x + 1 == g.firstChunk + g.nChunks) {   // the group is complete: over to the finalising thread, which owns its sd_agg from here
                    {
                        std::lock_guard<std::mutex> lk(fmu);
                        finQueue.push_back(g.agg);
                    }
                    g.agg = nullptr;
                    fcv.notify_all();
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    added = x + 1;
                }
                cv.notify_all();
            }
            tSearched = nowS();
        }   // (the finalising thread has written every complete group)
        for (Group &g : groups)
            if (g.agg) {
                sd_agg_destroy(g.agg);
                g.agg = nullptr;
            }
        if (failure.empty() && finStatus != SD_OK) failure = finError;
        if (failure.empty() && failed.load())
            for (auto &o : outs)
                if (o && o->rc != SD_OK) {
                    failure = o->err;
                    break;
                }
        if (failure.empty() && failed.load()) failure = "a worker failed";
    }   // (workers joined)
    if (!failure.empty()) return fail(failure);
    for (int s = 0; s < std::min(numIt, 8); s++)
        info(a, "iteration %d: prefilter %.2f s | align %.2f s | result2profile %.2f s (summed over %d workers)%s\n", s, sec[s][IT_PREF], sec[s][IT_ALIGN],
             sec[s][IT_R2P], nWorkers,
             nWorkers == 1 ? (" | process CPU " + std::to_string(cpu[s][IT_PREF]) + " / " + std::to_string(cpu[s][IT_ALIGN]) + " / " + std::to_string(cpu[s][IT_R2P]) + " s").c_str() : "");
    for (int s = 0; s < std::min(numIt, 8); s++)
        info(a, "iteration %d: %llu prefilter rows, %llu after subtracting the aligned targets, %llu alignments calculated, %llu passed the thresholds\n", s,
             (unsigned long long) cnt[s][0], (unsigned long long) cnt[s][1], (unsigned long long) cnt[s][2], (unsigned long long) cnt[s][3]);
    info(a, "%llu prefilter hits, %llu alignments calculated, %llu sequence pairs passed the thresholds\n", (unsigned long long) prefHits,
         (unsigned long long) aligned, (unsigned long long) accepted);
    info(a, "prefilter stats: kmers %llu index_hits %llu diagonals %llu diag_len %llu query_residues %llu\n", (unsigned long long) pfStats[0],
         (unsigned long long) pfStats[1], (unsigned long long) pfStats[2], (unsigned long long) pfStats[3], (unsigned long long) pfStats[4]);
    if (S.profile) {   // kernel event times by name, all workers: one JSON object on stderr
        std::string js = "{";
        std::vector<std::pair<std::string, std::pair<double, uint64_t> > > tot;
        for (auto &Wp : workers) {
            char buf[8192];
            buf[0] = 0;
            sd_profile_names(Wp->ctx, buf, sizeof(buf));
            for (char *tok = strtok(buf, ","); tok; tok = strtok(nullptr, ",")) {
                double ms = 0;
                uint64_t n = 0;
                if (sd_profile_get(Wp->ctx, tok, &ms, &n) != SD_OK) continue;
                bool found = false;
                for (auto &e : tot)
                    if (e.first == tok) {
                        e.second.first += ms;
                        e.second.second += n;
                        found = true;
                    }
                if (!found) tot.push_back(std::make_pair(std::string(tok), std::make_pair(ms, n)));
            }
        }
        for (size_t i = 0; i < tot.size(); i++) {
            char line[256];
            snprintf(line, sizeof(line), "%s\"%s\": [%.3f, %llu]", i ? ", " : "", tot[i].first.c_str(), tot[i].second.first,
                     (unsigned long long) tot[i].second.second);
            js += line;
        }
        js += "}";
        fprintf(stderr, "[iter profile] %s\n", js.c_str());
    }
    const double tEnd = nowS();
    info(a, "%llu clusters from %llu set pairs\n", (unsigned long long) nCluAll, (unsigned long long) nEntriesAll);
    info(a, "%llu clusters with %llu hits written\n", (unsigned long long) nCluLines, (unsigned long long) nHitLines);
    info(a, "in-memory iterations (%d workers, chunks of <= %u queries, %zu query sets): load %.2f s | target resident %.2f s | iterations %.2f s (aggregation "
            "%.2f s inside; clusterhits + TSV of the finished sets %.2f s beside them) | after the last chunk %.2f s | total %.2f s\n",
         nWorkers, chunkQ, groups.size(), tLoaded - tStart, tResident - tLoaded, tSearched - tResident, tAgg, tFinalize, tEnd - tSearched, tEnd - tStart);
    if (getenv("SD_DEBUG_TIMING"))
        fprintf(stderr, "[iter] load %.2f s | target resident %.2f s | iterations %.2f s (agg %.2f, finalise %.2f beside) | tail %.2f s | total %.2f s\n",
                tLoaded - tStart, tResident - tLoaded, tSearched - tResident, tAgg, tFinalize, tEnd - tSearched, tEnd - tStart);
    if (notComputed)
        return fail(std::to_string(notComputed) + " queries need the reference's double-overflow route (or have >= 2^32 index hits) and were taken as "
                    "queries without rows; every other result is complete");
    return 0;
}

}  // namespace sdcli
