// The three hot modules behind the reference's command lines and DB files:
//   prefilter   <queryDB> <targetDB> <resultDB>             (M/src/prefiltering/Main.cpp:13, Prefiltering.cpp:570-951)
//   align       <queryDB> <targetDB> <prefDB> <alnDB>        (M/src/alignment/Main.cpp:12, Alignment.cpp:244-542)
//   clusterhits <querySetDB> <targetSetDB> <matches> <out>   (R/src/util/ClusterHits.cpp:215-511)
// DB in, C ABI of libsdgpu.so (HIP kernels) in the middle, DB out.  No compute here, no CPU fallback: without a GPU
// sd_ctx_create fails and the module exits non-zero.
#include "sd_cli.h"
#include "sd_align_core.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace sdcli {

namespace {

struct HostH {
    sd_host *h = nullptr;
    bool own = true;   // false: the resident host object of a workflow (sd_cli.h)
    ~HostH() { if (h && own) sd_host_destroy(h); }
    int open(int threads) {
        if (resident().enabled) {
            h = resident().host(threads);
            own = false;
            return h ? SD_OK : SD_ENOMEM;
        }
        return sd_host_create(threads, &h);
    }
};
struct CtxH {
    sd_ctx *c = nullptr;
    bool own = true;   // false: the resident context of a workflow (sd_cli.h)
    ~CtxH() { if (c && own) sd_ctx_destroy(c); }
    // the module's context: the workflow's resident one when that is on, else its own
    int open(int device) {
        if (resident().enabled) {
            int rc = SD_OK;
            c = resident().ctx(device, &rc);
            own = false;
            return rc;
        }
        return sd_ctx_create(device, &c);
    }
};
struct SeqSetH {
    sd_seqset *s = nullptr;
    ~SeqSetH() { if (s && own) sd_seqset_destroy(s); }
    void reset() { if (s && own) sd_seqset_destroy(s); s = nullptr; }
    bool own = true;
};
struct TargetH {
    sd_target *t = nullptr;
    bool own = true;
    ~TargetH() { if (t && own) sd_target_destroy(t); }
};
struct IndexH {
    sd_host_index *ix = nullptr;
    ~IndexH() { if (ix) sd_host_index_destroy(ix); }
};

// what the modules refuse (the reference has these paths; this build does not)
int checkCommon(const Args &a) {
    if (a.integer("--compressed", 0) != 0) return fail("--compressed 1 is not supported");
    const std::string sm = a.multi("--sub-mat", "aa", "blosum62.out");
    if (sm != "blosum62.out") return fail("--sub-mat " + sm + ": only blosum62.out is built into this path");
    if (a.integer("--gpu", 0) != 0) return fail("--gpu 1 selects the reference's CUDA ungapped prefilter (a different algorithm); run without it");
    return 0;
}

int deviceOf(const Args &a) {
    if (a.has("--device")) return (int) a.integer("--device", 0);
    const char *lr = getenv("LOCAL_RANK");
    return lr ? atoi(lr) : 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
int prefilterSetupFromArgs(const Args &a, sd_host *host, const SeqDb &tdb, bool profileQueries, PrefSetup &s) {
    // parameters the way Prefiltering's constructor derives them (Prefiltering.cpp:180-215,1005-1065)
    s.k = (int) a.integer("-k", 0);
    if (s.k == 0) s.k = sd_host_auto_kmer_size(tdb.totalResidues());
    if (s.k != 6 && s.k != 7) return fail("-k " + std::to_string(s.k) + ": k-mer sizes 6 and 7 are implemented");
    const float sens = (float) a.real("-s", 4.0);
    const long long kScore = strtoll(a.multi("--k-score", profileQueries ? "prof" : "seq", "2147483647").c_str(), nullptr, 10);
    s.kmerThr = kScore != INT_MAX ? (int) kScore
                                  : (profileQueries ? sd_host_profile_kmer_threshold(sens, s.k) : sd_host_kmer_threshold(sens, s.k));
    s.indexThr = profileQueries ? 0 : s.kmerThr;   // profile searches index every k-mer (Prefiltering.cpp:525-527)
    s.mask = a.integer("--mask", 1) != 0;
    s.maskProb = a.real("--mask-prob", 0.9);
    s.includeIdentity = a.flag("--add-self-matches", false);
    s.compBias = a.integer("--comp-bias-corr", 1) != 0;
    sd_prefilter_params &par = s.par;
    memset(&par, 0, sizeof(par));
    par.kmerSize = s.k;
    par.kmerThr = s.kmerThr;
    par.maxHitsPerQuery = (int32_t) std::min<long long>(a.integer("--max-seqs", 300), tdb.n);   // Prefiltering.cpp:184
    par.minDiagScore = (int32_t) a.integer("--min-ungapped-score", 15);
    par.binSize = a.has("--bin-size") ? (uint32_t) a.integer("--bin-size", 0)
                                      : sd_host_bin_size(tdb.n, (uint64_t) a.integer("--l2-cache-size", 0));
    par.covMode = (int32_t) a.integer("--cov-mode", 0);
    par.covThr = (float) a.real("-c", 0.0);
    // the writer's coverage pre-filter only exists for these modes (Prefiltering.cpp:856-858)
    if (!(par.covMode == 0 || par.covMode == 2 || par.covMode == 5)) par.covThr = 0.0f;
    sd_host_matrix(host, 2, par.ungappedMatrix, nullptr, nullptr);
    if (par.maxHitsPerQuery < 1) par.maxHitsPerQuery = 1;
    return 0;
}

int prefilterModule(const Args &a) {
    if (a.pos.size() != 3) return fail("usage: prefilter <queryDB> <targetDB> <resultDB> [options]");
    if (int rc = checkCommon(a)) return rc;
    const std::string ssm = a.multi("--seed-sub-mat", "aa", "VTML80.out");
    if (ssm != "VTML80.out") return fail("--seed-sub-mat " + ssm + ": only VTML80.out is built into this path");
    if (a.integer("--spaced-kmer-mode", 1) != 1 || a.has("--spaced-kmer-pattern"))
        return fail("only the default spaced k-mer patterns are supported (--spaced-kmer-mode 1)");
    if (a.integer("--exact-kmer-matching", 0) != 0) return fail("--exact-kmer-matching 1 is not supported");
    if (!a.flag("--diag-score", true)) return fail("--diag-score 0 is not supported");
    if (a.integer("--target-search-mode", 0) != 0) return fail("--target-search-mode 1 is not supported");
    if (a.integer("--mask-lower-case", 0) != 0 || a.integer("--mask-n-repeat", 0) != 0)
        return fail("--mask-lower-case / --mask-n-repeat are not supported");
    if (a.integer("--split", 0) > 1) return fail("--split > 1: the whole target index is resident in HBM here; run with --split 0 or 1");
    if (a.multi("--alph-size", "aa", "21") != "21") return fail("--alph-size aa:21 only");
    if (a.real("--comp-bias-corr-scale", 1.0) != 1.0) return fail("--comp-bias-corr-scale 1 only");
    if (a.has("--taxon-list") && !a.str("--taxon-list", "").empty()) return fail("--taxon-list is not supported");
    const bool compBias = a.integer("--comp-bias-corr", 1) != 0;
    const int threads = threadsOf(a);

    Lap lap("prefilter");
    HostH host;
    if (host.open(threads) != SD_OK) return fail("sd_host_create failed");
    std::string err;
    const bool sameDb = a.pos[0] == a.pos[1];
    std::shared_ptr<SeqDb> tdb = loadTargetDb(a.pos[1], host.h, &err);
    std::unique_ptr<SeqDb> qdbOwn;
    if (!tdb) return fail(err);
    if (tdb->profile) return fail("profile target databases are not supported on this path");
    SeqDb *qdb = tdb.get();
    if (!sameDb) {
        qdbOwn.reset(new SeqDb());
        if (!qdbOwn->load(a.pos[0], host.h, &err)) return fail(err);
        qdb = qdbOwn.get();
    }
    lap.mark("load DBs");
    info(a, "Query database size: %u type: %s\nTarget database size: %u type: Aminoacid\n", qdb->n,
         qdb->profile ? "Profile" : "Aminoacid", tdb->n);

    PrefSetup PS;
    if (int rcS = prefilterSetupFromArgs(a, host.h, *tdb, qdb->profile, PS)) return rcS;
    const int k = PS.k, kmerThr = PS.kmerThr;
    const bool mask = PS.mask, includeIdentity = PS.includeIdentity;
    const double maskProb = PS.maskProb;

    CtxH ctx;
    int rc = ctx.open(deviceOf(a));
    if (rc != SD_OK) return fail("no usable HIP device (sd_ctx_create returned " + std::to_string(rc) + "); this path has no CPU fallback");

    lap.mark("context");
    // target side: TARGET.idx when a createindex file with matching parameters lies next to the DB (PrefilteringIndexReader
    // layout, sd_mod_index.cpp), else IndexBuilder::fillDatabase on the device (sd_target_build: mask + lists); resident in HBM afterwards.
    // Profile searches index every k-mer (Prefiltering.cpp:525-527)
    const int indexThr = qdb->profile ? 0 : kmerThr;
    IndexH index;
    LoadedIndex loaded;
    std::string why;
    uint64_t nEntries = 0, maskedRes = 0;
    const uint32_t *kOff = nullptr, *eSeq = nullptr;
    const uint16_t *ePos = nullptr;
    const uint8_t *masked = nullptr;
    const uint64_t *kBase = nullptr;
    const int16_t *s2, *s3;
    const uint16_t *i2, *i3;
    uint32_t sz2, sz3;
    sd_host_ext_matrix(host.h, 2, &s2, &i2, &sz2);
    sd_host_ext_matrix(host.h, 3, &s3, &i3, &sz3);
    TargetH target;
    // a workflow's resident target of exactly this index (same DB, k, threshold, masking, device) is taken as it is
    char tkey[96];
    snprintf(tkey, sizeof(tkey), "|%d|%d|%d|%.6f|%d", k, indexThr, mask ? 1 : 0, maskProb, deviceOf(a));
    const std::string targetKey = a.pos[1] + tkey;
    if (resident().enabled && resident().targets.count(targetKey)) {
        const Resident::TargetEntry &te = resident().targets[targetKey];
        target.t = te.t;
        target.own = false;
        nEntries = te.nEntries;
        maskedRes = te.masked;
        info(a, "Target index resident from the previous module of this workflow\n");
    }
    const int got = target.t ? 1 : loadTargetIndex(a.pos[1], k, indexThr, mask ? 1 : 0, tdb->n, tdb->totalResidues(), loaded, &why);
    if (got < 0) return fail(why);
    if (got == 0) {
        kOff = loaded.offsets.data();
        kBase = loaded.blockBase.empty() ? nullptr : loaded.blockBase.data();
        eSeq = loaded.entrySeq.data();
        ePos = loaded.entryPos.data();
        masked = loaded.masked.data();
        nEntries = loaded.nEntries;
        info(a, "Use index %s.idx\n", a.pos[1].c_str());
    } else {
        if (sddb::fileExists(a.pos[1] + ".idx.index")) info(a, "Index file not used: %s\n", why.c_str());
    }
    if (target.t) {
        // resident
    } else if (got != 0 && !getenv("SD_INDEX_HOST")) {
        // IndexBuilder::fillDatabase on the device (sd_target_build): mask, k-mer lists, list starts
        double ratios[21 * 21];
        int8_t self[21];
        sd_host_index_tables(host.h, ratios, self);
        uint64_t st[4] = {0, 0, 0, 0};
        rc = sd_target_build(ctx.c, k, indexThr, mask ? 1 : 0, maskProb, tdb->residues.data(), tdb->offsets.data(), tdb->n, ratios, self, s2, i2,
                             s3, i3, &target.t, st);
        if (rc != SD_OK) return failCtx(ctx.c, rc, "sd_target_build");
        nEntries = st[0];
        maskedRes = st[1];
    } else if (got != 0) {
        rc = sd_host_index_build(host.h, tdb->residues.data(), tdb->offsets.data(), tdb->n, k, indexThr, mask ? 1 : 0, maskProb, &index.ix);
        if (rc != SD_OK) return fail("sd_host_index_build failed (" + std::to_string(rc) + ")");
        uint64_t tableSize = 0;
        sd_host_index_info(index.ix, &tableSize, &nEntries, &maskedRes);
        sd_host_index_arrays(index.ix, &kOff, &eSeq, &ePos, &masked);
        sd_host_index_block_base(index.ix, &kBase, nullptr);
    }
    info(a, "Index table k-mer threshold: %d at k-mer size %d\nIndex statistics\nEntries:          %llu\n", kmerThr, k,
         (unsigned long long) nEntries);
    if (!target.t) {
        rc = sd_target_create_wide(ctx.c, k, kOff, kBase, eSeq, ePos, nEntries, masked, tdb->offsets.data(), tdb->n, s2, i2, s3, i3, &target.t);
        if (rc != SD_OK) return failCtx(ctx.c, rc, "sd_target_create");
    }
    if (resident().enabled && target.own) {   // stays for the next module of the workflow
        // one resident index per (DB, device): an index of the same DB with other parameters -- the sequence-threshold index of
        // iteration 0 once the profile iterations (threshold 0, a larger index) begin -- would only hold HBM until the workflow ends
        char dsuf[24];
        snprintf(dsuf, sizeof(dsuf), "|%d", deviceOf(a));
        const std::string pfx = a.pos[1] + "|", suf = dsuf;
        for (auto it = resident().targets.begin(); it != resident().targets.end();) {
            const std::string &key = it->first;
            if (key.compare(0, pfx.size(), pfx) == 0 && key.size() >= suf.size() && key.compare(key.size() - suf.size(), suf.size(), suf) == 0) {
                info(a, "Resident target index %s replaced\n", key.c_str());
                sd_target_destroy(it->second.t);
                it = resident().targets.erase(it);
            } else {
                ++it;
            }
        }
        Resident::TargetEntry te;
        te.t = target.t;
        te.nEntries = nEntries;
        te.masked = maskedRes;
        resident().targets[targetKey] = te;
        target.own = false;
    }

    lap.mark("target index");
    const sd_prefilter_params par = PS.par;

    sddb::Writer out;
    if (!out.open(a.pos[2], sddb::DBTYPE_PREFILTER_RES, &err)) return fail(err);

    const uint32_t chunk = (uint32_t) std::max<long long>(1, a.integer("--chunk-queries", 16384));
    std::vector<sd_hit> hits;
    std::vector<uint32_t> counts, ident;
    std::vector<int8_t> diagBias;
    std::vector<int16_t> kmerBias;
    std::vector<uint64_t> off;
    std::string text;
    uint64_t totalHits = 0, notComputed = 0;
    for (uint32_t c0 = 0; c0 < qdb->n; c0 += chunk) {
        const uint32_t c1 = std::min(qdb->n, c0 + chunk), nq = c1 - c0;
        const uint64_t r0 = qdb->offsets[c0], r1 = qdb->offsets[c1];
        off.resize((size_t) nq + 1);
        for (uint32_t i = 0; i <= nq; i++) off[i] = qdb->offsets[c0 + i] - r0;
        ident.resize(nq);
        for (uint32_t i = 0; i < nq; i++) {
            uint32_t id = UINT32_MAX;
            if (sameDb) id = c0 + i;
            else if (includeIdentity) {
                const size_t t = tdb->rd.idOfKey(qdb->keys[c0 + i]);
                if (t != SIZE_MAX) id = (uint32_t) t;
            }
            ident[i] = id;
        }
        hits.resize((size_t) nq * par.maxHitsPerQuery);
        counts.assign(nq, 0);
        if (qdb->profile) {
            rc = sd_prefilter_profile_batch(ctx.c, target.t, &par, nq, qdb->residues.data() + r0, off.data(),
                                            qdb->sortedScore.data() + r0 * 20, qdb->sortedIndex.data() + r0 * 20,
                                            qdb->alnProfile.data() + r0 * 21, ident.data(), hits.data(), counts.data(), nullptr);
        } else {
            diagBias.assign(r1 - r0 + 1, 0);
            kmerBias.assign(r1 - r0 + 1, 0);
            if (compBias)
                sd_host_comp_bias(host.h, qdb->residues.data() + r0, off.data(), nq, k, nullptr, diagBias.data(), kmerBias.data());
            rc = sd_prefilter_batch(ctx.c, target.t, &par, nq, qdb->residues.data() + r0, off.data(), kmerBias.data(),
                                    diagBias.data(), ident.data(), hits.data(), counts.data(), nullptr);
        }
        if (rc != SD_OK) return failCtx(ctx.c, rc, "sd_prefilter_batch");
        lap.mark("chunk: bias + device");
        // QueryMatcher::prefilterHitToBuffer (QueryMatcher.h:118-130): targetKey \t score \t (int16) diagonal
        char line[64];
        for (uint32_t i = 0; i < nq; i++) {
            text.clear();
            if (counts[i] == UINT32_MAX) {   // per-query error slot of sd_prefilter_batch: not computed
                if (notComputed < 5) fprintf(stderr, "sdgpu prefilter: query %u was not computed: %s\n", qdb->keys[c0 + i], sd_last_error(ctx.c));
                notComputed++;
                counts[i] = 0;
            }
            const sd_hit *row = hits.data() + (size_t) i * par.maxHitsPerQuery;
            for (uint32_t x = 0; x < counts[i]; x++) {
                const int len = snprintf(line, sizeof(line), "%u\t%d\t%d\n", tdb->keys[row[x].seqId], row[x].score,
                                         (int) (int16_t) row[x].diagonal);
                text.append(line, (size_t) len);
            }
            totalHits += counts[i];
            if (!out.write(qdb->keys[c0 + i], text.data(), text.size())) return fail("cannot write " + a.pos[2]);
        }
    }
    lap.mark("chunks: text + write");
    if (!out.close(&err)) return fail(err);
    lap.mark("close");
    info(a, "%llu prefilter hits written for %u queries\n", (unsigned long long) totalHits, qdb->n);
    if (notComputed)
        return fail(std::to_string(notComputed) + " queries need the reference's double-overflow route (or have >= 2^32 index hits) and were "
                    "written as empty entries; every other entry is complete");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
namespace {

int alignPairs(sd_ctx *ctx, const sd_sw_params &par, sd_seqset *qs, sd_seqset *ts, const SeqDb &qdb, const SeqDb &tdb,
               const std::vector<uint32_t> &qIdOfLocal, const std::vector<uint32_t> &pq, const std::vector<uint32_t> &pt,
               const std::vector<uint8_t> &ident, bool compact, std::vector<uint32_t> &outIdx, std::vector<sd_sw_result> &res,
               BtPool &pool) {
    const uint32_t n = (uint32_t) pq.size();
    res.resize(std::max<uint32_t>(n, 1));
    outIdx.resize(std::max<uint32_t>(n, 1));
    // a first guess that a second call rarely has to correct (a too small pool costs the whole batch again): a backtrace has at
    // most qLen + tLen columns and, for the full-length homologs that dominate, about min(qLen, tLen) of them
    uint64_t cap = 1u << 20;
    if (par.swMode == 2) {
        uint64_t est = 0;
        for (uint32_t i = 0; i < n; i++) est += (uint64_t) std::min(qdb.lens[qIdOfLocal[pq[i]]], tdb.lens[pt[i]]) + 16;
        cap += est + est / 4;
    }
    bool exact = false;
    for (;;) {
        pool.reserve(cap);
        uint64_t used = 0;
        int rc;
        uint32_t nOut = n;
        if (compact)
            rc = sd_sw_align_batch_compact(ctx, &par, qs, ts, n, pq.data(), pt.data(), ident.data(), outIdx.data(), res.data(),
                                           &nOut, pool.p.get(), pool.cap, &used);
        else
            rc = sd_sw_align_batch(ctx, &par, qs, ts, n, pq.data(), pt.data(), ident.data(), res.data(), pool.p.get(), pool.cap, &used);
        if (rc == SD_ENOMEM && !exact) {   // pool too small: the exact bound is sum(qLen + tLen)
            uint64_t need = 64;
            for (uint32_t i = 0; i < n; i++) need += (uint64_t) qdb.lens[qIdOfLocal[pq[i]]] + (uint64_t) tdb.lens[pt[i]];
            cap = need;
            exact = true;
            continue;
        }
        if (rc != SD_OK) return rc;
        if (compact) {
            res.resize(nOut);
            outIdx.resize(nOut);
        } else {
            for (uint32_t i = 0; i < n; i++) outIdx[i] = i;
        }
        return SD_OK;
    }
}

struct SeqSetGuard {
    sd_seqset *s = nullptr;
    ~SeqSetGuard() { if (s) sd_seqset_destroy(s); }
};

}  // namespace

int alignSetupFromArgs(const Args &a, sd_host *host, uint64_t targetResidues, AlignSetup &s) {
    if (a.flag("--wrapped-scoring", false)) return fail("--wrapped-scoring is a nucleotide mode");
    if (a.integer("--alt-ali", 0) != 0) return fail("--alt-ali > 0 is not supported");
    if (a.integer("--alignment-output-mode", 0) != 0) return fail("--alignment-output-mode 0 only");
    if (a.real("--score-bias", 0.0) != 0.0) return fail("--score-bias 0 only");
    if (a.real("--corr-score-weight", 0.0) != 0.0) return fail("--corr-score-weight 0 only");
    if (a.multi("--gap-open", "aa", "11") != "11" || a.multi("--gap-extend", "aa", "1") != "1")
        return fail("gap costs other than --gap-open 11 --gap-extend 1 need other E-value parameters than the built-in preset");
    if (a.real("--comp-bias-corr-scale", 1.0) != 1.0) return fail("--comp-bias-corr-scale 1 only");
    s.compBias = a.integer("--comp-bias-corr", 1) != 0;
    int alignmentMode = (int) a.integer("--alignment-mode", 0);
    if (alignmentMode == 4) return fail("Use rescorediagonal for ungapped alignment mode.");
    bool addBacktrace = a.flag("-a", false);
    s.realign = a.flag("--realign", false);
    s.realignScoreBias = (float) a.real("--realign-score-bias", -0.2);
    if (s.realign && !(s.realignScoreBias == -0.2f || s.realignScoreBias == 0.0f))
        return fail("--realign-score-bias: -0.2 (default) and 0 are built in");
    float covThr = (float) a.real("-c", 0.0);
    s.canCovThr = covThr;
    s.covMode = (int) a.integer("--cov-mode", 0);
    const float seqIdThr = (float) a.real("--min-seq-id", 0.0);
    // Alignment::Alignment (Alignment.cpp:31-57)
    if (addBacktrace) alignmentMode = 3;
    int realignSwMode = 0;
    auto initSWMode = [](int mode, float cov, float sid) {   // Alignment::initSWMode (:170-192)
        switch (mode) {
            case 0: return (cov > 0.0f && sid == 0.0f) ? 1 : ((cov > 0.0f && sid > 0.0f) ? 2 : 0);
            case 2: return 1;
            case 3: return 2;
            default: return 0;
        }
    };
    float realignCov = 0.0f;
    if (s.realign) {
        realignSwMode = initSWMode(std::max(alignmentMode, 2), 0.0f, 0.0f);
        alignmentMode = 1;
        realignCov = covThr;
        covThr = 0.0f;
        addBacktrace = true;
    }
    s.swMode = initSWMode(alignmentMode, (float) a.real("-c", 0.0), seqIdThr);
    memset(&s.par, 0, sizeof(s.par));
    s.par.gapOpen = 11;
    s.par.gapExtend = 1;
    sd_host_matrix(host, 0, s.par.matrix, nullptr, nullptr);
    s.par.covMode = s.covMode;
    s.par.covThr = covThr;
    s.par.evalThr = a.real("-e", 0.001);
    s.par.swMode = s.swMode;
    s.par.dbResidues = targetResidues;
    s.rpar = s.par;   // the realigner (Alignment.cpp:296-303,419): score-biased matrix, E-value gate off
    if (s.realign) {
        sd_host_matrix(host, s.realignScoreBias == 0.0f ? 0 : 2, s.rpar.matrix, nullptr, nullptr);
        s.rpar.covThr = realignCov;
        s.rpar.evalThr = FLT_MAX;
        s.rpar.swMode = realignSwMode;
    }
    memset(&s.crit, 0, sizeof(s.crit));
    s.crit.evalThr = s.par.evalThr;
    s.crit.seqIdThr = seqIdThr;
    s.crit.alnLenThr = (int32_t) a.integer("--min-aln-len", 0);
    s.crit.covMode = s.covMode;
    s.crit.covThr = s.realign ? realignCov : covThr;
    s.crit.seqIdMode = (int32_t) a.integer("--seq-id-mode", 0);
    s.crit.swMode = s.swMode;
    s.crit.addBacktrace = addBacktrace ? 1 : 0;
    s.crit.realign = s.realign ? 1 : 0;
    s.crit.realignSwMode = realignSwMode;
    s.crit.realignMaxSeqs = (int32_t) std::min<long long>(a.integer("--realign-max-seqs", INT_MAX), INT_MAX);
    s.crit.maxAccept = (uint32_t) std::min<long long>(a.integer("--max-accept", INT_MAX), INT_MAX);
    s.crit.maxRejected = (uint32_t) std::min<long long>(a.integer("--max-rejected", INT_MAX), INT_MAX);
    s.stopRules = s.crit.maxAccept != (uint32_t) INT_MAX || s.crit.maxRejected != (uint32_t) INT_MAX;
    s.includeIdentity = a.flag("--add-self-matches", false);
    return 0;
}

int alignChunkCore(sd_ctx *ctx, sd_host *host, const AlignSetup &s, const SeqDb &qdb, const SeqDb &tdb, sd_seqset *tset, AlignChunk &c, Lap *lap,
                   const char **what) {
    static const char *none = "";
    const char *dummyWhat;
    if (!what) what = &dummyWhat;
    *what = none;
    int rc = SD_OK;
    const std::vector<uint32_t> &localQ = c.localQ, &pq = c.pq, &pt = c.pt;
    const std::vector<uint8_t> &ident = c.ident;
    const uint32_t nq = (uint32_t) localQ.size();
    // queries of the chunk as one sequence set on the device
    c.qoff.assign((size_t) nq + 1, 0);
    c.qlen.resize(nq);
    for (uint32_t i = 0; i < nq; i++) {
        c.qlen[i] = qdb.lens[localQ[i]];
        c.qoff[i + 1] = c.qoff[i] + (uint64_t) c.qlen[i];
    }
    c.qres.resize(c.qoff[nq] + 1);
    for (uint32_t i = 0; i < nq; i++) memcpy(c.qres.data() + c.qoff[i], qdb.residues.data() + qdb.offsets[localQ[i]], (size_t) c.qlen[i]);
    SeqSetGuard qset;
    if (nq) {
        if (qdb.profile) {
            c.qaln.resize((c.qoff[nq] + 1) * 21);
            for (uint32_t i = 0; i < nq; i++)
                memcpy(c.qaln.data() + c.qoff[i] * 21, qdb.alnProfile.data() + qdb.offsets[localQ[i]] * 21, (size_t) c.qlen[i] * 21);
            rc = sd_profileset_create(ctx, c.qres.data(), c.qoff.data(), nq, c.qaln.data(), &qset.s);
        } else {
            c.qbias.assign(c.qoff[nq] + 1, 0);
            if (s.compBias) sd_host_comp_bias(host, c.qres.data(), c.qoff.data(), nq, 6, c.qbias.data(), nullptr, nullptr);
            rc = sd_seqset_create(ctx, c.qres.data(), c.qoff.data(), nq, c.qbias.data(), &qset.s);
        }
        if (rc != SD_OK) {
            *what = "sd_seqset_create(queries)";
            return rc;
        }
    }
    if (lap) lap->mark("chunk: query set");
    // the pairs that are aligned (pre-rejected ones are not)
    c.apq.clear();
    c.apt.clear();
    c.aid.clear();
    c.aIdx.clear();
    c.apq.reserve(pq.size());
    for (size_t i = 0; i < pq.size(); i++)
        if (ident[i] != 2) {
            c.apq.push_back(pq[i]);
            c.apt.push_back(pt[i]);
            c.aid.push_back(ident[i]);
            c.aIdx.push_back((uint32_t) i);
        }
    c.aligned = c.apq.size();
    const bool compact = s.swMode == 2 && !s.stopRules;
    if (!c.apq.empty()) {
        rc = alignPairs(ctx, s.par, qset.s, tset, qdb, tdb, localQ, c.apq, c.apt, c.aid, compact, c.idxOut, c.res, c.pool);
        if (rc != SD_OK) {
            *what = "sd_sw_align_batch";
            return rc;
        }
    } else {
        c.res.clear();
        c.idxOut.clear();
    }
    if (lap) lap->mark("chunk: alignPairs");
    // record list handed to the criteria: compact -> only the reportable records; otherwise every pair in prefilter
    // order, pre-rejected ones as records that fail every criterion (E-value NaN)
    c.recQ.clear();
    c.recT.clear();
    c.recIdent.clear();
    std::vector<sd_sw_result> *recs = &c.res;
    if (compact) {
        c.recQ.resize(c.res.size());
        c.recT.resize(c.res.size());
        c.recIdent.resize(c.res.size());
        for (size_t x = 0; x < c.res.size(); x++) {
            c.recQ[x] = c.apq[c.idxOut[x]];
            c.recT[x] = c.apt[c.idxOut[x]];
            c.recIdent[x] = c.aid[c.idxOut[x]];
        }
    } else {
        c.full.resize(pq.size());
        sd_sw_result dummy;
        memset(&dummy, 0, sizeof(dummy));
        dummy.qStart = dummy.tStart = dummy.qEnd = dummy.tEnd = -1;
        dummy.evalue = NAN;
        for (size_t i = 0; i < pq.size(); i++) c.full[i] = dummy;
        for (size_t x = 0; x < c.aIdx.size(); x++) c.full[c.aIdx[x]] = c.res[x];
        c.recQ = pq;
        c.recT = pt;
        c.recIdent.resize(pq.size());
        for (size_t i = 0; i < pq.size(); i++) c.recIdent[i] = ident[i] == 1 ? 1 : 0;
        recs = &c.full;
    }
    c.order.resize(std::max<size_t>(recs->size(), 1));
    c.counts.assign(std::max<uint32_t>(nq, 1), 0);
    rc = sd_host_accept_sort(&s.crit, nq, (uint32_t) recs->size(), c.recQ.data(), c.recT.data(), recs->data(), c.recIdent.data(),
                             c.qlen.data(), tdb.lens.data(), tdb.keys.data(), c.order.data(), c.counts.data());
    if (rc != SD_OK) {
        *what = "sd_host_accept_sort";
        return rc;
    }
    uint64_t nAcc = 0;
    for (uint32_t i = 0; i < nq; i++) nAcc += c.counts[i];
    c.accepted = nAcc;
    c.outRecs = recs;
    c.outOrder = &c.order;
    c.outCounts = &c.counts;
    c.outT = &c.recT;
    c.outIdent = &c.recIdent;
    c.outPool = &c.pool;
    if (s.realign && nAcc > 0) {
        // second pass over the accepted records, in their order (Alignment.cpp:408-440)
        c.pq2.resize(nAcc);
        c.pt2.resize(nAcc);
        c.ident2.resize(nAcc);
        uint64_t w = 0;
        for (uint32_t q = 0; q < nq; q++)
            for (uint32_t x = 0; x < c.counts[q]; x++, w++) {
                const uint32_t i = c.order[w];
                c.pq2[w] = q;
                c.pt2[w] = c.recT[i];
                c.ident2[w] = c.recIdent[i];
            }
        // the realigner's query profile: composition bias against the score-biased matrix (a profile query carries its
        // scores itself and is reused)
        SeqSetGuard qset2;
        sd_seqset *rq = qset.s;
        if (!qdb.profile && s.compBias && s.realignScoreBias != 0.0f) {
            c.qbias2.assign(c.qoff[nq] + 1, 0);
            sd_host_sw_comp_bias(host, 2, c.qres.data(), c.qoff.data(), nq, c.qbias2.data());
            rc = sd_seqset_create(ctx, c.qres.data(), c.qoff.data(), nq, c.qbias2.data(), &qset2.s);
            if (rc != SD_OK) {
                *what = "sd_seqset_create(realign queries)";
                return rc;
            }
            rq = qset2.s;
        }
        rc = alignPairs(ctx, s.rpar, rq, tset, qdb, tdb, localQ, c.pq2, c.pt2, c.ident2, false, c.idx2, c.res2, c.pool2);
        if (rc != SD_OK) {
            *what = "sd_sw_align_batch(realign)";
            return rc;
        }
        c.merged.resize(nAcc);
        c.order2.resize(nAcc);
        c.counts2.assign(nq, 0);
        rc = sd_host_realign_select(&s.crit, nq, c.counts.data(), c.order.data(), c.recT.data(), recs->data(), c.res2.data(),
                                    c.ident2.data(), c.qlen.data(), tdb.lens.data(), tdb.keys.data(), c.merged.data(),
                                    c.order2.data(), c.counts2.data());
        if (rc != SD_OK) {
            *what = "sd_host_realign_select";
            return rc;
        }
        c.outRecs = &c.merged;
        c.outOrder = &c.order2;
        c.outCounts = &c.counts2;
        c.accT = c.pt2;
        c.outT = &c.accT;
        c.outIdent = &c.ident2;
        c.outPool = &c.pool2;
    } else if (s.realign) {
        c.counts2.assign(std::max<uint32_t>(nq, 1), 0);
        c.outCounts = &c.counts2;
    }
    if (lap) lap->mark("chunk: accept / sort (+ realign)");
    return SD_OK;
}

int alignModule(const Args &a) {
    if (a.pos.size() != 4) return fail("usage: align <queryDB> <targetDB> <prefilterDB> <alignmentDB> [options]");
    if (int rc = checkCommon(a)) return rc;
    const int threads = threadsOf(a);
    Lap lap("align");
    HostH host;
    if (host.open(threads) != SD_OK) return fail("sd_host_create failed");
    std::string err;
    const bool sameDb = a.pos[0] == a.pos[1];
    std::shared_ptr<SeqDb> tdb = loadTargetDb(a.pos[1], host.h, &err);
    std::unique_ptr<SeqDb> qdbOwn;
    if (!tdb) return fail(err);
    if (tdb->profile) return fail("profile target databases are not supported on this path");
    SeqDb *qdb = tdb.get();
    if (!sameDb) {
        qdbOwn.reset(new SeqDb());
        if (!qdbOwn->load(a.pos[0], host.h, &err)) return fail(err);
        qdb = qdbOwn.get();
    }
    AlignSetup S;
    if (int rcS = alignSetupFromArgs(a, host.h, tdb->totalResidues(), S)) return rcS;
    const int swMode = S.swMode, covMode = S.covMode;
    const float canCovThr = S.canCovThr;
    const bool includeIdentity = S.includeIdentity;
    lap.mark("load DBs");
    sddb::Reader pref;
    if (!pref.open(a.pos[2], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    info(a, "%s\nQuery database size: %u type: %s\nTarget database size: %u type: Aminoacid\n",
         swMode == 0 ? "Compute score only" : (swMode == 1 ? "Compute score and coverage" : "Compute score, coverage and sequence identity"),
         qdb->n, qdb->profile ? "Profile" : "Aminoacid", tdb->n);

    CtxH ctx;
    int rc = ctx.open(deviceOf(a));
    if (rc != SD_OK) return fail("no usable HIP device (sd_ctx_create returned " + std::to_string(rc) + "); this path has no CPU fallback");

    SeqSetH tset;
    const std::string tsetKey = a.pos[1] + "|" + std::to_string(deviceOf(a));
    if (resident().enabled && resident().seqSets.count(tsetKey)) {
        tset.s = resident().seqSets[tsetKey];   // the target sequences are on the device already
        tset.own = false;
    } else {
        rc = sd_seqset_create(ctx.c, tdb->residues.data(), tdb->offsets.data(), tdb->n, nullptr, &tset.s);
        if (rc != SD_OK) return failCtx(ctx.c, rc, "sd_seqset_create(targets)");
        if (resident().enabled) {
            resident().seqSets[tsetKey] = tset.s;
            tset.own = false;
        }
    }

    lap.mark("context + target sequences on the device");
    sddb::Writer out;
    int outType = sddb::withExtended(sddb::DBTYPE_ALIGNMENT_RES, sddb::extendedType(pref.dbtype()));
    if (!out.open(a.pos[3], outType, &err)) return fail(err);
    sd_alntext *text = nullptr;
    sd_alntext_create(&text);
    std::unique_ptr<sd_alntext, void (*)(sd_alntext *)> textGuard(text, sd_alntext_destroy);

    const uint64_t maxPairs = 4000000;
    const size_t nEntries = pref.size();
    uint64_t alignmentsNum = 0, passedNum = 0;
    AlignChunk C;
    std::vector<uint32_t> &localQ = C.localQ, &pq = C.pq, &pt = C.pt;
    std::vector<uint8_t> &ident = C.ident;
    // lines per prefilter entry (one pass over the DB on all threads): the chunks are cut from these, and a chunk's lines are then
    // parsed in parallel into their places
    std::vector<uint32_t> lineCount(nEntries, 0);
#pragma omp parallel for schedule(dynamic, 256)
    for (size_t e = 0; e < nEntries; e++) {
        uint32_t c = 0;
        for (const char *d = pref.data(e); *d != '\0';) {
            const char *nl = strchr(d, '\n');
            c++;
            if (!nl) break;
            d = nl + 1;
        }
        lineCount[e] = c;
    }
    lap.mark("count prefilter lines");
    std::vector<uint64_t> pairOff;
    for (size_t e0 = 0; e0 < nEntries;) {
        // chunk of entries bounded by pairs
        localQ.clear();
        size_t e1 = e0;
        std::vector<uint32_t> entryLocal;   // local query index of entry (UINT32_MAX: empty entry)
        pairOff.assign(1, 0);
        while (e1 < nEntries && (pairOff.back() < maxPairs || e1 == e0) && localQ.size() < 20000) {
            if (lineCount[e1] == 0) {
                entryLocal.push_back(UINT32_MAX);
                pairOff.push_back(pairOff.back());
                e1++;
                continue;
            }
            const uint32_t qKey = pref.key(e1);
            const size_t qId = qdb->rd.idOfKey(qKey);
            if (qId == SIZE_MAX)
                return fail("Query sequence " + std::to_string(qKey) + " is required in the prefiltering, but is not contained in the query sequence database.");
            entryLocal.push_back((uint32_t) localQ.size());
            localQ.push_back((uint32_t) qId);
            pairOff.push_back(pairOff.back() + lineCount[e1]);
            e1++;
        }
        pq.resize(pairOff.back());
        pt.resize(pairOff.back());
        ident.resize(pairOff.back());
        uint32_t missingKey = UINT32_MAX;
        bool missing = false;
#pragma omp parallel for schedule(dynamic, 64)
        for (size_t e = e0; e < e1; e++) {
            const uint32_t lq = entryLocal[e - e0];
            if (lq == UINT32_MAX) continue;
            const uint32_t qKey = pref.key(e);
            const float qL = (float) qdb->lens[localQ[lq]];
            uint64_t w = pairOff[e - e0];
            for (const char *d = pref.data(e); *d != '\0';) {
                const uint32_t tKey = (uint32_t) strtoul(d, nullptr, 10);
                while (*d != '\n' && *d != '\0') d++;
                if (*d == '\n') d++;
                const size_t tId = tdb->rd.idOfKey(tKey);
                if (tId == SIZE_MAX) {
#pragma omp critical(sd_align_missing)
                    {
                        missing = true;
                        missingKey = tKey;
                    }
                    break;
                }
                // Util::canBeCovered pre-check (Alignment.cpp:370-373): a rejected pair, never aligned
                const bool can = sd_host_can_be_covered(canCovThr, covMode, qL, (float) tdb->lens[tId]) != 0;
                pq[w] = lq;
                pt[w] = (uint32_t) tId;
                // 2 marks the pre-rejected pair: kept only so that --max-rejected counts it
                ident[w] = !can ? 2 : ((qKey == tKey && (includeIdentity || sameDb)) ? 1 : 0);
                w++;
            }
        }
        if (missing)
            return fail("Sequence " + std::to_string(missingKey) + " is required in the prefiltering, but is not contained in the target sequence database!");
        lap.mark("chunk: parse prefilter entries");
        const uint32_t nq = (uint32_t) localQ.size();
        const char *what = "";
        rc = alignChunkCore(ctx.c, host.h, S, *qdb, *tdb, tset.s, C, &lap, &what);
        if (rc != SD_OK) return failCtx(ctx.c, rc, what);
        alignmentsNum += C.aligned;
        passedNum += C.accepted;
        rc = sd_alntext_format(text, &S.crit, nq, C.outCounts->data(), C.outOrder->data(), C.outT->data(), C.outRecs->data(), C.outIdent->data(),
                               C.outPool->data(), C.qlen.data(), tdb->lens.data(), tdb->keys.data());
        if (rc != SD_OK) return fail("sd_alntext_format failed (" + std::to_string(rc) + ")");
        lap.mark("chunk: format");
        const char *txt;
        const uint64_t *eoff;
        sd_alntext_get(text, &txt, &eoff);
        for (size_t e = e0; e < e1; e++) {
            const uint32_t lq = entryLocal[e - e0];
            if (lq == UINT32_MAX) {
                if (!out.write(pref.key(e), "", 0)) return fail("cannot write " + a.pos[3]);
            } else if (!out.write(pref.key(e), txt + eoff[lq], (size_t) (eoff[lq + 1] - eoff[lq]))) {
                return fail("cannot write " + a.pos[3]);
            }
        }
        lap.mark("chunk: write");
        e0 = e1;
    }
    if (!out.close(&err)) return fail(err);
    lap.mark("close");
    info(a, "%llu alignments calculated\n%llu sequence pairs passed the thresholds\n", (unsigned long long) alignmentsNum,
         (unsigned long long) passedNum);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
int clusterhitsModule(const Args &a) {
    if (a.pos.size() != 4) return fail("usage: clusterhits <querySetDB> <targetSetDB> <matchesDB> <clustersDB> [options]");
    if (a.integer("--compressed", 0) != 0) return fail("--compressed 1 is not supported");
    if (a.flag("--cluster-use-weight", false)) return fail("--cluster-use-weight 1 is not supported");
    std::string err;
    Lap lap("clusterhits");
    const std::shared_ptr<const SetInfo> qsP = loadSetInfo(a.pos[0], false, &err);
    if (!qsP) return fail(err);
    const bool sameDb = a.pos[0] == a.pos[1];
    const std::shared_ptr<const SetInfo> tsP = sameDb ? qsP : loadSetInfo(a.pos[1], false, &err);
    if (!tsP) return fail(err);
    const SetInfo &qs = *qsP, &ts = *tsP;
    lap.mark("set info");
    sddb::Reader res, hdr;
    if (!res.open(a.pos[2], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    if (!hdr.open(a.pos[2] + "_h", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    if (hdr.size() != res.size()) return fail("matches and matches_h differ in size");
    const bool dbOut = a.flag("--db-output", false);

    sd_ch_params par;
    par.maxGeneGap = (uint32_t) a.integer("--max-gene-gap", 3);
    par.clusterSize = (uint32_t) a.integer("--cluster-size", 2);
    par.alpha = a.real("--alpha", 1.0);
    par.pCluThr = (float) a.real("--cluster-pval", 0.01);
    par.pMHThr = (float) a.real("--multihit-pval", 0.01);

    // entries -> flat arrays (R/src/util/ClusterHits.cpp:300-353)
    std::vector<uint64_t> hitOff(1, 0);
    std::vector<uint32_t> qPos, tPos, Nq, entryQSet, entryTSet;
    std::vector<uint8_t> strands;
    std::vector<double> pval;
    std::vector<std::pair<const char *, uint32_t> > lines;   // hit line (with its '\n')
    uint32_t maxOrf = 0;
    for (uint32_t s : qs.setSize) maxOrf = std::max(maxOrf, s);
    for (uint32_t s : ts.setSize) maxOrf = std::max(maxOrf, s);
    uint32_t maxPos = 0;
    // the matches parsed on all threads, each into lists of its own, then laid out back to back in entry order
    struct ParsedMatch {
        std::vector<std::pair<const char *, uint32_t> > lines;
        std::vector<uint32_t> qPos, tPos;
        std::vector<uint8_t> strands;
        std::vector<double> pval;
        unsigned long qSet = 0, tSet = 0, nq = 0;
        uint32_t maxPos = 0;
        std::string error;
    };
    std::vector<ParsedMatch> parsed;
    const size_t parseBlock = 8192;
    for (size_t b0 = 0; b0 < res.size(); b0 += parseBlock) {
        const size_t b1 = std::min(res.size(), b0 + parseBlock);
        parsed.resize(b1 - b0);
#pragma omp parallel for schedule(dynamic, 8)
        for (size_t i = b0; i < b1; i++) {
            ParsedMatch &pm = parsed[i - b0];
            pm.lines.clear();
            pm.qPos.clear();
            pm.tPos.clear();
            pm.strands.clear();
            pm.pval.clear();
            pm.error.clear();
            pm.maxPos = 0;
            const char *h = hdr.data(i);
            char *end;
            pm.qSet = strtoul(h, &end, 10);
            pm.tSet = strtoul(end, &end, 10);
            pm.nq = strtoul(end, &end, 10);
            // six tab separated columns are required (ClusterHits.cpp:303-307)
            int cols = 0;
            for (const char *c = h; *c && *c != '\n'; c++) cols += (*c == '\t');
            if (cols < 5) {
                pm.error = "Invalid header record";
                continue;
            }
            const char *d = res.data(i);
            while (*d != '\0') {
                const char *ls = d;
                char *e2;
                const unsigned long qid = strtoul(d, &e2, 10);
                const unsigned long tid = strtoul(e2, &e2, 10);
                const double p = strtod(e2, nullptr);
                while (*d != '\n' && *d != '\0') d++;
                if (*d == '\n') d++;
                if (qid >= qs.nameOfKey.size() || qs.nameOfKey[qid].empty()) {
                    pm.error = "Invalid query lookup record";
                    break;
                }
                if (tid >= ts.nameOfKey.size() || ts.nameOfKey[tid].empty()) {
                    pm.error = "Invalid target lookup record";
                    break;
                }
                pm.lines.push_back(std::make_pair(ls, (uint32_t) (d - ls)));
                pm.qPos.push_back(qs.posOfKey[qid]);
                pm.tPos.push_back(ts.posOfKey[tid]);
                pm.maxPos = std::max(pm.maxPos, std::max(qs.posOfKey[qid], ts.posOfKey[tid]));
                pm.strands.push_back((uint8_t) (qs.strandOfKey[qid] | (ts.strandOfKey[tid] << 1)));
                pm.pval.push_back(p);
            }
        }
        for (size_t i = b0; i < b1; i++) {
            const ParsedMatch &pm = parsed[i - b0];
            if (!pm.error.empty()) return fail(pm.error);
            const size_t K = pm.lines.size();
            if (K <= 1) continue;   // a single hit is no cluster (ClusterHits.cpp:359-361)
            lines.insert(lines.end(), pm.lines.begin(), pm.lines.end());
            qPos.insert(qPos.end(), pm.qPos.begin(), pm.qPos.end());
            tPos.insert(tPos.end(), pm.tPos.begin(), pm.tPos.end());
            strands.insert(strands.end(), pm.strands.begin(), pm.strands.end());
            pval.insert(pval.end(), pm.pval.begin(), pm.pval.end());
            maxPos = std::max(maxPos, pm.maxPos);
            hitOff.push_back(lines.size());
            Nq.push_back((uint32_t) pm.nq);
            entryQSet.push_back((uint32_t) pm.qSet);
            entryTSet.push_back((uint32_t) pm.tSet);
        }
    }
    parsed.clear();
    parsed.shrink_to_fit();
    lap.mark("parse matches");
    const uint32_t nPairs = (uint32_t) Nq.size();
    const uint64_t total = hitOff.back();
    std::vector<uint32_t> clusterOf(std::max<uint64_t>(total, 1), UINT32_MAX), rank(std::max<uint64_t>(total, 1), 0),
        nClusters(std::max<uint32_t>(nPairs, 1), 0), cSize(std::max<uint64_t>(total, 1), 0);
    std::vector<double> pCO(std::max<uint64_t>(total, 1), 0.0), pMH(std::max<uint64_t>(total, 1), 0.0);
    if (nPairs > 0) {
        CtxH ctx;
        int rc = sd_ctx_create(deviceOf(a), &ctx.c);
        if (rc != SD_OK) return fail("no usable HIP device (sd_ctx_create returned " + std::to_string(rc) + "); this path has no CPU fallback");
        const uint32_t lgN = std::max(maxOrf, maxPos) + 8;
        std::vector<double> lg(lgN);
        sd_host_lgamma_table(lg.data(), lgN);
        rc = sd_clusterhits_batch(ctx.c, &par, nPairs, hitOff.data(), qPos.data(), tPos.data(), strands.data(), pval.data(), Nq.data(),
                                  lg.data(), lgN, clusterOf.data(), rank.data(), nClusters.data(), pCO.data(), pMH.data(), cSize.data());
        if (rc != SD_OK) return failCtx(ctx.c, rc, "sd_clusterhits_batch");
    }
    lap.mark("device");
    sddb::Writer out, outH;
    if (!out.open(a.pos[3], dbOut ? res.dbtype() : (int) sddb::DBTYPE_OMIT_FILE, &err)) return fail(err);
    if (!outH.open(a.pos[3] + "_h", sddb::DBTYPE_GENERIC_DB, &err)) return fail(err);
    uint32_t key = 0;
    // the clusters of a block of set pairs formatted on all threads (members in their rank order), written in order
    struct PairText {
        std::string body, head;
        std::vector<uint32_t> bodyEnd, headEnd;   // per cluster
    };
    std::vector<PairText> texts;
    const uint32_t writeBlock = 4096;
    for (uint32_t e0 = 0; e0 < nPairs; e0 += writeBlock) {
        const uint32_t e1 = std::min(nPairs, e0 + writeBlock);
        texts.resize(e1 - e0);
#pragma omp parallel
        {
            std::vector<uint32_t> member, start;
            char co[32], mh[32];
#pragma omp for schedule(dynamic, 8)
            for (uint32_t e = e0; e < e1; e++) {
                PairText &pt = texts[e - e0];
                pt.body.clear();
                pt.head.clear();
                pt.bodyEnd.clear();
                pt.headEnd.clear();
                const uint64_t off = hitOff[e], end = hitOff[e + 1];
                const uint32_t nC = nClusters[e];
                start.assign((size_t) nC + 1, 0);
                for (uint32_t c = 0; c < nC; c++) start[c + 1] = start[c] + cSize[off + c];
                member.assign(start[nC], 0);
                for (uint64_t h = off; h < end; h++)
                    if (clusterOf[h] < nC) member[start[clusterOf[h]] + rank[h]] = (uint32_t) (h - off);
                for (uint32_t c = 0; c < nC; c++) {
                    for (uint32_t x = start[c]; x < start[c + 1]; x++) pt.body.append(lines[off + member[x]].first, lines[off + member[x]].second);
                    snprintf(co, sizeof(co), "%.3E", pCO[off + c]);
                    snprintf(mh, sizeof(mh), "%.3E", pMH[off + c]);
                    pt.head += std::to_string(entryQSet[e]) + "\t" + std::to_string(entryTSet[e]) + "\t" + co + "\t" + mh + "\t" +
                               std::to_string(cSize[off + c]) + "\n";
                    pt.bodyEnd.push_back((uint32_t) pt.body.size());
                    pt.headEnd.push_back((uint32_t) pt.head.size());
                }
            }
        }
        for (uint32_t e = e0; e < e1; e++) {
            const PairText &pt = texts[e - e0];
            uint32_t b0 = 0, h0 = 0;
            for (size_t c = 0; c < pt.bodyEnd.size(); c++) {
                if (!out.write(key, pt.body.data() + b0, pt.bodyEnd[c] - b0) || !outH.write(key, pt.head.data() + h0, pt.headEnd[c] - h0))
                    return fail("cannot write " + a.pos[3]);
                b0 = pt.bodyEnd[c];
                h0 = pt.headEnd[c];
                key++;
            }
        }
    }
    lap.mark("write clusters");
    if (!out.close(&err) || !outH.close(&err)) return fail(err);
    if (!dbOut) ::remove((a.pos[3] + ".index").c_str());
    lap.mark("close");
    info(a, "%u clusters from %u set pairs\n", key, nPairs);
    return 0;
}

}  // namespace sdcli
