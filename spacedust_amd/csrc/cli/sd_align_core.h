// The body of `align` for one chunk of queries without the DB text on either side: pairs in, accepted records in compareHits order
// out (Alignment::run, M/src/alignment/Alignment.cpp:340-470).  alignModule (sd_mod_hot.cpp) parses a prefilter DB into the pairs and
// formats the records; the in-memory iterative search (sd_mod_iter.cpp) hands the prefilter rows over as they are and keeps the records.
#ifndef SD_ALIGN_CORE_H
#define SD_ALIGN_CORE_H

#include "sd_cli.h"

#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace sdcli {

// backtrace pool of the align calls: raw bytes (a std::vector would zero a gigabyte on every growth)
struct BtPool {
    std::unique_ptr<char[]> p;
    size_t cap = 0;
    void reserve(size_t n) {
        if (cap >= n) return;
        p.reset(new char[n]);
        cap = n;
    }
    const char *data() const { return p.get(); }
};

// what Prefiltering's constructor derives from the command line (Prefiltering.cpp:180-215,1005-1065)
struct PrefSetup {
    int k = 6, kmerThr = 0, indexThr = 0;
    bool mask = true, includeIdentity = false, compBias = true;
    double maskProb = 0.9;
    sd_prefilter_params par;
};
int prefilterSetupFromArgs(const Args &a, sd_host *host, const SeqDb &tdb, bool profileQueries, PrefSetup &s);

// what Alignment's constructor derives from the command line (Alignment.cpp:31-57,170-192,296-303)
struct AlignSetup {
    sd_sw_params par, rpar;      // first pass; the realigner (score-biased matrix, E-value gate off)
    sd_aln_criteria crit;
    int swMode = 0, covMode = 0;
    bool realign = false, compBias = true, includeIdentity = false, stopRules = false;
    float realignScoreBias = -0.2f, canCovThr = 0.0f;
};
// fills s from the module's flags; returns 0, or the exit code of fail(...)
int alignSetupFromArgs(const Args &a, sd_host *host, uint64_t targetResidues, AlignSetup &s);

// scratch and results of one chunk (reused between chunks)
struct AlignChunk {
    // inputs the caller fills: the chunk's queries (ids in qdb) and its pairs in prefilter order; ident: 1 identity pair, 2 a pair
    // Util::canBeCovered rejects before any alignment (kept so that --max-rejected counts it), 0 otherwise
    std::vector<uint32_t> localQ, pq, pt;
    std::vector<uint8_t> ident;
    // results: records rec[order[x]] of local query q for x in its slice of `counts`; recT / recIdent / pool indexed like rec
    const std::vector<sd_sw_result> *outRecs = nullptr;
    const std::vector<uint32_t> *outOrder = nullptr, *outCounts = nullptr, *outT = nullptr;
    const std::vector<uint8_t> *outIdent = nullptr;
    const BtPool *outPool = nullptr;
    std::vector<int32_t> qlen;
    uint64_t aligned = 0, accepted = 0;
    // scratch
    std::vector<uint32_t> idxOut, order, counts, order2, counts2, accT, recQ, recT, pq2, pt2, apq, apt, aIdx, idx2;
    std::vector<uint8_t> ident2, recIdent, aid, qres;
    std::vector<uint64_t> qoff;
    std::vector<int8_t> qbias, qaln, qbias2;
    std::vector<sd_sw_result> res, res2, merged, full;
    BtPool pool, pool2;
};
// aligns the chunk's pairs and applies the criteria / sort / --realign pass.  lap (nullable): the module's SD_DEBUG_TIMING marks.
// Returns SD_OK or a C-ABI error code (sd_last_error(ctx) says why for device errors; *what names the failing call).
int alignChunkCore(sd_ctx *ctx, sd_host *host, const AlignSetup &s, const SeqDb &qdb, const SeqDb &tdb, sd_seqset *tset, AlignChunk &c, Lap *lap,
                   const char **what);

// sd_setdb over a loaded DB (+ set membership by id when a lookup exists)
struct SetDbArrays {
    std::vector<uint32_t> setId, pos;
    std::vector<uint8_t> strand;
    sd_setdb view;
    void fill(const SeqDb &db, const SetInfo *sets) {
        memset(&view, 0, sizeof(view));
        view.residues = db.residues.data();
        view.offsets = db.offsets.data();
        view.n = db.n;
        view.keys = db.keys.data();
        if (sets) {
            setId.resize(db.n);
            pos.resize(db.n);
            strand.resize(db.n);
            for (uint32_t i = 0; i < db.n; i++) {
                const uint32_t k = db.keys[i];
                setId[i] = k < sets->setOfKey.size() ? sets->setOfKey[k] : 0;
                pos[i] = k < sets->posOfKey.size() ? sets->posOfKey[k] : 0;
                strand[i] = k < sets->strandOfKey.size() ? sets->strandOfKey[k] : 0;
            }
            view.setId = setId.data();
            view.posInSet = pos.data();
            view.strand = strand.data();
            view.nSets = sets->nSets;
        }
        if (db.profile) {
            view.alnProfile = db.alnProfile.data();
            view.sortedScore = db.sortedScore.data();
            view.sortedIndex = db.sortedIndex.data();
        }
    }
};

inline void packNames(const std::vector<std::string> &v, std::string &blob, std::vector<uint64_t> &off) {
    off.assign(v.size() + 1, 0);
    for (size_t i = 0; i < v.size(); i++) off[i + 1] = off[i] + v[i].size();
    blob.clear();
    blob.reserve(off.back());
    for (const std::string &s : v) blob += s;
}

// result2profile's parameters (M/src/util/result2profile.cpp:16-60); par.qid points into this object
struct R2pSetup {
    sd_r2p_params par;
    std::string qid;
    double evalProfile = 0.001;
    R2pSetup() {}
    R2pSetup(const R2pSetup &) = delete;
    R2pSetup &operator=(const R2pSetup &) = delete;
};
int r2pSetupFromArgs(const Args &a, R2pSetup &s);

// `clustersearch --num-iterations N` with every hand-over between the modules in memory (sd_mod_iter.cpp)
int iterativeClusterSearchInMemory(const Args &a, const std::string &Q, const std::string &T, const std::string &tsvPath,
                                   const std::vector<std::string> &prefFlags, const std::vector<std::string> &alnFlags,
                                   const std::vector<std::string> &profFlags, const std::string &eUser, const std::string &eProfile);

}  // namespace sdcli
#endif
