// oracle/sd_oracle.cpp -- TEST INFRASTRUCTURE.  Plain scalar C++ restatement of the
// clustersearch hot path (prefilter -> Smith-Waterman -> clusterhits) of soedinglab/spacedust.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (spacedust_amd/) never does.
//
// Pinning: every function here is checked (tests/test_oracle_golden.py, tests/test_oracle_clusterhits_ref.py; golden vectors
// made from the real reference by tools/make_golden*.py)
//   * against the real reference classes compiled into oracle/_ref/libsdref.so, and
//   * against the reference's own known answers on examples/ (index entries 1 784 989, masked
//     residues 11 546, prefilter/alignment md5s recorded in SURVEY.md 8(c), run_regression.sh counts).
// Support code that is not on the hot path (matrix derivation, float composition bias, masking,
// index construction, E-values) is shared with the product's host library (spacedust_amd/csrc/host)
// and is pinned by the same comparisons.
//
// Citations: M/ = /root/reference/lib/mmseqs/, R/ = /root/reference/.
#include "sd_host.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace sd;

namespace {

struct OracleCtx {
    SubMat blosum2, ungapped2, seed8;   // SW (bias 0); diagonal scoring and seeds (scoreBias -0.2, Prefiltering.cpp:991)
    ExtMatrix two, three;
    bool haveExt;
    int threads;
};

struct OracleTarget {
    OracleCtx *ctx;
    TargetIndex idx;
    uint32_t nSeq;
};

// ------------------------------------------------------------------------------------------
// ungapped diagonal score (M/src/prefiltering/UngappedAlignment.cpp:30-43,416-430)
// prof: int8 [L][21]
// ------------------------------------------------------------------------------------------
// one real diagonal (computeSingelSequenceScores, UngappedAlignment.cpp:416-430)
int diagScoreReal(const int8_t *prof, int qL, const uint8_t *t, int tL, int diagonal, unsigned int minDist) {
    int maxv = 0, score = 0;
    if (diagonal >= 0 && minDist < (unsigned int) qL) {
        int n = std::min(tL, qL - (int) minDist);
        const int8_t *p = prof + (size_t) minDist * ALPH;
        for (int pos = 0; pos < n; pos++) {
            score += p[(size_t) pos * ALPH + t[pos]];
            score = score < 0 ? 0 : score;
            maxv = score > maxv ? score : maxv;
        }
    } else if (diagonal < 0 && minDist < (unsigned int) tL) {
        int n = std::min(tL - (int) minDist, qL);
        const uint8_t *tt = t + minDist;
        for (int pos = 0; pos < n; pos++) {
            score += prof[(size_t) pos * ALPH + tt[pos]];
            score = score < 0 ? 0 : score;
            maxv = score > maxv ? score : maxv;
        }
    }
    return maxv;
}

// scoreSingleSequence (UngappedAlignment.cpp:437-447): with 32 768 residues or more on either side the 16-bit diagonal
// is ambiguous and computeLongScore (:312-329) takes the best of every real diagonal it can stand for.
// NOT restated: the reference's batched route for DIAGONALBINSIZE hits on one diagonal (:235-300) looks up
// hits[hitIdx] where it means hits[seqs[hitIdx].id] (:290), so a target of >= 32 768 residues that falls into a full
// batch of eight is scored as another sequence of the batch (0 when that one is short).  Oracle and device score the
// sequence itself; tests/golden/long_vectors.npz keeps no diagonal with eight hits.
int diagScore(const int8_t *prof, int qL, const uint8_t *t, int tL, uint16_t diag16) {
    if (qL >= 32768 || tL >= 32768) {
        int total = 0;
        for (unsigned int div = 1; div <= 1 + (unsigned int) tL / 32768; div++) {
            const int real = (int) diag16 - (int) div * 65536;
            total = std::max(total, diagScoreReal(prof, qL, t, tL, real, (unsigned int) std::abs(real)));
        }
        for (unsigned int div = 0; div <= (unsigned int) qL / 65536; div++) {
            const int real = (int) diag16 + (int) div * 65536;
            total = std::max(total, diagScoreReal(prof, qL, t, tL, real, (unsigned int) std::abs(real)));
        }
        return total;
    }
    const int d = (int16_t) diag16;
    const unsigned short minDist = (unsigned short) std::min((unsigned short) (0 - diag16), (unsigned short) (diag16 - 0));
    return diagScoreReal(prof, qL, t, tL, d, minDist);
}

struct Cand {
    uint32_t id;
    uint16_t diag;
    uint8_t count;
};

}  // namespace

extern "C" {

void *or_ctx_create(int threads) {
    OracleCtx *c = new OracleCtx();
    initSubMat(c->blosum2, MAT_BLOSUM62, 2.0f, 0.0f);
    initSubMat(c->ungapped2, MAT_BLOSUM62, 2.0f, -0.2f);
    initSubMat(c->seed8, MAT_VTML80, 8.0f, -0.2f);
    c->haveExt = false;
    c->threads = threads;
    return c;
}

static void ensureExt(OracleCtx *c) {
    if (c->haveExt) return;
    buildExtMatrix(c->seed8, 2, c->two, c->threads);
    buildExtMatrix(c->seed8, 3, c->three, c->threads);
    c->haveExt = true;
}

int or_get_matrix(void *vc, int which, short *out, double *pback, unsigned char *aa2num) {
    OracleCtx *c = (OracleCtx *) vc;
    const SubMat &m = which == 0 ? c->blosum2 : (which == 1 ? c->seed8 : c->ungapped2);
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++) out[i * ALPH + j] = m.sub[i][j];
    for (int i = 0; i < ALPH; i++) pback[i] = m.pBack[i];
    memcpy(aa2num, m.aa2num, 256);
    return ALPH;
}

void or_map_sequence(void *vc, const char *seq, size_t len, unsigned char *out) {
    mapSequence(((OracleCtx *) vc)->seed8, seq, len, out);
}

void or_compbias(void *vc, int which, const unsigned char *num, int L, float scale, float *out) {
    OracleCtx *c = (OracleCtx *) vc;
    calcLocalAaBiasCorrection(which == 0 ? c->blosum2 : (which == 1 ? c->seed8 : c->ungapped2), num, L, out, scale);
}

size_t or_ext_matrix(void *vc, int which, short *score, unsigned short *index) {
    OracleCtx *c = (OracleCtx *) vc;
    ensureExt(c);
    ExtMatrix &m = which == 3 ? c->three : c->two;
    if (score != NULL) {
        memcpy(score, m.score.data(), m.score.size() * sizeof(short));
        memcpy(index, m.index.data(), m.index.size() * sizeof(unsigned short));
    }
    return m.size;
}

size_t or_kmer_list(void *vc, int k, const unsigned char *window, int thr, unsigned int *out, size_t cap) {
    OracleCtx *c = (OracleCtx *) vc;
    ensureExt(c);
    std::vector<uint32_t> v;
    generateKmerList(c->three, c->two, k, window, thr, v);
    size_t n = std::min(cap, v.size());
    memcpy(out, v.data(), n * sizeof(uint32_t));
    return v.size();
}

int or_mask(void *vc, unsigned char *num, int L, double maskProb) {
    OracleCtx *c = (OracleCtx *) vc;
    MaskCtx m;
    initMaskCtx(c->seed8, m);
    return tantanMask(m, num, L, maskProb);
}

void *or_target_create(void *vc, const unsigned char *seqs, const uint64_t *offsets, uint32_t nSeq, int k,
                       int kmerThr, int mask, double maskProb) {
    OracleCtx *c = (OracleCtx *) vc;
    OracleTarget *t = new OracleTarget();
    t->ctx = c;
    t->nSeq = nSeq;
    buildTargetIndex(c->seed8, seqs, offsets, nSeq, k, kmerThr, mask != 0, maskProb, c->threads, t->idx);
    return t;
}

void or_target_destroy(void *vt) { delete (OracleTarget *) vt; }

uint64_t or_target_info(void *vt, uint64_t *tableSize, uint64_t *masked) {
    OracleTarget *t = (OracleTarget *) vt;
    *tableSize = t->idx.tableSize;
    *masked = t->idx.maskedResidues;
    return t->idx.entrySeq.size();
}

void or_target_dump(void *vt, uint32_t *offsets, uint32_t *entrySeq, uint16_t *entryPos, unsigned char *masked) {
    OracleTarget *t = (OracleTarget *) vt;
    memcpy(offsets, t->idx.offsets.data(), t->idx.offsets.size() * sizeof(uint32_t));
    memcpy(entrySeq, t->idx.entrySeq.data(), t->idx.entrySeq.size() * sizeof(uint32_t));
    memcpy(entryPos, t->idx.entryPos.data(), t->idx.entryPos.size() * sizeof(uint16_t));
    if (masked != NULL) memcpy(masked, t->idx.masked.data(), t->idx.masked.size());
}

// ------------------------------------------------------------------------------------------
// Prefilter, one query (QueryMatcher::matchQuery, M/src/prefiltering/QueryMatcher.cpp:85-211;
// step numbers refer to SURVEY.md Appendix A.1).
// Returns the number of result hits; outputs are in the reference's final order.
// stats: [0] #similar k-mers, [1] #index entries matched, [2] #candidates scored, [3] sum of diagonal lengths
// ------------------------------------------------------------------------------------------
struct ProfileQuery {            // Sequence::mapProfile output for one profile query
    const int8_t *aln;           // [qL][21]
    const int16_t *sortedScore;  // [qL][20]
    const uint8_t *sortedIndex;  // [qL][20]
};

static int64_t prefilterQueryImpl(void *vt, const unsigned char *q, int qL, uint32_t identityId, int kmerThr,
                                  uint32_t maxHits, int minDiagScore, uint32_t binSize, int compBias,
                                  uint32_t *outId, int32_t *outScore, uint16_t *outDiag, uint64_t *stats,
                                  const ProfileQuery *pq) {
    OracleTarget *T = (OracleTarget *) vt;
    OracleCtx *c = T->ctx;
    ensureExt(c);
    const TargetIndex &ix = T->idx;
    const int k = ix.k, span = ix.span;
    const uint32_t dbSize = T->nSeq;
    maxHits = std::min(maxHits, dbSize);

    // steps 1-2: composition bias (seed matrix) and diagonal profile
    std::vector<float> cb(qL > 0 ? qL : 1, 0.0f);
    if (compBias && !pq) calcLocalAaBiasCorrection(c->seed8, q, qL, cb.data(), 1.0f);   // profiles: no correction (QueryMatcher.cpp:93-99)
    std::vector<int8_t> prof((size_t) (qL > 0 ? qL : 1) * ALPH);
    if (pq) memcpy(prof.data(), pq->aln, (size_t) qL * ALPH);   // UngappedAlignment::createProfile, profile branch (:398-405)
    for (int pos = 0; pos < qL && !pq; pos++) {
        float a = cb[pos];
        float r = (float) ((a < 0.0) ? (double) (a / 4) - 0.5 : (double) (a / 4) + 0.5);
        int8_t corr = (int8_t) (char) r;
        for (int aa = 0; aa < ALPH; aa++)
            prof[(size_t) pos * ALPH + aa] = (int8_t) (c->ungapped2.sub[q[pos]][aa] + corr);
    }

    // steps 3-5: similar k-mers per position, index lists appended in stream order
    std::vector<uint32_t> hitSeq;
    std::vector<uint16_t> hitDiag;
    std::vector<uint32_t> kmers;
    uint64_t nKmers = 0;
    uint8_t window[8];
    // the hit buffer holds maxDbMatches entries; a list that would fill it closes the current chunk first
    // (QueryMatcher.cpp:281-316): chunkStart[] = stream positions where a new chunk begins
    const uint64_t maxDbMatches = std::max<uint64_t>(1000000, dbSize) * 2;
    std::vector<size_t> chunkStart(1, 0);
    uint64_t inBuffer = 0;
    for (int i = 0; i + span <= qL; i++) {
        bool hasX = false;
        float bias = 0;
        for (int p = 0; p < k; p++) {
            window[p] = q[i + ix.seedPos[p]];
            hasX |= (window[p] == X_CODE);
            bias += cb[i + ix.seedPos[p]];
        }
        if (hasX) continue;
        short b = (short) ((bias < 0.0) ? (double) bias - 0.5 : (double) bias + 0.5);
        short thr = (short) std::max(kmerThr - b, 0);
        if (pq) {
            const int16_t *rowScore[8];
            const uint8_t *rowIndex[8];
            for (int p = 0; p < k; p++) {   // Sequence::nextProfileKmer (:294-305)
                rowScore[p] = pq->sortedScore + (size_t) (i + ix.seedPos[p]) * 20;
                rowIndex[p] = pq->sortedIndex + (size_t) (i + ix.seedPos[p]) * 20;
            }
            generateProfileKmerList(rowScore, rowIndex, k, thr, kmers);
        } else {
            generateKmerList(c->three, c->two, k, window, thr, kmers);
        }
        nKmers += kmers.size();
        for (size_t z = 0; z < kmers.size(); z++) {
            uint32_t a = ix.offsets[kmers[z]], e = ix.offsets[kmers[z] + 1];
            if (inBuffer + (e - a) >= maxDbMatches) {
                chunkStart.push_back(hitSeq.size());
                inBuffer = 0;
            }
            inBuffer += e - a;
            for (uint32_t x = a; x < e; x++) {
                hitSeq.push_back(ix.entrySeq[x]);
                hitDiag.push_back((uint16_t) (i - ix.entryPos[x]));
            }
        }
    }
    // two overflows in a row take the merge + score + keep-max route of :289-303, which is not restated
    if (chunkStart.size() > 2) return -2;
    chunkStart.push_back(hitSeq.size());

    // step 6: double-diagonal detection per chunk, bins ascending, stream order inside a bin
    // (CacheFriendlyOperations.cpp:38-48,185-272,337-347)
    const uint32_t mask = binSize - 1;
    std::vector<uint8_t> prev8(dbSize, 0), mark(dbSize, 0);
    std::vector<Cand> cands;
    std::vector<Cand> tmp;
    for (size_t ch = 0; ch + 1 < chunkStart.size(); ch++) {
        std::vector<std::vector<uint32_t> > binIdx(binSize);
        for (size_t h = chunkStart[ch]; h < chunkStart[ch + 1]; h++) binIdx[hitSeq[h] & mask].push_back((uint32_t) h);
        std::fill(prev8.begin(), prev8.end(), 0);
        for (uint32_t b = 0; b < binSize; b++) {
            tmp.clear();
            const std::vector<uint32_t> &v = binIdx[b];
            for (size_t n = 0; n < v.size(); n++) {
                const uint32_t id = hitSeq[v[n]];
                const uint8_t cur = (uint8_t) hitDiag[v[n]];
                if (cur == prev8[id]) {
                    Cand cd;
                    cd.id = id;
                    cd.diag = hitDiag[v[n]];
                    cd.count = 0;
                    tmp.push_back(cd);
                }
                prev8[id] = cur;
            }
            for (size_t n = tmp.size(); n-- > 0;) mark[tmp[n].id] = (uint8_t) ((uint8_t) tmp[n].diag + 1);
            for (size_t n = 0; n < tmp.size(); n++) {
                if (mark[tmp[n].id] != (uint8_t) tmp[n].diag) cands.push_back(tmp[n]);
                mark[tmp[n].id] = (uint8_t) tmp[n].diag;
            }
        }
    }
    if (chunkStart.size() > 2) {
        // one overflow: the two result lists are merged with mergeElementsByDiagonal (:323-326 ->
        // CacheFriendlyOperations.cpp:84-115): re-binned in arrival order, consecutive equal 8-bit diagonals of a
        // target collapse to the first
        std::vector<Cand> merged;
        for (uint32_t b = 0; b < binSize; b++) {
            tmp.clear();
            for (size_t n = 0; n < cands.size(); n++)
                if ((cands[n].id & mask) == b) tmp.push_back(cands[n]);
            for (size_t n = tmp.size(); n-- > 0;) mark[tmp[n].id] = (uint8_t) ((uint8_t) tmp[n].diag + 1);
            for (size_t n = 0; n < tmp.size(); n++) {
                if (mark[tmp[n].id] != (uint8_t) tmp[n].diag) merged.push_back(tmp[n]);
                mark[tmp[n].id] = (uint8_t) tmp[n].diag;
            }
        }
        cands.swap(merged);
    }
    const uint64_t foundDiagonalsSize = std::max<uint64_t>(1000000, dbSize);
    if (cands.size() >= foundDiagonalsSize / 2) return -3;   // unsorted branch not restated

    // step 7: ungapped diagonal scores on the masked target sequences
    uint64_t diagLenSum = 0;
    for (size_t n = 0; n < cands.size(); n++) {
        const uint8_t *t = ix.masked.data() + ix.seqOffsets[cands[n].id];
        int tL = (int) (ix.seqOffsets[cands[n].id + 1] - ix.seqOffsets[cands[n].id]);
        int s = diagScore(prof.data(), qL, t, tL, cands[n].diag);
        cands[n].count = (uint8_t) std::min(255, s);
        int d = (int16_t) cands[n].diag;
        diagLenSum += d >= 0 ? std::max(0, std::min(tL, qL - d)) : std::max(0, std::min(tL + d, qL));
    }

    // step 8: keep the per-target maximum (CacheFriendlyOperations.cpp:350-380); the list is already
    // bin-major and re-binning is stable.
    std::vector<uint8_t> best(dbSize, 0);
    std::vector<Cand> kept;
    {
        size_t start = 0;
        while (start < cands.size()) {
            const uint32_t b = cands[start].id & mask;
            size_t end = start;
            while (end < cands.size() && (cands[end].id & mask) == b) end++;
            for (size_t n = start; n < end; n++) best[cands[n].id] = std::max(best[cands[n].id], cands[n].count);
            for (size_t n = start; n < end; n++) {
                bool found = best[cands[n].id] == cands[n].count;
                if (found) {
                    kept.push_back(cands[n]);
                    best[cands[n].id] = 0;
                }
            }
            start = end;
        }
    }

    // step 9: score histogram, cut, stable descending counting sort (QueryMatcher.h:206-216, .cpp:498-523)
    unsigned int scoreSizes[256];
    memset(scoreSizes, 0, sizeof(scoreSizes));
    for (size_t n = 0; n < kept.size(); n++) scoreSizes[kept[n].count]++;
    size_t foundHits = 0;
    unsigned int diagonalThr = 0;
    for (diagonalThr = 255; diagonalThr > 0; diagonalThr--) {
        foundHits += scoreSizes[diagonalThr];
        if (foundHits >= maxHits) break;
    }
    diagonalThr = std::max((unsigned int) minDiagScore, diagonalThr);
    std::vector<Cand> sorted;
    auto radix = [&](const std::vector<Cand> &in, unsigned int thr, std::vector<Cand> &out) {
        out.clear();
        for (int s = 255; s >= (int) thr; s--)
            for (size_t n = 0; n < in.size(); n++)
                if (in[n].count == s) out.push_back(in[n]);
    };
    radix(kept, diagonalThr, sorted);
    const unsigned int maxDiagonalScoreThr = 255;   // UCHAR_MAX - getQueryBias() (= 0)
    int rescale = 0;
    unsigned short resThr = (unsigned short) diagonalThr;
    if (diagonalThr >= maxDiagonalScoreThr) {
        // rescoreHits (QueryMatcher.cpp:525-544)
        int maxSelf = diagScore(prof.data(), qL, q, qL, 0);
        maxSelf = maxSelf - (int) maxDiagonalScoreThr;
        maxSelf = std::max(1, maxSelf);
        maxSelf = std::min(maxSelf, (int) USHRT_MAX);
        float fltMaxSelf = (float) maxSelf;
        std::vector<Cand> resc;
        for (size_t n = 0; n < sorted.size() && sorted[n].count >= maxDiagonalScoreThr; n++) {
            const uint8_t *t = ix.masked.data() + ix.seqOffsets[sorted[n].id];
            int tL = (int) (ix.seqOffsets[sorted[n].id + 1] - ix.seqOffsets[sorted[n].id]);
            unsigned int ns = (unsigned int) diagScore(prof.data(), qL, t, tL, sorted[n].diag);
            ns -= maxDiagonalScoreThr;
            float sc = (float) std::min(ns, (unsigned int) USHRT_MAX);
            Cand cd = sorted[n];
            cd.count = (unsigned char) ((sc / fltMaxSelf) * (float) UCHAR_MAX + 0.5);
            resc.push_back(cd);
        }
        radix(resc, 0, sorted);
        rescale = maxSelf;
        resThr = 0;
    }

    // step 10: result list (QueryMatcher.cpp:364-420) and final order (QueryMatcher.h:38-48)
    struct Hit {
        uint32_t id;
        int score;
        uint16_t diag;
    };
    std::vector<Hit> res;
    if (identityId != UINT_MAX) {
        Hit h;
        h.id = identityId;
        h.score = USHRT_MAX;
        h.diag = 0;
        res.push_back(h);
    }
    for (size_t n = 0; n < sorted.size() && res.size() < maxHits; n++) {
        if (sorted[n].count >= resThr && sorted[n].id != identityId) {
            Hit h;
            h.id = sorted[n].id;
            h.score = sorted[n].count;
            h.diag = sorted[n].diag;
            if (rescale != 0) {
                h.score = (int) (255 + ((unsigned int) sorted[n].count * (unsigned int) rescale / 255));
            } else if (sorted[n].count >= 255) {
                const uint8_t *t = ix.masked.data() + ix.seqOffsets[h.id];
                int tL = (int) (ix.seqOffsets[h.id + 1] - ix.seqOffsets[h.id]);
                h.score = diagScore(prof.data(), qL, t, tL, h.diag);
            }
            res.push_back(h);
        }
    }
    auto cmp = [](const Hit &a, const Hit &b) {
        if (abs(a.score) > abs(b.score)) return true;
        if (abs(b.score) > abs(a.score)) return false;
        return a.id < b.id;
    };
    if (res.size() > 1) {
        if (identityId != UINT_MAX) std::sort(res.begin() + 1, res.end(), cmp);
        else std::sort(res.begin(), res.end(), cmp);
    }
    for (size_t n = 0; n < res.size(); n++) {
        outId[n] = res[n].id;
        outScore[n] = res[n].score;
        outDiag[n] = res[n].diag;
    }
    if (stats != NULL) {
        stats[0] = nKmers;
        stats[1] = hitSeq.size();
        stats[2] = cands.size();
        stats[3] = diagLenSum;
    }
    return (int64_t) res.size();
}

int64_t or_prefilter_query(void *vt, const unsigned char *q, int qL, uint32_t identityId, int kmerThr,
                           uint32_t maxHits, int minDiagScore, uint32_t binSize, int compBias,
                           uint32_t *outId, int32_t *outScore, uint16_t *outDiag, uint64_t *stats) {
    return prefilterQueryImpl(vt, q, qL, identityId, kmerThr, maxHits, minDiagScore, binSize, compBias, outId, outScore,
                              outDiag, stats, nullptr);
}

// profile query (a22): letters / alignment profile / sorted k-mer generator rows as sd::mapProfile produces them
int64_t or_prefilter_query_profile(void *vt, const unsigned char *letters, const signed char *aln,
                                   const int16_t *sortedScore, const unsigned char *sortedIndex, int qL, uint32_t identityId,
                                   int kmerThr, uint32_t maxHits, int minDiagScore, uint32_t binSize, uint32_t *outId,
                                   int32_t *outScore, uint16_t *outDiag, uint64_t *stats) {
    ProfileQuery pq = {(const int8_t *) aln, sortedScore, sortedIndex};
    return prefilterQueryImpl(vt, letters, qL, identityId, kmerThr, maxHits, minDiagScore, binSize, 0, outId, outScore,
                              outDiag, stats, &pq);
}

// Sequence::mapProfile restated in the product's host code (sd::mapProfile); exposed for the golden tests
void or_map_profile(const char *data, uint32_t L, unsigned char *letters, unsigned char *consensus, signed char *aln,
                    int16_t *sortedScore, unsigned char *sortedIndex) {
    mapProfile(data, L, letters, consensus, (int8_t *) aln, sortedScore, sortedIndex);
}
size_t or_profile_kmer_list(const int16_t *sortedScore /* [k][20] */, const unsigned char *sortedIndex, int k, int thr,
                            unsigned int *out, size_t cap) {
    const int16_t *sc[8];
    const uint8_t *ix[8];
    for (int p = 0; p < k; p++) {
        sc[p] = sortedScore + (size_t) p * 20;
        ix[p] = sortedIndex + (size_t) p * 20;
    }
    std::vector<uint32_t> v;
    generateProfileKmerList(sc, ix, k, thr, v);
    for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
    return v.size();
}

// single diagonal score, for kernel unit tests
int or_diag_score(const int8_t *prof, int qL, const unsigned char *t, int tL, uint16_t diag) {
    return diagScore(prof, qL, t, tL, diag);
}

// ------------------------------------------------------------------------------------------
// Smith-Waterman score pass (sw_sse2_byte / sw_sse2_word, M/src/alignment/StripedSmithWaterman.cpp:639-940,
// 943-1214) in scalar form.  `lanes` = 32 reproduces the AVX2 byte kernel, 16 the word kernel:
// the striped kernels update E from the H that only contains the F of the *same SIMD lane segment*
// (segLen = ceil(n/lanes) consecutive query rows), then complete H with the lazy-F loop
// ("disallow adjacent insertion and then deletion", :819,1119).  Derivation in DESIGN.md.
//   prof: int16 [21][n] linear profile (mat[t][q_j] + compositionBias[j]) of the n query rows used
//   dir 0: target columns 0..tL-1 ;  dir 1: tL-1 down to 0 (reverse pass)
//   terminate: stop after the first column whose maximum equals it (0 = never)
//   overflowAt: byte-mode abort when a new maximum m satisfies m + bias >= 255 (0 = off)
// out[0] score (255 on byte overflow) out[1] end_db out[2] end_query
// ------------------------------------------------------------------------------------------
static void swPass(const int16_t *prof, int n, const uint8_t *t, int tL, int lanes, int dir, int go, int ge,
                   int terminate, int biasForOverflow, bool byteMode, int *out) {
    const int segLen = (n + lanes - 1) / lanes;
    std::vector<int> H(n, 0), Hn(n, 0), E(n, 0), Hmax(n, 0);
    int maxv = 0;
    int end_db = byteMode ? -1 : 0;
    bool overflow = false;
    int begin = 0, end = tL, step = 1;
    if (dir == 1) {
        begin = tL - 1;
        end = -1;
        step = -1;
    }
    for (int i = begin; i != end; i += step) {
        const int16_t *p = prof + (size_t) t[i] * n;
        int Fl = 0, Ff = 0, colMax = 0;
        for (int q = 0; q < n; q++) {
            if (q % segLen == 0) Fl = 0;
            int diag = q > 0 ? H[q - 1] : 0;
            int h = diag + p[q];
            if (h < 0) h = 0;
            if (!byteMode && h > 32767) h = 32767;   // simdi16_adds: the word kernel's H saturates (:1069)
            int hpre = std::max(std::max(h, E[q]), Fl);
            int g = std::max(hpre, Ff);
            Hn[q] = g;
            colMax = std::max(colMax, g);
            int open = std::max(hpre - go, 0);
            E[q] = std::max(std::max(E[q] - ge, 0), open);
            Fl = std::max(std::max(Fl - ge, 0), open);
            Ff = std::max(std::max(Ff - ge, 0), std::max(g - go, 0));
        }
        H.swap(Hn);
        if (colMax > maxv) {
            maxv = colMax;
            if (byteMode && maxv + biasForOverflow >= 255) {
                overflow = true;
                break;
            }
            end_db = i;
            Hmax = H;
        }
        if (terminate != 0 && colMax == terminate) break;
    }
    int end_query = n - 1;
    for (int q = 0; q < n; q++) {
        if (Hmax[q] == maxv) {
            end_query = std::min(end_query, q);
            break;
        }
    }
    out[0] = overflow ? 255 : maxv;
    out[1] = end_db;
    out[2] = end_query;
}

int or_sw_pass(const int16_t *prof, int n, const unsigned char *t, int tL, int lanes, int dir, int go, int ge,
               int terminate, int bias, int byteMode, int *out) {
    swPass(prof, n, t, tL, lanes, dir, go, ge, terminate, bias, byteMode != 0, out);
    return 0;
}

// ------------------------------------------------------------------------------------------
// banded traceback (SmithWaterman::banded_sw, StripedSmithWaterman.cpp:1348-1600) on the sub-rectangle
// q[0..qLen) x t[0..tLen); sc(i,j) = mat[q_i][t_j] + cb[i].  Emits the expanded backtrace string
// (computerBacktrace, :548-581).  Returns its length, or -1 on a traceback error.
// The band arrays and the direction matrix are indexed exactly as in the reference (set_u/set_d),
// including what stale band-edge cells hold.
// ------------------------------------------------------------------------------------------
static int bandedTraceback(const SubMat &m, const uint8_t *q, const int8_t *cb, int qLen, const uint8_t *t, int tLen,
                           int score, int go, int ge, std::string &bt, const int8_t *aln = nullptr /* profile query: [qLen][21] */) {
    int band = abs(tLen - qLen) + 1;
    std::vector<int> h_b, e_b, h_c;
    std::vector<int8_t> direction;
    int64_t width, width_d;
    int maxv = 0;
    do {
        width = (int64_t) band * 2 + 3;
        width_d = (int64_t) band * 2 + 1;
        if ((int64_t) h_b.size() < width + 1) {
            h_b.resize(width + 1, 0);
            e_b.resize(width + 1, 0);
            h_c.resize(width + 1, 0);
        }
        if ((int64_t) direction.size() < width_d * qLen * 3 + 1) direction.resize(width_d * qLen * 3 + 1, 0);
        for (int64_t j = 1; j < width - 1; j++) h_b[j] = 0;
        for (int i = 0; i < qLen; i++) {
            int beg = 0, end = tLen - 1, u = 0;
            int j = i - band;
            beg = beg > j ? beg : j;
            j = i + band;
            end = end < j ? end : j;
            int64_t edge = end + 1 < width - 1 ? end + 1 : width - 1;
            int f = h_b[0] = e_b[0] = h_b[edge] = e_b[edge] = h_c[0] = 0;
            int8_t *dl = direction.data() + width_d * i * 3;
            for (j = beg; j <= end; j++) {
                auto setU = [&](int ii, int jj) { int x = ii - band; x = x > 0 ? x : 0; return jj - x + 1; };
                auto setD = [&](int ii, int jj, int p) { int x = ii - band; x = x > 0 ? x : 0; x = jj - x; return x * 3 + p; };
                u = setU(i, j);
                int e = setU(i - 1, j);
                int b = setU(i, j - 1);
                int d = setU(i - 1, j - 1);
                int de = setD(i, j, 0), df = setD(i, j, 1), dh = setD(i, j, 2);
                int temp1 = i == 0 ? -go : h_b[e] - go;
                int temp2 = i == 0 ? -ge : e_b[e] - ge;
                e_b[u] = temp1 > temp2 ? temp1 : temp2;
                dl[de] = temp1 > temp2 ? 3 : 2;
                temp1 = h_c[b] - go;
                temp2 = f - ge;
                f = temp1 > temp2 ? temp1 : temp2;
                dl[df] = temp1 > temp2 ? 5 : 4;
                int f1 = f > 0 ? f : 0;
                int e1 = e_b[u] > 0 ? e_b[u] : 0;
                temp1 = e1 > f1 ? e1 : f1;
                temp2 = h_b[d] + (aln ? (int) aln[(size_t) i * ALPH + t[j]] : m.sub[q[i]][t[j]] + cb[i]);   // banded_sw's profile branch (:1472-1474)
                h_c[u] = temp1 > temp2 ? temp1 : temp2;
                if (h_c[u] > maxv) maxv = h_c[u];
                if (temp1 <= temp2) dl[dh] = 1;
                else dl[dh] = e1 > f1 ? dl[de] : dl[df];
            }
            for (j = 1; j <= u; j++) h_b[j] = h_c[j];
        }
        band *= 2;
    } while (maxv < score);
    band /= 2;

    // traceback
    int i = qLen - 1, j = tLen - 1;
    int state = 2;
    std::string rev;
    char op = 'M';
    const int8_t *dl = direction.data() + width_d * (qLen - 1) * 3;
    while (i > 0 || j > 0) {
        int x = i - band;
        x = x > 0 ? x : 0;
        x = j - x;
        int idx = x * 3 + state;
        switch (dl[idx]) {
            case 1: --i; --j; state = 2; dl -= width_d * 3; op = 'M'; break;
            case 2: --i; state = 0; dl -= width_d * 3; op = 'I'; break;
            case 3: --i; state = 2; dl -= width_d * 3; op = 'I'; break;
            case 4: --j; state = 1; op = 'D'; break;
            case 5: --j; state = 2; op = 'D'; break;
            default: return -1;
        }
        rev.push_back(op);
    }
    // the reference closes the CIGAR with the first cell: run of `op` is extended by one if it is 'M',
    // otherwise a single 'M' is added (:1559-1576)
    rev.push_back('M');
    bt.assign(rev.rbegin(), rev.rend());
    return (int) bt.size();
}

int or_banded_traceback(void *vc, const unsigned char *q, const int8_t *cb, int qLen, const unsigned char *t, int tLen,
                        int score, int go, int ge, char *out, int cap) {
    std::string bt;
    int n = bandedTraceback(((OracleCtx *) vc)->blosum2, q, cb, qLen, t, tLen, score, go, ge, bt);
    if (n < 0) return n;
    int c = std::min(cap - 1, n);
    memcpy(out, bt.data(), c);
    out[c] = 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// Full pair alignment (ssw_align_private<SEQ_SEQ>, :310-545 + Matcher::getSWResult, Matcher.cpp:60-142)
// out[0] score [1] qStart [2] qEnd [3] tStart [4] tEnd [5] identical [6] backtrace length [7] word-mode flag
// returns evalue.  swMode as in the reference (0 score, 1 +coverage/start, 2 +backtrace).
// ------------------------------------------------------------------------------------------
static double swAlignImpl(void *vc, const unsigned char *q, int qL, const unsigned char *t, int tL, uint64_t dbResidues,
                          int swMode, double evalThr, int covMode, float covThr, int compBias, int isIdentity, int *out,
                          char *backtrace, int btCap, const int8_t *aln /* profile query: [qL][21], else NULL */) {
    OracleCtx *c = (OracleCtx *) vc;
    const SubMat &m = c->blosum2;
    const int go = 11, ge = 1;
    Evaluer ev;
    initEvaluer(ev, dbResidues);
    std::vector<int8_t> cb8(qL > 0 ? qL : 1, 0);
    if (compBias) swCompBias8(m, q, qL, cb8.data());
    int minCb = 0;
    for (int i = 0; i < qL; i++) minCb = std::min(minCb, (int) cb8[i]);
    int matMin = 0;
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++) matMin = std::min(matMin, (int) m.sub[i][j]);
    int bias = abs(matMin) + abs(minCb);
    if (aln) {   // ssw_init's PROFILE branch (:1271-1281): min over the L x 20 profile scores
        int pm = 0;
        for (int i = 0; i < qL; i++)
            for (int a = 0; a < 20; a++) pm = std::min(pm, (int) aln[(size_t) i * ALPH + a]);
        bias = abs(pm);
    }
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[1] = -1;
    out[3] = -1;
    std::string bt;
    if (backtrace != NULL && btCap > 0) backtrace[0] = 0;

    if (isIdentity) {
        // scoreIdentical (:1675-1710)
        short score = 0;
        for (int pos = 0; pos < tL; pos++)
            score += aln ? (short) aln[(size_t) pos * ALPH + t[pos]] : (short) (m.sub[t[pos]][q[pos]] + cb8[pos]);
        out[0] = (uint32_t) (int) score;
        out[1] = swMode == 0 ? -1 : 0;
        out[3] = swMode == 0 ? -1 : 0;
        out[2] = tL - 1;
        out[4] = tL - 1;
        out[5] = tL;
        out[6] = tL;
        if (backtrace != NULL) {
            int cc = std::min(btCap - 1, tL);
            memset(backtrace, 'M', cc);
            backtrace[cc] = 0;
        }
        return computeEvalue(ev, (double) (uint32_t) (int) score, qL);
    }

    // forward profile
    std::vector<int16_t> prof((size_t) ALPH * qL);
    for (int a = 0; a < ALPH; a++)
        for (int j = 0; j < qL; j++)
            prof[(size_t) a * qL + j] = aln ? (int16_t) aln[(size_t) j * ALPH + a] : (int16_t) (m.sub[a][q[j]] + cb8[j]);
    int r[3];
    int word = 0;
    swPass(prof.data(), qL, t, tL, 32, 0, go, ge, 255, bias, true, r);
    if (r[0] == 255) {
        swPass(prof.data(), qL, t, tL, 16, 0, go, ge, USHRT_MAX, 0, false, r);
        word = 1;
    }
    out[7] = word;
    const int score1 = r[0], dbEnd = r[1], qEnd = r[2];
    out[0] = score1;
    out[4] = dbEnd;
    out[2] = qEnd;
    if (dbEnd == -1) return 0.0;
    double evalue = computeEvalue(ev, score1, qL);
    bool hasLowerEvalue = evalue > evalThr;
    float qCov = computeCov(0, qEnd, qL), tCov = computeCov(0, dbEnd, tL);
    bool hasLowerCoverage = !hasCoverage(covThr, covMode, qCov, tCov);
    if (swMode == 0 || hasLowerEvalue || hasLowerCoverage) return evalue;

    // reverse pass on query[0..qEnd] reversed x target[0..dbEnd] scanned downwards (:400-463)
    const int n = qEnd + 1;
    std::vector<int16_t> rprof((size_t) ALPH * n);
    for (int a = 0; a < ALPH; a++)
        for (int j = 0; j < n; j++)
            rprof[(size_t) a * n + j] = aln ? (int16_t) aln[(size_t) (qEnd - j) * ALPH + a] : (int16_t) (m.sub[a][q[qEnd - j]] + cb8[qEnd - j]);
    int rr[3];
    if (word == 0) swPass(rprof.data(), n, t, dbEnd + 1, 32, 1, go, ge, score1, bias, true, rr);
    else swPass(rprof.data(), n, t, dbEnd + 1, 16, 1, go, ge, score1, 0, false, rr);
    if (rr[0] != score1) {
        out[7] |= 2;   // "Score of forward/backward SW differ" -- fatal in the reference (:466-473)
        return evalue;
    }
    const int dbStart = rr[1], qStart = qEnd - rr[2];
    out[3] = dbStart;
    out[1] = qStart;
    qCov = computeCov(qStart, qEnd, qL);
    tCov = computeCov(dbStart, dbEnd, tL);
    hasLowerCoverage = !hasCoverage(covThr, covMode, qCov, tCov);
    if (swMode == 1 || hasLowerCoverage) return evalue;

    int len = bandedTraceback(m, q + qStart, cb8.data() + qStart, qEnd - qStart + 1, t + dbStart,
                              dbEnd - dbStart + 1, score1, go, ge, bt, aln ? aln + (size_t) qStart * ALPH : nullptr);
    if (len < 0) {
        out[7] |= 4;
        return evalue;
    }
    int ids = 0, qp = qStart, tp = dbStart;
    for (size_t x = 0; x < bt.size(); x++) {
        if (bt[x] == 'M') {
            ids += (t[tp] == q[qp]);
            qp++;
            tp++;
        } else if (bt[x] == 'I') qp++;
        else tp++;
    }
    out[5] = ids;
    out[6] = (int) bt.size();
    if (backtrace != NULL && btCap > 0) {
        int cc = std::min(btCap - 1, (int) bt.size());
        memcpy(backtrace, bt.data(), cc);
        backtrace[cc] = 0;
    }
    return evalue;
}

double or_sw_align(void *vc, const unsigned char *q, int qL, const unsigned char *t, int tL, uint64_t dbResidues,
                   int swMode, double evalThr, int covMode, float covThr, int compBias, int isIdentity, int *out,
                   char *backtrace, int btCap) {
    return swAlignImpl(vc, q, qL, t, tL, dbResidues, swMode, evalThr, covMode, covThr, compBias, isIdentity, out, backtrace,
                       btCap, nullptr);
}

// profile query (ssw_align_private<PROFILE_SEQ>): letters = the profile's query letters (identity counting only),
// aln = alignment profile int8 [qL][21]
double or_sw_align_profile(void *vc, const unsigned char *letters, const signed char *aln, int qL, const unsigned char *t,
                           int tL, uint64_t dbResidues, int swMode, double evalThr, int covMode, float covThr, int isIdentity,
                           int *out, char *backtrace, int btCap) {
    return swAlignImpl(vc, letters, qL, t, tL, dbResidues, swMode, evalThr, covMode, covThr, 0, isIdentity, out, backtrace,
                       btCap, (const int8_t *) aln);
}

// ------------------------------------------------------------------------------------------
// clusterhits, one (query set, target set) entry (R/src/util/ClusterHits.cpp:295-492), dense K x K
// restatement with the reference's argmax/tie/stale-dmin behaviour kept literally.
//   in : K hits (qPos, tPos, strands (bit0 q, bit1 t), pval), Nq, d, cls, alpha, thresholds, lGamma table
//   out: clusterOf[K] (cluster ordinal in emission order or UINT32_MAX), memberOrder[K] (hit indices grouped
//        per emitted cluster in node append order), clusterSize/pCO/pMH per emitted cluster; returns #clusters
// ------------------------------------------------------------------------------------------
namespace {
double chLogGamma(double x) {
    // Lanczos approximation as in ClusterHits.cpp:23-63
    static const double r10 = 10.900511;
    static const double dk[11] = {2.48574089138753565546e-5, 1.05142378581721974210, -3.45687097222016235469,
                                  4.51227709466894823700, -2.98285225323576655721, 1.05639711577126713077,
                                  -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                                  4.63399473359905636708e-6, -2.71994908488607703910e-9};
    static const double gc = 2 * sqrt(exp(1.0) / M_PI);
    if (x < 0.5) return log(M_PI) - log(abs(sin(M_PI * x))) - chLogGamma(1 - x);
    if (x == 1) return 0.0;
    double sum = dk[0];
    sum += dk[1] / (x + 0);
    sum += dk[2] / (x + 1);
    sum += dk[3] / (x + 2);
    sum += dk[4] / (x + 3);
    sum += dk[5] / (x + 4);
    sum += dk[6] / (x + 5);
    sum += dk[7] / (x + 6);
    sum += dk[8] / (x + 7);
    sum += dk[9] / (x + 8);
    sum += dk[10] / (x + 9);
    return log(gc) + (x - 0.5) * log(x + r10 - 0.5) - (x - 0.5) + log(sum);
}
struct ChHit {
    double pval;
    unsigned int qPos, tPos;
    bool qStrand, tStrand;
    int idx;
};
double chClusterPval(const double *lg, int k, int m, double q0 = 0.001) {
    return 2 * lg[m + 1] - 2 * lg[m - k + 1] - lg[k + 1] + k * log(q0);
}
double chOrderingPval(const double *lg, int k, int m) { return log(1 - 1.0 * m / k) - m * log(2) - lg[m + 1]; }
double chScore(const double *lg, std::vector<ChHit> &c) {
    if (c.size() == 0) return 0.0;
    unsigned int iMax = 0, iMin = INT_MAX, jMax = 0, jMin = INT_MAX;
    for (size_t l = 0; l < c.size(); l++) {
        iMax = std::max(iMax, c[l].qPos);
        iMin = std::min(iMin, c[l].qPos);
        jMax = std::max(jMax, c[l].tPos);
        jMin = std::min(jMin, c[l].tPos);
    }
    int spanI = iMax - iMin + 1, spanJ = jMax - jMin + 1;
    int span = spanI > spanJ ? spanI : spanJ;
    int k = (int) c.size();
    std::sort(c.begin(), c.end(), [](const ChHit &a, const ChHit &b) { return a.qPos < b.qPos; });
    int m = 0;
    for (size_t l = 0; l + 1 < c.size(); l++) {
        bool sameOrder = c[l + 1].tPos > c[l].tPos;
        bool s1 = c[l].qStrand == c[l].tStrand;
        bool s2 = c[l + 1].qStrand == c[l + 1].tStrand;
        if ((s1 == sameOrder) && (s2 == sameOrder)) m++;
    }
    return -0.5 * chClusterPval(lg, k, span) - 0.5 * chOrderingPval(lg, k, m);
}
bool chCompatible(const std::vector<ChHit> &a, const std::vector<ChHit> &b, unsigned int d) {
    unsigned int iMax1 = 0, iMin1 = INT_MAX, jMax1 = 0, jMin1 = INT_MAX;
    for (size_t l = 0; l < a.size(); l++) {
        iMax1 = std::max(iMax1, a[l].qPos); iMin1 = std::min(iMin1, a[l].qPos);
        jMax1 = std::max(jMax1, a[l].tPos); jMin1 = std::min(jMin1, a[l].tPos);
    }
    unsigned int iMax2 = 0, iMin2 = INT_MAX, jMax2 = 0, jMin2 = INT_MAX;
    for (size_t l = 0; l < b.size(); l++) {
        iMax2 = std::max(iMax2, b[l].qPos); iMin2 = std::min(iMin2, b[l].qPos);
        jMax2 = std::max(jMax2, b[l].tPos); jMin2 = std::min(jMin2, b[l].tPos);
    }
    return (std::min(jMin1 - jMax2, jMin2 - jMax1) <= d && std::min(iMin1 - iMax2, iMin2 - iMax1) <= d);   // unsigned
}
double chGroupScore(const double *lg, const std::vector<std::vector<int> > &nodes, const std::vector<ChHit> &match,
                    int i, int j, unsigned int d) {
    std::vector<ChHit> c1, c2, c;
    if (nodes[i].size() != 0 && nodes[j].size() != 0) {
        for (size_t m = 0; m < nodes[i].size(); m++) c1.push_back(match[nodes[i][m]]);
        for (size_t n = 0; n < nodes[j].size(); n++) c2.push_back(match[nodes[j][n]]);
        if (chCompatible(c1, c2, d)) {
            c.insert(c.begin(), c1.begin(), c1.end());
            c.insert(c.end(), c2.begin(), c2.end());
        }
    }
    return chScore(lg, c);
}
double chMultihit(const double *lg, const std::vector<ChHit> &cluster, int Nq, double alpha) {
    size_t k = 0;
    double r = 0;
    double pvalThreshold = alpha / (Nq + 1);
    double logPvalThr = log(pvalThreshold);
    for (size_t i = 0; i < cluster.size(); ++i) {
        double logPvalue = log(cluster[i].pval);
        if (logPvalue < logPvalThr) {
            k++;
            r -= logPvalue - logPvalThr;
        }
    }
    if (r == 0) return 1.0;
    if (std::isinf(r)) return 0.0;
    double expMinusR = exp(-r);
    if (expMinusR == 0) return 0.0;
    double sum = 0;
    for (size_t i = 0; i < k - 1; ++i) sum += pow(r, i) / exp(lg[i + 1]);
    return expMinusR * sum;
}
}  // namespace

void or_lgamma_table(double *out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) out[i] = chLogGamma(i * 1.0);
}

int or_clusterhits(uint32_t K, const uint32_t *qPos, const uint32_t *tPos, const uint8_t *strands, const double *pval,
                   uint32_t Nq, uint32_t d, uint32_t cls, double alpha, float pCluThr, float pMHThr,
                   const double *lg, uint32_t *clusterOf, uint32_t *memberOrder, uint32_t *clusterSize, double *pCO,
                   double *pMH, double *mergeTrace /* optional: 3 doubles per merge (i1,i2,score) */, uint32_t *nMerges) {
    for (uint32_t i = 0; i < K; i++) clusterOf[i] = UINT32_MAX;
    if (nMerges) *nMerges = 0;
    if (K == 1 || K == 0) return 0;
    std::vector<ChHit> match(K);
    for (uint32_t i = 0; i < K; i++) {
        match[i].pval = pval[i];
        match[i].qPos = qPos[i];
        match[i].tPos = tPos[i];
        match[i].qStrand = strands[i] & 1;
        match[i].tStrand = (strands[i] >> 1) & 1;
        match[i].idx = (int) i;
    }
    std::vector<std::vector<double> > D(K, std::vector<double>(K, 0.0));
    std::vector<int> dmin(K, 0);
    std::vector<std::vector<int> > nodes(K);
    for (uint32_t n = 0; n < K; n++) nodes[n].push_back(n);
    for (uint32_t i = 0; i < K; i++) {
        for (uint32_t j = 0; j < K; j++) {
            if (i == j) D[i][j] = 0.0;
            else D[i][j] = chGroupScore(lg, nodes, match, i, j, d);
            dmin[i] = (D[i][j] > D[i][dmin[i]]) ? j : dmin[i];
        }
    }
    double maxScore = DBL_MAX;
    bool isFirstIter = true;
    double sMin = -0.5 * chClusterPval(lg, 2, d + 1) - 0.5 * chOrderingPval(lg, 2, 1);
    uint32_t merges = 0;
    while (isFirstIter || (maxScore >= sMin)) {
        size_t i1 = 0, i2 = 0;
        for (size_t i = 0; i < K; i++) i1 = (D[i][dmin[i]] > D[i1][dmin[i1]]) ? i : i1;
        i2 = dmin[i1];
        maxScore = D[i1][i2];
        if (maxScore != 0) {
            if (isFirstIter) isFirstIter = false;
        } else {
            break;
        }
        if (mergeTrace) {
            mergeTrace[3 * merges] = (double) i1;
            mergeTrace[3 * merges + 1] = (double) i2;
            mergeTrace[3 * merges + 2] = maxScore;
        }
        merges++;
        for (size_t n = 0; n < nodes[i2].size(); n++) nodes[i1].push_back(nodes[i2][n]);
        nodes[i2].clear();
        for (size_t j = 0; j < K; j++) {
            if (i1 == j || i2 == j) {
                D[i1][j] = 0.0;
                D[j][i1] = 0.0;
            } else {
                D[i1][j] = chGroupScore(lg, nodes, match, (int) i1, (int) j, d);
                D[j][i1] = D[i1][j];
            }
            D[i2][j] = 0.0;
            D[j][i2] = 0.0;
            if (j != 0) dmin[i1] = (D[i1][j] > D[i1][dmin[i1]]) ? (int) j : dmin[i1];
            else dmin[i1] = (int) j;
            if (j != i1 && j != i2) dmin[j] = (D[j][i1] > D[j][dmin[j]]) ? (int) i1 : dmin[j];
        }
    }
    if (nMerges) *nMerges = merges;
    int nClu = 0;
    uint32_t w = 0;
    for (size_t i = 0; i < nodes.size(); i++) {
        if (nodes[i].size() >= cls) {
            std::vector<ChHit> cluster;
            for (size_t j = 0; j < nodes[i].size(); j++) cluster.push_back(match[nodes[i][j]]);
            std::vector<ChHit> tmp = cluster;
            double co = exp(-chScore(lg, tmp));
            // the reference sorts `cluster` in place (findConservedPairs) before printing, so members
            // are emitted in ascending qPos order
            cluster = tmp;
            double mh = chMultihit(lg, cluster, (int) Nq, alpha);
            if (co <= pCluThr && mh <= pMHThr) {
                pCO[nClu] = co;
                pMH[nClu] = mh;
                clusterSize[nClu] = (uint32_t) cluster.size();
                for (size_t j = 0; j < cluster.size(); j++) {
                    clusterOf[cluster[j].idx] = nClu;
                    memberOrder[w++] = cluster[j].idx;
                }
                nClu++;
            }
        }
    }
    return nClu;
}

double or_evalue(uint64_t dbResidues, double score, double qLen) {
    Evaluer ev;
    initEvaluer(ev, dbResidues);
    return computeEvalue(ev, score, qLen);
}
double or_bitscore(double score) {
    Evaluer ev;
    initEvaluer(ev, 1);
    return computeBitScore(ev, score);
}

}  // extern "C"
