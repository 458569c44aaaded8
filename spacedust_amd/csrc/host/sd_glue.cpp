// Host glue between `align` and `clusterhits`, fused in-process: what the reference
// does with five sub-commands and four text DBs round trips --
//   Alignment::checkCriteria + sort (M/src/alignment/Alignment.cpp:389-405,548-567, Matcher.h:157-168)
//   besthitbyset   (R/src/util/besthitbyset.cpp:41-144, simple-best-hit mode)
//   mergeresultsbyset (M/src/util/mergeresultsbyset.cpp:49-65)
//   combinehits    (R/src/util/combinehits.cpp:74-234, multihit aggregation mode, p0 = 1e-6)
//   summarizeresults (R/src/util/SummarizeResults.cpp:77-112)
// -- including the %.3E re-quantisation of the E-value / log P / P between the steps, because those rounded
// text values are what the reference's next step parses and compares.
#include "sd_host.h"
#include "spacedust_gpu.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct BestHit {
    uint32_t q, t;          // protein indices (DB keys)
    uint32_t qSet, tSet;
    double pval;            // strtod of the %.3E text written by combinehits
    char pvalText[16];
    char seqIdText[8];
    char evalText[16];
    int32_t qStart, qEnd, qLen, tStart, tEnd, tLen;
    std::string cigar;
};

double computeLogPval(double eval, double logCalibration) {   // besthitbyset.cpp:10-20
    if (eval == 0) return log(DBL_MIN) - logCalibration;
    else if (eval > 0 && eval < 10e-4) return log(eval) - logCalibration;
    else return log(1 - exp(-eval)) - logCalibration;
}

}  // namespace

struct sd_agg {
    std::vector<uint32_t> qSetOf, tSetOf;
    uint32_t nQSets = 0, nTSets = 0;
    double evalThr = 10.0;
    int covMode = 2;
    float covThr = 0.8f;
    int alnLenThr = 30;
    float seqIdThr = 0.0f;
    bool filterSelfMatch = true;
    std::vector<BestHit> best;       // after besthitbyset + combinehits filter, any order until finish()
    uint64_t nAligned = 0, nAccepted = 0;
    // finish(): entries sorted by (qSet, tSet), hits by query key
    std::vector<uint64_t> entryOff;
    std::vector<uint32_t> entryQSet, entryTSet;
};

extern "C" {

int sd_agg_create(const uint32_t *qSetOf, uint32_t nQ, const uint32_t *tSetOf, uint32_t nT, uint32_t nQSets,
                  uint32_t nTSets, double evalThr, int covMode, float covThr, int alnLenThr, int filterSelfMatch,
                  sd_agg **out) {
    if (!qSetOf || !tSetOf || !out) return SD_EINVAL;
    sd_agg *a = new sd_agg();
    a->qSetOf.assign(qSetOf, qSetOf + nQ);
    a->tSetOf.assign(tSetOf, tSetOf + nT);
    a->nQSets = nQSets;
    a->nTSets = nTSets;
    a->evalThr = evalThr;
    a->covMode = covMode;
    a->covThr = covThr;
    a->alnLenThr = alnLenThr;
    a->filterSelfMatch = filterSelfMatch != 0;
    *out = a;
    return SD_OK;
}

void sd_agg_destroy(sd_agg *a) { delete a; }

// pairs of one query must be contiguous.  qLen/tLen per pair.
int sd_agg_add(sd_agg *a, uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const sd_sw_result *res,
               const uint8_t *isIdentity, const int32_t *qLen, const int32_t *tLen, const char *btPool) {
    if (!a || (nPairs && (!pairQ || !pairT || !res || !qLen || !tLen))) return SD_EINVAL;
    // group boundaries
    std::vector<uint32_t> groupStart;
    for (uint32_t i = 0; i < nPairs; i++)
        if (i == 0 || pairQ[i] != pairQ[i - 1]) groupStart.push_back(i);
    groupStart.push_back(nPairs);
    const size_t nGroups = groupStart.size() - 1;
    std::vector<std::vector<BestHit> > perGroup(nGroups);
    uint64_t accepted = 0;
    const double logPvalThr = log(10e-7);   // combinehits.cpp:101-103
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : accepted)
    for (size_t g = 0; g < nGroups; g++) {
        struct Cand {
            uint32_t i;
            double eval;
            int bits;
            int dbLen;
            uint32_t key;
            float seqId;
        };
        std::vector<Cand> cands;
        for (uint32_t i = groupStart[g]; i < groupStart[g + 1]; i++) {
            const sd_sw_result &r = res[i];
            const bool ident = isIdentity && isIdentity[i];
            Cand c;
            c.i = i;
            c.eval = r.evalue;
            c.bits = static_cast<int>(sd_host_bitscore((double) (uint32_t) r.score) + 0.5);   // Matcher.cpp:130
            c.dbLen = tLen[i];
            c.key = pairT[i];
            if (ident) {
                c.seqId = 1.0f;   // Alignment.cpp:382-387
            } else {
                if (r.btLen <= 0 || r.qStart < 0 || r.tStart < 0) continue;   // stopped at a gate: fails checkCriteria
                const float qcov = sd::computeCov(r.qStart, r.qEnd, qLen[i]);
                const float dbcov = sd::computeCov(r.tStart, r.tEnd, tLen[i]);
                c.seqId = static_cast<float>(r.identical) / static_cast<float>(r.btLen);   // Util::computeSeqId, SEQ_ID_ALN_LEN
                const bool ok = (r.evalue <= a->evalThr) && (c.seqId >= a->seqIdThr) &&
                                sd::hasCoverage(a->covThr, a->covMode, qcov, dbcov) && (r.btLen >= a->alnLenThr);
                if (!ok) continue;
            }
            cands.push_back(c);
        }
        accepted += cands.size();
        // Matcher::compareHits order (Matcher.h:157-168)
        std::sort(cands.begin(), cands.end(), [](const Cand &x, const Cand &y) {
            if (x.eval != y.eval) return x.eval < y.eval;
            if (x.bits != y.bits) return x.bits > y.bits;
            if (x.dbLen != y.dbLen) return x.dbLen < y.dbLen;
            return x.key < y.key;
        });
        // besthitbyset: first (= smallest text E-value, strict <) hit per target set, sets ascending
        std::vector<std::pair<uint32_t, const Cand *> > bestOfSet;
        {
            std::vector<std::pair<uint32_t, double> > bestEval;
            for (const Cand &c : cands) {
                const uint32_t ts = a->tSetOf[c.key];
                char txt[32];
                snprintf(txt, sizeof(txt), "%.3E", c.eval);
                const double ev = strtod(txt, NULL);
                size_t s = 0;
                for (; s < bestOfSet.size(); s++)
                    if (bestOfSet[s].first == ts) break;
                if (s == bestOfSet.size()) {
                    bestOfSet.push_back(std::make_pair(ts, (const Cand *) NULL));
                    bestEval.push_back(std::make_pair(ts, DBL_MAX));
                }
                if (ev < bestEval[s].second) {
                    bestEval[s].second = ev;
                    bestOfSet[s].second = &c;
                }
            }
        }
        std::sort(bestOfSet.begin(), bestOfSet.end(), [](const std::pair<uint32_t, const Cand *> &x, const std::pair<uint32_t, const Cand *> &y) { return x.first < y.first; });
        const uint32_t q = pairQ[groupStart[g]];
        const uint32_t qs = a->qSetOf[q];
        for (size_t s = 0; s < bestOfSet.size(); s++) {
            const Cand *c = bestOfSet[s].second;
            if (c == NULL) continue;
            const uint32_t ts = bestOfSet[s].first;
            if (a->filterSelfMatch && qs == ts) continue;   // combinehits.cpp:83
            const sd_sw_result &r = res[c->i];
            BestHit b;
            snprintf(b.evalText, sizeof(b.evalText), "%.3E", c->eval);
            const double evParsed = strtod(b.evalText, NULL);
            char lpText[32];
            snprintf(lpText, sizeof(lpText), "%.3E", computeLogPval(evParsed, log(1)));   // besthitbyset.cpp:129
            const double logP = strtod(lpText, NULL);
            if (!(logP < logPvalThr)) continue;                                            // combinehits.cpp:107-112
            snprintf(b.pvalText, sizeof(b.pvalText), "%.3E", exp(logP));                   // combinehits.cpp:218-221
            b.pval = strtod(b.pvalText, NULL);
            char *e = sd::seqIdToBuffer(c->seqId, b.seqIdText);
            *e = '\0';
            b.q = q; b.t = c->key; b.qSet = qs; b.tSet = ts;
            b.qStart = r.qStart; b.qEnd = r.qEnd; b.qLen = qLen[c->i];
            b.tStart = r.tStart; b.tEnd = r.tEnd; b.tLen = tLen[c->i];
            if (btPool && r.btLen > 0) b.cigar = sd::compressBacktrace(btPool + r.btOffset, (size_t) r.btLen);
            perGroup[g].push_back(b);
        }
    }
    a->nAligned += nPairs;
    a->nAccepted += accepted;
    for (size_t g = 0; g < nGroups; g++)
        for (size_t x = 0; x < perGroup[g].size(); x++) a->best.push_back(std::move(perGroup[g][x]));
    return SD_OK;
}

// sort into (qSet, tSet) entries; hits inside an entry by query key (mergeresultsbyset order)
int sd_agg_finish(sd_agg *a, uint64_t *nEntries, uint64_t *nHits) {
    std::sort(a->best.begin(), a->best.end(), [](const BestHit &x, const BestHit &y) {
        if (x.qSet != y.qSet) return x.qSet < y.qSet;
        if (x.tSet != y.tSet) return x.tSet < y.tSet;
        return x.q < y.q;
    });
    a->entryOff.clear();
    a->entryQSet.clear();
    a->entryTSet.clear();
    for (size_t i = 0; i < a->best.size(); i++) {
        if (i == 0 || a->best[i].qSet != a->best[i - 1].qSet || a->best[i].tSet != a->best[i - 1].tSet) {
            a->entryOff.push_back(i);
            a->entryQSet.push_back(a->best[i].qSet);
            a->entryTSet.push_back(a->best[i].tSet);
        }
    }
    a->entryOff.push_back(a->best.size());
    if (nEntries) *nEntries = a->entryQSet.size();
    if (nHits) *nHits = a->best.size();
    return SD_OK;
}

int sd_agg_stats(sd_agg *a, uint64_t *nAligned, uint64_t *nAccepted) {
    *nAligned = a->nAligned;
    *nAccepted = a->nAccepted;
    return SD_OK;
}

int sd_agg_get(sd_agg *a, uint64_t *entryOff, uint32_t *entryQSet, uint32_t *entryTSet, uint32_t *hitQ, uint32_t *hitT,
               double *pval) {
    memcpy(entryOff, a->entryOff.data(), a->entryOff.size() * sizeof(uint64_t));
    memcpy(entryQSet, a->entryQSet.data(), a->entryQSet.size() * sizeof(uint32_t));
    memcpy(entryTSet, a->entryTSet.data(), a->entryTSet.size() * sizeof(uint32_t));
    for (size_t i = 0; i < a->best.size(); i++) {
        hitQ[i] = a->best[i].q;
        hitT[i] = a->best[i].t;
        pval[i] = a->best[i].pval;
    }
    return SD_OK;
}

// summarizeresults: one "#..." line per emitted cluster followed by its member lines (ascending query position).
// names: concatenated lookup names + offsets; sources: per set.  clusterKeyBase numbers the clusters.
// canonical != 0 drops the cluster key and the leading ">qname" field (the `cut -f2-` form used for comparisons).
int sd_agg_write_tsv(sd_agg *a, const char *path, const uint32_t *clusterOfHit, const uint32_t *rankInCluster,
                     const uint32_t *nClusters, const double *pCO, const double *pMH, const uint32_t *clusterSize,
                     const char *qNames, const uint64_t *qNameOff, const char *tNames, const uint64_t *tNameOff,
                     const char *qSources, const uint64_t *qSourceOff, const char *tSources, const uint64_t *tSourceOff,
                     int canonical, uint64_t *nClusterLines, uint64_t *nHitLines) {
    FILE *f = fopen(path, "w");
    if (!f) return SD_EINVAL;
    uint64_t key = 0, nc = 0, nh = 0;
    std::vector<uint32_t> order;
    for (size_t e = 0; e + 1 < a->entryOff.size(); e++) {
        const uint64_t off = a->entryOff[e], end = a->entryOff[e + 1];
        const uint32_t n = nClusters[e];
        for (uint32_t c = 0; c < n; c++) {
            order.assign(clusterSize[off + c], 0);
            for (uint64_t h = off; h < end; h++)
                if (clusterOfHit[h] == c) order[rankInCluster[h]] = (uint32_t) (h - off);
            const uint32_t qs = a->entryQSet[e], ts = a->entryTSet[e];
            char co[32], mh[32];
            snprintf(co, sizeof(co), "%.3E", pCO[off + c]);   // SSTR(double) = "{:.3E}" (M/src/commons/Util.cpp:658-660)
            snprintf(mh, sizeof(mh), "%.3E", pMH[off + c]);
            if (!canonical) fprintf(f, "#%llu\t", (unsigned long long) key);
            fprintf(f, "%.*s\t%.*s\t%s\t%s\t%u\n", (int) (qSourceOff[qs + 1] - qSourceOff[qs]), qSources + qSourceOff[qs],
                    (int) (tSourceOff[ts + 1] - tSourceOff[ts]), tSources + tSourceOff[ts], co, mh, clusterSize[off + c]);
            nc++;
            for (uint32_t m = 0; m < clusterSize[off + c]; m++) {
                const BestHit &b = a->best[off + order[m]];
                if (!canonical) fprintf(f, ">%.*s\t", (int) (qNameOff[b.q + 1] - qNameOff[b.q]), qNames + qNameOff[b.q]);
                fprintf(f, "%.*s\t%s\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%s\n", (int) (tNameOff[b.t + 1] - tNameOff[b.t]),
                        tNames + tNameOff[b.t], b.pvalText, b.seqIdText, b.evalText, b.qStart, b.qEnd, b.qLen, b.tStart,
                        b.tEnd, b.tLen, b.cigar.c_str());
                nh++;
            }
            key++;
        }
    }
    fclose(f);
    if (nClusterLines) *nClusterLines = nc;
    if (nHitLines) *nHitLines = nh;
    return SD_OK;
}

}  // extern "C"
