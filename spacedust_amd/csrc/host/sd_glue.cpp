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
#include <memory>
#include <string>
#include <vector>
#include <omp.h>
#include <parallel/algorithm>
#include <charconv>

namespace {

struct BestHit {
    uint32_t q, t;          // protein indices (DB keys)
    uint32_t qSet, tSet;
    double pval;            // strtod of the %.3E text written by combinehits
    char pvalText[16];
    char seqIdText[8];
    char evalText[16];
    int32_t qStart, qEnd, qLen, tStart, tEnd, tLen;
    uint32_t arena;         // which thread's cigar arena
    uint32_t cigarLen;
    uint64_t cigarOff;
};

// snprintf("%.3E") followed by strtod, i.e. what one reference module writes and the next one parses.  to_chars / from_chars produce
// the same digits and the same double (both are exact conversions).  The four digits are found without the exact conversion whenever
// that is safe: x * 10^(3 - e) in 80-bit arithmetic is within 1e-15 of the true quotient, so unless it lies within 1e-6 of a rounding
// boundary (then: to_chars) floor(. + 0.5) is the digit string printf rounds the exact value to.  (to_chars was 65 of the 85 ns of a round trip,
// and a matched hit makes three of them.)
struct Pow10L {
    long double p[700];   // 10^(i - 350), correctly rounded to the 64-bit significand (glibc's strtold)
    Pow10L() {
        char t[16];
        for (int i = 0; i < 700; i++) {
            snprintf(t, sizeof(t), "1e%d", i - 350);
            p[i] = strtold(t, nullptr);
        }
    }
};
const Pow10L kPow10L;
// (the shortcut below reads the 64-bit significand of an x87 long double; any other long double takes the exact conversions)
constexpr bool kX87LongDouble = sizeof(long double) == 16 && LDBL_MANT_DIG == 64;

double quantise3ESlow(double v, char *text) {
    std::to_chars_result r = std::to_chars(text, text + 24, v, std::chars_format::scientific, 3);
    *r.ptr = '\0';
    for (char *p = text; p < r.ptr; p++)
        if (*p >= 'a' && *p <= 'z') *p = (char) (*p - 'a' + 'A');   // 'E'; "INF" / "NAN" as %E writes them
    double back = 0.0;
    if (std::from_chars(text, r.ptr, back).ec != std::errc()) back = strtod(text, nullptr);   // out of range: strtod's +-HUGE_VAL / 0
    return back;
}

double quantise3E(double x, char *text) {
    const double v = fabs(x);   // printf rounds the magnitude (to nearest, ties to even digit): the sign is copied
    if (!kX87LongDouble || !(v >= 1e-300 && v <= 1e300)) return quantise3ESlow(x, text);   // zero, subnormal, huge, NaN
    uint64_t bits;
    memcpy(&bits, &v, 8);
    const int e2 = (int) (bits >> 52) - 1022;   // v = f * 2^e2 with 0.5 <= f < 1 (v is normal here)
    int e10 = (int) floor((double) (e2 - 1) * 0.30102999566398120);   // <= floor(log10 v) <= this + 1
    long double scaled = (long double) v * kPow10L.p[3 - e10 + 350];
    if (scaled >= 10000.0L) {
        e10++;
        scaled = (long double) v * kPow10L.p[3 - e10 + 350];
    }
    if (!(scaled >= 1000.0L && scaled < 10000.0L)) return quantise3ESlow(x, text);
    const int fi = (int) scaled;   // truncation = floor (positive)
    const long double frac = scaled - (long double) fi;
    if (fabsl(frac - 0.5L) < 1e-6L || frac < 1e-6L || frac > 1.0L - 1e-6L) return quantise3ESlow(x, text);
    int m = fi + (frac > 0.5L ? 1 : 0);
    if (m == 10000) {
        m = 1000;
        e10++;
    }
    char *p = text;
    if (x < 0) *p++ = '-';
    *p++ = (char) ('0' + m / 1000);
    *p++ = '.';
    *p++ = (char) ('0' + m / 100 % 10);
    *p++ = (char) ('0' + m / 10 % 10);
    *p++ = (char) ('0' + m % 10);
    *p++ = 'E';
    int ae = e10;
    if (ae < 0) {
        *p++ = '-';
        ae = -ae;
    } else {
        *p++ = '+';
    }
    if (ae >= 100) *p++ = (char) ('0' + ae / 100);
    *p++ = (char) ('0' + ae / 10 % 10);
    *p++ = (char) ('0' + ae % 10);
    *p = '\0';
    // strtod of that text = m * 10^(e10 - 3) rounded once.  In 80-bit arithmetic the product is within one unit of its 64-bit significand
    // of the true value (table entry and product: half a unit each), and rounding it to 53 bits gives the correctly rounded double unless
    // its eleven extra bits are within a few units of the half-way pattern (1 in 250: from_chars decides)
    const long double prod = (long double) m * kPow10L.p[e10 - 3 + 350];
    uint64_t sig;
    memcpy(&sig, &prod, 8);   // x87 layout: the 64-bit significand, then sign + exponent
    const int low = (int) (sig & 0x7FFu);
    double back;
    if (low >= 0x400 - 4 && low <= 0x400 + 4) {
        back = 0.0;
        const char *digits = x < 0 ? text + 1 : text;
        if (std::from_chars(digits, (const char *) p, back).ec != std::errc()) back = strtod(digits, nullptr);
    } else {
        back = (double) prod;
    }
    return x < 0 ? -back : back;
}

// Append-only storage in fixed blocks: what is appended never moves.  The per-thread hit lists of a proteome-scale range grow to
// hundreds of MB; as std::vectors they were copied at every doubling (the "push_back" share of sd_agg_add: 10 - 40 % of its CPU time).
template <typename T, unsigned LOG2>
struct Blocks {
    std::vector<std::unique_ptr<T[]> > blocks;
    size_t n = 0;
    size_t size() const { return n; }
    T &operator[](size_t i) { return blocks[i >> LOG2][i & ((1u << LOG2) - 1)]; }
    const T &operator[](size_t i) const { return blocks[i >> LOG2][i & ((1u << LOG2) - 1)]; }
    void push_back(const T &v) {
        if ((n >> LOG2) == blocks.size()) blocks.emplace_back(new T[(size_t) 1 << LOG2]);
        (*this)[n] = v;
        n++;
    }
    // room for `len` contiguous items: the index of the first (the tail of a block that cannot hold them stays unused)
    size_t appendRun(const T *src, size_t len) {
        const size_t cap = (size_t) 1 << LOG2;
        if (len > cap) return SIZE_MAX;
        if (n == blocks.size() * cap || (n & (cap - 1)) + len > cap) {
            n = blocks.size() * cap;
            blocks.emplace_back(new T[cap]);
        }
        const size_t at = n;
        memcpy(&(*this)[at], src, len * sizeof(T));
        n += len;
        return at;
    }
    void swap(Blocks &o) {
        blocks.swap(o.blocks);
        std::swap(n, o.n);
    }
};

double computeLogPval(double eval, double logCalibration) {   // besthitbyset.cpp:10-20
    if (eval == 0) return log(DBL_MIN) - logCalibration;
    else if (eval > 0 && eval < 10e-4) return log(eval) - logCalibration;
    else return log(1 - exp(-eval)) - logCalibration;
}

}  // namespace

struct sd_agg {
    std::vector<uint32_t> qSetOf, tSetOf;
    std::vector<int32_t> qLen, tLen;
    uint32_t nQSets = 0, nTSets = 0;
    double evalThr = 10.0;
    int covMode = 2;
    float covThr = 0.8f;
    int alnLenThr = 30;
    float seqIdThr = 0.0f;
    bool filterSelfMatch = true;
    bool listOrder = false;   // sd_agg_set_list_order
    bool cigarPool = false;   // sd_agg_set_pool_form: the pool of sd_agg_add holds run-length text (sd_sw_set_cigar_pool)
    // after besthitbyset + combinehits filter: per worker thread, any order until finish()
    struct HitKey { uint64_t cell; uint32_t q, idx; };  // cell = qSet * nTSets + tSet (64 bit: 30 000 x 30 000 sets and more)
    std::vector<uint32_t> qDbKey, tDbKey;               // optional DB keys (sd_agg_set_keys): order inside entries, compareHits tie-break
    std::vector<Blocks<BestHit, 13> > tBest;
    std::vector<Blocks<HitKey, 14> > tKey;
    std::vector<Blocks<char, 20> > tCigar;   // a CIGAR is contiguous inside a 1-MB block
    std::vector<const BestHit *> best;   // finish(): (qSet, tSet, q) order
    uint64_t nAligned = 0, nAccepted = 0;
    // finish(): entries sorted by (qSet, tSet), hits by query key
    std::vector<uint64_t> entryOff;
    std::vector<uint32_t> entryQSet, entryTSet;
};

extern "C" {

int sd_agg_create(const uint32_t *qSetOf, const int32_t *qLen, uint32_t nQ, const uint32_t *tSetOf, const int32_t *tLen,
                  uint32_t nT, uint32_t nQSets, uint32_t nTSets, double evalThr, int covMode, float covThr, int alnLenThr,
                  int filterSelfMatch, sd_agg **out) {
    if (!qSetOf || !tSetOf || !qLen || !tLen || !out) return SD_EINVAL;
    sd_agg *a = new sd_agg();
    a->qSetOf.assign(qSetOf, qSetOf + nQ);
    a->tSetOf.assign(tSetOf, tSetOf + nT);
    a->qLen.assign(qLen, qLen + nQ);
    a->tLen.assign(tLen, tLen + nT);
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

// the two text conversions of this file, callable on their own (tests/test_host_aggregation.py compares them with printf / strtod and a letter-by-letter loop)
int sd_host_quantise_3e(double v, char *text, double *back) {
    if (!text || !back) return SD_EINVAL;
    char t[32];
    *back = quantise3E(v, t);
    memcpy(text, t, 16);
    text[15] = '\0';
    return SD_OK;
}

int sd_host_compress_backtrace(const char *bt, uint64_t n, char *out, uint64_t cap, uint64_t *len) {
    if ((n && !bt) || !len) return SD_EINVAL;
    std::string c;
    sd::compressBacktraceAppend(bt ? bt : "", (size_t) n, c);
    *len = c.size();
    if (out) {
        if (cap < c.size()) return SD_ENOMEM;
        memcpy(out, c.data(), c.size());
    }
    return SD_OK;
}

// pairs of one query must be contiguous
int sd_agg_add(sd_agg *a, uint32_t nPairs, uint32_t qBase, const uint32_t *pairQ, const uint32_t *pairT,
               const sd_sw_result *res, const uint8_t *isIdentity, const char *btPool) {
    if (!a || (nPairs && (!pairQ || !pairT || !res))) return SD_EINVAL;
    {
        int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
        for (uint32_t i = 0; i < nPairs; i++)
            bad |= ((uint64_t) qBase + pairQ[i] >= a->qSetOf.size() || pairT[i] >= a->tSetOf.size() ||
                    // a CIGAR is at most two characters per backtrace letter and must fit one 2^20-byte block of the arena: checked here,
                    // before anything is appended (sequences are <= 65 535 residues, so a backtrace has < 2^17 letters)
                    (btPool && res[i].btLen > (1 << 19)) || (btPool && a->cigarPool && ((uint32_t) res[i].flags >> 8) > (1u << 20))) ? 1 : 0;
        if (bad) {
            if (getenv("SD_DEBUG_TIMING")) fprintf(stderr, "[sd_agg_add] rejected: a pair index outside the sets, or a backtrace of more than 2^19 letters\n");
            return SD_EINVAL;   // (nothing was added)
        }
    }
    const bool dbg = getenv("SD_DEBUG_TIMING") != NULL;
    const double t0 = omp_get_wtime();
    // group boundaries
    std::vector<uint32_t> groupStart;
    for (uint32_t i = 0; i < nPairs; i++)
        if (i == 0 || pairQ[i] != pairQ[i - 1]) groupStart.push_back(i);
    groupStart.push_back(nPairs);
    const size_t nGroups = groupStart.size() - 1;
    uint64_t accepted = 0;
    const double logPvalThr = log(10e-7);   // combinehits.cpp:101-103
    const int T = std::max(1, std::min(omp_get_max_threads(), 64));
    if ((int) a->tBest.size() < T) {
        a->tBest.resize(T);
        a->tCigar.resize(T);
        a->tKey.resize(T);
    }
    const double t1 = omp_get_wtime();
    struct Cand {
        uint32_t i;
        double eval;
        int score;   // (the rounded bit score, Matcher::compareHits' second criterion, is derived from it when two E-values tie)
        int dbLen;
        uint32_t key;   // DB key of the target (Matcher::compareHits' last criterion)
        uint32_t t;     // target index
        float seqId;
        double teval;   // list-order mode: the value the E-value's %.3E text parses to
    };
    int tooLong = 0;
#pragma omp parallel num_threads(T) reduction(+ : accepted) reduction(| : tooLong)
    {
        const int th = omp_get_thread_num();
        // thread-private containers (the shared vectors' headers sit on common cache lines), swapped back at the end
        Blocks<BestHit, 13> myBest;
        Blocks<char, 20> myCigar;
        Blocks<sd_agg::HitKey, 14> myKey;
        std::string cigar;
        myBest.swap(a->tBest[th]);
        myCigar.swap(a->tCigar[th]);
        myKey.swap(a->tKey[th]);
        // best candidate per target set: Alignment sorts a query's accepted hits with Matcher::compareHits
        // (Matcher.h:157-168) and besthitbyset keeps, per set, the first one whose %.3E text E-value is strictly
        // smaller than what it has (besthitbyset.cpp:88-101).  Rounding to text is monotone, so that is the
        // compareHits-minimum of the set; no sort and no text round trip are needed to find it.
        std::vector<Cand> slotBest;
        std::vector<uint32_t> slotSet, stamp(a->nTSets, 0), slotOf(a->nTSets, 0);
        std::vector<std::pair<uint32_t, const Cand *> > bestOfSet;
        uint32_t gen = 0;
        auto better = [](const Cand &x, const Cand &y) {
            if (x.eval != y.eval) return x.eval < y.eval;
            if (x.score != y.score) {   // Matcher.cpp:130: static_cast<int>(bit score + 0.5)
                const int xb = static_cast<int>(sd_host_bitscore((double) (uint32_t) x.score) + 0.5);
                const int yb = static_cast<int>(sd_host_bitscore((double) (uint32_t) y.score) + 0.5);
                if (xb != yb) return xb > yb;
            }
            if (x.dbLen != y.dbLen) return x.dbLen < y.dbLen;
            return x.key < y.key;
        };
#pragma omp for schedule(dynamic, 64)
        for (size_t g = 0; g < nGroups; g++) {
            gen++;
            slotBest.clear();
            slotSet.clear();
            const uint32_t q = qBase + pairQ[groupStart[g]];
            const int32_t qL = a->qLen[q];
            for (uint32_t i = groupStart[g]; i < groupStart[g + 1]; i++) {
                const sd_sw_result &r = res[i];
                const bool ident = isIdentity && isIdentity[i];
                Cand c;
                c.i = i;
                c.eval = r.evalue;
                c.dbLen = a->tLen[pairT[i]];
                c.key = a->tDbKey.empty() ? pairT[i] : a->tDbKey[pairT[i]];
                c.t = pairT[i];
                if (ident) {
                    c.seqId = 1.0f;   // Alignment.cpp:382-387
                } else {
                    if (r.btLen <= 0 || r.qStart < 0 || r.tStart < 0) continue;   // stopped at a gate: fails checkCriteria
                    const float qcov = sd::computeCov(r.qStart, r.qEnd, qL);
                    const float dbcov = sd::computeCov(r.tStart, r.tEnd, c.dbLen);
                    c.seqId = static_cast<float>(r.identical) / static_cast<float>(r.btLen);   // Util::computeSeqId, SEQ_ID_ALN_LEN
                    const bool ok = (r.evalue <= a->evalThr) && (c.seqId >= a->seqIdThr) &&
                                    sd::hasCoverage(a->covThr, a->covMode, qcov, dbcov) && (r.btLen >= a->alnLenThr);
                    if (!ok) continue;
                }
                c.score = r.score;
                accepted++;
                const uint32_t ts = a->tSetOf[c.t];
                c.teval = 0.0;
                if (a->listOrder) {
                    char txt[32];
                    c.teval = quantise3E(c.eval, txt);
                }
                if (stamp[ts] != gen) {
                    stamp[ts] = gen;
                    slotOf[ts] = (uint32_t) slotBest.size();
                    slotBest.push_back(c);
                    slotSet.push_back(ts);
                } else if (a->listOrder ? c.teval < slotBest[slotOf[ts]].teval : better(c, slotBest[slotOf[ts]])) {
                    slotBest[slotOf[ts]] = c;
                }
            }
            bestOfSet.clear();
            for (size_t x = 0; x < slotBest.size(); x++) bestOfSet.push_back(std::make_pair(slotSet[x], (const Cand *) &slotBest[x]));
            std::sort(bestOfSet.begin(), bestOfSet.end(),
                      [](const std::pair<uint32_t, const Cand *> &x, const std::pair<uint32_t, const Cand *> &y) { return x.first < y.first; });
            const uint32_t qs = a->qSetOf[q];
            for (size_t s = 0; s < bestOfSet.size(); s++) {
                const Cand *c = bestOfSet[s].second;
                if (c == NULL) continue;
                const uint32_t ts = bestOfSet[s].first;
                if (a->filterSelfMatch && qs == ts) continue;   // combinehits.cpp:83
                const sd_sw_result &r = res[c->i];
                // combinehits keeps a hit only if its log P, after two %.3E text round trips, is below log(10e-7) = -13.8155.  An
                // E-value of 1.1e-6 and more cannot get there: its text form is >= 1.0994e-6, log of that -13.7208 (larger E-values
                // take the other branch of ComputelogPval, >= -6.9), and the second rounding moves it by at most 5e-4 relative, to
                // >= -13.7277.  Four in five best hits of a proteome-scale search end here, before any formatting.
                if (c->eval >= 1.1e-6) continue;
                BestHit b;
                char evalText[32] = {0}, lpText[32], pvalText[32] = {0};   // (the record fields are zero behind the text: the same bytes every run)
                const double evParsed = quantise3E(c->eval, evalText);
                const double logP = quantise3E(computeLogPval(evParsed, log(1)), lpText);       // besthitbyset.cpp:129
                if (!(logP < logPvalThr)) continue;                                            // combinehits.cpp:107-112
                b.pval = quantise3E(exp(logP), pvalText);                                      // combinehits.cpp:218-221
                memcpy(b.evalText, evalText, sizeof(b.evalText));
                memcpy(b.pvalText, pvalText, sizeof(b.pvalText));
                b.evalText[sizeof(b.evalText) - 1] = '\0';
                b.pvalText[sizeof(b.pvalText) - 1] = '\0';
                memset(b.seqIdText, 0, sizeof(b.seqIdText));
                char *e = sd::seqIdToBuffer(c->seqId, b.seqIdText);
                *e = '\0';
                b.q = q; b.t = c->t; b.qSet = qs; b.tSet = ts;
                b.qStart = r.qStart; b.qEnd = r.qEnd; b.qLen = qL;
                b.tStart = r.tStart; b.tEnd = r.tEnd; b.tLen = c->dbLen;
                b.arena = (uint32_t) th;
                b.cigarOff = 0;
                b.cigarLen = 0;
                if (btPool && r.btLen > 0 && a->cigarPool) {   // the text came from the device (flags bits 8..: its length)
                    const size_t nTxt = (size_t) ((uint32_t) r.flags >> 8);
                    const size_t at = myCigar.appendRun(btPool + r.btOffset, nTxt);
                    if (at == SIZE_MAX) {
                        tooLong = 1;
                        continue;
                    }
                    b.cigarOff = at;
                    b.cigarLen = (uint32_t) nTxt;
                } else if (btPool && r.btLen > 0) {
                    cigar.clear();
                    sd::compressBacktraceAppend(btPool + r.btOffset, (size_t) r.btLen, cigar);
                    const size_t at = myCigar.appendRun(cigar.data(), cigar.size());
                    if (at == SIZE_MAX) {   // a CIGAR of more than 2^20 characters: no sequence of <= 65 535 residues has one
                        tooLong = 1;
                        continue;
                    }
                    b.cigarOff = at;
                    b.cigarLen = (uint32_t) cigar.size();
                }
                sd_agg::HitKey hk;
                hk.cell = (uint64_t) qs * a->nTSets + ts; hk.q = a->qDbKey.empty() ? q : a->qDbKey[q]; hk.idx = (uint32_t) myBest.size();
                myKey.push_back(hk);
                myBest.push_back(b);
            }
        }
        myBest.swap(a->tBest[th]);
        myCigar.swap(a->tCigar[th]);
        myKey.swap(a->tKey[th]);
    }
    if (tooLong) return SD_EINVAL;
    a->nAligned += nPairs;
    a->nAccepted += accepted;
    if (dbg) fprintf(stderr, "[sd_agg_add] pairs %u groups %zu threads %d: setup %.1f ms, parallel %.1f ms\n", nPairs, nGroups, T,
                     (t1 - t0) * 1e3, (omp_get_wtime() - t1) * 1e3);
    return SD_OK;
}

// sort into (qSet, tSet) entries; hits inside an entry by query key (mergeresultsbyset order)
int sd_agg_finish(sd_agg *a, uint64_t *nEntries, uint64_t *nHits) {
    if (!a) return SD_EINVAL;
    size_t total = 0;
    for (size_t t = 0; t < a->tBest.size(); t++) total += a->tBest[t].size();
    // (cell, query key) is unique per hit, so the sorted order does not depend on how the hits were spread over threads;
    // sorting the hits themselves (not a dense table over all nQSets x nTSets cells) keeps this linear in the output
    struct Ref {
        uint64_t cell;
        uint32_t q;
        const BestHit *hit;
    };
    std::vector<Ref> refs(total);
    size_t w = 0;
    for (size_t t = 0; t < a->tKey.size(); t++)
        for (size_t x = 0; x < a->tKey[t].size(); x++) {
            const sd_agg::HitKey &k = a->tKey[t][x];
            refs[w].cell = k.cell;
            refs[w].q = k.q;
            refs[w].hit = &a->tBest[t][k.idx];
            w++;
        }
    __gnu_parallel::sort(refs.begin(), refs.end(), [](const Ref &x, const Ref &y) { return x.cell != y.cell ? x.cell < y.cell : x.q < y.q; });
    a->best.resize(total);
    a->entryOff.clear();
    a->entryQSet.clear();
    a->entryTSet.clear();
    for (size_t i = 0; i < total; i++) {
        a->best[i] = refs[i].hit;
        if (i == 0 || refs[i].cell != refs[i - 1].cell) {
            a->entryOff.push_back(i);
            a->entryQSet.push_back((uint32_t) (refs[i].cell / a->nTSets));
            a->entryTSet.push_back((uint32_t) (refs[i].cell % a->nTSets));
        }
    }
    a->entryOff.push_back(total);
    if (nEntries) *nEntries = a->entryQSet.size();
    if (nHits) *nHits = a->best.size();
    return SD_OK;
}

// on != 0: the records of a query arrive in the order of the lines of its alignment DB entry, and that order is NOT Matcher::compareHits
// order -- the merged result of an iterative search (mergedbs concatenates the iterations' sorted lists).  besthitbyset then keeps, per
// target set, the first line whose %.3E text E-value is strictly smaller than what it holds (besthitbyset.cpp:88-101): among equal
// E-values -- equal raw scores of one query -- the line of the EARLIER iteration, not the compareHits minimum.  (For one sorted list
// the two rules pick the same line; the streaming pipeline hands its records over in device order and needs the default rule.)
int sd_agg_set_list_order(sd_agg *a, int on) {
    if (!a) return SD_EINVAL;
    a->listOrder = on != 0;
    return SD_OK;
}

int sd_agg_set_pool_form(sd_agg *a, int cigarText) {
    if (!a) return SD_EINVAL;
    a->cigarPool = cigarText != 0;
    return SD_OK;
}

int sd_agg_set_keys(sd_agg *a, const uint32_t *qKeys, const uint32_t *tKeys) {
    if (!a) return SD_EINVAL;
    if (qKeys) a->qDbKey.assign(qKeys, qKeys + a->qSetOf.size());
    if (tKeys) a->tDbKey.assign(tKeys, tKeys + a->tSetOf.size());
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
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < a->best.size(); i++) {
        hitQ[i] = a->best[i]->q;
        hitT[i] = a->best[i]->t;
        pval[i] = a->best[i]->pval;
    }
    return SD_OK;
}

// summarizeresults: one "#..." line per emitted cluster followed by its member lines (ascending query position).
// names: concatenated lookup names + offsets; sources: per set.  clusterKeyBase numbers the clusters.
// canonical != 0 drops the cluster key and the leading ">qname" field (the `cut -f2-` form used for comparisons).
// ---- cluster records: the result of a range as one self-contained byte string -- what a rank hands to sd_gather_results
// and what the TSV is written from (here, or on the root from the gathered buffer).  Little endian, per cluster:
//   u32 members, u32 qSet, u32 tSet, u32 0, f64 pCO, f64 pMH, then per member (in cluster order):
//   u32 q, u32 t, char pval[16], char seqId[8], char eval[16], i32 qStart, qEnd, qLen, tStart, tEnd, tLen, u32 cigarLen,
//   cigar bytes padded to a multiple of 4
namespace {
struct RecCluster {
    uint32_t members, qSet, tSet, pad;
    double pCO, pMH;
};
struct RecMember {
    uint32_t q, t;
    char pval[16], seqId[8], eval[16];
    int32_t qStart, qEnd, qLen, tStart, tEnd, tLen;
    uint32_t cigarLen;
};
static_assert(sizeof(RecCluster) == 32 && sizeof(RecMember) == 76, "record layout");
}  // namespace

int sd_agg_records(sd_agg *a, const uint32_t *clusterOfHit, const uint32_t *rankInCluster, const uint32_t *nClusters,
                   const double *pCO, const double *pMH, const uint32_t *clusterSize, void *out, uint64_t cap, uint64_t *bytes) {
    if (!a || !bytes) return SD_EINVAL;
    // Linear in the hits and parallel over the entries (a 1 000-proteome step holds 3 500 entries with 180 clusters of 750 hits each:
    // looking for every cluster's members among all hits of its entry was a second per step, paid twice -- size probe and build -- by every
    // rank of a multi-GPU run behind its last kernel).  Pass 1: bytes per entry; pass 2: every entry written at its offset, the members
    // of a cluster placed through the clusters' start positions (sizes in clusterSize[off + c], rank inside from rankInCluster).
    const size_t nE = a->entryOff.empty() ? 0 : a->entryOff.size() - 1;
    std::vector<uint64_t> at(nE + 1, 0);
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(| : bad)
    for (size_t e = 0; e < nE; e++) {
        const uint64_t off = a->entryOff[e], end = a->entryOff[e + 1];
        uint64_t sz = (uint64_t) nClusters[e] * sizeof(RecCluster), members = 0, sizes = 0;
        if (nClusters[e] > end - off) bad |= 1;
        else
            for (uint32_t c = 0; c < nClusters[e]; c++) sizes += clusterSize[off + c];
        for (uint64_t h = off; h < end; h++)
            if (clusterOfHit[h] != UINT32_MAX) {
                if (clusterOfHit[h] >= nClusters[e] || clusterOfHit[h] >= end - off || rankInCluster[h] >= clusterSize[off + clusterOfHit[h]]) bad |= 1;
                sz += sizeof(RecMember) + ((a->best[h]->cigarLen + 3u) & ~3u);
                members++;
            }
        if (members != sizes) bad |= 1;   // every member slot of every cluster is some hit's (cluster, rank)
        at[e + 1] = sz;
    }
    if (bad) return SD_EINVAL;
    for (size_t e = 0; e < nE; e++) at[e + 1] += at[e];
    *bytes = at[nE];
    char *o = (char *) out;
    if (!o) return SD_OK;
    if (at[nE] > cap) return SD_ENOMEM;
#pragma omp parallel
    {
        std::vector<uint32_t> start, order;
#pragma omp for schedule(dynamic, 16)
        for (size_t e = 0; e < nE; e++) {
            const uint64_t off = a->entryOff[e], end = a->entryOff[e + 1];
            const uint32_t nC = nClusters[e];
            if (nC == 0) continue;
            start.assign((size_t) nC + 1, 0);
            for (uint32_t c = 0; c < nC; c++) start[c + 1] = start[c] + clusterSize[off + c];
            order.assign(start[nC], 0);
            for (uint64_t h = off; h < end; h++)
                if (clusterOfHit[h] != UINT32_MAX) order[start[clusterOfHit[h]] + rankInCluster[h]] = (uint32_t) (h - off);
            uint64_t w = at[e];
            for (uint32_t c = 0; c < nC; c++) {
                const uint32_t m = clusterSize[off + c];
                RecCluster rc = {m, a->entryQSet[e], a->entryTSet[e], 0, pCO[off + c], pMH[off + c]};
                memcpy(o + w, &rc, sizeof(rc));
                w += sizeof(RecCluster);
                for (uint32_t x = 0; x < m; x++) {
                    const BestHit &b = *a->best[off + order[start[c] + x]];
                    const uint64_t need = sizeof(RecMember) + ((b.cigarLen + 3u) & ~3u);
                    RecMember rm;
                    memset(&rm, 0, sizeof(rm));
                    rm.q = b.q;
                    rm.t = b.t;
                    memcpy(rm.pval, b.pvalText, sizeof(rm.pval));
                    memcpy(rm.seqId, b.seqIdText, sizeof(rm.seqId));
                    memcpy(rm.eval, b.evalText, sizeof(rm.eval));
                    rm.qStart = b.qStart; rm.qEnd = b.qEnd; rm.qLen = b.qLen;
                    rm.tStart = b.tStart; rm.tEnd = b.tEnd; rm.tLen = b.tLen;
                    rm.cigarLen = b.cigarLen;
                    memcpy(o + w, &rm, sizeof(rm));
                    memset(o + w + sizeof(rm), 0, need - sizeof(rm));
                    if (b.cigarLen) memcpy(o + w + sizeof(rm), &a->tCigar[b.arena][b.cigarOff], b.cigarLen);
                    w += need;
                }
            }
        }
    }
    return SD_OK;
}

// A record buffer that came over the wire (RCCL / TCP gather) is checked before anything indexes with it: whole records, set and
// sequence indices inside the name / source tables the writer will use.
int sd_records_check(const void *records, uint64_t bytes, uint32_t nQSets, uint32_t nTSets, uint32_t nQ, uint32_t nT, uint64_t *nClusters,
                     uint64_t *nMembers) {
    if (bytes && !records) return SD_EINVAL;
    const char *p = (const char *) records, *endP = p + bytes;
    uint64_t nc = 0, nm = 0;
    while (p < endP) {
        RecCluster rc;
        if ((uint64_t) (endP - p) < sizeof(rc)) return SD_EINVAL;
        memcpy(&rc, p, sizeof(rc));
        p += sizeof(rc);
        if (rc.qSet >= nQSets || rc.tSet >= nTSets) return SD_EINVAL;
        nc++;
        for (uint32_t m = 0; m < rc.members; m++) {
            RecMember rm;
            if ((uint64_t) (endP - p) < sizeof(rm)) return SD_EINVAL;
            memcpy(&rm, p, sizeof(rm));
            p += sizeof(rm);
            const uint64_t padded = ((uint64_t) rm.cigarLen + 3u) & ~3ull;
            if (rm.q >= nQ || rm.t >= nT || (uint64_t) (endP - p) < padded) return SD_EINVAL;
            p += padded;
            nm++;
        }
    }
    if (nClusters) *nClusters = nc;
    if (nMembers) *nMembers = nm;
    return SD_OK;
}

// summarizeresults (R/src/util/SummarizeResults.cpp:77-112) from cluster records: '#key source source pCO pMH size' per
// cluster, '>query target pval seqId eval coordinates cigar' per member; canonical: without the '#key' / '>query' columns
int sd_records_write_tsv(const void *records, uint64_t bytes, const char *path, int append, uint64_t firstClusterKey,
                         const char *qNames, const uint64_t *qNameOff, const char *tNames, const uint64_t *tNameOff,
                         const char *qSources, const uint64_t *qSourceOff, const char *tSources, const uint64_t *tSourceOff,
                         int canonical, uint64_t *nClusterLines, uint64_t *nHitLines) {
    if ((bytes && !records) || !path) return SD_EINVAL;
    FILE *f = fopen(path, append ? "a" : "w");
    if (!f) return SD_EINVAL;
    const char *p = (const char *) records, *endP = p + bytes;
    uint64_t key = firstClusterKey, nc = 0, nh = 0;
    int status = SD_OK;
    while (p < endP) {
        RecCluster rc;
        if ((uint64_t) (endP - p) < sizeof(rc)) { status = SD_EINVAL; break; }
        memcpy(&rc, p, sizeof(rc));
        p += sizeof(rc);
        char co[32], mh[32];
        snprintf(co, sizeof(co), "%.3E", rc.pCO);   // SSTR(double) = "{:.3E}" (M/src/commons/Util.cpp:658-660)
        snprintf(mh, sizeof(mh), "%.3E", rc.pMH);
        if (!canonical) fprintf(f, "#%llu\t", (unsigned long long) key);
        fprintf(f, "%.*s\t%.*s\t%s\t%s\t%u\n", (int) (qSourceOff[rc.qSet + 1] - qSourceOff[rc.qSet]), qSources + qSourceOff[rc.qSet],
                (int) (tSourceOff[rc.tSet + 1] - tSourceOff[rc.tSet]), tSources + tSourceOff[rc.tSet], co, mh, rc.members);
        nc++;
        for (uint32_t m = 0; m < rc.members && status == SD_OK; m++) {
            RecMember rm;
            if ((uint64_t) (endP - p) < sizeof(rm)) { status = SD_EINVAL; break; }
            memcpy(&rm, p, sizeof(rm));
            p += sizeof(rm);
            const uint64_t padded = (rm.cigarLen + 3u) & ~3u;
            if ((uint64_t) (endP - p) < padded) { status = SD_EINVAL; break; }
            if (!canonical) fprintf(f, ">%.*s\t", (int) (qNameOff[rm.q + 1] - qNameOff[rm.q]), qNames + qNameOff[rm.q]);
            fprintf(f, "%.*s\t%.16s\t%.8s\t%.16s\t%d\t%d\t%d\t%d\t%d\t%d\t%.*s\n", (int) (tNameOff[rm.t + 1] - tNameOff[rm.t]), tNames + tNameOff[rm.t],
                    rm.pval, rm.seqId, rm.eval, rm.qStart, rm.qEnd, rm.qLen, rm.tStart, rm.tEnd, rm.tLen, (int) rm.cigarLen, p);
            p += padded;
            nh++;
        }
        if (status != SD_OK) break;
        key++;
    }
    fclose(f);
    if (nClusterLines) *nClusterLines = nc;
    if (nHitLines) *nHitLines = nh;
    return status;
}

int sd_agg_write_tsv_from(sd_agg *a, const char *path, int append, uint64_t firstClusterKey, const uint32_t *clusterOfHit,
                          const uint32_t *rankInCluster, const uint32_t *nClusters, const double *pCO, const double *pMH,
                          const uint32_t *clusterSize, const char *qNames, const uint64_t *qNameOff, const char *tNames,
                          const uint64_t *tNameOff, const char *qSources, const uint64_t *qSourceOff, const char *tSources,
                          const uint64_t *tSourceOff, int canonical, uint64_t *nClusterLines, uint64_t *nHitLines) {
    if (!a || !path) return SD_EINVAL;
    // one path to the file: the records, then the text
    uint64_t bytes = 0;
    int rc = sd_agg_records(a, clusterOfHit, rankInCluster, nClusters, pCO, pMH, clusterSize, nullptr, 0, &bytes);
    if (rc != SD_OK) return rc;
    std::vector<char> rec(bytes);
    rc = sd_agg_records(a, clusterOfHit, rankInCluster, nClusters, pCO, pMH, clusterSize, rec.data(), bytes, &bytes);
    if (rc != SD_OK) return rc;
    return sd_records_write_tsv(rec.data(), bytes, path, append, firstClusterKey, qNames, qNameOff, tNames, tNameOff, qSources, qSourceOff,
                                tSources, tSourceOff, canonical, nClusterLines, nHitLines);
}

int sd_agg_write_tsv(sd_agg *a, const char *path, const uint32_t *clusterOfHit, const uint32_t *rankInCluster,
                     const uint32_t *nClusters, const double *pCO, const double *pMH, const uint32_t *clusterSize,
                     const char *qNames, const uint64_t *qNameOff, const char *tNames, const uint64_t *tNameOff,
                     const char *qSources, const uint64_t *qSourceOff, const char *tSources, const uint64_t *tSourceOff,
                     int canonical, uint64_t *nClusterLines, uint64_t *nHitLines) {
    return sd_agg_write_tsv_from(a, path, 0, 0, clusterOfHit, rankInCluster, nClusters, pCO, pMH, clusterSize, qNames, qNameOff,
                                 tNames, tNameOff, qSources, qSourceOff, tSources, tSourceOff, canonical, nClusterLines, nHitLines);
}

}  // extern "C"
