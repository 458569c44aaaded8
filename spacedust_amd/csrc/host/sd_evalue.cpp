// E-value / bit score for blosum62 with gap open 11, extend 1 -- the preset Gumbel parameters
// of M/src/alignment/EvalueComputation.h:64-69 evaluated with the finite-size-corrected "area"
// of M/lib/alp/sls_pvalues.cpp:366-520 (get_appr_tail_prob_with_cov_without_errors with
// blast_=false, compute_only_area_=true) and evaluePerArea = K exp(-lambda s)
// (M/lib/alp/sls_alignment_evaluer.hpp:154-157).  Plus the small text helpers of Matcher/Util.
#include "sd_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#if defined(__AVX2__)
#include <immintrin.h>
#endif

namespace sd {

void initEvaluer(Evaluer &e, uint64_t dbResidues) {
    // {"blosum62.out", 11, 1, true, {lambda, K, a1, b1, a2, b2, alpha1, beta1, alpha2, beta2, sigma, tau}}
    e.lambda = 0.27359865037097330642;
    e.K = 0.044620920658722244834;
    const double a1 = 1.5938724404943873658, b1 = -19.959867650284412122;
    const double a2 = 1.5938724404943873658, b2 = -19.959867650284412122;
    const double alpha1 = 30.455610143099914211, beta1 = -622.28684628915891608;
    const double alpha2 = 30.455610143099914211, beta2 = -622.28684628915891608;
    e.sigma = 29.602444874818868215;
    e.tau = -601.81087985041381216;
    // sls_alignment_evaluer.cpp:679-721: a_I<-a2, a_J<-a1, alpha_I<-alpha2, alpha_J<-alpha1, b/beta likewise
    e.aI = a2; e.bI = b2; e.alphaI = alpha2; e.betaI = beta2;
    e.aJ = a1; e.bJ = b1; e.alphaJ = alpha1; e.betaJ = beta1;
    // sls_pvalues.cpp:343-361 (nat_cut_off_in_max = 2.0)
    e.viThr = std::max(2.0 * e.alphaI / e.lambda, 0.0);
    e.vjThr = std::max(2.0 * e.alphaJ / e.lambda, 0.0);
    e.cThr = std::max(2.0 * e.sigma / e.lambda, 0.0);
    e.logK = std::log(e.K);
    e.dbResidues = (double) dbResidues;
}

static inline double normalProbability(double x) { return 0.5 * erfc(-sqrt(0.5) * x); }   // sls_basic.hpp:195-198

// Finite-size corrected search-space area for a score (ALP's Gumbel area approximation as AlignmentEvaluer uses it, with
// compute_only_area_; the database side is seqlen2).  For each of the two sequences: the length left after the expected
// edge loss (linear in the score), divided by the standard deviation of that loss (variance linear in the score, floored),
// gives a normal quantile z; the side contributes  left * Phi(z) + sd * phi(z).  The area is the product of the two
// contributions plus a covariance term  max(cThr, sigma * score + tau) * Phi(z1) * Phi(z2).
// The operations keep the reference's association and order (its AVX2 build contracts a*b+c into FMA, and the printed
// %.3E digits depend on it): one side after the other, products formed before the differences that use them.
namespace {
struct AreaSide {
    double phiCdf;   // Phi(z)
    double part;     // left * Phi(z) + sd * phi(z)
};
inline AreaSide areaSide(double score, double length, double lossSlope, double lossOffset, double varSlope, double varOffset,
                         double varFloor) {
    static const double kPi = 3.1415926535897932384626433832795;
    static const double kInvSqrt2Pi = 1 / sqrt(2.0 * kPi);
    const double loss = lossSlope * score + lossOffset;
    const double left = length - loss;
    const double variance = std::max(varFloor, varSlope * score + varOffset);
    const double sd = sqrt(variance);
    double z;
    if (sd == 0.0) z = 1e100;
    else z = left / sd;
    AreaSide r;
    r.phiCdf = normalProbability(z);
    const double negDensity = -kInvSqrt2Pi * exp(-0.5 * z * z);
    const double leftTerm = left * r.phiCdf;
    const double sdTerm = sd * negDensity;
    r.part = leftTerm - sdTerm;
    return r;
}
}  // namespace

static double area(const Evaluer &e, double score, double queryLen, double dbLen) {
    const AreaSide dbSide = areaSide(score, dbLen, e.aI, e.bI, e.alphaI, e.betaI, e.viThr);
    const AreaSide querySide = areaSide(score, queryLen, e.aJ, e.bJ, e.alphaJ, e.betaJ, e.vjThr);
    const double covariance = std::max(e.cThr, e.sigma * score + e.tau);
    const double bothInside = dbSide.phiCdf * querySide.phiCdf;
    const double covTerm = covariance * bothInside;
    const double product = dbSide.part * querySide.part;
    return product + covTerm;
}

double computeEvalue(const Evaluer &e, double score, double qLen) {
    const double epa = e.K * exp(-e.lambda * score);
    const double a = area(e, score, qLen, e.dbResidues);
    return epa * a;
}

double computeBitScore(const Evaluer &e, double score) { return (e.lambda * score - e.logK) / log(2.0); }

bool canBeCovered(float covThr, int covMode, float queryLength, float targetLength) {
    switch (covMode) {
        case 0: return ((queryLength / targetLength >= covThr) && (targetLength / queryLength >= covThr));
        case 2: return ((targetLength / queryLength) >= covThr);
        case 1: return ((queryLength / targetLength) >= covThr);
        case 3: return ((targetLength / queryLength) >= covThr) && (targetLength / queryLength) <= 1.0;
        case 4: return ((queryLength / targetLength) >= covThr) && (queryLength / targetLength) <= 1.0;
        case 5: return (std::min(targetLength, queryLength) / std::max(targetLength, queryLength)) >= covThr;
        default: return true;
    }
}

bool hasCoverage(float covThr, int covMode, float queryCov, float targetCov) {
    switch (covMode) {
        case 0: return ((queryCov >= covThr) && (targetCov >= covThr));
        case 2: return (queryCov >= covThr);
        case 1: return (targetCov >= covThr);
        default: return true;
    }
}

float computeCov(unsigned startPos, unsigned endPos, unsigned len) {
    return (std::min(len, std::max(startPos, endPos)) - std::min(startPos, endPos) + 1) / (float) len;
}

char *u32toa(uint32_t v, char *buf) {
    char tmp[12];
    int n = 0;
    do {
        tmp[n++] = (char) ('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *buf++ = tmp[--n];
    return buf;
}

char *i32toa(int32_t v, char *buf) {
    if (v < 0) {
        *buf++ = '-';
        return u32toa((uint32_t) (-(int64_t) v), buf);
    }
    return u32toa((uint32_t) v, buf);
}

char *seqIdToBuffer(float seqId, char *buffer) {
    // Util::fastSeqIdToBuffer (M/src/commons/Util.cpp:222-251): truncation, not rounding.  For seqId == 1.0 the
    // reference writes "1.000" but returns the position OF the terminator, not one past it as its integer branch does,
    // so the separator every caller stores at ret[-1] (Matcher.cpp:286-287) lands on the last zero: the text every
    // alignment DB carries for an identity of one is "1.00" (the reference binary's `result` DB: 6 229 such lines on
    // config 1).
    if (seqId == 1.0) {
        memcpy(buffer, "1.00", 4);
        return buffer + 4;
    }
    *buffer++ = '0';
    *buffer++ = '.';
    if (seqId < 0.10) *buffer++ = '0';
    if (seqId < 0.01) *buffer++ = '0';
    return i32toa((int) (seqId * 1000), buffer);
}

std::string compressBacktrace(const char *bt, size_t n) {
    std::string ret;
    compressBacktraceAppend(bt, n, ret);
    return ret;
}

namespace {
struct Dec3 {
    struct { char d[4]; } t[1000];   // the decimal digits of 0 .. 999, left-aligned, and their number in d[3]
    Dec3() {
        for (int v = 0; v < 1000; v++) {
            char b[8];
            const int len = (int) (u32toa((uint32_t) v, b) - b);
            memset(t[v].d, '0', 3);
            memcpy(t[v].d, b, (size_t) len);
            t[v].d[3] = (char) len;
        }
    }
};
const Dec3 kDec3;
}  // namespace

// Matcher::compressAlignment (M/src/alignment/Matcher.cpp:166-185): run lengths, starting in state 'M' with a count of 0 (a
// backtrace that does not begin with a match begins "0M").  Run boundaries are found 32 letters at a time (the letters against their
// left neighbours, one compare + movemask):
// a protein alignment is mostly match runs of dozens of letters, and this loop was 60 % of the aggregation stage's CPU time.
void compressBacktraceAppend(const char *bt, size_t n, std::string &ret) {
    char state = 'M';
    size_t runStart = 0, i = 0;
    char out[256], *o = out;   // runs are written here and appended in pieces (one append per run was the rest of the cost)
    auto emit = [&](size_t count) {
        if (o > out + sizeof(out) - 16) {
            ret.append(out, (size_t) (o - out));
            o = out;
        }
        if (count < 1000) {   // no branch on the number of digits (run lengths are as good as random: every branch here mispredicts)
            memcpy(o, kDec3.t[count].d, 4);
            o += kDec3.t[count].d[3];
        } else {
            o = u32toa((uint32_t) count, o);
        }
        *o++ = state;
    };
    if (n && bt[0] != 'M') {
        emit(0);
        state = bt[0];
    }
    auto boundary = [&](size_t pos) {
        emit(pos - runStart);
        state = bt[pos];
        runStart = pos;
    };
    i = 1;
#if defined(__AVX2__)   // (the build's flags are the reference's AVX2 flags; a host without AVX2 takes the letter-by-letter loop below)
    for (; i + 32 <= n; i += 32) {   // letters i .. i + 31 against their left neighbours
        const __m256i cur = _mm256_loadu_si256((const __m256i *) (bt + i));
        const __m256i left = _mm256_loadu_si256((const __m256i *) (bt + i - 1));
        uint32_t ne = ~(uint32_t) _mm256_movemask_epi8(_mm256_cmpeq_epi8(cur, left));
        while (ne) {
            boundary(i + (size_t) __builtin_ctz(ne));
            ne &= ne - 1;
        }
    }
#endif
    for (; i < n; i++)
        if (bt[i] != bt[i - 1]) boundary(i);
    emit(n - runStart);
    ret.append(out, (size_t) (o - out));
}

}  // namespace sd
