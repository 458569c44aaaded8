// result2profile: the host step between the iterations of `search --num-iterations` -- alignment DB ->
// profile DB.  Restated from the reference's semantics, float operation by float operation, because the next iteration
// consumes the rounded int8 scores:
//   MSA from the backtraces, query-gap-free      M/src/alignment/MultipleAlignment.cpp:46-215 (computeMSA, noDeletionMSA)
//   redundancy filter (HH-suite style)           M/src/alignment/MsaFilter.cpp:85-530
//   sequence weights, context specific weights,  M/src/alignment/PSSMCalculator.cpp:169-240 (computePSSMFromMSA), :305-372,
//   Neff, pseudo counts, log-odds PSSM              :374-401, :420-588, :251-265
//   global composition bias of the PSSM          M/src/commons/SubstitutionMatrix.cpp:205-243
//   tantan masking of the query positions        M/src/commons/Masker.cpp:57-80
//   25-byte profile record                       PSSMCalculator.cpp:671-687, M/src/commons/Sequence.h:458-471
// Compiled by g++ with the reference's AVX2 flags (-mavx2 -mfma, GCC's -ffp-contract=fast): the mixed float/double
// expressions below keep the reference's literal types and grouping so that they contract the same way, and the one
// approximate instruction the reference uses (rcpps + one Newton step, PSSMCalculator.cpp:497-504) is used here too.
#include "sd_host.h"
#include "spacedust_gpu.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <string>
#include <vector>
#include <omp.h>

namespace {

enum { AA = 20, ANY = 20, NAA = 20, GAP = 21, ENDGAP = 22 };
constexpr int BLK = 32;   // VECSIZE_INT * 4 of the AVX2 build: MsaFilter compares rows in 32-byte blocks

#define SD_MAY_ALIAS(x) x __attribute__((__may_alias__))

// MathUtil::flog2 / fpow2 (M/src/commons/MathUtil.h:121-163): literal types as in the reference
inline float flog2(float x) {
    if (x <= 0) return -128;
    SD_MAY_ALIAS(int) *px = (int *) (&x);
    float e = (float) (((*px & 0x7F800000) >> 23) - 0x7f);
    *px = ((*px & 0x007FFFFF) | 0x3f800000);
    x -= 1.0;
    x *= (1.441740 + x * (-0.7077702 + x * (0.4123442 + x * (-0.1903190 + x * 0.0440047))));
    return x + e;
}

inline double fpow2(float x) {
    if (x >= FLT_MAX_EXP) return FLT_MAX;
    if (x <= FLT_MIN_EXP) return 0.0f;
    SD_MAY_ALIAS(int) *px = (int *) (&x);
    float tx = (x - 0.5f) + (3 << 22);
    SD_MAY_ALIAS(int) *ix = (int *) (&tx);
    int lx = *ix - 0x4b400000;
    float dx = x - (float) (lx);
    x = 1.0f + dx * (0.693019f + dx * (0.241404f + dx * (0.0520749f + dx * 0.0134929f)));
    *px += (lx << 23);
    return x;
}

inline float normalizeTo1(float *array, int length, const double *def_array = NULL) {   // MathUtil::NormalizeTo1
    float sum = 0.0f;
    for (int k = 0; k < length; k++) sum += array[k];
    if (sum != 0.0f) {
        float fac = 1.0 / sum;
        for (int i = 0; i < length; i++) array[i] *= fac;
    } else if (def_array) {
        for (int i = 0; i < length; i++) array[i] = def_array[i];
    }
    return sum;
}

inline unsigned char neffToChar(const float neff) {   // MathUtil::convertNeffToChar
    float retVal = std::min(255.0f, 1.0f + 64.0f * flog2(neff));
    return std::max(static_cast<unsigned char>(1), static_cast<unsigned char>(retVal + 0.5));
}

// ScalarProd20 (M/lib/simd/simd.h:901-953): five 4-lane products, pairwise adds, two shuffle-add rounds
inline float scalarProd20(const float *qi, const float *tj) {
    float __attribute__((aligned(16))) res;
    __m128 P, R;
    const __m128 *Qi = (const __m128 *) qi;
    const __m128 *Tj = (const __m128 *) tj;
    __m128 P1 = _mm_mul_ps(*(Qi), *(Tj));
    __m128 P2 = _mm_mul_ps(*(Qi + 1), *(Tj + 1));
    __m128 R1 = _mm_add_ps(P1, P2);
    __m128 P3 = _mm_mul_ps(*(Qi + 2), *(Tj + 2));
    __m128 P4 = _mm_mul_ps(*(Qi + 3), *(Tj + 3));
    __m128 R2 = _mm_add_ps(P3, P4);
    __m128 P5 = _mm_mul_ps(*(Qi + 4), *(Tj + 4));
    R = _mm_add_ps(R1, R2);
    R = _mm_add_ps(R, P5);
    P = _mm_shuffle_ps(R, R, _MM_SHUFFLE(2, 0, 2, 0));
    R = _mm_shuffle_ps(R, R, _MM_SHUFFLE(3, 1, 3, 1));
    R = _mm_add_ps(R, P);
    P = _mm_shuffle_ps(R, R, _MM_SHUFFLE(2, 0, 2, 0));
    R = _mm_shuffle_ps(R, R, _MM_SHUFFLE(3, 1, 3, 1));
    R = _mm_add_ps(R, P);
    _mm_store_ss(&res, R);
    return res;
}

// one worker's scratch space
struct Work {
    // MSA rows: stride padded to 32-byte blocks with at least one block of GAP after the last column
    std::vector<char> msa;
    std::vector<const char *> rows, X;
    size_t stride = 0;
    std::vector<unsigned> queryGaps;
    // filter
    std::vector<char> keep, in, inkk, display;
    std::vector<char *> keepLocal;
    std::vector<int> seqidPrev, first, last, nres, ksort, N, Nmax, idmaxwin;
    // PSSM
    std::vector<float> seqWeight, wi, matchWeight, pseudo, profile, neffM;
    std::vector<char> pssm;
    std::vector<unsigned char> consensus, masked;
    std::vector<int> nseqs, naa;
    std::vector<float> wContrib;   // [L][24]
    std::vector<int> n;            // [L + 1][24]
    std::vector<float> f;          // [L][23]
    std::vector<float> pNull;
};

// ---- MultipleAlignment::computeMSA with noDeletionMSA = true: every row has exactly centerL columns -----------------
void buildMsa(Work &w, const uint8_t *center, int centerL, size_t nEdges, const uint8_t *const *edgeSeq, const int32_t *qStart,
              const int32_t *tStart, const char *const *bt, const uint32_t *btLen) {
    const size_t setSize = nEdges + 1;
    w.stride = ((size_t) (centerL + 1) / BLK + 2) * BLK;   // MultipleAlignment::initX(centerSeq->L + 1, ...)
    w.msa.assign(w.stride * setSize, (char) GAP);
    w.rows.resize(setSize);
    for (size_t k = 0; k < setSize; k++) w.rows[k] = w.msa.data() + k * w.stride;
    // (computeQueryGaps only matters when deletions are kept; with noDeletionMSA the gap counts are never written out)
    char *row0 = w.msa.data();
    for (int p = 0; p < centerL; p++) row0[p] = (char) center[p];
    for (size_t e = 0; e < nEdges; e++) {
        char *row = w.msa.data() + (e + 1) * w.stride;
        const uint8_t *seq = edgeSeq[e];
        unsigned queryPos = (unsigned) qStart[e], targetPos = (unsigned) tStart[e];
        size_t bufferPos = 0;
        if (targetPos == 0xFFFFFFFFu) continue;   // row stays all gaps (MultipleAlignment.cpp:112-119)
        for (int p = 0; p < qStart[e]; p++) row[bufferPos++] = (char) GAP;
        const char *b = bt[e];
        const size_t n = btLen[e];
        for (size_t a = 0; a < n; a++) {
            if (b[a] == 'I') {
                row[bufferPos++] = (char) GAP;
                queryPos++;
            } else if (b[a] == 'D') {
                while (a < n && b[a] == 'D') {   // target residues against a query gap are dropped
                    targetPos++;
                    a++;
                }
                if (a >= n) break;
                if (b[a] == 'I') {
                    row[bufferPos++] = (char) GAP;
                    queryPos++;
                } else if (b[a] == 'M') {
                    row[bufferPos++] = (char) seq[targetPos];
                    queryPos++;
                    targetPos++;
                }
            } else if (b[a] == 'M') {
                row[bufferPos++] = (char) seq[targetPos];
                queryPos++;
                targetPos++;
            }
        }
        // the rest of the row is GAP already
    }
    // numeric residues: the reference maps letters back through aa2num; residues here are numeric from the start and X = 20
}

// ---- MsaFilter::filter (single qid bucket is the common case; the bucketed variant follows the same code) -----------
size_t filterMsa(Work &w, const int8_t *subMatrix /* 21x21 of the -0.2 biased blosum62 */, int N_in_total, int L, int coverage,
                 const std::vector<int> &qid_vec, float qsc, int max_seqid, int Ndiff, int filterMinEnable) {
    const float PLTY_GAPOPEN = 6.0f, PLTY_GAPEXTD = 1.0f;
    std::vector<const char *> &X_in = w.rows;
    w.keep.assign(N_in_total, 0);
    w.in.assign(N_in_total + 1, 0);
    w.inkk.assign(N_in_total + 1, 0);
    w.display.assign(N_in_total + 2, 0);
    w.seqidPrev.assign(N_in_total + 1, 0);
    w.first.assign(N_in_total, 0);
    w.last.assign(N_in_total, 0);
    w.nres.assign(N_in_total, 0);
    w.ksort.assign(N_in_total, 0);
    w.X.assign(N_in_total, nullptr);
    w.keepLocal.assign(N_in_total, nullptr);
    w.N.assign(L + 2, 0);
    w.Nmax.assign(L + 2, 0);
    w.idmaxwin.assign(L + 2, 0);
    char *keep = w.keep.data(), *in = w.in.data(), *inkk = w.inkk.data();
    int *first = w.first.data(), *last = w.last.data(), *nres = w.nres.data(), *ksort = w.ksort.data();
    int *N = w.N.data(), *Nmax = w.Nmax.data(), *idmaxwin = w.idmaxwin.data(), *seqid_prev = w.seqidPrev.data();
    const char **X = w.X.data();
    char **keep_local = w.keepLocal.data();
    int N_keep_total = 0;
    for (size_t qid_idx = 0; qid_idx < qid_vec.size(); qid_idx++) {
        int n = 0;
        int N_in_bucket = 0;
        int qid;
        if (qid_vec.size() == 1) {
            if (N_in_total < filterMinEnable) {
                memset(keep, 1, N_in_total * sizeof(char));
                keep[0] = 2;
                N_keep_total = N_in_total - 1;
                break;
            }
            qid = qid_vec[0];
            N_in_bucket = N_in_total;
            for (int k = 0; k < N_in_total; k++) {
                X[k] = X_in[k];
                keep_local[k] = &keep[k];
            }
        } else {
            if (qid_idx == qid_vec.size() - 1) break;
            qid = 0;
            X[0] = X_in[0];
            keep_local[0] = &keep[0];
            const char *query = X_in[0];
            N_in_bucket++;
            for (int k = 1; k < N_in_total; k++) {
                int nr = 0, nid = 0;
                for (int i = 0; i < L; ++i) {
                    nr += (X_in[k][i] < NAA);
                    nid += (X_in[k][i] == query[i] && X_in[k][i] < NAA);
                }
                int seqid = static_cast<int>(100.0f * (static_cast<float>(nid) / static_cast<float>(nr)));
                if (seqid > qid_vec[qid_idx] && seqid <= qid_vec[qid_idx + 1]) {
                    X[N_in_bucket] = X_in[k];
                    keep_local[N_in_bucket] = &keep[k];
                    N_in_bucket++;
                }
            }
            if (N_in_bucket < filterMinEnable) {
                for (int k = 1; k < N_in_bucket; k++) *keep_local[k] = 1;
                *keep_local[0] = 2;
                N_keep_total += N_in_bucket - 1;
                continue;
            }
        }
        int N_in = N_in_bucket;
        int seqid1 = 20;
        const int WFIL = 25;
        int diffNmax = Ndiff, diffNmax_prev = 0;
        int seqid, seqid_step = 0;
        float diff_min_frac;
        float qdiff_max_frac = 0.9999 - 0.01 * qid;
        int diff = 0, diff_suff, qdiff_max, cov_kj, first_kj, last_kj;
        int kk, jj, k, j, i;
        int kfirst = 0;
        for (k = 0; k < N_in; ++k) *keep_local[k] = (k == 0) ? 2 : 1;
        for (n = k = 0; k < N_in; ++k) {
            if (*keep_local[k] == 2) {
                in[k] = 2;
                n++;
            } else {
                in[k] = 0;
            }
        }
        for (k = 0; k < N_in; ++k) {
            for (i = 0; i < L; ++i)
                if (X[k][i] < NAA) break;
            first[k] = i;
            for (i = (L - 1); i > 0; i--)
                if (X[k][i] < NAA) break;
            last[k] = i;
        }
        for (k = 0; k < N_in; ++k) {
            int nr = 0;
            for (i = first[k]; i <= last[k]; ++i)
                if (X[k][i] < NAA) nr++;
            nres[k] = nr;
            if (nr == 0) *keep_local[k] = 0;
        }
        {
            std::vector<std::pair<int, int> > tmpSort(N_in);
            for (k = 0; k < N_in; ++k) {
                tmpSort[k].first = nres[k];
                tmpSort[k].second = k;
            }
            std::stable_sort(tmpSort.begin() + 1, tmpSort.end(),
                             [](const std::pair<int, int> &l, const std::pair<int, int> &r) { return l.first > r.first; });
            for (k = 0; k < N_in; ++k) ksort[k] = tmpSort[k].second;
        }
        for (kk = 0; kk < N_in; ++kk) inkk[kk] = in[ksort[kk]];
        for (i = 0; i < first[kfirst]; ++i) N[i] = 0;
        for (i = first[kfirst]; i <= last[kfirst]; ++i) N[i] = 1;
        for (i = last[kfirst] + 1; i < L; ++i) N[i] = 0;
        for (i = 0; i < L; ++i) {
            Nmax[i] = 0;
            idmaxwin[i] = -1;
        }
        for (k = 0; k < N_in; ++k) seqid_prev[k] = -1;
        if (Ndiff <= 0 || Ndiff >= N_in) {
            seqid1 = max_seqid;
            Ndiff = N_in;
            diffNmax = Ndiff;
        }
        for (k = 0; k < N_in; ++k) {
            if (*keep_local[k] == 0 || *keep_local[k] == 2) continue;
            if (100 * nres[k] < coverage * L) {
                *keep_local[k] = 0;
                continue;
            }
            float qsc_sum = 0.0;
            if (qsc > -10) {
                float qsc_min = qsc * nres[k];
                int gapq = 0, gapk = 0;
                for (int i2 = first[k]; i2 <= last[k]; ++i2) {
                    if (X[k][i2] < 20) {
                        gapk = 0;
                        if (X[kfirst][i2] < 20) {
                            gapq = 0;
                            qsc_sum += static_cast<float>(subMatrix[(int) X[kfirst][i2] * 21 + (int) X[k][i2]]);
                        } else if (X[kfirst][i2] == ANY)
                            continue;
                        else if (gapq++)
                            qsc_sum -= PLTY_GAPEXTD;
                        else
                            qsc_sum -= PLTY_GAPOPEN;
                    } else if (X[k][i2] == ANY)
                        continue;
                    else if (X[kfirst][i2] < 20) {
                        gapq = 0;
                        if (gapk++) qsc_sum -= PLTY_GAPEXTD;
                        else qsc_sum -= PLTY_GAPOPEN;
                    }
                }
                if (qsc_sum < qsc_min) {
                    *keep_local[k] = 0;
                    continue;
                }
            }
            if (qdiff_max_frac < 0.999) {
                qdiff_max = int(qdiff_max_frac * nres[k] + 0.9999);
                diff = 0;
                for (int i2 = first[k]; i2 <= last[k]; ++i2)
                    if (X[k][i2] < NAA && X[k][i2] != X[kfirst][i2] && ++diff >= qdiff_max) break;
                if (diff >= qdiff_max) {
                    *keep_local[k] = 0;
                    continue;
                }
            }
        }
        int nn = 0;
        for (k = 0; k < N_in; ++k)
            if (*keep_local[k] > 0) nn++;
        if (nn == 0) {   // unreachable while the query row is marked 2; kept for the reference's control flow
            for (k = 0; k < N_in; k++) {
                if (w.display[k] != 2) {
                    *keep_local[k] = 1;
                    break;
                }
            }
        }
        if (seqid1 > max_seqid) {
            N_keep_total += nn;
            continue;
        }
        seqid = seqid1;
        while (seqid <= max_seqid) {
            bool stop = true;
            diffNmax_prev = diffNmax;
            diffNmax = 0;
            for (i = 0; i < L; ++i) {
                int max = 0;
                for (j = std::max(0, std::min(L - 2 * WFIL + 1, i - WFIL)); j < std::min(L, std::max(2 * WFIL, i + WFIL)); ++j)
                    if (N[j] > max) max = N[j];
                if (Nmax[i] < max) Nmax[i] = max;
                if (Nmax[i] < Ndiff) {
                    stop = false;
                    idmaxwin[i] = seqid;
                    if (diffNmax < Ndiff - Nmax[i]) diffNmax = Ndiff - Nmax[i];
                }
            }
            if (stop) break;
            for (kk = 0; kk < N_in; ++kk) {
                if (inkk[kk]) continue;
                k = ksort[kk];
                if (!(*keep_local[k])) continue;
                if (*keep_local[k] == 2) {
                    inkk[kk] = 2;
                    continue;
                }
                if (seqid >= 100) {
                    in[k] = inkk[kk] = 1;
                    n++;
                    continue;
                }
                float seqidk = seqid1;
                for (i = first[k]; i <= last[k]; ++i)
                    if (idmaxwin[i] > seqidk) seqidk = idmaxwin[i];
                if (seqid == seqid_prev[k]) continue;
                seqid_prev[k] = seqid;
                diff_min_frac = 0.9999 - 0.01 * seqidk;
                for (jj = 0; jj < kk; ++jj) {
                    if (!inkk[jj]) continue;
                    j = ksort[jj];
                    first_kj = std::max(first[k], first[j]);
                    last_kj = std::min(last[k], last[j]);
                    cov_kj = last_kj - first_kj + 1;
                    diff_suff = int(diff_min_frac * std::min(nres[k], cov_kj) + 0.999);
                    diff = 0;
                    // the reference walks 32-byte blocks (AVX2): whole blocks count, the loop leaves at block borders
                    const int first_blk = first_kj / BLK;
                    const int last_blk = last_kj / BLK + 1;
                    const int first_diff = std::abs(first_blk * BLK - first_kj);
                    const int last_diff = std::abs(last_blk * BLK - (last_kj + 1));
                    cov_kj += (first_diff + last_diff);
                    const char *xk = X[k], *xj = X[j];
                    for (int b = first_blk; b < last_blk && diff < diff_suff; ++b) {
                        int noAa = 0, differ = 0;
                        for (int u = b * BLK; u < (b + 1) * BLK; u++) {
                            const bool gapOr = (xk[u] > (NAA - 1)) || (xj[u] > (NAA - 1));
                            noAa += gapOr;
                            differ += !(gapOr || xk[u] == xj[u]);
                        }
                        cov_kj -= noAa;
                        diff += differ;
                    }
                    if (diff < diff_suff && float(diff) <= diff_min_frac * cov_kj && cov_kj > 0) break;
                }
                if (jj >= kk) {
                    in[k] = inkk[kk] = 1;
                    n++;
                    for (i = first[k]; i <= last[k]; ++i) N[i]++;
                }
            }
            seqid_step = std::max(1, std::min(5, diffNmax / (diffNmax_prev - diffNmax + 1) * seqid_step / 2));
            seqid += seqid_step;
        }
        for (k = 0; k < N_in; ++k) *keep_local[k] = in[k];
        N_keep_total += n - 1;
    }
    // shuffleSequences: kept rows move to the front, order preserved
    for (int i = 0, j = 0; j < N_in_total; j++) {
        if (keep[j] != 0) {
            if (i < j) std::swap(X_in[i], X_in[j]);
            i++;
        }
    }
    return (size_t) N_keep_total + 1;
}

// ---- PSSMCalculator ---------------------------------------------------------------------------------------------------
void sequenceWeights(float *seqWeight, size_t L, size_t setSize, const char *const *msa) {
    std::vector<unsigned> number_res(setSize);
    std::fill(seqWeight, seqWeight + setSize, 1e-6);
    for (size_t k = 0; k < setSize; ++k) {
        unsigned nr = 0;
        for (size_t pos = 0; pos < L; pos++)
            if (msa[k][pos] != GAP) nr++;
        number_res[k] = nr;
    }
    for (size_t pos = 0; pos < L; pos++) {
        int nl[AA];
        std::fill(nl, nl + AA, 0);
        for (size_t k = 0; k < setSize; ++k) {
            if (msa[k][pos] != GAP) {
                const unsigned aa_pos = (unsigned char) msa[k][pos];
                if (aa_pos < AA) nl[aa_pos]++;
            }
        }
        int distinct = 0;
        for (size_t aa = 0; aa < AA; ++aa)
            if (nl[aa]) ++distinct;
        for (size_t k = 0; k < setSize; ++k) {
            if (msa[k][pos] != GAP && distinct != 0) {
                const unsigned aa_pos = (unsigned char) msa[k][pos];
                if (aa_pos < AA) seqWeight[k] += 1.0f / (float(nl[aa_pos]) * float(distinct) * (float(number_res[k]) + 30.0f));
            }
        }
    }
}

void contextSpecificWeights(Work &w, const sd::SubMat &m, float *matchWeight, const float *wg, float *Neff_M, size_t L, size_t setSize,
                            char *const *X) {
    const float MAXENDGAPFRAC = 0.1;
    const int NCOLMIN = 20;
    constexpr int NW = 24;   // (NAA + 3) rounded up to a multiple of 8 floats
    int nseqi = 0;
    w.n.assign((L + 1) * NW, 0);
    w.wContrib.assign((L + 1) * NW, 0.0f);
    w.f.assign((L + 1) * (NAA + 3), 0.0f);
    w.naa.assign(L + 1, 0);
    w.nseqs.assign(L + 1, 0);
    float *wi = w.wi.data();
    int *n = w.n.data();
    float *wc = w.wContrib.data();
    float *f = w.f.data();
    for (size_t k = 0; k < setSize; ++k) {
        for (size_t i = 0; i < L && X[k][i] == GAP; ++i) X[k][i] = ENDGAP;
        for (int i = (int) L - 1; i >= 0 && X[k][i] == GAP; i--) X[k][i] = ENDGAP;
    }
    for (size_t i = 0; i < L; i++) {
        bool change = false;
        for (size_t k = 0; k < setSize; ++k) {
            if ((i == 0 && X[k][i] < ANY) || (i != 0 && X[k][i - 1] >= ANY && X[k][i] < ANY)) {
                change = true;
                nseqi++;
                for (size_t j = 0; j < L; ++j) n[j * NW + (int) X[k][j]]++;
            } else if (i != 0 && X[k][i - 1] < ANY && X[k][i] >= ANY) {
                change = true;
                nseqi--;
                for (size_t j = 0; j < L; ++j) n[j * NW + (int) X[k][j]]--;
            }
        }
        w.nseqs[i] = nseqi;
        if (change) {
            int ncol = 0;
            for (size_t k = 0; k < setSize; ++k) wi[k] = 1E-8;
            int jmin, jmax;
            for (jmin = 0; jmin < static_cast<int>(L) && n[jmin * NW + ENDGAP] > MAXENDGAPFRAC * nseqi; ++jmin) {
            }
            for (jmax = (int) L - 1; jmax >= 0 && n[jmax * NW + ENDGAP] > MAXENDGAPFRAC * nseqi; --jmax) {
            }
            ncol = jmax - jmin + 1;
            if (ncol < NCOLMIN) {
                for (size_t k = 0; k < setSize; ++k) wi[k] = (X[k][i] < ANY) ? wg[k] : 0.0f;
            } else {
                for (int j = jmin; j <= jmax; ++j) {
                    w.naa[j] = 0;
                    for (int a = 0; a < ANY; ++a) w.naa[j] += (n[j * NW + a] ? 1 : 0);
                }
                for (int j = jmin; j <= jmax; ++j) {
                    // w_contrib[j][a] = 1 / (naa[j] * n[j][a]) through rcpps and one Newton-Raphson step
                    // (PSSMCalculator.cpp:494-505): rcp + rcp - x * (rcp * rcp), eight amino acids at a time
                    const __m256 naa_j = _mm256_cvtepi32_ps(_mm256_set1_epi32(w.naa[j]));
                    const int aa_size = (ANY + 8 - 1) / 8;
                    for (int a = 0; a < aa_size; ++a) {
                        const __m256 nja = _mm256_cvtepi32_ps(_mm256_loadu_si256((const __m256i *) (n + j * NW + a * 8)));
                        const __m256 res = _mm256_mul_ps(nja, naa_j);
                        const __m256 rcp = _mm256_rcp_ps(res);
                        const __m256 mul = _mm256_mul_ps(res, _mm256_mul_ps(rcp, rcp));
                        _mm256_storeu_ps(wc + j * NW + a * 8, _mm256_sub_ps(_mm256_add_ps(rcp, rcp), mul));
                    }
                    for (int a = ANY; a < NAA + 3; ++a) wc[j * NW + a] = 0.0f;
                }
                for (size_t k = 0; k < setSize; ++k) {
                    if (X[k][i] >= ANY) continue;
                    for (int j = jmin; j <= jmax; ++j) wi[k] += wc[j * NW + (int) X[k][j]];
                }
            }
            Neff_M[i] = 0.0;
            for (int j = jmin; j <= jmax; ++j) memset(f + j * (NAA + 3), 0, ANY * sizeof(float));
            for (size_t k = 0; k < setSize; ++k) {
                if (X[k][i] >= ANY) continue;
                for (int j = jmin; j <= jmax; ++j) f[j * (NAA + 3) + (int) X[k][j]] += wi[k];
            }
            for (int j = jmin; j <= jmax; ++j) {
                normalizeTo1(f + j * (NAA + 3), NAA);
                for (int a = 0; a < 20; ++a)
                    if (f[j * (NAA + 3) + a] > 1E-10) Neff_M[i] -= f[j * (NAA + 3) + a] * flog2(f[j * (NAA + 3) + a]);
            }
            if (ncol > 0) Neff_M[i] = fpow2(Neff_M[i] / ncol);
            else Neff_M[i] = 1.0;
        } else {
            if (i == 0) Neff_M[i] = 0.0f;
            else Neff_M[i] = Neff_M[i - 1];
        }
        for (int a = 0; a < 20; ++a) matchWeight[i * AA + a] = 0.0;
        // (X[k][i] can be 20..22 here: the reference adds those weights to the slots after the 20 amino acids of this
        // position, i.e. to the first slots of the next position, which the next iteration zeroes again)
        for (size_t k = 0; k < setSize; ++k) matchWeight[i * AA + (int) X[k][i]] += wi[k];
        normalizeTo1(matchWeight + i * AA, NAA, m.pBack);
    }
    for (size_t k = 0; k < setSize; ++k) {
        for (size_t i = 0; i < L && X[k][i] == ENDGAP; ++i) X[k][i] = GAP;
        for (int i = (int) L - 1; i >= 0 && X[k][i] == ENDGAP; i--) X[k][i] = GAP;
    }
}

void matchWeights(const sd::SubMat &m, float *matchWeight, const float *seqWeight, size_t setSize, size_t L, const char *const *msa) {
    for (size_t pos = 0; pos < L; pos++) {
        memset(matchWeight + pos * AA, 0, AA * sizeof(float));
        for (size_t k = 0; k < setSize; ++k) {
            if (msa[k][pos] != GAP) {
                const unsigned aa_pos = (unsigned char) msa[k][pos];
                if (aa_pos < AA) matchWeight[pos * AA + aa_pos] += seqWeight[k];
            }
        }
        normalizeTo1(&matchWeight[pos * AA], AA, m.pBack);
    }
}

void neffM(const float *frequency, const float *seqWeight, float *Neff_M, size_t L, size_t setSize, const char *const *msa) {
    float Neff_HMM = 0.0f;
    for (size_t pos = 0; pos < L; pos++) {
        float sum = 0.0f;
        for (size_t aa = 0; aa < AA; ++aa) {
            float freq_pos_aa = frequency[pos * AA + aa];
            if (freq_pos_aa > 1E-10) sum -= freq_pos_aa * flog2(freq_pos_aa);
        }
        Neff_HMM += fpow2(sum);
    }
    Neff_HMM /= L;
    float Nlim = fmax(10.0, Neff_HMM + 1.0);
    float scale = flog2((Nlim - Neff_HMM) / (Nlim - 1.0));
    for (size_t pos = 0; pos < L; pos++) {
        float w_M = -1.0 / setSize;
        for (size_t k = 0; k < setSize; ++k)
            if (msa[k][pos] != GAP) w_M += seqWeight[k];
        Neff_M[pos] = (w_M < 0) ? 1.0 : Nlim - (Nlim - 1.0) * fpow2(scale * w_M);
    }
}

}  // namespace

struct sd_r2p {
    sd::SubMat m;             // blosum62 at bit factor 2, score bias -0.2 (result2profile.cpp:126)
    float *R[21];             // subMatrixPseudoCounts[a][b] = P(a|b) (BaseMatrix.cpp:110-123), rows 16-byte aligned
    std::vector<float> Rbacking;
    int8_t sub[21 * 21];
    sd::MaskCtx mask;
};

extern "C" {

int sd_r2p_create(sd_r2p **out) {
    if (!out) return SD_EINVAL;
    sd_r2p *r = new sd_r2p();
    sd::initSubMat(r->m, sd::MAT_BLOSUM62, 2.0f, -0.2f);
    r->Rbacking.assign(21 * 24 + 8, 0.0f);
    float *base = r->Rbacking.data();
    while (((uintptr_t) base) % 32) base++;
    // P(a|b) is taken against the background generateSubMatrix derives from the joint matrix itself (row sums, X fixed at
    // ANY_BACK = 1e-5; BaseMatrix.cpp:97-123), not against the member pBack the log-odds use
    double rowBack[21];
    for (int i = 0; i < 21; i++) {
        rowBack[i] = 0;
        for (int j = 0; j < 21; j++) rowBack[i] += r->m.probMatrix[i][j];
    }
    rowBack[20] = 1E-5;
    for (int i = 0; i < 21; i++) {
        r->R[i] = base + i * 24;
        for (int j = 0; j < 21; j++) r->R[i][j] = r->m.probMatrix[i][j] / (rowBack[j]);
    }
    for (int i = 0; i < 21; i++)
        for (int j = 0; j < 21; j++) r->sub[i * 21 + j] = (int8_t) r->m.sub[i][j];
    sd::initMaskCtx(r->m, r->mask);
    *out = r;
    return SD_OK;
}

void sd_r2p_destroy(sd_r2p *r) { delete r; }

int sd_r2p_batch(sd_r2p *r, const sd_r2p_params *par, uint32_t nQ, const uint8_t *qLetters, const uint64_t *qOff,
                 const uint64_t *edgeOff, const uint32_t *edgeT, const int32_t *edgeQStart, const int32_t *edgeTStart,
                 const char *btPool, const uint64_t *btOff, const uint8_t *tResidues, const uint64_t *tOff, char *outProfiles,
                 uint8_t *outConsensus) {
    if (!r || !par || !qLetters || !qOff || !edgeOff || !outProfiles) return SD_EINVAL;
    if (par->pcMode != 0) return SD_EUNSUPPORTED;   // context specific pseudo counts need the K4000 library
    std::vector<int> qid_vec;
    {
        const char *s = par->qid ? par->qid : "0.0";
        while (*s) {
            char *e;
            const float v = (float) strtod(s, &e);
            qid_vec.push_back(static_cast<int>(v * 100));
            s = (*e == ',') ? e + 1 : e;
            if (e == s && *e) break;
        }
        if (qid_vec.empty()) qid_vec.push_back(0);
        std::sort(qid_vec.begin(), qid_vec.end());
    }
    int status = SD_OK;
#pragma omp parallel
    {
        Work w;
        std::vector<const uint8_t *> edgeSeq;
        std::vector<const char *> bt;
        std::vector<uint32_t> btLen;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t q = 0; q < nQ; q++) {
            const uint8_t *center = qLetters + qOff[q];
            const int L = (int) (qOff[q + 1] - qOff[q]);
            char *out = outProfiles + qOff[q] * 25;
            if (L == 0) continue;
            const uint64_t e0 = edgeOff[q], e1 = edgeOff[q + 1];
            const size_t nE = (size_t) (e1 - e0);
            edgeSeq.resize(nE);
            bt.resize(nE);
            btLen.resize(nE);
            for (size_t e = 0; e < nE; e++) {
                edgeSeq[e] = tResidues + tOff[edgeT[e0 + e]];
                bt[e] = btPool + btOff[e0 + e];
                btLen[e] = (uint32_t) (btOff[e0 + e + 1] - btOff[e0 + e]);
            }
            size_t setSize = nE + 1;
            if (nE == 0) {   // singleSequenceMSA
                w.stride = ((size_t) L / BLK + 2) * BLK;
                w.msa.assign(w.stride, (char) GAP);
                for (int p = 0; p < L; p++) w.msa[p] = (char) center[p];
                w.rows.assign(1, w.msa.data());
            } else {
                buildMsa(w, center, L, nE, edgeSeq.data(), edgeQStart + e0, edgeTStart + e0, bt.data(), btLen.data());
            }
            size_t filtered = setSize;
            if (par->filterMsa)
                filtered = filterMsa(w, r->sub, (int) setSize, L, (int) (par->covMSAThr * 100), qid_vec, par->qsc,
                                     (int) (par->filterMaxSeqId * 100), par->Ndiff, par->filterMinEnable);
            // computePSSMFromMSA(filteredSetSize, centerLength, msa, wg, 0.0)
            w.seqWeight.assign(filtered, 0.0f);
            w.wi.assign(filtered, 0.0f);
            w.matchWeight.assign((size_t) (L + 2) * AA, 0.0f);
            w.pseudo.assign((size_t) (L + 1) * AA, 0.0f);
            w.profile.assign((size_t) (L + 1) * AA, 0.0f);
            w.neffM.assign(L + 1, 0.0f);
            w.pssm.assign((size_t) (L + 1) * AA, 0);
            w.consensus.assign(L + 1, 0);
            const char *const *rows = w.rows.data();
            sequenceWeights(w.seqWeight.data(), L, filtered, rows);
            normalizeTo1(w.seqWeight.data(), (int) filtered);
            if (!par->wg) {
                contextSpecificWeights(w, r->m, w.matchWeight.data(), w.seqWeight.data(), w.neffM.data(), L, filtered,
                                       (char *const *) rows);
            } else {
                matchWeights(r->m, w.matchWeight.data(), w.seqWeight.data(), filtered, L, rows);
                neffM(w.matchWeight.data(), w.seqWeight.data(), w.neffM.data(), L, filtered, rows);
            }
            // consensus (PSSMCalculator.cpp:652-667): the letter is mapped back to its number for the record
            for (int pos = 0; pos < L; pos++) {
                float maxw = 1E-8;
                int maxa = ANY;
                for (int aa = 0; aa < AA; ++aa) {
                    float prob = w.matchWeight[(size_t) pos * AA + aa];
                    if (prob - r->m.pBack[aa] > maxw) {
                        maxw = prob - r->m.pBack[aa];
                        maxa = aa;
                    }
                }
                w.consensus[pos] = (unsigned char) maxa;
            }
            if (par->pca > 0.0f) {
                // preparePseudoCounts + computePseudoCounts (PSSMCalculator.cpp:274-282,374-392)
                float __attribute__((aligned(32))) freq[24];
                for (int pos = 0; pos < L; pos++) {
                    memcpy(freq, &w.matchWeight[(size_t) pos * AA], AA * sizeof(float));
                    for (int aa = 0; aa < AA; aa++) w.pseudo[(size_t) pos * AA + aa] = scalarProd20(r->R[aa], freq);
                }
                for (int pos = 0; pos < L; pos++) {
                    float tau = fmin(1.0, par->pca / (1.0 + w.neffM[pos] / par->pcb));
                    for (int aa = 0; aa < AA; ++aa) {
                        float pseudoCounts = tau * w.pseudo[(size_t) pos * AA + aa];
                        float frequencySignal = (1.0 - tau) * w.matchWeight[(size_t) pos * AA + aa];
                        w.profile[(size_t) pos * AA + aa] = frequencySignal + pseudoCounts;
                    }
                }
            } else {
                for (int i = 0; i < L * AA; i++) w.profile[i] = w.matchWeight[i];
            }
            // computeLogPSSM(subMat, pssm, profile, 8.0, L, 0.0) (PSSMCalculator.cpp:251-265)
            for (int pos = 0; pos < L; pos++) {
                for (int aa = 0; aa < AA; aa++) {
                    const float aaProb = w.profile[(size_t) pos * AA + aa];
                    float logProb = flog2(aaProb / r->m.pBack[aa]);
                    const float bitFactor = 8.0, scoreBias = 0.0;
                    float pssmVal = bitFactor * logProb + bitFactor * scoreBias;
                    pssmVal = static_cast<char>((pssmVal < 0.0) ? pssmVal - 0.5 : pssmVal + 0.5);
                    float truncPssmVal = std::min(pssmVal, 127.0f);
                    truncPssmVal = std::max(-128.0f, truncPssmVal);
                    w.pssm[(size_t) pos * AA + aa] = truncPssmVal;
                }
            }
            if (par->compBiasCorr) {   // SubstitutionMatrix::calcGlobalAaBiasCorrection (SubstitutionMatrix.cpp:205-243)
                w.pNull.assign(L, 0.0f);
                char *ps = w.pssm.data();
                const int windowSize = 40;
                for (int pos = 0; pos < L; pos++)
                    for (int aa = 0; aa < 20; aa++) w.pNull[pos] += r->m.pBack[aa] * static_cast<float>(ps[pos * AA + aa]);
                for (int i = 0; i < L; i++) {
                    const int minPos = std::max(0, (i - windowSize / 2));
                    const int maxPos = std::min(L, (i + windowSize / 2));
                    const int windowLength = maxPos - minPos;
                    float aaSum[20];
                    memset(aaSum, 0, sizeof(float) * 20);
                    for (int j = minPos; j < maxPos; j++) {
                        if (i == j) continue;
                        for (int aa = 0; aa < 20; aa++) aaSum[aa] += ps[j * AA + aa] - w.pNull[j];
                    }
                    for (int aa = 0; aa < 20; aa++) ps[i * AA + aa] = static_cast<int>(ps[i * AA + aa] - aaSum[aa] / windowLength);
                }
            }
            if (par->maskProfile) {   // Masker::maskPssm: tantan on the query letters, masked positions score -1 everywhere
                w.masked.assign(center, center + L);
                sd::tantanMask(r->mask, w.masked.data(), L, par->maskProb);
                for (int pos = 0; pos < L; pos++)
                    if (w.masked[pos] == sd::X_CODE)
                        for (int aa = 0; aa < AA; aa++) w.pssm[(size_t) pos * AA + aa] = -1;
            }
            // Profile::toBuffer (PSSMCalculator.cpp:671-687)
            for (int pos = 0; pos < L; pos++) {
                char *rec = out + (size_t) pos * 25;
                memcpy(rec, &w.pssm[(size_t) pos * AA], AA);
                rec[20] = (char) center[pos];
                rec[21] = (char) w.consensus[pos];
                rec[22] = (char) neffToChar(w.neffM[pos]);
                rec[23] = 0;
                rec[24] = 0;
                if (outConsensus) outConsensus[qOff[q] + pos] = w.consensus[pos];
            }
        }
    }
    return status;
}

}  // extern "C"
