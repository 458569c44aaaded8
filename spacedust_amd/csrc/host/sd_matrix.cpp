// Substitution matrices, composition bias, extended 2-/3-mer tables, similar-k-mer
// enumeration (host side).  See sd_host.h for the reference citations.
#include "sd_host.h"

#include <immintrin.h>
#include <vector>
#include <memory>
#include <mutex>
#include "sd_matrix_data.inc"

#include <algorithm>
#include <cctype>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <utility>

namespace sd {

static const double ANY_BACK = 1E-5;   // M/src/commons/BaseMatrix.cpp:10

void initSubMat(SubMat &m, MatrixId id, float bitFactor, float scoreBias) {
    const char *alphabet = id == MAT_BLOSUM62 ? BLOSUM62_ALPHABET : VTML80_ALPHABET;
    const char *lambdaStr = id == MAT_BLOSUM62 ? BLOSUM62_LAMBDA : VTML80_LAMBDA;
    const char **bg = id == MAT_BLOSUM62 ? BLOSUM62_BACKGROUND : VTML80_BACKGROUND;
    const char *(*hb)[21] = id == MAT_BLOSUM62 ? BLOSUM62_HALFBITS : VTML80_HALFBITS;
    m.alphabetSize = ALPH;
    m.bitFactor = bitFactor;
    m.name = id == MAT_BLOSUM62 ? "blosum62.out" : "VTML80.out";

    // letter mapping: header order, then the alias rules of SubstitutionMatrix.cpp:257-298
    uint8_t base[256];
    memset(base, 0xFF, sizeof(base));
    for (int i = 0; i < ALPH; i++) {
        base[(int) alphabet[i]] = (uint8_t) i;
        m.num2aa[i] = alphabet[i];
    }
    m.num2aa[ALPH] = '\0';
    for (int letter = 0; letter < 256; letter++) {
        int up = toupper(letter);
        uint8_t v;
        switch (up) {
            case 'J': v = base[(int) 'L']; break;
            case 'U':
            case 'O': v = base[(int) 'X']; break;
            case 'Z': v = base[(int) 'E']; break;
            case 'B': v = base[(int) 'D']; break;
            default:
                v = (up < 256 && base[up] != 0xFF) ? base[up] : base[(int) 'X'];
                break;
        }
        m.aa2num[letter] = v;
    }

    // readProbMatrix (SubstitutionMatrix.cpp:327-418)
    m.lambda = strtod(lambdaStr, NULL);
    for (int i = 0; i < ALPH; i++) m.pBack[i] = strtod(bg[i], NULL);
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++) m.probMatrix[i][j] = strtod(hb[i][j], NULL);
    bool xIsPositive = false;
    for (int j = 0; j < ALPH; j++) {
        if (m.probMatrix[X_CODE][j] > 0 || m.probMatrix[j][X_CODE] > 0) {
            xIsPositive = true;
            break;
        }
    }
    if (xIsPositive == false) {
        for (int i = 0; i < ALPH - 1; i++) m.pBack[i] = m.pBack[i] * (1.0 - m.pBack[X_CODE]);
    }
    for (int i = 0; i < ALPH; i++)
        for (int j = 0; j < ALPH; j++)
            m.probMatrix[i][j] = std::exp(m.lambda * m.probMatrix[i][j]) * m.pBack[i] * m.pBack[j];

    // generateSubMatrix (BaseMatrix.cpp:97-159): background re-summed from the joint matrix
    double pb[ALPH];
    for (int i = 0; i < ALPH; i++) {
        pb[i] = 0;
        for (int j = 0; j < ALPH; j++) pb[i] += m.probMatrix[i][j];
    }
    pb[ALPH - 1] = ANY_BACK;
    for (int i = 0; i < ALPH; i++) {
        for (int j = 0; j < ALPH; j++) {
            double s = std::log2(m.probMatrix[i][j] / (pb[i] * pb[j]));
            double v = ((double) bitFactor * s + (double) scoreBias);
            m.sub[i][j] = (short) ((v < 0.0) ? v - 0.5 : v + 0.5);
        }
    }
}

void mapSequence(const SubMat &m, const char *seq, size_t len, uint8_t *out) {
    for (size_t i = 0; i < len; i++) out[i] = m.aa2num[(unsigned char) seq[i]];
}

// The correction of a residue depends only on (residue type, window length, integer window sum): the 21-step
// float/double accumulation over the background that follows the sum is memoised per matrix (same operations, same
// rounding, evaluated once per distinct triple).
namespace {
struct BiasMemo {
    short sub[ALPH][ALPH];
    double pBack[ALPH];
    int alphabetSize = 0;
    int lo = 0, span = 0;
    std::vector<uint32_t> tab;   // float bit patterns; EMPTY = not computed yet
};
constexpr uint32_t BIAS_EMPTY = 0x7FC0DEADu;   // a NaN payload the computation cannot produce from finite inputs
std::mutex biasMemoLock;
std::vector<std::unique_ptr<BiasMemo> > biasMemos;

BiasMemo *biasMemoFor(const SubMat &m) {
    std::lock_guard<std::mutex> g(biasMemoLock);
    for (auto &p : biasMemos)
        if (p->alphabetSize == m.alphabetSize && memcmp(p->sub, m.sub, sizeof(m.sub)) == 0 && memcmp(p->pBack, m.pBack, sizeof(m.pBack)) == 0)
            return p.get();
    std::unique_ptr<BiasMemo> b(new BiasMemo());
    memcpy(b->sub, m.sub, sizeof(m.sub));
    memcpy(b->pBack, m.pBack, sizeof(m.pBack));
    b->alphabetSize = m.alphabetSize;
    int mn = 0, mx = 0;
    for (int a = 0; a < ALPH; a++)
        for (int c = 0; c < ALPH; c++) {
            mn = std::min(mn, (int) m.sub[a][c]);
            mx = std::max(mx, (int) m.sub[a][c]);
        }
    b->lo = 40 * mn - mx;
    b->span = (40 * mx - mn) - b->lo + 1;
    b->tab.assign((size_t) ALPH * 41 * b->span, BIAS_EMPTY);
    biasMemos.push_back(std::move(b));
    return biasMemos.back().get();
}

inline float biasTail(const SubMat &m, const short *row, int sum, int windowLength) {
    float d = (float) sum;
    // "deltaS_i /= -1.0 * float(W)": the double literal promotes the division
    d = (float) ((double) d / (-1.0 * (double) (float) windowLength));
    for (int a = 0; a < m.alphabetSize; a++) {
        d = (float) ((double) d + m.pBack[a] * (double) (float) row[a]);
    }
    return d;
}
}  // namespace

void calcLocalAaBiasCorrection(const SubMat &m, const uint8_t *seq, int N, float *out, float scale) {
    const int windowSize = 40;
    BiasMemo *memo = biasMemoFor(m);
    // the window sums are integer: eight interior positions at a time with AVX2 gathers from an int32 copy of the matrix
    // (the float tail below keeps the reference's expression order per position)
    alignas(32) int32_t tab[ALPH * ALPH + 7];
    for (int a = 0; a < ALPH; a++)
        for (int b = 0; b < ALPH; b++) tab[a * ALPH + b] = m.sub[a][b];
    alignas(32) int32_t vsum[8];
    int vecFrom = -1;   // first position of the block whose sums are in vsum
    for (int i = 0; i < N; i++) {
        const int minPos = std::max(0, (i - windowSize / 2));
        const int maxPos = std::min(N, (i + windowSize / 2));
        const int windowLength = maxPos - minPos;
        int sum = 0;
        const short *row = m.sub[seq[i]];
        if (i >= vecFrom && i < vecFrom + 8 && vecFrom >= 0) {
            sum = vsum[i - vecFrom];
        } else if (i >= windowSize / 2 && i + 7 + windowSize / 2 <= N) {
            const __m256i rowBase = _mm256_mullo_epi32(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *) (seq + i))),
                                                       _mm256_set1_epi32(ALPH));
            __m256i acc = _mm256_setzero_si256();
            for (int off = -windowSize / 2; off < windowSize / 2; off++) {
                const __m256i sj = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *) (seq + i + off)));
                acc = _mm256_add_epi32(acc, _mm256_i32gather_epi32(tab, _mm256_add_epi32(rowBase, sj), 4));
            }
            const __m256i si = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *) (seq + i)));
            acc = _mm256_sub_epi32(acc, _mm256_i32gather_epi32(tab, _mm256_add_epi32(rowBase, si), 4));
            _mm256_store_si256((__m256i *) vsum, acc);
            vecFrom = i;
            sum = vsum[0];
        } else {
            for (int j = minPos; j < maxPos; j++) sum += row[seq[j]];
            sum -= row[seq[i]];
        }
        float d;
        const int rel = sum - memo->lo;
        if (rel >= 0 && rel < memo->span && windowLength <= 40) {
            uint32_t &slot = memo->tab[((size_t) seq[i] * 41 + windowLength) * memo->span + rel];
            uint32_t bits = __atomic_load_n(&slot, __ATOMIC_RELAXED);
            if (bits == BIAS_EMPTY) {
                d = biasTail(m, row, sum, windowLength);
                memcpy(&bits, &d, 4);
                __atomic_store_n(&slot, bits, __ATOMIC_RELAXED);
            } else {
                memcpy(&d, &bits, 4);
            }
        } else {
            d = biasTail(m, row, sum, windowLength);
        }
        out[i] = scale * d;
    }
}

void biasTableFull(const SubMat &m, std::vector<float> &tab, int &lo, int &span) {
    int mn = 0, mx = 0;
    for (int a = 0; a < ALPH; a++)
        for (int c = 0; c < ALPH; c++) {
            mn = std::min(mn, (int) m.sub[a][c]);
            mx = std::max(mx, (int) m.sub[a][c]);
        }
    lo = 40 * mn - mx;
    span = (40 * mx - mn) - lo + 1;
    tab.assign((size_t) ALPH * 41 * span, 0.0f);
#pragma omp parallel for collapse(2) schedule(static)
    for (int r = 0; r < ALPH; r++)
        for (int w = 1; w <= 40; w++)
            for (int x = 0; x < span; x++) tab[((size_t) r * 41 + w) * span + x] = biasTail(m, m.sub[r], lo + x, w);
}

void swCompBias8(const SubMat &blosum2, const uint8_t *seq, int N, int8_t *out) {
    std::vector<float> cb(N > 0 ? N : 1);
    calcLocalAaBiasCorrection(blosum2, seq, N, cb.data(), 1.0f);
    for (int i = 0; i < N; i++) {
        // a double expression in the reference (float -/+ the 0.5 literal), converted on assignment
        double dv = (cb[i] < 0.0) ? (double) cb[i] - 0.5 : (double) cb[i] + 0.5;
        out[i] = (int8_t) dv;
    }
}

void diagCompBias8(const SubMat &seed8, const uint8_t *seq, int N, int8_t *out) {
    std::vector<float> cb(N > 0 ? N : 1);
    calcLocalAaBiasCorrection(seed8, seq, N, cb.data(), 1.0f);
    diagCompBias8From(cb.data(), N, out);
}

void diagCompBias8From(const float *cb, int N, int8_t *out) {
    for (int i = 0; i < N; i++) {
        float a = cb[i];
        // float aaCorrBias = (a < 0.0) ? a/4 - 0.5 : a/4 + 0.5;  (double expression stored to float)
        float r = (float) ((a < 0.0) ? (double) (a / 4) - 0.5 : (double) (a / 4) + 0.5);
        out[i] = (int8_t) (char) r;
    }
}

void kmerThrBias16(const SubMat &seed8, const uint8_t *seq, int N, const uint8_t *seedPos, int k, int span,
                   int16_t *out) {
    std::vector<float> cb(N > 0 ? N : 1);
    calcLocalAaBiasCorrection(seed8, seq, N, cb.data(), 1.0f);
    kmerThrBias16From(cb.data(), N, seedPos, k, span, out);
}

void kmerThrBias16From(const float *cb, int N, const uint8_t *seedPos, int k, int span, int16_t *out) {
    for (int i = 0; i + span <= N; i++) {
        float b = 0;
        for (int p = 0; p < k; p++) b += cb[i + seedPos[p]];
        out[i] = (int16_t) ((b < 0.0) ? (double) b - 0.5 : (double) b + 0.5);
    }
}

int spacedPattern(int k, uint8_t *pos) {
    static const int8_t s6[] = {1, 1, 0, 1, 0, 1, 0, 0, 1, 1};
    static const int8_t s7[] = {1, 1, 0, 1, 0, 1, 1, 0, 0, 1, 1};
    const int8_t *s = k == 6 ? s6 : s7;
    int span = k == 6 ? 10 : 11;
    int n = 0;
    for (int i = 0; i < span; i++)
        if (s[i]) pos[n++] = (uint8_t) i;
    return span;
}

void buildExtMatrix(const SubMat &seed8, int wordLen, ExtMatrix &out, int threads) {
    const int A = ALPH - 1;
    uint32_t size = 1;
    for (int i = 0; i < wordLen; i++) size *= A;
    out.wordLen = wordLen;
    out.size = size;
    out.score.assign((size_t) size * size, 0);
    out.index.assign((size_t) size * size, 0);
    // enumeration order of the reference's Cartesian product: position 0 outermost
    // (ExtendedSubstitutionMatrix.cpp:105-126); index = sum a_j * 20^j (Indexer.h:21-84)
    std::vector<uint16_t> permIdx(size);
    std::vector<uint8_t> perm((size_t) size * wordLen);
    for (uint32_t p = 0; p < size; p++) {
        uint32_t r = p;
        uint32_t idx = 0;
        for (int j = wordLen - 1; j >= 0; j--) {
            perm[(size_t) p * wordLen + j] = (uint8_t) (r % A);
            r /= A;
        }
        uint32_t pw = 1;
        for (int j = 0; j < wordLen; j++) {
            idx += perm[(size_t) p * wordLen + j] * pw;
            pw *= A;
        }
        permIdx[p] = (uint16_t) idx;
    }
#pragma omp parallel num_threads(threads)
    {
        std::vector<std::pair<short, uint16_t> > tmp(size);
#pragma omp for schedule(static)
        for (uint32_t i = 0; i < size; i++) {
            const uint8_t *wi = &perm[(size_t) i * wordLen];
            for (uint32_t j = 0; j < size; j++) {
                const uint8_t *wj = &perm[(size_t) j * wordLen];
                short s = 0;
                for (int z = 0; z < wordLen; z++) s += seed8.sub[wi[z]][wj[z]];
                tmp[j].first = s;
                tmp[j].second = permIdx[j];
            }
            std::stable_sort(tmp.begin(), tmp.end(),
                             [](const std::pair<short, uint16_t> &a, const std::pair<short, uint16_t> &b) {
                                 return a.first > b.first;
                             });
            size_t row = (size_t) permIdx[i] * size;
            for (uint32_t z = 0; z < size; z++) {
                out.score[row + z] = tmp[z].first;
                out.index[row + z] = tmp[z].second;
            }
        }
    }
}

size_t generateKmerList(const ExtMatrix &three, const ExtMatrix &two, int k, const uint8_t *window, int thr,
                        std::vector<uint32_t> &out) {
    // divide strategy after the reversal at KmerGenerator.cpp:84-85: k=6 -> [3,3]; k=7 -> [2,2,3]
    const ExtMatrix *mats[3];
    int steps[3];
    int nSteps;
    if (k == 6) {
        nSteps = 2; steps[0] = 3; steps[1] = 3; mats[0] = &three; mats[1] = &three;
    } else {
        nSteps = 3; steps[0] = 2; steps[1] = 2; steps[2] = 3; mats[0] = &two; mats[1] = &two; mats[2] = &three;
    }
    const int A = ALPH - 1;
    uint32_t partIdx[3];
    uint64_t mult[3];
    short best[3], rest[3];
    int before = 0;
    uint64_t pw = 1;
    for (int s = 0; s < nSteps; s++) {
        uint32_t idx = 0, p = 1;
        for (int j = 0; j < steps[s]; j++) {
            idx += window[before + j] * p;
            p *= A;
        }
        partIdx[s] = idx;
        mult[s] = pw;
        best[s] = mats[s]->score[(size_t) idx * mats[s]->size];
        before += steps[s];
        for (int j = 0; j < steps[s]; j++) pw *= A;
    }
    rest[nSteps - 1] = 0;
    for (int s = nSteps - 1; s >= 1; s--) rest[s - 1] = (short) (best[s] + rest[s]);

    const short threshold = (short) thr;
    short cutoff1 = (short) (threshold - rest[0]);
    std::vector<short> sA, sB;
    std::vector<uint32_t> iA, iB;
    {
        const ExtMatrix &m0 = *mats[0];
        const int16_t *sc = &m0.score[(size_t) partIdx[0] * m0.size];
        const uint16_t *ix = &m0.index[(size_t) partIdx[0] * m0.size];
        for (uint32_t pos = 0; pos < m0.size && sc[pos] >= cutoff1; pos++) {
            sA.push_back(sc[pos]);
            iA.push_back(ix[pos]);
        }
    }
    for (int s = 0; s < nSteps - 1; s++) {
        const ExtMatrix &mn = *mats[s + 1];
        const int16_t *sc = &mn.score[(size_t) partIdx[s + 1] * mn.size];
        const uint16_t *ix = &mn.index[(size_t) partIdx[s + 1] * mn.size];
        sB.clear();
        iB.clear();
        for (size_t i = 0; i < sA.size(); i++) {
            const short si = sA[i];
            const short cutoff2 = (short) (threshold - si - rest[s + 1]);
            for (uint32_t j = 0; j < mn.size && sc[j] >= cutoff2; j++) {
                sB.push_back((short) (si + sc[j]));
                iB.push_back((uint32_t) (iA[i] + (uint64_t) ix[j] * mult[s + 1]));
            }
        }
        sA.swap(sB);
        iA.swap(iB);
    }
    out.swap(iA);
    return out.size();
}

int kmerThreshold(float sensitivity, int k) {
    float best;
    if (k == 5) {
        float base = 160.75;
        best = base - (sensitivity * 12.75);
    } else if (k == 6) {
        float base = 163.2;
        best = base - (sensitivity * 8.917);
    } else {
        float base = 186.15;
        best = base - (sensitivity * 11.22);
    }
    return static_cast<int>(best);
}

int autoKmerSize(uint64_t targetResidues) { return targetResidues < 3350000000ULL ? 6 : 7; }

unsigned diagonalBinSize(uint64_t dbsize, uint64_t l2) {
    for (unsigned b = 2; b <= 1024; b *= 2) {
        if (dbsize / b < l2) return b;
    }
    return 2048;
}

}  // namespace sd

// Text form (.out layout) of the embedded matrix data, for tools that take a matrix file's contents
// (e.g. handing "blosum62.out:<text>" to the reference's SubstitutionMatrix on a box without /root/reference).
extern "C" int sd_host_matrix_text(int which, char *buf, size_t cap) {
    const bool bl = which == 0;
    const char *alphabet = bl ? BLOSUM62_ALPHABET : VTML80_ALPHABET;
    const char *lambdaStr = bl ? BLOSUM62_LAMBDA : VTML80_LAMBDA;
    const char **bg = bl ? BLOSUM62_BACKGROUND : VTML80_BACKGROUND;
    const char *(*hb)[21] = bl ? BLOSUM62_HALFBITS : VTML80_HALFBITS;
    std::string s = "# Background (precomputed optional):";
    for (int i = 0; i < 21; i++) {
        s += " ";
        s += bg[i];
    }
    s += "\n# Lambda     (precomputed optional): ";
    s += lambdaStr;
    s += "\n  ";
    for (int i = 0; i < 21; i++) {
        s += " ";
        s += alphabet[i];
    }
    s += "\n";
    for (int i = 0; i < 21; i++) {
        s += alphabet[i];
        for (int j = 0; j < 21; j++) {
            s += " ";
            s += hb[i][j];
        }
        s += "\n";
    }
    if (s.size() + 1 > cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int) s.size();
}
