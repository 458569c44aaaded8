// result2profile: the host step between the iterations of `search --num-iterations` -- the alignments of
// a centre sequence (or profile) become its profile for the next iteration.  What the reference computes
// (M/src/util/result2profile.cpp:239-282), in the order of this file:
//   1. a query-anchored multiple alignment from the backtraces, one row per hit, exactly L columns (target residues
//      opposite a query gap are dropped)                                   M/src/alignment/MultipleAlignment.cpp:46-215
//   2. a redundancy / diversity filter over the rows (HH-suite's scheme: per-row gates against the centre, then a greedy
//      selection that raises the allowed pairwise identity only where the alignment is still thin)
//                                                                          M/src/alignment/MsaFilter.cpp:85-530
//   3. sequence weights -- per column sub-alignment weights (default) or global position-based weights (--wg) --, the
//      number of effective sequences per column, substitution-matrix pseudo counts, log-odds scores
//                                                                          M/src/alignment/PSSMCalculator.cpp:169-588
//   4. the global composition bias of the scores                          M/src/commons/SubstitutionMatrix.cpp:205-243
//   5. tantan masking of the centre                                       M/src/commons/Masker.cpp:57-80
//   6. the 25-byte record per position                                    PSSMCalculator.cpp:671-687, Sequence.h:458-471
// The next iteration consumes the rounded int8 scores, so every float expression below keeps the reference's operand types
// and grouping (single precision accumulators, the double literals, the approximate reciprocal with one Newton step); the
// structure around them -- row spans as one record per row, the gates and the pairwise comparison as functions, the greedy
// selection over an explicit candidate order -- is this file's own.  Compiled by g++ with the reference's AVX2 flags.
// Pinned byte for byte against the reference's classes (oracle/_ref/libsdref_r2p.so) on all 5 898 regression queries and on
// the parameter corners: tests/test_result2profile.py.
#include "sd_host.h"
#include "spacedust_gpu.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <memory>
#include <numeric>
#include <string>
#include <vector>
#include <omp.h>

namespace {

// cell codes of the alignment: residues 0..19, 20 = any residue (X), 21 = gap, 22 = gap before the first / behind the last residue
enum : int { kResidues = 20, kAny = 20, kGap = 21, kEndGap = 22 };
constexpr int kBlock = 32;   // rows are compared 32 cells at a time (the reference's AVX2 register width in bytes)

#define SD_MAY_ALIAS(x) x __attribute__((__may_alias__))

// MathUtil::flog2 / fpow2 (M/src/commons/MathUtil.h:121-163): polynomial approximations, literal types as in the reference
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

// scale to sum 1 in single precision; an all-zero vector takes the fallback distribution (MathUtil::NormalizeTo1)
inline float scaleToOne(float *v, int n, const double *fallback = nullptr) {
    float total = 0.0f;
    for (int x = 0; x < n; x++) total += v[x];
    if (total != 0.0f) {
        float factor = 1.0 / total;
        for (int x = 0; x < n; x++) v[x] *= factor;
    } else if (fallback) {
        for (int x = 0; x < n; x++) v[x] = fallback[x];
    }
    return total;
}

inline unsigned char effectiveCountByte(const float neff) {   // MathUtil::convertNeffToChar
    float scaled = std::min(255.0f, 1.0f + 64.0f * flog2(neff));
    return std::max(static_cast<unsigned char>(1), static_cast<unsigned char>(scaled + 0.5));
}

// 20-term dot product in the summation order of the reference's SSE helper (M/lib/simd/simd.h:901-953): five 4-lane
// products, pairwise adds, two shuffle-add rounds
inline float dot20(const float *a, const float *b) {
    float __attribute__((aligned(16))) out;
    const __m128 *A = (const __m128 *) a;
    const __m128 *B = (const __m128 *) b;
    const __m128 s01 = _mm_add_ps(_mm_mul_ps(A[0], B[0]), _mm_mul_ps(A[1], B[1]));
    const __m128 s23 = _mm_add_ps(_mm_mul_ps(A[2], B[2]), _mm_mul_ps(A[3], B[3]));
    __m128 acc = _mm_add_ps(_mm_add_ps(s01, s23), _mm_mul_ps(A[4], B[4]));
    for (int round = 0; round < 2; round++) {
        const __m128 even = _mm_shuffle_ps(acc, acc, _MM_SHUFFLE(2, 0, 2, 0));
        const __m128 odd = _mm_shuffle_ps(acc, acc, _MM_SHUFFLE(3, 1, 3, 1));
        acc = _mm_add_ps(odd, even);
    }
    _mm_store_ss(&out, acc);
    return out;
}

// ---- 1. the alignment -----------------------------------------------------------------------------------------------
// Rows of `stride` cells, padded with gaps to whole comparison blocks plus one (reads of whole blocks never leave a row).
struct Alignment {
    std::vector<char> cells;
    std::vector<char *> row;
    size_t stride = 0;
    int columns = 0;

    void reset(int L, size_t nRows, size_t columnsForStride) {
        columns = L;
        stride = (columnsForStride / kBlock + 2) * kBlock;
        cells.assign(stride * nRows, (char) kGap);
        row.resize(nRows);
        for (size_t r = 0; r < nRows; r++) row[r] = cells.data() + r * stride;
    }
};

struct Hit {                 // one alignment of the centre
    const uint8_t *target;   // numeric residues of the target sequence
    int32_t centreStart, targetStart;
    const char *path;        // expanded backtrace: M, I (target gap), D (centre gap)
    uint32_t pathLength;
};

// row of a hit: gaps up to the alignment start, then one cell per centre position the path covers
void placeHit(char *cells, const Hit &h) {
    if ((uint32_t) h.targetStart == 0xFFFFFFFFu) return;   // no coordinates: an all-gap row (MultipleAlignment.cpp:112-119)
    size_t column = (size_t) std::max(h.centreStart, 0);   // cells before it are gaps already
    uint32_t t = (uint32_t) h.targetStart;
    for (uint32_t s = 0; s < h.pathLength; s++) {
        const char step = h.path[s];
        if (step == 'M') {
            cells[column++] = (char) h.target[t++];
        } else if (step == 'I') {
            column++;   // a gap cell
        } else if (step == 'D') {
            // target residues opposite a centre gap leave no cell; the step that ends the run is consumed with it
            while (s < h.pathLength && h.path[s] == 'D') {
                t++;
                s++;
            }
            if (s >= h.pathLength) break;
            if (h.path[s] == 'M') cells[column++] = (char) h.target[t++];
            else if (h.path[s] == 'I') column++;
        }
    }
}

void buildAlignment(Alignment &A, const uint8_t *centre, int L, const std::vector<Hit> &hits) {
    if (hits.empty()) A.reset(L, 1, (size_t) L);   // a centre without hits: MultipleAlignment::singleSequenceMSA
    else A.reset(L, hits.size() + 1, (size_t) L + 1);
    for (int p = 0; p < L; p++) A.row[0][p] = (char) centre[p];
    for (size_t h = 0; h < hits.size(); h++) placeHit(A.row[h + 1], hits[h]);
}

// ---- 2. the diversity filter ----------------------------------------------------------------------------------------
struct FilterSettings {
    int coveragePercent;        // --cov: residues of a row, in percent of the centre length
    std::vector<int> identityLadder;   // --qid in percent, ascending; one value: a minimum identity with the centre
    float scorePerResidue;      // --qsc
    int maxPairIdentity;        // --max-seq-id in percent
    int diversity;              // --diff: rows wanted per 50-column window
    int minRowsToFilter;        // --filter-min-enable
};

struct RowSpan {
    int first, last;   // first / last column holding a residue (first = L, last = 0 for a row without residues)
    int residues;      // residues in between
};

enum Mark : char { kDropped = 0, kCandidate = 1, kCentre = 2 };

class DiversityFilter {
public:
    // marks the rows to keep, moves them to the front of A.row (order preserved) and returns their number
    size_t run(Alignment &A, size_t nRows, const int8_t *scoreMatrix, const FilterSettings &s);

private:
    static constexpr int kWindow = 25;   // half width of the window the per-column row count is maximised over

    const int8_t *scores = nullptr;
    int L = 0;
    std::vector<RowSpan> span;
    std::vector<char> mark;          // per alignment row
    // state of one selection (over the rows of one bucket)
    std::vector<int> member;         // bucket position -> alignment row
    std::vector<int> order;          // candidate order: centre first, then by descending residue count
    std::vector<char> accepted;      // per bucket position, in `order`: 0 no, 1 accepted, 2 the centre
    std::vector<int> lastTestedAt;   // per bucket position: the identity level the row was last compared at
    std::vector<int> rowsAt, rowsNear, levelAt;   // per column: accepted rows covering it, their windowed maximum, identity level in force
    int wanted = 0;                  // rows wanted per window; a bucket without a usable target resets it for the ones after it too

    void measure(const Alignment &A, size_t nRows);
    bool passesGates(const char *row, const RowSpan &r, const char *centre, const FilterSettings &s, float maxCentreDiffFraction) const;
    bool tooSimilar(const char *a, const RowSpan &ra, const char *b, const RowSpan &rb, float minDiffFraction) const;
    int select(const Alignment &A, const FilterSettings &s, int minCentreIdentity);
};

void DiversityFilter::measure(const Alignment &A, size_t nRows) {
    span.resize(nRows);
    for (size_t r = 0; r < nRows; r++) {
        const char *c = A.row[r];
        RowSpan &m = span[r];
        m.first = 0;
        while (m.first < L && c[m.first] >= kResidues) m.first++;
        m.last = L - 1;
        while (m.last > 0 && c[m.last] >= kResidues) m.last--;
        m.residues = 0;
        for (int i = m.first; i <= m.last; i++) m.residues += c[i] < kResidues;
    }
}

// the three gates a row passes on its own: coverage of the centre, average substitution score against the centre, identity
// with the centre
bool DiversityFilter::passesGates(const char *row, const RowSpan &r, const char *centre, const FilterSettings &s,
                                  float maxCentreDiffFraction) const {
    if (100 * r.residues < s.coveragePercent * L) return false;
    if (s.scorePerResidue > -10) {
        const float gapOpen = 6.0f, gapExtend = 1.0f;
        const float required = s.scorePerResidue * r.residues;
        float total = 0.0;
        int centreGapRun = 0, rowGapRun = 0;   // a run's first gap costs gapOpen, the following ones gapExtend
        for (int i = r.first; i <= r.last; i++) {
            const int a = centre[i], b = row[i];
            if (b < kResidues) {
                rowGapRun = 0;
                if (a < kResidues) {
                    centreGapRun = 0;
                    total += static_cast<float>(scores[a * 21 + b]);
                } else if (a != kAny) {
                    total -= centreGapRun++ ? gapExtend : gapOpen;
                }
            } else if (b != kAny && a < kResidues) {
                centreGapRun = 0;
                total -= rowGapRun++ ? gapExtend : gapOpen;
            }
        }
        if (total < required) return false;
    }
    if (maxCentreDiffFraction < 0.999) {
        const int limit = int(maxCentreDiffFraction * r.residues + 0.9999);
        int differing = 0;
        for (int i = r.first; i <= r.last && differing < limit; i++) differing += row[i] < kResidues && row[i] != centre[i];
        if (differing >= limit) return false;
    }
    return true;
}

// Are rows a (the candidate) and b (an accepted row) more alike than the identity level allows?  Differences are counted over
// the overlap of their spans, block by block (the count stops at a block border once it is enough); cells where either row
// has no residue are taken out of the compared length.
bool DiversityFilter::tooSimilar(const char *a, const RowSpan &ra, const char *b, const RowSpan &rb, float minDiffFraction) const {
    const int from = std::max(ra.first, rb.first), to = std::min(ra.last, rb.last);
    int compared = to - from + 1;
    const int enough = int(minDiffFraction * std::min(ra.residues, compared) + 0.999);
    const int blockBegin = from / kBlock, blockEnd = to / kBlock + 1;
    compared += std::abs(blockBegin * kBlock - from) + std::abs(blockEnd * kBlock - (to + 1));   // whole blocks are walked
    int differing = 0;
    for (int blk = blockBegin; blk < blockEnd && differing < enough; blk++) {
        // one block = one 32-byte register: cells where either row has no residue, and paired cells that differ
        const __m256i va = _mm256_loadu_si256((const __m256i *) (a + blk * kBlock));
        const __m256i vb = _mm256_loadu_si256((const __m256i *) (b + blk * kBlock));
        const __m256i lastResidue = _mm256_set1_epi8((char) (kResidues - 1));
        const __m256i noPair = _mm256_or_si256(_mm256_cmpgt_epi8(va, lastResidue), _mm256_cmpgt_epi8(vb, lastResidue));
        const __m256i unequal = _mm256_andnot_si256(_mm256_or_si256(noPair, _mm256_cmpeq_epi8(va, vb)), _mm256_set1_epi8((char) 0xFF));
        compared -= __builtin_popcount((unsigned) _mm256_movemask_epi8(noPair));
        differing += __builtin_popcount((unsigned) _mm256_movemask_epi8(unequal));
    }
    return differing < enough && float(differing) <= minDiffFraction * compared && compared > 0;
}

// One selection over the rows in `member` (member[0] is the centre).  Returns the number of rows kept besides the centre.
int DiversityFilter::select(const Alignment &A, const FilterSettings &s, int minCentreIdentity) {
    const int n = (int) member.size();
    const char *centre = A.row[member[0]];
    const RowSpan &centreSpan = span[member[0]];
    // candidates: everything but rows without residues; the centre is fixed (it is accepted from the start below, whatever
    // its mark says -- a centre of X only has no residues either)
    for (int p = 0; p < n; p++) mark[member[p]] = span[member[p]].residues == 0 ? kDropped : (p == 0 ? kCentre : kCandidate);
    order.resize(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin() + 1, order.end(), [&](int x, int y) { return span[member[x]].residues > span[member[y]].residues; });
    accepted.assign(n, 0);
    accepted[0] = 2;   // order[0] is bucket position 0
    lastTestedAt.assign(n, -1);
    rowsAt.assign(L, 0);
    for (int i = centreSpan.first; i <= centreSpan.last; i++) rowsAt[i] = 1;
    rowsNear.assign(L, 0);
    levelAt.assign(L, -1);
    int startLevel = 20;
    if (wanted <= 0 || wanted >= n) {   // no usable diversity target: a single pass at the maximal identity
        startLevel = s.maxPairIdentity;
        wanted = n;   // (stays for the buckets that follow, as in the reference, MsaFilter.cpp:223-227)
    }
    const float maxCentreDiffFraction = 0.9999 - 0.01 * minCentreIdentity;
    int survivors = 0;
    for (int p = 0; p < n; p++) {
        const int r = member[p];
        if (mark[r] == kCandidate && !passesGates(A.row[r], span[r], centre, s, maxCentreDiffFraction)) mark[r] = kDropped;
        survivors += mark[r] != kDropped;
    }
    if (survivors == 0) mark[member[0]] = kCandidate;
    // an empty ladder (--max-seq-id below 20 %): the gates were the whole filter.  The reference counts the centre twice on
    // this way out (MsaFilter.cpp:353-355), so does the caller's row count
    if (startLevel > s.maxPairIdentity) return survivors;
    int kept = 1;   // the centre
    int shortfall = wanted, shortfallBefore = 0, step = 0;
    for (int level = startLevel; level <= s.maxPairIdentity; level += step) {
        // where is the alignment still thinner than wanted?  Those columns take the current identity level.
        bool thickEnough = true;
        shortfallBefore = shortfall;
        shortfall = 0;
        for (int i = 0; i < L; i++) {
            const int lo = std::max(0, std::min(L - 2 * kWindow + 1, i - kWindow)), hi = std::min(L, std::max(2 * kWindow, i + kWindow));
            int most = 0;
            for (int j = lo; j < hi; j++) most = std::max(most, rowsAt[j]);
            rowsNear[i] = std::max(rowsNear[i], most);
            if (rowsNear[i] < wanted) {
                thickEnough = false;
                levelAt[i] = level;
                shortfall = std::max(shortfall, wanted - rowsNear[i]);
            }
        }
        if (thickEnough) break;
        for (int o = 0; o < n; o++) {
            if (accepted[o]) continue;
            const int p = order[o], r = member[p];
            if (mark[r] == kDropped) continue;
            if (level >= 100) {   // identical rows allowed: everything left is taken
                accepted[o] = 1;
                kept++;
                continue;
            }
            // the level that applies to this row: the highest in force on a column it covers
            float rowLevel = startLevel;
            for (int i = span[r].first; i <= span[r].last; i++)
                if (levelAt[i] > rowLevel) rowLevel = levelAt[i];
            if (lastTestedAt[p] == level) continue;
            lastTestedAt[p] = level;
            const float minDiffFraction = 0.9999 - 0.01 * rowLevel;
            bool redundant = false;
            for (int earlier = 0; earlier < o && !redundant; earlier++)
                if (accepted[earlier])
                    redundant = tooSimilar(A.row[r], span[r], A.row[member[order[earlier]]], span[member[order[earlier]]], minDiffFraction);
            if (!redundant) {
                accepted[o] = 1;
                kept++;
                for (int i = span[r].first; i <= span[r].last; i++) rowsAt[i]++;
            }
        }
        // the next level: larger steps while the shortfall shrinks slowly
        step = std::max(1, std::min(5, shortfall / (shortfallBefore - shortfall + 1) * step / 2));
    }
    for (int o = 0; o < n; o++) mark[member[order[o]]] = accepted[o];
    return kept - 1;
}

size_t DiversityFilter::run(Alignment &A, size_t nRows, const int8_t *scoreMatrix, const FilterSettings &s) {
    scores = scoreMatrix;
    L = A.columns;
    mark.assign(nRows, kDropped);
    wanted = s.diversity;
    measure(A, nRows);
    size_t keptBesidesCentre = 0;
    if (s.identityLadder.size() == 1) {
        if ((int) nRows < s.minRowsToFilter) {
            std::fill(mark.begin(), mark.end(), (char) kCandidate);
            mark[0] = kCentre;
            keptBesidesCentre = nRows - 1;
        } else {
            member.resize(nRows);
            std::iota(member.begin(), member.end(), 0);
            keptBesidesCentre = (size_t) select(A, s, s.identityLadder[0]);
        }
    } else {
        // several --qid values: the rows are binned by their identity with the centre, (ladder[b], ladder[b + 1]], and every
        // bin is thinned on its own (no centre-identity gate inside a bin)
        const char *centre = A.row[0];
        std::vector<int> identity(nRows, -1);
        for (size_t r = 1; r < nRows; r++) {
            int residues = 0, equal = 0;
            for (int i = 0; i < L; i++) {
                residues += A.row[r][i] < kResidues;
                equal += A.row[r][i] == centre[i] && A.row[r][i] < kResidues;
            }
            identity[r] = static_cast<int>(100.0f * (static_cast<float>(equal) / static_cast<float>(residues)));
        }
        for (size_t b = 0; b + 1 < s.identityLadder.size(); b++) {
            member.assign(1, 0);
            for (size_t r = 1; r < nRows; r++)
                if (identity[r] > s.identityLadder[b] && identity[r] <= s.identityLadder[b + 1]) member.push_back((int) r);
            if ((int) member.size() < s.minRowsToFilter) {
                for (size_t p = 1; p < member.size(); p++) mark[member[p]] = kCandidate;
                mark[0] = kCentre;
                keptBesidesCentre += member.size() - 1;
            } else {
                keptBesidesCentre += (size_t) select(A, s, 0);
            }
        }
    }
    // kept rows to the front, in their order
    size_t front = 0;
    for (size_t r = 0; r < nRows; r++) {
        if (mark[r] == kDropped) continue;
        if (front < r) std::swap(A.row[front], A.row[r]);
        front++;
    }
    return keptBesidesCentre + 1;
}

// ---- 3. weights, effective sequence numbers, pseudo counts, scores ---------------------------------------------------
struct ProfileScratch {
    std::vector<float> globalWeight, localWeight;   // per row
    std::vector<float> frequency;                    // [L + 2][20]: weighted residue frequencies per column
    std::vector<float> pseudo, mixed;                // [L + 1][20]
    std::vector<float> effective;                    // per column: number of effective sequences
    std::vector<char> score;                         // [L + 1][20]
    std::vector<unsigned char> consensus, masked;
    std::vector<int> counts;                         // [L + 1][24]: rows of the current sub-alignment per column and cell code
    std::vector<float> share;                        // [L + 1][24]: 1 / (distinct residues x rows with that residue)
    std::vector<float> subFrequency;                 // [L + 1][23]
    std::vector<int> distinct;
    std::vector<uint32_t> active;                    // rows with a residue in the current column
    std::vector<float> nullScore;
};

// Position-based sequence weights over the whole alignment (Henikoff & Henikoff) with a length prior: a residue shared by c
// of the rows in a column of d distinct residues gives each of them 1 / (c * d * (row residues + 30)).
void globalWeights(float *weight, int L, size_t nRows, const char *const *row) {
    std::vector<unsigned> rowCells(nRows);
    std::fill(weight, weight + nRows, 1e-6);
    for (size_t r = 0; r < nRows; r++) {
        unsigned cells = 0;
        for (int i = 0; i < L; i++) cells += row[r][i] != kGap;
        rowCells[r] = cells;
    }
    for (int i = 0; i < L; i++) {
        int seen[kResidues] = {0};
        for (size_t r = 0; r < nRows; r++) {
            const unsigned c = (unsigned char) row[r][i];
            if (c < (unsigned) kResidues) seen[c]++;
        }
        int distinct = 0;
        for (int a = 0; a < kResidues; a++) distinct += seen[a] != 0;
        if (distinct == 0) continue;
        for (size_t r = 0; r < nRows; r++) {
            const unsigned c = (unsigned char) row[r][i];
            if (c < (unsigned) kResidues) weight[r] += 1.0f / (float(seen[c]) * float(distinct) * (float(rowCells[r]) + 30.0f));
        }
    }
}

// Weights per column from the sub-alignment of the rows that have a residue there (HH-suite's position-specific weighting):
// the rows taking part change only where some row starts or ends a residue run, and only then is anything recomputed.  Over
// the columns jmin..jmax where at most a tenth of those rows have an end gap, a row's weight is the sum of the shares of its
// cells; the frequencies under those weights give the column's number of effective sequences (2^average entropy).
void columnWeights(ProfileScratch &w, const double *background, float *frequency, const float *globalWeight, float *effective, int L,
                   size_t nRows, char *const *row) {
    constexpr int kCodes = 24;    // cell codes 0..22 padded to eight-float groups
    constexpr int kFreq = 23;
    const float maxEndGapFraction = 0.1;
    const int minColumns = 20;
    w.counts.assign((size_t) (L + 1) * kCodes, 0);
    w.share.assign((size_t) (L + 1) * kCodes, 0.0f);
    w.subFrequency.assign((size_t) (L + 1) * kFreq, 0.0f);
    w.distinct.assign(L + 1, 0);
    int *count = w.counts.data();
    float *share = w.share.data();
    float *sub = w.subFrequency.data();
    float *local = w.localWeight.data();
    // gaps outside a row's first / last residue become end gaps for the duration
    for (size_t r = 0; r < nRows; r++) {
        for (int i = 0; i < L && row[r][i] == kGap; i++) row[r][i] = kEndGap;
        for (int i = L - 1; i >= 0 && row[r][i] == kGap; i--) row[r][i] = kEndGap;
    }
    int participating = 0;
    for (int i = 0; i < L; i++) {
        bool changed = false;
        for (size_t r = 0; r < nRows; r++) {
            const bool here = row[r][i] < kAny, before = i != 0 && row[r][i - 1] < kAny;
            if (here && !before) {
                changed = true;
                participating++;
                for (int j = 0; j < L; j++) count[j * kCodes + (int) row[r][j]]++;
            } else if (i != 0 && before && !here) {
                changed = true;
                participating--;
                for (int j = 0; j < L; j++) count[j * kCodes + (int) row[r][j]]--;
            }
        }
        if (changed) {
            for (size_t r = 0; r < nRows; r++) local[r] = 1E-8;
            int jmin = 0, jmax = L - 1;
            while (jmin < L && count[jmin * kCodes + kEndGap] > maxEndGapFraction * participating) jmin++;
            while (jmax >= 0 && count[jmax * kCodes + kEndGap] > maxEndGapFraction * participating) jmax--;
            const int width = jmax - jmin + 1;
            if (width < minColumns) {
                for (size_t r = 0; r < nRows; r++) local[r] = (row[r][i] < kAny) ? globalWeight[r] : 0.0f;
            } else {
                for (int j = jmin; j <= jmax; j++) {
                    int d = 0;
                    for (int a = 0; a < kAny; a++) d += count[j * kCodes + a] != 0;
                    w.distinct[j] = d;
                }
                for (int j = jmin; j <= jmax; j++) {
                    // share[j][a] = 1 / (distinct[j] * count[j][a]) as an approximate reciprocal refined once,
                    // y' = 2y - x y^2 (PSSMCalculator.cpp:494-505), eight residues at a time
                    const __m256 d = _mm256_cvtepi32_ps(_mm256_set1_epi32(w.distinct[j]));
                    for (int g = 0; g < (kAny + 7) / 8; g++) {
                        const __m256 x = _mm256_mul_ps(_mm256_cvtepi32_ps(_mm256_loadu_si256((const __m256i *) (count + j * kCodes + g * 8))), d);
                        const __m256 y = _mm256_rcp_ps(x);
                        _mm256_storeu_ps(share + j * kCodes + g * 8, _mm256_sub_ps(_mm256_add_ps(y, y), _mm256_mul_ps(x, _mm256_mul_ps(y, y))));
                    }
                    for (int a = kAny; a < kFreq; a++) share[j * kCodes + a] = 0.0f;
                }
                // a row's weight = the sum of its cells' shares in column order.  Eight rows at a time: eight independent
                // chains of additions instead of one (every row still adds its own shares in the same order)
                w.active.clear();
                for (size_t r = 0; r < nRows; r++)
                    if (row[r][i] < kAny) w.active.push_back((uint32_t) r);
                const size_t nActive = w.active.size();
                size_t g = 0;
                for (; g + 8 <= nActive; g += 8) {
                    const char *c0 = row[w.active[g]], *c1 = row[w.active[g + 1]], *c2 = row[w.active[g + 2]], *c3 = row[w.active[g + 3]];
                    const char *c4 = row[w.active[g + 4]], *c5 = row[w.active[g + 5]], *c6 = row[w.active[g + 6]], *c7 = row[w.active[g + 7]];
                    float a0 = local[w.active[g]], a1 = local[w.active[g + 1]], a2 = local[w.active[g + 2]], a3 = local[w.active[g + 3]];
                    float a4 = local[w.active[g + 4]], a5 = local[w.active[g + 5]], a6 = local[w.active[g + 6]], a7 = local[w.active[g + 7]];
                    for (int j = jmin; j <= jmax; j++) {
                        const float *sj = share + j * kCodes;
                        a0 += sj[(int) c0[j]]; a1 += sj[(int) c1[j]]; a2 += sj[(int) c2[j]]; a3 += sj[(int) c3[j]];
                        a4 += sj[(int) c4[j]]; a5 += sj[(int) c5[j]]; a6 += sj[(int) c6[j]]; a7 += sj[(int) c7[j]];
                    }
                    local[w.active[g]] = a0; local[w.active[g + 1]] = a1; local[w.active[g + 2]] = a2; local[w.active[g + 3]] = a3;
                    local[w.active[g + 4]] = a4; local[w.active[g + 5]] = a5; local[w.active[g + 6]] = a6; local[w.active[g + 7]] = a7;
                }
                for (; g < nActive; g++) {
                    const size_t r = w.active[g];
                    for (int j = jmin; j <= jmax; j++) local[r] += share[j * kCodes + (int) row[r][j]];
                }
            }
            // effective sequences of the sub-alignment
            effective[i] = 0.0;
            for (int j = jmin; j <= jmax; j++) memset(sub + j * kFreq, 0, kAny * sizeof(float));
            for (size_t r = 0; r < nRows; r++) {
                if (row[r][i] >= kAny) continue;
                for (int j = jmin; j <= jmax; j++) sub[j * kFreq + (int) row[r][j]] += local[r];
            }
            for (int j = jmin; j <= jmax; j++) {
                scaleToOne(sub + j * kFreq, kResidues);
                for (int a = 0; a < kResidues; a++)
                    if (sub[j * kFreq + a] > 1E-10) effective[i] -= sub[j * kFreq + a] * flog2(sub[j * kFreq + a]);
            }
            if (width > 0) effective[i] = fpow2(effective[i] / width);
            else effective[i] = 1.0;
        } else {
            effective[i] = i == 0 ? 0.0f : effective[i - 1];
        }
        // the column's residue frequencies under the current weights.  Cells 20..22 add their weight to the slots behind the
        // twenty of this column -- the first of the next column, which are cleared when that column's turn comes (the
        // reference does the same; `frequency` has a spare column)
        for (int a = 0; a < kResidues; a++) frequency[i * kResidues + a] = 0.0;
        for (size_t r = 0; r < nRows; r++) frequency[i * kResidues + (int) row[r][i]] += local[r];
        scaleToOne(frequency + i * kResidues, kResidues, background);
    }
    for (size_t r = 0; r < nRows; r++) {
        for (int i = 0; i < L && row[r][i] == kEndGap; i++) row[r][i] = kGap;
        for (int i = L - 1; i >= 0 && row[r][i] == kEndGap; i--) row[r][i] = kGap;
    }
}

// --wg: one weight per row everywhere
void globalFrequencies(const double *background, float *frequency, const float *weight, size_t nRows, int L, const char *const *row) {
    for (int i = 0; i < L; i++) {
        float *f = frequency + (size_t) i * kResidues;
        memset(f, 0, kResidues * sizeof(float));
        for (size_t r = 0; r < nRows; r++) {
            const unsigned c = (unsigned char) row[r][i];
            if (c < (unsigned) kResidues) f[c] += weight[r];
        }
        scaleToOne(f, kResidues, background);
    }
}

// --wg: effective sequences per column from the profile's average entropy, scaled by the weight present in the column
void globalEffective(const float *frequency, const float *weight, float *effective, int L, size_t nRows, const char *const *row) {
    float average = 0.0f;
    for (int i = 0; i < L; i++) {
        float entropy = 0.0f;
        for (int a = 0; a < kResidues; a++) {
            const float f = frequency[(size_t) i * kResidues + a];
            if (f > 1E-10) entropy -= f * flog2(f);
        }
        average += fpow2(entropy);
    }
    average /= L;
    float ceiling = fmax(10.0, average + 1.0);
    float scale = flog2((ceiling - average) / (ceiling - 1.0));
    for (int i = 0; i < L; i++) {
        float present = -1.0 / nRows;
        for (size_t r = 0; r < nRows; r++)
            if (row[r][i] != kGap) present += weight[r];
        effective[i] = (present < 0) ? 1.0 : ceiling - (ceiling - 1.0) * fpow2(scale * present);
    }
}

}  // namespace

struct sd_r2p {
    sd::SubMat matrix;                // blosum62 at bit factor 2, score bias -0.2 (result2profile.cpp:126)
    float *conditional[21];           // P(a|b) of the matrix (BaseMatrix.cpp:110-123), rows 16-byte aligned
    std::vector<float> conditionalStore;
    int8_t scores[21 * 21];
    sd::MaskCtx mask;
};

// steps 3b .. 6 of the header comment: from the weighted frequencies and effective sequence numbers of a centre to its record
static void finishProfile(const sd_r2p *r, const sd_r2p_params *par, const double *background, const uint8_t *centre, int L, uint64_t qOffQ,
                          ProfileScratch &w, char *outProfiles, uint8_t *outConsensus) {
    const uint64_t qOff[1] = {qOffQ};
    const uint32_t q = 0;
    // consensus: the residue most enriched over the background (PSSMCalculator.cpp:652-667)
    for (int i = 0; i < L; i++) {
        float best = 1E-8;
        int which = kAny;
        for (int a = 0; a < kResidues; a++) {
            const float f = w.frequency[(size_t) i * kResidues + a];
            if (f - background[a] > best) {
                best = f - background[a];
                which = a;
            }
        }
        w.consensus[i] = (unsigned char) which;
    }
    // substitution-matrix pseudo counts, mixed in with a weight that falls with the effective sequence number
    // (PSSMCalculator.cpp:274-282,374-392)
    if (par->pca > 0.0f) {
        float __attribute__((aligned(32))) column[24];
        for (int i = 0; i < L; i++) {
            memcpy(column, &w.frequency[(size_t) i * kResidues], kResidues * sizeof(float));
            for (int a = 0; a < kResidues; a++) w.pseudo[(size_t) i * kResidues + a] = dot20(r->conditional[a], column);
        }
        for (int i = 0; i < L; i++) {
            float tau = fmin(1.0, par->pca / (1.0 + w.effective[i] / par->pcb));
            for (int a = 0; a < kResidues; a++) {
                float fromPrior = tau * w.pseudo[(size_t) i * kResidues + a];
                float fromData = (1.0 - tau) * w.frequency[(size_t) i * kResidues + a];
                w.mixed[(size_t) i * kResidues + a] = fromData + fromPrior;
            }
        }
    } else {
        for (int x = 0; x < L * kResidues; x++) w.mixed[x] = w.frequency[x];
    }
    // log-odds in eighth bits, rounded half away from zero through a char, clamped (PSSMCalculator.cpp:251-265)
    for (int i = 0; i < L; i++) {
        for (int a = 0; a < kResidues; a++) {
            const float p = w.mixed[(size_t) i * kResidues + a];
            float logOdds = flog2(p / background[a]);
            const float bitFactor = 8.0, scoreBias = 0.0;
            float v = bitFactor * logOdds + bitFactor * scoreBias;
            v = static_cast<char>((v < 0.0) ? v - 0.5 : v + 0.5);
            w.score[(size_t) i * kResidues + a] = std::max(-128.0f, std::min(v, 127.0f));
        }
    }
    if (par->compBiasCorr) {
        // every column gives up the average excess (over its background expectation) of its 40-column neighbourhood
        // (SubstitutionMatrix.cpp:205-243)
        w.nullScore.assign(L, 0.0f);
        char *sc = w.score.data();
        const int window = 40;
        for (int i = 0; i < L; i++)
            for (int a = 0; a < kResidues; a++) w.nullScore[i] += background[a] * static_cast<float>(sc[i * kResidues + a]);
        for (int i = 0; i < L; i++) {
            const int lo = std::max(0, i - window / 2), hi = std::min(L, i + window / 2);
            const int span = hi - lo;
            float excess[kResidues];
            memset(excess, 0, sizeof(excess));
            for (int j = lo; j < hi; j++) {
                if (j == i) continue;
                for (int a = 0; a < kResidues; a++) excess[a] += sc[j * kResidues + a] - w.nullScore[j];
            }
            for (int a = 0; a < kResidues; a++) sc[i * kResidues + a] = static_cast<int>(sc[i * kResidues + a] - excess[a] / span);
        }
    }
    if (par->maskProfile) {   // tantan on the centre's letters: masked positions score -1 against everything
        w.masked.assign(centre, centre + L);
        sd::tantanMask(r->mask, w.masked.data(), L, par->maskProb);
        for (int i = 0; i < L; i++)
            if (w.masked[i] == sd::X_CODE) memset(&w.score[(size_t) i * kResidues], -1, kResidues);
    }
    char *out = outProfiles + qOff[q] * 25;
    for (int i = 0; i < L; i++) {
        char *rec = out + (size_t) i * 25;
        memcpy(rec, &w.score[(size_t) i * kResidues], kResidues);
        rec[20] = (char) centre[i];
        rec[21] = (char) w.consensus[i];
        rec[22] = (char) effectiveCountByte(w.effective[i]);
        rec[23] = 0;
        rec[24] = 0;
        if (outConsensus) outConsensus[qOff[q] + i] = w.consensus[i];
    }
}

extern "C" {

int sd_r2p_create(sd_r2p **out) {
    if (!out) return SD_EINVAL;
    sd_r2p *r = new sd_r2p();
    sd::initSubMat(r->matrix, sd::MAT_BLOSUM62, 2.0f, -0.2f);
    r->conditionalStore.assign(21 * 24 + 8, 0.0f);
    float *base = r->conditionalStore.data();
    while (((uintptr_t) base) % 32) base++;
    // P(a|b) against the background the joint matrix itself implies (row sums, X fixed at 1e-5; BaseMatrix.cpp:97-123), not
    // against the member background the log-odds use
    double implied[21];
    for (int a = 0; a < 21; a++) {
        implied[a] = 0;
        for (int b = 0; b < 21; b++) implied[a] += r->matrix.probMatrix[a][b];
    }
    implied[20] = 1E-5;
    for (int a = 0; a < 21; a++) {
        r->conditional[a] = base + a * 24;
        for (int b = 0; b < 21; b++) r->conditional[a][b] = r->matrix.probMatrix[a][b] / (implied[b]);
    }
    for (int a = 0; a < 21; a++)
        for (int b = 0; b < 21; b++) r->scores[a * 21 + b] = (int8_t) r->matrix.sub[a][b];
    sd::initMaskCtx(r->matrix, r->mask);
    *out = r;
    return SD_OK;
}

void sd_r2p_destroy(sd_r2p *r) { delete r; }

// the device half (csrc/hip/sd_r2p.hip): position-specific weights, frequencies and effective sequence numbers of a batch of
// filtered alignments
struct R2pTaskH {   // = R2pTask of sd_r2p.hip
    uint64_t cellOff, cmOff, weightOff, colOff, scratchOff;
    uint32_t nRows, L, stride, rowStride;
};
#ifdef SD_HOST_STAGES_ONLY   // oracle/Makefile: the CPU checker links the host stages without the device library
}
static int sdR2pColumnWeightsDevice(sd_ctx *, uint32_t, const void *, const char *, uint64_t, uint64_t, const float *, uint64_t, uint64_t,
                                    uint64_t, const float *, uint32_t, const double *, float *, float *) {
    return SD_EINVAL;
}
extern "C" {
#else
int sdR2pColumnWeightsDevice(sd_ctx *ctx, uint32_t nTasks, const void *tasksHost, const char *cells, uint64_t cellBytes, uint64_t cmBytes,
                             const float *globalWeight, uint64_t nWeights, uint64_t nColumns, uint64_t scratchElems, const float *rcpTable,
                             uint32_t rcpN, const double *background, float *freqOut, float *effOut);
#endif

static int r2pBatchImpl(sd_ctx *ctx, sd_r2p *r, const sd_r2p_params *par, uint32_t nQ, const uint8_t *qLetters, const uint64_t *qOff,
                        const uint64_t *edgeOff, const uint32_t *edgeT, const int32_t *edgeQStart, const int32_t *edgeTStart,
                        const char *btPool, const uint64_t *btOff, const uint8_t *tResidues, const uint64_t *tOff, char *outProfiles,
                        uint8_t *outConsensus);

int sd_r2p_batch(sd_r2p *r, const sd_r2p_params *par, uint32_t nQ, const uint8_t *qLetters, const uint64_t *qOff,
                 const uint64_t *edgeOff, const uint32_t *edgeT, const int32_t *edgeQStart, const int32_t *edgeTStart,
                 const char *btPool, const uint64_t *btOff, const uint8_t *tResidues, const uint64_t *tOff, char *outProfiles,
                 uint8_t *outConsensus) {
    return r2pBatchImpl(nullptr, r, par, nQ, qLetters, qOff, edgeOff, edgeT, edgeQStart, edgeTStart, btPool, btOff, tResidues, tOff,
                        outProfiles, outConsensus);
}

int sd_r2p_batch_device(sd_ctx *ctx, sd_r2p *r, const sd_r2p_params *par, uint32_t nQ, const uint8_t *qLetters, const uint64_t *qOff,
                        const uint64_t *edgeOff, const uint32_t *edgeT, const int32_t *edgeQStart, const int32_t *edgeTStart,
                        const char *btPool, const uint64_t *btOff, const uint8_t *tResidues, const uint64_t *tOff, char *outProfiles,
                        uint8_t *outConsensus) {
    if (!ctx) return SD_EINVAL;
    return r2pBatchImpl(ctx, r, par, nQ, qLetters, qOff, edgeOff, edgeT, edgeQStart, edgeTStart, btPool, btOff, tResidues, tOff,
                        outProfiles, outConsensus);
}

}  // extern "C"

// share = 1 / x for x = 0 .. n - 1 as PSSMCalculator computes it (:494-505): the CPU's approximate reciprocal, one Newton step
static void reciprocalTable(std::vector<float> &table, uint32_t n) {
    table.assign(((size_t) n + 7) / 8 * 8, 0.0f);
    for (uint32_t g = 0; g < n; g += 8) {
        const __m256 x = _mm256_cvtepi32_ps(_mm256_add_epi32(_mm256_set1_epi32((int) g), _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7)));
        const __m256 y = _mm256_rcp_ps(x);
        _mm256_storeu_ps(table.data() + g, _mm256_sub_ps(_mm256_add_ps(y, y), _mm256_mul_ps(x, _mm256_mul_ps(y, y))));
    }
}

static int r2pBatchImpl(sd_ctx *ctx, sd_r2p *r, const sd_r2p_params *par, uint32_t nQ, const uint8_t *qLetters, const uint64_t *qOff,
                        const uint64_t *edgeOff, const uint32_t *edgeT, const int32_t *edgeQStart, const int32_t *edgeTStart,
                        const char *btPool, const uint64_t *btOff, const uint8_t *tResidues, const uint64_t *tOff, char *outProfiles,
                        uint8_t *outConsensus) {
    if (!r || !par || !qLetters || !qOff || !edgeOff || !outProfiles) return SD_EINVAL;
    if (par->pcMode != 0) return SD_EUNSUPPORTED;   // context specific pseudo counts need the K4000 library
    FilterSettings fs;
    {   // --qid: comma separated fractions -> ascending percentages
        const char *s = par->qid ? par->qid : "0.0";
        while (*s) {
            char *end;
            const float v = (float) strtod(s, &end);
            if (end == s) break;
            fs.identityLadder.push_back(static_cast<int>(v * 100));
            s = (*end == ',') ? end + 1 : end;
        }
        if (fs.identityLadder.empty()) fs.identityLadder.push_back(0);
        std::sort(fs.identityLadder.begin(), fs.identityLadder.end());
    }
    fs.coveragePercent = (int) (par->covMSAThr * 100);
    fs.scorePerResidue = par->qsc;
    fs.maxPairIdentity = (int) (par->filterMaxSeqId * 100);
    fs.diversity = par->Ndiff;
    fs.minRowsToFilter = par->filterMinEnable;
    const double *background = r->matrix.pBack;
    if (ctx && !par->wg) {
        // ---- device path: groups of centres whose filtered alignments fit the staging budget
        //   A (host, a thread per centre)  alignment, diversity filter, global weights
        //   B (device)                     position-specific weights, frequencies, effective sequence numbers
        //   C (host, a thread per centre)  pseudo counts, scores, composition bias, masking, record
        struct Prepared {
            std::vector<char> cells;   // [nRows][stride], kept rows in order
            std::vector<float> gw;
            uint32_t nRows = 0, stride = 0;
        };
        const uint64_t budget = 1ull << 30;   // bytes of alignment cells per device call
        uint32_t g0 = 0;
        while (g0 < nQ) {
            // a group: by the UNfiltered size (rows x columns), which bounds the filtered one
            uint32_t g1 = g0;
            uint64_t est = 0;
            while (g1 < nQ) {
                const uint64_t Lq = qOff[g1 + 1] - qOff[g1];
                const uint64_t add = (edgeOff[g1 + 1] - edgeOff[g1] + 1) * (Lq + 4);
                if (g1 > g0 && est + add > budget) break;
                est += add;
                g1++;
            }
            const uint32_t nG = g1 - g0;
            std::vector<Prepared> prep(nG);
            const bool dbg = getenv("SD_DEBUG_TIMING") != nullptr;
            const double tA0 = omp_get_wtime();
#pragma omp parallel
            {
                Alignment A;
                DiversityFilter filter;
                std::vector<Hit> hits;
#pragma omp for schedule(dynamic, 8)
                for (uint32_t x = 0; x < nG; x++) {
                    const uint32_t q = g0 + x;
                    const uint8_t *centre = qLetters + qOff[q];
                    const int L = (int) (qOff[q + 1] - qOff[q]);
                    if (L == 0) continue;
                    hits.resize((size_t) (edgeOff[q + 1] - edgeOff[q]));
                    for (size_t h = 0; h < hits.size(); h++) {
                        const uint64_t e = edgeOff[q] + h;
                        hits[h].target = tResidues + tOff[edgeT[e]];
                        hits[h].centreStart = edgeQStart[e];
                        hits[h].targetStart = edgeTStart[e];
                        hits[h].path = btPool + btOff[e];
                        hits[h].pathLength = (uint32_t) (btOff[e + 1] - btOff[e]);
                    }
                    buildAlignment(A, centre, L, hits);
                    size_t nRows = hits.size() + 1;
                    if (par->filterMsa) nRows = filter.run(A, nRows, r->scores, fs);
                    Prepared &P = prep[x];
                    P.nRows = (uint32_t) nRows;
                    P.stride = (uint32_t) ((L + 3) / 4 * 4);
                    P.cells.assign((size_t) nRows * P.stride, (char) kGap);
                    for (size_t rr = 0; rr < nRows; rr++) memcpy(&P.cells[rr * P.stride], A.row[rr], (size_t) L);
                    P.gw.assign(nRows, 0.0f);
                    globalWeights(P.gw.data(), L, nRows, A.row.data());
                    scaleToOne(P.gw.data(), (int) nRows);
                }
            }
            const double tA1 = omp_get_wtime();
            // staging
            std::vector<R2pTaskH> tasks;
            std::vector<uint32_t> taskQ;
            uint64_t cellBytes = 0, cmBytes = 0, nWeights = 0, nColumns = 0, scratch = 0;
            uint32_t maxRows = 1;
            for (uint32_t x = 0; x < nG; x++) {
                const Prepared &P = prep[x];
                if (P.nRows == 0) continue;
                R2pTaskH t;
                t.nRows = P.nRows;
                t.L = (uint32_t) (qOff[g0 + x + 1] - qOff[g0 + x]);
                t.stride = P.stride;
                t.rowStride = (P.nRows + 63) / 64 * 64;
                t.cellOff = cellBytes;
                t.cmOff = cmBytes;
                t.weightOff = nWeights;
                t.colOff = nColumns;
                t.scratchOff = scratch;
                cellBytes += (uint64_t) P.nRows * P.stride;
                cmBytes += (uint64_t) t.L * t.rowStride;
                nWeights += P.nRows;
                nColumns += t.L;
                scratch += (uint64_t) (t.L + 1) * 24;
                maxRows = std::max(maxRows, P.nRows);
                tasks.push_back(t);
                taskQ.push_back(g0 + x);
            }
            if (!tasks.empty()) {
                // (raw arrays: every byte is written before it is read, and zeroing half a gigabyte per group is time)
                std::unique_ptr<char[]> cells(new char[cellBytes + 64]);
                std::unique_ptr<float[]> freq(new float[nColumns * kResidues + 1]), eff(new float[nColumns + 1]);
                std::vector<float> gw(nWeights + 1), table;
#pragma omp parallel for schedule(dynamic, 16)
                for (size_t k = 0; k < tasks.size(); k++) {
                    const Prepared &P = prep[taskQ[k] - g0];
                    memcpy(&cells[tasks[k].cellOff], P.cells.data(), P.cells.size());
                    memcpy(&gw[tasks[k].weightOff], P.gw.data(), P.gw.size() * sizeof(float));
                }
                const uint32_t rcpN = (maxRows + 1) * 20 + 8;
                reciprocalTable(table, rcpN);
                // launch order: the longest alignments first -- the work per alignment grows with the square of its columns, and the
                // batch is as slow as its last workgroup (the offsets into the staging arrays travel with a task)
                std::vector<uint32_t> ord(tasks.size());
                for (size_t k = 0; k < ord.size(); k++) ord[k] = (uint32_t) k;
                std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) {
                    if (tasks[x].L != tasks[y].L) return tasks[x].L > tasks[y].L;
                    return tasks[x].nRows > tasks[y].nRows;
                });
                std::vector<R2pTaskH> launch(tasks.size());
                for (size_t k = 0; k < ord.size(); k++) launch[k] = tasks[ord[k]];
                const double tB0 = omp_get_wtime();
                const int rc = sdR2pColumnWeightsDevice(ctx, (uint32_t) tasks.size(), launch.data(), cells.get(), cellBytes, cmBytes, gw.data(),
                                                        nWeights, nColumns, scratch, table.data(), rcpN, background, freq.get(), eff.get());
                if (rc != SD_OK) return rc;
                const double tB1 = omp_get_wtime();
                if (dbg)
                    fprintf(stderr, "[r2p] group of %u centres: %.1f MB cells, %llu rows; build + filter %.2f s, staging %.2f s, device %.2f s\n", nG,
                            cellBytes / 1e6, (unsigned long long) nWeights, tA1 - tA0, tB0 - tA1, tB1 - tB0);
#pragma omp parallel
                {
                    ProfileScratch w;
#pragma omp for schedule(dynamic, 8)
                    for (size_t k = 0; k < tasks.size(); k++) {
                        const uint32_t q = taskQ[k];
                        const int L = (int) tasks[k].L;
                        w.frequency.assign((size_t) (L + 2) * kResidues, 0.0f);
                        memcpy(w.frequency.data(), &freq[tasks[k].colOff * kResidues], (size_t) L * kResidues * sizeof(float));
                        w.effective.assign(L + 1, 0.0f);
                        memcpy(w.effective.data(), &eff[tasks[k].colOff], (size_t) L * sizeof(float));
                        w.pseudo.assign((size_t) (L + 1) * kResidues, 0.0f);
                        w.mixed.assign((size_t) (L + 1) * kResidues, 0.0f);
                        w.score.assign((size_t) (L + 1) * kResidues, 0);
                        w.consensus.assign(L + 1, 0);
                        finishProfile(r, par, background, qLetters + qOff[q], L, qOff[q], w, outProfiles, outConsensus);
                    }
                }
            }
            g0 = g1;
        }
        return SD_OK;
    }
#pragma omp parallel
    {
        Alignment A;
        DiversityFilter filter;
        ProfileScratch w;
        std::vector<Hit> hits;
#pragma omp for schedule(dynamic, 8)
        for (uint32_t q = 0; q < nQ; q++) {
            const uint8_t *centre = qLetters + qOff[q];
            const int L = (int) (qOff[q + 1] - qOff[q]);
            if (L == 0) continue;
            hits.resize((size_t) (edgeOff[q + 1] - edgeOff[q]));
            for (size_t h = 0; h < hits.size(); h++) {
                const uint64_t e = edgeOff[q] + h;
                hits[h].target = tResidues + tOff[edgeT[e]];
                hits[h].centreStart = edgeQStart[e];
                hits[h].targetStart = edgeTStart[e];
                hits[h].path = btPool + btOff[e];
                hits[h].pathLength = (uint32_t) (btOff[e + 1] - btOff[e]);
            }
            buildAlignment(A, centre, L, hits);
            size_t nRows = hits.size() + 1;
            if (par->filterMsa) nRows = filter.run(A, nRows, r->scores, fs);
            char *const *row = A.row.data();

            w.globalWeight.assign(nRows, 0.0f);
            w.localWeight.assign(nRows, 0.0f);
            w.frequency.assign((size_t) (L + 2) * kResidues, 0.0f);
            w.pseudo.assign((size_t) (L + 1) * kResidues, 0.0f);
            w.mixed.assign((size_t) (L + 1) * kResidues, 0.0f);
            w.effective.assign(L + 1, 0.0f);
            w.score.assign((size_t) (L + 1) * kResidues, 0);
            w.consensus.assign(L + 1, 0);
            globalWeights(w.globalWeight.data(), L, nRows, row);
            scaleToOne(w.globalWeight.data(), (int) nRows);
            if (!par->wg) {
                columnWeights(w, background, w.frequency.data(), w.globalWeight.data(), w.effective.data(), L, nRows, row);
            } else {
                globalFrequencies(background, w.frequency.data(), w.globalWeight.data(), nRows, L, row);
                globalEffective(w.frequency.data(), w.globalWeight.data(), w.effective.data(), L, nRows, row);
            }
            finishProfile(r, par, background, centre, L, qOff[q], w, outProfiles, outConsensus);
        }
    }
    return SD_OK;
}
