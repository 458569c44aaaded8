// Target index construction on the device: sd_target_build.
//
// What IndexBuilder::fillDatabase does on the host (M/src/prefiltering/IndexBuilder.cpp:55-239; sd_index.cpp is the host
// restatement this file is tested against, entry for entry):
//   1. every target is tantan-masked to X (M/src/commons/Masker.cpp:15-55, M/lib/tantan/tantan.cpp:308-460),
//   2. per target, the spaced k-mers without X whose self score reaches kmerThr are collected, each distinct k-mer once per
//      target at its smallest position (IndexTable.h:131-166,376-392),
//   3. the lists are ordered by (target, position) (IndexTable.h:182-189).
//
// On the MI355X:
//   ib_mask_kernel     one lane per target sequence, 64 sequences of similar length per wavefront (the host deals them out by
//                      length).  The 50 repeat-offset states of tantan's HMM live in 100 VGPRs, the window of the last 50
//                      residues in 53 more, the 21 x 21 likelihood ratios in LDS; forward pass, backward pass, mask bytes.
//                      Double precision with the operation order and the fused multiply-adds of the host build (4 partial
//                      sums over the offsets combined as (s0 + s2) + (s1 + s3), see sd_tantan.cpp), so the float posterior
//                      that is compared with --mask-prob has the same bits.  The forward posteriors (one float per residue)
//                      and the rescaling factors are kept between the passes in a scratch buffer laid out [position][lane].
//   ib_count / ib_emit per k-mer range [lo, hi) that fits the sort buffers: (k-mer - lo, target << 16 | position) records of
//                      the masked sequences in (target, position) order -- a wavefront walks its sequences 64 positions at a
//                      time, ballots keep the order -- then a stable radix sort by k-mer (hipCUB; setup path, not the search),
//   ib_keep_* / ib_place  adjacent records of one (k-mer, target) collapse to the first (= smallest position), the survivors are
//                      appended to the entry array (8 B each, the layout the prefilter kernels read) and counted per k-mer,
//   offsets            64-bit exclusive scan of the counts; stored as 32-bit starts, relative to one 64-bit base per 65 536
//                      k-mers when the index has 2^32 entries or more (the wide form of sd_target_create_wide).
// Algorithmic bytes: 1 B read + 1 B written per residue for the mask (+ 8 B of scratch traffic), 8 B written per entry; the
// sort moves 12 B x 2 x 4 passes per record.
#include "sd_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#pragma clang fp contract(off)   // every fused multiply-add below is written out: the host build's contractions, no others

namespace {

// the radix sort of the (k-mer, record) pairs: this library's own kernels
#include "sd_scan_sort.h"

constexpr int TT_OFF = 50;       // maxCycleLength (Masker.cpp:20-32)
constexpr int TT_SCALE = 16;     // rescaling period (tantan.cpp)
constexpr int TT_RAMP = 52;      // positions walked one at a time (the first 50 reach back to the sequence start only)
constexpr int IB_X = 20;         // residue code of X
constexpr int IB_ALPH = 21;

struct TantanPar {
    double b2b, f2b, f2f0;
    double b2f[TT_OFF];
};
__constant__ TantanPar c_tt;
__constant__ uint8_t c_ibSeed[8];
__constant__ int c_ibSelf[IB_ALPH];
__constant__ uint32_t c_ibPow[8];

struct MaskLane {
    double fg[TT_OFF];
    uint32_t W[TT_OFF + 3];   // residues (as byte offsets into a row of likelihood ratios) around the current block
    const char *lr;           // LDS
    __device__ __forceinline__ double ratio(uint32_t row, uint32_t w) const { return *(const double *) (lr + row + w); }
};

// forward step over all 50 offsets; the offsets' residues are W[SH + i]
template <int SH>
__device__ __forceinline__ double fwdFull(MaskLane &m, uint32_t row, double b) {
    const double f2f0 = c_tt.f2f0;
    double s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 48; i++) {
        const double f = m.fg[i];
        s[i & 3] += f;
        m.fg[i] = fma(c_tt.b2f[i], b, f * f2f0) * m.ratio(row, m.W[SH + i]);
    }
    double sum = (s[0] + s[2]) + (s[1] + s[3]);
#pragma unroll
    for (int i = 48; i < TT_OFF; i++) {
        const double f = m.fg[i];
        sum += f;
        m.fg[i] = fma(c_tt.b2f[i], b, f * f2f0) * m.ratio(row, m.W[SH + i]);
    }
    return sum;
}
// ... over the first maxOffset (wave-uniform, <= 50) offsets, residues W[i]
__device__ __forceinline__ double fwdPart(MaskLane &m, uint32_t row, double b, int maxOffset) {
    const double f2f0 = c_tt.f2f0;
    const int full = maxOffset & ~3;
    double s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 48; i++) {
        if (i < full) {
            const double f = m.fg[i];
            s[i & 3] += f;
            m.fg[i] = fma(c_tt.b2f[i], b, f * f2f0) * m.ratio(row, m.W[i]);
        }
    }
    double sum = (s[0] + s[2]) + (s[1] + s[3]);
#pragma unroll
    for (int i = 0; i < TT_OFF; i++) {
        if (i >= full && i < maxOffset) {
            const double f = m.fg[i];
            sum += f;
            m.fg[i] = fma(c_tt.b2f[i], b, f * f2f0) * m.ratio(row, m.W[i]);
        }
    }
    return sum;
}
template <int SH>
__device__ __forceinline__ double bwdFull(MaskLane &m, uint32_t row, double toBg) {
    const double f2f0 = c_tt.f2f0;
    double s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 48; i++) {
        const double f = m.fg[i] * m.ratio(row, m.W[SH + i]);
        s[i & 3] = fma(c_tt.b2f[i], f, s[i & 3]);
        m.fg[i] = fma(f, f2f0, toBg);
    }
    double sum = (s[0] + s[2]) + (s[1] + s[3]);
#pragma unroll
    for (int i = 48; i < TT_OFF; i++) {
        const double f = m.fg[i] * m.ratio(row, m.W[SH + i]);
        sum = fma(c_tt.b2f[i], f, sum);
        m.fg[i] = fma(f, f2f0, toBg);
    }
    return sum;
}
__device__ __forceinline__ double bwdPart(MaskLane &m, uint32_t row, double toBg, int maxOffset) {
    const double f2f0 = c_tt.f2f0;
    const int full = maxOffset & ~3;
    double s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 48; i++) {
        if (i < full) {
            const double f = m.fg[i] * m.ratio(row, m.W[i]);
            s[i & 3] = fma(c_tt.b2f[i], f, s[i & 3]);
            m.fg[i] = fma(f, f2f0, toBg);
        }
    }
    double sum = (s[0] + s[2]) + (s[1] + s[3]);
#pragma unroll
    for (int i = 0; i < TT_OFF; i++) {
        if (i >= full && i < maxOffset) {
            const double f = m.fg[i] * m.ratio(row, m.W[i]);
            sum = fma(c_tt.b2f[i], f, sum);
            m.fg[i] = fma(f, f2f0, toBg);
        }
    }
    return sum;
}

// One lane = one sequence; the lanes of a wavefront walk their sequences in step from residue 0 (position `pos` is
// wave-uniform; a lane whose sequence is shorter keeps computing on clamped residues, its stores are off and what it needs
// from the end of its own sequence -- z after the forward pass, the initial state of the backward pass -- is taken / set at
// the position where its sequence ends).
// probT / scaleT: per wavefront rows x 64 floats and rows / 16 x 64 doubles (rows = its longest sequence rounded up to 16),
// starting at row waveRow[w].
__global__ void __launch_bounds__(64)
ib_mask_kernel(const uint8_t *__restrict__ res, const uint64_t *__restrict__ seqOff, const uint32_t *__restrict__ order, uint32_t nSeq,
               const double *__restrict__ lrTable, const uint64_t *__restrict__ waveRow, float *__restrict__ probT,
               double *__restrict__ scaleT, double minMaskProb, uint8_t *__restrict__ masked, unsigned long long *__restrict__ nMasked,
               uint32_t waveBase /* first wavefront of this launch: the scratch holds the rows of [waveBase, waveBase + gridDim.x) */) {
    __shared__ double lr[IB_ALPH * IB_ALPH];
    for (int i = threadIdx.x; i < IB_ALPH * IB_ALPH; i += 64) lr[i] = lrTable[i];
    __syncthreads();
    const uint32_t wave = waveBase + blockIdx.x, lane = threadIdx.x;
    const uint32_t slot = wave * 64 + lane;
    const bool have = slot < nSeq;
    const uint32_t seq = have ? order[slot] : 0;
    const uint64_t base = have ? seqOff[seq] : 0;
    const int L = have ? (int) (seqOff[seq + 1] - base) : 0;
    int maxLen = L;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) maxLen = max(maxLen, __shfl_xor(maxLen, o, 64));
    if (maxLen == 0) return;
    const uint8_t *s = res + base;
    uint8_t *out = masked + base;
    const uint64_t row0 = waveRow[wave] - waveRow[waveBase];
    float *prob = probT + row0 * 64 + lane;
    double *scale = scaleT + row0 / TT_SCALE * 64 + lane;
    const double b2b = c_tt.b2b, f2b = c_tt.f2b;
    auto at = [&](int i) -> uint32_t {   // residue i as a byte offset into a row of lr (clamped: lanes outside their sequence)
        i = i >= L ? L - 1 : i;
        i = i < 0 ? 0 : i;
        return L > 0 ? (uint32_t) s[i] * 8u : 0u;
    };
    MaskLane m;
    m.lr = (const char *) lr;
#pragma unroll
    for (int i = 0; i < TT_OFF; i++) m.fg[i] = 0.0;
#pragma unroll
    for (int j = 0; j < TT_OFF + 3; j++) m.W[j] = 0;
    double bg = 1.0, z = 1.0;

    auto endOfStep = [&](int pos, double fromFg, double b) {
        double nb = fma(b, b2b, fromFg * f2b);
        if ((pos & (TT_SCALE - 1)) == TT_SCALE - 1) {
            const double sc = 1.0 / nb;
            if (pos < L) scale[(size_t) (pos / TT_SCALE) * 64] = sc;
            nb *= sc;
#pragma unroll
            for (int j = 0; j < TT_OFF; j++) m.fg[j] *= sc;
        }
        bg = nb;
        if (pos < L) prob[(size_t) pos * 64] = (float) nb;
        if (__ballot(pos == L - 1)) {   // some sequence ends here: its normalisation constant
            double all = 0.0;
#pragma unroll
            for (int j = 0; j < TT_OFF; j++) all += m.fg[j];
            const double zc = fma(nb, b2b, all * f2b);
            z = pos == L - 1 ? zc : z;
        }
    };

    // ---------------- forward: the first positions one at a time (W[i] = residue pos - 1 - i) ...
    const int rampEnd = min(TT_RAMP, maxLen);
    for (int pos = 0; pos < rampEnd; pos++) {
        const uint32_t cur = at(pos);
        const double b = bg;
        const double fromFg = fwdPart(m, cur * IB_ALPH, b, min(pos, TT_OFF));
        endOfStep(pos, fromFg, b);
#pragma unroll
        for (int j = TT_OFF + 2; j >= 1; j--) m.W[j] = m.W[j - 1];
        m.W[0] = cur;
    }
    // ... then blocks of four: W[j] = residue P0 + 2 - j, position P0 + u reads its offsets at W[3 - u + i]
    if (maxLen > TT_RAMP) {
#pragma unroll
        for (int j = TT_OFF + 2; j >= 3; j--) m.W[j] = m.W[j - 3];
        for (int P0 = TT_RAMP; P0 < maxLen; P0 += 4) {
            const uint32_t r0 = at(P0), r1 = at(P0 + 1), r2 = at(P0 + 2), r3 = at(P0 + 3);
            m.W[2] = r0; m.W[1] = r1; m.W[0] = r2;
            {
                const double b = bg;
                endOfStep(P0, fwdFull<3>(m, r0 * IB_ALPH, b), b);
            }
            if (P0 + 1 < maxLen) {
                const double b = bg;
                endOfStep(P0 + 1, fwdFull<2>(m, r1 * IB_ALPH, b), b);
            }
            if (P0 + 2 < maxLen) {
                const double b = bg;
                endOfStep(P0 + 2, fwdFull<1>(m, r2 * IB_ALPH, b), b);
            }
            if (P0 + 3 < maxLen) {
                const double b = bg;
                endOfStep(P0 + 3, fwdFull<0>(m, r3 * IB_ALPH, b), b);
            }
#pragma unroll
            for (int j = TT_OFF + 2; j >= 4; j--) m.W[j] = m.W[j - 4];
            m.W[3] = r3;
        }
    }

    // ---------------- backward, from the wavefront's last position down; a lane joins at its own last residue
    const double f2bInit = f2b;
    unsigned long long nX = 0;
    auto backStep = [&](int pos, uint32_t cur, auto &&offsets) {
        if (__ballot(pos == L - 1)) {
            const bool start = pos == L - 1;
            bg = start ? b2b : bg;
#pragma unroll
            for (int j = 0; j < TT_OFF; j++) m.fg[j] = start ? f2bInit : m.fg[j];
        }
        const bool live = pos < L;
        const float fwd = live ? prob[(size_t) pos * 64] : 0.f;
        const double nonRepeat = (double) fwd * bg / z;
        const float p = 1 - (float) nonRepeat;
        if (live) {
            const bool x = (double) p >= minMaskProb;   // float against double, tantan.cpp:527
            out[pos] = x ? (uint8_t) IB_X : (uint8_t) (cur >> 3);
            nX += x ? 1 : 0;
        }
        if ((pos & (TT_SCALE - 1)) == TT_SCALE - 1) {
            const double sc = live ? scale[(size_t) (pos / TT_SCALE) * 64] : 1.0;
            bg *= sc;
#pragma unroll
            for (int j = 0; j < TT_OFF; j++) m.fg[j] *= sc;
        }
        const double toBg = f2b * bg;
        const double toFg = offsets(cur * IB_ALPH, toBg);
        bg = fma(b2b, bg, toFg);
    };
    int pos = maxLen - 1;
    if (maxLen > TT_RAMP) {
        int P0 = (maxLen - 1) & ~3;
#pragma unroll
        for (int j = 0; j < TT_OFF + 3; j++) m.W[j] = at(P0 + 2 - j);
        for (; P0 >= TT_RAMP; P0 -= 4) {
            const uint32_t r3 = at(P0 + 3);
            if (P0 + 3 < maxLen) backStep(P0 + 3, r3, [&](uint32_t row, double toBg) { return bwdFull<0>(m, row, toBg); });
            if (P0 + 2 < maxLen) backStep(P0 + 2, m.W[0], [&](uint32_t row, double toBg) { return bwdFull<1>(m, row, toBg); });
            if (P0 + 1 < maxLen) backStep(P0 + 1, m.W[1], [&](uint32_t row, double toBg) { return bwdFull<2>(m, row, toBg); });
            backStep(P0, m.W[2], [&](uint32_t row, double toBg) { return bwdFull<3>(m, row, toBg); });
#pragma unroll
            for (int j = 0; j < TT_OFF - 1; j++) m.W[j] = m.W[j + 4];   // W'[j] = residue P0 - 2 - j
            m.W[TT_OFF - 1] = at(P0 - 51);
            m.W[TT_OFF] = at(P0 - 52);
            m.W[TT_OFF + 1] = at(P0 - 53);
            m.W[TT_OFF + 2] = at(P0 - 54);
        }
        pos = TT_RAMP - 1;   // W[i] = residue TT_RAMP - 2 - i = pos - 1 - i
    } else {
#pragma unroll
        for (int j = 0; j < TT_OFF + 3; j++) m.W[j] = at(pos - 1 - j);
    }
    for (; pos >= 0; pos--) {
        const uint32_t cur = at(pos);
        const int maxOffset = min(pos, TT_OFF);
        backStep(pos, cur, [&](uint32_t row, double toBg) { return bwdPart(m, row, toBg, maxOffset); });
#pragma unroll
        for (int j = 0; j < TT_OFF + 2; j++) m.W[j] = m.W[j + 1];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) nX += __shfl_xor(nX, o, 64);
    if (lane == 0 && nX) atomicAdd(nMasked, nX);
}

// ---------------------------------------------------------------------------------------------
// k-mer records
// ---------------------------------------------------------------------------------------------
constexpr int IB_SEQ_PER_WAVE = 16;
constexpr uint32_t IB_COARSE_BINS = 4096;

// k-mer index at position i of a masked sequence, 0xFFFFFFFF if it holds an X, runs off the end or scores below the threshold
template <int K>
__device__ __forceinline__ uint32_t kmerAt(const uint8_t *__restrict__ s, int L, int i, int span, int thr) {
    if (i + span > L) return 0xFFFFFFFFu;
    uint32_t idx = 0;
    int score = 0;
    bool x = false;
#pragma unroll
    for (int p = 0; p < K; p++) {
        const uint32_t a = s[i + c_ibSeed[p]];
        x |= a >= (uint32_t) IB_X;
        score += c_ibSelf[a < (uint32_t) IB_ALPH ? a : IB_X];
        idx += a * c_ibPow[p];
    }
    if (x || (thr > 0 && score < thr)) return 0xFFFFFFFFu;
    return idx;
}

// MODE 0: histogram of the valid k-mers over IB_COARSE_BINS equal ranges of the k-mer space (the host cuts the ranges of the
//         passes from it);  MODE 1: records of [lo, hi) per wavefront;  MODE 2: the records themselves at waveBase[w] + ...
template <int K, int MODE>
__global__ void __launch_bounds__(256)
ib_records_kernel(const uint8_t *__restrict__ masked, const uint64_t *__restrict__ seqOff, uint32_t nSeq, int span, int thr,
                  uint32_t binWidth, unsigned long long *__restrict__ coarse, uint32_t lo, uint32_t hi,
                  uint32_t *__restrict__ waveCount, const uint64_t *__restrict__ waveBase, uint32_t *__restrict__ keys,
                  uint64_t *__restrict__ vals) {
    __shared__ uint32_t hist[MODE == 0 ? IB_COARSE_BINS : 1];
    if (MODE == 0) {
        for (uint32_t i = threadIdx.x; i < IB_COARSE_BINS; i += 256) hist[i] = 0;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t s0 = (uint64_t) wave * IB_SEQ_PER_WAVE;
    uint64_t written = MODE == 2 && s0 < nSeq ? waveBase[wave] : 0;
    uint32_t count = 0;
    for (uint64_t q = s0; q < s0 + IB_SEQ_PER_WAVE && q < nSeq; q++) {
        const uint64_t b = seqOff[q];
        const int L = (int) (seqOff[q + 1] - b);
        const uint8_t *s = masked + b;
        for (int i0 = 0; i0 + span <= L; i0 += 64) {
            const uint32_t km = kmerAt<K>(s, L, i0 + (int) lane, span, thr);
            if (MODE == 0) {
                if (km != 0xFFFFFFFFu) atomicAdd(&hist[km / binWidth], 1u);
            } else {
                const bool in = km != 0xFFFFFFFFu && km >= lo && km < hi;
                const unsigned long long b64 = __ballot(in);
                if (MODE == 2 && in) {
                    const uint64_t at = written + __popcll(b64 & ((1ull << lane) - 1));
                    keys[at] = km - lo;
                    vals[at] = (q << 16) | (uint64_t) (i0 + (int) lane);
                }
                written += __popcll(b64);
                count += __popcll(b64);
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < IB_COARSE_BINS; i += 256)
            if (hist[i]) atomicAdd(&coarse[i], (unsigned long long) hist[i]);
    }
    if (MODE == 1 && lane == 0 && s0 < nSeq) waveCount[wave] = count;
}

// ---- exclusive scan of up to 2^32 32-bit counts into 64-bit sums (three launches: tile sums, tile bases, apply)
constexpr int IS_TILE = 4096, IS_NT = 256;
__global__ void __launch_bounds__(IS_NT) is_tile_sums(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ tileSum) {
    __shared__ unsigned long long part[IS_NT / 64];
    const uint64_t base = (uint64_t) blockIdx.x * IS_TILE;
    unsigned long long v = 0;
    for (int x = threadIdx.x; x < IS_TILE; x += IS_NT)
        if (base + x < n) v += in[base + x];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tileSum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ void __launch_bounds__(1024) is_tile_bases(uint64_t *__restrict__ tileSum, uint64_t nTiles, uint64_t *__restrict__ total) {
    // one workgroup: every thread owns a contiguous run of tiles
    __shared__ unsigned long long part[1024];
    const uint64_t per = (nTiles + 1023) / 1024;
    const uint64_t b = threadIdx.x * per, e = min(nTiles, b + per);
    unsigned long long v = 0;
    for (uint64_t i = b; i < e; i++) v += tileSum[i];
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) {
            const unsigned long long c = part[i];
            part[i] = run;
            run += c;
        }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint64_t i = b; i < e; i++) {
        const unsigned long long c = tileSum[i];
        tileSum[i] = run;
        run += c;
    }
}
__global__ void __launch_bounds__(IS_NT) is_apply(const uint32_t *__restrict__ in, uint64_t n, const uint64_t *__restrict__ tileBase,
                                                  uint64_t *__restrict__ out) {
    __shared__ unsigned long long wsum[IS_NT / 64];
    const uint64_t base = (uint64_t) blockIdx.x * IS_TILE;
    constexpr int PER = IS_TILE / IS_NT;
    uint32_t c[PER];
    unsigned long long mine = 0;
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const uint64_t i = base + (uint64_t) threadIdx.x * PER + x;
        c[x] = i < n ? in[i] : 0u;
        mine += c[x];
    }
    unsigned long long incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(incl, o, 64);
        if ((int) (threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned long long run = tileBase[blockIdx.x] + incl - mine;
    for (int w = 0; w < (int) (threadIdx.x >> 6); w++) run += wsum[w];
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const uint64_t i = base + (uint64_t) threadIdx.x * PER + x;
        if (i < n) out[i] = run;
        run += c[x];
    }
}

// ---- survivors of the sorted records: the first of every (k-mer, target) run
constexpr int IK_TILE = 2048, IK_NT = 256;
__device__ __forceinline__ bool ikKept(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals, uint64_t i) {
    return i == 0 || keys[i] != keys[i - 1] || (vals[i] >> 16) != (vals[i - 1] >> 16);
}
__global__ void __launch_bounds__(IK_NT) ib_keep_count(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals, uint64_t n,
                                                       uint32_t *__restrict__ tileCount) {
    __shared__ uint32_t part[IK_NT / 64];
    const uint64_t base = (uint64_t) blockIdx.x * IK_TILE;
    uint32_t c = 0;
    for (int x = threadIdx.x; x < IK_TILE; x += IK_NT)
        if (base + x < n) c += ikKept(keys, vals, base + x) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tileCount[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// appends the survivors to the entry array (in order) and counts them per k-mer
__global__ void __launch_bounds__(IK_NT) ib_place(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals, uint64_t n,
                                                  const uint64_t *__restrict__ tileBase, uint64_t entryBase, uint32_t lo,
                                                  uint2 *__restrict__ entries, uint32_t *__restrict__ counts) {
    __shared__ uint32_t wsum[IK_NT / 64];
    const uint64_t base = (uint64_t) blockIdx.x * IK_TILE;
    constexpr int PER = IK_TILE / IK_NT;
    uint32_t mine = 0;
    bool kept[PER];
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const uint64_t i = base + (uint64_t) threadIdx.x * PER + x;
        kept[x] = i < n && ikKept(keys, vals, i);
        mine += kept[x] ? 1u : 0u;
    }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, 64);
        if ((int) (threadIdx.x & 63) >= o) incl += up;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t at = entryBase + tileBase[blockIdx.x] + incl - mine;
    for (int w = 0; w < (int) (threadIdx.x >> 6); w++) at += wsum[w];
#pragma unroll
    for (int x = 0; x < PER; x++) {
        if (!kept[x]) continue;
        const uint64_t i = base + (uint64_t) threadIdx.x * PER + x;
        const uint64_t v = vals[i];
        entries[at++] = make_uint2((uint32_t) (v >> 16), (uint32_t) (v & 0xFFFFu));
        atomicAdd(&counts[lo + keys[i]], 1u);
    }
}

// list starts as the prefilter kernels read them: 32-bit, relative to base[i >> 16] in the wide form
__global__ void __launch_bounds__(256) ib_offsets(const uint64_t *__restrict__ start, uint64_t n, int wide, uint32_t *__restrict__ offsets,
                                                  uint64_t *__restrict__ blockBase) {
    const uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = wide ? start[i & ~0xFFFFull] : 0;
    if (wide && (i & 0xFFFFu) == 0) blockBase[i >> 16] = b;
    offsets[i] = (uint32_t) (start[i] - b);
}

template <typename T>
struct Scoped {   // device allocation released at scope exit
    T *p = nullptr;
    ~Scoped() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **) &p, std::max<size_t>(n, 1) * sizeof(T)); }
    void release() { if (p) (void) hipFree(p); p = nullptr; }
};

int scan32to64(sd_ctx *ctx, const uint32_t *in, uint64_t n, uint64_t *out, uint64_t *dTotal) {
    const uint64_t nTiles = (n + IS_TILE - 1) / IS_TILE;
    Scoped<uint64_t> tile;
    SD_HIP(ctx, tile.alloc(nTiles));
    hipLaunchKernelGGL(is_tile_sums, dim3((unsigned) nTiles), dim3(IS_NT), 0, ctx->stream, in, n, tile.p);
    hipLaunchKernelGGL(is_tile_bases, dim3(1), dim3(1024), 0, ctx->stream, tile.p, nTiles, dTotal);
    hipLaunchKernelGGL(is_apply, dim3((unsigned) nTiles), dim3(IS_NT), 0, ctx->stream, in, n, (const uint64_t *) tile.p, out);
    SD_HIP(ctx, hipGetLastError());
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

double firstRepeatOffsetProb(double probMult, int maxRepeatOffset) {
    if (probMult < 1 || probMult > 1) return (1 - probMult) / (1 - std::pow(probMult, maxRepeatOffset));
    return 1.0 / maxRepeatOffset;
}

}  // namespace

extern "C" int sd_target_build(sd_ctx *ctx, int kmerSize, int kmerThr, int mask, double maskProb, const uint8_t *residues,
                               const uint64_t *seqOffsets, uint32_t nSeq, const double *maskRatios, const int8_t *selfScore,
                               const int16_t *ext2Score, const uint16_t *ext2Index, const int16_t *ext3Score,
                               const uint16_t *ext3Index, sd_target **out, uint64_t *stats) {
    if (!ctx || !out || !residues || !seqOffsets || !maskRatios || !selfScore || !ext3Score || !ext3Index) return SD_EINVAL;
    if (kmerSize != 6 && kmerSize != 7) return sdFail(ctx, SD_EUNSUPPORTED, "k=%d: the device implements k=6 and k=7", kmerSize);
    if (kmerSize == 7 && (!ext2Score || !ext2Index)) return sdFail(ctx, SD_EINVAL, "k=7 needs the 2-mer score matrix");
    uint32_t longest = 0;
    for (uint32_t i = 0; i < nSeq; i++) {
        const uint64_t len = seqOffsets[i + 1] - seqOffsets[i];
        if (len > 65535)
            return sdFail(ctx, SD_EINVAL, "target %u has %llu residues; index positions are 16 bit (limit 65535, --max-seq-len)", i,
                          (unsigned long long) len);
        longest = std::max<uint32_t>(longest, (uint32_t) len);
    }
    // the kernels read their tables (seed positions, powers, tantan constants) from __constant__ memory: one build at a time
    static std::mutex buildMutex;
    std::lock_guard<std::mutex> buildLock(buildMutex);
    (void) hipSetDevice(ctx->device);
    sd_target *t = new sd_target();
    struct Guard {
        sd_target *t;
        ~Guard() { if (t) sd_target_destroy(t); }
    } guard{t};
    t->ctx = ctx;
    t->k = kmerSize;
    t->nSeq = nSeq;
    t->tableSize = kmerSize == 6 ? 64000000ull : 1280000000ull;
    t->hSeqOff.assign(seqOffsets, seqOffsets + nSeq + 1);
    const uint64_t total = seqOffsets[nSeq];
    const int span = kmerSize == 6 ? 10 : 11;
    {
        static const uint8_t s6[8] = {0, 1, 3, 5, 8, 9, 0, 0}, s7[8] = {0, 1, 3, 5, 6, 9, 10, 0};
        uint32_t pw[8] = {0};
        uint32_t p = 1;
        for (int i = 0; i < kmerSize; i++) {
            pw[i] = p;
            p *= 20;
        }
        int self[IB_ALPH];
        for (int a = 0; a < IB_ALPH; a++) self[a] = selfScore[a];
        SD_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_ibSeed), kmerSize == 6 ? s6 : s7, 8));
        SD_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_ibPow), pw, sizeof(pw)));
        SD_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_ibSelf), self, sizeof(self)));
    }
    auto up = [&](void **d, const void *h, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes + 64);
        if (e != hipSuccess) return e;
        return hipMemcpy(*d, h, bytes, hipMemcpyDefault);
    };
    SD_HIP(ctx, up((void **) &t->dSeqOff, seqOffsets, (nSeq + 1) * sizeof(uint64_t)));
    SD_HIP(ctx, up((void **) &t->dExt3Score, ext3Score, (size_t) 8000 * 8000 * sizeof(int16_t)));
    SD_HIP(ctx, up((void **) &t->dExt3Index, ext3Index, (size_t) 8000 * 8000 * sizeof(uint16_t)));
    {
        const int rcCum = sdBuildExt3Cum(ctx, t);
        if (rcCum != SD_OK) return rcCum;
    }
    if (ext2Score && ext2Index) {
        SD_HIP(ctx, up((void **) &t->dExt2Score, ext2Score, (size_t) 400 * 400 * sizeof(int16_t)));
        SD_HIP(ctx, up((void **) &t->dExt2Index, ext2Index, (size_t) 400 * 400 * sizeof(uint16_t)));
    }
    SD_HIP(ctx, hipMalloc((void **) &t->dMasked, std::max<uint64_t>(total, 1) + 64));

    // ---------------- 1. masking
    uint64_t nMaskedResidues = 0;
    {
        Scoped<uint8_t> dRes;
        SD_HIP(ctx, dRes.alloc(total + 64));
        SD_HIP(ctx, hipMemcpy(dRes.p, residues, total, hipMemcpyDefault));
        if (!mask || nSeq == 0) {
            SD_HIP(ctx, hipMemcpyAsync(t->dMasked, dRes.p, total, hipMemcpyDeviceToDevice, ctx->stream));
            SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            TantanPar par;
            const double repeatProb = 0.005, repeatEndProb = 0.05, decay = 0.9;
            par.b2b = 1 - repeatProb;
            par.f2b = repeatEndProb;
            par.f2f0 = 1 - repeatEndProb;
            double p = repeatProb * firstRepeatOffsetProb(decay, TT_OFF);
            for (int i = 0; i < TT_OFF; i++) {
                par.b2f[i] = p;
                p *= decay;
            }
            SD_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_tt), &par, sizeof(par)));
            // sequences dealt to the wavefronts by length (counting sort, stable)
            std::vector<uint32_t> order(nSeq);
            {
                std::vector<uint32_t> cnt((size_t) longest + 2, 0);
                for (uint32_t i = 0; i < nSeq; i++) cnt[(size_t) (seqOffsets[i + 1] - seqOffsets[i]) + 1]++;
                for (size_t l = 1; l < cnt.size(); l++) cnt[l] += cnt[l - 1];
                for (uint32_t i = 0; i < nSeq; i++) order[cnt[(size_t) (seqOffsets[i + 1] - seqOffsets[i])]++] = i;
            }
            const uint32_t nWaves = (nSeq + 63) / 64;
            std::vector<uint64_t> waveRow((size_t) nWaves + 1, 0);
            for (uint32_t w = 0; w < nWaves; w++) {
                const uint32_t lastSeq = order[std::min<uint64_t>((uint64_t) w * 64 + 63, nSeq - 1)];
                const uint64_t len = seqOffsets[lastSeq + 1] - seqOffsets[lastSeq];
                waveRow[w + 1] = waveRow[w] + ((len + TT_SCALE - 1) / TT_SCALE) * TT_SCALE;
            }
            Scoped<uint32_t> dOrder;
            Scoped<uint64_t> dWaveRow;
            Scoped<float> dProb;
            Scoped<double> dScale, dLr;
            Scoped<unsigned long long> dN;
            // the forward posteriors and rescaling factors between the passes: 64 lanes x (4 + 8 / 16) bytes per row of a wavefront,
            // four to five times the residue bytes of the whole DB -- so the wavefronts run in slices against a scratch budget (an
            // eighth of the device memory, SD_INDEX_MASK_BUDGET bytes for the tests; never less than the longest wavefront needs)
            uint64_t maskBudgetRows;
            {
                size_t freeB = 0, totalB = 0;
                (void) hipMemGetInfo(&freeB, &totalB);
                uint64_t budget = getenv("SD_INDEX_MASK_BUDGET") ? strtoull(getenv("SD_INDEX_MASK_BUDGET"), nullptr, 10) : (uint64_t) totalB / 8;
                budget = std::min<uint64_t>(budget, (uint64_t) freeB / 2);
                maskBudgetRows = std::max<uint64_t>(budget / (64 * (sizeof(float) + sizeof(double) / TT_SCALE + 1)), 1);
            }
            uint64_t sliceRows = 0;   // the largest slice
            {
                uint32_t w0 = 0;
                while (w0 < nWaves) {
                    uint32_t w1 = w0 + 1;
                    while (w1 < nWaves && waveRow[w1 + 1] - waveRow[w0] <= maskBudgetRows) w1++;
                    sliceRows = std::max(sliceRows, waveRow[w1] - waveRow[w0]);
                    w0 = w1;
                }
            }
            SD_HIP(ctx, dOrder.alloc(nSeq));
            SD_HIP(ctx, dWaveRow.alloc(nWaves + 1));
            SD_HIP(ctx, dProb.alloc(sliceRows * 64));
            SD_HIP(ctx, dScale.alloc(sliceRows / TT_SCALE * 64 + 64));
            SD_HIP(ctx, dLr.alloc(IB_ALPH * IB_ALPH));
            SD_HIP(ctx, dN.alloc(1));
            SD_HIP(ctx, hipMemcpy(dOrder.p, order.data(), (size_t) nSeq * sizeof(uint32_t), hipMemcpyHostToDevice));
            SD_HIP(ctx, hipMemcpy(dWaveRow.p, waveRow.data(), ((size_t) nWaves + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
            SD_HIP(ctx, hipMemcpy(dLr.p, maskRatios, IB_ALPH * IB_ALPH * sizeof(double), hipMemcpyHostToDevice));
            SD_HIP(ctx, hipMemsetAsync(dN.p, 0, sizeof(unsigned long long), ctx->stream));
            for (uint32_t w0 = 0; w0 < nWaves;) {   // slices of wavefronts, one after the other on the stream (they share the scratch)
                uint32_t w1 = w0 + 1;
                while (w1 < nWaves && waveRow[w1 + 1] - waveRow[w0] <= maskBudgetRows) w1++;
                ProfScope ps(ctx, "index_mask");
                hipLaunchKernelGGL(ib_mask_kernel, dim3(w1 - w0), dim3(64), 0, ctx->stream, (const uint8_t *) dRes.p, (const uint64_t *) t->dSeqOff,
                                   (const uint32_t *) dOrder.p, nSeq, (const double *) dLr.p, (const uint64_t *) dWaveRow.p, dProb.p, dScale.p,
                                   maskProb, t->dMasked, dN.p, w0);
                w0 = w1;
            }
            SD_HIP(ctx, hipGetLastError());
            SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            unsigned long long h = 0;
            SD_HIP(ctx, hipMemcpy(&h, dN.p, sizeof(h), hipMemcpyDeviceToHost));
            nMaskedResidues = h;
        }
    }

    // ---------------- 2. records per k-mer range, sorted; survivors appended to the entries
    const uint32_t nWavesR = (uint32_t) (((uint64_t) nSeq + IB_SEQ_PER_WAVE - 1) / IB_SEQ_PER_WAVE);
    const unsigned gridR = (nWavesR + 3) / 4;
    const uint32_t binWidth = (uint32_t) ((t->tableSize + IB_COARSE_BINS - 1) / IB_COARSE_BINS);
    std::vector<unsigned long long> coarse(IB_COARSE_BINS, 0);
    if (nSeq) {
        Scoped<unsigned long long> dCoarse;
        SD_HIP(ctx, dCoarse.alloc(IB_COARSE_BINS));
        SD_HIP(ctx, hipMemsetAsync(dCoarse.p, 0, IB_COARSE_BINS * sizeof(unsigned long long), ctx->stream));
        {
            ProfScope ps(ctx, "index_kmer_histogram");
            if (kmerSize == 6)
                hipLaunchKernelGGL((ib_records_kernel<6, 0>), dim3(gridR), dim3(256), 0, ctx->stream, (const uint8_t *) t->dMasked,
                                   (const uint64_t *) t->dSeqOff, nSeq, span, kmerThr, binWidth, dCoarse.p, 0u, 0u, (uint32_t *) nullptr,
                                   (const uint64_t *) nullptr, (uint32_t *) nullptr, (uint64_t *) nullptr);
            else
                hipLaunchKernelGGL((ib_records_kernel<7, 0>), dim3(gridR), dim3(256), 0, ctx->stream, (const uint8_t *) t->dMasked,
                                   (const uint64_t *) t->dSeqOff, nSeq, span, kmerThr, binWidth, dCoarse.p, 0u, 0u, (uint32_t *) nullptr,
                                   (const uint64_t *) nullptr, (uint32_t *) nullptr, (uint64_t *) nullptr);
        }
        SD_HIP(ctx, hipGetLastError());
        SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SD_HIP(ctx, hipMemcpy(coarse.data(), dCoarse.p, IB_COARSE_BINS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    uint64_t nRecords = 0, biggestBin = 0;
    for (unsigned long long c : coarse) {
        nRecords += c;
        biggestBin = std::max<uint64_t>(biggestBin, c);
    }
    // records per pass: bounded by the sort's 32-bit item count and by SD_INDEX_PASS (tests: many small passes)
    uint64_t cap = 1ull << 30;
    if (getenv("SD_INDEX_PASS")) cap = std::max<uint64_t>(1, strtoull(getenv("SD_INDEX_PASS"), nullptr, 10));
    cap = std::max(cap, biggestBin);
    if (cap >= (1ull << 31))
        return sdFail(ctx, SD_EUNSUPPORTED, "one of the %u k-mer ranges holds %llu index records (limit 2^31 per sort)", IB_COARSE_BINS,
                      (unsigned long long) biggestBin);
    SD_HIP(ctx, hipMalloc((void **) &t->dEntries, (std::max<uint64_t>(nRecords, 1) + 8) * sizeof(uint2)));
    SD_HIP(ctx, hipMalloc((void **) &t->dOffsets, (t->tableSize + 1) * sizeof(uint32_t) + 64));
    SD_HIP(ctx, hipMemsetAsync(t->dOffsets, 0, (t->tableSize + 1) * sizeof(uint32_t), ctx->stream));   // counts first
    uint64_t nEntries = 0, nPasses = 0;
    if (nRecords) {
        const uint64_t bufN = std::min<uint64_t>(cap, nRecords);
        Scoped<uint32_t> k0, k1, dWaveCount, dTileCount;
        Scoped<uint64_t> v0, v1, dWaveBase, dTileBase, dTot;
        Scoped<uint8_t> dTmp;
        SD_HIP(ctx, k0.alloc(bufN));
        SD_HIP(ctx, k1.alloc(bufN));
        SD_HIP(ctx, v0.alloc(bufN));
        SD_HIP(ctx, v1.alloc(bufN));
        SD_HIP(ctx, dWaveCount.alloc(nWavesR));
        SD_HIP(ctx, dWaveBase.alloc(nWavesR));
        SD_HIP(ctx, dTileCount.alloc((bufN + IK_TILE - 1) / IK_TILE));
        SD_HIP(ctx, dTileBase.alloc((bufN + IK_TILE - 1) / IK_TILE));
        SD_HIP(ctx, dTot.alloc(1));
        SD_HIP(ctx, dTmp.alloc(sdRadixSortCountsBytes()));   // the radix sort's count matrix (sd_scan_sort.h)
        uint32_t bin = 0;
        while (bin < IB_COARSE_BINS) {
            uint32_t e = bin;
            uint64_t n = 0;
            while (e < IB_COARSE_BINS && n + coarse[e] <= cap) n += coarse[e++];
            const uint32_t lo = bin * binWidth;
            const uint32_t hi = (uint32_t) std::min<uint64_t>((uint64_t) e * binWidth, t->tableSize);
            bin = e;
            if (n == 0) continue;
            nPasses++;
            ProfScope ps(ctx, "index_pass");
            if (kmerSize == 6)
                hipLaunchKernelGGL((ib_records_kernel<6, 1>), dim3(gridR), dim3(256), 0, ctx->stream, (const uint8_t *) t->dMasked,
                                   (const uint64_t *) t->dSeqOff, nSeq, span, kmerThr, binWidth, (unsigned long long *) nullptr, lo, hi,
                                   dWaveCount.p, (const uint64_t *) nullptr, (uint32_t *) nullptr, (uint64_t *) nullptr);
            else
                hipLaunchKernelGGL((ib_records_kernel<7, 1>), dim3(gridR), dim3(256), 0, ctx->stream, (const uint8_t *) t->dMasked,
                                   (const uint64_t *) t->dSeqOff, nSeq, span, kmerThr, binWidth, (unsigned long long *) nullptr, lo, hi,
                                   dWaveCount.p, (const uint64_t *) nullptr, (uint32_t *) nullptr, (uint64_t *) nullptr);
            SD_HIP(ctx, hipGetLastError());
            int rc = scan32to64(ctx, dWaveCount.p, nWavesR, dWaveBase.p, dTot.p);
            if (rc != SD_OK) return rc;
            uint64_t got = 0;
            SD_HIP(ctx, hipMemcpy(&got, dTot.p, sizeof(got), hipMemcpyDeviceToHost));
            if (got != n) return sdFail(ctx, SD_EHIP, "sd_target_build: k-mer range [%u, %u) holds %llu records, the histogram said %llu", lo, hi,
                                        (unsigned long long) got, (unsigned long long) n);
            if (kmerSize == 6)
                hipLaunchKernelGGL((ib_records_kernel<6, 2>), dim3(gridR), dim3(256), 0, ctx->stream, (const uint8_t *) t->dMasked,
                                   (const uint64_t *) t->dSeqOff, nSeq, span, kmerThr, binWidth, (unsigned long long *) nullptr, lo, hi,
                                   (uint32_t *) nullptr, (const uint64_t *) dWaveBase.p, k0.p, v0.p);
            else
                hipLaunchKernelGGL((ib_records_kernel<7, 2>), dim3(gridR), dim3(256), 0, ctx->stream, (const uint8_t *) t->dMasked,
                                   (const uint64_t *) t->dSeqOff, nSeq, span, kmerThr, binWidth, (unsigned long long *) nullptr, lo, hi,
                                   (uint32_t *) nullptr, (const uint64_t *) dWaveBase.p, k0.p, v0.p);
            SD_HIP(ctx, hipGetLastError());
            int bits = 1;
            while (bits < 32 && (1ull << bits) < (uint64_t) (hi - lo)) bits++;
            // stable LSD radix sort of (k-mer, record) pairs by the range's k-mer bits (sd_scan_sort.h, 8 bits per pass), ping-pong between
            // the two buffer pairs: with an odd number of passes the input pair doubles as the scratch pair (it is read by the first pass
            // only and first written by the second) and the result lands in (k1, v1), with an even number in (k0, v0)
            const int passes = std::max(1, (bits + 7) / 8);
            const uint32_t *sk;
            const uint64_t *sv;
            if (passes & 1) {
                SD_HIP(ctx, sdRadixSortPairs<uint64_t>(ctx->stream, k0.p, v0.p, k1.p, v1.p, k0.p, v0.p, (uint32_t) n, 0, bits, (uint32_t *) dTmp.p));
                sk = k1.p;
                sv = v1.p;
            } else {
                SD_HIP(ctx, sdRadixSortPairs<uint64_t>(ctx->stream, k0.p, v0.p, k0.p, v0.p, k1.p, v1.p, (uint32_t) n, 0, bits, (uint32_t *) dTmp.p));
                sk = k0.p;
                sv = v0.p;
            }
            const unsigned tiles = (unsigned) ((n + IK_TILE - 1) / IK_TILE);
            hipLaunchKernelGGL(ib_keep_count, dim3(tiles), dim3(IK_NT), 0, ctx->stream, sk, sv, n, dTileCount.p);
            SD_HIP(ctx, hipGetLastError());
            rc = scan32to64(ctx, dTileCount.p, tiles, dTileBase.p, dTot.p);
            if (rc != SD_OK) return rc;
            uint64_t kept = 0;
            SD_HIP(ctx, hipMemcpy(&kept, dTot.p, sizeof(kept), hipMemcpyDeviceToHost));
            hipLaunchKernelGGL(ib_place, dim3(tiles), dim3(IK_NT), 0, ctx->stream, sk, sv, n, (const uint64_t *) dTileBase.p, nEntries, lo,
                               t->dEntries, t->dOffsets);
            SD_HIP(ctx, hipGetLastError());
            SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            nEntries += kept;
        }
    }
    // ---------------- 3. list starts
    {
        Scoped<uint64_t> dStart, dTot;
        SD_HIP(ctx, dStart.alloc(t->tableSize + 1));
        SD_HIP(ctx, dTot.alloc(1));
        int rc = scan32to64(ctx, t->dOffsets, t->tableSize + 1, dStart.p, dTot.p);
        if (rc != SD_OK) return rc;
        uint64_t sum = 0;
        SD_HIP(ctx, hipMemcpy(&sum, dTot.p, sizeof(sum), hipMemcpyDeviceToHost));
        if (sum != nEntries) return sdFail(ctx, SD_EHIP, "sd_target_build: %llu entries placed, %llu counted", (unsigned long long) nEntries,
                                           (unsigned long long) sum);
        const int wide = nEntries > 0xFFFFFFFFull || getenv("SD_INDEX_WIDE") != nullptr;
        if (wide) SD_HIP(ctx, hipMalloc((void **) &t->dBlockBase, (((t->tableSize + 2) >> 16) + 1) * sizeof(uint64_t)));
        if (wide) SD_HIP(ctx, hipMemsetAsync(t->dBlockBase, 0, (((t->tableSize + 2) >> 16) + 1) * sizeof(uint64_t), ctx->stream));
        hipLaunchKernelGGL(ib_offsets, dim3((unsigned) ((t->tableSize + 1 + 255) / 256)), dim3(256), 0, ctx->stream, (const uint64_t *) dStart.p,
                           t->tableSize + 1, wide, t->dOffsets, t->dBlockBase);
        SD_HIP(ctx, hipGetLastError());
        SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    t->nEntries = nEntries;
    if (stats) {
        stats[0] = nEntries;
        stats[1] = nMaskedResidues;
        stats[2] = nPasses;
        stats[3] = nRecords;
    }
    guard.t = nullptr;
    *out = t;
    return SD_OK;
}

namespace {
// triple x = (k-mer, sequence, position): is it an entry of the index?  Lists are ordered by sequence.
__global__ void __launch_bounds__(256) ib_check_triples(const uint32_t *__restrict__ offsets, const uint64_t *__restrict__ blockBase,
                                                        const uint2 *__restrict__ entries, const uint32_t *__restrict__ kmer,
                                                        const uint32_t *__restrict__ seq, const uint32_t *__restrict__ pos, uint64_t n,
                                                        unsigned long long *__restrict__ missing) {
    const uint64_t x = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;
    const uint64_t km = kmer[x];
    uint64_t lo = (blockBase ? blockBase[km >> 16] : 0) + offsets[km];
    uint64_t hi = (blockBase ? blockBase[(km + 1) >> 16] : 0) + offsets[km + 1];
    const uint32_t want = seq[x];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (entries[mid].x < want) lo = mid + 1;
        else hi = mid;
    }
    const uint64_t end = (blockBase ? blockBase[(km + 1) >> 16] : 0) + offsets[km + 1];
    if (!(lo < end && entries[lo].x == want && entries[lo].y == pos[x])) atomicAdd(missing, 1ull);
}
// number of entries that belong to one of the (ascending) sample sequences
__global__ void __launch_bounds__(256) ib_count_sample(const uint2 *__restrict__ entries, uint64_t nEntries, const uint32_t *__restrict__ sample,
                                                       uint32_t nSample, unsigned long long *__restrict__ count) {
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < nEntries; i += (uint64_t) gridDim.x * 256) {
        const uint32_t s = entries[i].x;
        uint32_t lo = 0, hi = nSample;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sample[mid] < s) lo = mid + 1;
            else hi = mid;
        }
        mine += (lo < nSample && sample[lo] == s) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}
}  // namespace

// Sampled check of an index too large to download (10^4 proteomes: 72 GB of entries): for n (k-mer, sequence, position)
// triples -- the complete entries of the nSample sequences `sample` (ascending), computed elsewhere -- *missing = how many are
// not entries of the index, *inSample = how many entries of the index belong to the sample sequences (equal to n exactly when
// the index holds nothing else for them); the masked residues of [resBegin, resEnd) are copied to maskedOut (nullable).
extern "C" int sd_target_sample_check(sd_ctx *ctx, const sd_target *t, const uint32_t *kmer, const uint32_t *seq, const uint32_t *pos,
                                      uint64_t n, const uint32_t *sample, uint32_t nSample, uint64_t *missing, uint64_t *inSample,
                                      uint64_t resBegin, uint64_t resEnd, uint8_t *maskedOut) {
    if (!ctx || !t || !missing || !inSample || (n && (!kmer || !seq || !pos)) || (nSample && !sample)) return SD_EINVAL;
    (void) hipSetDevice(ctx->device);
    Scoped<uint32_t> dK, dS, dP, dSample;
    Scoped<unsigned long long> dOut;
    SD_HIP(ctx, dK.alloc(n));
    SD_HIP(ctx, dS.alloc(n));
    SD_HIP(ctx, dP.alloc(n));
    SD_HIP(ctx, dSample.alloc(nSample));
    SD_HIP(ctx, dOut.alloc(2));
    SD_HIP(ctx, hipMemcpy(dK.p, kmer, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    SD_HIP(ctx, hipMemcpy(dS.p, seq, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    SD_HIP(ctx, hipMemcpy(dP.p, pos, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    SD_HIP(ctx, hipMemcpy(dSample.p, sample, (size_t) nSample * sizeof(uint32_t), hipMemcpyHostToDevice));
    SD_HIP(ctx, hipMemsetAsync(dOut.p, 0, 2 * sizeof(unsigned long long), ctx->stream));
    if (n)
        hipLaunchKernelGGL(ib_check_triples, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t *) t->dOffsets,
                           (const uint64_t *) t->dBlockBase, (const uint2 *) t->dEntries, (const uint32_t *) dK.p, (const uint32_t *) dS.p,
                           (const uint32_t *) dP.p, n, dOut.p);
    if (t->nEntries && nSample)
        hipLaunchKernelGGL(ib_count_sample, dim3(8192), dim3(256), 0, ctx->stream, (const uint2 *) t->dEntries, t->nEntries,
                           (const uint32_t *) dSample.p, nSample, dOut.p + 1);
    SD_HIP(ctx, hipGetLastError());
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long h[2] = {0, 0};
    SD_HIP(ctx, hipMemcpy(h, dOut.p, sizeof(h), hipMemcpyDeviceToHost));
    *missing = h[0];
    *inSample = h[1];
    if (maskedOut && resEnd > resBegin) SD_HIP(ctx, hipMemcpy(maskedOut, t->dMasked + resBegin, resEnd - resBegin, hipMemcpyDeviceToHost));
    return SD_OK;
}

// the pieces of a target as the device holds them (tests: device-built index against the host-built one).  Any pointer may
// be NULL; entries = nEntries x (sequence id, position); starts = absolute list starts, tableSize + 1 of them
extern "C" int sd_target_download(sd_ctx *ctx, const sd_target *t, uint64_t *nEntries, uint64_t *tableSize, uint8_t *masked,
                                  uint64_t *starts, uint32_t *entrySeq, uint16_t *entryPos) {
    if (!ctx || !t) return SD_EINVAL;
    if (nEntries) *nEntries = t->nEntries;
    if (tableSize) *tableSize = t->tableSize;
    if (masked) SD_HIP(ctx, hipMemcpy(masked, t->dMasked, t->hSeqOff.back(), hipMemcpyDeviceToHost));
    if (starts) {
        std::vector<uint32_t> off(t->tableSize + 1);
        SD_HIP(ctx, hipMemcpy(off.data(), t->dOffsets, off.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        std::vector<uint64_t> bb;
        if (t->dBlockBase) {
            bb.resize(((t->tableSize + 2) >> 16) + 1);
            SD_HIP(ctx, hipMemcpy(bb.data(), t->dBlockBase, bb.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        }
        for (uint64_t i = 0; i <= t->tableSize; i++) starts[i] = (bb.empty() ? 0 : bb[i >> 16]) + off[i];
    }
    if (entrySeq || entryPos) {
        const uint64_t CH = 1ull << 24;
        std::vector<uint2> buf(CH);
        for (uint64_t b = 0; b < t->nEntries; b += CH) {
            const uint64_t n = std::min(CH, t->nEntries - b);
            SD_HIP(ctx, hipMemcpy(buf.data(), t->dEntries + b, n * sizeof(uint2), hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n; i++) {
                if (entrySeq) entrySeq[b + i] = buf[i].x;
                if (entryPos) entryPos[b + i] = (uint16_t) buf[i].y;
            }
        }
    }
    return SD_OK;
}
