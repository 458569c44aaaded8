// k-mer prefilter on gfx950 (QueryMatcher::matchQuery, M/src/prefiltering/QueryMatcher.cpp:85-211) for a
// batch of queries.  Integer / byte work bound by HBM (and the 256 MB Infinity Cache for the u32 k-mer
// offset table); no MFMA.  Stages (SURVEY.md A.1 step numbers):
//   K1 count_kmers   (3-4)  one wavefront per query position: branch-and-bound product of the two sorted
//                           3-mer rows, counted with per-lane binary searches
//   K2 emit_kmers    (4-5)  same enumeration, writes every similar k-mer with its index-list (start,len)
//   K3 gather_hits   (5)    copies the (seqId,pos) index lists into the hit stream: THE HBM-bound stream
//                           (6 B read + 6 B written per hit, coalesced within a list)
//   sort             (6)    stable LSD radix sort of the stream by (query,target) -- keeps stream order
//                           inside a target, which is what the double-diagonal state machine consumes
//   K4 match_diag    (6)    per hit: same low-8-bit diagonal as the previous hit of the target, and not the
//                           same as the previous *matched* one (CacheFriendlyOperations.cpp:185-272)
//   K5 score_diag    (7)    ungapped Kadane score along each candidate diagonal on the masked targets
//   K6 keep_max      (8)    first element with the per-target maximum (CacheFriendlyOperations.cpp:350-380)
//   K7 select_hits   (9-11) per query: score histogram, cut at maxHits, order of the cut = (score desc,
//                           bin = seqId & (BINSIZE-1), stream position), final (score desc, seqId) order,
//                           coverage pre-filter (Prefiltering.cpp:856-863)
// Between the hit stream and the match sits the hot-target filter (hot_filter_kernel): hits of targets that cannot emit a
// candidate are dropped after one look.  The reference's overflow path (> 2*max(1e6,dbSize) hits per query), the rescoring
// path (threshold >= 255) and result lists of any length (select_hits_big_kernel) are computed; what the device does not
// compute is reported per query (outCount = UINT32_MAX) -- never silently approximated.
#include <memory>
#include "sd_common.h"


#include <algorithm>
#include <cstring>


namespace {

// device buffer view backed by the context's grow-only workspace (same .p / .alloc() surface as DevBuf)
template <typename T>
struct WsView {
    sd_ctx *ctx;
    const char *key;
    T *p = nullptr;
    size_t n = 0;
    WsView(sd_ctx *c, const char *k) : ctx(c), key(k) {}
    hipError_t alloc(size_t count) {
        n = count;
        return wsGet(ctx, key, count, &p);
    }
};

constexpr int SPAN6 = 10;
__constant__ uint8_t c_seed6[6] = {0, 1, 3, 5, 8, 9};   // spaced seed 1101010011 (M/src/commons/Sequence.h:23)

// number of entries >= cutoff in a descending row of n int16
__device__ __forceinline__ int countGE(const int16_t *__restrict__ row, int n, int cutoff) {
    int lo = 0, hi = n;   // first index with row[idx] < cutoff
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if ((int) row[mid] >= cutoff) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

struct PosInfo {
    uint32_t q;
    int i;
    bool ok;
    uint32_t idx0, idx1;
    int thr;
};

__device__ __forceinline__ PosInfo decodePos(uint64_t p, const uint64_t *__restrict__ posBase, uint32_t nQ,
                                             const uint8_t *__restrict__ qRes, const uint64_t *__restrict__ qOff,
                                             const int16_t *__restrict__ kmerBias, int kmerThr,
                                             const uint32_t *__restrict__ posQuery = nullptr /* pos_query_kernel's answer */) {
    PosInfo r;
    // binary search: last q with posBase[q] <= p -- eleven dependent reads in front of everything else a wavefront does for its
    // position; the hot kernels take the answer from posQuery (computed once per sub-batch, 64 positions per wavefront)
    uint32_t lo = 0, hi = nQ;
    if (posQuery) {
        lo = posQuery[p];
    } else {
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (posBase[mid] <= p) lo = mid;
            else hi = mid;
        }
    }
    r.q = lo;
    r.i = (int) (p - posBase[lo]);
    const uint8_t *s = qRes + qOff[lo] + r.i;
    uint8_t w[6];
    bool hasX = false;
#pragma unroll
    for (int x = 0; x < 6; x++) {
        w[x] = s[c_seed6[x]];
        hasX |= (w[x] >= 20);
    }
    r.ok = !hasX;
    r.idx0 = w[0] + 20u * w[1] + 400u * w[2];
    r.idx1 = w[3] + 20u * w[4] + 400u * w[5];
    int b = kmerBias[qOff[lo] + r.i];
    int t = kmerThr - b;
    r.thr = t > 0 ? t : 0;
    return r;
}

// Index list of a k-mer: (start, length).  WIDE (indexes of 2^32 entries and more): the 32-bit slot is relative to a 64-bit
// base per 65 536 k-mers; the high half of the start goes to a second stream array.
template <bool WIDE>
__device__ __forceinline__ void idxList(const uint32_t *__restrict__ off, const uint64_t *__restrict__ base, uint32_t km,
                                        uint32_t &startLo, uint32_t &startHi, uint32_t &len) {
    const uint32_t s = off[km], e = off[km + 1];
    if (WIDE) {
        const uint64_t S = base[km >> 16] + s, E = base[(km + 1) >> 16] + e;
        startLo = (uint32_t) S;
        startHi = (uint32_t) (S >> 32);
        len = (uint32_t) (E - S);
    } else {
        startLo = s;
        startHi = 0;
        len = e - s;
    }
}

// query of every position of a sub-batch (last q with posBase[q] <= p), a thread per position
__global__ void __launch_bounds__(256)
pos_query_kernel(uint64_t nPos, const uint64_t *__restrict__ posBase, uint32_t nQ, uint32_t *__restrict__ posQuery) {
    const uint64_t p = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    if (p >= nPos) return;
    uint32_t lo = 0, hi = nQ;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (posBase[mid] <= p) lo = mid;
        else hi = mid;
    }
    posQuery[p] = lo;
}

// countGE of a whole 3-mer row from the cumulative table (sd_target::dExt3Cum): one read instead of a 13-step search.  Exactly the
// same number -- the rows are sorted by descending score, so "first index below the cutoff" is "entries at or above it".
constexpr int EXT3_CUM_SPAN = 256;
__device__ __forceinline__ int countGETab(const uint16_t *__restrict__ cumRow, int lo, int cutoff) {
    const int c = cutoff - lo;
    return c <= 0 ? 8000 : (c >= EXT3_CUM_SPAN ? 0 : (int) cumRow[c]);
}

__global__ void ext3_minmax_kernel(const int16_t *__restrict__ score, uint64_t n, int *__restrict__ mm /* [min, max] */) {
    int lo = 32767, hi = -32768;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const int v = score[i];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

// one workgroup per row: histogram of the row's scores, then cum[c] = entries with score >= lo + c
__global__ void __launch_bounds__(256)
ext3_cum_kernel(const int16_t *__restrict__ score, int lo, uint16_t *__restrict__ cum) {
    __shared__ uint32_t hist[EXT3_CUM_SPAN];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int16_t *row = score + (size_t) blockIdx.x * 8000;
    for (int j = threadIdx.x; j < 8000; j += 256) atomicAdd(&hist[(int) row[j] - lo], 1u);
    __syncthreads();
    uint32_t sum = 0;
    for (int v = threadIdx.x; v < EXT3_CUM_SPAN; v++) sum += hist[v];
    cum[(size_t) blockIdx.x * EXT3_CUM_SPAN + threadIdx.x] = (uint16_t) sum;
}

// K1: count similar k-mers per position
__global__ void __launch_bounds__(256)
count_kmers_kernel(uint64_t nPos, const uint64_t *__restrict__ posBase, uint32_t nQ, const uint8_t *__restrict__ qRes,
                   const uint64_t *__restrict__ qOff, const int16_t *__restrict__ kmerBias, int kmerThr,
                   const int16_t *__restrict__ ext3Score, uint32_t *__restrict__ kmerCount,
                   const uint16_t *__restrict__ ext3Cum /* nullable */, int ext3Lo, const uint32_t *__restrict__ posQuery /* nullable */) {
    const uint64_t p = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= nPos) return;
    PosInfo pi = decodePos(p, posBase, nQ, qRes, qOff, kmerBias, kmerThr, posQuery);
    uint32_t total = 0;
    if (pi.ok) {
        const int16_t *row0 = ext3Score + (size_t) pi.idx0 * 8000;
        const int16_t *row1 = ext3Score + (size_t) pi.idx1 * 8000;
        const int best1 = row1[0];
        const int cutoff1 = (int) (short) (pi.thr - best1);
        if (ext3Cum) {
            const uint16_t *cum0 = ext3Cum + (size_t) pi.idx0 * EXT3_CUM_SPAN, *cum1 = ext3Cum + (size_t) pi.idx1 * EXT3_CUM_SPAN;
            const int n0 = countGETab(cum0, ext3Lo, cutoff1);
            for (int a = lane; a < n0; a += 64) total += (uint32_t) countGETab(cum1, ext3Lo, (int) (short) (pi.thr - (int) row0[a]));
        } else {
            const int n0 = countGE(row0, 8000, cutoff1);
            for (int a = lane; a < n0; a += 64) {
                const int cutoff2 = (int) (short) (pi.thr - (int) row0[a]);
                total += (uint32_t) countGE(row1, 8000, cutoff2);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
    if (lane == 0) kmerCount[p] = total;
}

// grid-stride: a dispatch carries its size in work-items as 32 bits, and an index of 2^32 entries and more has more entries
// than that
__global__ void interleave_entries_kernel(uint64_t n, const uint32_t *__restrict__ seq, const uint16_t *__restrict__ pos,
                                          uint2 *__restrict__ out) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = make_uint2(seq[i], (uint32_t) pos[i]);
}

// ---- k = 7 (targets >= 3.35e9 residues, IndexTable.h:439-449): spaced seed 11010110011 (span 11, Sequence.h:24), divide
// strategy [2,2,3] after the reversal at KmerGenerator.cpp:84-85.  generateKmerList's two array products
// (KmerGenerator.cpp:107-216) nest as  for a in list0: for b in list1(a): for c in list2(a,b)  with the `short` cutoffs
//   a: s0[a] >= thr - best1 - best2;  b: s1[b] >= thr - s0[a] - best2;  c: s2[c] >= thr - (s0[a] + s1[b])
// and k-mer index i0[a] + 400 i1[b] + 160000 i2[c]; that loop order is the order of the hit stream.
constexpr int SPAN7 = 11;
__constant__ uint8_t c_seed7[7] = {0, 1, 3, 5, 6, 9, 10};

struct PosInfo7 {
    uint32_t q;
    int i;
    bool ok;
    uint32_t idx0, idx1, idx2;
    int thr;
};

__device__ __forceinline__ PosInfo7 decodePos7(uint64_t p, const uint64_t *__restrict__ posBase, uint32_t nQ,
                                               const uint8_t *__restrict__ qRes, const uint64_t *__restrict__ qOff,
                                               const int16_t *__restrict__ kmerBias, int kmerThr,
                                               const uint32_t *__restrict__ posQuery = nullptr) {
    PosInfo7 r;
    uint32_t lo = 0, hi = nQ;
    if (posQuery) {
        lo = posQuery[p];
    } else {
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (posBase[mid] <= p) lo = mid;
            else hi = mid;
        }
    }
    r.q = lo;
    r.i = (int) (p - posBase[lo]);
    const uint8_t *s = qRes + qOff[lo] + r.i;
    uint8_t w[7];
    bool hasX = false;
#pragma unroll
    for (int x = 0; x < 7; x++) {
        w[x] = s[c_seed7[x]];
        hasX |= (w[x] >= 20);
    }
    r.ok = !hasX;
    r.idx0 = w[0] + 20u * w[1];
    r.idx1 = w[2] + 20u * w[3];
    r.idx2 = w[4] + 20u * w[5] + 400u * w[6];
    int b = kmerBias[qOff[lo] + r.i];
    int t = kmerThr - b;
    r.thr = t > 0 ? t : 0;
    return r;
}

__global__ void __launch_bounds__(256)
count_kmers7_kernel(uint64_t nPos, const uint64_t *__restrict__ posBase, uint32_t nQ, const uint8_t *__restrict__ qRes,
                    const uint64_t *__restrict__ qOff, const int16_t *__restrict__ kmerBias, int kmerThr,
                    const int16_t *__restrict__ ext2Score, const int16_t *__restrict__ ext3Score,
                    uint32_t *__restrict__ kmerCount, const uint16_t *__restrict__ ext3Cum /* nullable */, int ext3Lo,
                    const uint32_t *__restrict__ posQuery /* nullable */) {
    const uint64_t p = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= nPos) return;
    PosInfo7 pi = decodePos7(p, posBase, nQ, qRes, qOff, kmerBias, kmerThr, posQuery);
    uint32_t total = 0;
    if (pi.ok) {
        const int16_t *row0 = ext2Score + (size_t) pi.idx0 * 400;
        const int16_t *row1 = ext2Score + (size_t) pi.idx1 * 400;
        const int16_t *row2 = ext3Score + (size_t) pi.idx2 * 8000;
        const int best2 = row2[0];
        const int rest0 = (int) (short) ((int) row1[0] + best2);
        const int n0 = countGE(row0, 400, (int) (short) (pi.thr - rest0));
        for (int a = 0; a < n0; a++) {
            const int sa = row0[a];
            const int nb = countGE(row1, 400, (int) (short) (pi.thr - sa - best2));
            for (int b = lane; b < nb; b += 64) {
                const int sab = (int) (short) (sa + (int) row1[b]);
                const int cutoff3 = (int) (short) (pi.thr - sab);
                total += (uint32_t) (ext3Cum ? countGETab(ext3Cum + (size_t) pi.idx2 * EXT3_CUM_SPAN, ext3Lo, cutoff3) : countGE(row2, 8000, cutoff3));
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
    if (lane == 0) kmerCount[p] = total;
}

template <bool WIDE>
__global__ void __launch_bounds__(256)
emit_kmers7_kernel(uint64_t nPos, const uint64_t *__restrict__ posBase, uint32_t nQ, const uint8_t *__restrict__ qRes,
                   const uint64_t *__restrict__ qOff, const int16_t *__restrict__ kmerBias, int kmerThr,
                   const int16_t *__restrict__ ext2Score, const uint16_t *__restrict__ ext2Index,
                   const int16_t *__restrict__ ext3Score, const uint16_t *__restrict__ ext3Index,
                   const uint32_t *__restrict__ idxOffsets, const uint64_t *__restrict__ kmerBase,
                   uint32_t *__restrict__ kStart, uint32_t *__restrict__ kLen, uint32_t *__restrict__ kPos,
                   const uint64_t *__restrict__ blockBase, uint32_t *__restrict__ kStartHi,
                   const uint16_t *__restrict__ ext3Cum /* nullable: countGETab */, int ext3Lo,
                   const uint32_t *__restrict__ posQuery /* nullable */) {
    const uint64_t p = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= nPos) return;
    PosInfo7 pi = decodePos7(p, posBase, nQ, qRes, qOff, kmerBias, kmerThr, posQuery);
    if (!pi.ok) return;
    const int16_t *row0 = ext2Score + (size_t) pi.idx0 * 400;
    const int16_t *row1 = ext2Score + (size_t) pi.idx1 * 400;
    const int16_t *row2 = ext3Score + (size_t) pi.idx2 * 8000;
    const uint16_t *ix0 = ext2Index + (size_t) pi.idx0 * 400;
    const uint16_t *ix1 = ext2Index + (size_t) pi.idx1 * 400;
    const uint16_t *ix2 = ext3Index + (size_t) pi.idx2 * 8000;
    const int best2 = row2[0];
    const int rest0 = (int) (short) ((int) row1[0] + best2);
    const int n0 = countGE(row0, 400, (int) (short) (pi.thr - rest0));
    uint64_t base = kmerBase[p];
    const uint32_t qi = (pi.q << 16) | (uint32_t) pi.i;
    // runs flattened over the wavefront as in emit_kmers_kernel: slot t of a chunk -> owning lane by binary search
    __shared__ uint32_t sIncl[4][64];
    __shared__ uint32_t sKab[4][64];
    uint32_t *myIncl = sIncl[threadIdx.x >> 6], *myKab = sKab[threadIdx.x >> 6];
    for (int a = 0; a < n0; a++) {
        const int sa = row0[a];
        const uint32_t ka = ix0[a];
        const int nb = countGE(row1, 400, (int) (short) (pi.thr - sa - best2));
        for (int b0 = 0; b0 < nb; b0 += 64) {
            const int b = b0 + lane;
            uint32_t c = 0;
            if (b < nb) {
                const int sab = (int) (short) (sa + (int) row1[b]);
                const int cutoff3 = (int) (short) (pi.thr - sab);
                c = (uint32_t) (ext3Cum ? countGETab(ext3Cum + (size_t) pi.idx2 * EXT3_CUM_SPAN, ext3Lo, cutoff3) : countGE(row2, 8000, cutoff3));
            }
            uint32_t incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            const uint32_t chunkTotal = __shfl(incl, 63, 64);
            __builtin_amdgcn_wave_barrier();
            myIncl[lane] = incl;
            myKab[lane] = b < nb ? ka + 400u * (uint32_t) ix1[b] : 0u;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t t0 = 0; t0 < chunkTotal; t0 += 256) {
                uint32_t km[4], s4[4], e4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t0 + (uint32_t) u * 64 + (uint32_t) lane;
                    km[u] = 0;
                    if (t < chunkTotal) {
                        int lo = 0, hi = 63;
#pragma unroll
                        for (int st = 0; st < 6; st++) {
                            const int mid = (lo + hi) >> 1;
                            if (myIncl[mid] > t) hi = mid;
                            else lo = mid + 1;
                        }
                        const uint32_t before = lo ? myIncl[lo - 1] : 0u;
                        km[u] = myKab[lo] + 160000u * (uint32_t) ix2[t - before];
                    }
                }
                uint32_t h4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) idxList<WIDE>(idxOffsets, blockBase, km[u], s4[u], h4[u], e4[u]);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t t = t0 + (uint32_t) u * 64 + (uint32_t) lane;
                    if (t < chunkTotal) {
                        kStart[base + t] = s4[u];
                        if (WIDE) kStartHi[base + t] = h4[u];
                        kLen[base + t] = e4[u];
                        kPos[base + t] = qi;
                    }
                }
            }
            base += chunkTotal;
        }
    }
}

// ---- profile queries (Sequence::nextProfileKmer, Sequence.cpp:294-305; KmerGenerator::setDivideStrategy(ScoreMatrix**),
// KmerGenerator.cpp:30-39): every seed position contributes its own list of 20 (score, amino acid) pairs sorted by
// descending score, divide strategy 1+1+...+1.  generateKmerList's k-1 array products (:143-165) are run stage by
// stage like the reference does, a wavefront per query position: the partial list lives in a per-wavefront scratch
// buffer (L2 resident), lanes take parents, each appends its run of children after an exclusive scan of the run
// lengths -- parent-major, child-minor, which is the reference's order.  The last product writes the k-mer stream
// (EMIT) or only counts it.  Cutoffs are the reference's `short` expressions.
// Two tiers of scratch: every position first runs with PROFILE_PARTIAL_CAP partial k-mers per stage on a wide grid; the
// few positions that outgrow it (a partial list is never longer than the final one, and the reference's own buffer
// holds 2^23 k-mers, KmerGenerator.h:45) are flagged and redone by a narrow grid with PROFILE_PARTIAL_CAP_BIG.
constexpr uint32_t PROFILE_PARTIAL_CAP = 1u << 16;
constexpr uint32_t PROFILE_PARTIAL_CAP_BIG = 1u << 23;
constexpr uint32_t PROFILE_GRID = 2048, PROFILE_GRID_BIG = 16;

template <bool EMIT>
__global__ void __launch_bounds__(64)
profile_kmers_kernel(uint64_t nPos, const uint64_t *__restrict__ posBase, uint32_t nQ, const uint8_t *__restrict__ qLetters,
                     const uint64_t *__restrict__ qOff, const int16_t *__restrict__ sortedScore,
                     const uint8_t *__restrict__ sortedIndex, int kmerThr, int k, uint2 *__restrict__ scratch,
                     uint32_t *__restrict__ kmerCount, const uint32_t *__restrict__ idxOffsets,
                     const uint64_t *__restrict__ kmerBase, uint32_t *__restrict__ kStart, uint32_t *__restrict__ kLen,
                     uint32_t *__restrict__ kPos, int *__restrict__ errFlag, uint32_t cap /* scratch entries per list */,
                     uint8_t *__restrict__ big /* per position: 1 = outgrew the small tier */, int bigTier,
                     const uint64_t *__restrict__ blockBase /* nullable: wide index */, uint32_t *__restrict__ kStartHi,
                     uint64_t *__restrict__ joinElems /* EMIT, nullable: the k-mer-major join's stream element instead of the three lookup
                                                         arrays (round 6: profile queries take the join, sd_pf_join.h) -- no index access here */) {
    __shared__ int16_t rowS[7][20];
    __shared__ uint8_t rowI[7][20];
    const int lane = threadIdx.x;
    uint2 *listA = scratch + (size_t) blockIdx.x * 2 * cap;
    uint2 *listB = listA + cap;
    const uint8_t *seed = k == 6 ? c_seed6 : c_seed7;
    const int thr = kmerThr > 0 ? kmerThr : 0;   // no composition bias for profiles (QueryMatcher.cpp:93-99,237-238)
    for (uint64_t p = blockIdx.x; p < nPos; p += gridDim.x) {
        if ((big[p] != 0) != (bigTier != 0)) continue;   // block-uniform
        uint32_t lo = 0, hi = nQ;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (posBase[mid] <= p) lo = mid;
            else hi = mid;
        }
        const uint32_t q = lo;
        const int i = (int) (p - posBase[lo]);
        const uint64_t r0 = qOff[q] + (uint64_t) i;
        bool hasX = false;
        for (int x = 0; x < k; x++) hasX |= qLetters[r0 + seed[x]] >= 20;   // kmerContainsX on the query letters (Sequence.h:397-403)
        __syncthreads();   // the previous position's rows are no longer read
        for (int x = lane; x < k * 20; x += 64) {
            const int sIdx = x / 20, e = x % 20;
            rowS[sIdx][e] = sortedScore[(r0 + seed[sIdx]) * 20 + e];
            rowI[sIdx][e] = sortedIndex[(r0 + seed[sIdx]) * 20 + e];
        }
        __syncthreads();
        uint32_t totalOut = 0;
        if (!hasX) {
            int rest[7];
            rest[k - 1] = 0;
            for (int x = k - 1; x >= 1; x--) rest[x - 1] = (int) (short) ((int) rowS[x][0] + rest[x]);
            const int cutoff1 = (int) (short) (thr - rest[0]);
            uint32_t nA = 0;
            while (nA < 20 && (int) rowS[0][nA] >= cutoff1) nA++;
            if ((uint32_t) lane < nA) listA[lane] = make_uint2((uint32_t) (int) rowS[0][lane], (uint32_t) rowI[0][lane]);
            uint32_t mult = 1;
            uint64_t base = EMIT ? kmerBase[p] : 0;
            const uint32_t qi = (q << 16) | (uint32_t) i;
            bool overflow = false;
            for (int st = 0; st + 1 < k && !overflow; st++) {
                mult *= 20u;
                const bool last = st + 2 == k;
                const int16_t *rs = rowS[st + 1];
                const uint8_t *ri = rowI[st + 1];
                const int restNext = rest[st + 1];
                uint32_t nB = 0;
                __threadfence_block();
                for (uint32_t c0 = 0; c0 < nA; c0 += 64) {
                    const uint32_t e = c0 + lane;
                    int si = 0;
                    uint32_t ki = 0, c = 0;
                    if (e < nA) {
                        // agent-scope loads: the entry was written by another lane of this wavefront a stage ago and the
                        // vector L1 may still hold the line from the buffer's previous use
                        const uint32_t px = __hip_atomic_load(&listA[e].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ki = __hip_atomic_load(&listA[e].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        si = (int) (short) px;
                        const int cutoff2 = (int) (short) (thr - si - restNext);
                        while (c < 20 && (int) rs[c] >= cutoff2) c++;
                    }
                    uint32_t incl = c;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        uint32_t o = __shfl_up(incl, off, 64);
                        if (lane >= off) incl += o;
                    }
                    const uint32_t excl = incl - c;
                    const uint32_t chunkTotal = __shfl(incl, 63, 64);
                    if (!last) {
                        if (nB + chunkTotal > cap) {
                            overflow = true;
                            break;
                        }
                        for (uint32_t j = 0; j < c; j++)
                            listB[nB + excl + j] = make_uint2((uint32_t) (int) (short) (si + (int) rs[j]), ki + (uint32_t) ri[j] * mult);
                    } else if (EMIT) {
                        const uint64_t w = base + nB + excl;
                        if (joinElems) {   // kmer << 38 | stream index << 8 | low byte of the query position (jpElem, sd_pf_join.h)
                            for (uint32_t j = 0; j < c; j++) {
                                const uint32_t kmer = ki + (uint32_t) ri[j] * mult;
                                joinElems[w + j] = ((uint64_t) kmer << 38) | ((w + j) << 8) | (uint64_t) (i & 0xFF);
                            }
                        } else
                        for (uint32_t j = 0; j < c; j++) {
                            const uint32_t kmer = ki + (uint32_t) ri[j] * mult;
                            uint32_t s0, h0, l0;
                            if (blockBase) {
                                idxList<true>(idxOffsets, blockBase, kmer, s0, h0, l0);
                                kStartHi[w + j] = h0;
                            } else {
                                idxList<false>(idxOffsets, blockBase, kmer, s0, h0, l0);
                            }
                            kStart[w + j] = s0;
                            kLen[w + j] = l0;
                            kPos[w + j] = qi;
                        }
                    }
                    nB += chunkTotal;
                }
                if (last) {
                    totalOut = nB;
                } else {
                    uint2 *t = listA;
                    listA = listB;
                    listB = t;
                    nA = nB;
                }
            }
            if (overflow) {   // only possible while counting: the emit pass runs a position in the tier that counted it
                if (lane == 0) {
                    if (!bigTier) big[p] = 1;
                    atomicMax(errFlag, bigTier ? 8 : 7);
                }
                totalOut = 0;
            }
        }
        if (!EMIT && lane == 0) kmerCount[p] = totalOut;
    }
}

// K2: emit k-mers (+ index list start/len, + owning position) in the reference's enumeration order
template <bool WIDE>
__global__ void __launch_bounds__(256)
emit_kmers_kernel(uint64_t nPos, const uint64_t *__restrict__ posBase, uint32_t nQ, const uint8_t *__restrict__ qRes,
                  const uint64_t *__restrict__ qOff, const int16_t *__restrict__ kmerBias, int kmerThr,
                  const int16_t *__restrict__ ext3Score, const uint16_t *__restrict__ ext3Index,
                  const uint32_t *__restrict__ idxOffsets, const uint64_t *__restrict__ kmerBase,
                  uint32_t *__restrict__ kStart, uint32_t *__restrict__ kLen, uint32_t *__restrict__ kPos,
                  const uint64_t *__restrict__ blockBase, uint32_t *__restrict__ kStartHi,
                   const uint16_t *__restrict__ ext3Cum /* nullable: countGETab */, int ext3Lo,
                   const uint32_t *__restrict__ posQuery /* nullable */) {
    const uint64_t p = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= nPos) return;
    PosInfo pi = decodePos(p, posBase, nQ, qRes, qOff, kmerBias, kmerThr, posQuery);
    if (!pi.ok) return;
    const int16_t *row0 = ext3Score + (size_t) pi.idx0 * 8000;
    const int16_t *row1 = ext3Score + (size_t) pi.idx1 * 8000;
    const uint16_t *ix0 = ext3Index + (size_t) pi.idx0 * 8000;
    const uint16_t *ix1 = ext3Index + (size_t) pi.idx1 * 8000;
    const int best1 = row1[0];
    const int cutoff1 = (int) (short) (pi.thr - best1);
    const int n0 = ext3Cum ? countGETab(ext3Cum + (size_t) pi.idx0 * EXT3_CUM_SPAN, ext3Lo, cutoff1) : countGE(row0, 8000, cutoff1);
    uint64_t base = kmerBase[p];
    const uint32_t qi = (pi.q << 16) | (uint32_t) pi.i;   // owning query (< 2^16 per sub-batch) and position (< 2^16)
    // The lanes first own prefixes (a) and count their suffix runs; the runs are then flattened: output slot t of the chunk
    // belongs to the lane o with excl[o] <= t < incl[o] (binary search over the 64 scan values in LDS), so every lane
    // produces k-mers -- the run lengths differ by orders of magnitude between the best and the last prefix -- and the
    // three stream writes are contiguous across the wavefront.  Four slots per lane keep eight index lookups in flight.
    __shared__ uint32_t sIncl[4][64];
    __shared__ uint32_t sK0[4][64];
    uint32_t *myIncl = sIncl[threadIdx.x >> 6], *myK0 = sK0[threadIdx.x >> 6];
    for (int a0 = 0; a0 < n0; a0 += 64) {
        const int a = a0 + lane;
        uint32_t c = 0;
        if (a < n0) {
            const int cutoff2 = (int) (short) (pi.thr - (int) row0[a]);
            c = (uint32_t) (ext3Cum ? countGETab(ext3Cum + (size_t) pi.idx1 * EXT3_CUM_SPAN, ext3Lo, cutoff2) : countGE(row1, 8000, cutoff2));
        }
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        const uint32_t chunkTotal = __shfl(incl, 63, 64);
        __builtin_amdgcn_wave_barrier();   // the previous chunk's readers are done
        myIncl[lane] = incl;
        myK0[lane] = a < n0 ? (uint32_t) ix0[a] : 0u;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t t0 = 0; t0 < chunkTotal; t0 += 256) {
            uint32_t km[4], s4[4], e4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t t = t0 + (uint32_t) u * 64 + (uint32_t) lane;
                km[u] = 0;
                if (t < chunkTotal) {
                    int lo = 0, hi = 63;   // smallest o with incl[o] > t
#pragma unroll
                    for (int st = 0; st < 6; st++) {
                        const int mid = (lo + hi) >> 1;
                        if (myIncl[mid] > t) hi = mid;
                        else lo = mid + 1;
                    }
                    const uint32_t before = lo ? myIncl[lo - 1] : 0u;
                    km[u] = myK0[lo] + 8000u * (uint32_t) ix1[t - before];
                }
            }
            uint32_t h4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) idxList<WIDE>(idxOffsets, blockBase, km[u], s4[u], h4[u], e4[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t t = t0 + (uint32_t) u * 64 + (uint32_t) lane;
                if (t < chunkTotal) {
                    kStart[base + t] = s4[u];
                    if (WIDE) kStartHi[base + t] = h4[u];
                    kLen[base + t] = e4[u];
                    kPos[base + t] = qi;
                }
            }
        }
        base += chunkTotal;
    }
}

// K3: gather index lists into the hit stream; key = (qLocal << tBits) | seqId, value = low 8 bits of the diagonal
// (all the double-diagonal match needs) << 24 | position of the hit in its query's stream; sub-batches holding a query with
// 2^24 hits and more set widePos: the value is then the whole position and the diagonal byte is moved into the key by
// coarse_scatter_kernel.  The full 16-bit diagonal stays in hitDiag, in stream order, and is looked up for the few
// surviving candidates only.
__global__ void __launch_bounds__(256)
gather_hits_kernel(uint64_t nKmers, const uint32_t *__restrict__ kStart, const uint32_t *__restrict__ kLen,
                   const uint32_t *__restrict__ kPos, const uint64_t *__restrict__ hitBase,
                   const uint64_t *__restrict__ posBase, uint32_t nQ, const uint2 *__restrict__ entries, int tBits,
                   const uint64_t *__restrict__ qHitBase,
                   uint32_t *__restrict__ hitKey, uint32_t *__restrict__ hitVal, uint16_t *__restrict__ hitDiag,
                   const uint32_t *__restrict__ kStartHi /* nullable: high half of the list starts of a wide index */,
                   int widePos /* the value word is the full stream position (queries with >= 2^24 hits); the diagonal byte is
                                  taken from hitDiag by coarse_scatter_kernel and moves into the key */) {
    const uint64_t kidx = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t len = 0, q = 0;
    uint64_t start = 0;
    int i = 0;
    uint64_t base = 0;
    if (kidx < nKmers) {
        len = kLen[kidx];
        if (len) {
            start = kStart[kidx];
            if (kStartHi) start |= (uint64_t) kStartHi[kidx] << 32;
            base = hitBase[kidx];
            const uint32_t qi = kPos[kidx];
            q = qi >> 16;
            i = (int) (qi & 0xFFFFu);
        }
    }
    // short lists: each lane copies its own; long lists: the wavefront copies them cooperatively
    const bool isLong = len > 8;
    if (len && !isLong) {
        // all (at most eight) entry reads are issued before the first is used
        uint32_t sid[8];
        uint16_t ep[8];
        const uint32_t rel = (uint32_t) (base - qHitBase[q]);
#pragma unroll
        for (uint32_t x = 0; x < 8; x++) {
            if (x < len) {
                const uint2 en = entries[start + x];
                sid[x] = en.x;
                ep[x] = (uint16_t) en.y;
            }
        }
#pragma unroll
        for (uint32_t x = 0; x < 8; x++) {
            if (x < len) {
                const uint16_t d = (uint16_t) (i - (int) ep[x]);
                hitKey[base + x] = (q << tBits) | sid[x];
                hitVal[base + x] = widePos ? rel + x : ((uint32_t) (d & 0xFF) << 24) | (rel + x);
                hitDiag[base + x] = d;
            }
        }
    }
    unsigned long long longMask = __ballot(isLong);
    while (longMask) {
        const int src = __ffsll((long long) longMask) - 1;
        longMask &= longMask - 1;
        const uint32_t l2 = __shfl(len, src, 64), q2 = __shfl(q, src, 64);
        const uint64_t s2 = ((uint64_t) __shfl((uint32_t) (start >> 32), src, 64) << 32) | __shfl((uint32_t) start, src, 64);
        const int i2 = __shfl(i, src, 64);
        const uint64_t b2 = ((uint64_t) __shfl((uint32_t) (base >> 32), src, 64) << 32) | __shfl((uint32_t) base, src, 64);
        for (uint32_t x = lane; x < l2; x += 64) {
            const uint2 en = entries[s2 + x];
            const uint32_t sid = en.x;
            const uint16_t d = (uint16_t) (i2 - (int) en.y);
            hitKey[b2 + x] = (q2 << tBits) | sid;
            const uint32_t rp = (uint32_t) (b2 + x - qHitBase[q2]);
            hitVal[b2 + x] = widePos ? rp : ((uint32_t) (d & 0xFF) << 24) | rp;
            hitDiag[b2 + x] = d;
        }
    }
}

// K4: double-diagonal match on the (query,target)-sorted stream
// A query whose hits overflow the reference's hit buffer is matched in two independent parts (QueryMatcher.cpp:281-316):
// split = stream position (inside the query) where the second part starts, 0xFFFFFFFF = no overflow.  The previous-hit
// state of a target does not carry over the split; the merge of the two result lists (:323-326) collapses equal
// consecutive diagonals across it like inside a part.
__device__ __forceinline__ bool flagA(uint64_t s, const uint32_t *__restrict__ key, const uint32_t *__restrict__ val,
                                      uint32_t split) {
    const uint8_t d8 = (uint8_t) (val[s] >> 24);
    bool first = (s == 0) || (key[s - 1] != key[s]);
    if (!first) first = ((val[s - 1] & 0xFFFFFFu) < split) != ((val[s] & 0xFFFFFFu) < split);
    const uint8_t prev = first ? (uint8_t) 0 : (uint8_t) (val[s - 1] >> 24);
    return d8 == prev;
}

__global__ void __launch_bounds__(256)
match_diag_kernel(uint64_t nHits, const uint32_t *__restrict__ key, const uint32_t *__restrict__ val,
                  const uint32_t *__restrict__ qSplit, int tBits, uint8_t *__restrict__ emit) {
    const uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nHits) return;
    const uint32_t split = qSplit[key[s] >> tBits];
    bool e = false;
    if (flagA(s, key, val, split)) {
        e = true;
        const uint8_t d8 = (uint8_t) (val[s] >> 24);
        uint64_t x = s;
        while (x > 0 && key[x - 1] == key[s]) {
            x--;
            if (flagA(x, key, val, split)) {
                e = ((uint8_t) (val[x] >> 24)) != d8;
                break;
            }
        }
    }
    emit[s] = e ? 1 : 0;
}

// hits of query q start at stream position qHitBase[q] (q = nQ: total)
__global__ void query_hit_base_kernel(uint32_t nQ, const uint64_t *__restrict__ posBase, const uint64_t *__restrict__ kmerBase,
                                      const uint64_t *__restrict__ hitBase, uint64_t *__restrict__ qHitBase) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q <= nQ) qHitBase[q] = hitBase[kmerBase[posBase[q]]];
}

constexpr uint32_t QUERY_UNSUPPORTED = 0xFFFFFFFEu;   // qSplit marker: the query needs a reference path the device lacks
constexpr int PF_SPLITS_MAX = 32;   // overflows of the reference's hit buffer carried per query (33 parts: 6.6*10^7 hits at the smallest buffer)

// where the reference's hit buffer (cap entries) overflows inside query q, as stream positions: every k-mer list that would
// fill the buffer (inBuffer + listSize >= cap) starts a new part (QueryMatcher.cpp:281-316); qSplit = the first one (the
// common case of one overflow is read from there), qSplits / qParts all of them
__global__ void query_split_kernel(uint32_t nQ, const uint64_t *__restrict__ posBase, const uint64_t *__restrict__ kmerBase,
                                   const uint64_t *__restrict__ hitBase, uint64_t cap, uint32_t *__restrict__ qSplit,
                                   uint32_t *__restrict__ qParts, uint32_t *__restrict__ qSplits /* [nQ][PF_SPLITS_MAX] */,
                                   int *__restrict__ flag, uint64_t posLimit /* 2^24, or ~2^32 with wide stream positions */) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nQ) return;
    const uint64_t k0 = kmerBase[posBase[q]], k1 = kmerBase[posBase[q + 1]];
    const uint64_t h0 = hitBase[k0], total = hitBase[k1] - h0;
    uint32_t first = 0xFFFFFFFFu, n = 0;
    bool unsupported = total >= posLimit;   // stream positions are carried in 24 (wide: 32) bits
    uint64_t cur = k0;   // first k-mer of the current part
    while (!unsupported && hitBase[k1] - hitBase[cur] >= cap) {
        // first k in [cur, k1) with hitBase[k + 1] - hitBase[cur] >= cap
        const uint64_t partBase = hitBase[cur];
        uint64_t lo = cur, hi = k1;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (hitBase[mid + 1] - partBase >= cap) hi = mid;
            else lo = mid + 1;
        }
        if (lo >= k1) break;
        const uint32_t pos = (uint32_t) (hitBase[lo] - h0);
        if (n < (uint32_t) PF_SPLITS_MAX) qSplits[(size_t) q * PF_SPLITS_MAX + n] = pos;
        if (n == 0) first = pos;
        n++;
        if (hitBase[lo + 1] - hitBase[lo] >= cap || n > (uint32_t) PF_SPLITS_MAX) unsupported = true;   // (a list as large as the buffer, :312-314)
        cur = lo;
    }
    if (unsupported) {
        atomicExch(flag, 1);
        first = QUERY_UNSUPPORTED;
        n = 0;
    }
    qSplit[q] = first;
    qParts[q] = n;
}

// queries marked QUERY_UNSUPPORTED take no part in the rest of the batch: their index lists are emptied
__global__ void __launch_bounds__(256)
drop_query_kmers_kernel(uint64_t nKmers, const uint32_t *__restrict__ kPos, const uint32_t *__restrict__ qSplit,
                        uint32_t *__restrict__ kLen) {
    const uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nKmers) return;
    if (kLen[k] && qSplit[kPos[k] >> 16] == QUERY_UNSUPPORTED) kLen[k] = 0;
}

#include "sd_pf_join.h"

// part of the reference's hit buffer a stream position (lookup path) / k-mer ordinal (join path) falls into: the number of
// splits at or before it
struct SplitView {
    uint32_t first;            // first split (0xFFFFFFFF: none)
    uint32_t n;                // number of splits
    const uint32_t *all;       // the query's splits when n >= 2
};
__device__ __forceinline__ SplitView splitView(const uint32_t *qSplit, const uint32_t *qParts, const uint32_t *qSplits, uint32_t q) {
    SplitView v;
    v.first = qSplit[q];
    v.n = qParts ? qParts[q] : 0u;
    v.all = v.n >= 2 ? qSplits + (size_t) q * PF_SPLITS_MAX : nullptr;
    return v;
}
__device__ __forceinline__ uint32_t partOfPos(const SplitView &v, uint32_t pos) {
    if (!v.all) return pos >= v.first ? 1u : 0u;
    uint32_t lo = 0, hi = v.n;   // number of splits <= pos
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v.all[mid] <= pos) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// Place of a hit in the reference's result list once the buffer has overflowed m >= 2 times (parts 0 .. m of the stream).
// Every merge of carried results with a new part (QueryMatcher.cpp:289-303) walks the bins backwards
// (CacheFriendlyOperations.cpp:119-147), so the carried list C_j after the j-th overflow reads
//     C_1 = part 0 forwards;   C_j = part j-1 backwards, then C_{j-1} backwards;
// and the final list is C_m followed by part m forwards (the last merge, :316-319, walks forwards).  Elements only ever drop
// out in between, so a survivor's place is that of its part in this sequence and its stream position inside the part, read
// in the part's direction; it is returned as a position of the same range (a permutation of the stream positions), with
// `backwards` telling whether hits of one position (join path: one k-mer's list) are read in reverse too.
__device__ __forceinline__ uint32_t overflowOrderPos(const SplitView &v, uint32_t pos, bool &backwards) {
    const uint32_t m = v.n;
    const uint32_t p = partOfPos(v, pos);
    backwards = false;
    if (!v.all || p >= m) return pos;
    auto place = [&](uint32_t part, bool &back) -> uint32_t {   // rank of a part (< m) in C_m, and its direction
        uint32_t r = 0;
        back = part != 0;
        for (uint32_t j = (part == 0 ? 2u : part + 2u); j <= m; j++) {
            r = 1 + (j - 2 - r);
            back = !back;
        }
        return r;
    };
    const uint32_t mine = place(p, backwards);
    uint32_t base = 0;
    for (uint32_t o = 0; o < m; o++) {
        bool b;
        if (o != p && place(o, b) < mine) base += v.all[o] - (o ? v.all[o - 1] : 0u);
    }
    const uint32_t lo = p ? v.all[p - 1] : 0u, hi = v.all[p];
    return base + (backwards ? hi - 1 - pos : pos - lo);
}

// ---------------------------------------------------------------------------------------------
// Bucketed double-diagonal match (replaces the global radix sort + match + flag scan + compaction).
// The hit stream is already grouped by query.  Per query one workgroup splits its segment into `bins` target
// ranges (a power of two chosen from the segment size so that a range holds ~10^3 hits), reordering every tile in
// LDS first so that the global writes are runs, not a scatter.  Per (query, range) bucket one workgroup then
// sorts in LDS -- counting sort on the target offset, ties put back into emission order with the stream position
// carried in the value -- runs the match on the sorted bucket and writes only the matched hits, compacted, to the
// front of the bucket's region.  Buckets are ordered (query, target range), so concatenating them gives the
// candidates in (query, target, emission) order directly.
// ---------------------------------------------------------------------------------------------
constexpr int PF_NB_MAX = 2048;        // bucket slots per query
constexpr int PF_LB_MAX = 11;          // log2(PF_NB_MAX)
constexpr int PF_BUCKET_CAP = 1024;    // hits the LDS bucket sort takes; larger buckets go to a second launch with
constexpr int PF_BUCKET_CAP_BIG = 6144;   // this capacity, and only beyond that the whole sub-batch falls back
constexpr int PF_FILL = 512;           // aimed hits per bucket: the bucket count is the next power of two of hits / PF_FILL
constexpr int PF_CNT_MAX = 4096;       // counting-sort bins (target offsets) per bucket

__device__ __forceinline__ int pfLog2Bins(uint64_t n, int tBits) {
    const uint64_t want = (n + PF_FILL - 1) / PF_FILL;
    int lb = 0;
    while ((1ull << lb) < want && lb < PF_LB_MAX) lb++;
    const int minLb = tBits > 12 ? tBits - 12 : 0;   // a bucket's target range must fit PF_CNT_MAX counters
    lb = lb < minLb ? minLb : lb;
    lb = lb > tBits ? tBits : lb;
    return lb;   // > PF_LB_MAX only when tBits > 12 + PF_LB_MAX: the caller falls back
}

// 16-bit counters packed two per LDS word (every count here is < 65536): add one / read one
__device__ __forceinline__ uint32_t pk16Add(uint32_t *w, uint32_t idx) {
    const uint32_t sh = (idx & 1u) * 16u;
    return (atomicAdd(&w[idx >> 1], 1u << sh) >> sh) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t pk16Get(const uint32_t *w, uint32_t idx) { return (w[idx >> 1] >> ((idx & 1u) * 16u)) & 0xFFFFu; }

// exclusive scan of n packed 16-bit counters (n even, n/2 words), in place; NT threads, part[] >= NT/64 + 1 words
template <int NT>
__device__ void pk16Scan(uint32_t *w, int n, uint32_t *part) {
    const int t = threadIdx.x, words = n >> 1;
    const int per = (words + NT - 1) / NT;
    const int b = t * per, e = min(words, b + per);
    uint32_t sum = 0;
    for (int x = b; x < e; x++) sum += (w[x] & 0xFFFFu) + (w[x] >> 16);
    // inclusive scan of the per-thread sums across the workgroup
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int wv = 0; wv < (t >> 6); wv++) base += part[wv];
    uint32_t run = base + incl - sum;
    for (int x = b; x < e; x++) {
        const uint32_t lo = w[x] & 0xFFFFu, hi = w[x] >> 16;
        w[x] = (run & 0xFFFFu) | (((run + lo) & 0xFFFFu) << 16);
        run += lo + hi;
    }
    __syncthreads();
}

// exclusive scan of arr[0..n) (32-bit, NT threads), in place
template <int NT>
__device__ void pfBlockScan(uint32_t *arr, int n, uint32_t *part /* >= NT / 64 + 1 */) {
    const int t = threadIdx.x;
    const int per = (n + NT - 1) / NT;
    const int b = t * per, e = min(n, b + per);
    uint32_t sum = 0;
    for (int x = b; x < e; x++) sum += arr[x];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int wv = 0; wv < (t >> 6); wv++) base += part[wv];
    uint32_t run = base + incl - sum;
    for (int x = b; x < e; x++) {
        const uint32_t v = arr[x];
        arr[x] = run;
        run += v;
    }
    __syncthreads();
}

// NT threads reorder TILE hits per step: the average run a tile writes into a bin is TILE / bins hits, and only runs of
// a cache line or more keep the stores from being partial-line writes (2 048-hit tiles over 512 bins: 32-byte runs,
// 1.8x the bytes at the memory side; 8 192: 128-byte runs)
template <int NT, int TILE>
__global__ void __launch_bounds__(NT)
partition_hits_kernel(uint32_t nQ, const uint64_t *__restrict__ qHitBase, int tBits, const uint32_t *__restrict__ inKey,
                      const uint32_t *__restrict__ inVal, const uint2 *__restrict__ inKV /* join path: interleaved input */,
                      uint2 *__restrict__ outKV /* (key, value) per hit */,
                      uint32_t *__restrict__ qLog2Bins, uint64_t *__restrict__ bktStart, uint32_t *__restrict__ bktCount,
                      int *__restrict__ flag,
                      const uint32_t *__restrict__ segCount /* nullable: hits of the segment that are left (hot_filter_kernel) */,
                      const uint64_t *__restrict__ outBase /* with segCount: where the segment's buckets start in outKV */,
                      const uint8_t *__restrict__ segDone /* nullable */) {
    constexpr int PER = TILE / NT;
    __shared__ uint32_t cursor[PF_NB_MAX];          // segment histogram, then the running write position per bin
    __shared__ uint32_t tcount[PF_NB_MAX / 2];      // per tile: packed 16-bit counts, then exclusive starts
    __shared__ uint32_t tsize[PF_NB_MAX / 2];       // per tile: packed 16-bit counts (kept for the cursor update)
    __shared__ uint32_t part[NT / 64 + 8];
    __shared__ uint32_t tileK[TILE], tileV[TILE];
    const uint32_t q = blockIdx.x;
    const int t = threadIdx.x;
    if (segDone && segDone[q]) return;   // matched as a whole by segment_match_kernel
    const uint64_t s = qHitBase[q], e = segCount ? s + segCount[q] : qHitBase[q + 1];
    const uint64_t n = e - s;
    const uint64_t o = segCount ? outBase[q] : s;
    int lb = pfLog2Bins(n, tBits);
    if (lb > PF_LB_MAX) {
        if (t == 0) {
            qLog2Bins[q] = (uint32_t) lb;
            if (n > 0) atomicExch(flag, 1);
        }
        return;
    }
    const uint32_t tMask = (1u << tBits) - 1;
    int bins, shift;
    uint32_t *over = &part[NT / 64 + 7];
    // histogram of the segment; a finer split is taken while some range holds more than an LDS bucket can sort
    for (;;) {
        bins = 1 << lb;
        shift = tBits - lb;
        for (int b = t; b < bins; b += NT) cursor[b] = 0;
        if (t == 0) *over = 0;
        __syncthreads();
        if (inKV)
            for (uint64_t i = s + t; i < e; i += NT) atomicAdd(&cursor[(inKV[i].x & tMask) >> shift], 1u);
        else
            for (uint64_t i = s + t; i < e; i += NT) atomicAdd(&cursor[(inKey[i] & tMask) >> shift], 1u);
        __syncthreads();
        uint32_t mx = 0;
        for (int b = t; b < bins; b += NT) mx = max(mx, cursor[b]);
        if (mx > (uint32_t) PF_BUCKET_CAP) atomicMax(over, mx);
        __syncthreads();
        const bool isOver = *over > (uint32_t) PF_BUCKET_CAP;
        __syncthreads();
        if (!isOver || lb >= PF_LB_MAX || lb >= tBits) break;
        lb++;
    }
    if (t == 0) qLog2Bins[q] = (uint32_t) lb;
    for (int b = t; b < bins; b += NT) bktCount[(size_t) q * PF_NB_MAX + b] = cursor[b];
    __syncthreads();
    pfBlockScan<NT>(cursor, bins, part);
    for (int b = t; b < bins; b += NT) bktStart[(size_t) q * PF_NB_MAX + b] = o + cursor[b];
    const int binsEven = bins < 2 ? 2 : bins;
    for (int x = t; x < binsEven / 2; x += NT) tcount[x] = 0;
    __syncthreads();
    for (uint64_t base = s; base < e; base += TILE) {
        const int tn = (int) min((uint64_t) TILE, e - base);
        uint32_t k[PER], v[PER], r[PER];
#pragma unroll
        for (int x = 0; x < PER; x++) {
            const int j = x * NT + t;
            if (j < tn) {
                if (inKV) {
                    const uint2 kv = inKV[base + j];
                    k[x] = kv.x;
                    v[x] = kv.y;
                } else {
                    k[x] = inKey[base + j];
                    v[x] = inVal[base + j];
                }
                r[x] = pk16Add(tcount, (k[x] & tMask) >> shift);
            }
        }
        __syncthreads();
        for (int x = t; x < binsEven / 2; x += NT) tsize[x] = tcount[x];
        __syncthreads();
        pk16Scan<NT>(tcount, binsEven, part);
#pragma unroll
        for (int x = 0; x < PER; x++) {
            const int j = x * NT + t;
            if (j < tn) {
                const uint32_t p = pk16Get(tcount, (k[x] & tMask) >> shift) + r[x];
                tileK[p] = k[x];
                tileV[p] = v[x];
            }
        }
        __syncthreads();
        for (int j = t; j < tn; j += NT) {
            const uint32_t kk = tileK[j];
            const uint32_t b = (kk & tMask) >> shift;
            const uint64_t g = o + cursor[b] + ((uint32_t) j - pk16Get(tcount, b));
            outKV[g] = make_uint2(kk, tileV[j]);
        }
        __syncthreads();
        for (int b = t; b < bins; b += NT) cursor[b] += pk16Get(tsize, b);
        __syncthreads();
        for (int x = t; x < binsEven / 2; x += NT) tcount[x] = 0;
        __syncthreads();
    }
}

// ---- coarse split for very hit-rich queries (large target sets: ~1.5*10^6 index hits per query at 3*10^6 targets).
// The per-query partition above has 2^11 bucket slots; beyond ~5*10^5 hits its buckets outgrow the LDS sort and one
// workgroup per query is too little parallelism.  So the query's segment is first split into C = 2^cBits target
// ranges by the top target bits -- many workgroups per query (segments of CP_SEG hits): count, offsets, scatter --
// and every (query, range) then goes through the partition / bucket machinery as a "virtual query" whose keys have
// cBits fewer target bits.  Order inside a range is not kept (the bucket sort restores emission order from the
// stream position in the value); (query, range, bucket) order is (query, target) order, as before.
constexpr uint32_t CP_SEG = 16384;
constexpr int CP_MAX_BITS = 6;

__device__ __forceinline__ uint32_t cpQueryOfSeg(uint32_t seg, uint32_t nQ, const uint32_t *__restrict__ segBase) {
    uint32_t lo = 0, hi = nQ;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segBase[mid] <= seg) lo = mid;
        else hi = mid;
    }
    return lo;
}

// target ranges per query for a sub-batch whose average query has avgQ hits (0: no split); see the comments where it is applied
inline int coarseBitsFor(uint64_t avgQ, int tBits) {
    const uint64_t perVQ = getenv("SD_PF_COARSE") ? (uint64_t) atoll(getenv("SD_PF_COARSE")) : 200000;
    const uint64_t perVQsplit = getenv("SD_PF_COARSE") ? perVQ : (getenv("SD_PF_COARSE_SPLIT") ? (uint64_t) atoll(getenv("SD_PF_COARSE_SPLIT")) : 50000);
    int cBits = 0;
    if (avgQ > perVQ)
        while (cBits < CP_MAX_BITS && tBits - (cBits + 1) >= 8 && (avgQ >> cBits) > perVQsplit) cBits++;
    return cBits;
}

__global__ void __launch_bounds__(256)
coarse_count_kernel(uint32_t nQ, const uint64_t *__restrict__ qHitBase, const uint32_t *__restrict__ segBase, int tBits, int cBits,
                    const uint32_t *__restrict__ inKey, const uint2 *__restrict__ inKV, uint32_t *__restrict__ segCount /* [segment][C] */,
                    const uint8_t *__restrict__ inR6 /* nullable: the hits' top six target bits, one byte per hit (written by the join's
                                                        scatter): 1 B per hit read here instead of 8 */) {
    __shared__ uint32_t hist[1 << CP_MAX_BITS];
    const uint32_t seg = blockIdx.x;
    const uint32_t q = cpQueryOfSeg(seg, nQ, segBase);
    const uint64_t s = qHitBase[q] + (uint64_t) (seg - segBase[q]) * CP_SEG;
    const uint64_t e = min(qHitBase[q + 1], s + CP_SEG);
    const int C = 1 << cBits;
    if (threadIdx.x < C) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tMask = (1u << tBits) - 1;
    const int shift = tBits - cBits;
    if (inR6) {   // four hits per 32-bit load (the array is 4-byte aligned and padded; a segment starts anywhere)
        const int down = CP_MAX_BITS - cBits;
        for (uint64_t w = (s & ~3ull) + 4ull * threadIdx.x; w < e; w += 1024) {
            const uint32_t v = *(const uint32_t *) (inR6 + w);
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (w + b >= s && w + b < e) atomicAdd(&hist[((v >> (8 * b)) & 0xFFu) >> down], 1u);
        }
        __syncthreads();
        if (threadIdx.x < C) segCount[(size_t) seg * C + threadIdx.x] = hist[threadIdx.x];
        return;
    }
    // (one load per thread and iteration, one LDS atomic per lane: a version with 8 loads in flight and one atomic per group of
    // lanes with the same range was 1.7 ms per 8 192 queries slower on the same box)
    for (uint64_t i = s + threadIdx.x; i < e; i += 256) atomicAdd(&hist[((inKV ? inKV[i].x : inKey[i]) & tMask) >> shift], 1u);
    __syncthreads();
    if (threadIdx.x < C) segCount[(size_t) seg * C + threadIdx.x] = hist[threadIdx.x];
}

// per query: counts [segment][range] -> start of that segment's share inside the query's hit segment, range-major;
// vqHitBase[q*C + c] = start of range c (absolute)
__global__ void __launch_bounds__(256)
coarse_offsets_kernel(uint32_t nQ, const uint64_t *__restrict__ qHitBase, const uint32_t *__restrict__ segBase, int cBits,
                      uint32_t *__restrict__ segCount, uint64_t *__restrict__ vqHitBase) {
    __shared__ uint32_t part[5];
    __shared__ uint32_t carry;
    const uint32_t q = blockIdx.x;
    const int C = 1 << cBits;
    const uint32_t s0 = segBase[q], nSeg = segBase[q + 1] - s0;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int c = 0; c < C; c++) {
        if (t == 0) vqHitBase[(size_t) q * C + c] = qHitBase[q] + carry;
        for (uint32_t x0 = 0; x0 < nSeg; x0 += 256) {
            const uint32_t x = x0 + t;
            const uint32_t v = x < nSeg ? segCount[(size_t) (s0 + x) * C + c] : 0;
            uint32_t incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, 64);
                if ((t & 63) >= off) incl += o;
            }
            if ((t & 63) == 63) part[t >> 6] = incl;
            __syncthreads();
            uint32_t base = carry;
            for (int w = 0; w < (t >> 6); w++) base += part[w];
            if (x < nSeg) segCount[(size_t) (s0 + x) * C + c] = base + incl - v;
            __syncthreads();
            if (t == 255) carry = base + incl;
            __syncthreads();
        }
    }
    if (q == nQ - 1 && t == 0) vqHitBase[(size_t) nQ * C] = qHitBase[nQ];
}

// CS_PER hits per thread and step, all loads issued before the first cursor is taken: with one 8-byte load per thread in flight the
// kernel moved 2.4 TB/s (32 wavefronts x 512 B per CU: a third of what the memory's latency asks for).  The order inside a range is not
// kept anyway (see above).  KV / DIAG: the input layout and the wide-position form are compiled apart (one 8-byte load per hit, no
// branches in the loop).
// Round 6: the step's 4 096 hits are put in range order in LDS before they leave.  Written straight from the registers a wavefront's
// store went to up to 64 different lines, and the address path takes them one line per clock: the PMC counters had the kernel
// issue-stalled 65 % of its wave cycles (profiles/r07i_pmc_stalls.txt: SQ_WAIT_INST_ANY 1.65e11 of 2.52e11, not LDS).  Per step:
// the lanes of a wavefront whose hits fall into one range find each other with cBits ballots (as before) and the first of them takes
// the group's room in the step's per-range counters; a scan of the counters gives every range its place in the staging buffer and
// its place in the segment's share of the output (the running cursors); the hits are written to the buffer, and then out slot by
// slot -- 64 consecutive slots are 64 consecutive hits of one range (or of two): a store of one or two runs of lines.
template <int CS_PER, bool KV, bool DIAG>
__device__ __forceinline__ void coarseScatterBody(uint64_t s, uint64_t e, uint64_t qs, uint32_t tMask, int shift, int cBits, uint32_t *cursor,
                                                  const uint32_t *__restrict__ inKey, const uint32_t *__restrict__ inVal,
                                                  const uint2 *__restrict__ inKV, uint2 *__restrict__ outKV, const uint16_t *__restrict__ hitDiag,
                                                  uint2 *stage /* [CS_STEP] */, uint8_t *stageRange /* [CS_STEP] */, uint32_t *cnt /* [C] */,
                                                  uint32_t *rstart /* [C + 1] */, uint32_t *gbase /* [C] */) {
    constexpr int CS_STEP = 256 * CS_PER;
    const uint32_t lowMask = (1u << shift) - 1;
    const int C = 1 << cBits;
    const int t = threadIdx.x, lane = t & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint64_t i0 = s; i0 < e; i0 += CS_STEP) {
        if (t < C) cnt[t] = 0;
        uint32_t k[CS_PER], v[CS_PER], d[CS_PER];
        bool live[CS_PER];
#pragma unroll
        for (int u = 0; u < CS_PER; u++) {
            const uint64_t x = i0 + (uint64_t) u * 256 + (uint64_t) t;
            live[u] = x < e;
            k[u] = 0;
            v[u] = 0;
            d[u] = 0;
            if (live[u]) {
                if (KV) {
                    const uint2 kv = inKV[x];
                    k[u] = kv.x;
                    v[u] = kv.y;
                } else {
                    k[u] = inKey[x];
                    v[u] = inVal[x];
                }
                if (DIAG) d[u] = (uint32_t) hitDiag[x] & 0xFFu;
            }
        }
        __syncthreads();   // the counters are cleared (and the previous step's staging buffer has been written out)
        uint32_t p[CS_PER], r[CS_PER];
#pragma unroll
        for (int u = 0; u < CS_PER; u++) {
            r[u] = (k[u] & tMask) >> shift;
            unsigned long long same = __ballot(live[u]);
            for (int bb = 0; bb < cBits; bb++) {
                const bool bit = (r[u] >> bb) & 1u;
                const unsigned long long with = __ballot(bit);
                same &= bit ? with : ~with;
            }
            // (a dead lane's `same` is not a group it belongs to: it is its own leader and takes nothing)
            const int leader = live[u] ? __ffsll((long long) same) - 1 : lane;
            uint32_t base = 0;
            if (live[u] && lane == leader) base = atomicAdd(&cnt[r[u]], (uint32_t) __popcll(same));
            p[u] = (uint32_t) __shfl((int) base, leader, 64) + (uint32_t) __popcll(same & below);
        }
        __syncthreads();
        if (t < 64) {   // one wavefront: where every range starts in the staging buffer, and in the output
            uint32_t c = t < C ? cnt[t] : 0u, incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            if (t < C) {
                rstart[t] = incl - c;
                gbase[t] = cursor[t];
                cursor[t] += c;
            }
            if (t == C - 1) rstart[C] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CS_PER; u++)
            if (live[u]) {
                const uint32_t at = rstart[r[u]] + p[u];
                stage[at] = make_uint2(DIAG ? (d[u] << shift) | (k[u] & lowMask) : k[u], v[u]);
                stageRange[at] = (uint8_t) r[u];
            }
        __syncthreads();
        const uint32_t n = rstart[C];
#pragma unroll
        for (int u = 0; u < CS_PER; u++) {
            const uint32_t x = (uint32_t) u * 256 + (uint32_t) t;
            if (x < n) {
                const uint32_t rr = stageRange[x];
                outKV[qs + gbase[rr] + (x - rstart[rr])] = stage[x];
            }
        }
    }
}

template <int CS_PER>
__global__ void __launch_bounds__(256)
coarse_scatter_kernel(uint32_t nQ, const uint64_t *__restrict__ qHitBase, const uint32_t *__restrict__ segBase, int tBits, int cBits,
                      const uint32_t *__restrict__ segOffset, const uint32_t *__restrict__ inKey,
                      const uint32_t *__restrict__ inVal, const uint2 *__restrict__ inKV,
                      uint2 *__restrict__ outKV /* (key, value) pairs: one 8-byte store per hit */,
                      const uint16_t *__restrict__ hitDiag /* wide stream positions: the diagonal byte goes into the key bits
                                                              above the virtual query's target bits (the range is implied
                                                              by the segment), nullptr otherwise */) {
    __shared__ uint32_t cursor[1 << CP_MAX_BITS], cnt[1 << CP_MAX_BITS], rstart[(1 << CP_MAX_BITS) + 1], gbase[1 << CP_MAX_BITS];
    constexpr int CS_STEP = 256 * CS_PER;
    __shared__ uint2 stage[CS_STEP];
    __shared__ uint8_t stageRange[CS_STEP];
    const uint32_t seg = blockIdx.x;
    const uint32_t q = cpQueryOfSeg(seg, nQ, segBase);
    const uint64_t qs = qHitBase[q];
    const uint64_t s = qs + (uint64_t) (seg - segBase[q]) * CP_SEG;
    const uint64_t e = min(qHitBase[q + 1], s + CP_SEG);
    const int C = 1 << cBits;
    if (threadIdx.x < C) cursor[threadIdx.x] = segOffset[(size_t) seg * C + threadIdx.x];
    __syncthreads();
    const uint32_t tMask = (1u << tBits) - 1;
    const int shift = tBits - cBits;
    if (inKV) {
        if (hitDiag) coarseScatterBody<CS_PER, true, true>(s, e, qs, tMask, shift, cBits, cursor, inKey, inVal, inKV, outKV, hitDiag, stage, stageRange, cnt, rstart, gbase);
        else coarseScatterBody<CS_PER, true, false>(s, e, qs, tMask, shift, cBits, cursor, inKey, inVal, inKV, outKV, hitDiag, stage, stageRange, cnt, rstart, gbase);
    } else {
        if (hitDiag) coarseScatterBody<CS_PER, false, true>(s, e, qs, tMask, shift, cBits, cursor, inKey, inVal, inKV, outKV, hitDiag, stage, stageRange, cnt, rstart, gbase);
        else coarseScatterBody<CS_PER, false, false>(s, e, qs, tMask, shift, cBits, cursor, inKey, inVal, inKV, outKV, hitDiag, stage, stageRange, cnt, rstart, gbase);
    }
}

// ---- hot-target filter (in front of partition_hits / bucket_match).
// The double-diagonal match (CacheFriendlyOperations.cpp:185-272) emits a hit only when the low byte of its diagonal equals the
// previous hit's of the same target in the query's stream -- 0 for the first hit of a target.  So a target can contribute a
// candidate only if two of its hits share the diagonal byte, or one of its hits has diagonal byte 0; call such a target hot.
// Hits of every other target are provably dead weight: they emit nothing, and the match of one target never looks at
// another's hits (also across the parts of an overflowed hit buffer, which only restrict what "previous" means).  On a
// proteome-scale target set a query's k-mers hit most targets at most a handful of times on unrelated diagonals: ~95 % of
// the hit stream belongs to targets that are not hot, and was carried through partition_hits (read twice, written once)
// and bucket_match (read, counted, sorted) only to be dropped there.
// One workgroup per (virtual) query segment, two streaming passes, no sort:
//   pass A  every hit with a non-zero diagonal byte enters a blocked Bloom filter keyed by (target, diagonal byte): both
//           bits of the key live in ONE 32-bit LDS word and are set by one returning atomic OR, so of two hits with the
//           same key the later one always sees both bits (no false negatives, whatever the interleaving); a key found
//           present -- and every hit with diagonal byte 0 -- marks its target in the `hot` bitmap (indexed by the low
//           HF_HOT_LOG2 target bits: aliasing targets only makes the filter keep more);
//   pass B  hits of hot targets are compacted to the front of the segment IN PLACE.  Tile by tile: all wavefronts hold
//           tile i in registers (and have the loads of tile i + 1 in flight) before any survivor of tile i is written, and
//           the survivors of tiles 0 .. i fit in front of tile i + 1, so no unread hit is overwritten.
// False positives (Bloom collisions, bitmap aliasing) cost bandwidth downstream, never correctness: bucket_match decides.
// (HF_PER, template parameter: hits per thread and tile of pass B; the next tile is in flight as well)
constexpr int HF_PER_A = 8;                // hits per thread and step of pass A

__device__ __forceinline__ uint32_t hfMix(uint32_t tgt, uint32_t d8) {
    uint32_t x = tgt * 0x9E3779B1u ^ (d8 * 0x85EBCA6Bu + 0x27D4EB2Fu);
    x ^= x >> 15;
    x *= 0x2C1B3C6Du;
    x ^= x >> 13;
    return x;
}

// NT threads, a Bloom filter of BW 32-bit words, a hot bitmap over the low HL target bits.  The filter is sized for
// BW * 64 / 5 keys (a proteome-scale query segment in one round); a longer segment is taken in rounds over classes of targets (a hash of the target picks the round), the
// Bloom filter cleared in between and the hot bitmap kept: the false-positive rate stays at the design point whatever the
// segment length, at one more read of the segment per round.
template <int NT, int BW, int HL, int HF_PER = 4>
__global__ void __launch_bounds__(NT)
hot_filter_kernel(uint32_t nVQ, const uint64_t *__restrict__ segBase, int tBits, uint32_t *key, uint32_t *val, uint2 *kv,
                  int wpBits /* wide stream positions: the diagonal byte sits in the key above the target bits */,
                  uint32_t minSeg /* shorter segments pass unfiltered */, uint32_t *__restrict__ segCount) {
    constexpr int TILE = NT * HF_PER;
    __shared__ uint32_t bloom[BW];
    __shared__ uint32_t hot[1 << (HL - 5)];
    __shared__ uint32_t outCur;
    const int t = threadIdx.x, lane = t & 63;
    // a workgroup takes the segments blockIdx.x, blockIdx.x + gridDim.x, ...: with one workgroup per segment that is one round; with
    // a grid of one workgroup per CU (SD_PF_HF_PERSIST) a workgroup keeps the CU it has waited for
    for (uint32_t q = blockIdx.x; q < nVQ; q += gridDim.x) {
    __syncthreads();   // (the previous segment's last reads of hot / outCur)
    const uint64_t s = segBase[q], e = segBase[q + 1];
    const uint64_t n = e - s;
    if (n < (uint64_t) minSeg || n >= 0xFFFFFFF0ull) {
        if (t == 0) segCount[q] = (uint32_t) n;
        continue;
    }
    for (int x = t; x < (1 << (HL - 5)); x += NT) hot[x] = 0;
    if (t == 0) outCur = 0;
    const uint32_t tMask = (1u << tBits) - 1;
    const uint32_t hotMask = (1u << HL) - 1;
    constexpr uint32_t KEYS_PER_ROUND = (uint32_t) BW * 64u / 5u;   // 2.5 bits per key
    const uint32_t rounds = (uint32_t) ((n + KEYS_PER_ROUND - 1) / KEYS_PER_ROUND);
    // ---- pass A (HF_PER_A independent loads per thread in flight)
    for (uint32_t r = 0; r < rounds; r++) {
        __syncthreads();
        for (int x = t; x < BW; x += NT) bloom[x] = 0;
        __syncthreads();
        for (uint64_t base = s; base < e; base += (uint64_t) NT * HF_PER_A) {
            uint32_t k[HF_PER_A], v[HF_PER_A];
#pragma unroll
            for (int j = 0; j < HF_PER_A; j++) {
                const uint64_t i = base + (uint64_t) j * NT + t;
                k[j] = 0xFFFFFFFFu;
                v[j] = 0;
                if (i < e) {
                    if (kv) {
                        const uint2 h = kv[i];
                        k[j] = h.x;
                        v[j] = h.y;
                    } else {
                        k[j] = key[i];
                        v[j] = val[i];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < HF_PER_A; j++) {
                const uint64_t i = base + (uint64_t) j * NT + t;
                if (i < e) {
                    const uint32_t tgt = k[j] & tMask;
                    const uint32_t d8 = wpBits ? (k[j] >> wpBits) & 0xFFu : v[j] >> 24;
                    bool isHot = false;
                    if (d8 == 0) {
                        isHot = r == 0;
                    } else if (rounds == 1 || (uint32_t) (((uint64_t) (tgt * 0x9E3779B1u) * rounds) >> 32) == r) {
                        const uint32_t x = hfMix(tgt, d8);
                        const uint32_t w = (uint32_t) (((uint64_t) x * (uint32_t) BW) >> 32);
                        const uint32_t m = (1u << (x & 31u)) | (1u << ((x >> 5) & 31u));
                        const uint32_t old = atomicOr(&bloom[w], m);
                        isHot = (old & m) == m;
                    }
                    if (isHot) atomicOr(&hot[(tgt & hotMask) >> 5], 1u << (tgt & 31u));
                }
            }
        }
    }
    __syncthreads();
    // ---- pass B: in-place compaction
    uint32_t ck[HF_PER], cv[HF_PER];
    auto loadTile = [&](uint64_t base, uint32_t *k, uint32_t *v) {
#pragma unroll
        for (int j = 0; j < HF_PER; j++) {
            const uint64_t i = base + (uint64_t) j * NT + t;
            k[j] = 0xFFFFFFFFu;
            v[j] = 0;
            if (i < e) {
                if (kv) {
                    const uint2 h = kv[i];
                    k[j] = h.x;
                    v[j] = h.y;
                } else {
                    k[j] = key[i];
                    v[j] = val[i];
                }
            }
        }
    };
    loadTile(s, ck, cv);
    for (uint64_t base = s; base < e; base += TILE) {
        uint32_t nk[HF_PER], nv[HF_PER];
        if (base + TILE < e) loadTile(base + TILE, nk, nv);
        // which of this tile's hits stay (uses the tile: its loads have completed before the barrier below is reached)
        uint32_t keepBits = 0, mine = 0;
        unsigned long long bal[HF_PER];
#pragma unroll
        for (int j = 0; j < HF_PER; j++) {
            const uint64_t i = base + (uint64_t) j * NT + t;
            bool keep = false;
            if (i < e) {
                const uint32_t tgt = ck[j] & tMask;
                keep = (hot[(tgt & hotMask) >> 5] >> (tgt & 31u)) & 1u;
            }
            bal[j] = __ballot(keep);
            if (keep) keepBits |= 1u << j;
            mine += (uint32_t) __popcll(bal[j]);
        }
        __syncthreads();   // every wavefront holds tile `base` in registers: its survivors may now overwrite the segment's front
        uint32_t wbase = 0;
        if (lane == 0 && mine) wbase = atomicAdd(&outCur, mine);   // mine = the wavefront's total (identical in every lane)
        wbase = __shfl(wbase, 0, 64);
#pragma unroll
        for (int j = 0; j < HF_PER; j++) {
            if (keepBits & (1u << j)) {
                const uint64_t p = s + wbase + (uint32_t) __popcll(bal[j] & ((1ull << lane) - 1ull));
                if (kv) {
                    kv[p] = make_uint2(ck[j], cv[j]);
                } else {
                    key[p] = ck[j];
                    val[p] = cv[j];
                }
            }
            wbase += (uint32_t) __popcll(bal[j]);
        }
#pragma unroll
        for (int j = 0; j < HF_PER; j++) {
            ck[j] = nk[j];
            cv[j] = nv[j];
        }
    }
    __syncthreads();
    if (t == 0) segCount[q] = outCur;
    }
}

// workgroup w -> (query, bin): bins of a query are contiguous, binBase[q] = sum of bins of the queries before
__device__ __forceinline__ bool pfSlotOf(uint32_t w, uint32_t nQ, const uint64_t *__restrict__ binBase, uint32_t &q, uint32_t &b) {
    if (w >= binBase[nQ]) return false;
    uint32_t lo = 0, hi = nQ;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (binBase[mid] <= w) lo = mid;
        else hi = mid;
    }
    q = lo;
    b = w - (uint32_t) binBase[lo];
    return true;
}

// NT threads sort buckets of up to CAP hits.  slotList == nullptr: workgroup index -> (query, bin) through binBase,
// buckets above CAP are appended to bigList (or flag the fallback when there is no list to append to);
// slotList != nullptr: the listed slots only (second launch with a larger CAP for the few oversize buckets).
template <int NT, int CAP>
__global__ void __launch_bounds__(NT)
bucket_match_kernel(uint32_t nQ, const uint64_t *__restrict__ binBase, const uint32_t *__restrict__ qLog2Bins, int tBits,
                    const uint64_t *__restrict__ bktStart, const uint32_t *__restrict__ bktCount,
                    const uint2 *__restrict__ inKV, uint32_t *__restrict__ outKey,
                    uint32_t *__restrict__ outVal, uint32_t *__restrict__ bktEmit, int *__restrict__ flag,
                    const uint32_t *__restrict__ slotList, uint32_t *__restrict__ bigList, uint32_t *__restrict__ bigCount,
                    uint32_t bigCap, const uint32_t *__restrict__ qSplit, const uint32_t *__restrict__ qParts,
                    const uint32_t *__restrict__ qSplits, int vqShift /* query = virtual query >> vqShift */,
                    int wpBits /* 0, or (wide stream positions) the virtual query's target bits: the diagonal byte sits in the
                                  key above them and the value is the position */,
                    const uint32_t *__restrict__ listCount /* with slotList: entries of the list (read on the device), null: the grid */,
                    uint32_t listFirst /* first list entry this launch takes */) {
    __shared__ uint32_t eK[CAP], eV[CAP];
    __shared__ uint32_t cnt[PF_CNT_MAX / 2];   // packed 16-bit: counts -> group starts -> group ends
    __shared__ uint32_t part[NT / 64 + 1];
    constexpr int PER = CAP / NT;
    static_assert(CAP % NT == 0 && PER <= 32 && CAP < 65536, "bucket geometry");
    uint32_t q, b;
    if (slotList) {
        if (listCount && listFirst + blockIdx.x >= *listCount) return;
        const uint32_t sl = slotList[listFirst + blockIdx.x];
        q = sl / PF_NB_MAX;
        b = sl % PF_NB_MAX;
    } else if (!pfSlotOf(blockIdx.x, nQ, binBase, q, b)) {
        return;
    }
    const size_t slot = (size_t) q * PF_NB_MAX + b;
    int n = (int) bktCount[slot];
    if (n == 0) return;
    const int t = threadIdx.x;
    if (n > CAP) {
        if (t == 0) {
            if (bigList) {
                const uint32_t at = atomicAdd(bigCount, 1u);
                if (at < bigCap) bigList[at] = (uint32_t) slot;
                else atomicExch(flag, 2);
            } else {
                atomicExch(flag, 2);
            }
        }
        return;
    }
    const uint64_t start = bktStart[slot];
    const int shift = tBits - (int) qLog2Bins[q];
    const SplitView sv = splitView(qSplit, qParts, qSplits, q >> vqShift);
    const bool manyParts = sv.all != nullptr;     // the buffer overflowed more than once: see keep_max_kernel
    const uint32_t offMask = (1u << shift) - 1;   // key & offMask = target offset inside the bucket's range
    const bool WP = wpBits != 0;
    const uint32_t posMask = WP ? 0xFFFFFFFFu : 0xFFFFFFu;
    const uint32_t tgtMask = WP ? (1u << wpBits) - 1 : 0xFFFFFFFFu;   // what identifies the target in a key
    auto d8of = [&](uint32_t key, uint32_t val) -> uint32_t { return WP ? (key >> wpBits) & 0xFFu : val >> 24; };
    const int nCnt = shift == 0 ? 2 : (1 << shift);
    for (int x = t; x < nCnt / 2; x += NT) cnt[x] = 0;
    __syncthreads();
    uint32_t k[PER], v[PER];
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const int j = x * NT + t;
        if (j < n) {
            const uint2 kv = inKV[start + j];
            k[x] = kv.x;
            v[x] = kv.y;
            pk16Add(cnt, k[x] & offMask);
        }
    }
    __syncthreads();
    // A target hit once in this bucket cannot be flagged unless the low byte of its diagonal is 0 (the match compares with
    // the previous hit of the same target, 0 for the first): such hits -- the majority on a proteome-scale target set,
    // where most k-mer hits are chance hits on unrelated targets -- leave here, before the sort.
    uint32_t keepMask = 0, mineKept = 0;
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const int j = x * NT + t;
        if (j < n) {
            const uint32_t o = k[x] & offMask;
            if (pk16Get(cnt, o) > 1 || d8of(k[x], v[x]) == 0) {
                keepMask |= 1u << x;
                mineKept++;
            }
        }
    }
    __syncthreads();   // every count read before the singles are taken out
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const int j = x * NT + t;
        if (j < n && !(keepMask & (1u << x))) {
            const uint32_t o = k[x] & offMask;
            atomicSub(&cnt[o >> 1], 1u << ((o & 1u) * 16u));   // the only hit of its counter
        }
    }
    {   // n := hits that stay
        uint32_t tot = mineKept;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
        __syncthreads();
        if ((t & 63) == 0) part[t >> 6] = tot;
        __syncthreads();
        tot = 0;
        for (int wv = 0; wv < NT / 64; wv++) tot += part[wv];
        __syncthreads();
        if (tot == 0) return;   // bktEmit[slot] stays 0 (cleared by the caller)
        n = (int) tot;
    }
    pk16Scan<NT>(cnt, nCnt, part);
#pragma unroll
    for (int x = 0; x < PER; x++) {
        if (keepMask & (1u << x)) {
            const uint32_t p = pk16Add(cnt, k[x] & offMask);   // cnt[o] ends as the end of group o
            eK[p] = k[x];
            eV[p] = v[x];
        }
    }
    __syncthreads();
    // emission order inside every target group: rank by stream position (low 24 bits of the value)
    uint32_t fin[PER];
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const int p = x * NT + t;
        if (p < n) {
            k[x] = eK[p];
            v[x] = eV[p];
            const uint32_t o = k[x] & offMask;
            const uint32_t gs = o ? pk16Get(cnt, o - 1) : 0, ge = pk16Get(cnt, o);
            uint32_t f = gs;
            if (ge - gs > 1) {
                const uint32_t me = v[x] & posMask;
                for (uint32_t y = gs; y < ge; y++) f += ((eV[y] & posMask) < me) ? 1u : 0u;
            }
            fin[x] = f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int x = 0; x < PER; x++) {
        const int p = x * NT + t;
        if (p < n) {
            eK[fin[x]] = k[x];
            eV[fin[x]] = v[x];
        }
    }
    __syncthreads();
    // match (QueryMatcher.cpp double-diagonal logic, as match_diag_kernel) on the sorted bucket
    const int per = (n + NT - 1) / NT;
    const int pb = t * per, pe = min(n, pb + per);
    uint32_t emitMask = 0, mine = 0;
    for (int p = pb; p < pe; p++) {
        const uint8_t d8 = (uint8_t) d8of(eK[p], eV[p]);
        auto flagAt = [&](int x) -> bool {
            const uint8_t dx = (uint8_t) d8of(eK[x], eV[x]);
            bool first = (x == 0) || (((eK[x - 1] ^ eK[x]) & tgtMask) != 0);
            if (!first) first = partOfPos(sv, eV[x - 1] & posMask) != partOfPos(sv, eV[x] & posMask);   // a new part of the hit buffer
            const uint8_t prev = first ? (uint8_t) 0 : (uint8_t) d8of(eK[x - 1], eV[x - 1]);
            return dx == prev;
        };
        bool em = false;
        if (flagAt(p)) {
            em = true;
            int x = p;
            while (x > 0 && ((eK[x - 1] ^ eK[p]) & tgtMask) == 0) {
                x--;
                // one overflow: the two result lists are merged, equal consecutive diagonals collapse across the split.  More:
                // every part keeps its own list here, the merges in between are keep_max_kernel's
                if (manyParts && partOfPos(sv, eV[x] & posMask) != partOfPos(sv, eV[p] & posMask)) break;
                if (flagAt(x)) {
                    em = ((uint8_t) d8of(eK[x], eV[x])) != d8;
                    break;
                }
            }
        }
        if (em) {
            emitMask |= 1u << (p - pb);
            mine++;
        }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    uint32_t w = incl - mine;
    for (int wv = 0; wv < (t >> 6); wv++) w += part[wv];
    for (int p = pb; p < pe; p++) {
        if (emitMask & (1u << (p - pb))) {
            // candidates leave in the standard form (query << tBits | target); q = query << vqShift | range already
            outKey[start + w] = WP ? (q << wpBits) | (eK[p] & tgtMask) : eK[p];
            outVal[start + w] = eV[p];
            w++;
        }
    }
    if (t == NT - 1) {
        uint32_t tot = 0;
        for (int wv = 0; wv < NT / 64; wv++) tot += part[wv];
        bktEmit[slot] = tot;
    }
}

// ---- the match on a whole filtered segment (after hot_filter_kernel a proteome-scale query keeps a few thousand hits).
// One workgroup per (virtual) query whose segment fits the LDS sorter: every hit becomes one 64-bit word
// target << 32 | stream position << 8 | diagonal byte (wide positions: target << 40 | position << 8 | byte), a bitonic sort of
// the words IS the (target, emission) order, and the double-diagonal match + compaction run on the sorted array exactly as in
// bucket_match_kernel.  The segment then looks like a query with one bucket (slot 0) that is already matched, so
// bucket_collect_kernel and everything behind it are unchanged; longer segments are left to partition_hits / bucket_match
// (segDone = 0).
template <int NT, int CAP>
__global__ void __launch_bounds__(NT)
segment_match_kernel(uint32_t nVQ, const uint64_t *__restrict__ segBase, const uint32_t *__restrict__ segCount,
                     const uint64_t *__restrict__ outBase, int tBits, const uint32_t *__restrict__ inKey,
                     const uint32_t *__restrict__ inVal, const uint2 *__restrict__ inKV, uint32_t *__restrict__ outKey,
                     uint32_t *__restrict__ outVal, uint32_t *__restrict__ qLog2Bins, uint64_t *__restrict__ bktStart,
                     uint32_t *__restrict__ bktCount, uint32_t *__restrict__ bktEmit, uint8_t *__restrict__ segDone,
                     const uint32_t *__restrict__ qSplit, const uint32_t *__restrict__ qParts, const uint32_t *__restrict__ qSplits,
                     int vqShift, int wpBits) {
    __shared__ unsigned long long sk[CAP];
    __shared__ uint32_t part[NT / 64 + 1];
    const int t = threadIdx.x;
    // a workgroup takes the segments blockIdx.x, blockIdx.x + gridDim.x, ...: the caller launches a few workgroups per CU that keep the
    // CU they have waited for (64 KB of LDS beside the score wavefronts of the other streams) instead of one per segment
    for (uint32_t q = blockIdx.x; q < nVQ; q += gridDim.x) {
    __syncthreads();   // (the previous segment's last reads of sk / part)
    const uint32_t n = segCount[q];
    if (n > (uint32_t) CAP) {
        if (t == 0) segDone[q] = 0;
        continue;
    }
    const size_t slot = (size_t) q * PF_NB_MAX;
    const uint64_t ob = outBase[q];
    if (t == 0) {
        segDone[q] = 1;
        qLog2Bins[q] = 0;
        bktCount[slot] = 0;
        bktStart[slot] = ob;
    }
    if (n == 0) continue;   // bktEmit[slot] stays 0 (cleared by the caller)
    const uint64_t s = segBase[q];
    const uint32_t tMask = (1u << tBits) - 1;
    const bool WP = wpBits != 0;
    const int tSh = WP ? 40 : 32;
    uint32_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (uint32_t x = t; x < np2; x += NT) {
        unsigned long long w = ~0ull;
        if (x < n) {
            uint32_t k, v;
            if (inKV) {
                const uint2 h = inKV[s + x];
                k = h.x;
                v = h.y;
            } else {
                k = inKey[s + x];
                v = inVal[s + x];
            }
            const uint32_t tgt = k & tMask;
            const uint32_t d8 = WP ? (k >> wpBits) & 0xFFu : v >> 24;
            const uint32_t pos = WP ? v : v & 0xFFFFFFu;
            w = ((unsigned long long) tgt << tSh) | ((unsigned long long) pos << 8) | d8;
        }
        sk[x] = w;
    }
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t p = t; p < np2 / 2; p += NT) {
                const uint32_t lo = (p / stride) * stride * 2 + (p % stride), hi = lo + stride;
                const bool asc = (lo & size) == 0;
                const unsigned long long a = sk[lo], b = sk[hi];
                if ((a > b) == asc) {
                    sk[lo] = b;
                    sk[hi] = a;
                }
            }
        }
    }
    __syncthreads();
    const SplitView sv = splitView(qSplit, qParts, qSplits, q >> vqShift);
    const bool manyParts = sv.all != nullptr;
    const uint32_t posMask = WP ? 0xFFFFFFFFu : 0xFFFFFFu;
    auto tgtOf = [&](uint32_t x) -> uint32_t { return (uint32_t) (sk[x] >> tSh); };
    auto posOf = [&](uint32_t x) -> uint32_t { return (uint32_t) (sk[x] >> 8) & posMask; };
    auto d8Of = [&](uint32_t x) -> uint8_t { return (uint8_t) (sk[x] & 0xFFu); };
    const uint32_t per = (n + NT - 1) / NT;
    const uint32_t pb = min(n, (uint32_t) t * per), pe = min(n, pb + per);
    // match (CacheFriendlyOperations.cpp:185-272; the same state machine as bucket_match_kernel)
    auto flagAt = [&](uint32_t x) -> bool {
        bool first = (x == 0) || tgtOf(x - 1) != tgtOf(x);
        if (!first) first = partOfPos(sv, posOf(x - 1)) != partOfPos(sv, posOf(x));   // a new part of the hit buffer
        const uint8_t prev = first ? (uint8_t) 0 : d8Of(x - 1);
        return d8Of(x) == prev;
    };
    auto emits = [&](uint32_t p) -> bool {
        if (!flagAt(p)) return false;
        bool em = true;
        uint32_t x = p;
        while (x > 0 && tgtOf(x - 1) == tgtOf(p)) {
            x--;
            if (manyParts && partOfPos(sv, posOf(x)) != partOfPos(sv, posOf(p))) break;
            if (flagAt(x)) {
                em = d8Of(x) != d8Of(p);
                break;
            }
        }
        return em;
    };
    static_assert(CAP / NT <= 32, "one mask bit per element of a thread's run");
    uint32_t mine = 0, emitMask = 0;
    for (uint32_t p = pb; p < pe; p++)
        if (emits(p)) {
            emitMask |= 1u << (p - pb);
            mine++;
        }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    uint32_t w = incl - mine;
    for (int wv = 0; wv < (t >> 6); wv++) w += part[wv];
    if (mine)
        for (uint32_t p = pb; p < pe; p++)
            if (emitMask & (1u << (p - pb))) {
                outKey[ob + w] = (q << tBits) | tgtOf(p);
                outVal[ob + w] = WP ? posOf(p) : ((uint32_t) d8Of(p) << 24) | posOf(p);
                w++;
            }
    if (t == NT - 1) {
        uint32_t tot = 0;
        for (int wv = 0; wv < NT / 64; wv++) tot += part[wv];
        bktEmit[slot] = tot;
    }
    }
}

// per-query bin counts -> running bin base (so that a flat workgroup index maps to (query, bin))
// the (query, bin) slot of every bucket workgroup, a thread per workgroup: bucket_match / bucket_collect read theirs instead of each
// searching binBase (eleven dependent reads at the head of a workgroup that lives ten microseconds)
__global__ void __launch_bounds__(256)
slot_list_kernel(uint32_t nSlotsTotal, uint32_t nQ, const uint64_t *__restrict__ binBase, uint32_t *__restrict__ slotList) {
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nSlotsTotal) return;
    uint32_t q = 0, b = 0;
    slotList[w] = pfSlotOf(w, nQ, binBase, q, b) ? q * (uint32_t) PF_NB_MAX + b : 0u;
}

__global__ void bin_count_kernel(uint32_t nQ, const uint32_t *__restrict__ qLog2Bins, uint32_t *__restrict__ qBins) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nQ) qBins[q] = qLog2Bins[q] <= (uint32_t) PF_LB_MAX ? (1u << qLog2Bins[q]) : 0u;
    if (q == nQ) qBins[q] = 0;
}

// emitted hits of every bucket -> the dense candidate arrays (offsets: exclusive scan of the emit counts in
// (query, bin) slot order)
__global__ void __launch_bounds__(64)
bucket_collect_kernel(uint32_t nQ, const uint64_t *__restrict__ binBase, const uint64_t *__restrict__ bktStart,
                      const uint32_t *__restrict__ bktEmit, const uint64_t *__restrict__ emitOff,
                      const uint32_t *__restrict__ inKey, const uint32_t *__restrict__ inVal, uint32_t *__restrict__ cKey,
                      uint32_t *__restrict__ cVal, const uint32_t *__restrict__ slotList /* nullable: slot_list_kernel */) {
    uint32_t q, b;
    size_t slot;
    if (slotList) {
        slot = slotList[blockIdx.x];
    } else {
        if (!pfSlotOf(blockIdx.x, nQ, binBase, q, b)) return;
        slot = (size_t) q * PF_NB_MAX + b;
    }
    const uint32_t n = bktEmit[slot];
    if (n == 0) return;
    const uint64_t src = bktStart[slot], dst = emitOff[slot];
    for (uint32_t x = threadIdx.x; x < n; x += 64) {
        cKey[dst + x] = inKey[src + x];
        cVal[dst + x] = inVal[src + x];
    }
}

// K5: ungapped diagonal score (UngappedAlignment.cpp:30-43,416-430); candidates are (key,val) pairs
__global__ void __launch_bounds__(256)
score_diag_kernel(uint32_t nCand, const uint32_t *__restrict__ cKey, const uint32_t *__restrict__ cVal,
                  const DiagSrc ds, int tBits,
                  const uint8_t *__restrict__ qRes, const uint64_t *__restrict__ qOff, const int8_t *__restrict__ diagBias,
                  const uint8_t *__restrict__ tMasked, const uint64_t *__restrict__ tOff, const int8_t *__restrict__ mat,
                  int32_t *__restrict__ cScore, uint32_t *__restrict__ cLen,
                  const int8_t *__restrict__ qProf /* profile queries: [position][21] replaces matrix row + bias */,
                  uint32_t posMask /* stream position bits of the value word: 2^24 - 1, all 32 with wide positions */,
                  int diagFromTarget /* join path: the diagonal from the target's residues (diagFromResidues) instead of the index */) {
    __shared__ int8_t smat[441];
    for (int x = threadIdx.x; x < 441; x += blockDim.x) smat[x] = mat[x];
    __syncthreads();
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCand) return;
    const uint32_t k = cKey[c];
    const uint32_t q = k >> tBits, sid = k & ((1u << tBits) - 1);
    const int qL = (int) (qOff[q + 1] - qOff[q]);
    const int tL = (int) (tOff[sid + 1] - tOff[sid]);
    const uint32_t cv = cVal[c];
    const uint16_t d16 = diagFromTarget ? diagFromResidues(ds, q, cv & posMask, sid, cv >> 24, tMasked + tOff[sid], tL) : diagOf(ds, q, cv & posMask, sid);
    const int d = (int) (int16_t) d16;
    const uint8_t *qs = qRes + qOff[q];
    const int8_t *qb = diagBias + qOff[q];
    const uint8_t *ts = tMasked + tOff[sid];
    const int8_t *qpBase = qProf ? qProf + qOff[q] * 21 : nullptr;
    // computeSingelSequenceScores (UngappedAlignment.cpp:416-430) on one real diagonal
    auto scoreOn = [&](int diagonal, unsigned dist, int &nOut) {
        int n = 0, q0 = 0, t0 = 0;
        if (diagonal >= 0 && dist < (unsigned) qL) {
            n = min(tL, qL - (int) dist);
            q0 = (int) dist;
        } else if (diagonal < 0 && dist < (unsigned) tL) {
            n = min(tL - (int) dist, qL);
            t0 = (int) dist;
        }
        int score = 0, best = 0;
        if (qpBase) {   // profile query: the row of the position replaces matrix row + bias
            const int8_t *qp = qpBase + (size_t) q0 * 21;
            for (int x = 0; x < n; x++) {
                score += (int) qp[(size_t) x * 21 + ts[t0 + x]];
                score = score < 0 ? 0 : score;
                best = score > best ? score : best;
            }
        } else {
            // eight cells per step: the three byte streams (query letters, their bias, target letters) as one unaligned 8-byte read
            // each -- a lane walks its own pair of sequences, so every read is a request of its own and their number is what the
            // kernel costs
            int x = 0;
            for (; x + 8 <= n; x += 8) {
                unsigned long long wq, wb, wt;
                __builtin_memcpy(&wq, qs + q0 + x, 8);
                __builtin_memcpy(&wb, qb + q0 + x, 8);
                __builtin_memcpy(&wt, ts + t0 + x, 8);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int qr = (int) ((wq >> (8 * j)) & 0xFFull), tr = (int) ((wt >> (8 * j)) & 0xFFull);
                    const int8_t bj = (int8_t) (uint8_t) ((wb >> (8 * j)) & 0xFFull);
                    score += (int) (int8_t) (smat[qr * 21 + tr] + bj);
                    score = score < 0 ? 0 : score;
                    best = score > best ? score : best;
                }
            }
            for (; x < n; x++) {
                const int qr = qs[q0 + x];
                score += (int) (int8_t) (smat[qr * 21 + ts[t0 + x]] + qb[q0 + x]);
                score = score < 0 ? 0 : score;
                best = score > best ? score : best;
            }
        }
        nOut = n;
        return best;
    };
    int n = 0, best = 0;
    if (qL >= 32768 || tL >= 32768) {
        // computeLongScore (UngappedAlignment.cpp:312-329): the 16-bit diagonal of a sequence this long is ambiguous, so
        // every real diagonal it can stand for is scored (d16 - 65536 * {1 .. 1 + tL / 32768}, d16 + 65536 * {0 .. qL /
        // 65536}) and the best one kept (cLen, a work statistic, counts all of them)
        for (int dv = 1; dv <= 1 + tL / 32768; dv++) {
            const int real = (int) d16 - dv * 65536;
            int nn;
            const int sc = scoreOn(real, (unsigned) abs(real), nn);
            best = sc > best ? sc : best;
            n += nn;
        }
        for (int dv = 0; dv <= qL / 65536; dv++) {
            const int real = (int) d16 + dv * 65536;
            int nn;
            const int sc = scoreOn(real, (unsigned) abs(real), nn);
            best = sc > best ? sc : best;
            n += nn;
        }
    } else {
        const uint16_t minDist = (uint16_t) min((int) (uint16_t) (0 - d16), (int) (uint16_t) d16);
        best = scoreOn(d, minDist, n);
    }
    cScore[c] = best;
    cLen[c] = (uint32_t) n;
}

// K5, cooperative form (round 6; sequence queries).  The one-thread-per-candidate kernel above walks a diagonal eight cells per step
// with three 8-byte reads that belong to this lane alone: ~38 dependent steps per candidate, every read a request of its own (PMC:
// 17.9 MB fetched per query for 4.4 MB of diagonals).  Here a candidate is scored by EIGHT lanes: per step the group reads 64
// consecutive cells of the three byte streams (one 64-byte run each), every lane folds its eight cells into the summary of a clamped
// running sum -- r' = max(0, r + s) is max-plus linear in r, so a block of cells is (T, E, P, B): total, the running score it leaves
// when entered with 0, its maximal prefix sum, the best score inside when entered with 0; entering with r gives r_out = max(r + T, E),
// best = max(r + P, B) -- and the eight summaries are composed in order by three shuffle steps (exact integer arithmetic: the same
// score and length as the serial walk, bit for bit).  A candidate costs ceil(n / 64) steps instead of n / 8.  The diagonal itself
// (binary searches over the k-mer stream and the index list: dependent reads) comes from a pass of its own with a thread per
// candidate, diag_of_kernel, so that no lane of a group waits for it.  Sequences of 32 768 residues and more (computeLongScore)
// keep the serial walk, done by the group's first lane.
struct SdBlk {
    int T, E, P, B;
};
__device__ __forceinline__ SdBlk sdBlkThen(const SdBlk &a, const SdBlk &b) {   // a's cells, then b's
    SdBlk r;
    r.T = a.T + b.T;
    r.E = max(a.E + b.T, b.E);
    r.P = max(a.P, a.T + b.P);
    r.B = max(max(a.B, b.B), a.E + b.P);
    return r;
}

__global__ void __launch_bounds__(256)
diag_of_kernel(uint32_t nCand, const uint32_t *__restrict__ cKey, const uint32_t *__restrict__ cVal, const DiagSrc ds, int tBits, uint32_t posMask,
               uint16_t *__restrict__ cDiag, const uint8_t *__restrict__ tMasked, const uint64_t *__restrict__ tOff,
               int diagFromTarget /* the diagonal from the target's residues (diagFromResidues) instead of the index */) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCand) return;
    const uint32_t k = cKey[c], cv = cVal[c];
    const uint32_t q = k >> tBits, sid = k & ((1u << tBits) - 1);
    if (diagFromTarget) cDiag[c] = diagFromResidues(ds, q, cv & posMask, sid, cv >> 24, tMasked + tOff[sid], (int) (tOff[sid + 1] - tOff[sid]));
    else cDiag[c] = diagOf(ds, q, cv & posMask, sid);
}

__global__ void __launch_bounds__(256)
score_diag_coop_kernel(uint32_t nCand, const uint32_t *__restrict__ cKey, const uint16_t *__restrict__ cDiag, int tBits,
                       const uint8_t *__restrict__ qRes, const uint64_t *__restrict__ qOff, const int8_t *__restrict__ diagBias,
                       const uint8_t *__restrict__ tMasked, const uint64_t *__restrict__ tOff, const int8_t *__restrict__ mat,
                       int32_t *__restrict__ cScore, uint32_t *__restrict__ cLen) {
    __shared__ int8_t smat[441];
    for (int x = threadIdx.x; x < 441; x += blockDim.x) smat[x] = mat[x];
    __syncthreads();
    const uint32_t c = blockIdx.x * 32u + (threadIdx.x >> 3);
    const int sub = threadIdx.x & 7;
    if (c >= nCand) return;   // (whole groups leave: the shuffles below stay inside a group)
    const uint32_t k = cKey[c];
    const uint32_t q = k >> tBits, sid = k & ((1u << tBits) - 1);
    const uint16_t d16 = cDiag[c];
    const int d = (int) (int16_t) d16;
    const int qL = (int) (qOff[q + 1] - qOff[q]);
    const int tL = (int) (tOff[sid + 1] - tOff[sid]);
    const uint8_t *qs = qRes + qOff[q];
    const int8_t *qb = diagBias + qOff[q];
    const uint8_t *ts = tMasked + tOff[sid];
    if (qL >= 32768 || tL >= 32768) {
        // computeLongScore (UngappedAlignment.cpp:312-329), serial, by the group's first lane: every real diagonal the 16-bit one can stand for
        if (sub != 0) return;
        auto scoreOn = [&](int diagonal, unsigned dist, int &nOut) {
            int n = 0, q0 = 0, t0 = 0;
            if (diagonal >= 0 && dist < (unsigned) qL) {
                n = min(tL, qL - (int) dist);
                q0 = (int) dist;
            } else if (diagonal < 0 && dist < (unsigned) tL) {
                n = min(tL - (int) dist, qL);
                t0 = (int) dist;
            }
            int score = 0, best = 0;
            for (int x = 0; x < n; x++) {
                score += (int) (int8_t) (smat[(int) qs[q0 + x] * 21 + ts[t0 + x]] + qb[q0 + x]);
                score = score < 0 ? 0 : score;
                best = score > best ? score : best;
            }
            nOut = n;
            return best;
        };
        int n = 0, best = 0;
        for (int dv = 1; dv <= 1 + tL / 32768; dv++) {
            const int real = (int) d16 - dv * 65536;
            int nn;
            const int sc = scoreOn(real, (unsigned) abs(real), nn);
            best = sc > best ? sc : best;
            n += nn;
        }
        for (int dv = 0; dv <= qL / 65536; dv++) {
            const int real = (int) d16 + dv * 65536;
            int nn;
            const int sc = scoreOn(real, (unsigned) abs(real), nn);
            best = sc > best ? sc : best;
            n += nn;
        }
        cScore[c] = best;
        cLen[c] = (uint32_t) n;
        return;
    }
    // computeSingelSequenceScores (UngappedAlignment.cpp:416-430) on the one real diagonal
    const unsigned dist = (unsigned) (uint16_t) min((int) (uint16_t) (0 - d16), (int) (uint16_t) d16);
    int n = 0, q0 = 0, t0 = 0;
    if (d >= 0 && dist < (unsigned) qL) {
        n = min(tL, qL - (int) dist);
        q0 = (int) dist;
    } else if (d < 0 && dist < (unsigned) tL) {
        n = min(tL - (int) dist, qL);
        t0 = (int) dist;
    }
    int run = 0, best = 0;   // the walk's state behind the chunks done so far (identical in the group's lanes)
    for (int x0 = 0; x0 < n; x0 += 64) {
        const int x = x0 + 8 * sub;
        SdBlk b = {0, 0, 0, 0};
        if (x + 8 <= n) {
            unsigned long long wq, wb, wt;
            __builtin_memcpy(&wq, qs + q0 + x, 8);
            __builtin_memcpy(&wb, qb + q0 + x, 8);
            __builtin_memcpy(&wt, ts + t0 + x, 8);
            int sum = 0, r = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int qr = (int) ((wq >> (8 * j)) & 0xFFull), tr = (int) ((wt >> (8 * j)) & 0xFFull);
                const int8_t bj = (int8_t) (uint8_t) ((wb >> (8 * j)) & 0xFFull);
                const int sc = (int) (int8_t) (smat[qr * 21 + tr] + bj);
                sum += sc;
                r = max(0, r + sc);
                b.P = j == 0 ? sum : max(b.P, sum);
                b.B = max(b.B, r);
            }
            b.T = sum;
            b.E = r;
        } else if (x < n) {
            int sum = 0, r = 0;
            for (int j = 0; x + j < n; j++) {
                const int sc = (int) (int8_t) (smat[(int) qs[q0 + x + j] * 21 + ts[t0 + x + j]] + qb[q0 + x + j]);
                sum += sc;
                r = max(0, r + sc);
                b.P = j == 0 ? sum : max(b.P, sum);
                b.B = max(b.B, r);
            }
            b.T = sum;
            b.E = r;
        }
        // (a lane without cells holds the neutral block: T = E = B = 0, P = 0 -- entering with r >= 0 leaves r and a best of r, which the
        // walk has already counted)
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            SdBlk o;
            o.T = __shfl_down(b.T, off, 8);
            o.E = __shfl_down(b.E, off, 8);
            o.P = __shfl_down(b.P, off, 8);
            o.B = __shfl_down(b.B, off, 8);
            if ((sub & (2 * off - 1)) == 0) b = sdBlkThen(b, o);
        }
        b.T = __shfl(b.T, 0, 8);
        b.E = __shfl(b.E, 0, 8);
        b.P = __shfl(b.P, 0, 8);
        b.B = __shfl(b.B, 0, 8);
        best = max(best, max(run + b.P, b.B));
        run = max(run + b.T, b.E);
    }
    if (sub == 0) {
        cScore[c] = best;
        cLen[c] = (uint32_t) n;
    }
}

// K6: keep the first element holding the per-(query,target) maximum of min(255,score).
// Queries whose hit buffer overflowed more than once (m >= 2 splits, parts 0..m): the reference merges the carried result list
// with the new part's at every overflow from the second on -- back to front, so equal neighbouring diagonal bytes collapse onto
// the LATER element --, scores the merged list and keeps one element per target, the first with the maximal 8-bit score in that
// reversed order (QueryMatcher.cpp:289-303, CacheFriendlyOperations.cpp:118-150,350-380); the last part is merged front to back
// (:84-116) and the final keepMaxScoreElementOnly takes the first maximum.  Per target that is: over parts 0 and 1 the LAST
// candidate in stream order with the maximal score (the last candidate of part 0 leaves first if its diagonal byte equals the
// first of part 1's); every middle part takes over with its own last maximum unless the carried element scores strictly
// higher; in the last part the first candidate leaves if its diagonal byte equals the carried element's, and the carried
// element wins ties.  The head of a (query, target) group walks the group.
__global__ void __launch_bounds__(256)
keep_max_kernel(uint32_t nCand, const uint32_t *__restrict__ cKey, const uint32_t *__restrict__ cVal, const int32_t *__restrict__ cScore,
                uint8_t *__restrict__ keep, int tBits, uint32_t posMask, const uint32_t *__restrict__ qSplit,
                const uint32_t *__restrict__ qParts, const uint32_t *__restrict__ qSplits, const DiagSrc ds) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nCand) return;
    const uint32_t k = cKey[c];
    const uint32_t q = k >> tBits;
    if (qParts[q] >= 2) {
        if (c > 0 && cKey[c - 1] == k) return;   // the group's head decides for all
        const SplitView sv = splitView(qSplit, qParts, qSplits, q);
        const uint32_t m = sv.n;
        uint32_t e = c;
        while (e < nCand && cKey[e] == k) e++;
        auto partOf = [&](uint32_t i) { return partOfPos(sv, cVal[i] & posMask); };
        auto cnt = [&](uint32_t i) { return min(255, cScore[i]); };
        auto d8 = [&](uint32_t i) { return (uint32_t) diagOf(ds, q, cVal[i] & posMask, k & ((1u << tBits) - 1)) & 0xFFu; };
        auto endOfPart = [&](uint32_t i, uint32_t p) {
            while (i < e && partOf(i) <= p) i++;
            return i;
        };
        // parts 0 and 1
        const uint32_t b0 = endOfPart(c, 0), b1 = endOfPart(b0, 1);
        const bool dropLast0 = b0 > c && b1 > b0 && d8(b0 - 1) == d8(b0);
        uint32_t best = 0xFFFFFFFFu;
        int bestCnt = -1;
        for (uint32_t i = c; i < b1; i++) {
            if (dropLast0 && i == b0 - 1) continue;
            if (cnt(i) >= bestCnt) {
                best = i;
                bestCnt = cnt(i);
            }
        }
        // middle parts 2 .. m - 1
        uint32_t cur = b1;
        for (uint32_t p = 2; p + 1 <= m; p++) {
            const uint32_t b = endOfPart(cur, p);
            uint32_t lb = 0xFFFFFFFFu;
            int lc = -1;
            for (uint32_t i = cur; i < b; i++)
                if (cnt(i) >= lc) {
                    lb = i;
                    lc = cnt(i);
                }
            if (lb != 0xFFFFFFFFu && lc >= bestCnt) {
                best = lb;
                bestCnt = lc;
            }
            cur = b;
        }
        // the last part
        const bool dropFirst = best != 0xFFFFFFFFu && cur < e && d8(cur) == d8(best);
        uint32_t winner = best;
        int wc = bestCnt;
        for (uint32_t i = cur; i < e; i++) {
            if (dropFirst && i == cur) continue;
            if (cnt(i) > wc) {
                winner = i;
                wc = cnt(i);
            }
        }
        for (uint32_t i = c; i < e; i++) keep[i] = i == winner ? 1 : 0;
        return;
    }
    const int mine = min(255, cScore[c]);
    bool ok = true;
    // no earlier element with count >= mine, no later element with count > mine
    for (uint32_t x = c; x > 0 && cKey[x - 1] == k; x--)
        if (min(255, cScore[x - 1]) >= mine) { ok = false; break; }
    if (ok)
        for (uint32_t x = c + 1; x < nCand && cKey[x] == k; x++)
            if (min(255, cScore[x]) > mine) { ok = false; break; }
    keep[c] = ok ? 1 : 0;
}

// K7: one workgroup per query; SEL_CAP = LDS sorter capacity (48 KB at 4 096; 8 192 for result lists beyond 2 047 hits)
__device__ __forceinline__ void bitonicSort(unsigned long long *keys, uint32_t *pay, int n /* power of two */) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < n / 2; t += blockDim.x) {
                const int lo = (t / stride) * stride * 2 + (t % stride);
                const int hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == asc) {
                    keys[lo] = b; keys[hi] = a;
                    const uint32_t pa = pay[lo]; pay[lo] = pay[hi]; pay[hi] = pa;
                }
            }
        }
    }
    __syncthreads();
}

template <int SEL_CAP>
__global__ void __launch_bounds__(256)
select_hits_kernel(uint32_t nQ, const uint32_t *__restrict__ kStartOfQ /* nQ+1, into kept arrays */,
                   const uint32_t *__restrict__ kKey, const uint32_t *__restrict__ kVal,
                   const int32_t *__restrict__ kScore, const DiagSrc ds, int tBits,
                   uint32_t binMask, int maxHits, int minDiag, const uint32_t *__restrict__ identityId,
                   const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff, int covMode, float covThr,
                   const uint8_t *__restrict__ qRes, const int8_t *__restrict__ diagBias, const int8_t *__restrict__ mat,
                   sd_hit *__restrict__ outHits, uint32_t *__restrict__ outCount, int *__restrict__ errFlag,
                   const int8_t *__restrict__ qProf /* nullable: profile queries */,
                   uint32_t posMask /* stream position bits of the value word */,
                   int joinBinBits /* -1, or (join path) log2(BINSIZE): the value word orders by k-mer ordinal, hits of one
                                      k-mer's list by sequence id */,
                   const uint32_t *__restrict__ qSplit, const uint32_t *__restrict__ qParts, const uint32_t *__restrict__ qSplits,
                   uint32_t *__restrict__ bigList /* nullable: queries for select_hits_big_kernel */, uint32_t *__restrict__ bigCount) {
    __shared__ unsigned long long keys[SEL_CAP];
    __shared__ uint32_t pay[SEL_CAP];
    __shared__ unsigned int hist[256];
    __shared__ int sThr, sCnt, sMaxSelf, sQual;
    const uint32_t q = blockIdx.x;
    if (q >= nQ) return;
    const uint32_t beg = kStartOfQ[q], end = kStartOfQ[q + 1];
    const uint32_t ident = identityId[q];
    const uint32_t dbSizeCap = (uint32_t) maxHits;
    for (int x = threadIdx.x; x < 256; x += blockDim.x) hist[x] = 0;
    __syncthreads();
    for (uint32_t x = beg + threadIdx.x; x < end; x += blockDim.x) atomicAdd(&hist[min(255, kScore[x])], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        // computeScoreThreshold (QueryMatcher.h:206-216) then max(minDiagScoreThr, .) (QueryMatcher.cpp:154-155)
        size_t found = 0;
        int thr = 255;
        for (thr = 255; thr > 0; thr--) {
            found += hist[thr];
            if (found >= (size_t) dbSizeCap) break;
        }
        thr = max(minDiag, thr);
        sThr = thr;
        int qual = 0;
        for (int c = thr; c < 256; c++) qual += (int) hist[c];
        sQual = qual;
        sMaxSelf = 1;
        if (thr >= 255) {
            // Truncated scores (QueryMatcher.cpp:157-170): the cut itself is the saturated 8-bit score, so the reference
            // rescales the true diagonal scores of the saturated hits against the query's self score (rescoreHits,
            // :525-544), re-sorts them by the rescaled byte (stable) and reports 255 + byte * maxSelf / 255.
            // Self score: the query against itself on diagonal 0 (UngappedAlignment::scoreSingleSequence).
            const uint8_t *qs = qRes + qOff[q];
            const int8_t *qb = diagBias + qOff[q];
            const int qL = (int) (qOff[q + 1] - qOff[q]);
            int score = 0, best = 0;
            for (int x = 0; x < qL; x++) {
                score += qProf ? (int) qProf[(qOff[q] + (uint64_t) x) * 21 + qs[x]] : (int) (int8_t) (mat[qs[x] * 21 + qs[x]] + qb[x]);
                score = score < 0 ? 0 : score;
                best = score > best ? score : best;
            }
            int ms = best - 255;
            ms = ms < 1 ? 1 : ms;
            ms = ms > 65535 ? 65535 : ms;
            sMaxSelf = ms;
        }
    }
    __syncthreads();
    const int thr = sThr;
    const bool rescored = thr >= 255;
    auto rescaledByte = [&](uint32_t e) -> uint32_t {
        unsigned int ns = (unsigned int) kScore[e] - 255u;
        const float sc = (float) min(ns, 65535u);
        const double dv = (double) ((sc / (float) sMaxSelf) * 255.0f) + 0.5;
        return (uint32_t) (uint8_t) (int) dv;
    };
    // order of the cut: (rescaled) count desc, then the order the reference's counting sort keeps: bin asc, stream
    // position asc
    const SplitView sv = splitView(qSplit, qParts, qSplits, q);   // more than one overflow: overflowOrderPos
    auto keyOf = [&](uint32_t x) -> unsigned long long {
        const uint32_t sid = kKey[x] & ((1u << tBits) - 1);
        const uint32_t top = rescored ? 255u - rescaledByte(x) : 255u - (uint32_t) min(255, kScore[x]);
        bool backwards = false;
        const uint32_t pos = sv.all ? overflowOrderPos(sv, kVal[x] & posMask, backwards) : kVal[x] & posMask;
        if (joinBinBits >= 0) {
            const uint32_t inList = (sid >> joinBinBits) & 0xFFFFFu;
            return ((unsigned long long) top << 56) | ((unsigned long long) (sid & binMask) << 44) |
                   ((unsigned long long) pos << 20) | (unsigned long long) (backwards ? 0xFFFFFu - inList : inList);
        }
        return ((unsigned long long) top << 56) | ((unsigned long long) (sid & binMask) << 40) | (unsigned long long) pos;
    };
    // Only the first maxHits + 1 elements of that order are ever read (at most one of them is the identity target).
    // Usually everything at or above the cut fits the LDS arrays; otherwise (very many tied candidates) the candidate
    // list is swept in windows that cannot overflow them, keeping the best maxHits + 1 between windows.
    const int K = min(maxHits + 1, SEL_CAP / 2);
    if (sQual > SEL_CAP && maxHits + 1 > SEL_CAP / 2) {
        // a result list longer than the windowed selection can carry (maxHits > 4095) with more candidates at the cut than
        // the LDS sorter holds: the query goes to select_hits_big_kernel (radix selection + a sort through global scratch;
        // the reference has no cap on the list length, QueryMatcher.cpp:45-46,386), the others are unaffected (block-uniform)
        if (threadIdx.x == 0) {
            if (bigList) bigList[atomicAdd(bigCount, 1u)] = q;
            else {
                atomicExch(errFlag, 2);
                outCount[q] = 0xFFFFFFFFu;
            }
        }
        return;
    }
    int n = 0;
    if (sQual <= SEL_CAP) {
        if (threadIdx.x == 0) sCnt = 0;
        __syncthreads();
        for (uint32_t x = beg + threadIdx.x; x < end; x += blockDim.x) {
            if (min(255, kScore[x]) >= thr) {
                const int slot = atomicAdd(&sCnt, 1);
                keys[slot] = keyOf(x);
                pay[slot] = x;
            }
        }
        __syncthreads();
        n = sCnt;
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        for (int x = n + threadIdx.x; x < np2; x += blockDim.x) { keys[x] = ~0ull; pay[x] = 0xFFFFFFFFu; }
        __syncthreads();
        if (np2 > 1) bitonicSort(keys, pay, np2);
    } else {
        // very many candidates at or above the cut (large target sets, or a cut at the minimum score): stream through
        // the list, buffer what still beats the K-th best key seen so far, sort + truncate whenever the buffer could overflow
        if (threadIdx.x == 0) sCnt = 0;
        __syncthreads();
        unsigned long long bound = ~0ull;
        auto sortAndTruncate = [&]() {
            const int m = sCnt;
            int np2 = 1;
            while (np2 < m) np2 <<= 1;
            for (int x = m + threadIdx.x; x < np2; x += blockDim.x) { keys[x] = ~0ull; pay[x] = 0xFFFFFFFFu; }
            __syncthreads();
            if (np2 > 1) bitonicSort(keys, pay, np2);
            __syncthreads();
            const int have = min(m, K);
            bound = have >= K ? keys[K - 1] : ~0ull;
            __syncthreads();
            if (threadIdx.x == 0) sCnt = have;
            __syncthreads();
        };
        for (uint32_t base = beg; base < end; base += blockDim.x) {
            if (sCnt + (int) blockDim.x > SEL_CAP) sortAndTruncate();   // uniform: sCnt is read after a barrier
            const uint32_t x = base + threadIdx.x;
            if (x < end && min(255, kScore[x]) >= thr) {
                const unsigned long long kk = keyOf(x);
                if (kk < bound) {
                    const int slot = atomicAdd(&sCnt, 1);
                    keys[slot] = kk;
                    pay[slot] = x;
                }
            }
            __syncthreads();
        }
        sortAndTruncate();
        n = sCnt;
    }
    __syncthreads();
    // take the first (maxHits - hasIdentity) with id != identity (getResult, QueryMatcher.cpp:364-420); the identity target
    // occurs at most once in the list (one entry per target), so this is a shift by one behind its position
    __shared__ int sTake, sIdentPos;
    if (threadIdx.x == 0) sIdentPos = n;
    __syncthreads();
    for (int x = threadIdx.x; x < n; x += blockDim.x)
        if ((kKey[pay[x]] & ((1u << tBits) - 1)) == ident) sIdentPos = x;
    __syncthreads();
    {
        const int identPos = sIdentPos;
        const int current0 = (ident != 0xFFFFFFFFu) ? 1 : 0;
        const int avail = n - (identPos < n ? 1 : 0);
        const int take = max(0, min(avail, maxHits - current0));
        constexpr int PERT = (SEL_CAP / 2 + 255) / 256;
        uint32_t eReg[PERT];
        unsigned long long kReg[PERT];
#pragma unroll
        for (int y = 0; y < PERT; y++) {
            const int j = y * 256 + (int) threadIdx.x;
            if (j < take) {
                const uint32_t e = pay[j + (j >= identPos ? 1 : 0)];
                const uint32_t sid = kKey[e] & ((1u << tBits) - 1);
                // final order: |score| desc, seqId asc (QueryMatcher.h:38-48); true score for saturated counts
                const int sc = kScore[e];
                const int cnt = min(255, sc);
                const int prefScore = rescored ? (int) (255u + rescaledByte(e) * (unsigned int) sMaxSelf / 255u)
                                               : (cnt >= 255 ? sc : cnt);
                eReg[y] = e;
                kReg[y] = ((unsigned long long) (0x7FFFFFFFu - (uint32_t) prefScore) << 32) | sid;
            }
        }
        __syncthreads();
#pragma unroll
        for (int y = 0; y < PERT; y++) {
            const int j = y * 256 + (int) threadIdx.x;
            if (j < take) {
                pay[j] = eReg[y];
                keys[j] = kReg[y];
            }
        }
        if (threadIdx.x == 0) sTake = take;
    }
    __syncthreads();
    const int take = sTake;
    int np2 = 1;
    while (np2 < take) np2 <<= 1;
    for (int x = take + threadIdx.x; x < np2; x += blockDim.x) { keys[x] = ~0ull; pay[x] = 0xFFFFFFFFu; }
    __syncthreads();
    if (np2 > 1) bitonicSort(keys, pay, np2);
    __syncthreads();
    {
        // coverage pre-filter (Util::canBeCovered, Util.cpp:477-494; only the modes runSplit applies) and the rows,
        // compacted in order: every thread owns a run of consecutive entries
        sd_hit *o = outHits + (size_t) q * maxHits;
        const float qLen = (float) (qOff[q + 1] - qOff[q]);
        auto covered = [&](uint32_t sid) {
            if (!(covThr > 0.0f)) return true;
            const float tLen = (float) (tOff[sid + 1] - tOff[sid]);
            switch (covMode) {
                case 0: return (qLen / tLen >= covThr) && (tLen / qLen >= covThr);
                case 2: return (tLen / qLen) >= covThr;
                case 5: return (fminf(tLen, qLen) / fmaxf(tLen, qLen)) >= covThr;
                default: return true;
            }
        };
        const uint32_t w0 = (ident != 0xFFFFFFFFu && covered(ident)) ? 1u : 0u;
        if (threadIdx.x == 0 && w0) { o[0].seqId = ident; o[0].score = 65535; o[0].diagonal = 0; o[0].pad = 0; }
        constexpr int PERO = (SEL_CAP / 2 + 255) / 256;
        __shared__ uint32_t oPart[5];
        const int b0 = (int) threadIdx.x * PERO;
        uint32_t sidReg[PERO], mask = 0, cnt = 0;
#pragma unroll
        for (int y = 0; y < PERO; y++) {
            const int x = b0 + y;
            if (x < take) {
                sidReg[y] = kKey[pay[x]] & ((1u << tBits) - 1);
                if (covered(sidReg[y])) { mask |= 1u << y; cnt++; }
            }
        }
        uint32_t incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o2 = __shfl_up(incl, off, 64);
            if ((threadIdx.x & 63) >= off) incl += o2;
        }
        if ((threadIdx.x & 63) == 63) oPart[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t w = w0 + incl - cnt;
        for (int wv = 0; wv < (int) (threadIdx.x >> 6); wv++) w += oPart[wv];
#pragma unroll
        for (int y = 0; y < PERO; y++) {
            if (mask & (1u << y)) {
                const int x = b0 + y;
                const uint32_t e = pay[x];
                o[w].seqId = sidReg[y];
                o[w].score = (int32_t) (0x7FFFFFFFu - (uint32_t) (keys[x] >> 32));   // as ordered
                o[w].diagonal = diagOf(ds, q, kVal[e] & posMask, sidReg[y]);
                o[w].pad = 0;
                w++;
            }
        }
        if (threadIdx.x == 255) outCount[q] = w;   // the last thread's end = total
    }
}

// K7b: result lists of any length (the reference caps a list at maxHitsPerQuery = min(--max-seqs, dbSize) and nothing else,
// QueryMatcher.cpp:45-46,364-420; clustersearch asks for --max-seqs > the number of target sets, R/src/workflow/clustersearch.cpp:16-17).
// select_hits_kernel keeps the candidates at or above the score cut in an LDS sorter; a query whose list is longer than half
// that sorter AND has more such candidates than it holds comes here, one workgroup per query:
//   1. the same cut (histogram, computeScoreThreshold, rescoring when the cut saturates);
//   2. the `need` = maxHits - [query is a target] first candidates of the cut order among those that are not the identity
//      target, found without sorting: most-significant-digit radix selection of the need-th smallest 64-bit order key
//      (keys are unique: they end in the candidate's place in the hit stream), eight counting passes over the candidates;
//   3. the selected candidates go to a global scratch slot with their final order key (score desc, sequence id asc);
//   4. bitonic sort of that slot: strides of a tile and more on global memory, everything below in LDS tiles;
//   5. coverage pre-filter and the rows, compacted in order.
constexpr int SELB_NT = 1024;
constexpr int SELB_TILE = 4096;

__global__ void __launch_bounds__(SELB_NT)
select_hits_big_kernel(const uint32_t *__restrict__ bigList, const uint32_t *__restrict__ kStartOfQ,
                       const uint32_t *__restrict__ kKey, const uint32_t *__restrict__ kVal, const int32_t *__restrict__ kScore,
                       const DiagSrc ds, int tBits, uint32_t binMask, int maxHits, int minDiag,
                       const uint32_t *__restrict__ identityId, const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff,
                       int covMode, float covThr, const uint8_t *__restrict__ qRes, const int8_t *__restrict__ diagBias,
                       const int8_t *__restrict__ mat, sd_hit *__restrict__ outHits, uint32_t *__restrict__ outCount,
                       const int8_t *__restrict__ qProf, uint32_t posMask, int joinBinBits, const uint32_t *__restrict__ qSplit,
                       const uint32_t *__restrict__ qParts, const uint32_t *__restrict__ qSplits,
                       unsigned long long *scrKeys, uint32_t *scrPay, uint32_t scrStride /* power of two >= maxHits */) {
    __shared__ unsigned long long lk[SELB_TILE];
    __shared__ uint32_t lp[SELB_TILE];
    __shared__ unsigned int hist[256];
    __shared__ int sThr, sMaxSelf;
    __shared__ unsigned long long sPrefix;
    __shared__ uint32_t sRemain, sCnt, sU;
    __shared__ uint32_t oPart[SELB_NT / 64 + 1];
    const uint32_t q = bigList[blockIdx.x];
    unsigned long long *gk = scrKeys + (size_t) blockIdx.x * scrStride;
    uint32_t *gp = scrPay + (size_t) blockIdx.x * scrStride;
    const uint32_t beg = kStartOfQ[q], end = kStartOfQ[q + 1];
    const uint32_t ident = identityId[q];
    const uint32_t tMask = (1u << tBits) - 1;
    const int t = threadIdx.x;
    for (int x = t; x < 256; x += SELB_NT) hist[x] = 0;
    if (t == 0) sU = 0;
    __syncthreads();
    for (uint32_t x = beg + t; x < end; x += SELB_NT) atomicAdd(&hist[min(255, kScore[x])], 1u);
    __syncthreads();
    if (t == 0) {
        size_t found = 0;
        int thr = 255;
        for (thr = 255; thr > 0; thr--) {
            found += hist[thr];
            if (found >= (size_t) (uint32_t) maxHits) break;
        }
        thr = max(minDiag, thr);
        sThr = thr;
        sMaxSelf = 1;
        if (thr >= 255) {   // rescoreHits (QueryMatcher.cpp:525-544): see select_hits_kernel
            const uint8_t *qs = qRes + qOff[q];
            const int8_t *qb = diagBias + qOff[q];
            const int qL = (int) (qOff[q + 1] - qOff[q]);
            int score = 0, best = 0;
            for (int x = 0; x < qL; x++) {
                score += qProf ? (int) qProf[(qOff[q] + (uint64_t) x) * 21 + qs[x]] : (int) (int8_t) (mat[qs[x] * 21 + qs[x]] + qb[x]);
                score = score < 0 ? 0 : score;
                best = score > best ? score : best;
            }
            int ms = best - 255;
            ms = ms < 1 ? 1 : ms;
            ms = ms > 65535 ? 65535 : ms;
            sMaxSelf = ms;
        }
    }
    __syncthreads();
    const int thr = sThr;
    const bool rescored = thr >= 255;
    auto rescaledByte = [&](uint32_t e) -> uint32_t {
        unsigned int ns = (unsigned int) kScore[e] - 255u;
        const float sc = (float) min(ns, 65535u);
        const double dv = (double) ((sc / (float) sMaxSelf) * 255.0f) + 0.5;
        return (uint32_t) (uint8_t) (int) dv;
    };
    const SplitView sv = splitView(qSplit, qParts, qSplits, q);
    auto keyOf = [&](uint32_t x) -> unsigned long long {   // the order of the cut, as in select_hits_kernel
        const uint32_t sid = kKey[x] & tMask;
        const uint32_t top = rescored ? 255u - rescaledByte(x) : 255u - (uint32_t) min(255, kScore[x]);
        bool backwards = false;
        const uint32_t pos = sv.all ? overflowOrderPos(sv, kVal[x] & posMask, backwards) : kVal[x] & posMask;
        if (joinBinBits >= 0) {
            const uint32_t inList = (sid >> joinBinBits) & 0xFFFFFu;
            return ((unsigned long long) top << 56) | ((unsigned long long) (sid & binMask) << 44) |
                   ((unsigned long long) pos << 20) | (unsigned long long) (backwards ? 0xFFFFFu - inList : inList);
        }
        return ((unsigned long long) top << 56) | ((unsigned long long) (sid & binMask) << 40) | (unsigned long long) pos;
    };
    auto inUniverse = [&](uint32_t x) -> bool { return min(255, kScore[x]) >= thr && (kKey[x] & tMask) != ident; };
    {
        uint32_t mine = 0;
        for (uint32_t x = beg + t; x < end; x += SELB_NT) mine += inUniverse(x) ? 1u : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
        if ((t & 63) == 0 && mine) atomicAdd(&sU, mine);
    }
    __syncthreads();
    const uint32_t U = sU;
    const int current0 = (ident != 0xFFFFFFFFu) ? 1 : 0;
    const uint32_t need = min(U, (uint32_t) max(0, maxHits - current0));
    unsigned long long T = ~0ull;   // keys <= T are selected
    if (U > need && need > 0) {
        if (t == 0) {
            sPrefix = 0;
            sRemain = need;
        }
        for (int shift = 56; shift >= 0; shift -= 8) {
            for (int x = t; x < 256; x += SELB_NT) hist[x] = 0;
            __syncthreads();
            const unsigned long long prefix = sPrefix;
            for (uint32_t x = beg + t; x < end; x += SELB_NT) {
                if (!inUniverse(x)) continue;
                const unsigned long long kk = keyOf(x);
                if (shift == 56 || (kk >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t) (kk >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (t == 0) {
                uint32_t cum = 0, d = 0;
                for (d = 0; d < 255; d++) {
                    if (cum + hist[d] >= sRemain) break;
                    cum += hist[d];
                }
                sPrefix = (prefix << 8) | d;
                sRemain -= cum;
            }
            __syncthreads();
        }
        T = sPrefix;
    }
    if (t == 0) sCnt = 0;
    __syncthreads();
    if (need > 0) {
        for (uint32_t x = beg + t; x < end; x += SELB_NT) {
            if (!inUniverse(x)) continue;
            if (T != ~0ull && keyOf(x) > T) continue;
            const uint32_t slot = atomicAdd(&sCnt, 1u);
            if (slot < scrStride) {
                const uint32_t sid = kKey[x] & tMask;
                const int sc = kScore[x];
                const int cnt = min(255, sc);
                const int prefScore = rescored ? (int) (255u + rescaledByte(x) * (unsigned int) sMaxSelf / 255u) : (cnt >= 255 ? sc : cnt);
                gk[slot] = ((unsigned long long) (0x7FFFFFFFu - (uint32_t) prefScore) << 32) | sid;
                gp[slot] = x;
            }
        }
    }
    __syncthreads();
    const uint32_t take = min(min(sCnt, need), scrStride);
    uint32_t np2 = 1;
    while (np2 < take) np2 <<= 1;
    for (uint32_t x = take + t; x < np2; x += SELB_NT) {
        gk[x] = ~0ull;
        gp[x] = 0xFFFFFFFFu;
    }
    __syncthreads();
    // ---- bitonic sort of gk / gp [0, np2), ascending
    auto loadTile = [&](uint32_t base, uint32_t cnt) {
        for (uint32_t x = t; x < cnt; x += SELB_NT) {
            lk[x] = gk[base + x];
            lp[x] = gp[base + x];
        }
        __syncthreads();
    };
    auto storeTile = [&](uint32_t base, uint32_t cnt) {
        __syncthreads();
        for (uint32_t x = t; x < cnt; x += SELB_NT) {
            gk[base + x] = lk[x];
            gp[base + x] = lp[x];
        }
        __syncthreads();
    };
    auto ldsSteps = [&](uint32_t base, uint32_t cnt, uint32_t size, uint32_t strideFrom) {   // strides strideFrom .. 1 of stage `size`
        for (uint32_t stride = strideFrom; stride > 0; stride >>= 1) {
            __syncthreads();
            for (uint32_t p = t; p < cnt / 2; p += SELB_NT) {
                const uint32_t lo = (p / stride) * stride * 2 + (p % stride), hi = lo + stride;
                const bool asc = ((base + lo) & size) == 0;
                const unsigned long long a = lk[lo], b = lk[hi];
                if ((a > b) == asc) {
                    lk[lo] = b;
                    lk[hi] = a;
                    const uint32_t pa = lp[lo];
                    lp[lo] = lp[hi];
                    lp[hi] = pa;
                }
            }
        }
    };
    if (np2 > 1) {
        const uint32_t tile = min(np2, (uint32_t) SELB_TILE);
        for (uint32_t base = 0; base < np2; base += tile) {
            loadTile(base, tile);
            for (uint32_t size = 2; size <= tile; size <<= 1) ldsSteps(base, tile, size, size >> 1);
            storeTile(base, tile);
        }
        for (uint32_t size = tile * 2; size <= np2; size <<= 1) {
            for (uint32_t stride = size >> 1; stride >= tile; stride >>= 1) {
                for (uint32_t p = t; p < np2 / 2; p += SELB_NT) {
                    const uint32_t lo = (p / stride) * stride * 2 + (p % stride), hi = lo + stride;
                    const bool asc = (lo & size) == 0;
                    const unsigned long long a = gk[lo], b = gk[hi];
                    if ((a > b) == asc) {
                        gk[lo] = b;
                        gk[hi] = a;
                        const uint32_t pa = gp[lo];
                        gp[lo] = gp[hi];
                        gp[hi] = pa;
                    }
                }
                __syncthreads();
            }
            for (uint32_t base = 0; base < np2; base += tile) {
                loadTile(base, tile);
                ldsSteps(base, tile, size, tile >> 1);
                storeTile(base, tile);
            }
        }
    }
    __syncthreads();
    // ---- rows: coverage pre-filter, compacted in order (every thread owns a run of consecutive entries)
    sd_hit *o = outHits + (size_t) q * maxHits;
    const float qLen = (float) (qOff[q + 1] - qOff[q]);
    auto covered = [&](uint32_t sid) {
        if (!(covThr > 0.0f)) return true;
        const float tLen = (float) (tOff[sid + 1] - tOff[sid]);
        switch (covMode) {
            case 0: return (qLen / tLen >= covThr) && (tLen / qLen >= covThr);
            case 2: return (tLen / qLen) >= covThr;
            case 5: return (fminf(tLen, qLen) / fmaxf(tLen, qLen)) >= covThr;
            default: return true;
        }
    };
    const uint32_t w0 = (ident != 0xFFFFFFFFu && covered(ident)) ? 1u : 0u;
    if (t == 0 && w0) { o[0].seqId = ident; o[0].score = 65535; o[0].diagonal = 0; o[0].pad = 0; }
    const uint32_t per = (take + SELB_NT - 1) / SELB_NT;
    const uint32_t b0 = min(take, (uint32_t) t * per), b1 = min(take, b0 + per);
    uint32_t cnt = 0;
    for (uint32_t x = b0; x < b1; x++) cnt += covered((uint32_t) (gk[x] & 0xFFFFFFFFull)) ? 1u : 0u;
    uint32_t incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o2 = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o2;
    }
    if ((t & 63) == 63) oPart[t >> 6] = incl;
    __syncthreads();
    uint32_t w = w0 + incl - cnt;
    for (int wv = 0; wv < (t >> 6); wv++) w += oPart[wv];
    for (uint32_t x = b0; x < b1; x++) {
        const unsigned long long kf = gk[x];
        const uint32_t sid = (uint32_t) (kf & 0xFFFFFFFFull);
        if (covered(sid)) {
            o[w].seqId = sid;
            o[w].score = (int32_t) (0x7FFFFFFFu - (uint32_t) (kf >> 32));
            o[w].diagonal = diagOf(ds, q, kVal[gp[x]] & posMask, sid);
            o[w].pad = 0;
            w++;
        }
    }
    if (t == SELB_NT - 1) outCount[q] = w;   // the last thread's end = total
}

__global__ void compact_kernel(uint64_t n, const uint8_t *__restrict__ flag, const uint64_t *__restrict__ pos,
                               const uint32_t *__restrict__ inKey, const uint32_t *__restrict__ inVal,
                               const int32_t *__restrict__ inScore, uint32_t *__restrict__ outKey,
                               uint32_t *__restrict__ outVal, int32_t *__restrict__ outScore) {
    const uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n || !flag[s]) return;
    const uint64_t w = pos[s];
    outKey[w] = inKey[s];
    outVal[w] = inVal[s];
    if (inScore) outScore[w] = inScore[s];
}

__global__ void flag_to_u64_kernel(uint64_t n, const uint8_t *__restrict__ flag, uint64_t *__restrict__ out) {
    const uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) out[s] = flag[s];
}

__global__ void widen_kernel(uint64_t n, const uint32_t *__restrict__ in, uint64_t *__restrict__ out) {
    const uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) out[s] = in[s];
}

// first kept index of every query (keys are sorted): start[q] = lower_bound(q << tBits)
__global__ void query_bounds_kernel(uint32_t nQ, uint32_t nKept, const uint64_t *__restrict__ nKeptPtr /* non-null: the count as the scan left it */,
                                    const uint32_t *__restrict__ kKey, int tBits, uint32_t *__restrict__ start) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nQ) return;
    if (nKeptPtr) nKept = (uint32_t) *nKeptPtr;
    if (q == nQ) { start[q] = nKept; return; }
    uint32_t lo = 0, hi = nKept;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if ((kKey[mid] >> tBits) < q) lo = mid + 1;
        else hi = mid;
    }
    start[q] = lo;
}

// per-query statistics from per-position / per-candidate data
__global__ void stats_kernel(uint32_t nQ, const uint64_t *__restrict__ posBase, const uint64_t *__restrict__ kmerBase,
                             const uint64_t *__restrict__ hitBase, uint64_t nKmers, uint64_t nHits,
                             uint64_t *__restrict__ stats) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nQ) return;
    const uint64_t p0 = posBase[q], p1 = posBase[q + 1];
    const uint64_t k0 = kmerBase[p0], k1 = kmerBase[p1];   // kmerBase has nPos+1 entries
    stats[4 * q] = k1 - k0;
    const uint64_t h0 = k0 < nKmers ? hitBase[k0] : nHits, h1 = k1 < nKmers ? hitBase[k1] : nHits;
    stats[4 * q + 1] = h1 - h0;
}

__global__ void cand_stats_kernel(uint32_t nCand, const uint32_t *__restrict__ cKey, const uint32_t *__restrict__ cLen,
                                  int tBits, unsigned long long *__restrict__ stats) {
    // candidates are ordered by query: a wavefront nearly always holds one query, so it adds its sums with one pair of
    // atomics (64 contended 64-bit atomics per wavefront made this diagnostic kernel cost as much as score_diag)
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = c < nCand;
    const uint32_t q = live ? cKey[c] >> tBits : 0xFFFFFFFFu;
    const unsigned long long len = live ? (unsigned long long) cLen[c] : 0ull;
    const uint32_t q0 = __shfl(q, 0, 64);
    if (__all(q == q0 || !live)) {
        unsigned long long cnt = live ? 1ull : 0ull, sum = len;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            cnt += __shfl_xor(cnt, off, 64);
            sum += __shfl_xor(sum, off, 64);
        }
        if ((threadIdx.x & 63) == 0 && q0 != 0xFFFFFFFFu) {
            atomicAdd(&stats[4 * (size_t) q0 + 2], cnt);
            atomicAdd(&stats[4 * (size_t) q0 + 3], sum);
        }
    } else if (live) {
        atomicAdd(&stats[4 * (size_t) q + 2], 1ull);
        atomicAdd(&stats[4 * (size_t) q + 3], len);
    }
}

// device-wide scans and the radix sort: this library's own kernels (sd_scan_sort.h)
#include "sd_scan_sort.h"
template <typename T, typename Tmp>
int exclusiveScanWiden(sd_ctx *ctx, const T *in, uint64_t *out, uint64_t n, Tmp &tmp) {
    if (n == 0) return SD_OK;
    if (tmp.n < sdScanTmpBytes(n)) SD_HIP(ctx, tmp.alloc(sdScanTmpBytes(n) + 256));
    SD_HIP(ctx, (sdScanLaunch<T, ScanSum64, false, uint64_t>(ctx->stream, in, out, n, (uint64_t *) tmp.p)));
    return SD_OK;
}
template <typename T, typename Tmp>
int exclusiveScan(sd_ctx *ctx, const T *in, uint64_t *out, uint64_t n, Tmp &tmp) {
    return exclusiveScanWiden(ctx, in, out, n, tmp);
}
// stable sort of (key, value) pairs by the key bits [beginBit, endBit) into (kOut, vOut); scratch in the workspace
int sortPairs(sd_ctx *ctx, const uint32_t *kIn, uint32_t *kOut, const uint32_t *vIn, uint32_t *vOut, uint64_t n, int beginBit, int endBit) {
    if (n == 0) return SD_OK;
    if (n >= 0xFFFFFFFFull) return sdFail(ctx, SD_EUNSUPPORTED, "sort of %llu elements", (unsigned long long) n);
    WsView<uint32_t> kTmp(ctx, "pf.sortK"), vTmp(ctx, "pf.sortV"), cnt(ctx, "pf.sortCounts");
    SD_HIP(ctx, kTmp.alloc(n));
    SD_HIP(ctx, vTmp.alloc(n));
    SD_HIP(ctx, cnt.alloc(sdRadixSortCountsBytes() / sizeof(uint32_t)));
    SD_HIP(ctx, sdRadixSortPairs(ctx->stream, kIn, vIn, kOut, vOut, kTmp.p, vTmp.p, (uint32_t) n, beginBit, endBit, cnt.p));
    return SD_OK;
}

inline unsigned gridFor(uint64_t n, unsigned block) { return (unsigned) ((n + block - 1) / block); }

}  // namespace

// The cumulative score table of the 3-mer rows (see countGETab): built once per target, 4 MB.  Scores that span more than 255
// values leave dExt3Cum null and the kernels keep searching the sorted rows.
int sdBuildExt3Cum(sd_ctx *ctx, sd_target *t) {
    if (!t || !t->dExt3Score || t->dExt3Cum) return SD_OK;
    if (getenv("SD_PF_CUM") && atoi(getenv("SD_PF_CUM")) == 0) return SD_OK;
    int *dMM = nullptr;
    SD_HIP(ctx, hipMalloc((void **) &dMM, 2 * sizeof(int)));
    const int init[2] = {32767, -32768};
    int mm[2] = {0, 0};
    hipError_t e = hipMemcpy(dMM, init, sizeof(init), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ext3_minmax_kernel, dim3(2048), dim3(256), 0, ctx->stream, (const int16_t *) t->dExt3Score, (uint64_t) 8000 * 8000, dMM);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(mm, dMM, sizeof(mm), hipMemcpyDeviceToHost);
    (void) hipFree(dMM);
    SD_HIP(ctx, e);
    if (mm[1] - mm[0] + 1 >= EXT3_CUM_SPAN) return SD_OK;   // (c = hi - lo + 1 must read 0: it has to be inside the table)
    SD_HIP(ctx, hipMalloc((void **) &t->dExt3Cum, (size_t) 8000 * EXT3_CUM_SPAN * sizeof(uint16_t)));
    t->ext3Lo = mm[0];
    hipLaunchKernelGGL(ext3_cum_kernel, dim3(8000), dim3(256), 0, ctx->stream, (const int16_t *) t->dExt3Score, mm[0], t->dExt3Cum);
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SD_OK;
}

extern "C" {

int sd_target_create(sd_ctx *ctx, int kmerSize, const uint32_t *kmerOffsets, const uint32_t *entrySeq,
                     const uint16_t *entryPos, uint64_t nEntries, const uint8_t *maskedResidues,
                     const uint64_t *seqOffsets, uint32_t nSeq, const int16_t *ext2Score, const uint16_t *ext2Index,
                     const int16_t *ext3Score, const uint16_t *ext3Index, sd_target **out) {
    return sd_target_create_wide(ctx, kmerSize, kmerOffsets, nullptr, entrySeq, entryPos, nEntries, maskedResidues, seqOffsets, nSeq,
                                 ext2Score, ext2Index, ext3Score, ext3Index, out);
}

int sd_target_create_wide(sd_ctx *ctx, int kmerSize, const uint32_t *kmerOffsets, const uint64_t *kmerBlockBase,
                          const uint32_t *entrySeq, const uint16_t *entryPos, uint64_t nEntries, const uint8_t *maskedResidues,
                          const uint64_t *seqOffsets, uint32_t nSeq, const int16_t *ext2Score, const uint16_t *ext2Index,
                          const int16_t *ext3Score, const uint16_t *ext3Index, sd_target **out) {
    if (!ctx || !out || !kmerOffsets || !maskedResidues || !seqOffsets || !ext3Score || !ext3Index) return SD_EINVAL;
    if (kmerSize != 6 && kmerSize != 7) return sdFail(ctx, SD_EUNSUPPORTED, "k=%d: the device implements k=6 and k=7", kmerSize);
    if (kmerSize == 7 && (!ext2Score || !ext2Index)) return sdFail(ctx, SD_EINVAL, "k=7 needs the 2-mer score matrix");
    if (nEntries > 0xFFFFFFFFull && !kmerBlockBase)
        return sdFail(ctx, SD_EINVAL, "an index of %llu entries needs the block bases of a wide index (sd_host_index_block_base)",
                      (unsigned long long) nEntries);
    for (uint32_t i = 0; i < nSeq; i++)
        if (seqOffsets[i + 1] - seqOffsets[i] > 65535)
            return sdFail(ctx, SD_EINVAL, "target %u has %llu residues; index positions are 16 bit (limit 65535, --max-seq-len)", i,
                          (unsigned long long) (seqOffsets[i + 1] - seqOffsets[i]));
    (void) hipSetDevice(ctx->device);
    sd_target *t = new sd_target();
    t->ctx = ctx;
    t->k = kmerSize;
    t->nSeq = nSeq;
    t->nEntries = nEntries;
    t->tableSize = kmerSize == 6 ? 64000000ull : 1280000000ull;   // 20^k
    t->hSeqOff.assign(seqOffsets, seqOffsets + nSeq + 1);
    const uint64_t total = seqOffsets[nSeq];
    auto up = [&](void **d, const void *h, size_t bytes) -> bool {
        if (hipMalloc(d, bytes + 64) != hipSuccess) return false;
        return hipMemcpy(*d, h, bytes, hipMemcpyDefault) == hipSuccess;   // host or device source (an index received device-to-device)
    };
    bool ok = up((void **) &t->dOffsets, kmerOffsets, (t->tableSize + 1) * sizeof(uint32_t));
    if (kmerBlockBase) ok = ok && up((void **) &t->dBlockBase, kmerBlockBase, (((t->tableSize + 2) >> 16) + 1) * sizeof(uint64_t));
    ok = ok && up((void **) &t->dEntrySeq, entrySeq, std::max<uint64_t>(nEntries, 1) * sizeof(uint32_t));
    ok = ok && up((void **) &t->dEntryPos, entryPos, std::max<uint64_t>(nEntries, 1) * sizeof(uint16_t));
    ok = ok && up((void **) &t->dMasked, maskedResidues, std::max<uint64_t>(total, 1));
    ok = ok && up((void **) &t->dSeqOff, seqOffsets, (nSeq + 1) * sizeof(uint64_t));
    ok = ok && up((void **) &t->dExt3Score, ext3Score, (size_t) 8000 * 8000 * sizeof(int16_t));
    ok = ok && up((void **) &t->dExt3Index, ext3Index, (size_t) 8000 * 8000 * sizeof(uint16_t));
    if (ext2Score && ext2Index) {
        ok = ok && up((void **) &t->dExt2Score, ext2Score, (size_t) 400 * 400 * sizeof(int16_t));
        ok = ok && up((void **) &t->dExt2Index, ext2Index, (size_t) 400 * 400 * sizeof(uint16_t));
    }
    // SD_INDEX_TEST_SHIFT (tests only, wide indexes): the entries start that many slots into their buffer and every block
    // base moves with them, so that list starts beyond 2^32 are exercised without an index of that size
    uint64_t testShift = 0;
    if (kmerBlockBase && getenv("SD_INDEX_TEST_SHIFT")) testShift = strtoull(getenv("SD_INDEX_TEST_SHIFT"), nullptr, 10);
    ok = ok && hipMalloc((void **) &t->dEntries, (std::max<uint64_t>(nEntries, 1) + testShift + 8) * sizeof(uint2)) == hipSuccess;
    if (ok && testShift) {
        std::vector<uint64_t> bb(kmerBlockBase, kmerBlockBase + ((t->tableSize + 2) >> 16) + 1);
        for (uint64_t &v : bb) v += testShift;
        ok = hipMemcpy(t->dBlockBase, bb.data(), bb.size() * sizeof(uint64_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        sd_target_destroy(t);
        return sdFail(ctx, SD_ENOMEM, "sd_target_create: device allocation/upload failed");
    }
    if (nEntries > 0) {
        hipLaunchKernelGGL(interleave_entries_kernel, dim3((unsigned) std::min<uint64_t>((nEntries + 255) / 256, 1u << 20)), dim3(256), 0, ctx->stream, nEntries,
                           t->dEntrySeq, t->dEntryPos, t->dEntries + testShift);
        if (sdStreamSync(ctx) != hipSuccess) {
            sd_target_destroy(t);
            return sdFail(ctx, SD_EHIP, "sd_target_create: interleaving the index entries failed");
        }
    }
    (void) hipFree(t->dEntrySeq);
    (void) hipFree(t->dEntryPos);
    t->dEntrySeq = nullptr;
    t->dEntryPos = nullptr;
    if (sdBuildExt3Cum(ctx, t) != SD_OK) {
        sd_target_destroy(t);
        return SD_EHIP;
    }
    *out = t;
    return SD_OK;
}

void sd_target_destroy(sd_target *t) {
    if (!t) return;
    void *ptrs[] = {t->dOffsets, t->dBlockBase, t->dEntrySeq, t->dEntryPos, t->dMasked, t->dSeqOff, t->dExt3Score, t->dExt3Index,
                    t->dExt2Score, t->dExt2Index, t->dEntries, t->dExt3Cum};
    for (void *p : ptrs)
        if (p) (void) hipFree(p);
    delete t;
}

namespace {
struct ProfileQueries {          // profile queries: Sequence::mapProfile's outputs, concatenated like the letters
    const int16_t *sortedScore;  // [position][20] descending
    const uint8_t *sortedIndex;  // [position][20] amino acids in that order
    const int8_t *aln;           // [position][21] alignment profile
};
}  // namespace

static int prefilterBatchImpl(sd_ctx *ctx, const sd_target *T, const sd_prefilter_params *par, uint32_t nQ,
                              const uint8_t *qResidues, const uint64_t *qOffsets, const int16_t *qKmerBias,
                              const int8_t *qDiagBias, const uint32_t *identityId, sd_hit *outHits, uint32_t *outCount,
                              uint64_t *stats, const ProfileQueries *prof) {
    if (!ctx || !T || !par || !qResidues || !qOffsets || !identityId || !outHits || !outCount) return SD_EINVAL;
    if (!prof && (!qKmerBias || !qDiagBias)) return SD_EINVAL;
    if (par->kmerSize != T->k) return sdFail(ctx, SD_EINVAL, "k-mer size mismatch");
    if (par->minDiagScore < 1) return sdFail(ctx, SD_EUNSUPPORTED, "minDiagScore must be >= 1");
    if (par->binSize == 0 || (par->binSize & (par->binSize - 1))) return sdFail(ctx, SD_EINVAL, "binSize must be a power of two");
    (void) hipSetDevice(ctx->device);
    sdD2HReset(ctx);   // (reads an earlier, failed call left pending)
    const int maxHits = (int) std::min<uint64_t>((uint64_t) par->maxHitsPerQuery, T->nSeq);
    int tBits = 1;
    while ((1ull << tBits) < T->nSeq) tBits++;
    const uint32_t maxBatchQ = 1u << std::min(32 - tBits, 16);
    const uint64_t maxDbMatches = std::max<uint64_t>(1000000, T->nSeq) * 2;   // QueryMatcher.cpp:43-47
    // Hits per sub-batch.  The k-mer stream of a sub-batch carries 30-bit stream indices (jpElem) and the lookup path keeps
    // 18 - 26 B per hit in its workspace, so those two stay at 2^30; the JOIN path's hit stream is bounded by its 32-bit
    // write cursors alone (< 0xFFFFFFF0 hits, checked where the sub-batch is sized) -- its tables (offsets + entries: 7.1 GB
    // at 1 000 proteomes) are walked once per sub-batch by join_count / join_scatter, so the fewer sub-batches the better:
    // against 3 * 10^6 targets a sub-batch is now maxBatchQ = 1 024 queries (1.5 * 10^9 hits) where the old bound of 2^30
    // (a limit of the library sort that left the hot path in round 5) cut it into 375-query pieces.  SD_PF_HIT_BUDGET: hits.
    const uint64_t KMER_BUDGET = 1ull << 30, LOOKUP_HIT_BUDGET = 1ull << 30;
    uint64_t HIT_BUDGET = 3ull << 30;
    if (const char *e = getenv("SD_PF_HIT_BUDGET")) HIT_BUDGET = std::max<uint64_t>(1ull << 16, std::min<uint64_t>(0xF0000000ull, (uint64_t) atoll(e)));

    // (workspace views, not allocations of this call: a hipFree at the end of every call waits for all streams of the device,
    // i.e. for the other lanes of a pipeline)
    WsView<int8_t> dMat(ctx, "pf.mat");
    SD_HIP(ctx, dMat.alloc(448));
    SD_HIP(ctx, hipMemcpyAsync(dMat.p, par->ungappedMatrix, 441, hipMemcpyHostToDevice, ctx->stream));
    WsView<int> dErr(ctx, "pf.err");
    SD_HIP(ctx, dErr.alloc(1));
    WsView<uint8_t> scanTmp(ctx, "pf.scanTmp"), sortTmp(ctx, "pf.sortTmp");

    for (uint32_t x = 0; x < nQ; x++)
        if (qOffsets[x + 1] - qOffsets[x] > 65535)
            return sdFail(ctx, SD_EINVAL, "query %u has %llu residues; k-mer positions are 16 bit (limit 65535, --max-seq-len)", x,
                          (unsigned long long) (qOffsets[x + 1] - qOffsets[x]));
    uint32_t qBeg = 0;
    bool forceLookup = false;   // this sub-batch is redone on the lookup path (a regime the join path does not carry)
    // queries per sub-batch: the k-mer stream, its sorted copy and the hit stream of a sub-batch are what the prefilter keeps in
    // HBM (~3.4 MB per query on a proteome-scale target); the tables are walked once per sub-batch (~1 GB at 100 proteomes)
    uint32_t batchQ = std::min<uint32_t>(maxBatchQ, 2048);
    if (const char *e = getenv("SD_PF_BATCH")) batchQ = std::max<uint32_t>(1, std::min<uint32_t>(maxBatchQ, (uint32_t) atoi(e)));
    while (qBeg < nQ) {
        // the rest of the call in equal sub-batches of at most batchQ queries
        const uint32_t left = nQ - qBeg, parts = (left + batchQ - 1) / batchQ;
        uint32_t bq = (left + parts - 1) / parts;
        std::unique_ptr<HostScope> hs(new HostScope(ctx, "pf.upload_count"));
        // ---- upload the sub-batch
        const uint64_t r0 = qOffsets[qBeg], r1 = qOffsets[qBeg + bq];
        std::vector<uint64_t> hOff(bq + 1), hPos(bq + 1);
        hPos[0] = 0;
        for (uint32_t x = 0; x <= bq; x++) hOff[x] = qOffsets[qBeg + x] - r0;
        for (uint32_t x = 0; x < bq; x++) {
            const int64_t L = (int64_t) (hOff[x + 1] - hOff[x]);
            hPos[x + 1] = hPos[x] + (uint64_t) std::max<int64_t>(0, L - (T->k == 6 ? SPAN6 : SPAN7) + 1);
        }
        const uint64_t nPos = hPos[bq];
        if (nPos * 64 >= (1ull << 32) && bq > 1) {   // a wavefront per position: a dispatch carries at most 2^32 work-items
            batchQ = std::max<uint32_t>(1, bq / 2);
            continue;
        }
        WsView<uint8_t> dQ(ctx, "pf.dQ");
        WsView<int16_t> dKB(ctx, "pf.dKB");
        WsView<int8_t> dDB(ctx, "pf.dDB");
        WsView<uint64_t> dQOff(ctx, "pf.dQOff");
        WsView<uint64_t> dPosBase(ctx, "pf.dPosBase");
        WsView<uint32_t> dIdent(ctx, "pf.dIdent");
        SD_HIP(ctx, dQ.alloc(r1 - r0 + 64));
        SD_HIP(ctx, dKB.alloc(r1 - r0 + 64));
        SD_HIP(ctx, dDB.alloc(r1 - r0 + 64));
        SD_HIP(ctx, dQOff.alloc(bq + 1));
        SD_HIP(ctx, dPosBase.alloc(bq + 1));
        SD_HIP(ctx, dIdent.alloc(bq));
        SD_HIP(ctx, hipMemcpyAsync(dQ.p, qResidues + r0, r1 - r0, hipMemcpyHostToDevice, ctx->stream));
        WsView<int16_t> dPS(ctx, "pf.dProfScore");
        WsView<uint8_t> dPI(ctx, "pf.dProfIndex");
        WsView<int8_t> dPA(ctx, "pf.dProfAln");
        const int8_t *dProfAln = nullptr;
        if (prof) {
            SD_HIP(ctx, dPS.alloc((r1 - r0) * 20 + 64));
            SD_HIP(ctx, dPI.alloc((r1 - r0) * 20 + 64));
            SD_HIP(ctx, dPA.alloc((r1 - r0) * 21 + 64));
            SD_HIP(ctx, hipMemcpyAsync(dPS.p, prof->sortedScore + r0 * 20, (r1 - r0) * 20 * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
            SD_HIP(ctx, hipMemcpyAsync(dPI.p, prof->sortedIndex + r0 * 20, (r1 - r0) * 20, hipMemcpyHostToDevice, ctx->stream));
            SD_HIP(ctx, hipMemcpyAsync(dPA.p, prof->aln + r0 * 21, (r1 - r0) * 21, hipMemcpyHostToDevice, ctx->stream));
            dProfAln = dPA.p;
        } else {
            SD_HIP(ctx, hipMemcpyAsync(dKB.p, qKmerBias + r0, (r1 - r0) * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
            SD_HIP(ctx, hipMemcpyAsync(dDB.p, qDiagBias + r0, r1 - r0, hipMemcpyHostToDevice, ctx->stream));
        }
        SD_HIP(ctx, hipMemcpyAsync(dQOff.p, hOff.data(), (bq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(ctx, hipMemcpyAsync(dPosBase.p, hPos.data(), (bq + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(ctx, hipMemcpyAsync(dIdent.p, identityId + qBeg, bq * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(ctx, hipMemsetAsync(dErr.p, 0, sizeof(int), ctx->stream));

        uint64_t nKmers = 0, nHits = 0;
        // k-mer-major join (sd_pf_join.h): k = 6, 32-bit list starts, sequence queries, the bucketed match, and a tie order of
        // the result cut that fits its key (sequence id bits above the bin: 20)
        const bool useBuckets = getenv("SD_PF_SORT") == nullptr;
        int binBits = 0;
        while ((1u << binBits) < par->binSize) binBits++;
        // (round 6: profile queries as well -- profile_kmers_kernel writes the join's stream elements; SD_PF_JOIN_PROFILE=0 keeps them on
        // the lookup path)
        const bool joinCandidate = !forceLookup && T->k == 6 && T->dBlockBase == nullptr && useBuckets && bq <= (uint32_t) JQ_MAX &&
                                   (!prof || !(getenv("SD_PF_JOIN_PROFILE") && atoi(getenv("SD_PF_JOIN_PROFILE")) == 0)) &&
                                   tBits - binBits <= 20 && binBits <= 12 && !(getenv("SD_PF_JOIN") && atoi(getenv("SD_PF_JOIN")) == 0);
        WsView<uint32_t> dQKmerBase(ctx, "pf.dQKmerBase");
        WsView<int> dJoinFlag(ctx, "pf.dJoinFlag");
        int hJoinFlag = 0;
        uint32_t profGrid = 0;
        bool profAnyBig = false;
        WsView<uint2> dProfScratch(ctx, "pf.dProfScratch");
        WsView<uint8_t> dProfBig(ctx, "pf.dProfBig");
        WsView<uint32_t> dKmerCount(ctx, "pf.dKmerCount");
        WsView<uint64_t> dKmerBase(ctx, "pf.dKmerBase");
        SD_HIP(ctx, dKmerCount.alloc(nPos + 1));
        SD_HIP(ctx, dKmerBase.alloc(nPos + 1));
        SD_HIP(ctx, hipMemsetAsync(dKmerCount.p, 0, (nPos + 1) * sizeof(uint32_t), ctx->stream));
        WsView<uint32_t> dPosQuery(ctx, "pf.dPosQuery");   // sequence queries: the query of every position, looked up once
        if (nPos > 0 && !prof) {
            SD_HIP(ctx, dPosQuery.alloc(nPos));
            hipLaunchKernelGGL(pos_query_kernel, dim3(gridFor(nPos, 256)), dim3(256), 0, ctx->stream, nPos, (const uint64_t *) dPosBase.p, bq, dPosQuery.p);
        }
        if (nPos > 0) {
            {
                ProfScope ps(ctx, "prefilter_count_kmers");
                if (prof) {
                    profGrid = (uint32_t) std::min<uint64_t>(nPos, PROFILE_GRID);
                    SD_HIP(ctx, dProfScratch.alloc(std::max((size_t) PROFILE_GRID * 2 * PROFILE_PARTIAL_CAP,
                                                            (size_t) PROFILE_GRID_BIG * 2 * PROFILE_PARTIAL_CAP_BIG)));
                    SD_HIP(ctx, dProfBig.alloc(nPos + 1));
                    SD_HIP(ctx, hipMemsetAsync(dProfBig.p, 0, nPos + 1, ctx->stream));
                    hipLaunchKernelGGL(profile_kmers_kernel<false>, dim3(profGrid), dim3(64), 0, ctx->stream, nPos, dPosBase.p, bq,
                                       dQ.p, dQOff.p, dPS.p, dPI.p, par->kmerThr, T->k, dProfScratch.p, dKmerCount.p,
                                       (const uint32_t *) nullptr, (const uint64_t *) nullptr, (uint32_t *) nullptr,
                                       (uint32_t *) nullptr, (uint32_t *) nullptr, dErr.p, PROFILE_PARTIAL_CAP, dProfBig.p, 0,
                                       (const uint64_t *) nullptr, (uint32_t *) nullptr, (uint64_t *) nullptr);
                    int hTier = 0;
                    SD_HIP(ctx, sdD2H(ctx, &hTier, dErr.p, sizeof(int)));
                    SD_HIP(ctx, sdStreamSync(ctx));
                    profAnyBig = hTier == 7;
                    if (profAnyBig) {
                        SD_HIP(ctx, hipMemsetAsync(dErr.p, 0, sizeof(int), ctx->stream));
                        hipLaunchKernelGGL(profile_kmers_kernel<false>, dim3(PROFILE_GRID_BIG), dim3(64), 0, ctx->stream, nPos, dPosBase.p,
                                           bq, dQ.p, dQOff.p, dPS.p, dPI.p, par->kmerThr, T->k, dProfScratch.p, dKmerCount.p,
                                           (const uint32_t *) nullptr, (const uint64_t *) nullptr, (uint32_t *) nullptr,
                                           (uint32_t *) nullptr, (uint32_t *) nullptr, dErr.p, PROFILE_PARTIAL_CAP_BIG, dProfBig.p, 1,
                                           (const uint64_t *) nullptr, (uint32_t *) nullptr, (uint64_t *) nullptr);
                    }
                } else if (T->k == 6)
                    hipLaunchKernelGGL(count_kmers_kernel, dim3(gridFor(nPos, 4)), dim3(256), 0, ctx->stream, nPos, dPosBase.p, bq,
                                       dQ.p, dQOff.p, dKB.p, par->kmerThr, T->dExt3Score, dKmerCount.p, (const uint16_t *) T->dExt3Cum, T->ext3Lo,
                                       (const uint32_t *) dPosQuery.p);
                else
                    hipLaunchKernelGGL(count_kmers7_kernel, dim3(gridFor(nPos, 4)), dim3(256), 0, ctx->stream, nPos, dPosBase.p, bq,
                                       dQ.p, dQOff.p, dKB.p, par->kmerThr, T->dExt2Score, T->dExt3Score, dKmerCount.p, (const uint16_t *) T->dExt3Cum, T->ext3Lo,
                                       (const uint32_t *) dPosQuery.p);
            }
            int rc = exclusiveScanWiden(ctx, dKmerCount.p, dKmerBase.p, nPos + 1, scanTmp);
            if (rc != SD_OK) return rc;
            int hErrCount = 0;
            if (joinCandidate) {
                SD_HIP(ctx, dQKmerBase.alloc(bq + 1));
                SD_HIP(ctx, dJoinFlag.alloc(1));
                SD_HIP(ctx, hipMemsetAsync(dJoinFlag.p, 0, sizeof(int), ctx->stream));
                hipLaunchKernelGGL(join_query_base_kernel, dim3(gridFor(bq + 1, 256)), dim3(256), 0, ctx->stream, bq, dPosBase.p,
                                   dKmerBase.p, dQKmerBase.p, dJoinFlag.p);
                SD_HIP(ctx, sdD2H(ctx, &hJoinFlag, dJoinFlag.p, sizeof(int)));
            }
            SD_HIP(ctx, sdD2H(ctx, &nKmers, dKmerBase.p + nPos, sizeof(uint64_t)));
            if (prof) SD_HIP(ctx, sdD2H(ctx, &hErrCount, dErr.p, sizeof(int)));
            SD_HIP(ctx, sdStreamSync(ctx));
            if (hErrCount == 8)
                return sdFail(ctx, SD_EUNSUPPORTED, "more than %u similar k-mers at one profile position (the reference truncates there, KmerGenerator.cpp:202-210)",
                              PROFILE_PARTIAL_CAP_BIG);
        }
        if (nKmers > KMER_BUDGET && bq > 1) {   // permissive thresholds: the k-mer list itself outgrows 32-bit device scans
            batchQ = std::max<uint32_t>(1, bq / 2);
            continue;
        }
        if (nKmers >= 0x7FFFFFFFull) return sdFail(ctx, SD_EUNSUPPORTED, "more than 2^31 similar k-mers for one query");
        hs.reset(new HostScope(ctx, "pf.emit"));
        WsView<uint32_t> dKStart(ctx, "pf.dKStart");
        WsView<uint32_t> dKLen(ctx, "pf.dKLen");
        WsView<uint32_t> dKPos(ctx, "pf.dKPos");
        WsView<uint64_t> dHitBase(ctx, "pf.dHitBase");
        WsView<uint32_t> dKStartHi(ctx, "pf.dKStartHi");   // wide indexes only: high half of the list starts
        const bool wideIdx = T->dBlockBase != nullptr;
        const bool useJoin = joinCandidate && !hJoinFlag && nKmers > 0;
        if (!useJoin) {
        SD_HIP(ctx, dKStart.alloc(nKmers + 1));
        if (wideIdx) SD_HIP(ctx, dKStartHi.alloc(nKmers + 1));
        SD_HIP(ctx, dKLen.alloc(nKmers + 1));
        SD_HIP(ctx, dKPos.alloc(nKmers + 1));
        SD_HIP(ctx, dHitBase.alloc(nKmers + 1));
        SD_HIP(ctx, hipMemsetAsync(dKLen.p, 0, (nKmers + 1) * sizeof(uint32_t), ctx->stream));
        }
        if (nKmers > 0 && !useJoin) {
            {
                ProfScope ps(ctx, "prefilter_emit_kmers");
                if (prof) {
                    hipLaunchKernelGGL(profile_kmers_kernel<true>, dim3(profGrid), dim3(64), 0, ctx->stream, nPos, dPosBase.p, bq,
                                       dQ.p, dQOff.p, dPS.p, dPI.p, par->kmerThr, T->k, dProfScratch.p, (uint32_t *) nullptr,
                                       T->dOffsets, dKmerBase.p, dKStart.p, dKLen.p, dKPos.p, dErr.p, PROFILE_PARTIAL_CAP, dProfBig.p, 0,
                                       (const uint64_t *) T->dBlockBase, wideIdx ? dKStartHi.p : (uint32_t *) nullptr, (uint64_t *) nullptr);
                    if (profAnyBig)
                        hipLaunchKernelGGL(profile_kmers_kernel<true>, dim3(PROFILE_GRID_BIG), dim3(64), 0, ctx->stream, nPos, dPosBase.p,
                                           bq, dQ.p, dQOff.p, dPS.p, dPI.p, par->kmerThr, T->k, dProfScratch.p, (uint32_t *) nullptr,
                                           T->dOffsets, dKmerBase.p, dKStart.p, dKLen.p, dKPos.p, dErr.p, PROFILE_PARTIAL_CAP_BIG,
                                           dProfBig.p, 1, (const uint64_t *) T->dBlockBase, wideIdx ? dKStartHi.p : (uint32_t *) nullptr, (uint64_t *) nullptr);
                } else if (T->k == 6) {
                    if (wideIdx)
                        hipLaunchKernelGGL(emit_kmers_kernel<true>, dim3(gridFor(nPos, 4)), dim3(256), 0, ctx->stream, nPos, dPosBase.p, bq,
                                           dQ.p, dQOff.p, dKB.p, par->kmerThr, T->dExt3Score, T->dExt3Index, T->dOffsets,
                                           dKmerBase.p, dKStart.p, dKLen.p, dKPos.p, (const uint64_t *) T->dBlockBase, dKStartHi.p,
                                           (const uint16_t *) T->dExt3Cum, T->ext3Lo, (const uint32_t *) dPosQuery.p);
                    else
                        hipLaunchKernelGGL(emit_kmers_kernel<false>, dim3(gridFor(nPos, 4)), dim3(256), 0, ctx->stream, nPos, dPosBase.p, bq,
                                           dQ.p, dQOff.p, dKB.p, par->kmerThr, T->dExt3Score, T->dExt3Index, T->dOffsets,
                                           dKmerBase.p, dKStart.p, dKLen.p, dKPos.p, (const uint64_t *) nullptr, (uint32_t *) nullptr,
                                           (const uint16_t *) T->dExt3Cum, T->ext3Lo, (const uint32_t *) dPosQuery.p);
                } else {
                    if (wideIdx)
                        hipLaunchKernelGGL(emit_kmers7_kernel<true>, dim3(gridFor(nPos, 4)), dim3(256), 0, ctx->stream, nPos, dPosBase.p, bq,
                                           dQ.p, dQOff.p, dKB.p, par->kmerThr, T->dExt2Score, T->dExt2Index, T->dExt3Score,
                                           T->dExt3Index, T->dOffsets, dKmerBase.p, dKStart.p, dKLen.p, dKPos.p,
                                           (const uint64_t *) T->dBlockBase, dKStartHi.p,
                                           (const uint16_t *) T->dExt3Cum, T->ext3Lo, (const uint32_t *) dPosQuery.p);
                    else
                        hipLaunchKernelGGL(emit_kmers7_kernel<false>, dim3(gridFor(nPos, 4)), dim3(256), 0, ctx->stream, nPos, dPosBase.p, bq,
                                           dQ.p, dQOff.p, dKB.p, par->kmerThr, T->dExt2Score, T->dExt2Index, T->dExt3Score,
                                           T->dExt3Index, T->dOffsets, dKmerBase.p, dKStart.p, dKLen.p, dKPos.p,
                                           (const uint64_t *) nullptr, (uint32_t *) nullptr,
                                           (const uint16_t *) T->dExt3Cum, T->ext3Lo, (const uint32_t *) dPosQuery.p);
                }
            }
            int rc = exclusiveScanWiden(ctx, dKLen.p, dHitBase.p, nKmers + 1, scanTmp);
            if (rc != SD_OK) return rc;
            SD_HIP(ctx, sdD2H(ctx, &nHits, dHitBase.p + nKmers, sizeof(uint64_t)));
            SD_HIP(ctx, sdStreamSync(ctx));
        }
        hs.reset(new HostScope(ctx, "pf.stats"));
        std::vector<uint64_t> hStats((size_t) bq * 4, 0);
        WsView<uint64_t> dStats(ctx, "pf.dStats");
        SD_HIP(ctx, dStats.alloc((size_t) bq * 4));
        SD_HIP(ctx, hipMemsetAsync(dStats.p, 0, (size_t) bq * 4 * sizeof(uint64_t), ctx->stream));
        WsView<uint32_t> dQSplit(ctx, "pf.dQSplit");
        WsView<int> dSplitFlag(ctx, "pf.dSplitFlag");
        WsView<uint32_t> dQParts(ctx, "pf.dQParts");     // overflows of the reference's hit buffer per query, and where (all of them)
        WsView<uint32_t> dQSplits(ctx, "pf.dQSplits");
        SD_HIP(ctx, dQSplit.alloc(bq));
        SD_HIP(ctx, dQParts.alloc(bq));
        SD_HIP(ctx, dQSplits.alloc((size_t) bq * PF_SPLITS_MAX));
        SD_HIP(ctx, dSplitFlag.alloc(1));
        SD_HIP(ctx, hipMemsetAsync(dSplitFlag.p, 0, sizeof(int), ctx->stream));
        std::vector<uint8_t> hUnsupported;
        // join path: the k-mer stream, its k-mer-sorted copy, the hits grouped by query group
        WsView<uint64_t> dElems(ctx, "pf.dElems");
        WsView<uint64_t> dSorted(ctx, "pf.dSorted");
        WsView<uint2> dHitsKV(ctx, "pf.dHitsKV");
        WsView<uint8_t> dHitR6(ctx, "pf.dHitR6");   // join path: a byte per hit for the coarse split's count pass (nullptr: not written)
        WsView<uint64_t> dQHitBase(ctx, "pf.dQHitBase");   // hits of query q start at dQHitBase[q] (either path)
        WsView<uint64_t> dVQHitBase(ctx, "pf.dVQHitBase");  // with a split into target ranges: hits of (query, range) start here
        int jcBits = 0;   // join path: the scatter wrote (query, target range) sub-segments of 2^jcBits ranges per query
        if (useJoin) {
            WsView<uint32_t> dKpCounts(ctx, "pf.dKpCounts");
            WsView<uint32_t> dKpTotal(ctx, "pf.dKpTotal");
            WsView<uint64_t> dKpBase(ctx, "pf.dKpBase");
            WsView<uint32_t> dJqCounts(ctx, "pf.dJqCounts");
            WsView<uint32_t> dQHits(ctx, "pf.dQHits");
            WsView<unsigned long long> dWgTotal(ctx, "pf.dWgTotal");
            WsView<uint32_t> dQEff(ctx, "pf.dQEff");
            const uint64_t nSortedCap = nKmers + (uint64_t) KP_BINS * JC;   // every k-mer range padded to whole join chunks
            WsView<uint16_t> dChunkBin(ctx, "pf.dChunkBin");
            SD_HIP(ctx, dElems.alloc(nKmers + 1));
            SD_HIP(ctx, dSorted.alloc(nSortedCap));
            SD_HIP(ctx, dChunkBin.alloc(nSortedCap / JC + 1));
            SD_HIP(ctx, dKpCounts.alloc((size_t) JP_WGS * KP_BINS));
            SD_HIP(ctx, dKpTotal.alloc(KP_BINS));
            SD_HIP(ctx, dKpBase.alloc(KP_BINS + 1));
            // (SD_JJ_WGS: fewer persistent workgroups of the join -- a smaller working set of index entries per XCD, for A/Bs)
            const int jjWgs = getenv("SD_JJ_WGS") ? std::max(8, std::min(JJ_WGS, atoi(getenv("SD_JJ_WGS")) / 8 * 8)) : JJ_WGS;
            SD_HIP(ctx, dJqCounts.alloc((size_t) JJ_WGS * bq));
            SD_HIP(ctx, dQHits.alloc(bq));
            SD_HIP(ctx, dWgTotal.alloc(JJ_WGS));
            {
                ProfScope ps(ctx, "prefilter_emit_kmers");
                if (prof) {   // the per-position products again, this time writing the stream (the tiers the count pass chose)
                    hipLaunchKernelGGL(profile_kmers_kernel<true>, dim3(profGrid), dim3(64), 0, ctx->stream, nPos, dPosBase.p, bq, dQ.p, dQOff.p, dPS.p,
                                       dPI.p, par->kmerThr, T->k, dProfScratch.p, (uint32_t *) nullptr, (const uint32_t *) nullptr, dKmerBase.p,
                                       (uint32_t *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr, dErr.p, PROFILE_PARTIAL_CAP, dProfBig.p, 0,
                                       (const uint64_t *) nullptr, (uint32_t *) nullptr, dElems.p);
                    if (profAnyBig)
                        hipLaunchKernelGGL(profile_kmers_kernel<true>, dim3(PROFILE_GRID_BIG), dim3(64), 0, ctx->stream, nPos, dPosBase.p, bq, dQ.p,
                                           dQOff.p, dPS.p, dPI.p, par->kmerThr, T->k, dProfScratch.p, (uint32_t *) nullptr, (const uint32_t *) nullptr,
                                           dKmerBase.p, (uint32_t *) nullptr, (uint32_t *) nullptr, (uint32_t *) nullptr, dErr.p,
                                           PROFILE_PARTIAL_CAP_BIG, dProfBig.p, 1, (const uint64_t *) nullptr, (uint32_t *) nullptr, dElems.p);
                } else
                hipLaunchKernelGGL(emit_kmers_join_kernel, dim3(gridFor(nPos, 4)), dim3(256), 0, ctx->stream, nPos, dPosBase.p, bq, dQ.p,
                                   dQOff.p, dKB.p, par->kmerThr, T->dExt3Score, T->dExt3Index, dKmerBase.p, dElems.p, (const uint16_t *) T->dExt3Cum, T->ext3Lo,
                                   (const uint32_t *) dPosQuery.p);
            }
            {
                ProfScope ps(ctx, "prefilter_kmer_partition");
                hipLaunchKernelGGL(kp_hist_kernel, dim3(JP_WGS), dim3(JP_NT), 0, ctx->stream, (const uint64_t *) dElems.p, nKmers, dKpCounts.p);
                hipLaunchKernelGGL(col_prefix_kernel, dim3(KP_BINS / 64), dim3(256), 0, ctx->stream, dKpCounts.p, JP_WGS, KP_BINS, dKpTotal.p);
                hipLaunchKernelGGL(small_scan_kernel, dim3(1), dim3(SS_NT), 0, ctx->stream, (const uint32_t *) dKpTotal.p, KP_BINS, dKpBase.p, (uint32_t) JC);
                hipLaunchKernelGGL(kp_scatter_kernel, dim3(JP_WGS), dim3(JP_NT), 0, ctx->stream, (const uint64_t *) dElems.p, nKmers,
                                   (const uint32_t *) dKpCounts.p, (const uint64_t *) dKpBase.p, (const uint32_t *) dQKmerBase.p, bq, dSorted.p);
                hipLaunchKernelGGL(kp_finish_kernel, dim3(KP_BINS), dim3(256), 0, ctx->stream, (const uint32_t *) dKpTotal.p,
                                   (const uint64_t *) dKpBase.p, dSorted.p, dChunkBin.p);
            }
            // (the join kernels read the padded length of the sorted stream where the partition left it: no round trip to the host)
            const uint64_t *nSortedPtr = dKpBase.p + KP_BINS;
            {
                ProfScope ps(ctx, "prefilter_join_count");
                hipLaunchKernelGGL(join_count_kernel, dim3(jjWgs), dim3(JJ_NT), 0, ctx->stream, (const uint64_t *) dSorted.p, nSortedPtr,
                                   (const uint16_t *) dChunkBin.p, (const uint32_t *) T->dOffsets, dJqCounts.p, (int) bq, dWgTotal.p);
                hipLaunchKernelGGL(col_prefix_kernel, dim3(gridFor(bq, 64)), dim3(256), 0, ctx->stream, dJqCounts.p, jjWgs, (int) bq, dQHits.p);
            }
            // overflow of the reference's hit buffer, in k-mer ordinals
            hipLaunchKernelGGL(query_splits_join_kernel, dim3(bq), dim3(256), 0, ctx->stream, bq, (const uint32_t *) dQKmerBase.p,
                               (const uint64_t *) dElems.p, (const uint32_t *) T->dOffsets, (const uint32_t *) dQHits.p, maxDbMatches,
                               dQSplit.p, dQParts.p, dQSplits.p, dSplitFlag.p);
            hipLaunchKernelGGL(join_stats_kernel, dim3(gridFor(bq, 256)), dim3(256), 0, ctx->stream, bq, (const uint32_t *) dQKmerBase.p,
                               (const uint32_t *) dQHits.p, dStats.p);
            std::vector<unsigned long long> hWg((size_t) jjWgs);
            std::vector<uint32_t> hQHits(bq);
            int hSplitFlagJ = 0;
            SD_HIP(ctx, sdD2H(ctx, hWg.data(), dWgTotal.p, (size_t) jjWgs * sizeof(unsigned long long)));
            SD_HIP(ctx, sdD2H(ctx, hQHits.data(), dQHits.p, bq * sizeof(uint32_t)));
            SD_HIP(ctx, sdD2H(ctx, &hSplitFlagJ, dSplitFlag.p, sizeof(int)));
            SD_HIP(ctx, sdStreamSync(ctx));
            uint64_t all = 0;
            for (unsigned long long v : hWg) all += v;
            if (all > HIT_BUDGET && bq > 1) {   // too many hits for one sub-batch: halve it and retry
                batchQ = std::max<uint32_t>(1, bq / 2);
                continue;
            }
            if (all >= 0xFFFFFFF0ull) {   // the per-query counters are 32 bit: this query goes the lookup path
                forceLookup = true;
                continue;
            }
            for (uint32_t x = 0; x < bq; x++) hStats[(size_t) x * 4 + 1] = hQHits[x];
            nHits = all;
            if (hSplitFlagJ) {
                // queries the device does not compute (more than PF_SPLITS_MAX overflows of the reference's hit buffer, or a single
                // index list as large as that buffer) are taken out of the batch and reported per query (outCount = UINT32_MAX); the
                // join leaves their lists out
                std::vector<uint32_t> hSplit(bq);
                SD_HIP(ctx, sdD2H(ctx, hSplit.data(), dQSplit.p, (size_t) bq * sizeof(uint32_t)));
                SD_HIP(ctx, sdStreamSync(ctx));
                hUnsupported.assign(bq, 0);
                uint32_t nBad = 0, firstBad = 0;
                for (uint32_t x = 0; x < bq; x++)
                    if (hSplit[x] == QUERY_UNSUPPORTED) {
                        hUnsupported[x] = 1;
                        nHits -= hQHits[x];
                        if (!nBad) firstBad = qBeg + x;
                        nBad++;
                    }
                sdFail(ctx, SD_EUNSUPPORTED, "%u quer%s of the batch [%u, %u) (first: %u) overflow the reference's hit buffer more than 32 times or hold "
                       "an index list as large as that buffer: reported with outCount = UINT32_MAX, the rest of the batch is computed",
                       nBad, nBad == 1 ? "y" : "ies", qBeg, qBeg + bq, firstBad);
            }
            // per-query segments of the hit array (the layout the bucket machinery takes), then the scatter pass of the join
            SD_HIP(ctx, dQEff.alloc(bq));
            SD_HIP(ctx, dQHitBase.alloc(bq + 1));
            hipLaunchKernelGGL(join_effective_totals_kernel, dim3(gridFor(bq, 256)), dim3(256), 0, ctx->stream, bq, (const uint32_t *) dQHits.p,
                               (const uint32_t *) dQSplit.p, dQEff.p);
            hipLaunchKernelGGL(small_scan_kernel, dim3(1), dim3(SS_NT), 0, ctx->stream, (const uint32_t *) dQEff.p, (int) bq, dQHitBase.p, 0u);
            // SD_JOIN_RANGES=1 (an experiment kept for the A/B, off by default): the scatter splits every query's hits into target
            // ranges right away (join_scatter_kernel<RANGES>) by the rule of the coarse split below, which it replaces on the join path.
            // Measured at 1 000 proteomes, isolated, per 8 192 queries: plain scatter 72.8 + coarse split 97.9 = 171 ms against ranged
            // count 90 + ranged scatter 150 - 197 ms, 2 379 vs 2 235 genome-pairs/s end to end -- a wavefront's 256 hits fall into ~40
            // (query, range) columns, i.e. 48-byte pieces, where the plain scatter writes 600-byte runs and the coarse split moves
            // 16 384-hit tiles into at most 64 ranges (2-KB runs): two streaming levels beat one fine-grained scatter.
            if (nHits > 0 && getenv("SD_JOIN_RANGES") && atoi(getenv("SD_JOIN_RANGES")) == 1) {
                const uint64_t avgQ = nHits / std::max<uint32_t>(bq, 1);
                const uint64_t perVQ = getenv("SD_PF_COARSE") ? (uint64_t) atoll(getenv("SD_PF_COARSE")) : 200000;
                const uint64_t perVQsplit = getenv("SD_PF_COARSE") ? perVQ : 100000;
                if (avgQ > perVQ)
                    while (jcBits < CP_MAX_BITS && tBits - (jcBits + 1) >= 8 && (avgQ >> jcBits) > perVQsplit) jcBits++;
                while (jcBits > 0 && ((uint64_t) bq << jcBits) > (uint64_t) JX_COLS_MAX) jcBits--;   // the columns' cursors live in LDS
            }
            if (nHits > 0 && jcBits > 0) {
                const int cols = (int) (bq << jcBits);
                WsView<uint32_t> dJrCounts(ctx, "pf.dJrCounts");
                WsView<uint32_t> dVQHits(ctx, "pf.dVQHits");
                SD_HIP(ctx, dJrCounts.alloc((size_t) JJ_WGS * cols));
                SD_HIP(ctx, dVQHits.alloc((size_t) cols + 1));
                SD_HIP(ctx, dVQHitBase.alloc((size_t) cols + 1));
                SD_HIP(ctx, dHitsKV.alloc(nHits));
                {
                    ProfScope ps(ctx, "prefilter_join_count");
                    hipLaunchKernelGGL((join_scatter_kernel<false, true, true>), dim3(jjWgs), dim3(JJ_NT), 0, ctx->stream, (const uint64_t *) dSorted.p,
                                       nSortedPtr, (const uint16_t *) dChunkBin.p, (const uint32_t *) T->dOffsets, (const uint2 *) T->dEntries, bq,
                                       (const uint32_t *) nullptr, cols, (const uint64_t *) nullptr, tBits, (const uint32_t *) dQSplit.p,
                                       (uint2 *) nullptr, jcBits, dJrCounts.p, (uint8_t *) nullptr, 0);
                    hipLaunchKernelGGL(col_prefix_kernel, dim3(gridFor((uint64_t) cols, 64)), dim3(256), 0, ctx->stream, dJrCounts.p, jjWgs, cols, dVQHits.p);
                    SD_HIP(ctx, hipMemsetAsync(dVQHits.p + cols, 0, sizeof(uint32_t), ctx->stream));
                    int rcV = exclusiveScanWiden(ctx, dVQHits.p, dVQHitBase.p, (uint64_t) cols + 1, scanTmp);
                    if (rcV != SD_OK) return rcV;
                }
                ProfScope ps(ctx, "prefilter_join_scatter");
                static const bool ntStore = !(getenv("SD_JOIN_NT") && atoi(getenv("SD_JOIN_NT")) == 0);
                auto kern = ntStore ? join_scatter_kernel<true, true, false> : join_scatter_kernel<false, true, false>;
                hipLaunchKernelGGL(kern, dim3(jjWgs), dim3(JJ_NT), 0, ctx->stream, (const uint64_t *) dSorted.p, nSortedPtr,
                                   (const uint16_t *) dChunkBin.p, (const uint32_t *) T->dOffsets, (const uint2 *) T->dEntries, bq,
                                   (const uint32_t *) dJrCounts.p, cols, (const uint64_t *) dVQHitBase.p, tBits,
                                   (const uint32_t *) dQSplit.p, dHitsKV.p, jcBits, (uint32_t *) nullptr, (uint8_t *) nullptr, 0);
            } else if (nHits > 0) {
                SD_HIP(ctx, dHitsKV.alloc(nHits));
                // the coarse split's count pass needs a hit's range and nothing else: one byte per hit beside the 8-byte stream (the top
                // six target bits: any split of up to 2^6 ranges reads its range off them) -- 1.5 MB per query written here for 12.3 MB
                // per query not read there (SD_PF_R6=0: the count pass reads the hits)
                if (tBits >= CP_MAX_BITS + 8 && coarseBitsFor(nHits / std::max<uint32_t>(bq, 1), tBits) > 0 &&
                    !(getenv("SD_PF_R6") && atoi(getenv("SD_PF_R6")) == 0))
                    SD_HIP(ctx, dHitR6.alloc(nHits + 8));
                ProfScope ps(ctx, "prefilter_join_scatter");
                // SD_JOIN_NT=0: plain stores (partial lines of neighbouring runs merge in the XCD's L2 before they are written back)
                static const bool ntStore = !(getenv("SD_JOIN_NT") && atoi(getenv("SD_JOIN_NT")) == 0);
                auto kern = ntStore ? join_scatter_kernel<true, false, false> : join_scatter_kernel<false, false, false>;
                hipLaunchKernelGGL(kern, dim3(jjWgs), dim3(JJ_NT), 0, ctx->stream, (const uint64_t *) dSorted.p, nSortedPtr,
                                   (const uint16_t *) dChunkBin.p, (const uint32_t *) T->dOffsets, (const uint2 *) T->dEntries, bq,
                                   (const uint32_t *) dJqCounts.p, (int) bq, (const uint64_t *) dQHitBase.p, tBits,
                                   (const uint32_t *) dQSplit.p, dHitsKV.p, 0, (uint32_t *) nullptr, dHitR6.p, tBits - CP_MAX_BITS);
            }
            SD_HIP(ctx, hipGetLastError());
        }
        if (!useJoin) {
        if (nHits > std::min(HIT_BUDGET, LOOKUP_HIT_BUDGET) && bq > 1) {   // too many hits for one sort: halve the sub-batch and retry
            batchQ = std::max<uint32_t>(1, bq / 2);
            continue;
        }
        if (nHits >= 0xFFFFFFFFull) return sdFail(ctx, SD_EUNSUPPORTED, "more than 2^32 hits in one query batch");
        hipLaunchKernelGGL(stats_kernel, dim3(gridFor(bq, 256)), dim3(256), 0, ctx->stream, bq, dPosBase.p, dKmerBase.p,
                           dHitBase.p, nKmers, nHits, dStats.p);
        // Stream positions: 24 bits of the value word beside the diagonal byte -- or, for sub-batches with a query of 2^24 hits
        // and more, the whole word, the diagonal byte travelling in the key (bucket path only: it needs the coarse split)
        const uint64_t posLimit = useBuckets ? 0xFFFFFFF0ull : (1ull << 24);
        // the reference's hit buffer holds maxDbMatches entries per query; where a query overflows it once, the match runs
        // on the two parts separately (query_split_kernel); two overflows are not implemented
        hipLaunchKernelGGL(query_split_kernel, dim3(gridFor(bq, 256)), dim3(256), 0, ctx->stream, bq, dPosBase.p, dKmerBase.p,
                           dHitBase.p, maxDbMatches, dQSplit.p, dQParts.p, dQSplits.p, dSplitFlag.p, posLimit);
        int hSplitFlag = 0;
        SD_HIP(ctx, sdD2H(ctx, &hSplitFlag, dSplitFlag.p, sizeof(int)));
        SD_HIP(ctx, sdD2H(ctx, hStats.data(), dStats.p, (size_t) bq * 4 * sizeof(uint64_t)));
        SD_HIP(ctx, sdStreamSync(ctx));
        if (hSplitFlag) {
            // Queries with >= 2^32 index hits (or more than PF_SPLITS_MAX overflows of the reference's hit buffer, or an index list as
            // large as that buffer) cannot be computed here.  They are taken out of the batch -- their index lists emptied,
            // offsets re-scanned -- and reported per query (outCount = UINT32_MAX); every other query is computed as usual.
            std::vector<uint32_t> hSplit(bq);
            SD_HIP(ctx, sdD2H(ctx, hSplit.data(), dQSplit.p, (size_t) bq * sizeof(uint32_t)));
            SD_HIP(ctx, sdStreamSync(ctx));
            hUnsupported.assign(bq, 0);
            uint32_t nBad = 0, firstBad = 0;
            for (uint32_t x = 0; x < bq; x++)
                if (hSplit[x] == QUERY_UNSUPPORTED) {
                    hUnsupported[x] = 1;
                    if (!nBad) firstBad = qBeg + x;
                    nBad++;
                }
            sdFail(ctx, SD_EUNSUPPORTED, "%u quer%s of the batch [%u, %u) (first: %u) have >= 2^32 index hits or overflow the reference's hit buffer "
                   "more than 32 times: reported with outCount = UINT32_MAX, the rest of the batch is computed",
                   nBad, nBad == 1 ? "y" : "ies", qBeg, qBeg + bq, firstBad);
            hipLaunchKernelGGL(drop_query_kmers_kernel, dim3(gridFor(nKmers, 256)), dim3(256), 0, ctx->stream, nKmers, dKPos.p, dQSplit.p, dKLen.p);
            int rc2 = exclusiveScanWiden(ctx, dKLen.p, dHitBase.p, nKmers + 1, scanTmp);
            if (rc2 != SD_OK) return rc2;
            SD_HIP(ctx, sdD2H(ctx, &nHits, dHitBase.p + nKmers, sizeof(uint64_t)));
            SD_HIP(ctx, hipMemsetAsync(dSplitFlag.p, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(query_split_kernel, dim3(gridFor(bq, 256)), dim3(256), 0, ctx->stream, bq, dPosBase.p, dKmerBase.p,
                               dHitBase.p, maxDbMatches, dQSplit.p, dQParts.p, dQSplits.p, dSplitFlag.p, posLimit);
            SD_HIP(ctx, sdStreamSync(ctx));
        }
        }   // lookup path

        hs.reset(new HostScope(ctx, "pf.gather_sort_match"));
        uint32_t nCand = 0;
        const uint64_t *nKeptPtr = nullptr;   // kept candidates (device); null: none
        bool bucketDone = false;
        bool widePos = false;
        if (useBuckets && !useJoin)
            for (uint32_t x = 0; x < bq && !widePos; x++)
                widePos = hStats[(size_t) x * 4 + 1] >= (1ull << 24) && !(x < hUnsupported.size() && hUnsupported[x]);
        const uint32_t posMask = widePos ? 0xFFFFFFFFu : 0xFFFFFFu;
        WsView<uint32_t> dKeyA(ctx, "pf.dKeyA");
        WsView<uint32_t> dKeyB(ctx, "pf.dKeyB");
        WsView<uint32_t> dValA(ctx, "pf.dValA");
        WsView<uint32_t> dValB(ctx, "pf.dValB");
        WsView<uint16_t> dDiag(ctx, "pf.dDiag");
        if (!useJoin) SD_HIP(ctx, dQHitBase.alloc(bq + 1));
        WsView<uint8_t> dEmit(ctx, "pf.dEmit");
        WsView<uint64_t> dEmitPos(ctx, "pf.dEmitPos");
        WsView<uint64_t> dEmit64(ctx, "pf.dEmit64");
        WsView<uint32_t> dCKey(ctx, "pf.dCKey");
        WsView<uint32_t> dCVal(ctx, "pf.dCVal");
        WsView<uint32_t> dCLen(ctx, "pf.dCLen");
        WsView<uint32_t> dKKey(ctx, "pf.dKKey");
        WsView<uint32_t> dKVal(ctx, "pf.dKVal");
        WsView<uint32_t> dQStart(ctx, "pf.dQStart");
        WsView<int32_t> dCScore(ctx, "pf.dCScore");
        WsView<int32_t> dKScore(ctx, "pf.dKScore");
        WsView<uint8_t> dKeep(ctx, "pf.dKeep");
        if (nHits > 0) {
            if (!useJoin) {
            SD_HIP(ctx, dKeyA.alloc(nHits));
            SD_HIP(ctx, dValA.alloc(nHits));
            SD_HIP(ctx, dKeyB.alloc(nHits));
            SD_HIP(ctx, dValB.alloc(nHits));
            SD_HIP(ctx, dDiag.alloc(nHits));
            hipLaunchKernelGGL(query_hit_base_kernel, dim3(gridFor(bq + 1, 256)), dim3(256), 0, ctx->stream, bq, dPosBase.p,
                               dKmerBase.p, dHitBase.p, dQHitBase.p);
            }
            if (!useJoin) {
                ProfScope ps(ctx, "prefilter_gather_hits");
                hipLaunchKernelGGL(gather_hits_kernel, dim3(gridFor(nKmers, 256)), dim3(256), 0, ctx->stream, nKmers, dKStart.p,
                                   dKLen.p, dKPos.p, dHitBase.p, dPosBase.p, bq, (const uint2 *) T->dEntries, tBits, dQHitBase.p,
                                   dKeyA.p, dValA.p, dDiag.p, wideIdx ? (const uint32_t *) dKStartHi.p : (const uint32_t *) nullptr, widePos ? 1 : 0);
            }
            // ---- double-diagonal match: bucketed LDS path, or (fallback / SD_PF_SORT=1) global radix sort + match
            if (useBuckets) {
                // very hit-rich queries (large target sets): coarse split into virtual queries first (see coarse_*_kernel)
                uint32_t nVQ = bq;
                const uint32_t nVQ0 = nVQ;
                const int tBits0 = tBits;
                int cBits = 0, tBitsV = tBits0;
                const uint64_t *pHitBase = dQHitBase.p;
                const uint32_t *pKey = dKeyA.p, *pVal = dValA.p;
                // join path: the hits arrive interleaved (key, value)
                const uint2 *pKV = useJoin ? (const uint2 *) dHitsKV.p : (const uint2 *) nullptr;
                WsView<uint32_t> dSegBase(ctx, "pf.dSegBase");
                WsView<uint32_t> dSegCount(ctx, "pf.dSegCount");
                WsView<uint2> dKVC(ctx, "pf.dKVC");
                if (useJoin && jcBits > 0) {   // the join wrote (query, target range) sub-segments: they are the virtual queries
                    cBits = jcBits;
                    nVQ = nVQ0 << cBits;
                    tBitsV = tBits0 - cBits;
                    pHitBase = dVQHitBase.p;
                } else {
                    // decided by the average query of the sub-batch (a few long queries are what the adaptive bucket count and
                    // the oversize-bucket launch are for): ~100 hits per bucket at 2^11 buckets
                    const uint64_t avgQ = nHits / std::max<uint32_t>(nVQ0, 1);
                    // no split up to 2 * 10^5 hits per average query (the filter takes such a segment in one round); beyond that, ranges of
                    // at most 10^5 hits: the filter's Bloom rounds, the partition and the bucket sort all run on half-size segments for one more
                    // level of the (cheap, streaming) split -- isolated prefilter at 1 000 proteomes 443 -> 390 ms per 8 192 queries, +4 % end
                    // to end; at 100 proteomes (1.5 * 10^5 hits per query) a split would only add its 24 B per hit (1 915 -> 1 767)
                    // (coarseBitsFor below is this rule; the join's scatter asks it whether a split will follow)
                    const uint64_t perVQ = getenv("SD_PF_COARSE") ? (uint64_t) atoll(getenv("SD_PF_COARSE")) : 200000;
                    // Round 6: ranges of at most 5 * 10^4 hits (SD_PF_COARSE_SPLIT; 10^5 until then) -- 32 ranges at 1 000 proteomes.  The filter's
                    // Bloom filter then has twice the bits per key: 7.2 % of the hit stream left instead of 12.3 % (4.7 % at 2.5 * 10^4), and what
                    // is left comes in segments the LDS sorter takes whole (segment_match) instead of through partition_hits / bucket_match.
                    // Isolated, per 8 192 queries, interleaved in one process (profiles/r06y_split.txt): 328 ms at 10^5 (coarse split 69, filter
                    // 74, partition + bucket + segment match 59), **315 ms at 5 * 10^4** (80 / 67 / 43), 323 ms at 2.5 * 10^4 (92 / 66 / 37: the
                    // split's cost grows with its ranges).  Round 5 had measured finer ranges as a loss (388 - 442 vs 385 ms): the filter read
                    // two arrays then and the split wrote them.  (Two more bitmaps in the filter -- a target must be hit twice to be nominated --
                    // were tried on top and removed: 7.4 % left instead of 7.2 %, the kernel 8 ms slower; what passes the filter at this
                    // granularity are targets that do have two hits on one diagonal byte, profiles/r06x_hf_multi.txt.)
                    const uint64_t perVQsplit = getenv("SD_PF_COARSE") ? perVQ : (getenv("SD_PF_COARSE_SPLIT") ? (uint64_t) atoll(getenv("SD_PF_COARSE_SPLIT")) : 50000);
                    (void) perVQsplit;
                    cBits = coarseBitsFor(avgQ, tBits0);
                    // wide stream positions need the split: it is where the diagonal byte moves into the key (8 free bits)
                    while (widePos && cBits < CP_MAX_BITS && tBits - (cBits + 1) >= 8 && (cBits < 1 || tBits - cBits > 24)) cBits++;
                    if (widePos && (cBits < 1 || tBits - cBits > 24))
                        return sdFail(ctx, SD_EUNSUPPORTED, "a query with >= 2^24 index hits against a target set of %u sequences", T->nSeq);
                    if (cBits > 0) {
                        // segments of CP_SEG hits per query.  The join path knows every query's hits on the host already (the read that
                        // sized the sub-batch), the lookup path reads the segment starts back; the segment table goes up through a pinned
                        // buffer, so nothing here waits for the stream a second time
                        std::vector<uint64_t> hQHB(nVQ0 + 1, 0);
                        if (useJoin) {
                            for (uint32_t x = 0; x < nVQ0; x++)
                                hQHB[x + 1] = hQHB[x] + ((x < hUnsupported.size() && hUnsupported[x]) ? 0ull : hStats[(size_t) x * 4 + 1]);
                        } else {
                            SD_HIP(ctx, sdD2H(ctx, hQHB.data(), pHitBase, (nVQ0 + 1) * sizeof(uint64_t)));
                            SD_HIP(ctx, sdStreamSync(ctx));
                        }
                        ProfScope ps(ctx, "prefilter_coarse_split");
                        const uint32_t C = 1u << cBits;
                        uint32_t *hSegBase = nullptr;
                        SD_HIP(ctx, pinGet(ctx, "pf.hSegBase", (size_t) nVQ0 + 1, &hSegBase));
                        hSegBase[0] = 0;
                        for (uint32_t x = 0; x < nVQ0; x++)
                            hSegBase[x + 1] = hSegBase[x] + (uint32_t) ((hQHB[x + 1] - hQHB[x] + CP_SEG - 1) / CP_SEG);
                        const uint32_t nSeg = hSegBase[nVQ0];
                        nVQ = nVQ0 * C;
                        tBitsV = tBits0 - cBits;
                        SD_HIP(ctx, dVQHitBase.alloc((size_t) nVQ + 1));
                        SD_HIP(ctx, dSegBase.alloc(nVQ0 + 1));
                        SD_HIP(ctx, dSegCount.alloc((size_t) std::max<uint32_t>(nSeg, 1) * C));
                        SD_HIP(ctx, dKVC.alloc(nHits));
                        SD_HIP(ctx, hipMemcpyAsync(dSegBase.p, hSegBase, (nVQ0 + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                        if (nSeg > 0)
                            hipLaunchKernelGGL(coarse_count_kernel, dim3(nSeg), dim3(256), 0, ctx->stream, nVQ0, pHitBase, dSegBase.p, tBits0,
                                               cBits, dKeyA.p, pKV, dSegCount.p, useJoin ? (const uint8_t *) dHitR6.p : (const uint8_t *) nullptr);
                        hipLaunchKernelGGL(coarse_offsets_kernel, dim3(nVQ0), dim3(256), 0, ctx->stream, nVQ0, pHitBase, dSegBase.p, cBits,
                                           dSegCount.p, dVQHitBase.p);
                        if (nSeg > 0) {   // SD_CS_PER=8: steps of 2 048 hits (16 KB of staging per workgroup instead of 36)
                            static const bool per8 = getenv("SD_CS_PER") && atoi(getenv("SD_CS_PER")) == 8;
                            auto csKern = per8 ? coarse_scatter_kernel<8> : coarse_scatter_kernel<16>;
                            hipLaunchKernelGGL(csKern, dim3(nSeg), dim3(256), 0, ctx->stream, nVQ0, pHitBase, dSegBase.p, tBits0,
                                               cBits, dSegCount.p, dKeyA.p, dValA.p, pKV, dKVC.p,
                                               widePos ? (const uint16_t *) dDiag.p : (const uint16_t *) nullptr);
                        }
                        // (hSegBase is pinned and persistent: the upload may still be reading it; the next sub-batch writes it only after
                        // several waits for this stream)
                        pHitBase = dVQHitBase.p;
                        pKey = nullptr;   // the ranges' hits are (key, value) pairs, like the join's stream
                        pVal = nullptr;
                        pKV = dKVC.p;
                    }
                }
                // hot-target filter (hot_filter_kernel): every (virtual) query's segment is compacted in place to the hits of
                // targets that can still emit a candidate; the bucket machinery below runs on what is left, laid out densely
                uint64_t nLeft = nHits;
                WsView<uint32_t> dHotCount(ctx, "pf.dHotCount");
                WsView<uint64_t> dHotBase(ctx, "pf.dHotBase");
                const uint32_t *pSegCount = nullptr;
                const uint64_t *pOutBase = nullptr;
                const bool useFilter = !(getenv("SD_PF_FILTER") && atoi(getenv("SD_PF_FILTER")) == 0);
                if (useFilter) {
                    const uint32_t minSeg = getenv("SD_PF_FILTER_MIN") ? (uint32_t) atoi(getenv("SD_PF_FILTER_MIN")) : 256u;
                    SD_HIP(ctx, dHotCount.alloc((size_t) nVQ + 1));
                    SD_HIP(ctx, dHotBase.alloc((size_t) nVQ + 1));
                    SD_HIP(ctx, hipMemsetAsync(dHotCount.p + nVQ, 0, sizeof(uint32_t), ctx->stream));
                    {
                        ProfScope ps(ctx, "prefilter_hot_filter");
                        const int geo = getenv("SD_PF_HF") ? atoi(getenv("SD_PF_HF")) : 0;
                        // One workgroup per CU that loops over the segments (SD_PF_HF_PERSIST=n: n per CU, 0: one workgroup per segment).  A
                        // 128-KB workgroup needs a drained CU; beside the score wavefronts of the other streams every new workgroup waited for
                        // one again, the persistent one keeps the CU it has waited for.  Round 5, interleaved runs on one box: the kernel's
                        // time inside the pipeline 7 - 11 s -> 2.7 s per 14 steps at 1 000 proteomes, 2 417 -> 2 444 genome-pairs/s (three runs
                        // each), 1 905 -> 1 929 at 100 proteomes (round 4 had measured the same kernel time and no gain in throughput)
                        const int hfPersist = getenv("SD_PF_HF_PERSIST") ? atoi(getenv("SD_PF_HF_PERSIST")) : 1;
                        const uint32_t hfGrid = hfPersist > 0 ? std::min<uint32_t>(nVQ, (uint32_t) hfPersist * (uint32_t) ctx->prop.multiProcessorCount) : nVQ;
#define SD_HF(NT_, BW_, HL_)                                                                                                              \
    hipLaunchKernelGGL((hot_filter_kernel<NT_, BW_, HL_>), dim3(hfGrid), dim3(NT_), 0, ctx->stream, nVQ, pHitBase, tBitsV, (uint32_t *) pKey, \
                       (uint32_t *) pVal, (uint2 *) pKV, widePos ? tBitsV : 0, minSeg, dHotCount.p)
                        if (geo == 1) SD_HF(512, 12288, 17);        // 64 KB
                        else if (geo == 2) SD_HF(512, 8192, 17);    // 48 KB
                        else if (geo == 3) SD_HF(256, 6144, 16);    // 32 KB
                        else if (geo == 4) SD_HF(1024, 16384, 18);  // 96 KB
                        else if (geo == 5) SD_HF(1024, 24576, 18);  // 128 KB, pass-B tiles of 4 096 hits
                        else if (geo == 6) hipLaunchKernelGGL((hot_filter_kernel<1024, 12288, 18, 8>), dim3(hfGrid), dim3(1024), 0, ctx->stream, nVQ, pHitBase, tBitsV,
                                                              (uint32_t *) pKey, (uint32_t *) pVal, (uint2 *) pKV, widePos ? tBitsV : 0, minSeg, dHotCount.p);   // 48 + 32 KB: two workgroups per CU
                        else if (geo == 8) hipLaunchKernelGGL((hot_filter_kernel<1024, 12288, 17, 8>), dim3(hfGrid), dim3(1024), 0, ctx->stream, nVQ, pHitBase, tBitsV,
                                                              (uint32_t *) pKey, (uint32_t *) pVal, (uint2 *) pKV, widePos ? tBitsV : 0, minSeg, dHotCount.p);   // 48 + 16 KB: two workgroups of 1 024 threads per CU (with SD_PF_HF_PERSIST=2)
                        else if (geo == 7) hipLaunchKernelGGL((hot_filter_kernel<1024, 16384, 17, 8>), dim3(hfGrid), dim3(1024), 0, ctx->stream, nVQ, pHitBase, tBitsV,
                                                              (uint32_t *) pKey, (uint32_t *) pVal, (uint2 *) pKV, widePos ? tBitsV : 0, minSeg, dHotCount.p);   // 64 + 16 KB: two workgroups per CU
                        else hipLaunchKernelGGL((hot_filter_kernel<1024, 24576, 18, 8>), dim3(hfGrid), dim3(1024), 0, ctx->stream, nVQ, pHitBase, tBitsV,
                                                (uint32_t *) pKey, (uint32_t *) pVal, (uint2 *) pKV, widePos ? tBitsV : 0, minSeg, dHotCount.p);   // 128 KB, tiles of 8 192: half the barriers (isolated 62.9 -> 60.5 ms per step)
#undef SD_HF
                    }
                    int rcF = exclusiveScanWiden(ctx, dHotCount.p, dHotBase.p, (uint64_t) nVQ + 1, scanTmp);
                    if (rcF != SD_OK) return rcF;
                    SD_HIP(ctx, sdD2H(ctx, &nLeft, dHotBase.p + nVQ, sizeof(uint64_t)));
                    SD_HIP(ctx, sdStreamSync(ctx));
                    pSegCount = dHotCount.p;
                    pOutBase = dHotBase.p;
                    if (ctx->profiling) {   // a counter beside the kernel times: hits that passed the filter (bench.py's algorithmic bytes)
                        sd_profile_entry &pe = ctx->profile["stat.prefilter_hits_left"];
                        pe.ms += (double) nLeft;
                        pe.launches += 1;
                    }
                    if (getenv("SD_DEBUG_TIMING"))
                        fprintf(stderr, "[prefilter] hot filter: %llu of %llu hits left (%.2f %%), %u segments\n", (unsigned long long) nLeft,
                                (unsigned long long) nHits, 100.0 * (double) nLeft / (double) std::max<uint64_t>(nHits, 1), nVQ);
                }
                if (useJoin) {   // the match's output arrays (lookup path: allocated for the gather, all hits)
                    SD_HIP(ctx, dKeyA.alloc(nLeft));
                    SD_HIP(ctx, dValA.alloc(nLeft));
                }
                const size_t nSlots = (size_t) nVQ * PF_NB_MAX;
                WsView<uint32_t> dSlotList(ctx, "pf.dSlotList");
                WsView<uint64_t> dBktStart(ctx, "pf.dBktStart");
                WsView<uint32_t> dBktCount(ctx, "pf.dBktCount");
                WsView<uint32_t> dBktEmit(ctx, "pf.dBktEmit");
                WsView<uint64_t> dEmitOff(ctx, "pf.dEmitOff");
                WsView<uint32_t> dQLog2(ctx, "pf.dQLog2");
                WsView<uint32_t> dQBins(ctx, "pf.dQBins");
                WsView<uint64_t> dBinBase(ctx, "pf.dBinBase");
                WsView<int> dFlag(ctx, "pf.dFlag");
                SD_HIP(ctx, dBktStart.alloc(nSlots));
                SD_HIP(ctx, dBktCount.alloc(nSlots));
                SD_HIP(ctx, dBktEmit.alloc(nSlots + 1));
                SD_HIP(ctx, dEmitOff.alloc(nSlots + 1));
                SD_HIP(ctx, dQLog2.alloc(nVQ));
                SD_HIP(ctx, dQBins.alloc(nVQ + 1));
                SD_HIP(ctx, dBinBase.alloc(nVQ + 1));
                SD_HIP(ctx, dFlag.alloc(1));
                WsView<uint2> dKVB(ctx, "pf.dKVB");
                SD_HIP(ctx, dKVB.alloc(nLeft));
                SD_HIP(ctx, hipMemsetAsync(dBktEmit.p, 0, (nSlots + 1) * sizeof(uint32_t), ctx->stream));
                SD_HIP(ctx, hipMemsetAsync(dFlag.p, 0, sizeof(int), ctx->stream));
                // the matched hits leave in dKeyA / dValA -- unless those hold the input (lookup path without the coarse split)
                uint32_t *outK = dKeyA.p, *outV = dValA.p;
                if (!pKV && pKey == dKeyA.p) {
                    outK = dKeyB.p;
                    outV = dValB.p;
                }
                WsView<uint8_t> dSegDone(ctx, "pf.dSegDone");
                const uint8_t *pSegDone = nullptr;
                if (useFilter && !(getenv("SD_PF_SEGMATCH") && atoi(getenv("SD_PF_SEGMATCH")) == 0)) {
                    // filtered segments that fit the LDS sorter are matched as a whole; the rest goes through partition / bucket_match
                    SD_HIP(ctx, dSegDone.alloc((size_t) nVQ + 1));
                    ProfScope ps(ctx, "prefilter_segment_match");
                    // (SD_PF_SM_PERSIST=n: n workgroups per CU looping over the segments; default one workgroup per segment -- measured
                    // round 5, interleaved: the kernel's in-pipeline time 3.5 - 3.9 s -> 1.5 s per 14 steps with n = 1 or 2, the throughput
                    // 2 347 / 2 436 (0) vs 2 434 (1) vs 2 225 / 2 330 (2): the waiting moves to the other kernels)
                    const int smPersist = getenv("SD_PF_SM_PERSIST") ? atoi(getenv("SD_PF_SM_PERSIST")) : 0;
                    const uint32_t smGrid = smPersist > 0 ? std::min<uint32_t>(nVQ, (uint32_t) smPersist * (uint32_t) ctx->prop.multiProcessorCount) : nVQ;
                    hipLaunchKernelGGL((segment_match_kernel<512, 8192>), dim3(smGrid), dim3(512), 0, ctx->stream, nVQ, pHitBase, pSegCount, pOutBase,
                                       tBitsV, pKey, pVal, pKV, outK, outV, dQLog2.p, dBktStart.p, dBktCount.p, dBktEmit.p, dSegDone.p,
                                       (const uint32_t *) dQSplit.p, (const uint32_t *) dQParts.p, (const uint32_t *) dQSplits.p, cBits,
                                       widePos ? tBitsV : 0);
                    pSegDone = dSegDone.p;
                }
                {
                    ProfScope ps(ctx, "prefilter_partition_hits");
                    // large tiles where the virtual queries are large (proteome-scale target sets), the small form for small inputs
                    const int tileMode = getenv("SD_PF_TILE") ? atoi(getenv("SD_PF_TILE")) : 1;
                    if (nLeft / std::max<uint32_t>(nVQ, 1) >= 16384 && tileMode == 2)
                        hipLaunchKernelGGL((partition_hits_kernel<1024, 8192>), dim3(nVQ), dim3(1024), 0, ctx->stream, nVQ, pHitBase, tBitsV, pKey,
                                           pVal, pKV, dKVB.p, dQLog2.p, dBktStart.p, dBktCount.p, dFlag.p, pSegCount, pOutBase, pSegDone);
                    else if (nLeft / std::max<uint32_t>(nVQ, 1) >= 16384 && tileMode == 1)
                        hipLaunchKernelGGL((partition_hits_kernel<512, 4096>), dim3(nVQ), dim3(512), 0, ctx->stream, nVQ, pHitBase, tBitsV, pKey,
                                           pVal, pKV, dKVB.p, dQLog2.p, dBktStart.p, dBktCount.p, dFlag.p, pSegCount, pOutBase, pSegDone);
                    else
                        hipLaunchKernelGGL((partition_hits_kernel<256, 2048>), dim3(nVQ), dim3(256), 0, ctx->stream, nVQ, pHitBase, tBitsV, pKey,
                                           pVal, pKV, dKVB.p, dQLog2.p, dBktStart.p, dBktCount.p, dFlag.p, pSegCount, pOutBase, pSegDone);
                }
                hipLaunchKernelGGL(bin_count_kernel, dim3(gridFor(nVQ + 1, 256)), dim3(256), 0, ctx->stream, nVQ, dQLog2.p, dQBins.p);
                int rc = exclusiveScanWiden(ctx, dQBins.p, dBinBase.p, nVQ + 1, scanTmp);
                if (rc != SD_OK) return rc;
                int hFlag = 0;
                uint64_t totalBins = 0;
                SD_HIP(ctx, sdD2H(ctx, &hFlag, dFlag.p, sizeof(int)));
                SD_HIP(ctx, sdD2H(ctx, &totalBins, dBinBase.p + nVQ, sizeof(uint64_t)));
                SD_HIP(ctx, sdStreamSync(ctx));
                if (hFlag == 0 && totalBins > 0) {
                    const uint32_t bigCap = (uint32_t) std::min<size_t>(nSlots, 1u << 26);   // every bucket may be oversize on very large target sets
                    WsView<uint32_t> dBigList(ctx, "pf.dBigList");
                    SD_HIP(ctx, dBigList.alloc(bigCap + 1));
                    uint32_t *dBigCount = dBigList.p + bigCap;
                    SD_HIP(ctx, hipMemsetAsync(dBigCount, 0, sizeof(uint32_t), ctx->stream));
                    {
                        ProfScope ps(ctx, "prefilter_bucket_match");
                        SD_HIP(ctx, dSlotList.alloc(totalBins));
                        hipLaunchKernelGGL(slot_list_kernel, dim3(gridFor(totalBins, 256)), dim3(256), 0, ctx->stream, (uint32_t) totalBins, nVQ,
                                           (const uint64_t *) dBinBase.p, dSlotList.p);
                        hipLaunchKernelGGL((bucket_match_kernel<128, PF_BUCKET_CAP>), dim3((unsigned) totalBins), dim3(128), 0, ctx->stream,
                                           nVQ, dBinBase.p, dQLog2.p, tBitsV, dBktStart.p, dBktCount.p, (const uint2 *) dKVB.p, outK,
                                           outV, dBktEmit.p, dFlag.p, (const uint32_t *) dSlotList.p, dBigList.p, dBigCount, bigCap,
                                           dQSplit.p, dQParts.p, dQSplits.p, cBits, widePos ? tBitsV : 0, (const uint32_t *) nullptr, 0u);
                    }
                    // the few buckets with one very hit-rich target (e.g. the query itself): their number stays on the device -- a launch of
                    // BIG_GRID workgroups that compare their index with the list's length takes them without a round trip to the host; a
                    // list longer than that (every bucket can be oversize on a very large target set) is finished after the next read
                    constexpr uint32_t BIG_GRID = 2048;
                    auto launchBig = [&](uint32_t first, uint32_t grid) {
                        ProfScope ps(ctx, "prefilter_bucket_match_big");
                        hipLaunchKernelGGL((bucket_match_kernel<256, PF_BUCKET_CAP_BIG>), dim3(grid), dim3(256), 0, ctx->stream, nVQ,
                                           dBinBase.p, dQLog2.p, tBitsV, dBktStart.p, dBktCount.p, (const uint2 *) dKVB.p, outK, outV,
                                           dBktEmit.p, dFlag.p, (const uint32_t *) dBigList.p, (uint32_t *) nullptr,
                                           (uint32_t *) nullptr, 0u, dQSplit.p, dQParts.p, dQSplits.p, cBits, widePos ? tBitsV : 0,
                                           (const uint32_t *) dBigCount, first);
                    };
                    launchBig(0, std::min<uint32_t>(BIG_GRID, bigCap));
                    rc = exclusiveScanWiden(ctx, dBktEmit.p, dEmitOff.p, nSlots + 1, scanTmp);
                    if (rc != SD_OK) return rc;
                    uint64_t nc64 = 0;
                    uint32_t nBig = 0;
                    SD_HIP(ctx, sdD2H(ctx, &nBig, dBigCount, sizeof(uint32_t)));
                    SD_HIP(ctx, sdD2H(ctx, &hFlag, dFlag.p, sizeof(int)));
                    SD_HIP(ctx, sdD2H(ctx, &nc64, dEmitOff.p + nSlots, sizeof(uint64_t)));
                    SD_HIP(ctx, sdStreamSync(ctx));
                    if (hFlag == 0 && nBig > BIG_GRID && nBig <= bigCap) {   // the rest of a long list, then the offsets again
                        for (uint32_t first = BIG_GRID; first < nBig; first += 1u << 20) launchBig(first, std::min<uint32_t>(nBig - first, 1u << 20));
                        rc = exclusiveScanWiden(ctx, dBktEmit.p, dEmitOff.p, nSlots + 1, scanTmp);
                        if (rc != SD_OK) return rc;
                        SD_HIP(ctx, sdD2H(ctx, &hFlag, dFlag.p, sizeof(int)));
                        SD_HIP(ctx, sdD2H(ctx, &nc64, dEmitOff.p + nSlots, sizeof(uint64_t)));
                        SD_HIP(ctx, sdStreamSync(ctx));
                    }
                    if (hFlag == 0) {
                        nCand = (uint32_t) nc64;
                        bucketDone = true;
                        if (nCand > 0) {
                            SD_HIP(ctx, dCKey.alloc(nCand));
                            SD_HIP(ctx, dCVal.alloc(nCand));
                            hipLaunchKernelGGL(bucket_collect_kernel, dim3((unsigned) totalBins), dim3(64), 0, ctx->stream, nVQ, dBinBase.p,
                                               dBktStart.p, dBktEmit.p, dEmitOff.p, outK, outV, dCKey.p, dCVal.p, (const uint32_t *) dSlotList.p);
                        }
                    }
                }
                if (!bucketDone && getenv("SD_DEBUG_TIMING"))
                    fprintf(stderr, "[prefilter] bucket path fell back: flag %d, %llu bins, %llu hits, %u queries\n", hFlag,
                            (unsigned long long) totalBins, (unsigned long long) nHits, bq);
                if (!bucketDone && useJoin) {   // the global-sort fallback belongs to the lookup path: redo the sub-batch there
                    forceLookup = true;
                    continue;
                }
                if (!bucketDone) {
                    // a bucket larger than the LDS capacity (or more target bits than the slots cover): redo this
                    // sub-batch with the global sort; dKeyA / dValA were overwritten by the emitted hits, so gather again
                    ProfScope ps(ctx, "prefilter_gather_hits");
                    hipLaunchKernelGGL(gather_hits_kernel, dim3(gridFor(nKmers, 256)), dim3(256), 0, ctx->stream, nKmers, dKStart.p,
                                       dKLen.p, dKPos.p, dHitBase.p, dPosBase.p, bq, (const uint2 *) T->dEntries, tBits, dQHitBase.p,
                                       dKeyA.p, dValA.p, dDiag.p, wideIdx ? (const uint32_t *) dKStartHi.p : (const uint32_t *) nullptr, 0);
                }
            }
            if (!bucketDone) {
            {
                ProfScope ps(ctx, "prefilter_sort_hits");
                // Hits arrive grouped by query in emission order.  A stable sort on the target bits alone makes every
                // (target, query) group contiguous with its emission order intact, which is all match_diag needs; the
                // few surviving candidates are put back into (query, target) order below.  19 target bits = 3 radix
                // passes instead of the 4 that (query, target) would take.
                const int rcS = sortPairs(ctx, dKeyA.p, dKeyB.p, dValA.p, dValB.p, nHits, 0, tBits);
                if (rcS != SD_OK) return rcS;
            }
            SD_HIP(ctx, dEmit.alloc(nHits + 1));
            SD_HIP(ctx, dEmitPos.alloc(nHits + 1));
            {
                ProfScope ps(ctx, "prefilter_match_diag");
                hipLaunchKernelGGL(match_diag_kernel, dim3(gridFor(nHits, 256)), dim3(256), 0, ctx->stream, nHits, dKeyB.p, dValB.p,
                                   dQSplit.p, tBits, dEmit.p);
            }
            SD_HIP(ctx, hipMemsetAsync(dEmit.p + nHits, 0, 1, ctx->stream));
            int rc = exclusiveScanWiden(ctx, dEmit.p, dEmitPos.p, nHits + 1, scanTmp);
            if (rc != SD_OK) return rc;
            uint64_t nc64 = 0;
            SD_HIP(ctx, sdD2H(ctx, &nc64, dEmitPos.p + nHits, sizeof(uint64_t)));
            SD_HIP(ctx, sdStreamSync(ctx));
            nCand = (uint32_t) nc64;
            }
        }
        if (!bucketDone && nHits > 0) {
            // the global-sort fallback matches across one split only: queries whose hit buffer overflows more than once are
            // reported there, not guessed
            std::vector<uint32_t> hParts(bq);
            SD_HIP(ctx, sdD2H(ctx, hParts.data(), dQParts.p, (size_t) bq * sizeof(uint32_t)));
            SD_HIP(ctx, sdStreamSync(ctx));
            bool any = false;
            for (uint32_t x = 0; x < bq; x++)
                if (hParts[x] >= 2) {
                    if (hUnsupported.empty()) hUnsupported.assign(bq, 0);
                    hUnsupported[x] = 1;
                    any = true;
                }
            if (any)
                sdFail(ctx, SD_EUNSUPPORTED, "queries whose hit buffer overflows more than once in a sub-batch that fell back to the global sort "
                       "(batch [%u, %u)): reported with outCount = UINT32_MAX", qBeg, qBeg + bq);
        }
        if (widePos && !bucketDone) {
            // the global-sort fallback carries 24-bit positions only: the heavy queries of this sub-batch are reported, not guessed
            if (hUnsupported.empty()) hUnsupported.assign(bq, 0);
            for (uint32_t x = 0; x < bq; x++)
                if (hStats[(size_t) x * 4 + 1] >= (1ull << 24)) hUnsupported[x] = 1;
            sdFail(ctx, SD_EUNSUPPORTED, "queries with >= 2^24 index hits in a sub-batch that fell back to the global sort (batch [%u, %u)): "
                   "reported with outCount = UINT32_MAX", qBeg, qBeg + bq);
        }
        hs.reset(new HostScope(ctx, "pf.score_keep"));
        DiagSrc diagSrc;
        diagSrc.hitDiag = useJoin ? (const uint16_t *) nullptr : (const uint16_t *) dDiag.p;
        diagSrc.qHitBase = dQHitBase.p;
        diagSrc.elems = dElems.p;
        diagSrc.kmerBase = dKmerBase.p;
        diagSrc.posBase = dPosBase.p;
        diagSrc.idxOffsets = T->dOffsets;
        diagSrc.entries = T->dEntries;
        if (nCand > 0) {
            if (!bucketDone) {
                SD_HIP(ctx, dCKey.alloc(nCand));
                SD_HIP(ctx, dCVal.alloc(nCand));
            }
            SD_HIP(ctx, dCScore.alloc(nCand));
            SD_HIP(ctx, dCLen.alloc(nCand));
            if (!bucketDone) {
                // compact into scratch, then a stable sort on the query bits restores (query, target, emission) order
                WsView<uint32_t> dCKey0(ctx, "pf.dCKey0");
                WsView<uint32_t> dCVal0(ctx, "pf.dCVal0");
                SD_HIP(ctx, dCKey0.alloc(nCand));
                SD_HIP(ctx, dCVal0.alloc(nCand));
                hipLaunchKernelGGL(compact_kernel, dim3(gridFor(nHits, 256)), dim3(256), 0, ctx->stream, nHits, dEmit.p, dEmitPos.p,
                                   dKeyB.p, dValB.p, (const int32_t *) nullptr, dCKey0.p, dCVal0.p, (int32_t *) nullptr);
                int endBit = tBits;
                uint32_t qb = bq - 1;
                while (qb) { endBit++; qb >>= 1; }
                endBit = std::min(32, std::max(endBit, tBits + 1));
                const int rcS = sortPairs(ctx, dCKey0.p, dCKey.p, dCVal0.p, dCVal.p, nCand, tBits, endBit);
                if (rcS != SD_OK) return rcS;
            }
            {
                ProfScope ps(ctx, "prefilter_score_diag");
                // SD_PF_SCORE_COOP=1 (sequence queries): the diagonal of every candidate first (a thread each), then eight lanes per candidate
                // down the diagonal (score_diag_coop_kernel).  Measured at 1 000 proteomes, interleaved in one process, identical rows
                // (profiles/r06n_score_coop.txt): 41.3 ms per 8 192 queries with the one-thread-per-candidate kernel, 42.4 ms with this one --
                // the walk is not what the kernel costs; the diagonal is (diagOf: the k-mer of the hit from the k-mer stream, its index list's
                // start, the entry of the target: three dependent reads of a line each per candidate, 8.5 MB per query).  Off.
                const bool coop = getenv("SD_PF_SCORE_COOP") && atoi(getenv("SD_PF_SCORE_COOP")) != 0;
                // the diagonal of a candidate from the target's residues (join path: 8-byte hits without a stored diagonal; k = 6 with
                // 24-bit ordinals beside the diagonal byte); SD_PF_DIAG_RES=0: from the index as before
                const int diagRes = (useJoin && T->k == 6 && posMask == 0xFFFFFFu && !(getenv("SD_PF_DIAG_RES") && atoi(getenv("SD_PF_DIAG_RES")) == 0)) ? 1 : 0;
                if (coop && !dProfAln) {
                    WsView<uint16_t> dCDiag(ctx, "pf.dCDiag");
                    SD_HIP(ctx, dCDiag.alloc(nCand));
                    hipLaunchKernelGGL(diag_of_kernel, dim3(gridFor(nCand, 256)), dim3(256), 0, ctx->stream, nCand, dCKey.p, dCVal.p, diagSrc, tBits,
                                       posMask, dCDiag.p, (const uint8_t *) T->dMasked, (const uint64_t *) T->dSeqOff, diagRes);
                    hipLaunchKernelGGL(score_diag_coop_kernel, dim3(gridFor(nCand, 32)), dim3(256), 0, ctx->stream, nCand, dCKey.p,
                                       (const uint16_t *) dCDiag.p, tBits, dQ.p, dQOff.p, dDB.p, T->dMasked, T->dSeqOff, dMat.p, dCScore.p, dCLen.p);
                } else {
                    hipLaunchKernelGGL(score_diag_kernel, dim3(gridFor(nCand, 256)), dim3(256), 0, ctx->stream, nCand, dCKey.p, dCVal.p,
                                       diagSrc, tBits, dQ.p, dQOff.p, dDB.p, T->dMasked, T->dSeqOff, dMat.p, dCScore.p, dCLen.p,
                                       dProfAln, posMask, diagRes);
                }
            }
            hipLaunchKernelGGL(cand_stats_kernel, dim3(gridFor(nCand, 256)), dim3(256), 0, ctx->stream, nCand, dCKey.p, dCLen.p, tBits,
                               (unsigned long long *) dStats.p);
            SD_HIP(ctx, dKeep.alloc(nCand));
            {
                ProfScope ps(ctx, "prefilter_keep_max");
                hipLaunchKernelGGL(keep_max_kernel, dim3(gridFor(nCand, 256)), dim3(256), 0, ctx->stream, nCand, dCKey.p, dCVal.p, dCScore.p, dKeep.p,
                                   tBits, posMask, (const uint32_t *) dQSplit.p, (const uint32_t *) dQParts.p, (const uint32_t *) dQSplits.p, diagSrc);
            }
        WsView<uint64_t> dK64(ctx, "pf.dK64");
        WsView<uint64_t> dKPos64(ctx, "pf.dKPos64");
            SD_HIP(ctx, dK64.alloc(nCand + 1));
            SD_HIP(ctx, dKPos64.alloc(nCand + 1));
            SD_HIP(ctx, hipMemsetAsync(dK64.p + nCand, 0, sizeof(uint64_t), ctx->stream));
            hipLaunchKernelGGL(flag_to_u64_kernel, dim3(gridFor(nCand, 256)), dim3(256), 0, ctx->stream, (uint64_t) nCand, dKeep.p, dK64.p);
            int rc = exclusiveScan(ctx, dK64.p, dKPos64.p, (uint64_t) nCand + 1, scanTmp);
            if (rc != SD_OK) return rc;
            // the number of kept candidates stays on the device (query_bounds_kernel reads it where the scan left it): the kept arrays
            // are sized by the candidates, no round trip to the host
            nKeptPtr = dKPos64.p + nCand;
            SD_HIP(ctx, dKKey.alloc((size_t) nCand + 1));
            SD_HIP(ctx, dKVal.alloc((size_t) nCand + 1));
            SD_HIP(ctx, dKScore.alloc((size_t) nCand + 1));
            hipLaunchKernelGGL(compact_kernel, dim3(gridFor(nCand, 256)), dim3(256), 0, ctx->stream, (uint64_t) nCand, dKeep.p,
                               dKPos64.p, dCKey.p, dCVal.p, dCScore.p, dKKey.p, dKVal.p, dKScore.p);
        } else {
            SD_HIP(ctx, dKKey.alloc(1));
            SD_HIP(ctx, dKVal.alloc(1));
            SD_HIP(ctx, dKScore.alloc(1));
            SD_HIP(ctx, dDiag.alloc(std::max<uint64_t>(nHits, 1)));
        }
        hs.reset(new HostScope(ctx, "pf.select"));
        SD_HIP(ctx, dQStart.alloc(bq + 1));
        hipLaunchKernelGGL(query_bounds_kernel, dim3(gridFor(bq + 1, 256)), dim3(256), 0, ctx->stream, bq, 0u, nKeptPtr, (const uint32_t *) dKKey.p, tBits, dQStart.p);
        WsView<sd_hit> dOut(ctx, "pf.dOut");
        WsView<uint32_t> dOutCount(ctx, "pf.dOutCount");
        SD_HIP(ctx, dOut.alloc((size_t) bq * maxHits));
        SD_HIP(ctx, dOutCount.alloc(bq));
        const bool selBig = maxHits + 1 > 2048;
        WsView<uint32_t> dSelBig(ctx, "pf.dSelBig");   // [bq] query list + [1] count
        if (selBig) {
            SD_HIP(ctx, dSelBig.alloc((size_t) bq + 1));
            SD_HIP(ctx, hipMemsetAsync(dSelBig.p + bq, 0, sizeof(uint32_t), ctx->stream));
        }
        {
            ProfScope ps(ctx, "prefilter_select_hits");
            if (maxHits + 1 <= 512 && !getenv("SD_PF_SEL4096"))   // short result lists: a 12-KB sorter instead of 48 KB (more workgroups per CU)
                hipLaunchKernelGGL(select_hits_kernel<1024>, dim3(bq), dim3(256), 0, ctx->stream, bq, dQStart.p, dKKey.p, dKVal.p, dKScore.p,
                                   diagSrc, tBits, par->binSize - 1, maxHits, par->minDiagScore, dIdent.p, dQOff.p, T->dSeqOff,
                                   par->covMode, par->covThr, dQ.p, dDB.p, dMat.p, dOut.p, dOutCount.p, dErr.p, dProfAln, posMask, useJoin ? binBits : -1,
                                   (const uint32_t *) dQSplit.p, (const uint32_t *) dQParts.p, (const uint32_t *) dQSplits.p,
                                   (uint32_t *) nullptr, (uint32_t *) nullptr);
            else if (maxHits + 1 <= 2048)
                hipLaunchKernelGGL(select_hits_kernel<4096>, dim3(bq), dim3(256), 0, ctx->stream, bq, dQStart.p, dKKey.p, dKVal.p, dKScore.p,
                                   diagSrc, tBits, par->binSize - 1, maxHits, par->minDiagScore, dIdent.p, dQOff.p, T->dSeqOff,
                                   par->covMode, par->covThr, dQ.p, dDB.p, dMat.p, dOut.p, dOutCount.p, dErr.p, dProfAln, posMask, useJoin ? binBits : -1,
                                   (const uint32_t *) dQSplit.p, (const uint32_t *) dQParts.p, (const uint32_t *) dQSplits.p,
                                   (uint32_t *) nullptr, (uint32_t *) nullptr);
            else   // longer lists: queries with more candidates at the cut than the LDS sorter holds go on to select_hits_big_kernel
                hipLaunchKernelGGL(select_hits_kernel<8192>, dim3(bq), dim3(256), 0, ctx->stream, bq, dQStart.p, dKKey.p, dKVal.p, dKScore.p,
                                   diagSrc, tBits, par->binSize - 1, maxHits, par->minDiagScore, dIdent.p, dQOff.p, T->dSeqOff,
                                   par->covMode, par->covThr, dQ.p, dDB.p, dMat.p, dOut.p, dOutCount.p, dErr.p, dProfAln, posMask, useJoin ? binBits : -1,
                                   (const uint32_t *) dQSplit.p, (const uint32_t *) dQParts.p, (const uint32_t *) dQSplits.p, dSelBig.p, dSelBig.p + bq);
        }
        SD_HIP(ctx, hipGetLastError());
        if (selBig) {
            // queries whose list is longer than the LDS sorter carries, with more candidates at the cut than it holds
            uint32_t nBigSel = 0;
            SD_HIP(ctx, sdD2H(ctx, &nBigSel, dSelBig.p + bq, sizeof(uint32_t)));
            SD_HIP(ctx, sdStreamSync(ctx));
            if (nBigSel > 0) {
                uint32_t stride = 1;
                while (stride < (uint32_t) maxHits) stride <<= 1;
                // scratch: 12 B per slot entry; at most ~1 GB per launch
                const uint32_t perLaunch = (uint32_t) std::max<uint64_t>(1, std::min<uint64_t>(nBigSel, (1ull << 30) / ((uint64_t) stride * 12)));
                WsView<unsigned long long> dSelKeys(ctx, "pf.dSelKeys");
                WsView<uint32_t> dSelPay(ctx, "pf.dSelPay");
                SD_HIP(ctx, dSelKeys.alloc((size_t) perLaunch * stride));
                SD_HIP(ctx, dSelPay.alloc((size_t) perLaunch * stride));
                ProfScope ps(ctx, "prefilter_select_hits_big");
                for (uint32_t b0 = 0; b0 < nBigSel; b0 += perLaunch) {
                    const uint32_t nb = std::min(perLaunch, nBigSel - b0);
                    hipLaunchKernelGGL(select_hits_big_kernel, dim3(nb), dim3(SELB_NT), 0, ctx->stream, (const uint32_t *) dSelBig.p + b0,
                                       (const uint32_t *) dQStart.p, (const uint32_t *) dKKey.p, (const uint32_t *) dKVal.p,
                                       (const int32_t *) dKScore.p, diagSrc, tBits, par->binSize - 1, maxHits, par->minDiagScore,
                                       (const uint32_t *) dIdent.p, (const uint64_t *) dQOff.p, (const uint64_t *) T->dSeqOff, par->covMode,
                                       par->covThr, (const uint8_t *) dQ.p, (const int8_t *) dDB.p, (const int8_t *) dMat.p, dOut.p,
                                       dOutCount.p, dProfAln, posMask, useJoin ? binBits : -1, (const uint32_t *) dQSplit.p,
                                       (const uint32_t *) dQParts.p, (const uint32_t *) dQSplits.p, dSelKeys.p, dSelPay.p, stride);
                }
                SD_HIP(ctx, hipGetLastError());
            }
        }
        hs.reset(new HostScope(ctx, "pf.download"));
        int hErr = 0;
        sd_hit *hOutP = nullptr;   // pinned, persistent: a pageable destination makes this copy a staged, synchronous one
        SD_HIP(ctx, pinGet(ctx, "pf.hOut", (size_t) bq * maxHits + 1, &hOutP));
        SD_HIP(ctx, sdD2H(ctx, &hErr, dErr.p, sizeof(int)));
        SD_HIP(ctx, hipMemcpyAsync(hOutP, dOut.p, (size_t) bq * maxHits * sizeof(sd_hit), hipMemcpyDeviceToHost, ctx->stream));
        SD_HIP(ctx, sdD2H(ctx, outCount + qBeg, dOutCount.p, bq * sizeof(uint32_t)));
        if (stats) SD_HIP(ctx, sdD2H(ctx, stats + (size_t) qBeg * 4, dStats.p, (size_t) bq * 4 * sizeof(uint64_t)));
        SD_HIP(ctx, sdStreamSync(ctx));
        (void) hErr;
        hs.reset(new HostScope(ctx, "pf.scatter"));
        // the caller's rows are par->maxHitsPerQuery wide
        for (uint32_t x = 0; x < bq; x++)
            if (outCount[qBeg + x] != UINT32_MAX)
                memcpy(outHits + (size_t) (qBeg + x) * par->maxHitsPerQuery, hOutP + (size_t) x * maxHits,
                       (size_t) std::min<uint32_t>(outCount[qBeg + x], maxHits) * sizeof(sd_hit));
        for (uint32_t x = 0; x < (uint32_t) hUnsupported.size(); x++)
            if (hUnsupported[x]) outCount[qBeg + x] = UINT32_MAX;   // per-query error slot: not computed (see above)
        hs.reset();
        forceLookup = false;
        qBeg += bq;
    }
    return SD_OK;
}

}  // extern "C"

int sd_prefilter_batch(sd_ctx *ctx, const sd_target *T, const sd_prefilter_params *par, uint32_t nQ,
                       const uint8_t *qResidues, const uint64_t *qOffsets, const int16_t *qKmerBias,
                       const int8_t *qDiagBias, const uint32_t *identityId, sd_hit *outHits, uint32_t *outCount,
                       uint64_t *stats) {
    return prefilterBatchImpl(ctx, T, par, nQ, qResidues, qOffsets, qKmerBias, qDiagBias, identityId, outHits, outCount, stats,
                              nullptr);
}

int sd_prefilter_profile_batch(sd_ctx *ctx, const sd_target *T, const sd_prefilter_params *par, uint32_t nQ,
                               const uint8_t *qLetters, const uint64_t *qOffsets, const int16_t *sortedScore,
                               const uint8_t *sortedIndex, const int8_t *alnProfile, const uint32_t *identityId,
                               sd_hit *outHits, uint32_t *outCount, uint64_t *stats) {
    if (!sortedScore || !sortedIndex || !alnProfile) return SD_EINVAL;
    ProfileQueries pq = {sortedScore, sortedIndex, alnProfile};
    std::vector<uint32_t> noIdentity;
    if (!identityId) {   // NULL: no query is in the target DB
        noIdentity.assign(nQ, 0xFFFFFFFFu);
        identityId = noIdentity.data();
    }
    return prefilterBatchImpl(ctx, T, par, nQ, qLetters, qOffsets, nullptr, nullptr, identityId, outHits, outCount, stats, &pq);
}
