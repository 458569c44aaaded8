// Device-wide prefix scans and a stable LSD radix sort of (uint32 key, 32- or 64-bit value) pairs -- this library's own (no hipCUB /
// rocPRIM anywhere in the tree).  Included by sd_prefilter.hip, sd_sw.hip and sd_index_build.hip inside their anonymous namespaces.
//
// Why not the library forms: the library's scan is a decoupled look-back (tiles spin on their predecessors) and its one-sweep
// sort runs 1 024-thread workgroups that do the same; both are fine on an idle device and take milliseconds when three other
// streams hold most CUs -- a spinning workgroup occupies the CU its predecessor is waiting for.  The forms
// here have no inter-workgroup waiting at all: reduce-then-scan in three launches, and per radix pass a count launch, a column
// prefix and a scatter launch over PERSISTENT workgroups that own contiguous tile ranges (so a digit's elements leave in
// workgroup order = input order: stable without atomics on global memory), 256-thread workgroups with <= 37 KB of LDS that fit
// beside the score wavefronts.
#ifndef SD_SCAN_SORT_H
#define SD_SCAN_SORT_H

// ---------------------------------------------------------------------------------------------------------------------------
// scans: out[i] = op(in[0..i)) (exclusive) or op(in[0..i]) (inclusive), accumulated in Acc
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 16, SCAN_BLOCK = 256, SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

struct ScanSum64 {
    typedef uint64_t Acc;
    static __device__ __forceinline__ Acc identity() { return 0; }
    static __device__ __forceinline__ Acc apply(Acc a, Acc b) { return a + b; }
};
struct ScanMax32 {
    typedef uint32_t Acc;
    static __device__ __forceinline__ Acc identity() { return 0; }
    static __device__ __forceinline__ Acc apply(Acc a, Acc b) { return a > b ? a : b; }
};

template <typename Acc>
__device__ __forceinline__ Acc scanShflUp(Acc v, int off) { return __shfl_up(v, off, 64); }
template <typename Acc>
__device__ __forceinline__ Acc scanShflXor(Acc v, int off) { return __shfl_xor(v, off, 64); }

template <typename T, typename Op>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_tile_sums_kernel(const T *__restrict__ in, uint64_t n, typename Op::Acc *__restrict__ tileSum) {
    typedef typename Op::Acc Acc;
    __shared__ Acc part[SCAN_BLOCK / 64];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
    Acc sum = Op::identity();
    if (base + SCAN_ITEMS <= n) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) sum = Op::apply(sum, (Acc) in[base + j]);
    } else {
        for (int j = 0; j < SCAN_ITEMS; j++)
            if (base + j < n) sum = Op::apply(sum, (Acc) in[base + j]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum = Op::apply(sum, scanShflXor(sum, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        Acc t = Op::identity();
        for (int w = 0; w < SCAN_BLOCK / 64; w++) t = Op::apply(t, part[w]);
        tileSum[blockIdx.x] = t;
    }
}

// in place: tileSum[i] <- op(tileSum[0..i)); one workgroup, eight consecutive tiles per thread and round
template <typename Op>
__global__ void __launch_bounds__(1024)
scan_tile_bases_kernel(typename Op::Acc *__restrict__ tileSum, uint32_t nTiles) {
    typedef typename Op::Acc Acc;
    constexpr int PER = 8;
    __shared__ Acc part[16];
    __shared__ Acc carry;
    if (threadIdx.x == 0) carry = Op::identity();
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t c0 = 0; c0 < nTiles; c0 += 1024 * PER) {
        const uint32_t i0 = c0 + threadIdx.x * PER;
        Acc v[PER];
        Acc sum = Op::identity();
#pragma unroll
        for (int j = 0; j < PER; j++) {
            v[j] = (i0 + j < nTiles) ? tileSum[i0 + j] : Op::identity();
            sum = Op::apply(sum, v[j]);
        }
        Acc incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const Acc o = scanShflUp(incl, off);
            if (lane >= off) incl = Op::apply(o, incl);
        }
        Acc excl = scanShflUp(incl, 1);   // the lanes before this one
        if (lane == 0) excl = Op::identity();
        if (lane == 63) part[wave] = incl;
        __syncthreads();
        Acc run = carry;
        for (int w = 0; w < wave; w++) run = Op::apply(run, part[w]);
        run = Op::apply(run, excl);
        Acc endOfThread = Op::apply(run, sum);
#pragma unroll
        for (int j = 0; j < PER; j++) {
            if (i0 + j < nTiles) tileSum[i0 + j] = run;
            run = Op::apply(run, v[j]);
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = endOfThread;
        __syncthreads();
    }
}

template <typename T, typename Op, bool INCLUSIVE, typename Out>
__global__ void __launch_bounds__(SCAN_BLOCK)
scan_apply_kernel(const T *__restrict__ in, uint64_t n, const typename Op::Acc *__restrict__ tileBase, Out *__restrict__ out) {
    typedef typename Op::Acc Acc;
    __shared__ Acc part[SCAN_BLOCK / 64];
    const uint64_t base = (uint64_t) blockIdx.x * SCAN_TILE + (uint64_t) threadIdx.x * SCAN_ITEMS;
    Acc v[SCAN_ITEMS];
    Acc sum = Op::identity();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        v[j] = (base + j < n) ? (Acc) in[base + j] : Op::identity();
        sum = Op::apply(sum, v[j]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Acc incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Acc o = scanShflUp(incl, off);
        if (lane >= off) incl = Op::apply(o, incl);
    }
    Acc excl = scanShflUp(incl, 1);
    if (lane == 0) excl = Op::identity();
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    Acc run = tileBase[blockIdx.x];
    for (int w = 0; w < wave; w++) run = Op::apply(run, part[w]);
    run = Op::apply(run, excl);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; j++) {
        if (INCLUSIVE) run = Op::apply(run, v[j]);
        if (base + j < n) out[base + j] = (Out) run;
        if (!INCLUSIVE) run = Op::apply(run, v[j]);
    }
}

// tmp: device scratch of at least ceil(n / SCAN_TILE) accumulators (the callers keep it in their workspace)
template <typename T, typename Op, bool INCLUSIVE, typename Out>
inline hipError_t sdScanLaunch(hipStream_t stream, const T *in, Out *out, uint64_t n, typename Op::Acc *tileSum) {
    if (n == 0) return hipSuccess;
    const uint64_t nTiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL((scan_tile_sums_kernel<T, Op>), dim3((unsigned) nTiles), dim3(SCAN_BLOCK), 0, stream, in, n, tileSum);
    hipLaunchKernelGGL((scan_tile_bases_kernel<Op>), dim3(1), dim3(1024), 0, stream, tileSum, (uint32_t) nTiles);
    hipLaunchKernelGGL((scan_apply_kernel<T, Op, INCLUSIVE, Out>), dim3((unsigned) nTiles), dim3(SCAN_BLOCK), 0, stream, in, n,
                       (const typename Op::Acc *) tileSum, out);
    return hipGetLastError();
}
inline size_t sdScanTmpBytes(uint64_t n) { return ((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(uint64_t); }

// ---------------------------------------------------------------------------------------------------------------------------
// stable LSD radix sort of (key, value) pairs, 8 bits per pass
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int RS_NT = 256;                 // threads per workgroup
constexpr int RS_PER = 16;                 // elements per thread and tile
constexpr int RS_TILE = RS_NT * RS_PER;    // 4 096 elements: 32 KB of staging
constexpr int RS_BINS = 256;
constexpr int RS_WGS_MAX = 1024;           // persistent workgroups (rows of the count matrix)

__device__ __forceinline__ uint32_t rsDigit(uint32_t key, int shift, uint32_t mask) { return (key >> shift) & mask; }

__global__ void __launch_bounds__(RS_NT)
rs_hist_kernel(const uint32_t *__restrict__ keys, uint32_t n, int shift, uint32_t mask, uint32_t *__restrict__ counts /* [gridDim.x][RS_BINS] */) {
    __shared__ uint32_t hist[RS_BINS];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t nTiles = (n + RS_TILE - 1) / RS_TILE, perWg = (nTiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * perWg, t1 = min(nTiles, t0 + perWg);
    for (uint32_t tile = t0; tile < t1; tile++) {
        const uint32_t base = tile * RS_TILE;
#pragma unroll
        for (int j = 0; j < RS_PER; j++) {
            const uint32_t i = base + (uint32_t) j * RS_NT + threadIdx.x;
            if (i < n) atomicAdd(&hist[rsDigit(keys[i], shift, mask)], 1u);
        }
    }
    __syncthreads();
    counts[(size_t) blockIdx.x * RS_BINS + threadIdx.x] = hist[threadIdx.x];
}

// counts[r][c] -> exclusive prefix down every column (in place) plus the exclusive prefix of the column totals: afterwards
// counts[r][c] = first output position of workgroup r's elements with digit c.  One workgroup of RS_BINS threads.
__global__ void __launch_bounds__(RS_BINS)
rs_prefix_kernel(uint32_t *__restrict__ counts, int rows) {
    __shared__ uint32_t part[RS_BINS / 64];
    const int c = threadIdx.x;
    uint32_t sum = 0;
    for (int r = 0; r < rows; r++) sum += counts[(size_t) r * RS_BINS + c];
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((c & 63) >= off) incl += o;
    }
    if ((c & 63) == 63) part[c >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < (c >> 6); w++) run += part[w];
    for (int r = 0; r < rows; r++) {
        const uint32_t v = counts[(size_t) r * RS_BINS + c];
        counts[(size_t) r * RS_BINS + c] = run;
        run += v;
    }
}

// A tile is ranked wave by wave: element j of wavefront w is tile position w * 1024 + j * 64 + lane, so (w, j, lane) is input
// order.  Inside a step the lanes with equal digits find each other with eight ballots (one per digit bit); the rank of a lane is
// the wavefront's running count of its digit plus the number of equal lanes below it.
template <typename V>   // value type: uint32_t (the prefilter's and the alignment stage's sorts) or uint64_t (the index build's records)
struct RsPair {
    uint32_t x;
    V y;
};
template <typename V>
__global__ void __launch_bounds__(RS_NT)
rs_scatter_kernel(const uint32_t *__restrict__ kIn, const V *__restrict__ vIn, uint32_t n, int shift, uint32_t mask,
                  const uint32_t *__restrict__ starts /* [gridDim.x][RS_BINS] from rs_prefix_kernel */, uint32_t *__restrict__ kOut,
                  V *__restrict__ vOut) {
    constexpr int NW = RS_NT / 64;
    __shared__ RsPair<V> stage[RS_TILE];
    __shared__ uint32_t wcnt[NW][RS_BINS];   // per wavefront: running count, then the wavefront's offset inside the digit
    __shared__ uint32_t binStart[RS_BINS];   // first staging slot of every digit
    __shared__ uint32_t cur[RS_BINS];        // this workgroup's next output position per digit
    __shared__ uint32_t part[NW];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    cur[t] = starts[(size_t) blockIdx.x * RS_BINS + t];
    const uint32_t nTiles = (n + RS_TILE - 1) / RS_TILE, perWg = (nTiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * perWg, t1 = min(nTiles, t0 + perWg);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint32_t tile = t0; tile < t1; tile++) {
        const uint32_t base = tile * RS_TILE;
        for (int x = t; x < NW * RS_BINS; x += RS_NT) (&wcnt[0][0])[x] = 0;
        __syncthreads();
        uint32_t k[RS_PER], rank[RS_PER];
        V v[RS_PER];
#pragma unroll
        for (int j = 0; j < RS_PER; j++) {
            const uint32_t i = base + (uint32_t) wv * (64 * RS_PER) + (uint32_t) j * 64 + (uint32_t) lane;
            k[j] = i < n ? kIn[i] : 0xFFFFFFFFu;
            v[j] = i < n ? vIn[i] : (V) 0;
        }
#pragma unroll
        for (int j = 0; j < RS_PER; j++) {
            const uint32_t i = base + (uint32_t) wv * (64 * RS_PER) + (uint32_t) j * 64 + (uint32_t) lane;
            const bool live = i < n;
            const uint32_t d = rsDigit(k[j], shift, mask);
            unsigned long long peers = __ballot(live);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const unsigned long long has = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? has : ~has;
            }
            rank[j] = 0;
            if (live) {
                rank[j] = wcnt[wv][d] + (uint32_t) __popcll(peers & below);
            }
            __builtin_amdgcn_wave_barrier();
            if (live && (peers & below) == 0) wcnt[wv][d] += (uint32_t) __popcll(peers);   // the lowest lane of the group
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // digit t: offsets of the wavefronts inside the digit, and the digit's total
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t c = wcnt[w][t];
            wcnt[w][t] = tot;
            tot += c;
        }
        uint32_t incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) part[wv] = incl;
        __syncthreads();
        uint32_t bs = incl - tot;
        for (int w = 0; w < wv; w++) bs += part[w];
        binStart[t] = bs;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RS_PER; j++) {
            const uint32_t i = base + (uint32_t) wv * (64 * RS_PER) + (uint32_t) j * 64 + (uint32_t) lane;
            if (i < n) {
                const uint32_t d = rsDigit(k[j], shift, mask);
                stage[binStart[d] + wcnt[wv][d] + rank[j]] = RsPair<V>{k[j], v[j]};
            }
        }
        __syncthreads();
        const uint32_t tn = min((uint32_t) RS_TILE, n - base);
#pragma unroll
        for (int j = 0; j < RS_PER; j++) {
            const uint32_t x = (uint32_t) j * RS_NT + (uint32_t) t;
            if (x < tn) {
                const RsPair<V> e = stage[x];
                const uint32_t d = rsDigit(e.x, shift, mask);
                const uint32_t p = cur[d] + (x - binStart[d]);
                kOut[p] = e.x;
                vOut[p] = e.y;
            }
        }
        __syncthreads();
        cur[t] += tot;
        __syncthreads();
    }
}

// scratch the sort needs: a second (key, value) buffer pair of n elements when the number of passes is even ... the caller hands
// over both buffer pairs; counts: RS_WGS_MAX * RS_BINS uint32
inline size_t sdRadixSortCountsBytes() { return (size_t) RS_WGS_MAX * RS_BINS * sizeof(uint32_t); }

// Sorts (stably) by key bits [beginBit, endBit).  The result lands in (kOut, vOut); (kTmp, vTmp) are scratch of n elements each
// (distinct from the inputs, which are left untouched).  n < 2^32.
template <typename V>
inline hipError_t sdRadixSortPairs(hipStream_t stream, const uint32_t *kIn, const V *vIn, uint32_t *kOut, V *vOut, uint32_t *kTmp,
                                   V *vTmp, uint32_t n, int beginBit, int endBit, uint32_t *counts) {
    if (n == 0) return hipSuccess;
    const int passes = std::max(1, (endBit - beginBit + 7) / 8);
    const uint32_t nTiles = (n + RS_TILE - 1) / RS_TILE;
    const unsigned wgs = (unsigned) std::min<uint32_t>(nTiles, RS_WGS_MAX);
    const uint32_t *sk = kIn;
    const V *sv = vIn;
    for (int p = 0; p < passes; p++) {
        // the last pass writes (kOut, vOut): the passes before it alternate so that it does
        uint32_t *dk = ((passes - 1 - p) & 1) ? kTmp : kOut;
        V *dv = ((passes - 1 - p) & 1) ? vTmp : vOut;
        const int shift = beginBit + 8 * p;
        const int bits = std::min(8, endBit - shift);
        const uint32_t mask = bits >= 8 ? 0xFFu : ((1u << std::max(bits, 1)) - 1u);
        hipLaunchKernelGGL(rs_hist_kernel, dim3(wgs), dim3(RS_NT), 0, stream, sk, n, shift, mask, counts);
        hipLaunchKernelGGL(rs_prefix_kernel, dim3(1), dim3(RS_BINS), 0, stream, counts, (int) wgs);
        hipLaunchKernelGGL(rs_scatter_kernel<V>, dim3(wgs), dim3(RS_NT), 0, stream, sk, sv, n, shift, mask, (const uint32_t *) counts, dk, dv);
        sk = dk;
        sv = dv;
    }
    return hipGetLastError();
}

#endif
