// k-mer-major join: the front half of the prefilter (similar k-mers -> index hits) without a random table access
// per k-mer.  Included by sd_prefilter.hip (inside its anonymous namespace).
//
// The reference walks a query position by position and copies the index list of every similar k-mer
// (QueryMatcher.cpp:213-346, IndexTable.h:176-180).  Done literally on the device that is one random 8-byte read into
// the 256-MB offset table plus one random read of a (mostly one- or two-entry) list per k-mer: 64-128 B fetched per
// 8 useful.  Here a sub-batch's similar k-mers (~4*10^8 over a table of 6.4*10^7 k-mers) are first partitioned by
// k-mer into 2 048 ranges, each range is then joined with its slice of the offset table and of the entry array --
// 128 KB + a few hundred KB, read once and served from the XCD's L2 afterwards -- and the hits leave the join already
// split into groups of queries, which is the input form of the per-query bucket machinery (partition_hits /
// bucket_match).  Order: a k-mer carries its ordinal in the query's enumeration (kOrd), and an index list holds a
// target at most once (IndexTable::addSequence adds a sequence's k-mer only at its first position,
// IndexTable.h:383-392, lists sorted by sequence id, :182-189), so (kOrd, seqId) orders the hits of a query exactly
// like the reference's hit buffer; the value word carries kOrd where the lookup path carries the stream position.
//
// Both partitions (k-mers by k-mer range, hits by query group) are done by JP_WGS persistent workgroups in two passes
// with no global atomics: pass one counts per (workgroup, bin), a column prefix turns the counts into private write
// cursors, pass two reorders tile by tile in LDS and appends runs at those cursors -- a workgroup's writes into a bin
// are consecutive in memory, so partial lines merge in its XCD's L2.

// Workgroup shapes.  These kernels run beside the alignment stage's kernels (other streams); half-CU shapes (512 threads / 88 KB
// for the partition, 256 threads / 33 KB for the join) were measured in the running pipeline and did not pay: +25 % isolated
// time, end-to-end throughput within noise -- unlike partition_hits, where 512 x 4 096 beats both neighbours.
constexpr int JP_WGS = 256;            // persistent workgroups of the k-mer partition (one per CU)
constexpr int JP_NT = 1024;
// k-mer ranges: bin = kmer >> KP_LOW.  A range's slices of the offset table and of the entry array must stay in an XCD's L2
// while its elements are joined (KP_LOW = 15: 128 KB + ~350 KB at 100 proteomes; 17: 512 KB + ~1.4 MB of 4 MB), and the hits a
// query has in one range (62 k-mers x ~1.2 at KP_LOW = 15, four times that at 17) are the runs join_scatter writes: longer runs
// = fewer partial lines, which the memory side turns into read-modify-writes.
#ifndef SD_KP_LOW
#define SD_KP_LOW 15
#endif
constexpr int KP_LOW = SD_KP_LOW;
constexpr int KP_BINS = KP_LOW == 15 ? 2048 : (KP_LOW == 16 ? 1024 : 512);   // >= 64 000 000 >> KP_LOW
constexpr int KP_SHIFT = 38 + KP_LOW;  // element >> KP_SHIFT = kmer >> KP_LOW
constexpr uint64_t KP_LOW_MASK = (1ull << KP_LOW) - 1;
static_assert((64000000 >> KP_LOW) < KP_BINS && KP_LOW <= 19, "k-mer ranges");
constexpr int KP_TILE = 16384;
constexpr int KP_PER = KP_TILE / JP_NT;
constexpr int JQ_MAX = 4096;           // queries per sub-batch
constexpr int JX_COLS_MAX = 15360;     // (query, target range) columns of the ranged scatter: 60 KB of LDS cursors
constexpr int JJ_WGS = 512;            // persistent workgroups of the join (two per CU)
constexpr int JJ_NT = 1024;
#ifndef SD_JE
#define SD_JE 2
#endif
constexpr int JE = SD_JE;              // k-mers per lane and step of the join (independent load chains in flight)
constexpr int JC = JJ_NT * JE;         // k-mers per join chunk: 64 * JE per wavefront
constexpr uint32_t JOIN_ORD_LIMIT = 1u << 24;   // k-mers per query the value word can order
constexpr uint64_t JP_PAD = ~0ull;     // filler behind a k-mer range (ranges are padded to whole join chunks)

// element of the k-mer stream: kmer << 38 | stream index (< 2^30) << 8 | low byte of the query position
__device__ __forceinline__ uint64_t jpElem(uint32_t kmer, uint64_t streamIdx, int i) {
    return ((uint64_t) kmer << 38) | (streamIdx << 8) | (uint64_t) (i & 0xFF);
}

// K2 (join form): similar k-mers in the reference's enumeration order, no index access
__global__ void __launch_bounds__(256)
emit_kmers_join_kernel(uint64_t nPos, const uint64_t *__restrict__ posBase, uint32_t nQ, const uint8_t *__restrict__ qRes,
                       const uint64_t *__restrict__ qOff, const int16_t *__restrict__ kmerBias, int kmerThr,
                       const int16_t *__restrict__ ext3Score, const uint16_t *__restrict__ ext3Index,
                       const uint64_t *__restrict__ kmerBase, uint64_t *__restrict__ elems,
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
    const int cutoff1 = (int) (short) (pi.thr - (int) row1[0]);
    const uint16_t *cum0 = ext3Cum ? ext3Cum + (size_t) pi.idx0 * EXT3_CUM_SPAN : nullptr;
    const uint16_t *cum1 = ext3Cum ? ext3Cum + (size_t) pi.idx1 * EXT3_CUM_SPAN : nullptr;
    const int n0 = ext3Cum ? countGETab(cum0, ext3Lo, cutoff1) : countGE(row0, 8000, cutoff1);
    uint64_t base = kmerBase[p];
    __shared__ uint32_t sIncl[4][64];
    __shared__ uint32_t sK0[4][64];
    uint32_t *myIncl = sIncl[threadIdx.x >> 6], *myK0 = sK0[threadIdx.x >> 6];
    for (int a0 = 0; a0 < n0; a0 += 64) {
        const int a = a0 + lane;
        uint32_t c = 0;
        if (a < n0) {
            const int cutoff2 = (int) (short) (pi.thr - (int) row0[a]);
            c = (uint32_t) (ext3Cum ? countGETab(cum1, ext3Lo, cutoff2) : countGE(row1, 8000, cutoff2));
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
        myK0[lane] = a < n0 ? (uint32_t) ix0[a] : 0u;
        __builtin_amdgcn_wave_barrier();
        for (uint32_t t = lane; t < chunkTotal; t += 64) {
            int lo = 0, hi = 63;   // smallest o with incl[o] > t
#pragma unroll
            for (int st = 0; st < 6; st++) {
                const int mid = (lo + hi) >> 1;
                if (myIncl[mid] > t) hi = mid;
                else lo = mid + 1;
            }
            const uint32_t before = lo ? myIncl[lo - 1] : 0u;
            const uint32_t km = myK0[lo] + 8000u * (uint32_t) ix1[t - before];
            elems[base + t] = jpElem(km, base + t, pi.i);
        }
        base += chunkTotal;
    }
}

// first stream index of every query (32 bit: a sub-batch holds at most 2^30 k-mers); flag[0] |= 1 when a query has
// JOIN_ORD_LIMIT k-mers or more
__global__ void join_query_base_kernel(uint32_t nQ, const uint64_t *__restrict__ posBase, const uint64_t *__restrict__ kmerBase,
                                       uint32_t *__restrict__ qKmerBase, int *__restrict__ flag) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nQ) return;
    const uint64_t b = kmerBase[posBase[q]];
    qKmerBase[q] = (uint32_t) b;
    if (q < nQ && kmerBase[posBase[q + 1]] - b >= JOIN_ORD_LIMIT) atomicOr(flag, 1);
}

// ---- partition of the k-mer stream by k-mer range
__global__ void __launch_bounds__(JP_NT)
kp_hist_kernel(const uint64_t *__restrict__ elems, uint64_t n, uint32_t *__restrict__ counts /* [JP_WGS][KP_BINS] */) {
    __shared__ uint32_t hist[KP_BINS];
    for (int b = threadIdx.x; b < KP_BINS; b += JP_NT) hist[b] = 0;
    __syncthreads();
    const uint64_t nTiles = (n + KP_TILE - 1) / KP_TILE, perWg = (nTiles + gridDim.x - 1) / gridDim.x;
    for (uint64_t t = blockIdx.x * perWg; t < min(nTiles, (blockIdx.x + 1) * perWg); t++) {
        const uint64_t base = t * KP_TILE;
#pragma unroll
        for (int j = 0; j < KP_PER; j++) {
            const uint64_t idx = base + (uint64_t) j * JP_NT + threadIdx.x;
            if (idx < n) atomicAdd(&hist[(uint32_t) (elems[idx] >> KP_SHIFT)], 1u);
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < KP_BINS; b += JP_NT) counts[(size_t) blockIdx.x * KP_BINS + b] = hist[b];
}

// counts[r][c] -> exclusive prefix down every column, in place; total[c] = column sum
__global__ void __launch_bounds__(256)
col_prefix_kernel(uint32_t *__restrict__ m, int rows /* multiple of 4 */, int cols, uint32_t *__restrict__ total) {
    __shared__ uint32_t part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), pt = threadIdx.x >> 6;
    const int RP = rows / 4;
    uint32_t sum = 0;
    if (c < cols)
        for (int r = pt * RP; r < (pt + 1) * RP; r++) sum += m[(size_t) r * cols + c];
    part[pt][threadIdx.x & 63] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int x = 0; x < pt; x++) run += part[x][threadIdx.x & 63];
    if (c < cols) {
        for (int r = pt * RP; r < (pt + 1) * RP; r++) {
            const uint32_t v = m[(size_t) r * cols + c];
            m[(size_t) r * cols + c] = run;
            run += v;
        }
        if (pt == 3) total[c] = run;
    }
}

// exclusive scan of n <= 4096 totals into 64-bit bases (n + 1 values), one workgroup; pad > 0: every total is rounded
// up to a multiple of pad first (k-mer ranges start on join-chunk boundaries)
constexpr int SS_NT = 1024;
__global__ void __launch_bounds__(SS_NT)
small_scan_kernel(const uint32_t *__restrict__ in, int n, uint64_t *__restrict__ out, uint32_t pad) {
    __shared__ uint64_t part[SS_NT / 64];
    const int t = threadIdx.x;
    uint64_t v[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        v[j] = (t * 4 + j < n) ? (uint64_t) in[t * 4 + j] : 0;
        if (pad) v[j] = (v[j] + pad - 1) / pad * pad;
        sum += v[j];
    }
    uint64_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint64_t o = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    uint64_t run = incl - sum;
    for (int w = 0; w < (t >> 6); w++) run += part[w];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (t * 4 + j <= n) out[t * 4 + j] = run;
        run += v[j];
    }
    if (t * 4 + 4 == n) out[n] = run;   // n = 4 * SS_NT: the total has no thread of its own
}

// exclusive scan of arr[0..n) in LDS by all NT threads of the workgroup (n <= 8 * NT); part[NT / 64] receives the total
template <int NT>
__device__ __forceinline__ void jpBlockScan2(uint32_t *arr, int n, uint32_t *part) {
    const int t = threadIdx.x;
    const int per = (n + NT - 1) / NT;
    const int lo = t * per, hi = min(n, lo + per);
    uint32_t v[8];
    uint32_t sum = 0;
    for (int x = lo; x < hi; x++) {
        v[x - lo] = arr[x];
        sum += v[x - lo];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += o;
    }
    if ((t & 63) == 63) part[t >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < (t >> 6); w++) run += part[w];
    for (int x = lo; x < hi; x++) {
        arr[x] = run;
        run += v[x - lo];
    }
    __syncthreads();
    if (t == NT - 1) part[NT / 64] = run;   // grand total
    __syncthreads();
}

// Element of the k-mer-sorted stream (written by kp_scatter_kernel): the range is implied by the position, so the word
// has room for the owning query and the k-mer's ordinal in it -- the join looks neither up:
//   k-mer & 0x7FFF << 44 | query (< 4096) << 32 | ordinal (< 2^24) << 8 | low byte of the query position
__device__ __forceinline__ uint64_t jpSortedElem(uint64_t e, uint32_t q, uint32_t ord) {
    return (((e >> 38) & KP_LOW_MASK) << 44) | ((uint64_t) q << 32) | ((uint64_t) ord << 8) | (e & 0xFFull);
}

__global__ void __launch_bounds__(JP_NT)
kp_scatter_kernel(const uint64_t *__restrict__ in, uint64_t n, const uint32_t *__restrict__ prefix /* [JP_WGS][KP_BINS] */,
                  const uint64_t *__restrict__ binBase, const uint32_t *__restrict__ qKmerBase, uint32_t nQ,
                  uint64_t *__restrict__ out) {
    __shared__ uint64_t stage[KP_TILE];
    __shared__ uint32_t cnt[KP_BINS], lstart[KP_BINS], cur[KP_BINS];
    __shared__ uint32_t part[JP_NT / 64 + 1];
    __shared__ uint32_t sQ0;
    const int t = threadIdx.x;
    for (int b = t; b < KP_BINS; b += JP_NT) {
        cur[b] = (uint32_t) binBase[b] + prefix[(size_t) blockIdx.x * KP_BINS + b];
        cnt[b] = 0;
    }
    __syncthreads();
    // a workgroup takes a contiguous range of tiles and its share of a bin lies behind the shares of the workgroups before
    // it: every bin stays in stream order -- i.e. ordered by query -- down to the tile, which is what lets the join write
    // a chunk's hits as a few long per-query runs
    const uint64_t nTiles = (n + KP_TILE - 1) / KP_TILE, perWg = (nTiles + gridDim.x - 1) / gridDim.x;
    for (uint64_t tile = blockIdx.x * perWg; tile < min(nTiles, (blockIdx.x + 1) * perWg); tile++) {
        const uint64_t base = tile * KP_TILE;
        const int tn = (int) min((uint64_t) KP_TILE, n - base);
        if (t == 0) {   // query of the tile's first k-mer: the tile's k-mers belong to it or to the next few
            uint32_t lo = 0, hi = nQ;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (qKmerBase[mid] <= (uint32_t) base) lo = mid;
                else hi = mid;
            }
            sQ0 = lo;
        }
        uint64_t e[KP_PER];
        uint32_t r[KP_PER];
#pragma unroll
        for (int j = 0; j < KP_PER; j++) {
            const int x = j * JP_NT + t;
            if (x < tn) e[j] = in[base + x];
        }
#pragma unroll
        for (int j = 0; j < KP_PER; j++) {
            const int x = j * JP_NT + t;
            if (x < tn) r[j] = atomicAdd(&cnt[(uint32_t) (e[j] >> KP_SHIFT)], 1u);
        }
        __syncthreads();
        for (int b = t; b < KP_BINS; b += JP_NT) lstart[b] = cnt[b];
        __syncthreads();
        jpBlockScan2<JP_NT>(lstart, KP_BINS, part);
#pragma unroll
        for (int j = 0; j < KP_PER; j++) {
            const int x = j * JP_NT + t;
            if (x < tn) stage[lstart[(uint32_t) (e[j] >> KP_SHIFT)] + r[j]] = e[j];
        }
        __syncthreads();
        const uint32_t q0 = sQ0;
#pragma unroll
        for (int j = 0; j < KP_PER; j++) {
            const int x = j * JP_NT + t;
            if (x < tn) {
                const uint64_t v = stage[x];
                const uint32_t b = (uint32_t) (v >> KP_SHIFT);
                const uint32_t sidx = (uint32_t) (v >> 8) & 0x3FFFFFFFu;
                uint32_t q = q0;
                while (q + 1 < nQ && qKmerBase[q + 1] <= sidx) q++;
                out[cur[b] + ((uint32_t) x - lstart[b])] = jpSortedElem(v, q, sidx - qKmerBase[q]);
            }
        }
        __syncthreads();
        for (int b = t; b < KP_BINS; b += JP_NT) {
            cur[b] += cnt[b];
            cnt[b] = 0;
        }
        __syncthreads();
    }
}

// behind every k-mer range: fillers up to the next chunk boundary, and the range of every chunk
__global__ void __launch_bounds__(256)
kp_finish_kernel(const uint32_t *__restrict__ total, const uint64_t *__restrict__ binBase, uint64_t *__restrict__ sorted,
                 uint16_t *__restrict__ chunkBin) {
    const uint32_t b = blockIdx.x;
    const uint64_t s = binBase[b], e = binBase[b + 1];
    for (uint64_t x = s + total[b] + threadIdx.x; x < e; x += 256) sorted[x] = JP_PAD;
    for (uint64_t c = s / JC + threadIdx.x; c < e / JC; c += 256) chunkBin[c] = (uint16_t) b;
}

// ---- the join.  Work split (identical in the count and the scatter pass): XCD x = workgroup & 7 owns the x-th eighth
// of the k-mer-sorted stream, its workgroups take that range's chunks round-robin, so the CUs of an XCD work on
// neighbouring chunks -- the same one or two k-mer ranges -- at any time and the table slices stay in that XCD's L2.
// A range is in query order (see kp_scatter_kernel), so a chunk's k-mers touch a few dozen queries and its hits are a
// few long runs: they are written straight to the per-query segments, positions handed out by LDS cursors (lanes of
// a wavefront mostly share the query, the returned positions are consecutive, the stores coalesce).
struct JoinSpan {
    uint64_t begin, end;   // this XCD's element range (whole chunks)
    uint32_t slot, nSlots;
};
__device__ __forceinline__ JoinSpan joinSpan(uint64_t n /* multiple of JC */) {
    const uint32_t xcd = blockIdx.x & 7u;
    uint64_t per = (n + 7) / 8;
    per = (per + JC - 1) / JC * JC;
    JoinSpan s;
    s.begin = min(n, (uint64_t) xcd * per);
    s.end = min(n, s.begin + per);
    s.slot = blockIdx.x >> 3;
    s.nSlots = gridDim.x >> 3;
    return s;
}

__global__ void __launch_bounds__(JJ_NT)
join_count_kernel(const uint64_t *__restrict__ sorted, const uint64_t *__restrict__ nPtr /* elements of the sorted stream: read on the device, no host round trip */, const uint16_t *__restrict__ chunkBin,
                  const uint32_t *__restrict__ idxOffsets, uint32_t *__restrict__ counts /* [JJ_WGS][cols] */, int cols,
                  unsigned long long *__restrict__ wgTotal /* [JJ_WGS]: 64-bit, the per-query counts are 32 */) {
    __shared__ uint32_t hist[JQ_MAX];
    __shared__ unsigned long long sTotal;
    const uint64_t n = *nPtr;
    unsigned long long mine = 0;
    if (threadIdx.x == 0) sTotal = 0;
    for (int x = threadIdx.x; x < cols; x += JJ_NT) hist[x] = 0;
    __syncthreads();
    const JoinSpan sp = joinSpan(n);
    const int lane = threadIdx.x & 63;
    const uint64_t step = (uint64_t) sp.nSlots * JC;
    uint64_t nx[JE];   // the next chunk's k-mers are on their way while this one is worked on
    uint32_t nxHi = 0;
    {
        const uint64_t b0 = sp.begin + (uint64_t) sp.slot * JC;
        if (b0 < sp.end) {
            nxHi = (uint32_t) chunkBin[b0 / JC] << KP_LOW;
#pragma unroll
            for (int j = 0; j < JE; j++) nx[j] = __builtin_nontemporal_load(sorted + b0 + (uint64_t) j * JJ_NT + threadIdx.x);
        }
    }
    for (uint64_t base = sp.begin + (uint64_t) sp.slot * JC; base < sp.end; base += step) {
        const uint32_t kmHi = nxHi;
        uint64_t e[JE];
        uint32_t len[JE];
#pragma unroll
        for (int j = 0; j < JE; j++) e[j] = nx[j];
        if (base + step < sp.end) {
            nxHi = (uint32_t) chunkBin[(base + step) / JC] << KP_LOW;
#pragma unroll
            for (int j = 0; j < JE; j++) nx[j] = __builtin_nontemporal_load(sorted + base + step + (uint64_t) j * JJ_NT + threadIdx.x);
        }
#pragma unroll
        for (int j = 0; j < JE; j++) {
            len[j] = 0;
            if (e[j] != JP_PAD) {
                const uint32_t km = kmHi | (uint32_t) (e[j] >> 44);
                uint32_t se[2];   // list start and end in one (4-byte aligned) 8-byte read
                __builtin_memcpy(se, idxOffsets + km, 8);
                len[j] = se[1] - se[0];
            }
        }
#pragma unroll
        for (int j = 0; j < JE; j++) {
            const uint32_t q = (uint32_t) (e[j] >> 32) & 0xFFFu;
            mine += len[j];
            // the wavefront's k-mers are in query order: one LDS add per query present, not one per lane
            unsigned long long todo = __ballot(len[j] != 0);
            while (todo) {
                const int leader = __ffsll((long long) todo) - 1;
                const uint32_t qL = __shfl(q, leader, 64);
                const bool in = len[j] != 0 && q == qL;
                uint32_t sum = in ? len[j] : 0u;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
                if (lane == leader) atomicAdd(&hist[qL], sum);
                todo &= ~__ballot(in);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sTotal, mine);
    __syncthreads();
    for (int x = threadIdx.x; x < cols; x += JJ_NT) counts[(size_t) blockIdx.x * cols + x] = hist[x];
    if (threadIdx.x == 0) wgTotal[blockIdx.x] = sTotal;
}

// per-query hit totals that take part (queries taken out of the batch count nothing)
__global__ void join_effective_totals_kernel(uint32_t nQ, const uint32_t *__restrict__ qHits, const uint32_t *__restrict__ qSplit,
                                             uint32_t *__restrict__ eff) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nQ) eff[q] = qSplit[q] == QUERY_UNSUPPORTED ? 0u : qHits[q];
}

// Every wavefront works through its 64 * JE k-mers of a chunk on its own (no workgroup barrier inside the loop, so the
// wavefronts of a CU hide each other's memory latency): list starts and lengths, a wave scan of the lengths, then the
// lists flattened over the lanes -- hit f of the wavefront belongs to the k-mer x with off[x] <= f < off[x + 1].
//
// RANGES (target sets beyond ~150 proteomes, where the average query of a sub-batch has more than 2 * 10^5 index hits): the hits
// leave the join already split into C = 2^cBits target ranges per query -- the columns are (query, range) pairs, the cursors sit in
// LDS as before (JX_COLS_MAX of them), every (query, range) sub-segment is the "virtual query" the hot-target filter and the bucket
// machinery take -- instead of per query, to be split by three more streaming passes over the whole hit stream (coarse_count /
// coarse_scatter: 24 B per hit).  The keys do not change: key >> (tBits - cBits) IS the virtual query.  The price is the COUNT form
// of this kernel: the same walk with a histogram of the columns in place of the stores (a hit's range is known only from its entry,
// so the count pass reads the entries too; join_count_kernel, which reads list lengths only, still sizes the sub-batch).  Inside a
// sub-segment the hits are in no particular order (the lanes take positions from LDS atomics): the bucket sort restores emission
// order from the k-mer ordinal in the value, as it does behind coarse_scatter_kernel.
template <bool NT_STORE, bool RANGES, bool COUNT>
__global__ void __launch_bounds__(JJ_NT)
join_scatter_kernel(const uint64_t *__restrict__ sorted, const uint64_t *__restrict__ nPtr, const uint16_t *__restrict__ chunkBin,
                    const uint32_t *__restrict__ idxOffsets, const uint2 *__restrict__ entries, uint32_t nQ,
                    const uint32_t *__restrict__ prefix /* [JJ_WGS][cols]: hits of the workgroups before this one, per column (scatter) */,
                    int cols, const uint64_t *__restrict__ qHitBase /* per column */, int tBits, const uint32_t *__restrict__ qSplit,
                    uint2 *__restrict__ outKV, int cBits, uint32_t *__restrict__ counts /* COUNT: [JJ_WGS][cols] */,
                    uint8_t *__restrict__ outR6 /* nullable (plain scatter): the top six target bits of every hit, at the hit's position --
                                                   what the coarse split's count pass reads instead of the 8-byte hits */,
                    int r6Shift) {
    __shared__ uint32_t qcur[RANGES ? JX_COLS_MAX : JQ_MAX];   // this workgroup's write position inside every column's segment (< 2^32 hits per sub-batch)
    constexpr int WE = 64 * JE;   // k-mers per wavefront and step
    __shared__ uint32_t wOff[JJ_NT / 64][WE + 1], wStart[JJ_NT / 64][WE], wKey[JJ_NT / 64][WE], wVal[JJ_NT / 64][WE];
    __shared__ uint8_t wOwn[JJ_NT / 64][256];   // per wavefront: the k-mer whose list starts at a flat slot of the current 256-hit batch
    static_assert(WE <= 256, "k-mer indices of a wavefront's step are bytes");
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint64_t n = *nPtr;
    const int rShift = tBits - cBits;
    const uint32_t rMask = (1u << cBits) - 1u;
    if (COUNT) {
        for (int c = t; c < cols; c += JJ_NT) qcur[c] = qSplit[(uint32_t) c >> cBits] == QUERY_UNSUPPORTED ? 0xFFFFFFFFu : 0u;
    } else if (RANGES) {
        for (int c = t; c < cols; c += JJ_NT)
            qcur[c] = qSplit[(uint32_t) c >> cBits] == QUERY_UNSUPPORTED ? 0xFFFFFFFFu : (uint32_t) qHitBase[c] + prefix[(size_t) blockIdx.x * cols + c];
    } else {
        for (uint32_t q = t; q < nQ; q += JJ_NT)
            qcur[q] = qSplit[q] == QUERY_UNSUPPORTED ? 0xFFFFFFFFu : (uint32_t) qHitBase[q] + prefix[(size_t) blockIdx.x * cols + q];
    }
    __syncthreads();
    uint32_t *eOff = wOff[wv], *eStart = wStart[wv], *eKey = wKey[wv], *eVal = wVal[wv];
    const JoinSpan sp = joinSpan(n);
    const uint64_t step = (uint64_t) sp.nSlots * JC;
    uint64_t nx[JE];   // the next chunk's k-mers are on their way while this one is worked on
    uint32_t nxHi = 0;
    {
        const uint64_t b0 = sp.begin + (uint64_t) sp.slot * JC;
        if (b0 < sp.end) {
            nxHi = (uint32_t) chunkBin[b0 / JC] << KP_LOW;
#pragma unroll
            for (int j = 0; j < JE; j++)   // streamed once: not to displace the table slices in L2
                nx[j] = __builtin_nontemporal_load(sorted + b0 + (uint64_t) wv * WE + (uint64_t) j * 64 + lane);
        }
    }
    for (uint64_t base = sp.begin + (uint64_t) sp.slot * JC; base < sp.end; base += step) {
        const uint32_t kmHi = nxHi;
        // the wavefront's k-mers: JE consecutive runs of 64 (element x = j * 64 + lane of the wavefront's step)
        uint64_t e[JE];
        uint32_t len[JE], st[JE];
#pragma unroll
        for (int j = 0; j < JE; j++) e[j] = nx[j];
        if (base + step < sp.end) {
            nxHi = (uint32_t) chunkBin[(base + step) / JC] << KP_LOW;
#pragma unroll
            for (int j = 0; j < JE; j++)
                nx[j] = __builtin_nontemporal_load(sorted + base + step + (uint64_t) wv * WE + (uint64_t) j * 64 + lane);
        }
#pragma unroll
        for (int j = 0; j < JE; j++) {
            len[j] = 0;
            st[j] = 0;
            if (e[j] != JP_PAD) {
                const uint32_t km = kmHi | (uint32_t) (e[j] >> 44);
                uint32_t se[2];   // list start and end in one (4-byte aligned) 8-byte read
                __builtin_memcpy(se, idxOffsets + km, 8);
                st[j] = se[0];
                len[j] = se[1] - se[0];
            }
        }
        uint32_t carry = 0;
        uint32_t myOff[JE];
#pragma unroll
        for (int j = 0; j < JE; j++) {
            const int x = j * 64 + lane;
            const uint32_t q = (uint32_t) (e[j] >> 32) & 0xFFFu;
            if (len[j] && qcur[RANGES ? (q << cBits) : q] == 0xFFFFFFFFu) len[j] = 0;   // query taken out of the batch
            eStart[x] = st[j];
            eKey[x] = q << tBits;
            eVal[x] = (uint32_t) (((e[j] & 0xFF) << 24) | ((e[j] >> 8) & 0xFFFFFF));
            uint32_t incl = len[j];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            myOff[j] = carry + incl - len[j];
            eOff[x] = myOff[j];
            carry += __shfl(incl, 63, 64);
        }
        const uint32_t total = carry;
        if (lane == 63) eOff[WE] = total;
        __builtin_amdgcn_wave_barrier();
        // Flat slot f of the wavefront's hits belongs to the last k-mer x with eOff[x] <= f.  Per batch of 256 slots every k-mer whose
        // list STARTS inside the batch writes its index at that slot (a byte in LDS), and a running maximum over the slots hands
        // every slot its owner (the k-mer indices grow with the slots; a list that began in an earlier batch is the carry) -- two
        // dozen shuffles per four hits where a binary search of eOff cost seven dependent LDS reads per hit
        // (short lists -- 1.2 entries per k-mer at 100 proteomes, one or two batches per step -- keep the search: the scan's fixed cost per
        // batch is what a batch of mostly empty slots cannot repay; isolated at 100 proteomes 12.5 ms per 8 192 queries with the search,
        // 16.8 with the scan; at 1 000 proteomes, twelve entries per k-mer, 73 with the search and 68 with the scan)
        const bool ownerScan = total > 3u * 256u;
        uint8_t *own = wOwn[wv];
        uint32_t ownCarry = 0;
        for (uint32_t f0 = 0; f0 < total; f0 += 4 * 64) {
            uint32_t x4[4], a4[4];
            if (ownerScan) {
            ((uint32_t *) own)[lane] = 0u;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < JE; j++)
                if (len[j] && myOff[j] >= f0 && myOff[j] < f0 + 256u) own[myOff[j] - f0] = (uint8_t) (j * 64 + lane);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t f = f0 + (uint32_t) j * 64 + (uint32_t) lane;
                uint32_t v = own[j * 64 + lane];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = __shfl_up(v, off, 64);
                    if (lane >= off) v = max(v, o);
                }
                v = max(v, ownCarry);
                ownCarry = __shfl(v, 63, 64);
                x4[j] = 0xFFFFFFFFu;
                if (f < total) {
                    x4[j] = v;
                    a4[j] = eStart[v] + (f - eOff[v]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t f = f0 + (uint32_t) j * 64 + (uint32_t) lane;
                x4[j] = 0xFFFFFFFFu;
                if (f < total) {
                    int lo = 0, hi = WE;   // last x with eOff[x] <= f
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (eOff[mid] <= f) lo = mid;
                        else hi = mid;
                    }
                    x4[j] = (uint32_t) lo;
                    a4[j] = eStart[lo] + (f - eOff[lo]);
                }
            }
            }
            uint2 en[4];
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x4[j] != 0xFFFFFFFFu) en[j] = entries[a4[j]];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool live = x4[j] != 0xFFFFFFFFu;
                uint32_t v = 0, kq = 0xFFFFFFFFu, pos = 0;
                if (live) {
                    v = eVal[x4[j]];
                    kq = eKey[x4[j]];
                }
                if (RANGES) {
                    // a column per (query, target range): every lane takes its own position (or counts itself)
                    if (live) {
                        const uint32_t col = ((kq >> tBits) << cBits) | ((en[j].x >> rShift) & rMask);
                        if (COUNT) atomicAdd(&qcur[col], 1u);
                        else pos = atomicAdd(&qcur[col], 1u);
                    }
                } else {
                // positions: one LDS add per query present in the wavefront; lanes of a query get consecutive positions
                unsigned long long todo = __ballot(live);
                while (todo) {
                    const int leader = __ffsll((long long) todo) - 1;
                    const uint32_t kL = __shfl(kq, leader, 64);
                    const unsigned long long mask = __ballot(live && kq == kL);
                    uint32_t b0 = 0;
                    if (lane == leader) b0 = atomicAdd(&qcur[kL >> tBits], (uint32_t) __popcll(mask));
                    b0 = __shfl(b0, leader, 64);
                    if (live && kq == kL) pos = b0 + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
                    todo &= ~mask;
                }
                }
                if (live && !COUNT) {
                    const unsigned long long kv = ((unsigned long long) (((((v >> 24) - en[j].y) & 0xFFu) << 24) | (v & 0xFFFFFFu)) << 32) | (kq | en[j].x);
                    if (NT_STORE) __builtin_nontemporal_store(kv, (unsigned long long *) outKV + pos);
                    else ((unsigned long long *) outKV)[pos] = kv;
                    if (!RANGES && outR6) outR6[pos] = (uint8_t) ((en[j].x >> r6Shift) & 63u);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (COUNT) {
        __syncthreads();
        for (int c = t; c < cols; c += JJ_NT) counts[(size_t) blockIdx.x * cols + c] = qcur[c] == 0xFFFFFFFFu ? 0u : qcur[c];
    }
}

// per (query, range) column: the query's first column carries the marker of a query taken out of the batch -- the other columns of
// such a query must not count either (the COUNT form initialises every column of it to the marker and writes 0)

// Where the reference's hit buffer (cap entries) overflows inside a query, in k-mer ordinals: whenever the next k-mer's list
// would fill the buffer, the buffered part is matched on its own and the buffer starts again with that list
// (QueryMatcher.cpp:281-316) -- so a split is the ordinal of the first k-mer of a part.  One workgroup per query; only queries
// with at least cap hits walk their k-mers (enumeration order, list lengths from the offset table), and only the blocks of
// 256 k-mers in which the running part can reach cap are walked one k-mer at a time.
__global__ void __launch_bounds__(256)
query_splits_join_kernel(uint32_t nQ, const uint32_t *__restrict__ qKmerBase, const uint64_t *__restrict__ elems,
                         const uint32_t *__restrict__ idxOffsets, const uint32_t *__restrict__ qHits, uint64_t cap,
                         uint32_t *__restrict__ qSplit, uint32_t *__restrict__ qParts, uint32_t *__restrict__ qSplits /* [nQ][PF_SPLITS_MAX] */,
                         int *__restrict__ flag) {
    __shared__ uint32_t lens[256];
    __shared__ unsigned long long part[4];
    __shared__ unsigned long long inPart;
    __shared__ uint32_t nSplit;
    __shared__ int stop;
    const uint32_t q = blockIdx.x;
    if ((uint64_t) qHits[q] < cap) {
        if (threadIdx.x == 0) {
            qSplit[q] = 0xFFFFFFFFu;
            qParts[q] = 0;
        }
        return;
    }
    const uint32_t k0 = qKmerBase[q], nk = qKmerBase[q + 1] - k0;
    uint32_t *out = qSplits + (size_t) q * PF_SPLITS_MAX;
    if (threadIdx.x == 0) {
        inPart = 0;
        nSplit = 0;
        stop = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t x0 = 0; x0 < nk && !stop; x0 += 256) {
        const uint32_t x = x0 + threadIdx.x;
        uint32_t len = 0;
        if (x < nk) {
            const uint32_t km = (uint32_t) (elems[k0 + x] >> 38);
            len = idxOffsets[km + 1] - idxOffsets[km];
        }
        lens[threadIdx.x] = len;
        unsigned long long sum = len;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
        if (lane == 0) part[wave] = sum;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long total = part[0] + part[1] + part[2] + part[3];
            if (inPart + total < cap) {
                inPart += total;   // no list of this block can fill the buffer
            } else {
                unsigned long long have = inPart;
                for (uint32_t j = 0; j < 256 && x0 + j < nk; j++) {
                    const unsigned long long l = lens[j];
                    if (have + l >= cap) {
                        if (nSplit < (uint32_t) PF_SPLITS_MAX) out[nSplit] = x0 + j;
                        nSplit++;
                        have = 0;
                        if (l >= cap) {   // a single list as large as the buffer: the reference stops matching here (:312-314)
                            stop = 1;
                            break;
                        }
                    }
                    have += l;
                }
                inPart = have;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const bool unsupported = stop || nSplit > (uint32_t) PF_SPLITS_MAX;
        qParts[q] = unsupported ? 0u : nSplit;
        qSplit[q] = unsupported ? QUERY_UNSUPPORTED : (nSplit ? out[0] : 0xFFFFFFFFu);
        if (unsupported) atomicExch(flag, 1);
    }
}

// per-query statistics of the join path: k-mers and index hits
__global__ void join_stats_kernel(uint32_t nQ, const uint32_t *__restrict__ qKmerBase, const uint32_t *__restrict__ qHits,
                                  uint64_t *__restrict__ stats) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nQ) return;
    stats[4 * q] = qKmerBase[q + 1] - qKmerBase[q];
    stats[4 * q + 1] = qHits[q];
}

// The 16-bit diagonal of a hit for the few that survive the match.  Lookup path: kept per stream position at gather
// time.  Join path: recomputed -- the query position from the k-mer's ordinal (binary search over the per-position
// stream bases), the target position from the k-mer's index list (sorted by sequence id, one entry per sequence).
struct DiagSrc {
    const uint16_t *hitDiag;     // lookup path (nullptr selects the join path)
    const uint64_t *qHitBase;
    const uint64_t *elems;       // join path: the query-major k-mer stream
    const uint64_t *kmerBase;    // first stream index of every position
    const uint64_t *posBase;     // first position of every query
    const uint32_t *idxOffsets;
    const uint2 *entries;
};

__device__ __forceinline__ uint16_t diagOf(const DiagSrc &S, uint32_t q, uint32_t ord, uint32_t sid) {
    if (S.hitDiag) return S.hitDiag[S.qHitBase[q] + ord];
    uint64_t lo = S.posBase[q], hi = S.posBase[q + 1];
    const uint64_t p0 = lo, sidx = S.kmerBase[lo] + ord;
    while (hi - lo > 1) {   // last position whose stream base is <= sidx
        const uint64_t mid = (lo + hi) >> 1;
        if (S.kmerBase[mid] <= sidx) lo = mid;
        else hi = mid;
    }
    const int i = (int) (lo - p0);
    const uint32_t km = (uint32_t) (S.elems[sidx] >> 38);
    uint32_t a = S.idxOffsets[km], b = S.idxOffsets[km + 1];
    while (b - a > 1) {   // the list's entry of sequence sid
        const uint32_t mid = (a + b) >> 1;
        if (S.entries[mid].x <= sid) a = mid;
        else b = mid;
    }
    return (uint16_t) (i - (int) S.entries[a].y);
}

// The same diagonal without the index (join path, k = 6).  A hit carries the low byte of its diagonal, d8 = (i - j) & 255, and its
// k-mer ordinal; i follows from the ordinal as above, so the target position is j = (i - d8) mod 256 + 256 m.  With at most 256
// k-mer positions left in the target behind the smallest candidate there is only that one (the hit exists); otherwise the candidates
// are told apart by the k-mer itself, read from the target's own masked residues -- the lines the diagonal walk reads next anyway --
// and the smallest position that holds the hit's k-mer is the indexed one (the index keeps a sequence's FIRST occurrence of a k-mer,
// IndexTable.h:383-392, and that occurrence is in the residue class).  Against diagOf this drops the reads of the offset table and
// of the entry list (a 128-byte line each per candidate) always, and the read of the k-mer stream for targets of up to 265 residues.
__device__ __forceinline__ uint16_t diagFromResidues(const DiagSrc &S, uint32_t q, uint32_t ord, uint32_t sid, uint32_t d8,
                                                     const uint8_t *__restrict__ ts /* the target's masked residues */, int tL) {
    uint64_t lo = S.posBase[q], hi = S.posBase[q + 1];
    const uint64_t p0 = lo, sidx = S.kmerBase[lo] + ord;
    while (hi - lo > 1) {   // last position whose stream base is <= sidx
        const uint64_t mid = (lo + hi) >> 1;
        if (S.kmerBase[mid] <= sidx) lo = mid;
        else hi = mid;
    }
    const int i = (int) (lo - p0);
    const int jMax = tL - SPAN6;   // last k-mer position of the target
    int j = (i - (int) d8) & 255;
    if (j + 256 > jMax) return (uint16_t) (i - j);
    const uint32_t km = (uint32_t) (S.elems[sidx] >> 38);
    for (; j <= jMax; j += 256) {
        uint32_t idx = 0, pw = 1;
        bool x = false;
#pragma unroll
        for (int p = 0; p < 6; p++) {
            const uint32_t a = ts[j + c_seed6[p]];
            x |= a >= 20u;
            idx += a * pw;
            pw *= 20u;
        }
        if (!x && idx == km) return (uint16_t) (i - j);
    }
    return diagOf(S, q, ord, sid);   // (not reached for a hit of the index these residues were indexed from)
}
