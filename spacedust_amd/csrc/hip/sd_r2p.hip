// result2profile on the device: the position-specific sequence weights of PSSMCalculator
// (M/src/alignment/PSSMCalculator.cpp:394-588, computeSequenceWeights / the sub-alignment per column of HH-suite) -- the
// O(columns^2 x rows) part of the step between search iterations; everything around it (alignment assembly, the greedy
// diversity filter, pseudo counts, scores, masking: O(columns x rows) or sequential by nature) stays in
// csrc/host/sd_result2profile.cpp, which calls sdR2pColumnWeightsDevice for a batch of filtered alignments.
//
// One workgroup per centre sequence; columns are walked in order (the row counts per column and cell code are updated where a
// row starts or ends, and the weights persist where nothing changes).  Inside a column every summation keeps the
// reference's order and operand types, so the profile bytes equal the host's:
//   * a row's weight = the sum of its cells' shares in column order               -> one lane per row (the alignment is kept
//     column-major for this loop: the lanes of a wavefront read neighbouring bytes);
//   * the sub-alignment's frequencies, summed over the rows in row order           -> one lane per column (row-major copy);
//   * share = 1 / (distinct x count): the reference takes an approximate reciprocal (rcpps) and refines it once -- the bits
//     depend on the CPU, so the HOST tabulates it for every integer product that can occur and the kernel looks it up;
//   * the entropy sum over (column, residue) is one chain of fused multiply-subtracts, run by one lane over terms prepared
//     by all (flog2: double Horner form with the host compiler's fused multiply-adds, read off its object code).
// `#pragma clang fp contract(off)` + explicit fma / fmaf: exactly the contractions g++ -ffp-contract=fast makes in the host
// code, nothing else.
#include "sd_common.h"

#include <algorithm>
#include <chrono>
#include <vector>

#include <cfloat>

#pragma clang fp contract(off)

namespace {

constexpr int R2P_NT = 512;      // threads per alignment: one per column in the column-parallel passes, one per row in the row-parallel ones
constexpr int R2P_ACT = 768;     // active rows whose index and weight are staged in LDS for the frequency pass (deeper alignments read them from memory)
constexpr int kCodes = 24;   // cell codes 0..22 (0..19 residues, 20 any, 21 gap, 22 end gap), padded
constexpr int kAnyC = 20, kGapC = 21, kEndGapC = 22, kRes = 20;

struct R2pTask {
    uint64_t cellOff;      // row-major cells: [nRows][stride]
    uint64_t cmOff;        // column-major scratch: [L][rowStride]
    uint64_t weightOff;    // global weights / local weights / active list: [nRows]
    uint64_t colOff;       // first position of the centre in freq / eff
    uint64_t scratchOff;   // count / share / sub / lg: (L + 1) * kCodes each
    uint32_t nRows, L, stride, rowStride;
};

// MathUtil::flog2 (M/src/commons/MathUtil.h:121-141) as g++ compiles it: the polynomial in double, four fused multiply-adds
__device__ __forceinline__ float r2pFlog2(float x) {
    if (x <= 0) return -128;
    int ix = __float_as_int(x);
    const float e = (float) (((ix & 0x7F800000) >> 23) - 0x7f);
    ix = (ix & 0x007FFFFF) | 0x3f800000;
    float m = __int_as_float(ix);
    m -= 1.0f;   // (the host subtracts the double 1.0 from the promoted float and rounds back: the same value)
    const double d = (double) m;
    double t = 0.0440047;
    t = fma(t, d, -0.1903190);
    t = fma(t, d, 0.4123442);
    t = fma(t, d, -0.7077702);
    t = fma(t, d, 1.441740);
    const float p = (float) (d * t);
    return p + e;
}

// MathUtil::fpow2 (:143-163), float Horner form with fused multiply-adds; the caller stores the result in a float
__device__ __forceinline__ float r2pFpow2(float x) {
    if (x >= (float) FLT_MAX_EXP) return FLT_MAX;
    if (x <= (float) FLT_MIN_EXP) return 0.0f;
    const float tx = (x - 0.5f) + (float) (3 << 22);
    const int lx = __float_as_int(tx) - 0x4b400000;
    const float dx = x - (float) lx;
    float t = 0.0134929f;
    t = fmaf(t, dx, 0.0520749f);
    t = fmaf(t, dx, 0.241404f);
    t = fmaf(t, dx, 0.693019f);
    t = fmaf(t, dx, 1.0f);
    return __int_as_float(__float_as_int(t) + (lx << 23));
}

// scale v[0..20) to sum 1 (MathUtil::NormalizeTo1): sequential float sum, factor = 1.0 / total in double rounded to float
__device__ __forceinline__ void r2pScale20(float *v, const double *fallback) {
    float total = 0.0f;
    for (int a = 0; a < kRes; a++) total += v[a];
    if (total != 0.0f) {
        const float factor = (float) (1.0 / (double) total);
        for (int a = 0; a < kRes; a++) v[a] *= factor;
    } else if (fallback) {
        for (int a = 0; a < kRes; a++) v[a] = (float) fallback[a];
    }
}

// LDSCOLS: alignments of up to this many columns keep the sub-alignment frequencies and their logarithms in LDS (60 KB at 320:
// two workgroups per CU).  Longer ones keep them in the global scratch, but never work on them there: a thread accumulates a
// column's frequencies in an LDS slice of its own, and the entropy chain -- one thread, 20 dependent FMAs per column -- reads
// them from LDS again, LDSCOLS columns staged at a time by the whole workgroup (the chain chasing global loads was 0.8 s for
// one alignment of 1 200 columns, the tail of the whole batch).
template <int LDSCOLS>
__global__ void __launch_bounds__(R2P_NT, 4)   // two workgroups per CU: four waves per SIMD
r2p_column_weights_kernel(const R2pTask *__restrict__ tasks, char *cellsRM, char *cellsCM, const float *__restrict__ globalWeight,
                          float *localWeight, uint32_t *activeList, const float *__restrict__ rcpTable, uint32_t rcpN,
                          const double *__restrict__ background, int *countG, float *shareG, float *subG, float *lgG,
                          float *__restrict__ freq, float *__restrict__ eff, int *__restrict__ errFlag,
                          long long *__restrict__ taskTicks /* NULL, or the 100 MHz ticks every task took (SD_DEBUG_TIMING) */) {
    const long long tick0 = taskTicks ? (long long) wall_clock64() : 0;
    // one array: a long alignment's per-thread accumulators (R2P_NT x kCodes) span both halves
    __shared__ float ldsF[2 * LDSCOLS * kCodes];
    static_assert(R2P_NT * kCodes <= 2 * LDSCOLS * kCodes, "the accumulator slices of a long alignment must fit");
    float *const subL = ldsF, *const lgL = ldsF + LDSCOLS * kCodes;
    __shared__ int sJmin, sJmax, sChanged, sActive, sPart;
    __shared__ uint32_t wavePart[R2P_NT / 64 + 1];
    __shared__ uint32_t changeList[R2P_NT];
    __shared__ uint32_t actL[R2P_ACT];
    __shared__ float actW[R2P_ACT];
    __shared__ int rowCell[R2P_NT];
    __shared__ float rowWeight[R2P_NT];
    __shared__ uint32_t sNChange;
    const R2pTask T = tasks[blockIdx.x];
    const int t = threadIdx.x, lane = t & 63;
    const int L = (int) T.L;
    const uint32_t nRows = T.nRows;
    char *rm = cellsRM + T.cellOff;
    char *cm = cellsCM + T.cmOff;
    const float *gw = globalWeight + T.weightOff;
    float *local = localWeight + T.weightOff;
    uint32_t *active = activeList + T.weightOff;
    int *count = countG + T.scratchOff;
    // the reciprocal table of a sub-alignment is dead before its logarithms are written: in LDS it shares their array
    float *share = L > LDSCOLS ? shareG + T.scratchOff : lgL;
    const bool big = L > LDSCOLS;
    float *sub = big ? subG + T.scratchOff : subL;
    float *lg = big ? lgG + T.scratchOff : lgL;
    float *fq = freq + T.colOff * kRes;
    float *ef = eff + T.colOff;
    // ---- gaps outside a row's first / last residue become end gaps; column-major copy; counters cleared
    for (uint32_t r = t; r < nRows; r += R2P_NT) {
        char *row = rm + (size_t) r * T.stride;
        for (int i = 0; i < L && row[i] == kGapC; i++) row[i] = (char) kEndGapC;
        for (int i = L - 1; i >= 0 && row[i] == kGapC; i--) row[i] = (char) kEndGapC;
        local[r] = 0.0f;
    }
    for (int x = t; x < (L + 1) * kCodes; x += R2P_NT) count[x] = 0;
    __syncthreads();
    for (int i = 0; i < L; i++)
        for (uint32_t r = t; r < nRows; r += R2P_NT) cm[(size_t) i * T.rowStride + r] = rm[(size_t) r * T.stride + i];
    if (t == 0) sPart = 0;
    __syncthreads();
    float prevEff = 0.0f;
    for (int i = 0; i < L; i++) {
        const char *colI = cm + (size_t) i * T.rowStride;
        const char *colP = i ? cm + (size_t) (i - 1) * T.rowStride : nullptr;
        // ---- rows that start / end a residue run here
        if (t == 0) {
            sNChange = 0;
            sChanged = 0;
        }
        __syncthreads();
        for (uint32_t r0 = 0; r0 < nRows; r0 += R2P_NT) {
            const uint32_t r = r0 + t;
            int delta = 0;
            if (r < nRows) {
                const bool here = colI[r] < kAnyC, before = i != 0 && colP[r] < kAnyC;
                if (here && !before) delta = 1;
                else if (i != 0 && before && !here) delta = -1;
            }
            if (delta) {
                const uint32_t at = atomicAdd(&sNChange, 1u);
                if (at < (uint32_t) R2P_NT) changeList[at] = (r << 1) | (delta > 0 ? 1u : 0u);
                atomicAdd(&sPart, delta);
            }
            __syncthreads();
            // (a pass looks at R2P_NT rows, so the list cannot overflow); drain it now
            const uint32_t nC = sNChange;
            for (uint32_t c = 0; c < nC; c++) {
                const uint32_t rr = changeList[c] >> 1;
                const int d = (changeList[c] & 1u) ? 1 : -1;
                const char *row = rm + (size_t) rr * T.stride;
                for (int j = t; j < L; j += R2P_NT) count[j * kCodes + (int) row[j]] += d;
            }
            __syncthreads();
            if (t == 0) {
                if (nC) sChanged = 1;
                sNChange = 0;
            }
            __syncthreads();
        }
        const bool changed = sChanged != 0;
        if (changed) {
            const int participating = sPart;
            // ---- the columns where at most a tenth of the participating rows have an end gap
            if (t == 0) {
                sJmin = L;
                sJmax = -1;
            }
            __syncthreads();
            const float limit = 0.1f * (float) participating;
            {
                int myMin = L, myMax = -1;
                for (int j = t; j < L; j += R2P_NT)
                    if (!((float) count[j * kCodes + kEndGapC] > limit)) {
                        myMin = min(myMin, j);
                        myMax = max(myMax, j);
                    }
                if (myMin < L) atomicMin(&sJmin, myMin);
                if (myMax >= 0) atomicMax(&sJmax, myMax);
            }
            __syncthreads();
            const int jmin = sJmin, jmax = sJmax;
            const int width = jmax - jmin + 1;
            // ---- the active rows, in row order
            if (t == 0) sActive = 0;
            __syncthreads();
            for (uint32_t r0 = 0; r0 < nRows; r0 += R2P_NT) {
                const uint32_t r = r0 + t;
                const bool act = r < nRows && colI[r] < kAnyC;
                const unsigned long long bal = __ballot(act);
                if (lane == 0) wavePart[t >> 6] = (uint32_t) __popcll(bal);
                __syncthreads();
                uint32_t base = (uint32_t) sActive;
                for (int w = 0; w < (t >> 6); w++) base += wavePart[w];
                if (act) active[base + (uint32_t) __popcll(bal & ((1ull << lane) - 1ull))] = r;
                __syncthreads();
                if (t == 0) {
                    uint32_t tot = 0;
                    for (int w = 0; w < R2P_NT / 64; w++) tot += wavePart[w];
                    sActive += (int) tot;
                }
                __syncthreads();
            }
            const uint32_t nActive = (uint32_t) sActive;
            if (width < 20) {
                for (uint32_t r = t; r < nRows; r += R2P_NT) local[r] = colI[r] < kAnyC ? gw[r] : 0.0f;
            } else {
                // share[j][a] = 1 / (distinct[j] * count[j][a]): the host's tabulated reciprocal
                for (int j = jmin + t; j <= jmax; j += R2P_NT) {
                    int d = 0;
                    for (int a = 0; a < kAnyC; a++) d += count[j * kCodes + a] != 0;
                    for (int a = 0; a < kAnyC; a++) {
                        const uint32_t x = (uint32_t) (count[j * kCodes + a] * d);
                        if (x >= rcpN) atomicExch(errFlag, 1);
                        share[j * kCodes + a] = rcpTable[min(x, rcpN - 1)];
                    }
                    for (int a = kAnyC; a < kCodes; a++) share[j * kCodes + a] = 0.0f;
                }
                for (uint32_t r = t; r < nRows; r += R2P_NT) local[r] = 1E-8f;
                __syncthreads();
                for (uint32_t x = t; x < nActive; x += R2P_NT) {
                    const uint32_t r = active[x];
                    float acc = 1E-8f;
                    // sixteen columns' cells and shares are requested before the first is added (the adds stay in column order)
                    int j = jmin;
                    for (; j + 16 <= jmax + 1; j += 16) {
                        int c[16];
                        float sv[16];
#pragma unroll
                        for (int u = 0; u < 16; u++) c[u] = (int) cm[(size_t) (j + u) * T.rowStride + r];
#pragma unroll
                        for (int u = 0; u < 16; u++) sv[u] = share[(j + u) * kCodes + c[u]];
#pragma unroll
                        for (int u = 0; u < 16; u++) acc += sv[u];
                    }
                    for (; j <= jmax; j++) acc += share[j * kCodes + (int) cm[(size_t) j * T.rowStride + r]];
                    local[r] = acc;
                }
            }
            __syncthreads();
            // ---- effective sequences of the sub-alignment: frequencies under the weights (rows in order), entropy.  This pass reads
            // every cell of the sub-alignment -- the bulk of the kernel's time -- so the active rows' indices and weights come from LDS
            const bool staged = nActive <= (uint32_t) R2P_ACT;
            if (staged)
                for (uint32_t x = t; x < nActive; x += R2P_NT) {
                    const uint32_t r = active[x];
                    actL[x] = r;
                    actW[x] = local[r];
                }
            __syncthreads();
            for (int j = jmin + t; j <= jmax; j += R2P_NT) {
                float *sj = big ? ldsF + t * kCodes : sub + j * kCodes;
                for (int a = 0; a < kCodes; a++) sj[a] = 0.0f;
                uint32_t x = 0;
                if (staged) {
                    for (; x + 16 <= nActive; x += 16) {   // sixteen rows' cells in flight; the adds stay in row order
                        int c[16];
#pragma unroll
                        for (int u = 0; u < 16; u++) c[u] = (int) rm[(size_t) actL[x + u] * T.stride + j];
#pragma unroll
                        for (int u = 0; u < 16; u++) sj[c[u]] += actW[x + u];
                    }
                    for (; x < nActive; x++) sj[(int) rm[(size_t) actL[x] * T.stride + j]] += actW[x];
                }
                for (; x + 8 <= nActive; x += 8) {   // eight rows' cells and weights in flight; the adds stay in row order
                    int c[8];
                    float wv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const uint32_t r = active[x + u];
                        c[u] = (int) rm[(size_t) r * T.stride + j];
                        wv[u] = local[r];
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) sj[c[u]] += wv[u];
                }
                for (; x < nActive; x++) {
                    const uint32_t r = active[x];
                    sj[(int) rm[(size_t) r * T.stride + j]] += local[r];
                }
                r2pScale20(sj, nullptr);
                // what the entropy chain reads: the frequency (0 where the reference skips the term: fma(-0, 0, e) = e) and its logarithm
                for (int a = 0; a < kRes; a++) {
                    const bool term = (double) sj[a] > 1E-10;
                    lg[j * kCodes + a] = term ? r2pFlog2(sj[a]) : 0.0f;
                    sub[j * kCodes + a] = term ? sj[a] : 0.0f;
                }
            }
            __syncthreads();
            {
                float e = 0.0f;   // thread 0's running sum, in column order
                for (int j0 = jmin; j0 <= jmax; j0 += LDSCOLS) {
                    const int nj = min(LDSCOLS, jmax + 1 - j0);
                    const float *cs = sub + j0 * kCodes, *cl = lg + j0 * kCodes;
                    if (big) {
                        for (int x = t; x < nj * kRes; x += R2P_NT) {
                            const int jj = x / kRes, a = x - jj * kRes;
                            subL[jj * kCodes + a] = sub[(j0 + jj) * kCodes + a];
                            lgL[jj * kCodes + a] = lg[(j0 + jj) * kCodes + a];
                        }
                        __syncthreads();
                        cs = subL;
                        cl = lgL;
                    }
                    if (t == 0) {
                        for (int j = 0; j < nj; j++) {
                            float sv[kRes], lv[kRes];
#pragma unroll
                            for (int a = 0; a < kRes; a++) {
                                sv[a] = cs[j * kCodes + a];
                                lv[a] = cl[j * kCodes + a];
                            }
#pragma unroll
                            for (int a = 0; a < kRes; a++) e = fmaf(-sv[a], lv[a], e);
                        }
                    }
                    if (big) __syncthreads();   // the staged columns are consumed before the next ones arrive
                }
                if (t == 0) ef[i] = width > 0 ? r2pFpow2(e / (float) width) : 1.0f;
            }
            __syncthreads();
            prevEff = ef[i];
        } else {
            if (t == 0) ef[i] = i == 0 ? 0.0f : prevEff;
        }
        // ---- the column's residue frequencies under the current weights (rows in order)
        {   // R2P_NT rows' cells and weights staged by the workgroup, then one thread per residue adds its rows in order
            float acc = 0.0f;
            for (uint32_t r0 = 0; r0 < nRows; r0 += R2P_NT) {
                const uint32_t n = min((uint32_t) R2P_NT, nRows - r0);
                if ((uint32_t) t < n) {
                    rowCell[t] = (int) colI[r0 + t];
                    rowWeight[t] = local[r0 + t];
                }
                __syncthreads();
                if (t < kRes)
                    for (uint32_t x = 0; x < n; x++)
                        if (rowCell[x] == t) acc += rowWeight[x];
                __syncthreads();
            }
            if (t < kRes) fq[i * kRes + t] = acc;
        }
        __syncthreads();
        if (t == 0) r2pScale20(fq + i * kRes, background);
        __syncthreads();
    }
    if (taskTicks && t == 0) taskTicks[blockIdx.x] = (long long) wall_clock64() - tick0;
}

}  // namespace

// host entry (called by sd_result2profile.cpp): tasks / cells / weights are host arrays of one batch
extern "C" int sdR2pColumnWeightsDevice(sd_ctx *ctx, uint32_t nTasks, const void *tasksHost /* R2pTask[nTasks] */, const char *cells, uint64_t cellBytes,
                             uint64_t cmBytes, const float *globalWeight, uint64_t nWeights, uint64_t nColumns, uint64_t scratchElems,
                             const float *rcpTable, uint32_t rcpN, const double *background, float *freqOut, float *effOut) {
    if (!ctx) return SD_EINVAL;
    if (nTasks == 0) return SD_OK;
    (void) hipSetDevice(ctx->device);
    R2pTask *dTasks = nullptr;
    char *dCells = nullptr, *dCm = nullptr;
    float *dGw = nullptr, *dLocal = nullptr, *dRcp = nullptr, *dShare = nullptr, *dSub = nullptr, *dLg = nullptr, *dFreq = nullptr, *dEff = nullptr;
    uint32_t *dActive = nullptr;
    int *dCount = nullptr, *dErr = nullptr;
    double *dBack = nullptr;
    SD_HIP(ctx, wsGet(ctx, "r2p.tasks", (size_t) nTasks, &dTasks));
    SD_HIP(ctx, wsGet(ctx, "r2p.cells", (size_t) cellBytes + 64, &dCells));
    SD_HIP(ctx, wsGet(ctx, "r2p.cm", (size_t) cmBytes + 64, &dCm));
    SD_HIP(ctx, wsGet(ctx, "r2p.gw", (size_t) nWeights + 1, &dGw));
    SD_HIP(ctx, wsGet(ctx, "r2p.local", (size_t) nWeights + 1, &dLocal));
    SD_HIP(ctx, wsGet(ctx, "r2p.active", (size_t) nWeights + 1, &dActive));
    SD_HIP(ctx, wsGet(ctx, "r2p.rcp", (size_t) rcpN + 1, &dRcp));
    SD_HIP(ctx, wsGet(ctx, "r2p.count", (size_t) scratchElems + 1, &dCount));
    SD_HIP(ctx, wsGet(ctx, "r2p.share", (size_t) scratchElems + 1, &dShare));
    SD_HIP(ctx, wsGet(ctx, "r2p.sub", (size_t) scratchElems + 1, &dSub));
    SD_HIP(ctx, wsGet(ctx, "r2p.lg", (size_t) scratchElems + 1, &dLg));
    SD_HIP(ctx, wsGet(ctx, "r2p.freq", (size_t) nColumns * kRes + 1, &dFreq));
    SD_HIP(ctx, wsGet(ctx, "r2p.eff", (size_t) nColumns + 1, &dEff));
    SD_HIP(ctx, wsGet(ctx, "r2p.err", 1, &dErr));
    SD_HIP(ctx, wsGet(ctx, "r2p.back", 24, &dBack));
    SD_HIP(ctx, hipMemcpyAsync(dTasks, tasksHost, (size_t) nTasks * sizeof(R2pTask), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(dCells, cells, cellBytes, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(dGw, globalWeight, nWeights * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(dRcp, rcpTable, (size_t) rcpN * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(dBack, background, kRes * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dErr, 0, sizeof(int), ctx->stream));
    const bool dbg = getenv("SD_DEBUG_TIMING") != nullptr;
    auto nowMs = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = nowMs();
    if (dbg) SD_HIP(ctx, sdStreamSync(ctx));
    const double t1 = nowMs();
    long long *dTicks = nullptr;
    if (dbg) SD_HIP(ctx, wsGet(ctx, "r2p.ticks", (size_t) nTasks, &dTicks));
    {
        // one launch, the caller's order (longest alignments first: the tail of the batch starts at time 0)
        ProfScope ps(ctx, "r2p_column_weights");
        hipLaunchKernelGGL(r2p_column_weights_kernel<320>, dim3(nTasks), dim3(R2P_NT), 0, ctx->stream, (const R2pTask *) dTasks, dCells, dCm,
                           (const float *) dGw, dLocal, dActive, (const float *) dRcp, rcpN, (const double *) dBack, dCount, dShare, dSub, dLg,
                           dFreq, dEff, dErr, dTicks);
    }
    SD_HIP(ctx, hipGetLastError());
    if (dbg) SD_HIP(ctx, sdStreamSync(ctx));
    const double t2 = nowMs();
    int hErr = 0;
    SD_HIP(ctx, hipMemcpyAsync(freqOut, dFreq, (size_t) nColumns * kRes * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(effOut, dEff, (size_t) nColumns * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(&hErr, dErr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, sdStreamSync(ctx));
    if (dbg) {
        std::vector<long long> ticks(nTasks);
        SD_HIP(ctx, hipMemcpy(ticks.data(), dTicks, (size_t) nTasks * sizeof(long long), hipMemcpyDeviceToHost));
        const R2pTask *th = (const R2pTask *) tasksHost;
        std::vector<uint32_t> o(nTasks);
        for (uint32_t k = 0; k < nTasks; k++) o[k] = k;
        std::sort(o.begin(), o.end(), [&](uint32_t x, uint32_t y) { return ticks[x] > ticks[y]; });
        double sum = 0;
        for (uint32_t k = 0; k < nTasks; k++) sum += (double) ticks[k] * 1e-5;
        fprintf(stderr, "[r2p device] workgroup time: sum %.0f ms; slowest", sum);
        for (uint32_t k = 0; k < std::min(nTasks, 5u); k++)
            fprintf(stderr, " %.1f ms (L %u, %u rows, launch slot %u)", (double) ticks[o[k]] * 1e-5, th[o[k]].L, th[o[k]].nRows, o[k]);
        fprintf(stderr, "\n");
    }
    if (dbg)
        fprintf(stderr, "[r2p device] %u tasks: upload %.1f ms, kernel %.1f ms, download %.1f ms\n", nTasks, t1 - t0, t2 - t1,
                nowMs() - t2);
    if (hErr) return sdFail(ctx, SD_EHIP, "result2profile: a (count x distinct) product outside the reciprocal table");
    return SD_OK;
}
