// Smith-Waterman on gfx950: score/end pass, start-position (reverse) pass and banded traceback.
//
// Semantics are those of the reference's striped SSE/AVX2 kernels, restated in scalar form in
// oracle/sd_oracle.cpp (swPass) and derived in DESIGN.md:
//     Hpre(q) = max(0, G[c-1][q-1] + P, E[q], Fl)        Fl: vertical gap opened inside the same SIMD segment
//     G(q)    = max(Hpre(q), Ff)                          Ff: exact vertical gap (what the lazy-F loop produces)
//     E'(q)   = max(E[q]-ge, Hpre(q)-go, 0)               E is opened from Hpre, NOT from G
//     Fl'     = max(Fl-ge, Hpre(q)-go, 0)  (reset to 0 where q % segLen == 0),  Ff' = max(Ff-ge, G(q)-go, 0)
// (M/src/alignment/StripedSmithWaterman.cpp:713-871 byte kernel, :1013-1149 word kernel).
//
// Mapping to CDNA4: integer DP, no MFMA.  A task (one pair, one pass) is owned by a 32-lane half
// wavefront; lane l holds RT consecutive query rows in VGPRs and the half-wave sweeps the target as a
// systolic array (lane l works on column k-l at step k), handing (G, Ff, Fl, residue) of its last row to
// lane l+1 each step.  The per-task query profile (int8 [21][32*RT]) lives in LDS and is the only
// indexed operand; every lane reads only its own RT bytes of the row selected by its current target
// residue.  HBM traffic is the two residue streams (algorithmic bytes qLen + tLen per task), so the
// kernel is VALU bound; bench.py reports it in GCUPS.
#include "sd_common.h"
#include <unistd.h>


#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <omp.h>
#include <numeric>

#include "../host/sd_host.h"
#include "sd_sw_pk.h"

namespace {

using sdpk::SwTask;

template <int RT>
__global__ void __launch_bounds__(64)
sw_score_kernel(const SwTask *__restrict__ tasks, uint32_t nTasks, const uint8_t *__restrict__ qRes,
                const int8_t *__restrict__ qBias, const uint8_t *__restrict__ tRes, const int8_t *__restrict__ mat,
                int go, int ge, int32_t *__restrict__ out, uint2 *__restrict__ boundary,
                const uint32_t *__restrict__ order /* nullable: task = tasks[order[x]] */,
                const int8_t *__restrict__ qProf /* nullable: profile queries, int8 [position][21] */) {
    constexpr int ROWS = 32 * RT;          // rows per strip
    constexpr int WORDS = RT / 4;          // profile dwords per lane per residue
    __shared__ uint32_t prof[2][21][ROWS / 4];
    __shared__ int8_t smat[441];
    for (int i = threadIdx.x; i < 441; i += 64) smat[i] = mat[i];
    __syncthreads();

    const int grp = threadIdx.x >> 5;
    const int l = threadIdx.x & 31;
    const uint32_t taskId = blockIdx.x * 2 + grp;
    SwTask tk;
    if (taskId < nTasks) {
        tk = tasks[order ? order[taskId] : taskId];
    } else {
        tk.n = 0; tk.tL = 0; tk.qOff = 0; tk.tOff = 0; tk.qStep = 1; tk.tStep = 1; tk.segLen = 1; tk.slot = 0; tk.boundOff = 0;
    }
    const int n = tk.n, tL = tk.tL;
    const int nStrips = (n + ROWS - 1) / ROWS;

    int bestVal = 0, bestCol = -1, bestRow = 0;

    for (int strip = 0; strip < nStrips; strip++) {
        const int q0 = strip * ROWS + l * RT;
        // ---- build this lane's slice of the query profile (SmithWaterman::createQueryProfile, :163-187)
        uint32_t segMask = 0;
        {
            int8_t *pb = (int8_t *) &prof[grp][0][0];
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const int qi = q0 + r;
                const bool valid = qi < n;
                int res = 20, cb = 0;
                int64_t pidx = 0;
                if (valid) {
                    const int64_t idx = (int64_t) tk.qOff + (int64_t) qi * tk.qStep;
                    res = qRes[idx];
                    cb = qBias[idx];
                    pidx = idx * 21;
                }
                if (valid && (qi % tk.segLen) == 0) segMask |= (1u << r);
                if (qProf) {
#pragma unroll
                    for (int a = 0; a < 21; a++) pb[a * ROWS + l * RT + r] = (int8_t) (valid ? (int) qProf[pidx + a] : -64);
                } else {
#pragma unroll
                    for (int a = 0; a < 21; a++) {
                        int v = valid ? (int) smat[a * 21 + res] + cb : -64;
                        pb[a * ROWS + l * RT + r] = (int8_t) v;
                    }
                }
            }
        }
        int H[RT], E[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) {
            H[r] = 0;
            E[r] = 0;
        }
        int outG = 0, outFf = 0, outFl = 0, curT = 20, prevInG = 0;
        const bool firstStrip = (strip == 0);
        const bool lastStrip = (strip == nStrips - 1);
        uint2 *bnd = boundary + tk.boundOff;
        if (!firstStrip) __threadfence();   // boundary values of the previous strip must be visible
        const int steps = (n > 0 && tL > 0) ? tL + 31 : 0;
        for (int k = 0; k < steps; k++) {
            const int c = k - l;
            int inG = __shfl_up(outG, 1, 32);
            int inFf = __shfl_up(outFf, 1, 32);
            int inFl = __shfl_up(outFl, 1, 32);
            int inT = __shfl_up(curT, 1, 32);
            const bool active = (c >= 0) && (c < tL);
            if (l == 0) {
                inG = 0; inFf = 0; inFl = 0;
                if (active) {
                    inT = tRes[(int64_t) tk.tOff + (int64_t) c * tk.tStep];
                    if (!firstStrip) {
                        unsigned long long w = __hip_atomic_load((unsigned long long *) &bnd[c], __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_AGENT);
                        inG = (int) (w & 0xFFFFu);
                        inFf = (int) ((w >> 16) & 0xFFFFu);
                        inFl = (int) ((w >> 32) & 0xFFFFu);
                    }
                }
            }
            if (active) {
                curT = inT;
                int diag = (c == 0) ? 0 : prevInG;
                prevInG = inG;
                int Fl = inFl, Ff = inFf;
                uint32_t pw[WORDS];
#pragma unroll
                for (int w = 0; w < WORDS; w++) pw[w] = prof[grp][curT][l * WORDS + w];
                int cm = 0;
#pragma unroll
                for (int r = 0; r < RT; r++) {
                    const int s = (int) (int8_t) (pw[r >> 2] >> (8 * (r & 3)));
                    if (segMask & (1u << r)) Fl = 0;
                    const int old = H[r];
                    const int h = min(diag + s, 32767);   // simdi16_adds: the word kernel's H saturates (sw_sse2_word, :1069)
                    const int hpre = max(max(h, E[r]), Fl);
                    const int g = max(hpre, Ff);
                    const int open = hpre - go;
                    E[r] = max(max(E[r] - ge, open), 0);
                    Fl = max(max(Fl - ge, open), 0);
                    Ff = max(max(Ff - ge, g - go), 0);
                    H[r] = g;
                    diag = old;
                    cm = max(cm, (g << 5) | (31 - r));
                }
                outG = H[RT - 1];
                outFf = Ff;
                outFl = Fl;
                if (l == 31 && !lastStrip) {
                    unsigned long long w = (unsigned long long) (uint32_t) outG | ((unsigned long long) (uint32_t) outFf << 16) |
                                           ((unsigned long long) (uint32_t) outFl << 32);
                    __hip_atomic_store((unsigned long long *) &bnd[c], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int colVal = cm >> 5;
                if (colVal > bestVal) {
                    bestVal = colVal;
                    bestCol = c;
                    bestRow = q0 + 31 - (cm & 31);
                }
            }
        }
    }
    // ---- reduce over the 32 lanes: max value, then smallest column, then smallest row
    unsigned long long key = ((unsigned long long) (uint32_t) bestVal << 40) |
                             ((unsigned long long) (uint32_t) (0xFFFFF - (bestCol < 0 ? 0xFFFFF : bestCol)) << 20) |
                             (unsigned long long) (uint32_t) (0xFFFFF - bestRow);
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(key, off, 32);
        key = o > key ? o : key;
    }
    if (l == 0 && taskId < nTasks) {
        const int v = (int) (key >> 40);
        const int col = 0xFFFFF - (int) ((key >> 20) & 0xFFFFF);
        const int row = 0xFFFFF - (int) (key & 0xFFFFF);
        out[3 * tk.slot + 0] = v;
        out[3 * tk.slot + 1] = (v == 0 || col == 0xFFFFF) ? -1 : col;
        out[3 * tk.slot + 2] = (v == 0) ? n - 1 : row;
    }
}

// ---------------------------------------------------------------------------------------------
// Banded traceback, one thread per alignment: SmithWaterman::banded_sw (StripedSmithWaterman.cpp:1348-1600)
// with its band-coordinate bookkeeping kept literally (what the band-edge cells hold decides ties).
// One launch evaluates one band width; the host doubles the band of the tasks whose maximum has not
// reached the alignment score (the reference's do/while, :1492-1493).
// ---------------------------------------------------------------------------------------------
struct TbTask {
    uint64_t qAbs;      // absolute index of query[qStart]
    uint64_t tAbs;      // absolute index of target[tStart]
    int32_t qLen, tLen; // sub-rectangle
    int32_t score;
    int32_t band;
    int32_t maxv;       // carried over band doublings
    uint32_t slot;
    uint64_t intOff;    // scratch: 3*(width+1) ints
    uint64_t dirOff;    // scratch: width_d*qLen*3 bytes
    uint64_t btOff;     // output (reversed while walking, fixed up at the end)
};

// Wave-parallel form: a 32-lane half wavefront owns one alignment.  The three band arrays (h_b, e_b, h_c)
// live in LDS and are indexed exactly like the reference's; a row is processed 32 band cells at a time.
// The only loop-carried dependence inside a row is the horizontal gap f[j] = max(h_c[j-1]-go, f[j-1]-ge);
// with T = max(e1, diag) >= 0 it reduces to f[p] = max(-ge, max_{k<p}(T[k]-go+ge*(k+1))) - ge*p, i.e. an
// exclusive prefix maximum across lanes (cross-lane scan, carried between 32-cell chunks).  Directions are
// packed to one byte per cell (bit0 dirE==3, bit1 dirF==5, bits2-3: 0 diag / 1 take dirE / 2 take dirF) and
// written row-major (coalesced); lane 0 then walks the path.
// GLOBAL: bands beyond the LDS classes (2*band+3 > 2047: a gap of > 1 000 residues inside the alignment) keep the three
// band arrays in global scratch in front of the task's direction bytes; the lanes of the half wavefront exchange them
// through agent-scope accesses with a fence where the LDS version has a wave barrier.  Rare and slow by design --
// what matters is that the reference's do/while (band doubling, :1492-1493) has no width the device refuses.
template <bool GLOBAL> __device__ __forceinline__ int bandLd(const int32_t *p) {
    if constexpr (GLOBAL) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <bool GLOBAL> __device__ __forceinline__ void bandSt(int32_t *p, int v) {
    if constexpr (GLOBAL) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool GLOBAL> __device__ __forceinline__ void bandSync() {
    if constexpr (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __builtin_amdgcn_wave_barrier();
}
__host__ __device__ __forceinline__ uint64_t tbGlobalIntBytes(int band) { return ((uint64_t) 3 * (2 * (uint64_t) band + 4) * 4 + 15) & ~15ull; }

// DPP moves with compile-time control words (the traceback kernels' cross-lane traffic)
__device__ __forceinline__ int dppI(int oldv, int src, int ctrl, int rowMask) {
    switch (ctrl) {   // the control word must be a compile-time constant
        case 0x138: return __builtin_amdgcn_update_dpp(oldv, src, 0x138, 0xf, 0xf, false);   // wave_shr:1
        case 0x130: return __builtin_amdgcn_update_dpp(oldv, src, 0x130, 0xf, 0xf, false);   // wave_shl:1
        case 0x111: return __builtin_amdgcn_update_dpp(oldv, src, 0x111, 0xf, 0xf, false);   // row_shr:1
        case 0x112: return __builtin_amdgcn_update_dpp(oldv, src, 0x112, 0xf, 0xf, false);
        case 0x114: return __builtin_amdgcn_update_dpp(oldv, src, 0x114, 0xf, 0xf, false);
        case 0x118: return __builtin_amdgcn_update_dpp(oldv, src, 0x118, 0xf, 0xf, false);
        default: return __builtin_amdgcn_update_dpp(oldv, src, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    }
    (void) rowMask;
}

__device__ __forceinline__ int dppZ(int src, int ctrl) {   // lanes without a source read 0
    switch (ctrl) {
        case 0x138: return __builtin_amdgcn_update_dpp(0, src, 0x138, 0xf, 0xf, true);
        case 0x130: return __builtin_amdgcn_update_dpp(0, src, 0x130, 0xf, 0xf, true);
        case 0x111: return __builtin_amdgcn_update_dpp(0, src, 0x111, 0xf, 0xf, true);
        case 0x112: return __builtin_amdgcn_update_dpp(0, src, 0x112, 0xf, 0xf, true);
        case 0x114: return __builtin_amdgcn_update_dpp(0, src, 0x114, 0xf, 0xf, true);
        default: return __builtin_amdgcn_update_dpp(0, src, 0x118, 0xf, 0xf, true);
    }
}

template <bool PROF, bool GLOBAL = false>
__global__ void __launch_bounds__(64)
sw_traceback_kernel(TbTask *__restrict__ tasks, uint32_t nTasks, const uint8_t *__restrict__ qRes,
                    const int8_t *__restrict__ qBias, const uint8_t *__restrict__ tRes, const int8_t *__restrict__ mat,
                    int go, int ge, int ldsStride /* ints per array */, int8_t *__restrict__ dirs, char *__restrict__ bt,
                    int32_t *__restrict__ res /* per task: btLen (-2 = band too small, -1 = traceback error), identical */,
                    const uint32_t *__restrict__ order /* nullable */,
                    int maxWidthLoop /* 0: one attempt; else keep doubling the band inside the kernel while
                                        2*band+3 <= maxWidthLoop (scratch must be sized for that width) */,
                    const int8_t *__restrict__ qProf /* nullable: profile queries, int8 [position][21] */) {
    extern __shared__ int32_t lds[];
    __shared__ int8_t smat[441];
    for (int i = threadIdx.x; i < 441; i += 64) smat[i] = mat[i];
    __syncthreads();
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const uint32_t lid = blockIdx.x * 2 + grp;
    const bool have = lid < nTasks;
    const uint32_t id = have ? (order ? order[lid] : lid) : 0;
    TbTask tk;
    if (have) tk = tasks[id];
    else { tk.qLen = 0; tk.tLen = 0; tk.band = 1; tk.score = 0; tk.maxv = 0; tk.qAbs = 0; tk.tAbs = 0; tk.slot = 0; tk.dirOff = 0; tk.btOff = 0; }
    const int qLen = tk.qLen, tLen = tk.tLen;
    int band = tk.band;
    int32_t *h_b, *e_b, *h_c;
    int8_t *direction;
    if constexpr (GLOBAL) {   // one attempt at the task's band per launch: [3 x (width + 1) ints][direction bytes]
        const int stride = band * 2 + 4;
        h_b = reinterpret_cast<int32_t *>(dirs + tk.dirOff);
        e_b = h_b + stride;
        h_c = e_b + stride;
        direction = dirs + tk.dirOff + tbGlobalIntBytes(band);
    } else {
        h_b = lds + (size_t) grp * 3 * ldsStride;
        e_b = h_b + ldsStride;
        h_c = e_b + ldsStride;
        direction = dirs + tk.dirOff;
    }
    const uint8_t *q = qRes + tk.qAbs;
    const int8_t *cb = qBias + tk.qAbs;
    const uint8_t *t = tRes + tk.tAbs;
    int maxv = tk.maxv;
    int width, width_d;
  for (;;) {
    width = band * 2 + 3;
    width_d = band * 2 + 1;
    // (the idle half of an odd last wavefront owns no scratch: in the global class its stores would land in another task's arrays)
    for (int x = l; (have || !GLOBAL) && x <= width && x < ldsStride; x += 32) { bandSt<GLOBAL>(h_b + x, 0); bandSt<GLOBAL>(e_b + x, 0); bandSt<GLOBAL>(h_c + x, 0); }
    bandSync<GLOBAL>();
    // Operands of the substitution score two rows ahead, the score itself one row ahead: neither the global loads (query
    // letter, bias, target letter of the row's first 32 cells) nor the matrix / profile lookup sit on a row's critical path.
    auto rowClamp = [&](int r) -> int { return r < qLen ? r : qLen - 1; };
    auto loadT = [&](int r) -> int {   // target letter of cell l of row r's first chunk
        int jn = ((r - band) > 0 ? (r - band) : 0) + l;
        jn = jn >= tLen ? tLen - 1 : jn;
        jn = jn < 0 ? 0 : jn;
        return (int) t[jn];
    };
    auto scoreOf = [&](int r, int qv, int tv) -> int {   // (without the bias: an add right behind the read would wait for it at the row's top)
        if (PROF) return (int) qProf[(tk.qAbs + (uint64_t) r) * 21 + tv];
        return (int) smat[21 * qv + tv];
    };
    int qa = 0, cba = 0, qb = 0, cbb = 0, tb = 0, sNext = 0;
    if (have && qLen > 0 && tLen > 0) {
        qa = q[0];
        cba = PROF ? 0 : cb[0];
        sNext = scoreOf(0, qa, loadT(0));
        const int r1 = rowClamp(1);
        qb = q[r1];
        cbb = PROF ? 0 : cb[r1];
        tb = loadT(r1);
    }
    for (int i = 0; i < qLen; i++) {
        int beg = 0, end = tLen - 1;
        int jj = i - band; beg = beg > jj ? beg : jj;
        jj = i + band; end = end < jj ? end : jj;
        const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
        if (l < 5) {   // h_b[0], e_b[0], h_b[edge], e_b[edge], h_c[0] = 0: one store of five lanes
            int32_t *z = l == 0 ? h_b : (l == 1 ? e_b : (l == 2 ? h_b + edge : (l == 3 ? e_b + edge : h_c)));
            bandSt<GLOBAL>(z, 0);
        }
        bandSync<GLOBAL>();
        const int sCur = sNext;
        const int r1 = rowClamp(i + 1), r2 = rowClamp(i + 2);
        sNext = scoreOf(r1, qb, tb);
        const int qc = q[r2], cbc = PROF ? 0 : (int) cb[r2], tc = loadT(r2);
        const int xi = (i - band) > 0 ? (i - band) : 0;
        const int xim = (i - 1 - band) > 0 ? (i - 1 - band) : 0;
        const int W = end - beg + 1;
        const int8_t *mrow = PROF ? qProf + (tk.qAbs + (uint64_t) i) * 21 : smat + 21 * qa;   // never a mixed (flat) pointer
        const int cbi = cba;
        int8_t *dl = direction + (long long) width_d * i;
        int carry = -ge, prevHc = 0, prevF = 0, uLast = 0;
        for (int p0 = 0; p0 < W; p0 += 32) {
            const int p = p0 + l;
            const bool valid = p < W;
            const int j = beg + p;
            const int u = j - xi + 1;
            int T = 0, eNew = 0, e1 = 0, diag = 0;
            bool dirE = false;
            if (valid) {
                const int e = j - xim + 1, d = (j - 1) - xim + 1;
                int t1 = i == 0 ? -go : bandLd<GLOBAL>(h_b + e) - go;
                int t2 = i == 0 ? -ge : bandLd<GLOBAL>(e_b + e) - ge;
                eNew = t1 > t2 ? t1 : t2;
                dirE = t1 > t2;
                e1 = eNew > 0 ? eNew : 0;
                const int sc = (p0 == 0 ? sCur : (int) mrow[t[j]]) + cbi;
                diag = bandLd<GLOBAL>(h_b + d) + sc;
                T = e1 > diag ? e1 : diag;
            }
            // exclusive prefix maximum of S = T - go + ge * (p + 1) over the 32 lanes by DPP moves (the half wavefront is two
            // rows of 16), carried as S + go + 1 >= 1 so that the 0 a zero-filled move delivers is the identity
            int incl = valid ? T + ge * (p + 1) + 1 : 0, o;
            o = dppZ(incl, 0x111); incl = incl > o ? incl : o;
            o = dppZ(incl, 0x112); incl = incl > o ? incl : o;
            o = dppZ(incl, 0x114); incl = incl > o ? incl : o;
            o = dppZ(incl, 0x118); incl = incl > o ? incl : o;
            o = dppI(0, incl, 0x142, 0xa); incl = incl > o ? incl : o;
            int excl = dppZ(incl, 0x138);
            excl = l == 0 ? 0 : excl;
            const int exclS = excl - go - 1;   // (no predecessor: -go - 1 < -ge <= carry)
            int g = carry > exclS ? carry : exclS;
            const int f = g - ge * p;
            const int hcv = T > f ? T : f;
            int hcP = dppZ(hcv, 0x138), fP = dppZ(f, 0x138);
            if (l == 0) { hcP = prevHc; fP = prevF; }
            if (valid) {
                const bool dirF = (hcP - go) > (fP - ge);
                const int f1 = f > 0 ? f : 0;
                const int tmp1 = e1 > f1 ? e1 : f1;
                int code = (dirE ? 1 : 0) | (dirF ? 2 : 0);
                if (!(tmp1 <= diag)) code |= (e1 > f1) ? 4 : 8;
                dl[j - xi] = (int8_t) code;
                bandSt<GLOBAL>(e_b + u, eNew);
                bandSt<GLOBAL>(h_c + u, hcv);
                maxv = hcv > maxv ? hcv : maxv;
            }
            // carries to the next chunk: lane 31 of this half (a full chunk whenever there is a next one)
            if (p0 + 32 < W) {
                const int cmA = __builtin_amdgcn_readlane(incl, 31), cmB = __builtin_amdgcn_readlane(incl, 63);
                const int hcA = __builtin_amdgcn_readlane(hcv, 31), hcB = __builtin_amdgcn_readlane(hcv, 63);
                const int fA = __builtin_amdgcn_readlane(f, 31), fB = __builtin_amdgcn_readlane(f, 63);
                const int cm = (grp ? cmB : cmA) - go - 1;
                carry = carry > cm ? carry : cm;
                prevHc = grp ? hcB : hcA;
                prevF = grp ? fB : fA;
            }
            const int lastValid = (W - p0) < 32 ? (W - p0 - 1) : 31;
            uLast = (beg + p0 + lastValid) - xi + 1;
            if constexpr (!GLOBAL) __builtin_amdgcn_wave_barrier();   // (GLOBAL: a lane only re-reads what other lanes wrote after the row's fence)
        }
        bandSync<GLOBAL>();
        // the reference copies h_c[1 .. uLast] to h_b here (:1477-1478).  The arrays are exchanged instead: the next row reads h_b at
        // indices 0 (zeroed above), 1 .. uLast (this row's values in either form) and, where its window grew by a column, uLast + 1 --
        // which is exactly the `edge` index it has just zeroed (end + 1 while the window still starts at column 0, width - 1
        // afterwards) -- so what the other entries hold never reaches a cell
        if (W > 0) {
            int32_t *sw = h_b;
            h_b = h_c;
            h_c = sw;
        }
        (void) uLast;
        qa = qb;
        cba = cbb;
        qb = qc;
        cbb = cbc;
        tb = tc;
    }
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        int o = __shfl_xor(maxv, off, 32);
        maxv = o > maxv ? o : maxv;
    }
    if (maxv >= tk.score) break;
    band *= 2;   // the reference's do/while doubles the band (:1492-1493)
    if (band * 2 + 3 > maxWidthLoop) {
        if (have && l == 0) {
            tasks[id].maxv = maxv;
            tasks[id].band = band;
            res[2 * id] = -2;
        }
        return;
    }
  }
    if (!have || l != 0) return;
    tasks[id].maxv = maxv;
    tasks[id].band = band;
    res[2 * id] = -3;   // band found, directions written: sw_traceback_walk_kernel walks the path
}

// The walk of the LDS-band / global-band kernel's tasks: inside sw_traceback_kernel it was lane 0 of a half wavefront chasing
// ~qLen + tLen dependent loads with 31 lanes parked.  Here 16 lanes own a task (four tasks per wavefront): while the walk is in
// the match state the lanes look at the next 16 cells down the diagonal at once (16 independent loads instead of 16 dependent
// ones), the leading run of cells whose code says "diagonal" is taken in one step -- a coalesced store of 'M's, identities by
// popcount -- and the first cell that is anything else goes through the one-cell step, executed by all 16 lanes alike (same
// address: one request) so that the walk state stays uniform over the group.
// traceback (:1498-1558) + expansion / identity count (computerBacktrace, :548-581)
constexpr int TBW_LANES = 16;
template <bool GLOBAL>
__global__ void __launch_bounds__(256)
sw_traceback_walk_kernel(const TbTask *__restrict__ tasks, uint32_t nTasks, const uint8_t *__restrict__ qRes,
                         const uint8_t *__restrict__ tRes, const int8_t *__restrict__ dirs, char *__restrict__ bt,
                         int32_t *__restrict__ res, const uint32_t *__restrict__ order /* nullable */) {
    const uint32_t lid = (blockIdx.x * blockDim.x + threadIdx.x) / TBW_LANES;
    const int l = threadIdx.x & (TBW_LANES - 1), sh = threadIdx.x & 63 & ~(TBW_LANES - 1);
    uint32_t id = 0;
    bool have = lid < nTasks;
    if (have) {
        id = order ? order[lid] : lid;
        have = res[2 * id] == -3;
    }
    TbTask tk;
    if (have) tk = tasks[id];
    else { tk.qLen = 1; tk.tLen = 1; tk.band = 1; tk.qAbs = 0; tk.tAbs = 0; tk.dirOff = 0; tk.btOff = 0; }
    const int qLen = tk.qLen, tLen = tk.tLen, band = tk.band;
    const int width_d = band * 2 + 1;
    const uint8_t *q = qRes + tk.qAbs;
    const uint8_t *t = tRes + tk.tAbs;
    const int8_t *direction = dirs + tk.dirOff + (GLOBAL ? tbGlobalIntBytes(band) : 0);
    // the path is walked from its end, so the characters are written from the end of the task's region backwards:
    // the finished backtrace is the last `len` bytes of [btOff, btOff + qLen + tLen + 2)
    int i = qLen - 1, j = tLen - 1, state = 2;
    char *o = bt + tk.btOff + (qLen + tLen + 2);
    int len = 0, ids = 0;
    bool bad = false;
    bool going = have && (i > 0 || j > 0);
    while (__ballot(going)) {
        {
            const int ii = i - l, jj = j - l;
            bool ok = going && state == 2 && (ii > 0 || jj > 0) && ii >= 0 && jj >= 0;
            int x = ii - band;
            x = x > 0 ? x : 0;
            x = jj - x;
            ok = ok && x >= 0 && x < width_d;
            bool same = false;
            if (ok) {
                const int code = direction[(long long) width_d * ii + x];
                ok = (code & 12) == 0;
                same = q[ii] == t[jj];
            }
            const uint32_t okMask = (uint32_t) (__ballot(ok) >> sh) & ((1u << TBW_LANES) - 1u);
            const uint32_t sameMask = (uint32_t) (__ballot(same) >> sh);
            const int run = __builtin_ctz(~okMask);   // (bit 16 of ~okMask is set: run <= 16)
            const uint32_t runMask = (1u << run) - 1u;
            if (l < run) o[-1 - l] = 'M';
            ids += __builtin_popcount(sameMask & runMask);
            i -= run;
            j -= run;
            o -= run;
            len += run;
        }
        if (going && (i > 0 || j > 0)) {
            if (i < 0 || j < 0) bad = true;
            int x = i - band;
            x = x > 0 ? x : 0;
            x = j - x;
            if (x < 0 || x >= width_d) bad = true;
            if (!bad) {
                const int code = direction[(long long) width_d * i + x];
                int dcode;
                const int dE = (code & 1) ? 3 : 2, dF = (code & 2) ? 5 : 4;
                if (state == 0) dcode = dE;
                else if (state == 1) dcode = dF;
                else dcode = (code & 4) ? dE : ((code & 8) ? dF : 1);
                char c;
                switch (dcode) {
                    case 1: ids += (q[i] == t[j]); --i; --j; state = 2; c = 'M'; break;
                    case 2: --i; state = 0; c = 'I'; break;
                    case 3: --i; state = 2; c = 'I'; break;
                    case 4: --j; state = 1; c = 'D'; break;
                    default: --j; state = 2; c = 'D'; break;
                }
                --o;
                if (l == 0) *o = c;
                len++;
            }
        }
        going = going && !bad && (i > 0 || j > 0);
    }
    if (!have || l != 0) return;
    if (bad || i != 0 || j != 0) {
        res[2 * id] = -1;
        return;
    }
    ids += (q[0] == t[0]);
    *--o = 'M';
    len++;
    res[2 * id] = len;
    res[2 * id + 1] = ids;
}

// ---------------------------------------------------------------------------------------------
// Narrow bands (2*band+3 <= 32, i.e. almost every alignment): the same banded_sw arithmetic with the three band
// arrays held in registers -- lane p of a 32-lane half wavefront owns array index p -- so a row costs a few DPP
// moves instead of LDS round trips: h_b[e], e_b[e] and h_b[d] are the lane itself or a neighbour depending on
// whether the band window moved (xi != xim), the horizontal-gap prefix maximum is a DPP scan, and the band is
// doubled inside the kernel while it still fits.  The query slice, its bias, the target slice and the 4-bit
// direction codes (16 bytes per row) live in LDS; lane 0 walks the path out of LDS.
// qCap / tCap: LDS capacity per task (rows, columns) of this launch class.
// ---------------------------------------------------------------------------------------------
template <bool PROF>
__global__ void __launch_bounds__(64)
sw_traceback_narrow_kernel(TbTask *__restrict__ tasks, uint32_t nTasks, const uint8_t *__restrict__ qRes,
                           const int8_t *__restrict__ qBias, const uint8_t *__restrict__ tRes, const int8_t *__restrict__ mat,
                           int go, int ge, int qCap, int tCap, char *__restrict__ bt, int32_t *__restrict__ res,
                           const uint32_t *__restrict__ order, const int8_t *__restrict__ qProf /* nullable: profile queries */) {
    extern __shared__ uint8_t ldsRaw[];
    __shared__ int8_t smat[441];
    for (int i = threadIdx.x; i < 441; i += 64) smat[i] = mat[i];
    const int grp = threadIdx.x >> 5, l = threadIdx.x & 31;
    const uint32_t lid = blockIdx.x * 2 + grp;
    const bool have = lid < nTasks;
    const uint32_t id = have ? (order ? order[lid] : lid) : 0;
    TbTask tk;
    if (have) tk = tasks[id];
    else { tk.qLen = 0; tk.tLen = 0; tk.band = 1; tk.score = 0; tk.maxv = 0; tk.qAbs = 0; tk.tAbs = 0; tk.slot = 0; tk.dirOff = 0; tk.btOff = 0; }
    const int qLen = tk.qLen, tLen = tk.tLen;
    const size_t perTask = (size_t) 16 * qCap + 2 * (size_t) qCap + tCap;
    uint8_t *dirs = ldsRaw + grp * perTask;           // 16 bytes (32 nibbles) per row
    uint8_t *sq = dirs + (size_t) 16 * qCap;          // query residues
    int8_t *scb = (int8_t *) (sq + qCap);             // query bias
    uint8_t *st = (uint8_t *) (scb + qCap);           // target residues
    for (int x = l; x < qLen; x += 32) {
        sq[x] = qRes[tk.qAbs + x];
        scb[x] = qBias[tk.qAbs + x];
    }
    for (int x = l; x < tLen; x += 32) st[x] = tRes[tk.tAbs + x];
    __syncthreads();

    const int NEG = -(1 << 28);
    int band = tk.band, maxv = tk.maxv;
    bool reached = false;
    // both half wavefronts run the same number of attempts / rows (uniform control flow for the DPP moves)
    for (;;) {
        const int width = band * 2 + 3;
        const bool fits = width <= 32;
        int hb = 0, eb = 0;
        const int rows = (fits && !reached) ? qLen : 0;
        const int rowsOther = __shfl_xor(rows, 32, 64);
        const int nRows = max(rows, rowsOther);
        // substitution score of this lane's cell: its operands (query letter, bias, target letter) are read two rows ahead and
        // the matrix / profile entry one row ahead, so that neither of the two dependent LDS round trips is waited for inside a
        // row (a wavefront issues in order: a wait in front of the first use stalls the recurrence behind it as well)
        // (no branch around the reads -- a half without rows reads its first entries: every address stays inside the task's LDS
        // slice / the profile -- so that the only waits are in front of the uses)
        auto cellOps = [&](int row, int &qv, int &tv, int &cbv) {
            int r = row < qLen ? row : qLen - 1;
            r = r > 0 ? r : 0;
            const int x0 = (r - band) > 0 ? (r - band) : 0;
            int jn = x0 + l - 1;
            jn = jn >= tLen ? tLen - 1 : jn;
            jn = jn > 0 ? jn : 0;
            tv = st[jn];
            qv = 0;
            cbv = 0;
            if (!PROF) {
                qv = sq[r];
                cbv = scb[r];
            }
        };
        // (the matrix entry and the bias stay separate until the cell uses them: an add right behind the read would put the wait for
        // the read at the top of the row)
        auto cellLook = [&](int row, int qv, int tv) -> int {
            int r = row < qLen ? row : qLen - 1;
            r = r > 0 ? r : 0;
            if (PROF) return (int) qProf[(tk.qAbs + (uint64_t) r) * 21 + (tv < 21 ? tv : 20)];
            return (int) smat[21 * qv + tv];   // (an idle half may form any index from its unwritten slice: LDS, no cell takes the value)
        };
        int qb, tb, cbb;
        cellOps(0, qb, tb, cbb);
        int sNext = cellLook(0, qb, tb), cNext = cbb;
        cellOps(1, qb, tb, cbb);
        for (int i = 0; i < nRows; i++) {
            const bool live = i < rows;
            const int sCur = sNext, cCur = cNext;
            sNext = cellLook(i + 1, qb, tb);
            cNext = cbb;
            cellOps(i + 2, qb, tb, cbb);
            const int xi = (i - band) > 0 ? (i - band) : 0;
            const int delta = (i - band) >= 1 ? 1 : 0;     // xi - xim
            int end = tLen - 1;
            end = end < i + band ? end : i + band;
            const int edge = end + 1 < width - 1 ? end + 1 : width - 1;
            const int W = end - xi + 1;
            const bool zeroed = live && (l == 0 || l == edge);
            hb = zeroed ? 0 : hb;
            eb = zeroed ? 0 : eb;
            // previous-row values at index e = u + delta and d = e - 1 (zero-filled DPP moves, no exec masking: every
            // lane computes, `valid` selects what is kept)
            const int hbUp = dppZ(hb, 0x130), ebUp = dppZ(eb, 0x130), hbDn = dppZ(hb, 0x138);
            const int hE = delta ? hbUp : hb, eE = delta ? ebUp : eb, hD = delta ? hb : hbDn;
            const int u = l;
            const bool valid = live && u >= 1 && u <= W;
            const int t1 = i == 0 ? -go : hE - go;
            const int t2 = i == 0 ? -ge : eE - ge;
            const int eNew = t1 > t2 ? t1 : t2;
            const bool dirE = t1 > t2;
            const int e1 = eNew > 0 ? eNew : 0;
            const int diag = hD + sCur + cCur;
            const int T = e1 > diag ? e1 : diag;
            // prefix maximum of S = T - go + ge*u, carried as S + go + 1 >= 1 so that 0 (what a zero-filled DPP move
            // delivers where there is no source lane) is the identity
            int incl = valid ? T + ge * u + 1 : 0, o;
            o = dppZ(incl, 0x111); incl = incl > o ? incl : o;
            o = dppZ(incl, 0x112); incl = incl > o ? incl : o;
            o = dppZ(incl, 0x114); incl = incl > o ? incl : o;
            o = dppZ(incl, 0x118); incl = incl > o ? incl : o;
            o = dppI(0, incl, 0x142, 0xa); incl = incl > o ? incl : o;
            int excl = dppZ(incl, 0x138);
            excl = l == 0 ? 0 : excl;
            const int gx = excl - go - 1;
            const int g = (-ge) > gx ? (-ge) : gx;
            const int f = g - ge * (u - 1);
            const int hcv = T > f ? T : f;
            int hcP = dppZ(hcv, 0x138), fP = dppZ(f, 0x138);
            hcP = u <= 1 ? 0 : hcP;
            fP = u <= 1 ? 0 : fP;
            const bool dirF = (hcP - go) > (fP - ge);
            const int f1 = f > 0 ? f : 0;
            const int tmp1 = e1 > f1 ? e1 : f1;
            int code = (dirE ? 1 : 0) | (dirF ? 2 : 0) | ((tmp1 > diag) ? ((e1 > f1) ? 4 : 8) : 0);
            code = valid ? code : 0;
            eb = valid ? eNew : eb;
            hb = valid ? hcv : hb;
            maxv = (valid && hcv > maxv) ? hcv : maxv;
            // two cells per byte: cell x = u - 1; odd u carries the low nibble and fetches its right neighbour's
            const int codeUp = dppZ(code, 0x130);
            if (valid && (u & 1)) dirs[(size_t) 16 * i + ((u - 1) >> 1)] = (uint8_t) (code | (codeUp << 4));
        }
        if (!reached && fits) {
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                const int o2 = __shfl_xor(maxv, off, 32);
                maxv = o2 > maxv ? o2 : maxv;
            }
            if (maxv >= tk.score) reached = true;
            else band *= 2;
        }
        const bool done = reached || band * 2 + 3 > 32 || !have;
        const bool doneOther = __shfl_xor(done ? 1 : 0, 32, 64) != 0;
        if (done && doneOther) break;
    }
    __syncthreads();
    if (have && l == 0) {
        tasks[id].maxv = maxv;
        tasks[id].band = band;
        if (!reached) res[2 * id] = -2;
    }
    // traceback (:1498-1558) + expansion / identity count (computerBacktrace, :548-581), written backwards.  The path is mostly
    // runs of diagonal steps: while the walk is in the match state the 32 lanes of the half wavefront look at the next 32 cells
    // down the diagonal at once, the leading run of cells whose code says "diagonal" is taken in one step (a coalesced store
    // of 'M's, identities by popcount), and the first cell that is anything else goes through the one-cell step below, which
    // every lane of the half executes redundantly (same LDS words: broadcasts) so that the walk state stays lane-uniform.
    const int width_d = band * 2 + 1;
    int i = qLen - 1, j = tLen - 1, state = 2;
    char *o = bt + tk.btOff + (qLen + tLen + 2);
    int len = 0, ids = 0;
    bool bad = false;
    bool going = have && reached && (i > 0 || j > 0);
    const int sh = grp * 32;
    while (__ballot(going)) {
        {
            const int ii = i - l, jj = j - l;
            bool ok = going && state == 2 && (ii > 0 || jj > 0) && ii >= 0 && jj >= 0;
            int x = ii - band;
            x = x > 0 ? x : 0;
            x = jj - x;
            ok = ok && x >= 0 && x < width_d;
            bool same = false;
            if (ok) {
                const int byte = dirs[(size_t) 16 * ii + (x >> 1)];
                const int code = (x & 1) ? (byte >> 4) : (byte & 15);
                ok = (code & 12) == 0;
                same = sq[ii] == st[jj];
            }
            const uint32_t okMask = (uint32_t) (__ballot(ok) >> sh);
            const uint32_t sameMask = (uint32_t) (__ballot(same) >> sh);
            const int run = okMask == 0xFFFFFFFFu ? 32 : __builtin_ctz(~okMask);
            const uint32_t runMask = run == 32 ? 0xFFFFFFFFu : ((1u << run) - 1u);
            if (l < run) o[-1 - l] = 'M';
            ids += __builtin_popcount(sameMask & runMask);
            i -= run;
            j -= run;
            o -= run;
            len += run;
        }
        if (going && (i > 0 || j > 0)) {
            if (i < 0 || j < 0) bad = true;
            int x = i - band;
            x = x > 0 ? x : 0;
            x = j - x;
            if (x < 0 || x >= width_d) bad = true;
            if (!bad) {
                const int byte = dirs[(size_t) 16 * i + (x >> 1)];
                const int code = (x & 1) ? (byte >> 4) : (byte & 15);
                int dcode;
                const int dE = (code & 1) ? 3 : 2, dF = (code & 2) ? 5 : 4;
                if (state == 0) dcode = dE;
                else if (state == 1) dcode = dF;
                else dcode = (code & 4) ? dE : ((code & 8) ? dF : 1);
                char c;
                switch (dcode) {
                    case 1: ids += (sq[i] == st[j]); --i; --j; state = 2; c = 'M'; break;
                    case 2: --i; state = 0; c = 'I'; break;
                    case 3: --i; state = 2; c = 'I'; break;
                    case 4: --j; state = 1; c = 'D'; break;
                    default: --j; state = 2; c = 'D'; break;
                }
                --o;
                if (l == 0) *o = c;
                len++;
            }
        }
        going = going && !bad && (i > 0 || j > 0);
    }
    if (!have || !reached || l != 0) return;
    if (bad || i != 0 || j != 0) {
        res[2 * id] = -1;
        return;
    }
    ids += (sq[0] == st[0]);
    *--o = 'M';
    len++;
    res[2 * id] = len;
    res[2 * id + 1] = ids;
}

template <int RT, int LW, bool MULTI, bool WIDE, bool SHARED>
void launchScorePk(sd_ctx *ctx, const SwTask *dTasks, const uint32_t *dOrder, uint32_t n, const sd_seqset *q,
                   const sd_seqset *t, const int8_t *dMat, int go, int ge, int32_t *dOut, uint2 *dBound) {
    if (n == 0) return;
    constexpr uint32_t perWave = 2 * (64 / LW);
    dim3 grid((n + perWave - 1) / perWave), block(64);
    // SD_SW_LDS_PAD: unused dynamic LDS per wavefront, i.e. a cap on how many score wavefronts a CU holds -- what it leaves
    // free (LDS, wave slots) is what the memory-bound prefilter workgroups of the other streams run in
    static const unsigned ldsPad = getenv("SD_SW_LDS_PAD") ? (unsigned) atoi(getenv("SD_SW_LDS_PAD")) : 0u;
    hipLaunchKernelGGL((sdpk::sw_score_pk_kernel<RT, LW, MULTI, WIDE, SHARED>), grid, block, ldsPad, ctx->stream, dTasks, n, q->dRes, q->dBias,
                       t->dRes, dMat, go, ge, dOut, dBound, dOrder, (const int8_t *) q->dProf);
}

template <int RT, int LW, int SHARED, int WAVES = 1>
void launchScorePkAligned(sd_ctx *ctx, const SwTask *dTasks, const uint32_t *dOrder, uint32_t n, const sd_seqset *q, const sd_seqset *t,
                          const int8_t *dMat, int go, int ge, int32_t *dOut) {
    if (n == 0) return;
    constexpr uint32_t perWg = 2 * (64 / LW) * WAVES;
    dim3 grid((n + perWg - 1) / perWg), block(64 * WAVES);
    static const unsigned ldsPad = getenv("SD_SW_LDS_PAD") ? (unsigned) atoi(getenv("SD_SW_LDS_PAD")) : 0u;
    hipLaunchKernelGGL((sdpk::sw_score_pk_aligned_kernel<RT, LW, SHARED, WAVES>), grid, block, ldsPad, ctx->stream, dTasks, n, q->dRes, q->dBias, t->dRes,
                       dMat, go, ge, dOut, dOrder, (const int8_t *) q->dProf);
}

// SD_SW_CHAIN=M: a wavefront of the aligned shared-profile kernel takes M consecutive quads of the pair list and chains those of one
// query (sw_score_pk_chain_kernel: one profile, one systolic ramp per chain); 0 / 1: one quad per wavefront
// (measured at 1 000 proteomes, profiles/r05_experiments.txt item 13: 4.6 % fewer VALU instructions per quad at M = 4 and the same
// throughput -- a class launch holds about six quads per wave slot, so M-quad workgroups coarsen the grid as much as they save: off.
// Read at every call: tests switch it inside one process.)
inline uint32_t sdSwChain() {
    const char *e = getenv("SD_SW_CHAIN");
    return e ? (uint32_t) std::min(64, std::max(0, atoi(e))) : 0u;
}
template <int RT>
void launchScorePkChain(sd_ctx *ctx, const SwTask *dTasks, const uint32_t *dOrder, uint32_t n, const sd_seqset *q, const sd_seqset *t,
                        const int8_t *dMat, int go, int ge, int32_t *dOut) {
    if (n == 0) return;
    const uint32_t nQuads = (n + 3) / 4, chainLen = sdSwChain();
    dim3 grid((nQuads + chainLen - 1) / chainLen), block(64);
    static const unsigned ldsPad = getenv("SD_SW_LDS_PAD") ? (unsigned) atoi(getenv("SD_SW_LDS_PAD")) : 0u;
    hipLaunchKernelGGL((sdpk::sw_score_pk_chain_kernel<RT>), grid, block, ldsPad, ctx->stream, dTasks, n, q->dRes, q->dBias, t->dRes, dMat, go, ge,
                       dOut, dOrder, (const int8_t *) q->dProf, chainLen);
}

int rtClass(int n) {
    if (n <= 128) return 4;
    if (n <= 256) return 8;
    if (n <= 512) return 16;
    return 32;
}

template <int RT>
void launchScore(sd_ctx *ctx, const SwTask *dTasks, uint32_t n, const sd_seqset *q, const sd_seqset *t,
                 const int8_t *dMat, int go, int ge, int32_t *dOut, uint2 *dBound) {
    if (n == 0) return;
    dim3 grid((n + 1) / 2), block(64);
    hipLaunchKernelGGL(sw_score_kernel<RT>, grid, block, 0, ctx->stream, dTasks, n, q->dRes, q->dBias, t->dRes, dMat,
                       go, ge, dOut, dBound, (const uint32_t *) nullptr, (const int8_t *) q->dProf);
}

template <int RT>
void launchScoreIdx(sd_ctx *ctx, const SwTask *dTasks, const uint32_t *dOrder, uint32_t n, const sd_seqset *q,
                    const sd_seqset *t, const int8_t *dMat, int go, int ge, int32_t *dOut, uint2 *dBound) {
    if (n == 0) return;
    dim3 grid((n + 1) / 2), block(64);
    hipLaunchKernelGGL(sw_score_kernel<RT>, grid, block, 0, ctx->stream, dTasks, n, q->dRes, q->dBias, t->dRes, dMat,
                       go, ge, dOut, dBound, dOrder, (const int8_t *) q->dProf);
}

// run a list of score tasks (any mix of sizes); results land in hOut[3*slot..]
// Tasks are ordered by (RT class, tL descending) with a parallel stable counting sort straight into a pinned
// staging buffer (two tasks share a wavefront, so neighbours should have similar lengths; biggest first).
int runScoreTasks(sd_ctx *ctx, std::vector<SwTask> &tasks, const sd_seqset *q, const sd_seqset *t, const int8_t *dMat,
                  int go, int ge, std::vector<int32_t> &hOut, uint32_t nSlots, uint64_t *cells, const char *tag = "a") {
    hOut.assign((size_t) nSlots * 3, 0);
    if (tasks.empty()) return SD_OK;
    HostScope hsAll(ctx, "score.total");
    const size_t n = tasks.size();
    constexpr int NB = 4 * 1024;
    auto keyOf = [](const SwTask &t) {
        const int c = rtClass(t.n);
        const int ci = c == 4 ? 0 : (c == 8 ? 1 : (c == 16 ? 2 : 3));
        return (uint32_t) ci * 1024u + (uint32_t) (1023 - std::min(t.tL >> 4, 1023));
    };
    SwTask *sorted = nullptr;
    SD_HIP(ctx, pinGet(ctx, (std::string("sw.tasks.") + tag).c_str(), n, &sorted));
    uint64_t boundTotal = 0, cellSum = 0;
    size_t classBegin[5] = {0, 0, 0, 0, 0};
    {
        int T = 1;
#pragma omp parallel
        {
#pragma omp single
            T = omp_get_num_threads();
        }
        T = std::max(1, std::min(T, 64));
        std::vector<uint32_t> hist((size_t) T * NB, 0);
#pragma omp parallel num_threads(T) reduction(+ : cellSum)
        {
            const int th = omp_get_thread_num();
            const size_t lo = n * th / T, hi = n * (th + 1) / T;
            uint32_t *hh = &hist[(size_t) th * NB];
            for (size_t i = lo; i < hi; i++) {
                hh[keyOf(tasks[i])]++;
                cellSum += (uint64_t) tasks[i].n * (uint64_t) tasks[i].tL;
            }
        }
        // exclusive prefix over (bucket major, thread minor)
        uint64_t run = 0;
        for (int bkt = 0; bkt < NB; bkt++) {
            if ((bkt & 1023) == 0) classBegin[bkt >> 10] = run;
            for (int th = 0; th < T; th++) {
                const uint32_t c = hist[(size_t) th * NB + bkt];
                hist[(size_t) th * NB + bkt] = (uint32_t) run;
                run += c;
            }
        }
        classBegin[4] = run;
#pragma omp parallel num_threads(T)
        {
            const int th = omp_get_thread_num();
            const size_t lo = n * th / T, hi = n * (th + 1) / T;
            uint32_t *hh = &hist[(size_t) th * NB];
            for (size_t i = lo; i < hi; i++) sorted[hh[keyOf(tasks[i])]++] = tasks[i];
        }
    }
    *cells += cellSum;
    for (size_t i = classBegin[3]; i < classBegin[4]; i++) {   // multi-strip tasks exist in class 32 only
        if (sorted[i].n > 1024) {
            sorted[i].boundOff = boundTotal;
            boundTotal += (uint64_t) sorted[i].tL;
        }
    }
    SwTask *dTasks = nullptr;
    int32_t *dOut = nullptr;
    uint2 *dBound = nullptr;
    int32_t *hPin = nullptr;
    SD_HIP(ctx, wsGet(ctx, "sw.tasks", n, &dTasks));
    SD_HIP(ctx, wsGet(ctx, "sw.out", (size_t) nSlots * 3, &dOut));
    SD_HIP(ctx, wsGet(ctx, "sw.bound", std::max<uint64_t>(boundTotal, 1), &dBound));
    SD_HIP(ctx, pinGet(ctx, "sw.hout", (size_t) nSlots * 3, &hPin));
    SD_HIP(ctx, hipMemcpyAsync(dTasks, sorted, n * sizeof(SwTask), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dOut, 0, (size_t) nSlots * 3 * sizeof(int32_t), ctx->stream));
    static const int classes[4] = {4, 8, 16, 32};
    for (int ci = 0; ci < 4; ci++) {
        const size_t begin = classBegin[ci], end = classBegin[ci + 1];
        if (end == begin) continue;
        const uint32_t cnt = (uint32_t) (end - begin);
        {
            ProfScope ps(ctx, "sw_score");
            switch (classes[ci]) {
                case 4: launchScore<4>(ctx, dTasks + begin, cnt, q, t, dMat, go, ge, dOut, dBound); break;
                case 8: launchScore<8>(ctx, dTasks + begin, cnt, q, t, dMat, go, ge, dOut, dBound); break;
                case 16: launchScore<16>(ctx, dTasks + begin, cnt, q, t, dMat, go, ge, dOut, dBound); break;
                default: launchScore<32>(ctx, dTasks + begin, cnt, q, t, dMat, go, ge, dOut, dBound); break;
            }
        }
        SD_HIP(ctx, hipGetLastError());
    }
    SD_HIP(ctx, hipMemcpyAsync(hPin, dOut, (size_t) nSlots * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, sdStreamSync(ctx));
    memcpy(hOut.data(), hPin, (size_t) nSlots * 3 * sizeof(int32_t));
    return SD_OK;
}


// ---------------------------------------------------------------------------------------------
// Device-resident orchestration of a batch of pairs: task construction, the gates between the passes and
// the ordering of tasks all happen on the GPU; the host only launches, reads six class boundaries per pass
// and receives the finished result records + a dense backtrace pool.
// ---------------------------------------------------------------------------------------------
// score-pass task classes (the pair sort carries the class in six bits: fewer than 64)
//   0          packed-int16 kernel, <= 128 rows (general form: per-row segment masks)
//   1 .. 20    aligned packed kernel (sw_score_pk_aligned_kernel): RT = ceil(n / 32) = 5 .. 24, i.e. 129 .. 768 rows in steps
//              of 32 -- a task pays for the rows of its class, and the reference's 32 SIMD segments are the 32 lanes
//   21 .. 23   2 / 3 / more strips of 512 rows (general form)
//   24 .. 37   the general packed kernel with the wide row code (word-kernel reruns; rows: <=128, <=192, <=224, <=256,
//              <=288, <=320, <=352, <=384, <=512, <=640, <=768, then 2 / 3 / more strips)
//   38 .. 41   int32 kernel (rows: <=128, <=256, <=512, more)
constexpr uint32_t N_NARROW_CLASSES = 24;
constexpr uint32_t FIRST_NARROW_MULTI = 21;
constexpr uint32_t N_PK_CLASSES = 14;     // classes of the wide form
constexpr uint32_t FIRST_MULTI_PK = 11;   // (wide form) classes of more than one strip
constexpr uint32_t FIRST_WIDE_CLASS = N_NARROW_CLASSES;
constexpr uint32_t FIRST_INT32_CLASS = FIRST_WIDE_CLASS + N_PK_CLASSES;
constexpr uint32_t N_SCORE_CLASSES = FIRST_INT32_CLASS + 4;
static_assert(N_SCORE_CLASSES < 63, "class boundaries are found by one 64-lane wavefront; six class bits in the pair key");
enum ScoreKernel { SCORE_PK = 0, SCORE_PK_WIDE = 1, SCORE_INT32 = 2 };
constexpr uint32_t KEY_INVALID = N_SCORE_CLASSES * 1024u;   // sorts after every class
__host__ __device__ __forceinline__ bool scoreClassIsMulti(uint32_t ci) {   // packed classes of more than one strip
    return (ci >= FIRST_NARROW_MULTI && ci < FIRST_WIDE_CLASS) || (ci >= FIRST_WIDE_CLASS + FIRST_MULTI_PK && ci < FIRST_INT32_CLASS);
}
// wideRowLimit: int16 cells hold a score for certain while min(n, tL) * (largest entry of this call's query profiles) <= 32 767
// (no saturation of the reference's word kernel to reproduce, no wrap); longer pairs take the int32 kernel
__device__ __forceinline__ uint32_t scoreKey(int n, int tL, int kernel, int wideRowLimit = 800) {
    int ci;
    if (kernel == SCORE_PK_WIDE && min(n, tL) > wideRowLimit) kernel = SCORE_INT32;
    if (kernel == SCORE_INT32 || tL > 65535) {
        ci = FIRST_INT32_CLASS + (n <= 128 ? 0 : (n <= 256 ? 1 : (n <= 512 ? 2 : 3)));
    } else if (kernel == SCORE_PK_WIDE) {
        if (n <= 192) ci = n <= 128 ? 0 : 1;
        else if (n <= 384) ci = 2 + (n - 193) / 32;                      // 224, 256, 288, 320, 352, 384 rows -> 2..7
        else if (n <= 768) ci = n <= 512 ? 8 : (n <= 640 ? 9 : 10);
        else ci = min((int) FIRST_MULTI_PK + (n + 511) / 512 - 2, 13);   // 2 strips -> 11, 3 strips -> 12, more -> 13
        ci += FIRST_WIDE_CLASS;
    } else {
        if (n <= 128) ci = 0;
        else if (n <= 768) ci = (n + 31) / 32 - 4;                       // RT = 5 .. 24 -> 1 .. 20
        else ci = min((int) FIRST_NARROW_MULTI + (n + 511) / 512 - 2, (int) FIRST_WIDE_CLASS - 1);
    }
    return (uint32_t) ci * 1024u + (uint32_t) (1023 - min(tL >> 4, 1023));
}

struct DevGateParams {
    int go, ge, matMin, swMode, covMode, wideRowLimit;
    float covThr;
    double evalThr;
    // Gumbel parameters (sd::Evaluer) for the device-side E-value gate
    double lambda, K, aI, bI, alphaI, betaI, aJ, bJ, alphaJ, betaJ, sigma, tau, viThr, vjThr, cThr, dbResidues;
};

__device__ double devEvalue(const DevGateParams &e, double y_, double qLen) {
    const double m_ = e.dbResidues, n_ = qLen;
    const double const_val = 0.3989422804014327;   // 1/sqrt(2 pi)
    double tmp = e.aI * y_ + e.bI;
    double m_li_y = m_ - tmp;
    double vi_y = fmax(e.viThr, e.alphaI * y_ + e.betaI);
    double sqrt_vi_y = sqrt(vi_y);
    double m_F = sqrt_vi_y == 0.0 ? 1e100 : m_li_y / sqrt_vi_y;
    double P_m_F = 0.5 * erfc(-sqrt(0.5) * m_F);
    double E_m_F = -const_val * exp(-0.5 * m_F * m_F);
    double p1 = m_li_y * P_m_F - sqrt_vi_y * E_m_F;
    tmp = e.aJ * y_ + e.bJ;
    double n_lj_y = n_ - tmp;
    double vj_y = fmax(e.vjThr, e.alphaJ * y_ + e.betaJ);
    double sqrt_vj_y = sqrt(vj_y);
    double n_F = sqrt_vj_y == 0.0 ? 1e100 : n_lj_y / sqrt_vj_y;
    double P_n_F = 0.5 * erfc(-sqrt(0.5) * n_F);
    double E_n_F = -const_val * exp(-0.5 * n_F * n_F);
    double p2 = n_lj_y * P_n_F - sqrt_vj_y * E_n_F;
    double c_y = fmax(e.cThr, e.sigma * y_ + e.tau);
    double area = p1 * p2 + c_y * (P_m_F * P_n_F);
    return e.K * exp(-e.lambda * y_) * area;
}

__device__ __forceinline__ float devCov(unsigned int startPos, unsigned int endPos, unsigned int len) {
    return (min(len, max(startPos, endPos)) - min(startPos, endPos) + 1) / (float) len;
}
__device__ __forceinline__ bool devHasCoverage(float covThr, int covMode, float qCov, float tCov) {
    switch (covMode) {
        case 0: return (qCov >= covThr) && (tCov >= covThr);
        case 2: return qCov >= covThr;
        case 1: return tCov >= covThr;
        default: return true;
    }
}

// A pair whose byte-kernel score is CERTAIN to saturate needs no byte-structure pass: the reference discards that pass's
// result and reruns the pair with the word kernel (:360-368), and the score of any ungapped stretch of one diagonal is a lower
// bound of the byte kernel's maximum (a diagonal-only path never meets the lane structure's vertical-gap quirk).  With the
// prefilter's diagonal at hand, k_pre_word walks that diagonal with the alignment's own scoring (matrix row + composition bias,
// or the profile row) and marks the pairs whose bound already reaches 255 - bias; they skip pass 1 and go straight to pass 2.
// On proteome-scale searches that is most true homologs, i.e. the longest-running tasks of pass 1.
__global__ void __launch_bounds__(256)
k_pre_word(uint32_t nPairs, const uint32_t *__restrict__ pairQ, const uint32_t *__restrict__ pairT, const uint16_t *__restrict__ pairDiag,
           const uint32_t *__restrict__ fwdKeys, const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff,
           const uint8_t *__restrict__ qRes, const int8_t *__restrict__ qBias, const uint8_t *__restrict__ tRes, const int8_t *__restrict__ mat,
           const int8_t *__restrict__ qProf, const int32_t *__restrict__ minBias, int matMin, uint8_t *__restrict__ preWord,
           uint32_t *__restrict__ pass1Keys) {
    __shared__ int8_t smat[441];
    for (int x = threadIdx.x; x < 441; x += blockDim.x) smat[x] = mat[x];
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    uint8_t pw = 0;
    const uint32_t key = fwdKeys[i];
    if (key != KEY_INVALID) {
        const uint64_t qo = qOff[pairQ[i]], to = tOff[pairT[i]];
        const int qL = (int) (qOff[pairQ[i] + 1] - qo), tL = (int) (tOff[pairT[i] + 1] - to);
        if (qL < 32768 && tL < 32768 && pairDiag[i] != 0x8000) {   // (longer sequences: the 16-bit diagonal is ambiguous; 0x8000 = caller has no hint)
            const int d = (int) (int16_t) pairDiag[i];
            int q0 = d >= 0 ? d : 0, t0 = d >= 0 ? 0 : -d;
            const int n = min(qL - q0, tL - t0);
            const int need = 255 - (abs(matMin) + abs(minBias[pairQ[i]]));
            int run = 0, best = 0;
            for (int x = 0; x < n && best < need; x++) {
                const int qr = qRes[qo + q0 + x], tr = tRes[to + t0 + x];
                run += qProf ? (int) qProf[(qo + q0 + x) * 21 + tr] : (int) smat[tr * 21 + qr] + (int) qBias[qo + q0 + x];
                run = run < 0 ? 0 : run;
                best = run > best ? run : best;
            }
            pw = best >= need ? 1 : 0;
        }
    }
    preWord[i] = pw;
    pass1Keys[i] = pw ? KEY_INVALID : key;
}

// forward tasks (32-lane structure) for every non-identity pair
__global__ void __launch_bounds__(256)
k_make_fwd(uint32_t nPairs, const uint32_t *__restrict__ pairQ, const uint32_t *__restrict__ pairT,
           const uint8_t *__restrict__ ident, const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff,
           SwTask *__restrict__ tasks, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
           sd_sw_result *__restrict__ res, int usePk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    const uint64_t qo = qOff[pairQ[i]], to = tOff[pairT[i]];
    const int qL = (int) (qOff[pairQ[i] + 1] - qo), tL = (int) (tOff[pairT[i] + 1] - to);
    SwTask tk;
    tk.qOff = qo; tk.tOff = to; tk.n = qL; tk.tL = tL; tk.qStep = 1; tk.tStep = 1;
    tk.segLen = max(1, (qL + 31) / 32);
    tk.slot = i; tk.boundOff = 0;
    const bool valid = !(ident && ident[i]) && qL > 0 && tL > 0;
    tasks[i] = tk;
    keys[i] = valid ? scoreKey(qL, tL, usePk ? SCORE_PK : SCORE_INT32) : KEY_INVALID;
    vals[i] = i;
    sd_sw_result r;
    r.score = 0; r.qStart = -1; r.qEnd = -1; r.tStart = -1; r.tEnd = -1; r.identical = 0; r.btLen = 0; r.flags = 0;
    r.evalue = 0.0; r.btOffset = 0;
    res[i] = r;
}

// class boundaries of a sorted key array: b[c] = lower_bound(c*1024), c = 0..4 (b[4] = number of valid tasks)
__global__ void k_bounds(const uint32_t *__restrict__ keys, uint32_t n, uint32_t step, uint32_t *__restrict__ b, int nb) {
    const int c = threadIdx.x;
    if (c >= nb) return;
    const uint64_t want = (uint64_t) c * step;   // (32 classes x 2^27: the end bound does not fit 32 bits)
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t) keys[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    b[c] = lo;
}

// after the 32-lane pass: pairs whose byte score saturates get a 16-lane task (:881,916,360-368)
__global__ void __launch_bounds__(256)
k_gate_word(uint32_t nPairs, const uint32_t *__restrict__ pairQ, const int32_t *__restrict__ out32,
            const int32_t *__restrict__ minBias, int matMin, SwTask *__restrict__ tasks, uint32_t *__restrict__ keys,
            uint32_t *__restrict__ vals, uint8_t *__restrict__ word, const uint32_t *__restrict__ fwdKeys, int usePk, int wideRowLimit,
            const uint8_t *__restrict__ preWord /* nullable: pairs that skipped pass 1 because their byte score saturates for certain */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    vals[i] = i;
    bool w = false;
    if (fwdKeys[i] != KEY_INVALID) {
        const int mb = minBias[pairQ[i]];
        const int bias = abs(matMin) + abs(mb);
        w = (preWord && preWord[i]) || out32[3 * i] + bias >= 255;
    }
    word[i] = w ? 1 : 0;
    if (w) {
        tasks[i].segLen = max(1, (tasks[i].n + 15) / 16);
        keys[i] = scoreKey(tasks[i].n, tasks[i].tL, usePk ? SCORE_PK_WIDE : SCORE_INT32, wideRowLimit);
    } else {
        keys[i] = KEY_INVALID;
    }
}

// gates after the score pass (:389-398) and reverse tasks
__global__ void __launch_bounds__(256)
k_gate_rev(uint32_t nPairs, DevGateParams gp, const uint32_t *__restrict__ pairQ, const uint32_t *__restrict__ pairT,
           const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff, const int32_t *__restrict__ out32,
           const int32_t *__restrict__ out16, const uint8_t *__restrict__ word, const uint32_t *__restrict__ fwdKeys,
           SwTask *__restrict__ tasks, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
           sd_sw_result *__restrict__ res, int usePk) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    vals[i] = i;
    keys[i] = KEY_INVALID;
    if (fwdKeys[i] == KEY_INVALID) return;
    const int32_t *src = word[i] ? &out16[3 * i] : &out32[3 * i];
    sd_sw_result r = res[i];
    r.score = src[0];
    r.tEnd = src[1];
    r.qEnd = src[2];
    r.flags = word[i] ? 1 : 0;
    if (word[i] && r.tEnd == -1) r.tEnd = 0;   // the word kernel initialises end_ref to 0 (:962)
    if (r.score == 0) r.qEnd = 0;   // no cell scored: the saved column is all zero and the first row matches it (:903-913)
    if (r.tEnd == -1) { res[i] = r; return; }
    const uint64_t qo = qOff[pairQ[i]], to = tOff[pairT[i]];
    const int qL = (int) (qOff[pairQ[i] + 1] - qo), tL = (int) (tOff[pairT[i] + 1] - to);
    const double ev = devEvalue(gp, (double) r.score, (double) qL);
    r.evalue = ev;   // provisional (gate only); the host recomputes it with the reference's arithmetic
    // a result within 1e-9 of the threshold is passed on and decided by the host's exact value
    const bool lowE = ev > gp.evalThr * (1.0 + 1e-9);
    const float qCov = devCov(0, r.qEnd, qL), tCov = devCov(0, r.tEnd, tL);
    const bool lowCov = !devHasCoverage(gp.covThr, gp.covMode, qCov, tCov);
    res[i] = r;
    if (gp.swMode == 0 || lowE || lowCov) return;
    SwTask tk;
    tk.qOff = qo + r.qEnd; tk.tOff = to + r.tEnd;
    tk.n = r.qEnd + 1; tk.tL = r.tEnd + 1; tk.qStep = -1; tk.tStep = -1;
    const int lanes = word[i] ? 16 : 32;
    tk.segLen = max(1, (tk.n + lanes - 1) / lanes);
    tk.slot = i; tk.boundOff = 0;
    tasks[i] = tk;
    keys[i] = scoreKey(tk.n, tk.tL, !usePk ? SCORE_INT32 : (word[i] ? SCORE_PK_WIDE : SCORE_PK), gp.wideRowLimit);
}

// traceback task classes: N_TB_NARROW register-band classes by query rows, then three LDS-band classes by band width
// (<=127, <=511, <=2047) and the global-band class for everything wider
// The register-band kernel keeps a task's direction codes (16 B per row), query / bias / target slices in LDS, and
// it is latency bound, so its throughput is its occupancy: classes by query rows keep the LDS request close to what
// the tasks need (a band <= 14 means |tLen - qLen| <= 13, so qCap + 16 target columns always suffice).
constexpr int N_TB_NARROW = 9;
__constant__ int c_tbNarrowQ[N_TB_NARROW] = {128, 192, 256, 320, 384, 512, 640, 768, 1024};
static const int TB_NARROW_Q[N_TB_NARROW] = {128, 192, 256, 320, 384, 512, 640, 768, 1024};
__device__ __forceinline__ bool tbNarrow(int band, int qLen, int tLen) { return band * 2 + 3 <= 32 && qLen <= 1024 && tLen <= 1024 + 13; }
__device__ __forceinline__ uint32_t tbKey(int band, int qLen, int tLen) {
    const int w = band * 2 + 3;
    int ci;
    if (tbNarrow(band, qLen, tLen)) {
        ci = 0;
        while (qLen > c_tbNarrowQ[ci]) ci++;
    } else {
        ci = N_TB_NARROW + (w <= 127 ? 0 : (w <= 511 ? 1 : (w <= 2047 ? 2 : 3)));
    }
    if (ci < N_TB_NARROW) {
        // register-band kernel: the two tasks of a wavefront run in lock step through max(attempts) x max(rows).  Tasks with the
        // same number of bands the kernel can try (initial band 1: 1, 2, 4, 8; 2-3: three; 4-7: two; 8-14: one) sit together,
        // most attempts first, longest query first inside
        const int tries = band <= 1 ? 4 : (band <= 3 ? 3 : (band <= 7 ? 2 : 1));
        return (uint32_t) ci * 4096u + (uint32_t) (4 - tries) * 1024u + (uint32_t) (1023 - min(max(qLen - 1, 0), 1023));
    }
    const unsigned long long work = (unsigned long long) ((2 * band + 1 + 31) / 32) * (unsigned long long) qLen;
    return (uint32_t) ci * 4096u + (uint32_t) (4095 - (int) min(work >> 3, 4095ull));
}
constexpr uint32_t N_TB_CLASSES = N_TB_NARROW + 4;   // + the global-band class (w > 2047)
constexpr uint32_t TBKEY_INVALID = N_TB_CLASSES * 4096u;
// global direction scratch (LDS-band kernel only), sized for the widest band of the task's class because the
// kernel keeps doubling the band inside its class
__device__ __forceinline__ uint64_t tbDirBytes(int band, int qLen, int tLen) {
    if (tbNarrow(band, qLen, tLen)) return 0ull;
    const int w = band * 2 + 3;
    if (w > 2047)   // global-band class: one attempt per round at exactly this band, band arrays in front of the directions
        return (tbGlobalIntBytes(band) + (uint64_t) (w - 2) * (uint64_t) qLen + 31) & ~15ull;
    const int wMax = w <= 127 ? 127 : (w <= 511 ? 511 : 2047);
    return ((uint64_t) (wMax - 2) * (uint64_t) qLen + 31) & ~15ull;   // multiples of 16: every task's region stays aligned
}

// start positions (:475-476), second coverage gate (:483-489) and traceback tasks
__global__ void __launch_bounds__(256)
k_gate_tb(uint32_t nPairs, DevGateParams gp, const uint32_t *__restrict__ pairQ, const uint32_t *__restrict__ pairT,
          const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff, const int32_t *__restrict__ outRev,
          const uint32_t *__restrict__ revKeys, TbTask *__restrict__ tb, uint32_t *__restrict__ keys,
          uint32_t *__restrict__ vals, uint64_t *__restrict__ dirBytes, uint64_t *__restrict__ btBytes,
          sd_sw_result *__restrict__ res, int *__restrict__ errFlag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    vals[i] = i;
    keys[i] = TBKEY_INVALID;
    dirBytes[i] = 0;
    btBytes[i] = 0;
    if (revKeys[i] == KEY_INVALID) return;
    sd_sw_result r = res[i];
    if (outRev[3 * i] != r.score) {
        r.flags |= 2;
        res[i] = r;
        atomicExch(errFlag, 1);
        return;
    }
    r.tStart = r.tEnd - outRev[3 * i + 1];
    r.qStart = r.qEnd - outRev[3 * i + 2];
    res[i] = r;
    const uint64_t qo = qOff[pairQ[i]], to = tOff[pairT[i]];
    const int qL = (int) (qOff[pairQ[i] + 1] - qo), tL = (int) (tOff[pairT[i] + 1] - to);
    const float qCov = devCov(r.qStart, r.qEnd, qL), tCov = devCov(r.tStart, r.tEnd, tL);
    const bool lowCov = !devHasCoverage(gp.covThr, gp.covMode, qCov, tCov);
    if (gp.swMode == 1 || lowCov) return;
    TbTask t;
    t.qAbs = qo + r.qStart; t.tAbs = to + r.tStart;
    t.qLen = r.qEnd - r.qStart + 1; t.tLen = r.tEnd - r.tStart + 1;
    t.score = r.score;
    t.band = abs(t.tLen - t.qLen) + 1;
    t.maxv = 0; t.slot = i; t.intOff = 0; t.dirOff = 0; t.btOff = 0;
    tb[i] = t;
    keys[i] = tbKey(t.band, t.qLen, t.tLen);
    dirBytes[i] = tbDirBytes(t.band, t.qLen, t.tLen);
    btBytes[i] = (uint64_t) t.qLen + t.tLen + 2;
}

// this round's tasks: everything whose direction scratch ends inside the budget (prefix of the pending tasks in pair
// order: the first one starts at 0); the others keep their keys and come again
__global__ void __launch_bounds__(256)
k_tb_round(uint32_t nPairs, const uint32_t *__restrict__ keys, const uint64_t *__restrict__ dirOff,
           const uint64_t *__restrict__ dirBytes, uint64_t budget, uint32_t *__restrict__ keysRound) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    const uint32_t k = keys[i];
    keysRound[i] = (k != TBKEY_INVALID && dirOff[i] + dirBytes[i] <= budget) ? k : TBKEY_INVALID;
}

__global__ void __launch_bounds__(256)
k_tb_offsets(uint32_t nPairs, const uint32_t *__restrict__ keys, const uint64_t *__restrict__ dirOff,
             const uint64_t *__restrict__ btOff, TbTask *__restrict__ tb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs || keys[i] == TBKEY_INVALID) return;
    tb[i].dirOff = dirOff[i];
    tb[i].btOff = btOff[i];
}

// after a traceback round: finished tasks publish their result, the others double their band (:1492-1493)
__global__ void __launch_bounds__(256)
k_tb_collect(uint32_t nPairs, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, TbTask *__restrict__ tb,
             const int32_t *__restrict__ tbRes, uint64_t *__restrict__ dirBytes, uint64_t *__restrict__ btLenOut,
             sd_sw_result *__restrict__ res, int *__restrict__ errFlag, int maxBand,
             const uint32_t *__restrict__ keysRound /* what this round ran: deferred tasks keep key and scratch size */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    vals[i] = i;
    if (keys[i] == TBKEY_INVALID) { dirBytes[i] = 0; return; }
    if (keysRound[i] == TBKEY_INVALID) return;
    const int len = tbRes[2 * i];
    if (len == -2) {
        const int band = tb[i].band;   // already doubled by the kernel that gave up
        // the full rectangle is inside the band once band >= max(qLen, tLen): a task that still misses its score then is
        // a traceback error, not a reason to double again
        if (band > maxBand) { atomicExch(errFlag, 2); keys[i] = TBKEY_INVALID; dirBytes[i] = 0; return; }
        keys[i] = tbKey(band, tb[i].qLen, tb[i].tLen);
        dirBytes[i] = tbDirBytes(band, tb[i].qLen, tb[i].tLen);
    } else if (len < 0) {
        atomicExch(errFlag, 2);
        keys[i] = TBKEY_INVALID;
        dirBytes[i] = 0;
    } else {
        res[i].btLen = len;
        res[i].identical = tbRes[2 * i + 1];
        btLenOut[i] = (uint64_t) len;
        keys[i] = TBKEY_INVALID;
        dirBytes[i] = 0;
    }
}

__global__ void __launch_bounds__(256)
k_bt_pack(uint32_t nPairs, const uint64_t *__restrict__ btLen, const uint64_t *__restrict__ dense,
          const TbTask *__restrict__ tb, const char *__restrict__ bt, char *__restrict__ pool, uint64_t poolBase,
          sd_sw_result *__restrict__ res) {
    // one wavefront per pair copies its backtrace into the dense pool
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nPairs) return;
    const uint64_t len = btLen[i];
    if (len == 0) return;
    const char *src = bt + tb[i].btOff + ((uint64_t) (tb[i].qLen + tb[i].tLen + 2) - len);   // written backwards from the end
    char *dst = pool + dense[i];
    for (uint64_t x = lane; x < len; x += 64) dst[x] = src[x];
    if (lane == 0) res[i].btOffset = poolBase + dense[i];
}

// ---- Matcher::compressAlignment on the device (M/src/alignment/Matcher.cpp:166-185; sd_sw_set_cigar_pool): the run-length text
// of a backtrace -- "57M2I103M" -- instead of its letters.  State 'M' with a count of 0 at the start (a backtrace that does not begin
// with a match begins "0M"), every run as decimal count + letter.  One wavefront per pair, 64 letters per step: a lane whose right
// neighbour differs ends a run, the run began at the last lane at or below it whose left neighbour differs (a ballot; the last start
// of the earlier steps is carried), and a scan of the text lengths of the step's runs places them.  WRITE = false: the text's length
// only (the pool offsets are a scan of these), WRITE = true: the text, the record's offset and -- flags bits 8.. -- its length.
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_bt_cigar(uint32_t nPairs, const uint64_t *__restrict__ btLen, const uint64_t *__restrict__ dense, const TbTask *__restrict__ tb,
           const char *__restrict__ bt, char *__restrict__ pool, uint64_t *__restrict__ cigLen, sd_sw_result *__restrict__ res) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= nPairs) return;
    const uint32_t len = (uint32_t) btLen[i];
    if (len == 0) {
        if (!WRITE && lane == 0) cigLen[i] = 0;
        return;
    }
    const char *src = bt + tb[i].btOff + ((uint64_t) (tb[i].qLen + tb[i].tLen + 2) - len);   // written backwards from the end
    char *dst = WRITE ? pool + dense[i] : nullptr;
    uint32_t out = 0;
    uint32_t runStart = 0;
    int left = src[0];   // the letter in front of the step's first one (in front of the first letter: itself, position 0 starts a run anyway)
    if (left != 'M') {
        if (WRITE && lane == 0) {
            dst[0] = '0';
            dst[1] = 'M';
        }
        out = 2;
    }
    for (uint32_t base = 0; base < len; base += 64) {
        const uint32_t x = base + (uint32_t) lane;
        const bool valid = x < len;
        const int c = valid ? (int) src[x] : -1;
        const int after = base + 64 < len ? (int) src[base + 64] : -2;   // the letter behind the step's last one (or the end)
        int prv = __shfl_up(c, 1, 64);
        if (lane == 0) prv = left;
        int nxt = __shfl_down(c, 1, 64);
        if (lane == 63) nxt = after;
        if (valid && x + 1 == len) nxt = -2;
        const bool isStart = valid && (x == 0 || prv != c);
        const bool isEnd = valid && nxt != c;
        const unsigned long long starts = __ballot(isStart);
        const unsigned long long mine = starts & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const uint32_t from = mine ? base + (uint32_t) (63 - __clzll((long long) mine)) : runStart;
        uint32_t count = 0, nd = 0;
        if (isEnd) {
            count = x - from + 1;
            nd = count < 10 ? 1 : count < 100 ? 2 : count < 1000 ? 3 : count < 10000 ? 4 : count < 100000 ? 5 : count < 1000000 ? 6 : 7;
        }
        uint32_t incl = isEnd ? nd + 1 : 0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (WRITE && isEnd) {
            char *o = dst + out + incl - 1;   // the run's letter; the digits go in front of it
            *o = (char) c;
            uint32_t v = count;
            for (uint32_t d = 0; d < nd; d++) {
                *--o = (char) ('0' + v % 10);
                v /= 10;
            }
        }
        out += __shfl(incl, 63, 64);
        if (starts) runStart = base + (uint32_t) (63 - __clzll((long long) starts));
        left = __shfl(c, 63, 64);
    }
    if (lane == 0) {
        if (WRITE) {
            res[i].btOffset = dense[i];
            res[i].flags = (res[i].flags & 0xFF) | (int32_t) (out << 8);
        } else {
            cigLen[i] = out;
        }
    }
}

// ---- besthitbyset on the device (sd_sw_align_batch_best_by_group).  Matcher::compareHits orders a query's accepted alignments by
// E-value, rounded bit score, target length, target key; for one query the E-value is a strictly decreasing function of the score
// and the bit score a function of it, so the first of a (query, target set) cell is the maximum of
//   score << 44 | (0xFFFF - target length) << 28 | (0x0FFFFFFF - key)
// over the cell's accepted pairs: one atomic maximum per pair into a table of queries x target sets, then a second look.
// "Strictly decreasing" ends where the double does: K m n exp(-lambda S) is 0.0 from a raw score of ~2 700 on (about 520
// identical residues: two near-identical paralogs of a long protein in one target genome) and collides in the last
// subnormals below that, and compareHits then decides by the ROUNDED bit score (0.39 bit per score unit: neighbouring scores
// tie), the shorter target, the smaller key -- not by the raw score.  So every accepted pair whose E-value is below
// BB_EVAL_EXACT (far above the first subnormal, far below anything two different scores can share) is kept whatever the
// cell's maximum is, and the host's selection -- compareHits itself -- decides among them as it does without the device's
// help; a pair above the bound can only lose to them (smaller score) and is dropped as before.
constexpr double BB_EVAL_EXACT = 1e-290;
struct BestByGroup {
    const uint32_t *groupOf, *groupKey;   // per target sequence (groupKey nullable: the index)
    uint32_t nGroups;
    float seqIdThr, covThr;
    int alnLenThr, covMode;
    double evalThr;
    double evalExact;   // BB_EVAL_EXACT (SD_BEST_EXACT=0 in the environment: 0.0, the round-5 behaviour, for the A/B of the test)
};
__device__ __forceinline__ float bbCov(uint32_t startPos, uint32_t endPos, uint32_t len) {   // sd::computeCov (Util::computeCov)
    return (float) (min(len, max(startPos, endPos)) - min(startPos, endPos) + 1u) / (float) len;
}
__device__ __forceinline__ bool bbAccepted(const sd_sw_result &r, const BestByGroup &B, uint32_t qL, uint32_t tL) {
    if (r.btLen <= 0 || r.qStart < 0 || r.tStart < 0) return false;   // stopped at a gate
    const float qcov = bbCov((uint32_t) r.qStart, (uint32_t) r.qEnd, qL), dbcov = bbCov((uint32_t) r.tStart, (uint32_t) r.tEnd, tL);
    const float seqId = (float) r.identical / (float) r.btLen;
    bool cov = true;
    if (B.covMode == 0) cov = qcov >= B.covThr && dbcov >= B.covThr;
    else if (B.covMode == 2) cov = qcov >= B.covThr;
    else if (B.covMode == 1) cov = dbcov >= B.covThr;
    // (the device's E-value stands within 1e-15 of the host's: a pair inside that band of the threshold is the worst of its cell
    // and can only be the maximum where nothing better exists, so keeping it costs a record and never displaces one)
    return r.evalue <= B.evalThr * (1.0 + 1e-9) && seqId >= B.seqIdThr && cov && r.btLen >= B.alnLenThr;
}
__device__ __forceinline__ unsigned long long bbPack(const sd_sw_result &r, uint32_t tL, uint32_t key) {
    return ((unsigned long long) min((uint32_t) r.score, 0xFFFFFu) << 44) | ((unsigned long long) (0xFFFFu - min(tL, 0xFFFFu)) << 28) |
           (unsigned long long) (0x0FFFFFFFu - min(key, 0x0FFFFFFFu));
}
__global__ void __launch_bounds__(256)
k_best_mark(uint32_t nPairs, const sd_sw_result *__restrict__ res, const uint8_t *__restrict__ ident, const uint32_t *__restrict__ pq,
            const uint32_t *__restrict__ pt, const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff, BestByGroup B,
            unsigned long long *__restrict__ table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs || ident[i]) return;
    const uint32_t q = pq[i], t = pt[i];
    const uint32_t qL = (uint32_t) (qOff[q + 1] - qOff[q]), tL = (uint32_t) (tOff[t + 1] - tOff[t]);
    const sd_sw_result r = res[i];
    if (!bbAccepted(r, B, qL, tL)) return;
    atomicMax(&table[(size_t) q * B.nGroups + B.groupOf[t]], bbPack(r, tL, B.groupKey ? B.groupKey[t] : t));
}
// a pair that is neither an identity pair nor the first of its cell leaves no record and no backtrace
__global__ void __launch_bounds__(256)
k_best_keep(uint32_t nPairs, sd_sw_result *__restrict__ res, const uint8_t *__restrict__ ident, const uint32_t *__restrict__ pq,
            const uint32_t *__restrict__ pt, const uint64_t *__restrict__ qOff, const uint64_t *__restrict__ tOff, BestByGroup B,
            const unsigned long long *__restrict__ table, uint64_t *__restrict__ btLen) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs || ident[i]) return;
    const uint32_t q = pq[i], t = pt[i];
    const uint32_t qL = (uint32_t) (qOff[q + 1] - qOff[q]), tL = (uint32_t) (tOff[t + 1] - tOff[t]);
    const sd_sw_result r = res[i];
    const bool keep = bbAccepted(r, B, qL, tL) &&
                      (r.evalue < B.evalExact || table[(size_t) q * B.nGroups + B.groupOf[t]] == bbPack(r, tL, B.groupKey ? B.groupKey[t] : t));
    if (!keep) {
        res[i].btLen = 0;
        btLen[i] = 0;
    }
}

// compact mode: which records go back to the host
__global__ void __launch_bounds__(256)
k_accept(uint32_t nPairs, const sd_sw_result *__restrict__ res, const uint8_t *__restrict__ ident, uint8_t *__restrict__ acc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    const sd_sw_result &r = res[i];
    acc[i] = (ident[i] || (r.btLen > 0 && r.qStart >= 0 && r.tStart >= 0)) ? 1 : 0;
}
__global__ void __launch_bounds__(256)
k_accept_compact(uint32_t nPairs, const sd_sw_result *__restrict__ res, const uint8_t *__restrict__ acc,
                 const uint64_t *__restrict__ pos, sd_sw_result *__restrict__ outRes, uint32_t *__restrict__ outIdx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs || !acc[i]) return;
    outRes[pos[i]] = res[i];
    outIdx[pos[i]] = i;
}
// device-wide scans and the radix sort of the task keys: this library's own kernels (sd_scan_sort.h), 256-thread workgroups
// without inter-workgroup waiting -- they run between the score wavefronts of the other lanes
#include "sd_scan_sort.h"

// exclusive sum of n values (any unsigned width) into 64-bit offsets
template <typename T>
int devExclusiveScan(sd_ctx *ctx, const T *in, uint64_t *out, size_t n) {
    uint64_t *tmp = nullptr;
    SD_HIP(ctx, wsGet(ctx, "al.scantmp", sdScanTmpBytes(n) / sizeof(uint64_t) + 32, &tmp));
    SD_HIP(ctx, (sdScanLaunch<T, ScanSum64, false, uint64_t>(ctx->stream, in, out, n, tmp)));
    return SD_OK;
}
// running maximum (inclusive) of n 32-bit values
int devInclusiveMax(sd_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n) {
    uint32_t *tmp = nullptr;
    SD_HIP(ctx, wsGet(ctx, "al.scantmp32", sdScanTmpBytes(n) / sizeof(uint32_t) + 32, &tmp));
    SD_HIP(ctx, (sdScanLaunch<uint32_t, ScanMax32, true, uint32_t>(ctx->stream, in, out, n, tmp)));
    return SD_OK;
}

// stable sort of (key, value) pairs by the key bits [0, endBit)
int devSortPairs(sd_ctx *ctx, const uint32_t *kIn, uint32_t *kOut, const uint32_t *vIn, uint32_t *vOut, size_t n, int endBit) {
    if (n == 0) return SD_OK;
    uint32_t *kTmp = nullptr, *vTmp = nullptr, *cnt = nullptr;
    SD_HIP(ctx, wsGet(ctx, "al.sortK", n, &kTmp));
    SD_HIP(ctx, wsGet(ctx, "al.sortV", n, &vTmp));
    SD_HIP(ctx, wsGet(ctx, "al.sortCounts", sdRadixSortCountsBytes() / sizeof(uint32_t), &cnt));
    SD_HIP(ctx, sdRadixSortPairs(ctx->stream, kIn, vIn, kOut, vOut, kTmp, vTmp, (uint32_t) n, 0, endBit, cnt));
    return SD_OK;
}

__global__ void __launch_bounds__(256)
k_bound_need(uint32_t nPairs, const SwTask *__restrict__ tasks, const uint32_t *__restrict__ keys, uint64_t *__restrict__ need) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    // units of uint2: the packed kernel hands over 16 bytes per column (strips of 512 rows beyond 768 rows), the
    // int32 kernel 8 bytes (strips of 1024 rows)
    uint64_t v = 0;
    if (keys[i] != KEY_INVALID) {
        const uint32_t ci = keys[i] >> 10;
        if (scoreClassIsMulti(ci)) v = 2ull * (uint64_t) tasks[i].tL;
        else if (ci == N_SCORE_CLASSES - 1 && tasks[i].n > 1024) v = (uint64_t) tasks[i].tL;
    }
    need[i] = v;
}
__global__ void __launch_bounds__(256)
k_bound_apply(uint32_t nPairs, SwTask *__restrict__ tasks, const uint64_t *__restrict__ off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPairs) return;
    tasks[i].boundOff = off[i];
}

// ---- tasks of one query two by two (shared-profile kernels) --------------------------------------------------------
// key = class (5 bits) | query (17 bits) | target-length bucket (10 bits): one stable sort puts the tasks of a
// (class, query) run together, longest target first; inside a run of a packed class consecutive tasks form pairs, an
// odd last task stays alone.  Invalid tasks sort last (all ones).
constexpr uint32_t PAIR_NONE = 0xFFFFFFFFu;
// SD_SW_QUADS=0: wavefronts of two independent shared-profile pairs (round 3's form) instead of four tasks of one query
inline bool sdSwQuads() {
    static const bool on = !(getenv("SD_SW_QUADS") && atoi(getenv("SD_SW_QUADS")) == 0);
    return on;
}
// SD_SW_WAVES=2: two wavefronts (eight tasks of one query) per workgroup of the aligned kernel share the profile
inline int sdSwWaves() {
    static const int w = (sdSwQuads() && getenv("SD_SW_WAVES") && atoi(getenv("SD_SW_WAVES")) == 2) ? 2 : 1;
    return w;
}
constexpr int PAIR_QUERY_BITS = 17;
__global__ void __launch_bounds__(256)
k_pair_keys(uint32_t n, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ pairQ, uint32_t *__restrict__ key2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = keys[i];
    // class (6 bits) | query (17 bits) | target length code (9 bits): invalid entries sort behind the last class
    key2[i] = k == KEY_INVALID ? 0xFFFFFFFFu : ((k >> 10) << 26) | (pairQ[i] << 9) | ((k & 1023u) >> 1);
}
__global__ void __launch_bounds__(256)
k_pair_heads(uint32_t n, const uint32_t *__restrict__ keyS, uint32_t *__restrict__ headPos) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const bool head = p == 0 || (keyS[p] >> 9) != (keyS[p - 1] >> 9);
    headPos[p] = head ? p : 0u;
}
// leader[p] = pairs that start at sorted position p: 1 for every second task of a (class, query) run; with `quads` the run's
// last task adds one EMPTY pair when the run holds an odd number of pairs, so that every wavefront of four tasks (two pairs)
// stays inside one query (sw_score_pk_aligned_kernel<.., SHARE = 2>: one profile per wavefront)
__global__ void __launch_bounds__(256)
k_pair_leaders(uint32_t n, const uint32_t *__restrict__ keyS, const uint32_t *__restrict__ runStart, uint8_t *__restrict__ leader, int quads) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p > n) return;
    uint8_t v = 0;
    if (p < n && (keyS[p] >> 26) < FIRST_INT32_CLASS) {
        const uint32_t o = p - runStart[p];
        v = (o & 1u) == 0 ? 1 : 0;
        if (quads) {   // quads = pairs per workgroup (2 per wavefront): the run is padded to a multiple with empty pairs behind it
            const bool lastOfRun = p + 1 >= n || (keyS[p + 1] >> 9) != (keyS[p] >> 9);
            const uint32_t pairs = (o + 1 + 1) / 2;   // pairs of the run = ceil((o + 1) / 2)
            if (lastOfRun) v += (uint8_t) (((uint32_t) quads - pairs % (uint32_t) quads) % (uint32_t) quads);
        }
    }
    leader[p] = v;
}
__global__ void __launch_bounds__(256)
k_pair_emit(uint32_t n, const uint32_t *__restrict__ keyS, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ runStartOf,
            const uint64_t *__restrict__ pairIdx, uint32_t *__restrict__ order2) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || (keyS[p] >> 26) >= FIRST_INT32_CLASS || ((p - runStartOf[p]) & 1u)) return;   // the first task of every pair (packed classes) writes it
    const uint64_t w = pairIdx[p];
    order2[2 * w] = vals[p];
    const bool mate = p + 1 < n && (keyS[p + 1] >> 9) == (keyS[p] >> 9);
    order2[2 * w + 1] = mate ? vals[p + 1] : PAIR_NONE;
}
// pair-index boundaries of the packed classes: b[c] = number of pairs of classes < c
__global__ void k_pair_bounds(const uint32_t *__restrict__ keyS, uint32_t n, const uint64_t *__restrict__ pairIdx,
                              uint32_t *__restrict__ b, int nb) {
    const int c = threadIdx.x;
    if (c >= nb) return;
    const uint32_t want = (uint32_t) c << 26;
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (keyS[mid] < want) lo = mid + 1;
        else hi = mid;
    }
    b[c] = (uint32_t) pairIdx[lo];
}

// sort (keys,vals) -> order, read the class boundaries, launch one score kernel per RT class
int devRunScore(sd_ctx *ctx, uint32_t nPairs, const uint32_t *dKeys, const uint32_t *dVals, uint32_t *dKeysSorted,
                uint32_t *dOrder, uint32_t *dBounds, SwTask *dTasks, uint64_t *dScanA, uint64_t *dScanB,
                const sd_seqset *q, const sd_seqset *t, const int8_t *dMat, int go, int ge, int32_t *dOut,
                uint32_t *nValid, const uint32_t *dPairQ /* non-null: every task scans its whole query, pair them up */) {
    const unsigned grid = (nPairs + 255) / 256;
    int rc = SD_OK;
    uint32_t *dOrder2 = nullptr, *dPairBounds = nullptr;
    if (dPairQ) {
        // one sort serves both the pairing of the packed classes and the plain order of the int32 classes
        uint32_t *dKey2 = nullptr, *dVals2 = nullptr, *dHead = nullptr, *dRunStart = nullptr;
        uint64_t *dPairIdx = nullptr;
        uint8_t *dLeader = nullptr;
        SD_HIP(ctx, wsGet(ctx, "sp.key2", (size_t) nPairs, &dKey2));
        SD_HIP(ctx, wsGet(ctx, "sp.vals2", (size_t) nPairs, &dVals2));
        SD_HIP(ctx, wsGet(ctx, "sp.head", (size_t) nPairs, &dHead));
        SD_HIP(ctx, wsGet(ctx, "sp.runstart", (size_t) nPairs, &dRunStart));
        SD_HIP(ctx, wsGet(ctx, "sp.leader", (size_t) nPairs + 1, &dLeader));
        SD_HIP(ctx, wsGet(ctx, "sp.pairidx", (size_t) nPairs + 1, &dPairIdx));
        const int padPairs = sdSwQuads() ? 2 * sdSwWaves() : 0;   // pairs per workgroup of the shared-profile kernels
        const size_t order2Cap = (size_t) 2 * (size_t) std::max(padPairs, 2) * nPairs + 8;   // worst case: every run one task, padded to a workgroup
        SD_HIP(ctx, wsGet(ctx, "sp.order2", order2Cap, &dOrder2));
        SD_HIP(ctx, hipMemsetAsync(dOrder2, 0xFF, order2Cap * sizeof(uint32_t), ctx->stream));   // empty pairs: PAIR_NONE
        SD_HIP(ctx, wsGet(ctx, "sp.bounds", 64, &dPairBounds));
        hipLaunchKernelGGL(k_pair_keys, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dKeys, dPairQ, dKey2);
        rc = devSortPairs(ctx, dKey2, dKeysSorted, dVals, dVals2, nPairs, 32);
        if (rc != SD_OK) return rc;
        dOrder = dVals2;
        hipLaunchKernelGGL(k_pair_heads, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dKeysSorted, dHead);
        rc = devInclusiveMax(ctx, dHead, dRunStart, nPairs);
        if (rc != SD_OK) return rc;
        hipLaunchKernelGGL(k_pair_leaders, dim3((nPairs + 256) / 256), dim3(256), 0, ctx->stream, nPairs, dKeysSorted, dRunStart, dLeader,
                           padPairs);
        rc = devExclusiveScan(ctx, (const uint8_t *) dLeader, dPairIdx, (size_t) nPairs + 1);
        if (rc != SD_OK) return rc;
        hipLaunchKernelGGL(k_pair_emit, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dKeysSorted, dVals2, dRunStart, dPairIdx, dOrder2);
        hipLaunchKernelGGL(k_pair_bounds, dim3(1), dim3(64), 0, ctx->stream, dKeysSorted, nPairs, dPairIdx, dPairBounds,
                           (int) FIRST_INT32_CLASS + 1);
        hipLaunchKernelGGL(k_bounds, dim3(1), dim3(64), 0, ctx->stream, dKeysSorted, nPairs, 1u << 26, dBounds, (int) N_SCORE_CLASSES + 1);
    } else {
        rc = devSortPairs(ctx, dKeys, dKeysSorted, dVals, dOrder, nPairs, 16);
        if (rc != SD_OK) return rc;
        hipLaunchKernelGGL(k_bounds, dim3(1), dim3(64), 0, ctx->stream, dKeysSorted, nPairs, 1024u, dBounds, (int) N_SCORE_CLASSES + 1);
    }
    // strip hand-off workspace for queries longer than one 1024-row strip
    hipLaunchKernelGGL(k_bound_need, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dTasks, dKeys, dScanA);
    SD_HIP(ctx, hipMemsetAsync(dScanA + nPairs, 0, sizeof(uint64_t), ctx->stream));
    rc = devExclusiveScan(ctx, dScanA, dScanB, (size_t) nPairs + 1);
    if (rc != SD_OK) return rc;
    hipLaunchKernelGGL(k_bound_apply, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dTasks, dScanB);
    uint32_t hb[N_SCORE_CLASSES + 1], hpb[FIRST_INT32_CLASS + 1];
    uint64_t boundTotal = 0;
    if (dPairQ) SD_HIP(ctx, sdD2H(ctx, hpb, dPairBounds, sizeof(hpb)));
    SD_HIP(ctx, sdD2H(ctx, hb, dBounds, sizeof(hb)));
    SD_HIP(ctx, sdD2H(ctx, &boundTotal, dScanB + nPairs, sizeof(uint64_t)));
    SD_HIP(ctx, sdStreamSync(ctx));
    *nValid = hb[N_SCORE_CLASSES];
    uint2 *dBound = nullptr;
    SD_HIP(ctx, wsGet(ctx, "sw.bound", std::max<uint64_t>(boundTotal, 1), &dBound));
    for (uint32_t ci = 0; ci < N_SCORE_CLASSES; ci++) {
        const uint32_t begin = hb[ci], cnt = hb[ci + 1] - hb[ci];
        if (cnt == 0) continue;
        static const char *const pkNames[N_PK_CLASSES] = {"rt4x32", "rt6x32", "rt7x32", "rt8x32", "rt9x32", "rt10x32", "rt11x32", "rt12x32",
                                                          "rt8x64", "rt10x64", "rt12x64", "rt8x64s2", "rt8x64s3", "rt8x64sN"};
        static const char *const i32Names[4] = {"sw_score.rt4", "sw_score.rt8", "sw_score.rt16", "sw_score.rt32"};
        // SD_SW_LONG32=0: queries of 385 .. 768 rows keep the 64-lane general kernels (half the LDS per wavefront, twice the ramp)
        static const bool long32 = !(getenv("SD_SW_LONG32") && atoi(getenv("SD_SW_LONG32")) == 0);
        static const bool useAligned = !(getenv("SD_SW_ALIGNED") && atoi(getenv("SD_SW_ALIGNED")) == 0);
        const bool shared = dPairQ != nullptr && ci < FIRST_INT32_CLASS;
        const int alignedRT = (ci >= 1 && ci < FIRST_NARROW_MULTI) ? (int) ci + 4 : 0;
        // start-position tasks (two profiles per pair in LDS) beyond 384 rows and the A/B switches use the general kernels
        const bool aligned = alignedRT > 0 && useAligned && (alignedRT <= 12 || (shared && long32));
        // the general kernel by rows: class 0, the multi-strip classes, and the aligned classes when they are not taken
        int generalIdx = 0;
        if (ci >= FIRST_NARROW_MULTI && ci < FIRST_WIDE_CLASS) generalIdx = (int) FIRST_MULTI_PK + (int) (ci - FIRST_NARROW_MULTI);
        else if (alignedRT) generalIdx = alignedRT <= 6 ? 1 : (alignedRT <= 12 ? alignedRT - 5 : (alignedRT <= 16 ? 8 : (alignedRT <= 20 ? 9 : 10)));
        char name[48];
        if (aligned) snprintf(name, sizeof(name), "sw_score_pk.a_seg%d", alignedRT);
        else if (ci < FIRST_WIDE_CLASS) snprintf(name, sizeof(name), "sw_score_pk.%s", pkNames[generalIdx]);
        else if (ci < FIRST_INT32_CLASS) snprintf(name, sizeof(name), "sw_score_pk.w_%s", pkNames[ci - FIRST_WIDE_CLASS]);
        else snprintf(name, sizeof(name), "%s", i32Names[ci - FIRST_INT32_CLASS]);
        ProfScope ps(ctx, name);
        const uint32_t *ord = dOrder + begin;
        uint32_t nOrd = cnt;
        if (shared) {   // explicit pairs: two entries per pair of the class
            ord = dOrder2 + 2 * (size_t) hpb[ci];
            nOrd = 2 * (hpb[ci + 1] - hpb[ci]);
        }
#define SD_PK(RT, LW, MULTI, WIDE)                                                                                          \
    do {                                                                                                                   \
        if (shared) launchScorePk<RT, LW, MULTI, WIDE, true>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut, dBound);     \
        else launchScorePk<RT, LW, MULTI, WIDE, false>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut, dBound);           \
    } while (0)
#define SD_PKA(RT)                                                                                               \
    case RT:                                                                                                     \
        if (shared && quads && sdSwWaves() == 2) launchScorePkAligned<RT, 32, 2, 2>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut);  \
        else if (shared && quads && sdSwChain() > 1) launchScorePkChain<RT>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut);  \
        else if (shared && quads) launchScorePkAligned<RT, 32, 2>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut);  \
        else if (shared) launchScorePkAligned<RT, 32, 1>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut);      \
        else launchScorePkAligned<RT, 32, 0>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut);                  \
        break;
#define SD_PKA16(SEG)                                                                                            \
    case SEG: launchScorePkAligned<2 * SEG, 16, 1>(ctx, dTasks, ord, nOrd, q, t, dMat, go, ge, dOut); break;
#define SD_PK_CLASS(WIDE, IDX)                                                \
        switch (IDX) {                                                        \
            case 0: SD_PK(4, 32, false, WIDE); break;                         \
            case 1: SD_PK(6, 32, false, WIDE); break;                         \
            case 2: SD_PK(7, 32, false, WIDE); break;                         \
            case 3: SD_PK(8, 32, false, WIDE); break;                         \
            case 4: SD_PK(9, 32, false, WIDE); break;                         \
            case 5: SD_PK(10, 32, false, WIDE); break;                        \
            case 6: SD_PK(11, 32, false, WIDE); break;                        \
            case 7: SD_PK(12, 32, false, WIDE); break;                        \
            case 8: SD_PK(8, 64, false, WIDE); break;                         \
            case 9: SD_PK(10, 64, false, WIDE); break;                        \
            case 10: SD_PK(12, 64, false, WIDE); break;                       \
            default: SD_PK(8, 64, true, WIDE); break;                         \
        }
        // SD_SW_LW16: shared-profile tasks of up to this many rows per segment run on 16-lane groups (two segments per lane)
        static const int lw16Max = getenv("SD_SW_LW16") ? atoi(getenv("SD_SW_LW16")) : (sdSwQuads() ? 0 : 8);
        const bool quads = sdSwQuads();   // measured: beyond 8 rows per segment the LDS footprint (two segments per lane, eight tasks per wavefront) costs more occupancy than the shorter ramp returns
        if (aligned && shared && alignedRT <= lw16Max) {
            switch (alignedRT) {
                SD_PKA16(5) SD_PKA16(6) SD_PKA16(7) SD_PKA16(8) SD_PKA16(9) SD_PKA16(10) SD_PKA16(11) SD_PKA16(12)
                default: break;
            }
        } else if (aligned) {
            switch (alignedRT) {
                SD_PKA(5) SD_PKA(6) SD_PKA(7) SD_PKA(8) SD_PKA(9) SD_PKA(10) SD_PKA(11) SD_PKA(12) SD_PKA(13) SD_PKA(14)
                SD_PKA(15) SD_PKA(16) SD_PKA(17) SD_PKA(18) SD_PKA(19) SD_PKA(20) SD_PKA(21) SD_PKA(22) SD_PKA(23) SD_PKA(24)
                default: break;
            }
        } else if (ci < FIRST_WIDE_CLASS) {
            SD_PK_CLASS(false, generalIdx)
        } else if (ci < FIRST_INT32_CLASS) {
            SD_PK_CLASS(true, (int) (ci - FIRST_WIDE_CLASS))
        } else {
            switch (ci - FIRST_INT32_CLASS) {
                case 0: launchScoreIdx<4>(ctx, dTasks, ord, cnt, q, t, dMat, go, ge, dOut, dBound); break;
                case 1: launchScoreIdx<8>(ctx, dTasks, ord, cnt, q, t, dMat, go, ge, dOut, dBound); break;
                case 2: launchScoreIdx<16>(ctx, dTasks, ord, cnt, q, t, dMat, go, ge, dOut, dBound); break;
                default: launchScoreIdx<32>(ctx, dTasks, ord, cnt, q, t, dMat, go, ge, dOut, dBound); break;
            }
        }
#undef SD_PKA
#undef SD_PKA16
#undef SD_PK_CLASS
#undef SD_PK
    }
    SD_HIP(ctx, hipGetLastError());
    return SD_OK;
}

__global__ void __launch_bounds__(256)
k_cells(uint32_t nPairs, const SwTask *__restrict__ tasks, const uint32_t *__restrict__ keys, unsigned long long *__restrict__ acc) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    if (i < nPairs && keys[i] != KEY_INVALID) c = (unsigned long long) tasks[i].n * (unsigned long long) tasks[i].tL;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(acc, c);
}

}  // namespace

// =============================================================================================
int sdFail(sd_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->lastError = buf;
    else fprintf(stderr, "spacedust_gpu: %s\n", buf);
    return code;
}

extern "C" {

int sd_ctx_create(int device, sd_ctx **out) { return sd_ctx_create_prio(device, 0, out); }

int sd_ctx_create_prio(int device, int priority, sd_ctx **out) { return sd_ctx_create_masked(device, priority, 0, out); }

int sd_ctx_create_masked(int device, int priority, int reserveCUs, sd_ctx **out) {
    if (out == nullptr) return SD_EINVAL;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        fprintf(stderr, "spacedust_gpu: no HIP device visible (%s); this library has no CPU fallback\n",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return SD_ENODEVICE;
    }
    if (device < 0 || device >= count) return SD_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return SD_ENODEVICE;
    sd_ctx *c = new sd_ctx();
    c->device = device;
    int least = 0, greatest = 0;   // numerically: least priority >= greatest priority
    (void) hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int prio = priority < 0 ? greatest : (priority > 0 ? least : (least + greatest) / 2);
    hipError_t se = hipSuccess;
    (void) hipGetDeviceProperties(&c->prop, device);
    const int nCU = c->prop.multiProcessorCount;
    if (reserveCUs > 0 && nCU > 0 && reserveCUs < nCU) {
        // the queue's CU mask: the driver deals the bits round the XCDs (bit i -> XCD i mod 8), so clearing the top bits takes the
        // same number of CUs from every XCD.  (A masked stream has the default priority: the masked-stream call takes none.)
        std::vector<uint32_t> mask((size_t) (nCU + 31) / 32, 0u);
        for (int b = 0; b < nCU - reserveCUs; b++) mask[(size_t) b / 32] |= 1u << (b % 32);
        se = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t) mask.size(), mask.data());
    } else {
        se = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio);
    }
    if (se != hipSuccess) {
        delete c;
        return SD_EHIP;
    }
    // SD_SYNC=spin keeps hipStreamSynchronize's busy wait (lowest wake-up latency, one core per waiting thread)
    const char *syncMode = getenv("SD_SYNC");
    const bool spin = syncMode && strcmp(syncMode, "spin") == 0;
    (void) hipEventCreate(&c->evStart);
    (void) hipEventCreate(&c->evStop);
    if (!spin) (void) hipEventCreateWithFlags(&c->evSync, hipEventDisableTiming);
    (void) hipGetDeviceProperties(&c->prop, device);
    *out = c;
    return SD_OK;
}

void sd_ctx_destroy(sd_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    if (ctx->evStart) (void) hipEventDestroy(ctx->evStart);
    if (ctx->evStop) (void) hipEventDestroy(ctx->evStop);
    if (ctx->evSync) (void) hipEventDestroy(ctx->evSync);
    for (auto &pp : ctx->profPending) {
        (void) hipEventDestroy(pp.a);
        (void) hipEventDestroy(pp.b);
    }
    for (hipEvent_t e : ctx->evPool) (void) hipEventDestroy(e);
    if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
    for (auto &kv : ctx->ws) if (kv.second.p) (void) hipFree(kv.second.p);
    for (auto &b : ctx->pool) if (b.p) (void) hipFree(b.p);
    for (auto &kv : ctx->pinned) if (kv.second.p) (void) hipHostFree(kv.second.p);
    for (auto &b : ctx->bounce) if (b.p) (void) hipHostFree(b.p);
    delete ctx;
}

const char *sd_last_error(sd_ctx *ctx) { return ctx ? ctx->lastError.c_str() : "no context"; }

int sd_device_memory(sd_ctx *ctx, uint64_t *freeBytes, uint64_t *totalBytes) {
    if (!ctx) return SD_EINVAL;
    (void) hipSetDevice(ctx->device);
    size_t f = 0, t = 0;
    SD_HIP(ctx, hipMemGetInfo(&f, &t));
    if (freeBytes) *freeBytes = f;
    if (totalBytes) *totalBytes = t;
    return SD_OK;
}

int sd_workspace_report(sd_ctx *ctx, uint64_t *deviceBytes, uint64_t *pinnedBytes, char *buf, size_t cap) {
    if (!ctx) return SD_EINVAL;
    uint64_t dev = 0, pin = 0;
    std::vector<std::pair<size_t, std::string> > rows;
    for (auto &kv : ctx->ws) {
        dev += kv.second.bytes;
        rows.push_back(std::make_pair(kv.second.bytes, kv.first));
    }
    for (auto &kv : ctx->pinned) pin += kv.second.bytes;
    const size_t pooled = poolBytes(ctx);   // device buffers of destroyed sequence sets kept for the next one (poolPut)
    if (pooled) {
        dev += pooled;
        rows.push_back(std::make_pair(pooled, std::string("pool(seqset buffers)")));
    }
    if (deviceBytes) *deviceBytes = dev;
    if (pinnedBytes) *pinnedBytes = pin;
    if (buf && cap) {
        std::sort(rows.begin(), rows.end(), [](const std::pair<size_t, std::string> &a, const std::pair<size_t, std::string> &b) { return a.first > b.first; });
        std::string s;
        for (auto &r : rows) s += r.second + " " + std::to_string(r.first) + "\n";
        snprintf(buf, cap, "%s", s.c_str());
    }
    return SD_OK;
}

int sd_workspace_release(sd_ctx *ctx) {
    if (!ctx) return SD_EINVAL;
    (void) hipSetDevice(ctx->device);
    SD_HIP(ctx, sdStreamSync(ctx));
    for (auto &kv : ctx->ws)
        if (kv.second.p) (void) hipFree(kv.second.p);
    ctx->ws.clear();
    {
        std::lock_guard<std::mutex> lock(ctx->poolMutex);
        for (auto &b : ctx->pool)
            if (b.p) (void) hipFree(b.p);
        ctx->pool.clear();
    }
    ctx->biasTablesUploaded = false;
    return SD_OK;
}

// identity of a physical GPU across processes and nodes: FNV-1a of the host name and the device's PCI bus id
int sd_device_identity(int device, uint64_t *id) {
    if (!id) return SD_EINVAL;
    char host[256] = {0}, bus[64] = {0};
    (void) gethostname(host, sizeof(host) - 1);
    if (hipDeviceGetPCIBusId(bus, (int) sizeof(bus), device) != hipSuccess) return SD_ENODEVICE;
    uint64_t h = 1469598103934665603ull;
    for (const char *p = host; *p; p++) h = (h ^ (uint8_t) *p) * 1099511628211ull;
    h = (h ^ 0xFFu) * 1099511628211ull;
    for (const char *p = bus; *p; p++) h = (h ^ (uint8_t) *p) * 1099511628211ull;
    *id = h;
    return SD_OK;
}

int sd_device_name(sd_ctx *ctx, char *buf, size_t cap) {
    if (!ctx || !buf || cap == 0) return SD_EINVAL;
    snprintf(buf, cap, "%s (%s, %d CUs)", ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.multiProcessorCount);
    return SD_OK;
}

int sd_synchronize(sd_ctx *ctx) {
    SD_HIP(ctx, sdStreamSync(ctx));
    return SD_OK;
}

int sd_profile_enable(sd_ctx *ctx, int on) {
    ctx->profiling = on != 0;
    return SD_OK;
}
int sd_profile_reset(sd_ctx *ctx) {
    sdProfDrain(ctx, true);
    ctx->profile.clear();
    return SD_OK;
}
int sd_profile_get(sd_ctx *ctx, const char *name, double *totalMs, uint64_t *launches) {
    sdProfDrain(ctx, true);
    auto it = ctx->profile.find(name);
    if (it == ctx->profile.end()) {
        *totalMs = 0;
        *launches = 0;
        return SD_OK;
    }
    *totalMs = it->second.ms;
    *launches = it->second.launches;
    return SD_OK;
}
int sd_profile_names(sd_ctx *ctx, char *buf, size_t cap) {
    sdProfDrain(ctx, true);
    std::string s;
    for (auto &kv : ctx->profile) {
        if (!s.empty()) s += ",";
        s += kv.first;
    }
    snprintf(buf, cap, "%s", s.c_str());
    return SD_OK;
}

int sd_seqset_create(sd_ctx *ctx, const uint8_t *residues, const uint64_t *offsets, uint32_t n, const int8_t *swCompBias,
                     sd_seqset **out) {
    if (!ctx || !residues || !offsets || !out) return SD_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        if (offsets[i + 1] - offsets[i] > 65535)
            return sdFail(ctx, SD_EINVAL, "sequence %u has %llu residues; the limit is 65535 (--max-seq-len)", i,
                          (unsigned long long) (offsets[i + 1] - offsets[i]));
    (void) hipSetDevice(ctx->device);
    sd_seqset *s = new sd_seqset();
    s->ctx = ctx;
    s->n = n;
    s->total = offsets[n];
    s->hOff.assign(offsets, offsets + n + 1);
    s->hRes.assign(residues, residues + s->total);
    if (swCompBias) s->hBias.assign(swCompBias, swCompBias + s->total);
    else s->hBias.assign(s->total, 0);
    s->hMinBias.assign(n, 0);
    s->maxEntryAdd = 0;
    if (swCompBias) {   // (a set without a bias -- the target side, 9 * 10^8 residues at 1 000 proteomes -- has nothing to scan)
        int mx = 0;
#pragma omp parallel for schedule(static) reduction(max : mx)
        for (uint32_t i = 0; i < n; i++) {
            int m = 0;
            for (uint64_t x = offsets[i]; x < offsets[i + 1]; x++) {
                m = std::min(m, (int) swCompBias[x]);
                mx = std::max(mx, (int) swCompBias[x]);
            }
            s->hMinBias[i] = m;
        }
        s->maxEntryAdd = mx;
    }
    const size_t pad = 64;
    if (poolGet(ctx, s->total + pad, (void **) &s->dRes, &s->bRes) != hipSuccess ||
        poolGet(ctx, s->total + pad, (void **) &s->dBias, &s->bBias) != hipSuccess ||
        poolGet(ctx, (n + 1) * sizeof(uint64_t), (void **) &s->dOff, &s->bOff) != hipSuccess) {
        (void) hipGetLastError();
        sd_seqset_destroy(s);
        return sdFail(ctx, SD_ENOMEM, "sd_seqset_create: device allocation of %llu bytes failed", (unsigned long long) s->total);
    }
    // on the context's stream (the set is used there), one wait for the three copies; a failed upload hands the half-built set
    // back (host copies freed, device buffers to the pool) instead of leaking it
    hipError_t e = hipMemcpyAsync(s->dRes, s->hRes.data(), s->total, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = swCompBias ? hipMemcpyAsync(s->dBias, s->hBias.data(), s->total, hipMemcpyHostToDevice, ctx->stream)
                       : hipMemsetAsync(s->dBias, 0, s->total, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(s->dOff, s->hOff.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = sdStreamSync(ctx);
    if (e != hipSuccess) {
        (void) sdStreamSyncRaw(ctx);   // nothing may still read the host copies
        sd_seqset_destroy(s);
        *out = nullptr;
        return sdFail(ctx, SD_EHIP, "sd_seqset_create: upload failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return SD_OK;
}

// Profile queries (Sequence::mapProfile output, Sequence.cpp:241-292; ssw_init's PROFILE branch, StripedSmithWaterman.cpp:1238-1301):
// the query letters stand in for the residues (identity counting, computerBacktrace :558) and the int8 alignment
// profile [position][21] (profile score / 4, X column 0) replaces "matrix row + composition bias" in every kernel.
int sd_profileset_create(sd_ctx *ctx, const uint8_t *queryLetters, const uint64_t *offsets, uint32_t n,
                         const int8_t *alnProfile, sd_seqset **out) {
    if (!alnProfile) return SD_EINVAL;
    int rc = sd_seqset_create(ctx, queryLetters, offsets, n, nullptr, out);
    if (rc != SD_OK) return rc;
    sd_seqset *s = *out;
    const size_t bytes = (size_t) s->total * 21;
    if (poolGet(ctx, bytes + 64, (void **) &s->dProf, &s->bProf) != hipSuccess) {
        (void) hipGetLastError();
        sd_seqset_destroy(s);
        *out = nullptr;
        return sdFail(ctx, SD_ENOMEM, "sd_profileset_create: device allocation of %llu bytes failed", (unsigned long long) bytes);
    }
    s->hProf.assign(alnProfile, alnProfile + bytes);
    {
        hipError_t e = hipMemcpyAsync(s->dProf, s->hProf.data(), bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = sdStreamSync(ctx);
        if (e != hipSuccess) {
            (void) sdStreamSyncRaw(ctx);
            sd_seqset_destroy(s);
            *out = nullptr;
            return sdFail(ctx, SD_EHIP, "sd_profileset_create: upload failed: %s", hipGetErrorString(e));
        }
    }
    s->hProfBias.assign(n, 0);
    for (uint32_t i = 0; i < n; i++) {
        int m = 0;   // min over the 20 amino-acid columns (matSize = L * PROFILE_AA_SIZE, :1277-1279)
        for (uint64_t x = offsets[i]; x < offsets[i + 1]; x++)
            for (int a = 0; a < 20; a++) m = std::min(m, (int) alnProfile[x * 21 + a]);
        s->hProfBias[i] = -m;
    }
    {
        int mx = 1;
        for (size_t x = 0; x < bytes; x++) mx = std::max(mx, (int) alnProfile[x]);
        s->maxEntryAdd = mx;
    }
    return SD_OK;
}

void sd_seqset_destroy(sd_seqset *s) {
    if (!s) return;
    // (a set is destroyed before its context, include/spacedust_gpu.h: its buffers go back to that context's pool)
    poolPut(s->ctx, s->dRes, s->bRes);
    poolPut(s->ctx, s->dProf, s->bProf);
    poolPut(s->ctx, s->dBias, s->bBias);
    poolPut(s->ctx, s->dOff, s->bOff);
    poolPut(s->ctx, s->dGroupOf, s->bGroupOf);
    poolPut(s->ctx, s->dGroupKey, s->bGroupKey);
    delete s;
}

int sd_seqset_set_groups(sd_seqset *s, const uint32_t *groupOf, uint32_t nGroups, const uint32_t *keys) {
    if (!s || !groupOf || nGroups == 0) return SD_EINVAL;
    sd_ctx *ctx = s->ctx;
    (void) hipSetDevice(ctx->device);
    for (uint32_t i = 0; i < s->n; i++)
        if (groupOf[i] >= nGroups) return sdFail(ctx, SD_EINVAL, "sd_seqset_set_groups: sequence %u is in group %u of %u", i, groupOf[i], nGroups);
    // the selection key carries 28 bits of the DB key (or of the index): larger keys cannot break ties there, so such a set keeps
    // every accepted record for the caller's own selection (sd_sw_align_batch_best_by_group then returns what _compact_diag returns)
    bool keysFit = s->n <= 0x0FFFFFFFu;
    if (keys)
        for (uint32_t i = 0; i < s->n && keysFit; i++) keysFit = keys[i] < 0x0FFFFFFFu;
    if (!s->dGroupOf && poolGet(ctx, (size_t) (s->n + 1) * sizeof(uint32_t), (void **) &s->dGroupOf, &s->bGroupOf) != hipSuccess)
        return sdFail(ctx, SD_ENOMEM, "sd_seqset_set_groups: device allocation failed");
    if (keys && !s->dGroupKey && poolGet(ctx, (size_t) (s->n + 1) * sizeof(uint32_t), (void **) &s->dGroupKey, &s->bGroupKey) != hipSuccess)
        return sdFail(ctx, SD_ENOMEM, "sd_seqset_set_groups: device allocation failed");
    SD_HIP(ctx, hipMemcpyAsync(s->dGroupOf, groupOf, (size_t) s->n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (keys) SD_HIP(ctx, hipMemcpyAsync(s->dGroupKey, keys, (size_t) s->n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, sdStreamSync(ctx));
    s->nGroups = keysFit ? nGroups : 0;
    return SD_OK;
}

int sd_sw_last_cells(sd_ctx *ctx, uint64_t *f, uint64_t *r, uint64_t *t) {
    if (f) *f = ctx->cellsFwd;
    if (r) *r = ctx->cellsRev;
    if (t) *t = ctx->cellsTb;
    return SD_OK;
}

int sd_sw_score_batch(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                      uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, int lanes, int reverse,
                      const int32_t *qEnd, const int32_t *tEnd, int32_t *out) {
    if (!ctx || !par || !queries || !targets || (lanes != 16 && lanes != 32)) return SD_EINVAL;
    (void) hipSetDevice(ctx->device);
    DevBuf<int8_t> dMat;
    SD_HIP(ctx, dMat.alloc(441));
    SD_HIP(ctx, hipMemcpy(dMat.p, par->matrix, 441, hipMemcpyHostToDevice));
    std::vector<SwTask> tasks;
    tasks.reserve(nPairs);
    for (uint32_t i = 0; i < nPairs; i++) {
        const uint64_t qo = queries->hOff[pairQ[i]], to = targets->hOff[pairT[i]];
        const int qL = (int) (queries->hOff[pairQ[i] + 1] - qo), tL = (int) (targets->hOff[pairT[i] + 1] - to);
        SwTask tk;
        if (!reverse) {
            tk.qOff = qo; tk.tOff = to; tk.n = qL; tk.tL = tL; tk.qStep = 1; tk.tStep = 1;
        } else {
            if (qEnd[i] < 0 || tEnd[i] < 0 || qEnd[i] >= qL || tEnd[i] >= tL) return sdFail(ctx, SD_EINVAL, "bad end position for pair %u", i);
            tk.qOff = qo + qEnd[i]; tk.tOff = to + tEnd[i]; tk.n = qEnd[i] + 1; tk.tL = tEnd[i] + 1; tk.qStep = -1; tk.tStep = -1;
        }
        tk.segLen = std::max(1, (tk.n + lanes - 1) / lanes);
        tk.slot = i;
        tk.boundOff = 0;
        if (tk.n > 0 && tk.tL > 0) tasks.push_back(tk);
    }
    std::vector<int32_t> h;
    uint64_t cells = 0;
    int rc = runScoreTasks(ctx, tasks, queries, targets, dMat.p, par->gapOpen, par->gapExtend, h, nPairs, &cells);
    if (rc != SD_OK) return rc;
    if (reverse) ctx->cellsRev = cells; else ctx->cellsFwd = cells;
    for (uint32_t i = 0; i < nPairs; i++) {
        int v = h[3 * i], col = h[3 * i + 1], row = h[3 * i + 2];
        out[3 * i] = v;
        if (!reverse) {
            out[3 * i + 1] = col;
            out[3 * i + 2] = row;
        } else {
            out[3 * i + 1] = col < 0 ? -1 : tEnd[i] - col;   // scan position -> target index
            out[3 * i + 2] = row;                            // rows counted from qEnd downwards (reference: qStart = qEnd - read)
        }
    }
    return SD_OK;
}

// compactIdx != nullptr: only the pairs that reached a result worth reporting (identity pairs and pairs that were
// not stopped at a gate) are returned, out[x] being the record of pair compactIdx[x], x < *nCompact
static int alignBatchImpl(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                          uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint8_t *isIdentity,
                          sd_sw_result *out, char *btPool, uint64_t btCap, uint64_t *btUsed, uint32_t *compactIdx,
                          uint32_t *nCompact, const uint16_t *pairDiag = nullptr /* the prefilter's diagonal of every pair (nullable) */,
                          const float *bestBy = nullptr /* non-null: {seqIdThr, alnLenThr}: sd_sw_align_batch_best_by_group */) {
    if (!ctx || !par || !queries || !targets || !out) return SD_EINVAL;
    if (compactIdx && (!nCompact || par->swMode != 2)) return sdFail(ctx, SD_EINVAL, "the compact variants need swMode 2 (records with backtraces)");
    (void) hipSetDevice(ctx->device);
    sdD2HReset(ctx);   // (reads an earlier, failed call left pending)
    if (btUsed) *btUsed = 0;
    if (nCompact) *nCompact = 0;
    if (nPairs == 0) return SD_OK;
    const int go = par->gapOpen, ge = par->gapExtend;
    ctx->cellsFwd = ctx->cellsRev = ctx->cellsTb = 0;
    sd::Evaluer ev;
    sd::initEvaluer(ev, par->dbResidues);
    int matMin = 0;
    for (int i = 0; i < 441; i++) matMin = std::min(matMin, (int) par->matrix[i]);
    for (uint32_t i = 0; i < nPairs; i++)
        if (pairQ[i] >= queries->n || pairT[i] >= targets->n) return sdFail(ctx, SD_EINVAL, "pair %u out of range", i);
    if (targets->dProf) return sdFail(ctx, SD_EUNSUPPORTED, "profile targets are not implemented (profile queries are)");
    std::unique_ptr<HostScope> hs(new HostScope(ctx, "align.upload"));
    DevGateParams gp;
    gp.go = go; gp.ge = ge; gp.matMin = matMin; gp.swMode = par->swMode; gp.covMode = par->covMode; gp.covThr = par->covThr;
    {   // the int16 kernels are exact while no cell can exceed 32 767: rows x the largest entry of a query profile of this call
        int matMax = 0;
        for (int x = 0; x < 441; x++) matMax = std::max(matMax, (int) par->matrix[x]);
        const int maxEntry = std::max(1, queries->dProf ? queries->maxEntryAdd : matMax + queries->maxEntryAdd);
        gp.wideRowLimit = 32767 / maxEntry;
    }
    gp.evalThr = par->evalThr;
    gp.lambda = ev.lambda; gp.K = ev.K; gp.aI = ev.aI; gp.bI = ev.bI; gp.alphaI = ev.alphaI; gp.betaI = ev.betaI;
    gp.aJ = ev.aJ; gp.bJ = ev.bJ; gp.alphaJ = ev.alphaJ; gp.betaJ = ev.betaJ; gp.sigma = ev.sigma; gp.tau = ev.tau;
    gp.viThr = ev.viThr; gp.vjThr = ev.vjThr; gp.cThr = ev.cThr; gp.dbResidues = ev.dbResidues;

    const size_t N = nPairs;
    const unsigned grid = (unsigned) ((N + 255) / 256);
    int8_t *dMat = nullptr;
    uint32_t *dPQ = nullptr, *dPT = nullptr, *dKeys = nullptr, *dVals = nullptr, *dKeysS = nullptr, *dOrder = nullptr,
             *dBounds = nullptr, *dFwdKeys = nullptr, *dRevKeys = nullptr;
    uint8_t *dIdent = nullptr, *dWord = nullptr;
    int32_t *dMinBias = nullptr, *dOut32 = nullptr, *dOut16 = nullptr, *dOutRev = nullptr, *dTbRes = nullptr;
    int *dErr = nullptr;
    SwTask *dTasks = nullptr;
    TbTask *dTb = nullptr;
    sd_sw_result *dRes = nullptr;
    uint64_t *dScanA = nullptr, *dScanB = nullptr, *dDirBytes = nullptr, *dDirOff = nullptr, *dBtBytes = nullptr,
             *dBtOff = nullptr, *dBtLen = nullptr, *dDense = nullptr;
    unsigned long long *dCells = nullptr;
    SD_HIP(ctx, wsGet(ctx, "al.mat", 448, &dMat));
    SD_HIP(ctx, wsGet(ctx, "al.pq", N, &dPQ));
    SD_HIP(ctx, wsGet(ctx, "al.pt", N, &dPT));
    SD_HIP(ctx, wsGet(ctx, "al.ident", N, &dIdent));
    SD_HIP(ctx, wsGet(ctx, "al.word", N, &dWord));
    SD_HIP(ctx, wsGet(ctx, "al.keys", N, &dKeys));
    SD_HIP(ctx, wsGet(ctx, "al.vals", N, &dVals));
    SD_HIP(ctx, wsGet(ctx, "al.keyss", N, &dKeysS));
    SD_HIP(ctx, wsGet(ctx, "al.order", N, &dOrder));
    SD_HIP(ctx, wsGet(ctx, "al.fwdkeys", N, &dFwdKeys));
    SD_HIP(ctx, wsGet(ctx, "al.revkeys", N, &dRevKeys));
    SD_HIP(ctx, wsGet(ctx, "al.bounds", 64, &dBounds));
    SD_HIP(ctx, wsGet(ctx, "al.minbias", queries->n, &dMinBias));
    SD_HIP(ctx, wsGet(ctx, "al.out32", N * 3, &dOut32));
    SD_HIP(ctx, wsGet(ctx, "al.out16", N * 3, &dOut16));
    SD_HIP(ctx, wsGet(ctx, "al.outrev", N * 3, &dOutRev));
    SD_HIP(ctx, wsGet(ctx, "al.tbres", N * 2, &dTbRes));
    SD_HIP(ctx, wsGet(ctx, "al.err", 4, &dErr));
    SD_HIP(ctx, wsGet(ctx, "al.tasks", N, &dTasks));
    SD_HIP(ctx, wsGet(ctx, "al.tb", N, &dTb));
    SD_HIP(ctx, wsGet(ctx, "al.res", N, &dRes));
    SD_HIP(ctx, wsGet(ctx, "al.scana", N + 1, &dScanA));
    SD_HIP(ctx, wsGet(ctx, "al.scanb", N + 1, &dScanB));
    SD_HIP(ctx, wsGet(ctx, "al.dirbytes", N + 1, &dDirBytes));
    SD_HIP(ctx, wsGet(ctx, "al.diroff", N + 1, &dDirOff));
    SD_HIP(ctx, wsGet(ctx, "al.btbytes", N + 1, &dBtBytes));
    SD_HIP(ctx, wsGet(ctx, "al.btoff", N + 1, &dBtOff));
    SD_HIP(ctx, wsGet(ctx, "al.btlen", N + 1, &dBtLen));
    SD_HIP(ctx, wsGet(ctx, "al.dense", N + 1, &dDense));
    SD_HIP(ctx, wsGet(ctx, "al.cells", 4, &dCells));
    SD_HIP(ctx, hipMemcpyAsync(dMat, par->matrix, 441, hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(dPQ, pairQ, N * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemcpyAsync(dPT, pairT, N * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (isIdentity) SD_HIP(ctx, hipMemcpyAsync(dIdent, isIdentity, N, hipMemcpyHostToDevice, ctx->stream));
    else SD_HIP(ctx, hipMemsetAsync(dIdent, 0, N, ctx->stream));
    // byte-kernel bias (ssw_init, :1276-1287): |min(matrix)| + |min(0, composition bias)| for sequences, |min(0, profile)| for profiles
    SD_HIP(ctx, hipMemcpyAsync(dMinBias, queries->dProf ? queries->hProfBias.data() : queries->hMinBias.data(),
                               queries->n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dErr, 0, 4 * sizeof(int), ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dCells, 0, 4 * sizeof(unsigned long long), ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dOut32, 0, N * 3 * sizeof(int32_t), ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dOut16, 0, N * 3 * sizeof(int32_t), ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dOutRev, 0, N * 3 * sizeof(int32_t), ctx->stream));
    SD_HIP(ctx, hipMemsetAsync(dBtLen, 0, (N + 1) * sizeof(uint64_t), ctx->stream));

    // ---- pass 1: forward, byte-kernel lane structure
    hs.reset(new HostScope(ctx, "align.fwd32"));
    // the packed-int16 kernel's recurrence needs go >= ge (see sd_sw_pk.h); SD_SW_INT32=1 forces the int32 kernel
    const int usePk = (go >= ge && go < 1024 && ge >= 0 && getenv("SD_SW_INT32") == nullptr) ? 1 : 0;
    uint32_t nValid = 0;
    hipLaunchKernelGGL(k_make_fwd, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dPQ, dPT, dIdent, queries->dOff, targets->dOff,
                       dTasks, dFwdKeys, dVals, dRes, usePk);
    // pairs whose byte score saturates for certain (k_pre_word) skip this pass
    uint32_t *dPass1Keys = dFwdKeys;
    uint8_t *dPreWord = nullptr;
    if (pairDiag && usePk && !getenv("SD_SW_NO_PREWORD")) {
        uint16_t *dDiag = nullptr;
        SD_HIP(ctx, wsGet(ctx, "al.diag", N, &dDiag));
        SD_HIP(ctx, wsGet(ctx, "al.preword", N, &dPreWord));
        SD_HIP(ctx, wsGet(ctx, "al.pass1keys", N, &dPass1Keys));
        SD_HIP(ctx, hipMemcpyAsync(dDiag, pairDiag, N * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_pre_word, dim3(grid), dim3(256), 0, ctx->stream, nPairs, (const uint32_t *) dPQ, (const uint32_t *) dPT,
                           (const uint16_t *) dDiag, (const uint32_t *) dFwdKeys, (const uint64_t *) queries->dOff, (const uint64_t *) targets->dOff,
                           (const uint8_t *) queries->dRes, (const int8_t *) queries->dBias, (const uint8_t *) targets->dRes, (const int8_t *) dMat,
                           (const int8_t *) queries->dProf, (const int32_t *) dMinBias, queries->dProf ? 0 : matMin, dPreWord, dPass1Keys);
    }
    hipLaunchKernelGGL(k_cells, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dTasks, dPass1Keys, dCells + 0);
    const uint32_t *dShare = (getenv("SD_SW_NOSHARE") || queries->n >= (1u << PAIR_QUERY_BITS)) ? nullptr : dPQ;   // forward passes scan whole queries: pair by query
    int rc = devRunScore(ctx, nPairs, dPass1Keys, dVals, dKeysS, dOrder, dBounds, dTasks, dScanA, dScanB, queries, targets, dMat, go,
                         ge, dOut32, &nValid, dShare);
    if (rc != SD_OK) return rc;
    // ---- pass 2: saturated pairs again with the word kernel's 16-lane structure
    hs.reset(new HostScope(ctx, "align.fwd16"));
    hipLaunchKernelGGL(k_gate_word, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dPQ, dOut32, dMinBias, queries->dProf ? 0 : matMin, dTasks, dKeys,
                       dVals, dWord, dFwdKeys, usePk, gp.wideRowLimit, (const uint8_t *) dPreWord);
    hipLaunchKernelGGL(k_cells, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dTasks, dKeys, dCells + 0);
    rc = devRunScore(ctx, nPairs, dKeys, dVals, dKeysS, dOrder, dBounds, dTasks, dScanA, dScanB, queries, targets, dMat, go, ge,
                     dOut16, &nValid, dShare);
    if (rc != SD_OK) return rc;
    // ---- gates + pass 3: start positions
    hs.reset(new HostScope(ctx, "align.rev"));
    hipLaunchKernelGGL(k_gate_rev, dim3(grid), dim3(256), 0, ctx->stream, nPairs, gp, dPQ, dPT, queries->dOff, targets->dOff, dOut32,
                       dOut16, dWord, dFwdKeys, dTasks, dRevKeys, dVals, dRes, usePk);
    hipLaunchKernelGGL(k_cells, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dTasks, dRevKeys, dCells + 1);
    rc = devRunScore(ctx, nPairs, dRevKeys, dVals, dKeysS, dOrder, dBounds, dTasks, dScanA, dScanB, queries, targets, dMat, go, ge,
                     dOutRev, &nValid, nullptr);   // start-position tasks scan per-pair prefixes: no shared profile
    if (rc != SD_OK) return rc;
    // ---- pass 4: banded traceback, band doubling on the device
    hs.reset(new HostScope(ctx, "align.traceback"));
    hipLaunchKernelGGL(k_gate_tb, dim3(grid), dim3(256), 0, ctx->stream, nPairs, gp, dPQ, dPT, queries->dOff, targets->dOff, dOutRev,
                       dRevKeys, dTb, dKeys, dVals, dDirBytes, dBtBytes, dRes, dErr);
    SD_HIP(ctx, hipMemsetAsync(dBtBytes + N, 0, sizeof(uint64_t), ctx->stream));
    rc = devExclusiveScan(ctx, dBtBytes, dBtOff, N + 1);
    if (rc != SD_OK) return rc;
    uint64_t btScratch = 0;
    SD_HIP(ctx, sdD2H(ctx, &btScratch, dBtOff + N, sizeof(uint64_t)));
    SD_HIP(ctx, sdStreamSync(ctx));
    char *dBt = nullptr;
    SD_HIP(ctx, wsGet(ctx, "tb.bt", btScratch + 64, &dBt));
    // direction scratch of the LDS-band kernel (1 B per band cell, sized for the widest band of the task's class): a
    // quarter of the device memory at most (288 GB HBM: 72 GB), the rest belongs to the index and the prefilter workspace
    uint64_t SCRATCH_BUDGET = 16ull << 30;
    {
        size_t freeB = 0, totalB = 0;
        // a quarter of the device at most -- and no more than a third of what is free right now (the index, the prefilter
        // lanes and the other alignment lane hold the rest)
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && totalB > 0)
            SCRATCH_BUDGET = std::max<uint64_t>(1ull << 30, std::min<uint64_t>(totalB / 4, freeB / 3));
        if (const char *e = getenv("SD_TB_BUDGET")) SCRATCH_BUDGET = std::max<uint64_t>(1u << 20, strtoull(e, nullptr, 10));   // tests: force slices
    }
    uint32_t *dKeysRound = nullptr;
    SD_HIP(ctx, wsGet(ctx, "tb.keysRound", N + 1, &dKeysRound));
    bool tbPending = false, budgetRaised = false;
    for (int round = 0; round < 4096; round++) {   // band doublings, and slices of what the direction scratch holds at a time
        tbPending = true;
        SD_HIP(ctx, hipMemsetAsync(dDirBytes + N, 0, sizeof(uint64_t), ctx->stream));
        rc = devExclusiveScan(ctx, dDirBytes, dDirOff, N + 1);
        if (rc != SD_OK) return rc;
        hipLaunchKernelGGL(k_tb_round, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dKeys, dDirOff, dDirBytes, SCRATCH_BUDGET, dKeysRound);
        hipLaunchKernelGGL(k_tb_offsets, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dKeysRound, dDirOff, dBtOff, dTb);
        rc = devSortPairs(ctx, dKeysRound, dKeysS, dVals, dOrder, nPairs, 16);
        if (rc != SD_OK) return rc;
        hipLaunchKernelGGL(k_bounds, dim3(1), dim3(64), 0, ctx->stream, dKeysS, nPairs, 4096u, dBounds, (int) N_TB_CLASSES + 1);
        uint32_t hb[N_TB_CLASSES + 1];
        uint64_t dirTotal = 0;
        SD_HIP(ctx, sdD2H(ctx, hb, dBounds, sizeof(hb)));
        SD_HIP(ctx, sdD2H(ctx, &dirTotal, dDirOff + N, sizeof(uint64_t)));
        SD_HIP(ctx, sdStreamSync(ctx));
        if (hb[N_TB_CLASSES] == 0) {
            if (dirTotal == 0) {   // nothing left
                tbPending = false;
                break;
            }
            // tasks are pending but not even the first fits the budget (a global-band task of a long protein, or a budget halved
            // after an allocation failure): the budget grows to what a single task needs -- or the call fails, never a silent exit
            if (SCRATCH_BUDGET >= dirTotal) return sdFail(ctx, SD_EHIP, "traceback: pending tasks but an empty round (budget %llu, pending %llu bytes)",
                                                          (unsigned long long) SCRATCH_BUDGET, (unsigned long long) dirTotal);
            SCRATCH_BUDGET = std::min<uint64_t>(dirTotal, SCRATCH_BUDGET * 2);
            budgetRaised = true;
            continue;
        }
        if (getenv("SD_DEBUG_TB")) {
            fprintf(stderr, "[tb] round %d: narrow", round);
            for (int ci = 0; ci < N_TB_NARROW; ci++) fprintf(stderr, " %u", hb[ci + 1] - hb[ci]);
            fprintf(stderr, " lds %u %u %u global %u dir %.1f MB\n", hb[N_TB_NARROW + 1] - hb[N_TB_NARROW], hb[N_TB_NARROW + 2] - hb[N_TB_NARROW + 1],
                    hb[N_TB_NARROW + 3] - hb[N_TB_NARROW + 2], hb[N_TB_NARROW + 4] - hb[N_TB_NARROW + 3], dirTotal / 1e6);
        }
        // more direction bytes than the budget (long result lists of homologs, e.g. --max-seqs 4000 on a 10 000-proteome target):
        // this round runs the prefix that fits (k_tb_round), the rest follows in later rounds
        int8_t *dDir = nullptr;
        if (wsGet(ctx, "tb.dir", std::min<uint64_t>(dirTotal, SCRATCH_BUDGET) + 64, &dDir) != hipSuccess) {
            (void) hipGetLastError();
            if (SCRATCH_BUDGET <= (256ull << 20) || budgetRaised)   // (a raised budget is what one task needs: halving it would only come back here)
                return sdFail(ctx, SD_ENOMEM, "traceback direction scratch: out of device memory (%llu bytes for one round)",
                              (unsigned long long) std::min<uint64_t>(dirTotal, SCRATCH_BUDGET));
            SCRATCH_BUDGET /= 2;   // another lane took the memory meanwhile: smaller slices
            continue;
        }
        for (int ci = 0; ci < N_TB_NARROW; ci++) {
            const uint32_t begin = hb[ci], cnt = hb[ci + 1] - hb[ci];
            if (cnt == 0) continue;
            static const char *const tbNames[N_TB_NARROW] = {"sw_traceback.narrow128", "sw_traceback.narrow192", "sw_traceback.narrow256",
                                                             "sw_traceback.narrow320", "sw_traceback.narrow384", "sw_traceback.narrow512",
                                                             "sw_traceback.narrow640", "sw_traceback.narrow768", "sw_traceback.narrow1024"};
            ProfScope ps(ctx, tbNames[ci]);
            const int qCap = TB_NARROW_Q[ci], tCap = qCap + 16;
            const size_t ldsBytes = 2 * ((size_t) 16 * qCap + 2 * (size_t) qCap + tCap);
            if (queries->dProf)
                hipLaunchKernelGGL(sw_traceback_narrow_kernel<true>, dim3((cnt + 1) / 2), dim3(64), ldsBytes, ctx->stream, dTb, cnt,
                                   queries->dRes, queries->dBias, targets->dRes, dMat, go, ge, qCap, tCap, dBt, dTbRes, dOrder + begin,
                                   (const int8_t *) queries->dProf);
            else
                hipLaunchKernelGGL(sw_traceback_narrow_kernel<false>, dim3((cnt + 1) / 2), dim3(64), ldsBytes, ctx->stream, dTb, cnt,
                                   queries->dRes, queries->dBias, targets->dRes, dMat, go, ge, qCap, tCap, dBt, dTbRes, dOrder + begin,
                                   (const int8_t *) nullptr);
        }
        static const int ldsClass[3] = {128, 512, 2048};
        for (int ci = 0; ci < 3; ci++) {
            const uint32_t begin = hb[N_TB_NARROW + ci], cnt = hb[N_TB_NARROW + 1 + ci] - hb[N_TB_NARROW + ci];
            if (cnt == 0) continue;
            ProfScope ps(ctx, "sw_traceback.lds");
            const int ldsStride = ldsClass[ci] + 1;
            const size_t ldsBytes = (size_t) 2 * 3 * ldsStride * sizeof(int32_t);
            if (queries->dProf)
                hipLaunchKernelGGL(sw_traceback_kernel<true>, dim3((cnt + 1) / 2), dim3(64), ldsBytes, ctx->stream, dTb, cnt, queries->dRes,
                                   queries->dBias, targets->dRes, dMat, go, ge, ldsStride, dDir, dBt, dTbRes, dOrder + begin,
                                   ldsClass[ci] - 1, (const int8_t *) queries->dProf);
            else
                hipLaunchKernelGGL(sw_traceback_kernel<false>, dim3((cnt + 1) / 2), dim3(64), ldsBytes, ctx->stream, dTb, cnt, queries->dRes,
                                   queries->dBias, targets->dRes, dMat, go, ge, ldsStride, dDir, dBt, dTbRes, dOrder + begin,
                                   ldsClass[ci] - 1, (const int8_t *) nullptr);
            hipLaunchKernelGGL(sw_traceback_walk_kernel<false>, dim3((cnt + 256 / TBW_LANES - 1) / (256 / TBW_LANES)), dim3(256), 0, ctx->stream, dTb, cnt, queries->dRes,
                               targets->dRes, dDir, dBt, dTbRes, dOrder + begin);
        }
        {   // bands beyond the LDS classes: band arrays in global scratch, one attempt per round
            const uint32_t begin = hb[N_TB_NARROW + 3], cnt = hb[N_TB_NARROW + 4] - hb[N_TB_NARROW + 3];
            if (cnt) {
                ProfScope ps(ctx, "sw_traceback.global");
                if (queries->dProf)
                    hipLaunchKernelGGL((sw_traceback_kernel<true, true>), dim3((cnt + 1) / 2), dim3(64), 0, ctx->stream, dTb, cnt, queries->dRes,
                                       queries->dBias, targets->dRes, dMat, go, ge, INT_MAX, dDir, dBt, dTbRes, dOrder + begin, 0,
                                       (const int8_t *) queries->dProf);
                else
                    hipLaunchKernelGGL((sw_traceback_kernel<false, true>), dim3((cnt + 1) / 2), dim3(64), 0, ctx->stream, dTb, cnt, queries->dRes,
                                       queries->dBias, targets->dRes, dMat, go, ge, INT_MAX, dDir, dBt, dTbRes, dOrder + begin, 0,
                                       (const int8_t *) nullptr);
                hipLaunchKernelGGL(sw_traceback_walk_kernel<true>, dim3((cnt + 256 / TBW_LANES - 1) / (256 / TBW_LANES)), dim3(256), 0, ctx->stream, dTb, cnt, queries->dRes,
                                   targets->dRes, dDir, dBt, dTbRes, dOrder + begin);
            }
        }
        SD_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_tb_collect, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dKeys, dVals, dTb, dTbRes, dDirBytes, dBtLen,
                           dRes, dErr, 4 * 65536, dKeysRound);
    }
    if (tbPending) return sdFail(ctx, SD_EHIP, "traceback: tasks still pending after 4096 rounds");
    if (getenv("SD_DEBUG_TB")) {   // band statistics of the finished tasks
        std::vector<TbTask> hT(nPairs);
        std::vector<uint64_t> hL(nPairs);
        SD_HIP(ctx, hipMemcpy(hT.data(), dTb, (size_t) nPairs * sizeof(TbTask), hipMemcpyDeviceToHost));
        SD_HIP(ctx, hipMemcpy(hL.data(), dBtLen, (size_t) nPairs * sizeof(uint64_t), hipMemcpyDeviceToHost));
        uint64_t nDone = 0, init6 = 0, init6final6 = 0, init6final14 = 0, final14 = 0;
        for (uint32_t i = 0; i < nPairs; i++) {
            if (hL[i] == 0) continue;
            nDone++;
            const int ib = abs(hT[i].tLen - hT[i].qLen) + 1, fb = hT[i].band;
            if (ib <= 6) { init6++; if (fb <= 6) init6final6++; else if (fb <= 14) init6final14++; }
            if (fb <= 14) final14++;
        }
        {
            static const int lim[8] = {1, 2, 3, 4, 6, 8, 14, 1 << 30};
            uint64_t hi[8] = {0}, hf[8] = {0}, rowsF[8] = {0};
            for (uint32_t i = 0; i < nPairs; i++) {
                if (hL[i] == 0) continue;
                const int ib = abs(hT[i].tLen - hT[i].qLen) + 1, fb = hT[i].band;
                int a = 0, b = 0;
                while (ib > lim[a]) a++;
                while (fb > lim[b]) b++;
                hi[a]++;
                hf[b]++;
                rowsF[b] += (uint64_t) hT[i].qLen;
            }
            fprintf(stderr, "[tb] band <=1 <=2 <=3 <=4 <=6 <=8 <=14 >14: initial");
            for (int x = 0; x < 8; x++) fprintf(stderr, " %llu", (unsigned long long) hi[x]);
            fprintf(stderr, " | final");
            for (int x = 0; x < 8; x++) fprintf(stderr, " %llu", (unsigned long long) hf[x]);
            fprintf(stderr, " | rows by final");
            for (int x = 0; x < 8; x++) fprintf(stderr, " %llu", (unsigned long long) rowsF[x]);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[tb] done %llu: final band<=14 %llu; initial<=6 %llu of which final<=6 %llu, final 7..14 %llu\n",
                (unsigned long long) nDone, (unsigned long long) final14, (unsigned long long) init6, (unsigned long long) init6final6,
                (unsigned long long) init6final14);
    }
    // ---- besthitbyset on the device: only identity pairs and the first accepted alignment of every (query, target set) cell keep
    // their record and their backtrace (a table of queries x target sets; beyond 2^27 cells the call returns everything)
    if (bestBy && compactIdx && targets->dGroupOf && targets->nGroups > 0 && (uint64_t) queries->n * targets->nGroups <= (1ull << 27)) {
        BestByGroup B;
        B.groupOf = targets->dGroupOf;
        B.groupKey = targets->dGroupKey;
        B.nGroups = targets->nGroups;
        B.seqIdThr = bestBy[0];
        B.alnLenThr = (int) bestBy[1];
        B.covThr = par->covThr;
        B.covMode = par->covMode;
        B.evalThr = par->evalThr;
        B.evalExact = (getenv("SD_BEST_EXACT") && atof(getenv("SD_BEST_EXACT")) == 0.0) ? 0.0 : BB_EVAL_EXACT;
        unsigned long long *dTable = nullptr;
        const size_t cellsN = (size_t) queries->n * targets->nGroups;
        SD_HIP(ctx, wsGet(ctx, "al.besttable", cellsN, &dTable));
        SD_HIP(ctx, hipMemsetAsync(dTable, 0, cellsN * sizeof(unsigned long long), ctx->stream));
        ProfScope ps(ctx, "align_best_by_group");
        hipLaunchKernelGGL(k_best_mark, dim3(grid), dim3(256), 0, ctx->stream, nPairs, (const sd_sw_result *) dRes, (const uint8_t *) dIdent,
                           (const uint32_t *) dPQ, (const uint32_t *) dPT, (const uint64_t *) queries->dOff, (const uint64_t *) targets->dOff, B, dTable);
        hipLaunchKernelGGL(k_best_keep, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dRes, (const uint8_t *) dIdent, (const uint32_t *) dPQ,
                           (const uint32_t *) dPT, (const uint64_t *) queries->dOff, (const uint64_t *) targets->dOff, B,
                           (const unsigned long long *) dTable, dBtLen);
    }
    // ---- dense backtrace pool + results back to the host
    hs.reset(new HostScope(ctx, "align.download"));
    // (sd_sw_set_cigar_pool: the pool holds run-length text, its offsets are a scan of the text lengths)
    const bool cigarPool = ctx->cigarPool && btPool != nullptr;
    uint64_t *dCigLen = nullptr;
    if (cigarPool) {
        SD_HIP(ctx, wsGet(ctx, "al.ciglen", N + 1, &dCigLen));
        SD_HIP(ctx, hipMemsetAsync(dCigLen + nPairs, 0, (N + 1 - nPairs) * sizeof(uint64_t), ctx->stream));
        hipLaunchKernelGGL(k_bt_cigar<false>, dim3((nPairs + 3) / 4), dim3(256), 0, ctx->stream, nPairs, (const uint64_t *) dBtLen,
                           (const uint64_t *) nullptr, (const TbTask *) dTb, (const char *) dBt, (char *) nullptr, dCigLen, (sd_sw_result *) nullptr);
    }
    rc = devExclusiveScan(ctx, cigarPool ? dCigLen : dBtLen, dDense, N + 1);
    if (rc != SD_OK) return rc;
    uint64_t poolBytes = 0;
    int hErr[4] = {0, 0, 0, 0};
    unsigned long long hCells[4] = {0, 0, 0, 0};
    SD_HIP(ctx, sdD2H(ctx, &poolBytes, dDense + N, sizeof(uint64_t)));
    SD_HIP(ctx, sdD2H(ctx, hErr, dErr, sizeof(hErr)));
    SD_HIP(ctx, sdD2H(ctx, hCells, dCells, sizeof(hCells)));
    SD_HIP(ctx, sdStreamSync(ctx));
    if (hErr[0] == 1) return sdFail(ctx, SD_EMISMATCH, "Score of forward/backward SW differ (fatal in the reference, StripedSmithWaterman.cpp:466-473)");
    if (hErr[0] == 2) return sdFail(ctx, SD_EHIP, "Trace back error");
    ctx->cellsFwd = hCells[0];
    ctx->cellsRev = hCells[1];
    // identity pairs need pool space too (scoreIdentical backtraces are written by the host below)
    uint64_t identBytes = 0;
    if (isIdentity && btPool)
        for (uint32_t i = 0; i < nPairs; i++)
            if (isIdentity[i]) identBytes += cigarPool ? 8 : targets->hOff[pairT[i] + 1] - targets->hOff[pairT[i]];
    if (poolBytes > 0 && btPool == nullptr) return sdFail(ctx, SD_EINVAL, "swMode 2 needs a backtrace pool");
    if (poolBytes + identBytes > btCap && btPool) return sdFail(ctx, SD_ENOMEM, "backtrace pool too small");
    char *dPool = nullptr;
    SD_HIP(ctx, wsGet(ctx, "al.pool", poolBytes + 64, &dPool));
    if (poolBytes > 0 && cigarPool)
        hipLaunchKernelGGL(k_bt_cigar<true>, dim3((nPairs + 3) / 4), dim3(256), 0, ctx->stream, nPairs, (const uint64_t *) dBtLen,
                           (const uint64_t *) dDense, (const TbTask *) dTb, (const char *) dBt, dPool, (uint64_t *) nullptr, dRes);
    else if (poolBytes > 0)
        hipLaunchKernelGGL(k_bt_pack, dim3((nPairs + 3) / 4), dim3(256), 0, ctx->stream, nPairs, dBtLen, dDense, dTb, dBt, dPool,
                           (uint64_t) 0, dRes);
    // records to bring back: all of them, or (compact mode) only identity pairs and pairs that passed every gate
    uint32_t nRec = nPairs;
    sd_sw_result *dRecSrc = dRes;
    uint32_t *hIdx = nullptr;
    if (compactIdx) {
        uint8_t *dAcc = nullptr;
        uint64_t *dAccPos = nullptr;
        sd_sw_result *dResC = nullptr;
        uint32_t *dIdxC = nullptr;
        SD_HIP(ctx, wsGet(ctx, "al.acc", N + 1, &dAcc));
        SD_HIP(ctx, wsGet(ctx, "al.accpos", N + 1, &dAccPos));
        SD_HIP(ctx, wsGet(ctx, "al.resc", N, &dResC));
        SD_HIP(ctx, wsGet(ctx, "al.idxc", N, &dIdxC));
        hipLaunchKernelGGL(k_accept, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dRes, dIdent, dAcc);
        SD_HIP(ctx, hipMemsetAsync(dAcc + N, 0, 1, ctx->stream));
        {
            const int rcS = devExclusiveScan(ctx, (const uint8_t *) dAcc, dAccPos, N + 1);
            if (rcS != SD_OK) return rcS;
        }
        hipLaunchKernelGGL(k_accept_compact, dim3(grid), dim3(256), 0, ctx->stream, nPairs, dRes, dAcc, dAccPos, dResC, dIdxC);
        uint64_t nAcc = 0;
        SD_HIP(ctx, sdD2H(ctx, &nAcc, dAccPos + N, sizeof(uint64_t)));
        SD_HIP(ctx, sdStreamSync(ctx));
        nRec = (uint32_t) nAcc;
        dRecSrc = dResC;
        SD_HIP(ctx, pinGet(ctx, "al.hidx", std::max<uint32_t>(nRec, 1), &hIdx));
        SD_HIP(ctx, hipMemcpyAsync(hIdx, dIdxC, (size_t) nRec * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    sd_sw_result *hRes = nullptr;
    char *hPool = nullptr;
    SD_HIP(ctx, pinGet(ctx, "al.hres", std::max<uint32_t>(nRec, 1), &hRes));
    SD_HIP(ctx, pinGet(ctx, "al.hpool", poolBytes + 64, &hPool));
    SD_HIP(ctx, hipMemcpyAsync(hRes, dRecSrc, (size_t) nRec * sizeof(sd_sw_result), hipMemcpyDeviceToHost, ctx->stream));
    if (poolBytes > 0) SD_HIP(ctx, hipMemcpyAsync(hPool, dPool, poolBytes, hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, sdStreamSync(ctx));
    hs.reset(new HostScope(ctx, "align.finish"));
    {
        const int nth = std::max(1, std::min(omp_get_max_threads(), 64));
        const size_t blk = (poolBytes + nth - 1) / nth;
#pragma omp parallel for schedule(static) num_threads(nth)
        for (int t = 0; t < nth; t++) {
            const size_t a = std::min<size_t>(poolBytes, (size_t) t * blk), b = std::min<size_t>(poolBytes, a + blk);
            if (b > a) memcpy(btPool + a, hPool + a, b - a);
        }
    }
    uint64_t btPos = poolBytes;
    // exact E-values (the device value only served the gate), identity pairs, and the rare results the device
    // let through because they sat within 1e-9 of the E-value threshold
#pragma omp parallel for schedule(static)
    for (uint32_t x = 0; x < nRec; x++) {
        const uint32_t i = hIdx ? hIdx[x] : x;
        sd_sw_result r = hRes[x];
        if (!(isIdentity && isIdentity[i])) {
            const int qL = (int) (queries->hOff[pairQ[i] + 1] - queries->hOff[pairQ[i]]);
            if (r.tEnd != -1 && qL > 0) {
                // the device's value (same formula, device libm: relative error ~1e-15) stands for pairs far above the
                // threshold -- the reference computes but never reports those; everything that can be reported gets
                // the host's bit-exact value
                if (!(r.evalue > 2.0 * par->evalThr)) r.evalue = sd::computeEvalue(ev, r.score, qL);
                if (r.evalue > par->evalThr && r.qStart != -1) {   // decided by the host's exact value
                    r.qStart = -1; r.tStart = -1; r.identical = 0; r.btLen = 0; r.btOffset = 0; r.flags &= 1;
                }
            } else {
                r.evalue = 0.0;
            }
        }
        out[x] = r;
        if (compactIdx) compactIdx[x] = i;
    }
    if (nCompact) *nCompact = nRec;
    if (isIdentity) {
        for (uint32_t x = 0; x < nRec; x++) {
            const uint32_t i = hIdx ? hIdx[x] : x;
            if (!isIdentity[i]) continue;
            // scoreIdentical (StripedSmithWaterman.cpp:1675-1710), host side, O(L)
            const int L = (int) (targets->hOff[pairT[i] + 1] - targets->hOff[pairT[i]]);
            const int qL = (int) (queries->hOff[pairQ[i] + 1] - queries->hOff[pairQ[i]]);
            if (qL != L) return sdFail(ctx, SD_EINVAL, "scoreIdentical has different lengths for pair %u", i);
            const uint8_t *q = queries->hRes.data() + queries->hOff[pairQ[i]];
            const int8_t *cb = queries->hBias.data() + queries->hOff[pairQ[i]];
            const uint8_t *t = targets->hRes.data() + targets->hOff[pairT[i]];
            short score = 0;
            if (queries->dProf) {   // profile_word_linear[target letter][position] (:1700-1703, :1293-1298)
                const int8_t *pr = queries->hProf.data() + queries->hOff[pairQ[i]] * 21;
                for (int p = 0; p < L; p++) score += (short) pr[(size_t) p * 21 + t[p]];
            } else {
                for (int p = 0; p < L; p++) score += (short) (par->matrix[t[p] * 21 + q[p]] + cb[p]);
            }
            sd_sw_result &r = out[x];
            r.score = (int32_t) (uint32_t) (int) score;
            r.qStart = par->swMode == 0 ? -1 : 0;
            r.tStart = par->swMode == 0 ? -1 : 0;
            r.qEnd = L - 1; r.tEnd = L - 1; r.identical = L; r.btLen = L; r.flags = 0;
            r.evalue = sd::computeEvalue(ev, (double) (uint32_t) r.score, qL);
            r.btOffset = 0;
            if (btPool && cigarPool) {   // "<L>M"
                char txt[16];
                const int nTxt = snprintf(txt, sizeof(txt), "%dM", L);
                memcpy(btPool + btPos, txt, (size_t) nTxt);
                r.btOffset = btPos;
                r.flags = nTxt << 8;
                btPos += (uint64_t) nTxt;
            } else if (btPool) {
                memset(btPool + btPos, 'M', L);
                r.btOffset = btPos;
                btPos += L;
            }
        }
    }
    if (btUsed) *btUsed = btPos;
    ctx->d2hRecordBytes += (uint64_t) nRec * (sizeof(sd_sw_result) + (compactIdx ? sizeof(uint32_t) : 0));
    ctx->d2hPoolBytes += poolBytes;
    hs.reset();
    return SD_OK;
}

int sd_sw_set_cigar_pool(sd_ctx *ctx, int on) {
    if (!ctx) return SD_EINVAL;
    ctx->cigarPool = on != 0;
    return SD_OK;
}

int sd_sw_download_bytes(sd_ctx *ctx, uint64_t *recordBytes, uint64_t *poolBytes) {
    if (!ctx) return SD_EINVAL;
    if (recordBytes) *recordBytes = ctx->d2hRecordBytes;
    if (poolBytes) *poolBytes = ctx->d2hPoolBytes;
    return SD_OK;
}

int sd_sw_align_batch(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                      uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint8_t *isIdentity,
                      sd_sw_result *out, char *btPool, uint64_t btCap, uint64_t *btUsed) {
    return alignBatchImpl(ctx, par, queries, targets, nPairs, pairQ, pairT, isIdentity, out, btPool, btCap, btUsed, nullptr, nullptr);
}

int sd_sw_align_batch_compact(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                              uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint8_t *isIdentity,
                              uint32_t *outIdx, sd_sw_result *out, uint32_t *nOut, char *btPool, uint64_t btCap,
                              uint64_t *btUsed) {
    if (!outIdx || !nOut) return SD_EINVAL;
    return alignBatchImpl(ctx, par, queries, targets, nPairs, pairQ, pairT, isIdentity, out, btPool, btCap, btUsed, outIdx, nOut);
}

int sd_sw_align_batch_compact_diag(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                                   uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint16_t *pairDiag,
                                   const uint8_t *isIdentity, uint32_t *outIdx, sd_sw_result *out, uint32_t *nOut, char *btPool,
                                   uint64_t btCap, uint64_t *btUsed) {
    if (!outIdx || !nOut) return SD_EINVAL;
    return alignBatchImpl(ctx, par, queries, targets, nPairs, pairQ, pairT, isIdentity, out, btPool, btCap, btUsed, outIdx, nOut, pairDiag);
}

int sd_selftest_sort_pairs(sd_ctx *ctx, const uint32_t *keys, const uint32_t *vals, uint32_t n, int beginBit, int endBit, uint32_t *outKeys,
                           uint32_t *outVals) {
    if (!ctx || (n && (!keys || !vals || !outKeys || !outVals)) || beginBit < 0 || endBit > 32 || beginBit >= endBit) return SD_EINVAL;
    if (n == 0) return SD_OK;
    (void) hipSetDevice(ctx->device);
    DevBuf<uint32_t> kIn, vIn, kOut, vOut, kTmp, vTmp, cnt;
    SD_HIP(ctx, kIn.alloc(n));
    SD_HIP(ctx, vIn.alloc(n));
    SD_HIP(ctx, kOut.alloc(n));
    SD_HIP(ctx, vOut.alloc(n));
    SD_HIP(ctx, kTmp.alloc(n));
    SD_HIP(ctx, vTmp.alloc(n));
    SD_HIP(ctx, cnt.alloc(sdRadixSortCountsBytes() / sizeof(uint32_t)));
    SD_HIP(ctx, hipMemcpy(kIn.p, keys, (size_t) n * 4, hipMemcpyHostToDevice));
    SD_HIP(ctx, hipMemcpy(vIn.p, vals, (size_t) n * 4, hipMemcpyHostToDevice));
    SD_HIP(ctx, sdRadixSortPairs(ctx->stream, kIn.p, vIn.p, kOut.p, vOut.p, kTmp.p, vTmp.p, n, beginBit, endBit, cnt.p));
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SD_HIP(ctx, hipMemcpy(outKeys, kOut.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    SD_HIP(ctx, hipMemcpy(outVals, vOut.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_selftest_scan(sd_ctx *ctx, const uint32_t *in, uint32_t n, uint64_t *exclusiveSum, uint32_t *runningMax) {
    if (!ctx || !in || !exclusiveSum || n == 0) return SD_EINVAL;
    (void) hipSetDevice(ctx->device);
    DevBuf<uint32_t> dIn, dMax;
    DevBuf<uint64_t> dOut;
    SD_HIP(ctx, dIn.alloc((size_t) n + 1));
    SD_HIP(ctx, dOut.alloc((size_t) n + 1));
    SD_HIP(ctx, hipMemset(dIn.p, 0, ((size_t) n + 1) * 4));
    SD_HIP(ctx, hipMemcpy(dIn.p, in, (size_t) n * 4, hipMemcpyHostToDevice));
    int rc = devExclusiveScan(ctx, (const uint32_t *) dIn.p, dOut.p, (size_t) n + 1);
    if (rc != SD_OK) return rc;
    if (runningMax) {
        SD_HIP(ctx, dMax.alloc(n));
        rc = devInclusiveMax(ctx, dIn.p, dMax.p, n);
        if (rc != SD_OK) return rc;
    }
    SD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SD_HIP(ctx, hipMemcpy(exclusiveSum, dOut.p, ((size_t) n + 1) * 8, hipMemcpyDeviceToHost));
    if (runningMax) SD_HIP(ctx, hipMemcpy(runningMax, dMax.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    return SD_OK;
}

int sd_sw_align_batch_best_by_group(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                                    uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint16_t *pairDiag,
                                    const uint8_t *isIdentity, float seqIdThr, int32_t alnLenThr, uint32_t *outIdx, sd_sw_result *out,
                                    uint32_t *nOut, char *btPool, uint64_t btCap, uint64_t *btUsed) {
    if (!outIdx || !nOut || !targets) return SD_EINVAL;
    if (!targets->dGroupOf) return sdFail(ctx, SD_EINVAL, "sd_sw_align_batch_best_by_group: the target set has no groups (sd_seqset_set_groups)");
    const float bb[2] = {seqIdThr, (float) alnLenThr};
    return alignBatchImpl(ctx, par, queries, targets, nPairs, pairQ, pairT, isIdentity, out, btPool, btCap, btUsed, outIdx, nOut, pairDiag, bb);
}

int sd_sw_align_batch_hostpath(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                      uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint8_t *isIdentity,
                      sd_sw_result *out, char *btPool, uint64_t btCap, uint64_t *btUsed) {
    if (!ctx || !par || !queries || !targets || !out) return SD_EINVAL;
    if (queries->dProf || targets->dProf) return sdFail(ctx, SD_EUNSUPPORTED, "the host-orchestrated A/B path takes sequence sets only");
    (void) hipSetDevice(ctx->device);
    const int go = par->gapOpen, ge = par->gapExtend;
    ctx->cellsFwd = ctx->cellsRev = ctx->cellsTb = 0;
    sd::Evaluer ev;
    sd::initEvaluer(ev, par->dbResidues);
    int matMin = 0;
    for (int i = 0; i < 441; i++) matMin = std::min(matMin, (int) par->matrix[i]);
    DevBuf<int8_t> dMat;
    SD_HIP(ctx, dMat.alloc(441));
    SD_HIP(ctx, hipMemcpy(dMat.p, par->matrix, 441, hipMemcpyHostToDevice));
    if (btUsed) *btUsed = 0;
    uint64_t btPos = 0;

    std::unique_ptr<HostScope> hs(new HostScope(ctx, "align.init"));
    std::vector<int> qLv(nPairs), tLv(nPairs);
    for (uint32_t i = 0; i < nPairs; i++) {
        if (pairQ[i] >= queries->n || pairT[i] >= targets->n) return sdFail(ctx, SD_EINVAL, "pair %u out of range", i);
        qLv[i] = (int) (queries->hOff[pairQ[i] + 1] - queries->hOff[pairQ[i]]);
        tLv[i] = (int) (targets->hOff[pairT[i] + 1] - targets->hOff[pairT[i]]);
        sd_sw_result &r = out[i];
        r.score = 0; r.qStart = -1; r.qEnd = -1; r.tStart = -1; r.tEnd = -1; r.identical = 0; r.btLen = 0; r.flags = 0;
        r.evalue = 0.0; r.btOffset = 0;
    }
    // ---- identity pairs: scoreIdentical (StripedSmithWaterman.cpp:1675-1710), host side, O(L)
    for (uint32_t i = 0; i < nPairs; i++) {
        if (!(isIdentity && isIdentity[i])) continue;
        const int L = tLv[i];
        if (qLv[i] != L) return sdFail(ctx, SD_EINVAL, "scoreIdentical has different lengths for pair %u", i);
        const uint8_t *q = queries->hRes.data() + queries->hOff[pairQ[i]];
        const int8_t *cb = queries->hBias.data() + queries->hOff[pairQ[i]];
        const uint8_t *t = targets->hRes.data() + targets->hOff[pairT[i]];
        short score = 0;
        for (int p = 0; p < L; p++) score += (short) (par->matrix[t[p] * 21 + q[p]] + cb[p]);
        sd_sw_result &r = out[i];
        r.score = (int32_t) (uint32_t) (int) score;
        r.qStart = par->swMode == 0 ? -1 : 0;
        r.tStart = par->swMode == 0 ? -1 : 0;
        r.qEnd = L - 1; r.tEnd = L - 1; r.identical = L; r.btLen = L;
        r.evalue = sd::computeEvalue(ev, (double) (uint32_t) r.score, qLv[i]);
        if (btPool) {
            if (btPos + (uint64_t) L > btCap) return sdFail(ctx, SD_ENOMEM, "backtrace pool too small");
            memset(btPool + btPos, 'M', L);
            r.btOffset = btPos;
            btPos += L;
        }
    }
    hs.reset(new HostScope(ctx, "align.fwd32"));
    // ---- pass 1: forward, byte-kernel lane structure (32 lanes)
    std::vector<SwTask> tasks;
    auto fwdTask = [&](uint32_t i, int lanes) {
        SwTask tk;
        tk.qOff = queries->hOff[pairQ[i]]; tk.tOff = targets->hOff[pairT[i]];
        tk.n = qLv[i]; tk.tL = tLv[i]; tk.qStep = 1; tk.tStep = 1;
        tk.segLen = std::max(1, (tk.n + lanes - 1) / lanes);
        tk.slot = i; tk.boundOff = 0;
        return tk;
    };
    {
        // every pair gets a slot-ordered task; identity / empty pairs are dropped with a parallel compaction
        std::vector<uint32_t> keepIdx(nPairs);
        uint32_t nKeep = 0;
        for (uint32_t i = 0; i < nPairs; i++)
            if (!(isIdentity && isIdentity[i]) && qLv[i] > 0 && tLv[i] > 0) keepIdx[nKeep++] = i;
        tasks.resize(nKeep);
#pragma omp parallel for schedule(static)
        for (uint32_t x = 0; x < nKeep; x++) tasks[x] = fwdTask(keepIdx[x], 32);
    }
    std::vector<int32_t> h;
    int rc = runScoreTasks(ctx, tasks, queries, targets, dMat.p, go, ge, h, nPairs, &ctx->cellsFwd, "f32");
    if (rc != SD_OK) return rc;
    hs.reset(new HostScope(ctx, "align.fwd16"));
    // ---- pass 2: pairs whose byte score saturates (max + bias >= 255, :881,916,360-368) rerun with the
    //      word kernel's 16-lane structure
    std::vector<SwTask> tasks2;
    std::vector<uint8_t> word(nPairs, 0);
    for (size_t x = 0; x < tasks.size(); x++) {
        const uint32_t i = tasks[x].slot;
        const int bias = std::abs(matMin) + std::abs(queries->hMinBias[pairQ[i]]);
        if (h[3 * i] + bias >= 255) {
            word[i] = 1;
            tasks2.push_back(fwdTask(i, 16));
        }
    }
    std::vector<int32_t> h2;
    rc = runScoreTasks(ctx, tasks2, queries, targets, dMat.p, go, ge, h2, nPairs, &ctx->cellsFwd, "f16");
    if (rc != SD_OK) return rc;
    hs.reset(new HostScope(ctx, "align.gates"));
    // ---- gates after the score pass (:389-398); E-values in parallel, task list built serially
    std::vector<SwTask> rtasks;
    std::vector<uint8_t> goRev(nPairs, 0), hasTask(nPairs, 0);
    for (size_t x = 0; x < tasks.size(); x++) hasTask[tasks[x].slot] = 1;
    // pair-major (contiguous) so that threads do not share cache lines of out[] / goRev[]
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < nPairs; i++) {
        if (!hasTask[i]) continue;
        const int32_t *src = word[i] ? &h2[3 * i] : &h[3 * i];
        sd_sw_result &r = out[i];
        r.score = src[0];
        r.tEnd = src[1];
        r.qEnd = src[2];
        r.flags = word[i] ? 1 : 0;
        if (word[i] && r.tEnd == -1) r.tEnd = 0;   // the word kernel initialises end_ref to 0 (:962)
        if (r.score == 0) r.qEnd = 0;
        if (r.tEnd == -1) { r.evalue = 0.0; continue; }
        r.evalue = sd::computeEvalue(ev, r.score, qLv[i]);
        const bool lowE = r.evalue > par->evalThr;
        const float qCov = sd::computeCov(0, r.qEnd, qLv[i]), tCov = sd::computeCov(0, r.tEnd, tLv[i]);
        const bool lowCov = !sd::hasCoverage(par->covThr, par->covMode, qCov, tCov);
        if (par->swMode == 0 || lowE || lowCov) continue;
        goRev[i] = 1;
    }
    for (size_t x = 0; x < tasks.size(); x++) {
        const uint32_t i = tasks[x].slot;
        if (!goRev[i]) continue;
        const sd_sw_result &r = out[i];
        SwTask tk;
        tk.qOff = queries->hOff[pairQ[i]] + r.qEnd; tk.tOff = targets->hOff[pairT[i]] + r.tEnd;
        tk.n = r.qEnd + 1; tk.tL = r.tEnd + 1; tk.qStep = -1; tk.tStep = -1;
        tk.segLen = std::max(1, (tk.n + (word[i] ? 16 : 32) - 1) / (word[i] ? 16 : 32));
        tk.slot = i; tk.boundOff = 0;
        rtasks.push_back(tk);
    }
    hs.reset(new HostScope(ctx, "align.rev"));
    // ---- pass 3: start positions (reverse pass, :400-476)
    std::vector<int32_t> hr;
    rc = runScoreTasks(ctx, rtasks, queries, targets, dMat.p, go, ge, hr, nPairs, &ctx->cellsRev, "rev");
    if (rc != SD_OK) return rc;
    std::vector<TbTask> tb;
    for (size_t x = 0; x < rtasks.size(); x++) {
        const uint32_t i = rtasks[x].slot;
        sd_sw_result &r = out[i];
        if (hr[3 * i] != r.score) {
            r.flags |= 2;
            return sdFail(ctx, SD_EMISMATCH, "Score of forward/backward SW differ: %d %d (pair %u)", r.score, hr[3 * i], i);
        }
        r.tStart = r.tEnd - hr[3 * i + 1];
        r.qStart = r.qEnd - hr[3 * i + 2];
        const float qCov = sd::computeCov(r.qStart, r.qEnd, qLv[i]), tCov = sd::computeCov(r.tStart, r.tEnd, tLv[i]);
        const bool lowCov = !sd::hasCoverage(par->covThr, par->covMode, qCov, tCov);
        if (par->swMode == 1 || lowCov) continue;
        TbTask t;
        t.qAbs = queries->hOff[pairQ[i]] + r.qStart;
        t.tAbs = targets->hOff[pairT[i]] + r.tStart;
        t.qLen = r.qEnd - r.qStart + 1;
        t.tLen = r.tEnd - r.tStart + 1;
        t.score = r.score;
        t.band = std::abs(t.tLen - t.qLen) + 1;
        t.maxv = 0;
        t.slot = i;
        tb.push_back(t);
    }
    hs.reset(new HostScope(ctx, "align.traceback"));
    // ---- pass 4: banded traceback in chunks that fit the scratch budget
    if (!tb.empty() && btPool == nullptr) return sdFail(ctx, SD_EINVAL, "swMode 2 needs a backtrace pool");
    const uint64_t SCRATCH_BUDGET = 8ull << 30;
    auto widthClass = [](int band) { int w = band * 2 + 3; return w <= 127 ? 128 : (w <= 511 ? 512 : (w <= 2047 ? 2048 : 0)); };   // 0: global-band class
    std::vector<TbTask> pending = tb;
    while (!pending.empty()) {
        for (size_t x = 0; x < pending.size(); x++)
            if (pending[x].band > 4 * 65536) return sdFail(ctx, SD_EHIP, "Trace back error (pair %u)", pending[x].slot);
        {   // order: LDS width class, then work (row chunks x rows) descending -- stable counting sort
            auto keyOf = [&](const TbTask &t) {
                const int c = widthClass(t.band);
                const int ci = c == 128 ? 0 : (c == 512 ? 1 : (c == 2048 ? 2 : 3));
                const uint64_t work = (uint64_t) ((2 * t.band + 1 + 31) / 32) * (uint64_t) t.qLen;
                return (uint32_t) ci * 4096u + (uint32_t) (4095 - std::min<uint64_t>(work >> 3, 4095));
            };
            std::vector<uint32_t> cnt(4 * 4096 + 1, 0);
            for (size_t i = 0; i < pending.size(); i++) cnt[keyOf(pending[i]) + 1]++;
            for (size_t i = 1; i < cnt.size(); i++) cnt[i] += cnt[i - 1];
            std::vector<TbTask> sortedTb(pending.size());
            for (size_t i = 0; i < pending.size(); i++) sortedTb[cnt[keyOf(pending[i])]++] = pending[i];
            pending.swap(sortedTb);
        }
        std::vector<TbTask> next;
        size_t pos = 0;
        while (pos < pending.size()) {
            uint64_t nDir = 0, nBt = 0;
            size_t end = pos;
            const int cls = widthClass(pending[pos].band);
            while (end < pending.size() && widthClass(pending[end].band) == cls) {
                TbTask &t = pending[end];
                const uint64_t width_d = (uint64_t) t.band * 2 + 1;
                const uint64_t ad = ((cls ? 0 : tbGlobalIntBytes(t.band)) + width_d * (uint64_t) t.qLen + 31) & ~15ull, ab = (uint64_t) t.qLen + t.tLen + 2;
                if (end > pos && (nDir + ad) > SCRATCH_BUDGET) break;
                t.intOff = 0; t.dirOff = nDir; t.btOff = nBt;
                nDir += ad; nBt += ab;
                ctx->cellsTb += width_d * (uint64_t) t.qLen;
                end++;
            }
            const uint32_t cnt = (uint32_t) (end - pos);
            struct { TbTask *p; } dT;
            struct { int32_t *p; } dRes;
            struct { int8_t *p; } dDir;
            struct { char *p; } dBt;
            if (wsGet(ctx, "tb.tasks", cnt, &dT.p) != hipSuccess || wsGet(ctx, "tb.dir", nDir, &dDir.p) != hipSuccess ||
                wsGet(ctx, "tb.bt", nBt, &dBt.p) != hipSuccess || wsGet(ctx, "tb.res", (size_t) cnt * 2, &dRes.p) != hipSuccess)
                return sdFail(ctx, SD_ENOMEM, "traceback scratch allocation failed (%llu bytes)", (unsigned long long) nDir);
            SD_HIP(ctx, hipMemcpyAsync(dT.p, &pending[pos], cnt * sizeof(TbTask), hipMemcpyHostToDevice, ctx->stream));
            {
                ProfScope ps(ctx, "sw_traceback");
                const int ldsStride = cls + 1;
                const size_t ldsBytes = (size_t) 2 * 3 * ldsStride * sizeof(int32_t);
                if (cls)
                    hipLaunchKernelGGL(sw_traceback_kernel<false>, dim3((cnt + 1) / 2), dim3(64), ldsBytes, ctx->stream, dT.p, cnt,
                                       queries->dRes, queries->dBias, targets->dRes, dMat.p, go, ge, ldsStride, dDir.p, dBt.p, dRes.p,
                                       (const uint32_t *) nullptr, 0, (const int8_t *) nullptr);
                else
                    hipLaunchKernelGGL((sw_traceback_kernel<false, true>), dim3((cnt + 1) / 2), dim3(64), 0, ctx->stream, dT.p, cnt,
                                       queries->dRes, queries->dBias, targets->dRes, dMat.p, go, ge, INT_MAX, dDir.p, dBt.p, dRes.p,
                                       (const uint32_t *) nullptr, 0, (const int8_t *) nullptr);
                if (cls)
                    hipLaunchKernelGGL(sw_traceback_walk_kernel<false>, dim3((cnt + 256 / TBW_LANES - 1) / (256 / TBW_LANES)), dim3(256), 0, ctx->stream, dT.p, cnt, queries->dRes,
                                       targets->dRes, dDir.p, dBt.p, dRes.p, (const uint32_t *) nullptr);
                else
                    hipLaunchKernelGGL(sw_traceback_walk_kernel<true>, dim3((cnt + 256 / TBW_LANES - 1) / (256 / TBW_LANES)), dim3(256), 0, ctx->stream, dT.p, cnt, queries->dRes,
                                       targets->dRes, dDir.p, dBt.p, dRes.p, (const uint32_t *) nullptr);
            }
            SD_HIP(ctx, hipGetLastError());
            TbTask *back = nullptr;
            char *hbt = nullptr;
            int32_t *hres = nullptr;
            SD_HIP(ctx, pinGet(ctx, "tb.back", cnt, &back));
            SD_HIP(ctx, pinGet(ctx, "tb.hbt", nBt, &hbt));
            SD_HIP(ctx, pinGet(ctx, "tb.hres", (size_t) cnt * 2, &hres));
            SD_HIP(ctx, hipMemcpyAsync(back, dT.p, cnt * sizeof(TbTask), hipMemcpyDeviceToHost, ctx->stream));
            SD_HIP(ctx, hipMemcpyAsync(hbt, dBt.p, nBt, hipMemcpyDeviceToHost, ctx->stream));
            SD_HIP(ctx, hipMemcpyAsync(hres, dRes.p, (size_t) cnt * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            SD_HIP(ctx, sdStreamSync(ctx));
            // pool offsets serially (cheap), payload copies in parallel
            std::vector<uint64_t> dst(cnt, 0);
            for (uint32_t x = 0; x < cnt; x++) {
                const TbTask &t = back[x];
                const int len = hres[2 * x];
                if (len == -2) {
                    TbTask nt = t;
                    nt.band = t.band;   // doubled by the kernel
                    next.push_back(nt);
                } else if (len < 0) {
                    return sdFail(ctx, SD_EHIP, "Trace back error for pair %u", t.slot);
                } else {
                    if (btPos + (uint64_t) len > btCap) return sdFail(ctx, SD_ENOMEM, "backtrace pool too small");
                    dst[x] = btPos;
                    btPos += len;
                }
            }
#pragma omp parallel for schedule(static)
            for (uint32_t x = 0; x < cnt; x++) {
                const int len = hres[2 * x];
                if (len < 0) continue;
                const TbTask &t = back[x];
                sd_sw_result &r = out[t.slot];
                memcpy(btPool + dst[x], hbt + t.btOff + ((size_t) (t.qLen + t.tLen + 2) - (size_t) len), len);
                r.btOffset = dst[x];
                r.btLen = len;
                r.identical = hres[2 * x + 1];
            }
            pos = end;
        }
        pending.swap(next);
    }
    hs.reset();
    if (btUsed) *btUsed = btPos;
    return SD_OK;
}

}  // extern "C"
