// Packed-int16 score pass: the kernel the forward and start-position passes of sd_sw_align_batch run on
// whenever the score provably fits (byte-kernel lane structure, StripedSmithWaterman.cpp:639-915).
//
// Mapping onto a CDNA4 wavefront
//   * a 32-lane half wavefront owns TWO (query,target) pairs: pair A lives in the low 16 bits of every
//     DP register, pair B in the high 16 bits, so each v_pk_* instruction advances two cells;
//   * lane l owns RT consecutive query rows of both pairs and walks the target one column per step,
//     one column behind lane l-1 (systolic anti-diagonal): the only traffic between lanes is the
//     (H, F_first, F_lazy, residue) hand-off, done with DPP wave_shr:1 moves -- no LDS, no bpermute;
//   * the int8 query profiles (21 residues + 1 neutral row) sit in LDS, read as RT/4 dwords per pair and
//     step; profile bytes are sign-extended and added to the diagonal in one SDWA v_add_u16 per cell;
//   * target residues are fetched 32 columns at a time (coalesced) and fed to lane 0 with v_readlane.
// DP state per row pair: H (for the next column's diagonal) and E, both in VGPRs.  The reference's
// recurrence (see sw_score_kernel) is evaluated with unsigned saturating subtractions, which supply every
// max(...,0) of the original for free, and with F_first' = max(F_first - ge, Hpre - go), which equals
// max(F_first - ge, G - go) whenever go >= ge (G - go = max(Hpre - go, F_first - go) <= the former).
// The column maximum carries the row in its low five bits (g*32 + 31-r, saturating): exact while g < 1024,
// and a saturated result (1023) is above every byte-kernel overflow threshold, so such pairs are rerun by
// the caller with the int32 kernel exactly like the reference reruns them with its word kernel.
#ifndef SD_SW_PK_H
#define SD_SW_PK_H

#include <hip/hip_runtime.h>
#include <cstdint>

namespace sdpk {

struct SwTask {
    uint64_t qOff;     // absolute index (into the residue array) of the first scanned query residue
    uint64_t tOff;     // absolute index of the first scanned target residue
    int32_t n;         // query rows used
    int32_t tL;        // target columns scanned
    int32_t qStep;     // +1 forward, -1 reverse pass
    int32_t tStep;
    int32_t segLen;    // ceil(n / lanes) of the reference kernel being reproduced
    uint32_t slot;     // output slot
    uint64_t boundOff; // offset (in uint2 units) into the strip boundary workspace (multi-strip tasks only)
};

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pkMax(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pkSubSat(uint32_t a, uint32_t b) {   // unsigned, saturating at 0
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pkSub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2) (__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pkAdd(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2) (__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pkLshr(uint32_t a, int s) {
    return __builtin_bit_cast(uint32_t, (u16x2) (__builtin_bit_cast(u16x2, a) >> (u16x2) ((unsigned short) s)));
}
__device__ __forceinline__ uint32_t pkAshr(uint32_t a, int s) {
    return __builtin_bit_cast(uint32_t, (s16x2) (__builtin_bit_cast(s16x2, a) >> (s16x2) ((short) s)));
}
// g*32 + code in both halves, saturating to int16
__device__ __forceinline__ uint32_t pkRowCode(uint32_t g, int code) {
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, 32, %2 op_sel_hi:[1,0,0] clamp" : "=v"(r) : "v"(g), "s"(code));
    return r;
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

// h_k = d_k + sext(byte k of pa) in the low half, d_k + sext(byte k of pb) in the high half, k = 0..3.
// The eight SDWA adds are ordered so that no instruction reads the destination of its predecessor (gfx950
// needs one wait state after a dst_sel write); the closing s_nop covers the first consumer.
__device__ __forceinline__ void addProfile4(uint32_t pa, uint32_t pb, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3,
                                            uint32_t &h0, uint32_t &h1, uint32_t &h2, uint32_t &h3) {
    asm("v_add_u16_sdwa %0, %4, sext(%8) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0\n\t"
        "v_add_u16_sdwa %1, %5, sext(%8) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_1\n\t"
        "v_add_u16_sdwa %2, %6, sext(%8) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2\n\t"
        "v_add_u16_sdwa %3, %7, sext(%8) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_3\n\t"
        "v_add_u16_sdwa %0, %4, sext(%9) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_0\n\t"
        "v_add_u16_sdwa %1, %5, sext(%9) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_1\n\t"
        "v_add_u16_sdwa %2, %6, sext(%9) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2\n\t"
        "v_add_u16_sdwa %3, %7, sext(%9) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_3\n\t"
        "s_nop 0"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3)
        : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(pa), "v"(pb));
}

__device__ __forceinline__ uint32_t dppShr1(uint32_t v) {
    // lane i receives lane i-1 (whole wavefront); lane 0 receives 0
    return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// value (wave-uniform) into one lane of a VGPR
template <int LANE>
__device__ __forceinline__ uint32_t writeLane(uint32_t value, uint32_t old) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(value), "n"(LANE));
    return old;
}
__device__ __forceinline__ uint32_t readLane(uint32_t v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t) __builtin_amdgcn_readlane((int) v, lane);
#else
    (void) lane;
    return v;
#endif
}

constexpr int PK_NEUTRAL = 21;   // profile row of -64s: columns outside the target

// RT rows per lane and pair, LW lanes per pair of tasks: LW = 32 puts two pairs (four tasks) on a wavefront,
// LW = 64 one pair.  ROWS = LW*RT query rows per strip.  MULTI: queries longer than one strip.
// WIDE: the column maximum is tracked per task in a 32-bit (value << 16 | 31-r) code instead of the packed
// five-bit one, which lifts the g < 1024 limit to the int16 range (scores of the reference's word kernel).
// SHARED: the two tasks of a pair have the same query (same rows, same profile): one profile per pair in LDS
// instead of two, which doubles the wavefronts a CU can hold; `order` then lists pairs explicitly, 0xFFFFFFFF
// standing for "no second task".
template <int RT, int LW, bool MULTI, bool WIDE, bool SHARED>
__global__ void __launch_bounds__(64)
sw_score_pk_kernel(const SwTask *__restrict__ tasks, uint32_t nTasks, const uint8_t *__restrict__ qRes,
                   const int8_t *__restrict__ qBias, const uint8_t *__restrict__ tRes, const int8_t *__restrict__ mat,
                   int go, int ge, int32_t *__restrict__ out, uint2 *__restrict__ boundary,
                   const uint32_t *__restrict__ order,
                   const int8_t *__restrict__ qProf /* profile queries: int8 [position][21] rows replace mat[q][.] + bias */) {
    constexpr int ROWS = LW * RT;
    constexpr int WORDS = (RT + 3) / 4;    // profile dwords per lane, pair and residue (RT need not be a multiple of 4:
    constexpr int RTP = 4 * WORDS;         //  the bytes past RT in the last dword are padding)
    constexpr int PSTRIDE = LW * WORDS;    // dwords per residue row of a profile
    constexpr int NGRP = 64 / LW;
    constexpr int NT = 2 * NGRP;   // tasks per wavefront
    __shared__ uint32_t prof[SHARED ? NGRP : NT][22][PSTRIDE];
    __shared__ int8_t smat[441];
    for (int i = threadIdx.x; i < 441; i += 64) smat[i] = mat[i];
    __syncthreads();

    const int lane = threadIdx.x;
    const int grp = lane / LW, l = lane % LW;
    // the tasks of this wavefront (uniform loads); tasks beyond the end are empty
    SwTask tk[NT];
#pragma unroll
    for (int x = 0; x < NT; x++) {
        const uint32_t id = blockIdx.x * NT + x;
        const uint32_t tid = id < nTasks ? (order ? order[id] : id) : 0xFFFFFFFFu;
        if (tid != 0xFFFFFFFFu) {
            tk[x] = tasks[tid];
        } else {
            tk[x].n = 0; tk[x].tL = 0; tk[x].qOff = 0; tk[x].tOff = 0; tk[x].qStep = 1; tk[x].tStep = 1; tk[x].segLen = 1;
            tk[x].slot = 0; tk[x].boundOff = 0;
        }
    }
    const SwTask A = (NGRP == 2 && grp) ? tk[NT - 2] : tk[0];
    SwTask B = (NGRP == 2 && grp) ? tk[NT - 1] : tk[1];
    const bool haveA = A.n > 0, haveB = B.n > 0;
    if (SHARED && !haveB) {   // lone task of its query: the second half idles on the same rows
        B = A;
        B.tL = 0;
    }
    int maxN = 0, maxTL = 0;
#pragma unroll
    for (int x = 0; x < NT; x++) {
        maxN = max(maxN, tk[x].n);
        maxTL = max(maxTL, tk[x].tL);
    }
    const int pairTL = max(A.tL, B.tL);
    const int nStrips = MULTI ? (maxN + ROWS - 1) / ROWS : (maxN > 0 ? 1 : 0);
    const int steps = (maxN > 0 && maxTL > 0) ? maxTL + LW - 1 : 0;
    uint4 *bnd = nullptr;
    if (MULTI) bnd = (uint4 *) (boundary + (A.tL >= B.tL ? A.boundOff : B.boundOff));

    const uint32_t goP = (uint32_t) go | ((uint32_t) go << 16), geP = (uint32_t) ge | ((uint32_t) ge << 16);
    unsigned long long keyA = 0, keyB = 0;   // (value, 0xFFFFF - column, 0xFFFFF - row), best over strips
    const uint32_t *profA = &prof[SHARED ? grp : 2 * grp][0][0] + l * WORDS;
    const uint32_t *profB = SHARED ? profA : &prof[2 * grp + 1][0][0] + l * WORDS;

    for (int strip = 0; strip < nStrips; strip++) {
        const int q0 = strip * ROWS + l * RT;
        // ---- query profiles of this strip (SmithWaterman::createQueryProfile, :163-187) and the lazy-F reset rows
        uint32_t mask[RTP];
        if (strip > 0) __syncthreads();   // the previous strip's profile reads are done
#pragma unroll
        for (int x = 0; x < (SHARED ? 1 : 2); x++) {
            const SwTask &T = x ? B : A;
            uint32_t *pw = &prof[SHARED ? grp : 2 * grp + x][0][0] + l * WORDS;
            int seg = T.segLen > 0 ? q0 % T.segLen : 0;
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                int res[4], cb[4];
                int64_t pidx[4];
                bool valid[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int qi = q0 + 4 * w + b;
                    valid[b] = (4 * w + b < RT) && qi < T.n;
                    res[b] = 20;
                    cb[b] = 0;
                    pidx[b] = 0;
                    if (valid[b]) {
                        const int64_t idx = (int64_t) T.qOff + (int64_t) qi * T.qStep;
                        res[b] = qRes[idx];
                        cb[b] = qBias[idx];
                        pidx[b] = idx * 21;
                    }
                    const bool reset = valid[b] && seg == 0;
                    seg = (seg + 1 == T.segLen) ? 0 : seg + 1;
                    const uint32_t m = reset ? 0u : 0xFFFFu;
                    if (SHARED) mask[4 * w + b] = m | (m << 16);
                    else if (x == 0) mask[4 * w + b] = m;
                    else mask[4 * w + b] |= m << 16;
                }
                if (qProf) {   // profile query: the position's own row
                    for (int a = 0; a < 21; a++) {
                        uint32_t word = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int v = valid[b] ? (int) qProf[pidx[b] + a] : -64;
                            word |= (uint32_t) (uint8_t) (int8_t) v << (8 * b);
                        }
                        pw[a * PSTRIDE + w] = word;
                    }
                } else {
                    for (int a = 0; a < 21; a++) {
                        uint32_t word = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int v = valid[b] ? (int) smat[a * 21 + res[b]] + cb[b] : -64;
                            word |= (uint32_t) (uint8_t) (int8_t) v << (8 * b);
                        }
                        pw[a * PSTRIDE + w] = word;
                    }
                }
                pw[PK_NEUTRAL * PSTRIDE + w] = 0xC0C0C0C0u;
            }
        }
        __syncthreads();

        uint32_t H[RTP], E[RTP];
#pragma unroll
        for (int r = 0; r < RTP; r++) {
            H[r] = 0;
            E[r] = 0;
        }
        uint32_t outG = 0, outFf = 0, outFl = 0, prevInG = 0;
        uint32_t curT = PK_NEUTRAL | (PK_NEUTRAL << 8);
        uint32_t bestv = 0, bestcm = 0, bestcol = 0;          // packed tracking (!WIDE)
        uint32_t bestA = 0, bestB = 0, bcolA = 0, bcolB = 0;   // per task (WIDE)
        uint32_t ccol = (uint32_t) (-l) & 0xFFFFu;
        ccol |= ccol << 16;
        const bool lastStrip = strip == nStrips - 1;
        const bool readBound = MULTI && strip > 0;

        auto loadChunk = [&](int c0) -> uint32_t {
            const int col = c0 + l;
            uint32_t a = PK_NEUTRAL, b = PK_NEUTRAL;
            if (col < A.tL) a = tRes[(int64_t) A.tOff + (int64_t) col * A.tStep];
            if (col < B.tL) b = tRes[(int64_t) B.tOff + (int64_t) col * B.tStep];
            return a | (b << 8);
        };
        auto loadBound = [&](int c0) -> uint4 {
            const int col = c0 + l;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (readBound && col < pairTL) {
                const unsigned long long lo = __hip_atomic_load((unsigned long long *) &bnd[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long hi = __hip_atomic_load((unsigned long long *) &bnd[col] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v.x = (uint32_t) lo; v.y = (uint32_t) (lo >> 32); v.z = (uint32_t) hi;
            }
            return v;
        };
        if (readBound) __threadfence();   // boundary values of the previous strip must be visible

        uint32_t chunk = loadChunk(0);
        uint4 bchunk = loadBound(0);
        for (int k0 = 0; k0 < steps; k0 += LW) {
            const uint32_t chunkNext = loadChunk(k0 + LW);
            uint4 bnext = make_uint4(0, 0, 0, 0);
            if (readBound) bnext = loadBound(k0 + LW);
            const int iEnd = min(LW, steps - k0);
#pragma unroll 1
            for (int i = 0; i < iEnd; i++) {
                // ---- hand-off from lane l-1; lane 0 of each half wavefront takes the target residues (and the
                //      previous strip's boundary) instead
                uint32_t inT = dppShr1(curT), inG = dppShr1(outG), inFf = dppShr1(outFf), inFl = dppShr1(outFl);
                inT = writeLane<0>(readLane(chunk, i), inT);
                if (NGRP == 2) inT = writeLane<32>(readLane(chunk, 32 + i), inT);
                if (readBound) {
                    inG = writeLane<0>(readLane(bchunk.x, i), inG);
                    inFf = writeLane<0>(readLane(bchunk.y, i), inFf);
                    inFl = writeLane<0>(readLane(bchunk.z, i), inFl);
                    if (NGRP == 2) {
                        inG = writeLane<32>(readLane(bchunk.x, 32 + i), inG);
                        inFf = writeLane<32>(readLane(bchunk.y, 32 + i), inFf);
                        inFl = writeLane<32>(readLane(bchunk.z, 32 + i), inFl);
                    }
                } else if (NGRP == 2) {
                    inG = writeLane<32>(0, inG);
                    inFf = writeLane<32>(0, inFf);
                    inFl = writeLane<32>(0, inFl);
                }
                curT = inT;
                const uint32_t *pa = profA + (curT & 0xFFu) * PSTRIDE;
                const uint32_t *pb = profB + (curT >> 8) * PSTRIDE;
                uint32_t h[RTP];
#pragma unroll
                for (int w = 0; w < WORDS; w++) {
                    const uint32_t d0 = (w == 0) ? prevInG : H[4 * w - 1];
                    addProfile4(pa[w], pb[w], d0, H[4 * w], H[4 * w + 1], H[4 * w + 2], h[4 * w], h[4 * w + 1], h[4 * w + 2],
                                h[4 * w + 3]);
                }
                prevInG = inG;
                uint32_t Fl = inFl, Ff = inFf, cm = 0, cmA = 0, cmB = 0;
#pragma unroll
                for (int r = 0; r < RT; r++) {
                    Fl &= mask[r];
                    const uint32_t hpre = pkMax(pkMax(h[r], E[r]), Fl);
                    const uint32_t g = pkMax(hpre, Ff);
                    const uint32_t open = pkSubSat(hpre, goP);
                    E[r] = pkMax(pkSubSat(E[r], geP), open);
                    Fl = pkMax(pkSubSat(Fl, geP), open);
                    Ff = pkMax(pkSubSat(Ff, geP), open);
                    H[r] = g;
                    if (WIDE) {
                        cmA = max(cmA, (g << 16) | (uint32_t) (31 - r));
                        cmB = max(cmB, (g & 0xFFFF0000u) | (uint32_t) (31 - r));
                    } else {
                        cm = pkMax(cm, pkRowCode(g, 31 - r));
                    }
                }
                outG = H[RT - 1];
                outFf = Ff;
                outFl = Fl;
                if (MULTI && !lastStrip && l == LW - 1) {
                    const int c = k0 + i - (LW - 1);
                    if (c >= 0 && c < pairTL) {
                        const unsigned long long lo = (unsigned long long) outG | ((unsigned long long) outFf << 32);
                        __hip_atomic_store((unsigned long long *) &bnd[c], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store((unsigned long long *) &bnd[c] + 1, (unsigned long long) outFl, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                // ---- column maximum: a strictly larger value takes over (first column wins, smallest row inside)
                if (WIDE) {
                    const bool upA = cmA > (bestA | 0xFFFFu), upB = cmB > (bestB | 0xFFFFu);
                    bestA = upA ? cmA : bestA;
                    bcolA = upA ? ccol : bcolA;
                    bestB = upB ? cmB : bestB;
                    bcolB = upB ? ccol : bcolB;
                } else {
                    const uint32_t v = pkLshr(cm, 5);
                    const uint32_t nb = pkMax(bestv, v);
                    const uint32_t m = pkAshr(pkSub(bestv, nb), 15);
                    bestcm = bfi(m, cm, bestcm);
                    bestcol = bfi(m, ccol, bestcol);
                    bestv = nb;
                }
                ccol = pkAdd(ccol, 0x00010001u);
            }
            chunk = chunkNext;
            bchunk = bnext;
        }
        // ---- this strip's candidates
        {
            const uint32_t vA = WIDE ? bestA >> 16 : bestv & 0xFFFFu, vB = WIDE ? bestB >> 16 : bestv >> 16;
            const uint32_t rcA = WIDE ? bestA & 31u : bestcm & 31u, rcB = WIDE ? bestB & 31u : (bestcm >> 16) & 31u;
            const uint32_t rowA = (uint32_t) (q0 + 31 - (int) rcA), rowB = (uint32_t) (q0 + 31 - (int) rcB);
            const uint32_t colA = WIDE ? bcolA & 0xFFFFu : bestcol & 0xFFFFu, colB = WIDE ? bcolB >> 16 : bestcol >> 16;
            if (vA > 0) {
                const unsigned long long kk = ((unsigned long long) vA << 40) | ((unsigned long long) (0xFFFFFu - colA) << 20) |
                                              (unsigned long long) (0xFFFFFu - rowA);
                keyA = kk > keyA ? kk : keyA;
            }
            if (vB > 0) {
                const unsigned long long kk = ((unsigned long long) vB << 40) | ((unsigned long long) (0xFFFFFu - colB) << 20) |
                                              (unsigned long long) (0xFFFFFu - rowB);
                keyB = kk > keyB ? kk : keyB;
            }
        }
    }
    // ---- reduce over the LW lanes: max value, then smallest column, then smallest row
#pragma unroll
    for (int off = LW / 2; off >= 1; off >>= 1) {
        const unsigned long long oa = __shfl_xor(keyA, off, LW), ob = __shfl_xor(keyB, off, LW);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
    }
    if (l == 0) {
        if (haveA) {
            const int v = (int) (keyA >> 40);
            out[3 * A.slot + 0] = v;
            out[3 * A.slot + 1] = v == 0 ? -1 : 0xFFFFF - (int) ((keyA >> 20) & 0xFFFFF);
            out[3 * A.slot + 2] = v == 0 ? A.n - 1 : 0xFFFFF - (int) (keyA & 0xFFFFF);
        }
        if (haveB) {
            const int v = (int) (keyB >> 40);
            out[3 * B.slot + 0] = v;
            out[3 * B.slot + 1] = v == 0 ? -1 : 0xFFFFF - (int) ((keyB >> 20) & 0xFFFFF);
            out[3 * B.slot + 2] = v == 0 ? B.n - 1 : 0xFFFFF - (int) (keyB & 0xFFFFF);
        }
    }
}

// ---- aligned form: the lanes ARE the reference's SIMD segments ---------------------------------------------------------
// A byte-structure task (segLen = ceil(n / 32), StripedSmithWaterman.cpp:639-915) with n in (32 (RT - 1), 32 RT] has
// segLen == RT: on a 32-lane group with RT rows per lane the reference's 32 segments are exactly the lanes.  The lazy-F
// chain of the lane structure, Fl, then restarts at row 0 of every lane in every column: no per-row reset masks, no Fl
// hand-off, and the first / last row of a lane drop the Fl instructions they do not need.  The class of a task is its RT
// (5 .. 24: 129 .. 768 rows), so a task pays for at most 31 padding rows and queries beyond 384 rows keep the 32-lane
// ramp.  Also here, against the per-column overhead of the general kernel (sw_score_pk_kernel):
//   * the residue stream carries LDS byte offsets of the profile rows (res * row stride, 16 bits per task), so a lane's
//     profile address is one SDWA add per task;
//   * the profile words of column k + 1 are requested before the cells of column k are computed (the residue a lane sees
//     next is the one its upper neighbour holds now), so LDS latency is off the dependency chain and one wavefront per
//     SIMD suffices where LDS is short;
//   * best-so-far bookkeeping in five instructions on the row-coded column maximum alone (value << 5 | 31 - row: a
//     strictly larger VALUE takes over, i.e. code > best | 31), the column as the wave-uniform step counter.
__device__ __forceinline__ uint32_t pkAshr15(uint32_t a) {
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t bfiAsm(uint32_t mask, uint32_t a, uint32_t b) {   // (a & mask) | (b & ~mask)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t addWord0(uint32_t base, uint32_t packed) {   // base + (packed & 0xFFFF)
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(base), "v"(packed));
    return r;
}
__device__ __forceinline__ uint32_t addWord1(uint32_t base, uint32_t packed) {   // base + (packed >> 16)
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(base), "v"(packed));
    return r;
}

// LW = 32: a lane is one segment (RT = segLen).  LW = 16: four 16-lane groups of a task pair each, a lane holds TWO segments
// (RT = 2 segLen, Fl restarts at rows 0 and RT / 2): half the systolic ramp (15 idle steps per task instead of 31), the
// per-column overhead spread over twice the rows, and the group boundaries are DPP row boundaries -- row_shr:1 hands G / Ff
// down with a zero entering every group's first lane, and the residues enter there from a register that row_ror:1 rotates
// once per column (no v_readlane / v_writelane at all).
template <int CTRL, bool ZERO_FILL>
__device__ __forceinline__ uint32_t dppMove(uint32_t old, uint32_t v) {
    return (uint32_t) __builtin_amdgcn_update_dpp((int) old, (int) v, CTRL, 0xf, 0xf, ZERO_FILL);
}

// SHARE: 0 = every task has its own profile; 1 = the two tasks of a pair have the same query (one profile per group);
// 2 = ALL tasks of the wavefront have the same query (the caller pads a query's run of pairs to whole wavefronts): one profile
// per wavefront -- half (LW = 32) or a quarter (LW = 16) of the LDS, which is what lets the memory-bound prefilter workgroups of
// the other streams live on the same CUs.
// WAVES (SHARE == 2 only): wavefronts per workgroup that share the one profile -- the caller pads a query's run of pairs to whole
// workgroups; the LDS a score wavefront holds is what bounds how many of them (and how many of the other streams' workgroups) a CU
// takes, and half a profile per wavefront lifts that bound to the wave slots
template <int RT, int LW, int SHARE, int WAVES = 1>
__global__ void __launch_bounds__(64 * WAVES)
sw_score_pk_aligned_kernel(const SwTask *__restrict__ tasks, uint32_t nTasks, const uint8_t *__restrict__ qRes,
                           const int8_t *__restrict__ qBias, const uint8_t *__restrict__ tRes, const int8_t *__restrict__ mat,
                           int go, int ge, int32_t *__restrict__ out, const uint32_t *__restrict__ order,
                           const int8_t *__restrict__ qProf) {
    static_assert(LW == 32 || (LW == 16 && RT % 2 == 0), "32 lanes of one segment or 16 lanes of two");
    static_assert(WAVES == 1 || SHARE == 2, "several wavefronts per workgroup share ONE profile");
    constexpr int SEG = RT * LW / 32;      // rows of a reference segment
    constexpr int WORDS = (RT + 3) / 4;
    constexpr int RTP = 4 * WORDS;
    constexpr int PSTRIDE = LW * WORDS;    // dwords per residue row of a profile
    constexpr int NGRP = 64 / LW;
    constexpr int NT = 2 * NGRP;           // tasks per wavefront: a task pair per group of LW lanes
    constexpr uint32_t ROWB = PSTRIDE * 4; // bytes per residue row
    static_assert(22 * ROWB < 65536, "row offsets travel in 16 bits");
    constexpr bool SHARED = SHARE != 0;
    __shared__ uint32_t prof[SHARE == 2 ? 1 : (SHARE == 1 ? NGRP : NT)][22][PSTRIDE];
    __shared__ int8_t smat[441];
    for (int i = threadIdx.x; i < 441; i += 64 * WAVES) smat[i] = mat[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LW, l = lane % LW;
    SwTask tk[NT];
#pragma unroll
    for (int x = 0; x < NT; x++) {
        const uint32_t id = (blockIdx.x * WAVES + (uint32_t) wave) * NT + x;
        const uint32_t tid = id < nTasks ? (order ? order[id] : id) : 0xFFFFFFFFu;
        if (tid != 0xFFFFFFFFu) {
            tk[x] = tasks[tid];
        } else {
            tk[x].n = 0; tk[x].tL = 0; tk[x].qOff = 0; tk[x].tOff = 0; tk[x].qStep = 1; tk[x].tStep = 1; tk[x].segLen = 1;
            tk[x].slot = 0; tk[x].boundOff = 0;
        }
    }
    SwTask A = tk[0], B = tk[1];
#pragma unroll
    for (int g = 1; g < NGRP; g++)
        if (grp == g) {
            A = tk[2 * g];
            B = tk[2 * g + 1];
        }
    const bool haveA = A.n > 0, haveB = B.n > 0;
    if (SHARED && !haveB) {   // lone task of its query: the second half idles on the same rows
        B = A;
        B.tL = 0;
    }
    int maxN = 0, maxTL = 0;
#pragma unroll
    for (int x = 0; x < NT; x++) {
        maxN = max(maxN, tk[x].n);
        maxTL = max(maxTL, tk[x].tL);
    }
    const int steps = (maxN > 0 && maxTL > 0) ? maxTL + LW - 1 : 0;
    const uint32_t goP = (uint32_t) go | ((uint32_t) go << 16), geP = (uint32_t) ge | ((uint32_t) ge << 16);
    const int q0 = l * RT;
    // ---- query profiles (SmithWaterman::createQueryProfile, :163-187)
#pragma unroll
    for (int x = 0; x < (SHARED ? 1 : 2); x++) {
        // (SHARE == 2: every group of every wavefront writes a share of the residue rows of the one profile; the first task of the
        // workgroup is a real one -- a run is padded at its end -- and names the query for all of them)
        SwTask T0 = tk[0];
        if (WAVES > 1) {
            const uint32_t id0 = blockIdx.x * WAVES * NT;
            const uint32_t tid0 = id0 < nTasks ? (order ? order[id0] : id0) : 0xFFFFFFFFu;
            if (tid0 != 0xFFFFFFFFu) T0 = tasks[tid0];
        }
        const SwTask &T = SHARE == 2 ? T0 : (x ? B : A);
        uint32_t *pw = &prof[SHARE == 2 ? 0 : (SHARE == 1 ? grp : 2 * grp + x)][0][0] + l * WORDS;
#pragma unroll
        for (int w = 0; w < WORDS; w++) {
            int res[4], cb[4];
            int64_t pidx[4];
            bool valid[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int qi = q0 + 4 * w + b;
                valid[b] = (4 * w + b < RT) && qi < T.n;
                res[b] = 20;
                cb[b] = 0;
                pidx[b] = 0;
                if (valid[b]) {
                    const int64_t idx = (int64_t) T.qOff + (int64_t) qi * T.qStep;
                    res[b] = qRes[idx];
                    cb[b] = qBias[idx];
                    pidx[b] = idx * 21;
                }
            }
            const int aFirst = SHARE == 2 ? wave * NGRP + grp : 0, aStep = SHARE == 2 ? NGRP * WAVES : 1;
            if (qProf) {   // profile query: the position's own row
                for (int a = aFirst; a < 21; a += aStep) {
                    uint32_t word = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int v = valid[b] ? (int) qProf[pidx[b] + a] : -64;
                        word |= (uint32_t) (uint8_t) (int8_t) v << (8 * b);
                    }
                    pw[a * PSTRIDE + w] = word;
                }
            } else {
                for (int a = aFirst; a < 21; a += aStep) {
                    uint32_t word = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int v = valid[b] ? (int) smat[a * 21 + res[b]] + cb[b] : -64;
                        word |= (uint32_t) (uint8_t) (int8_t) v << (8 * b);
                    }
                    pw[a * PSTRIDE + w] = word;
                }
            }
            pw[PK_NEUTRAL * PSTRIDE + w] = 0xC0C0C0C0u;
        }
    }
    __syncthreads();
    // LDS byte addresses of this lane's profile words (row 0); the residue stream adds the row offset
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t baseA = (uint32_t) (size_t) (lds_u32 *) (&prof[SHARE == 2 ? 0 : (SHARE == 1 ? grp : 2 * grp)][0][0] + l * WORDS);
    const uint32_t baseB = SHARED ? baseA : (uint32_t) (size_t) (lds_u32 *) (&prof[2 * grp + 1][0][0] + l * WORDS);
    auto ldsWord = [](uint32_t addr, int w) -> uint32_t { return *((const lds_u32 *) (size_t) addr + w); };

    uint32_t H[RTP], E[RTP];
#pragma unroll
    for (int r = 0; r < RTP; r++) {
        H[r] = 0;
        E[r] = 0;
    }
    uint32_t outG = 0, outFf = 0, prevInG = 0;
    uint32_t bestcm = 0, bestcol = 0;
    constexpr uint32_t NEUT = (uint32_t) PK_NEUTRAL * ROWB;
    auto loadChunk = [&](int c0) -> uint32_t {   // row offsets (bytes) of the residues of one column per lane, task A | task B << 16
        // LW = 32: lane l holds column c0 + l.  LW = 16: lane 0 holds c0, lane 15 c0 + 1, ... lane 1 c0 + 15 -- the order in which
        // row_ror:1 brings them to lane 0
        const int col = c0 + (LW == 16 ? (16 - l) & 15 : l);
        uint32_t a = PK_NEUTRAL, b = PK_NEUTRAL;
        if (col < A.tL) a = tRes[(int64_t) A.tOff + (int64_t) col * A.tStep];
        if (col < B.tL) b = tRes[(int64_t) B.tOff + (int64_t) col * B.tStep];
        return (a * ROWB) | ((b * ROWB) << 16);
    };
    uint32_t chunk = loadChunk(0);
    uint32_t chunkNext = loadChunk(LW);
    // column 0's residues enter at the first lane of either group; every other lane starts on the neutral row
    uint32_t T = NEUT | (NEUT << 16);
    if (LW == 16) {
        T = l == 0 ? chunk : T;
    } else {
        T = writeLane<0>(readLane(chunk, 0), T);
        T = writeLane<32>(readLane(chunk, 32), T);
    }
    uint32_t pa[WORDS], pb[WORDS];
    {
        const uint32_t aA = addWord0(baseA, T), aB = addWord1(baseB, T);
#pragma unroll
        for (int w = 0; w < WORDS; w++) {
            pa[w] = ldsWord(aA, w);
            pb[w] = ldsWord(aB, w);
        }
    }
    // one column step: consumes (T, pa, pb), produces the next column's (Tn, pan, pbn); two steps per loop iteration with the
    // roles of the two register sets swapped, so nothing is copied (an odd step count is rounded up: a step past the last
    // column runs on the neutral row and cannot raise a maximum)
    auto columnStep = [&](const int k, const uint32_t &Tc, const uint32_t (&pac)[WORDS], const uint32_t (&pbc)[WORDS], uint32_t &Tn,
                          uint32_t (&pan)[WORDS], uint32_t (&pbn)[WORDS]) {
        // ---- the residues and profile words of column k + 1 (needed one step from now)
        const int i1 = (k + 1) & (LW - 1);
        if (LW == 16) {
            if (i1 == 0) {
                chunk = chunkNext;
                chunkNext = loadChunk(k + 1 + LW);
            } else {
                chunk = dppMove<0x121 /* row_ror:1 */, false>(chunk, chunk);   // the next column's residues into lane 0 of every group
            }
            Tn = dppMove<0x111 /* row_shr:1 */, false>(chunk, Tc);   // lane 0 of a group keeps `old` = the fresh residues
        } else {
            if (i1 == 0) {
                chunk = chunkNext;
                chunkNext = loadChunk(k + 1 + LW);
            }
            Tn = dppShr1(Tc);
            Tn = writeLane<0>(readLane(chunk, i1), Tn);
            Tn = writeLane<32>(readLane(chunk, 32 + i1), Tn);
        }
        {
            const uint32_t aA = addWord0(baseA, Tn), aB = addWord1(baseB, Tn);
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                pan[w] = ldsWord(aA, w);
                pbn[w] = ldsWord(aB, w);
            }
        }
        // ---- hand-off from lane l - 1 (nothing enters the first lane of a group)
        uint32_t inG, inFf;
        if (LW == 16) {
            inG = dppMove<0x111, true>(0, outG);
            inFf = dppMove<0x111, true>(0, outFf);
        } else {
            inG = dppShr1(outG);
            inFf = dppShr1(outFf);
            inG = writeLane<32>(0, inG);
            inFf = writeLane<32>(0, inFf);
        }
        uint32_t h[RTP];
#pragma unroll
        for (int w = 0; w < WORDS; w++) {
            const uint32_t d0 = (w == 0) ? prevInG : H[4 * w - 1];
            addProfile4(pac[w], pbc[w], d0, H[4 * w], H[4 * w + 1], H[4 * w + 2], h[4 * w], h[4 * w + 1], h[4 * w + 2], h[4 * w + 3]);
        }
        prevInG = inG;
        uint32_t Fl = 0, Ff = inFf, cm = 0;
#pragma unroll
        for (int r = 0; r < RT; r++) {
            uint32_t hpre = pkMax(h[r], E[r]);
            if (r % SEG != 0) hpre = pkMax(hpre, Fl);   // a segment starts here: no vertical gap of the lane structure enters
            const uint32_t g = pkMax(hpre, Ff);
            const uint32_t open = pkSubSat(hpre, goP);
            E[r] = pkMax(pkSubSat(E[r], geP), open);
            if (r % SEG == 0) Fl = open;
            else if ((r + 1) % SEG != 0) Fl = pkMax(pkSubSat(Fl, geP), open);
            Ff = pkMax(pkSubSat(Ff, geP), open);
            H[r] = g;
            const uint32_t code = pkRowCode(g, 31 - r);
            cm = r == 0 ? code : pkMax(cm, code);
        }
        outG = H[RT - 1];
        outFf = Ff;
        // ---- a strictly larger value takes over (first column wins, smallest row inside)
        const uint32_t t = bestcm | 0x001F001Fu;
        const uint32_t m = pkAshr15(pkSub(t, cm));   // all ones where cm > t
        const uint32_t kk = (uint32_t) k | ((uint32_t) k << 16);
        bestcm = bfiAsm(m, cm, bestcm);
        bestcol = bfiAsm(m, kk, bestcol);
    };
    uint32_t T2, pa2[WORDS], pb2[WORDS];
#pragma unroll 1
    for (int k = 0; k < steps; k += 2) {
        columnStep(k, T, pa, pb, T2, pa2, pb2);
        columnStep(k + 1, T2, pa2, pb2, T, pa, pb);
    }
    // ---- candidates of this lane -> reduce over the 32 lanes: max value, then smallest column, then smallest row
    unsigned long long keyA = 0, keyB = 0;
    {
        const uint32_t vA = (bestcm & 0xFFFFu) >> 5, vB = bestcm >> 21;
        const uint32_t rowA = (uint32_t) (q0 + 31 - (int) (bestcm & 31u)), rowB = (uint32_t) (q0 + 31 - (int) ((bestcm >> 16) & 31u));
        const uint32_t colA = (bestcol & 0xFFFFu) - (uint32_t) l, colB = (bestcol >> 16) - (uint32_t) l;   // step - lane
        if (vA > 0) keyA = ((unsigned long long) vA << 40) | ((unsigned long long) (0xFFFFFu - colA) << 20) | (unsigned long long) (0xFFFFFu - rowA);
        if (vB > 0) keyB = ((unsigned long long) vB << 40) | ((unsigned long long) (0xFFFFFu - colB) << 20) | (unsigned long long) (0xFFFFFu - rowB);
    }
#pragma unroll
    for (int off = LW / 2; off >= 1; off >>= 1) {
        const unsigned long long oa = __shfl_xor(keyA, off, LW), ob = __shfl_xor(keyB, off, LW);
        keyA = oa > keyA ? oa : keyA;
        keyB = ob > keyB ? ob : keyB;
    }
    if (l == 0) {
        if (haveA) {
            const int v = (int) (keyA >> 40);
            out[3 * A.slot + 0] = v;
            out[3 * A.slot + 1] = v == 0 ? -1 : 0xFFFFF - (int) ((keyA >> 20) & 0xFFFFF);
            out[3 * A.slot + 2] = v == 0 ? A.n - 1 : 0xFFFFF - (int) (keyA & 0xFFFFF);
        }
        if (haveB) {
            const int v = (int) (keyB >> 40);
            out[3 * B.slot + 0] = v;
            out[3 * B.slot + 1] = v == 0 ? -1 : 0xFFFFF - (int) ((keyB >> 20) & 0xFFFFF);
            out[3 * B.slot + 2] = v == 0 ? B.n - 1 : 0xFFFFF - (int) (keyB & 0xFFFFF);
        }
    }
}


// ---- chained form of the aligned kernel: the quads of one query flow through the wavefront back to back --------------------
// sw_score_pk_aligned_kernel<RT, 32, 2> runs one quad (four tasks of one query: two 32-lane groups x two packed tasks) per wavefront
// and pays the systolic ramp -- 31 steps in which part of the lanes work on nothing -- for every quad of ~330 columns.  Here a
// wavefront takes up to `chainLen` CONSECUTIVE quads of the pair list and, while the query stays the same, feeds the next quad's
// first column into lane 0 in the step after the previous quad's last one: the boundary between two targets travels down the
// lanes one lane per step, so in the first 32 steps of a segment lane i (and only lane i, at step i) snapshots its running best
// of the quad it has just finished and clears its DP state (H, E, the diagonal hand-off, the best).  After those 32 steps every
// lane has handed in its snapshot and the finished quad is reduced and stored; the profile is built once per chain and the ramp
// is paid once (the 32 draining steps behind the last quad).  A segment is the longest of its four targets, rounded up to an even
// number of steps and to at least 32 (so that at most one boundary is in flight); columns past a target's end run on the neutral
// profile row and cannot raise a maximum, exactly as in the one-quad kernel.  The column of a best is kept as the chain's step
// counter (16 bits per task), so a chain ends early where the next segment would take it past 65 535.
template <int RT>
__global__ void __launch_bounds__(64)
sw_score_pk_chain_kernel(const SwTask *__restrict__ tasks, uint32_t nTasks, const uint8_t *__restrict__ qRes,
                         const int8_t *__restrict__ qBias, const uint8_t *__restrict__ tRes, const int8_t *__restrict__ mat,
                         int go, int ge, int32_t *__restrict__ out, const uint32_t *__restrict__ order,
                         const int8_t *__restrict__ qProf, uint32_t chainLen) {
    constexpr int LW = 32;
    constexpr int WORDS = (RT + 3) / 4;
    constexpr int RTP = 4 * WORDS;
    constexpr int PSTRIDE = LW * WORDS;    // dwords per residue row of the profile
    constexpr uint32_t ROWB = PSTRIDE * 4; // bytes per residue row
    static_assert(22 * ROWB < 65536, "row offsets travel in 16 bits");
    __shared__ uint32_t prof[22][PSTRIDE];
    __shared__ int8_t smat[441];
    for (int i = threadIdx.x; i < 441; i += 64) smat[i] = mat[i];
    __syncthreads();
    const int lane = threadIdx.x;
    const int grp = lane >> 5, l = lane & 31;
    const int q0 = l * RT;
    const uint32_t goP = (uint32_t) go | ((uint32_t) go << 16), geP = (uint32_t) ge | ((uint32_t) ge << 16);
    constexpr uint32_t NEUT = (uint32_t) PK_NEUTRAL * ROWB;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t base = (uint32_t) (size_t) (lds_u32 *) (&prof[0][0] + l * WORDS);
    auto ldsWord = [](uint32_t addr, int w) -> uint32_t { return *((const lds_u32 *) (size_t) addr + w); };

    const uint32_t nQuads = (nTasks + 3) / 4;
    uint32_t qd = blockIdx.x * chainLen;
    const uint32_t qdEnd = min(qd + chainLen, nQuads);
    auto loadQuad = [&](uint32_t quad, SwTask (&tk)[4]) {   // uniform loads; entries beyond the list and PAIR_NONE entries are empty tasks
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const uint32_t id = quad * 4 + x;
            const uint32_t tid = id < nTasks ? (order ? order[id] : id) : 0xFFFFFFFFu;
            if (tid != 0xFFFFFFFFu) {
                tk[x] = tasks[tid];
            } else {
                tk[x].n = 0; tk[x].tL = 0; tk[x].qOff = 0; tk[x].tOff = 0; tk[x].qStep = 1; tk[x].tStep = 1; tk[x].segLen = 1;
                tk[x].slot = 0; tk[x].boundOff = 0;
            }
        }
    };
    bool firstChain = true;
    while (qd < qdEnd) {
        SwTask tk[4];
        loadQuad(qd, tk);
        if (tk[0].n <= 0) {   // (a quad's first task is a real one: runs are padded at their end)
            qd++;
            continue;
        }
        // ---- the query's profile (SmithWaterman::createQueryProfile, :163-187): either group writes half of the residue rows
        if (!firstChain) __syncthreads();
        firstChain = false;
        const SwTask Q = tk[0];
        {
            uint32_t *pw = &prof[0][0] + l * WORDS;
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                int res[4], cb[4];
                int64_t pidx[4];
                bool valid[4];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int qi = q0 + 4 * w + b;
                    valid[b] = (4 * w + b < RT) && qi < Q.n;
                    res[b] = 20;
                    cb[b] = 0;
                    pidx[b] = 0;
                    if (valid[b]) {
                        const int64_t idx = (int64_t) Q.qOff + (int64_t) qi * Q.qStep;
                        res[b] = qRes[idx];
                        cb[b] = qBias[idx];
                        pidx[b] = idx * 21;
                    }
                }
                if (qProf) {   // profile query: the position's own row
                    for (int a = grp; a < 21; a += 2) {
                        uint32_t word = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int v = valid[b] ? (int) qProf[pidx[b] + a] : -64;
                            word |= (uint32_t) (uint8_t) (int8_t) v << (8 * b);
                        }
                        pw[a * PSTRIDE + w] = word;
                    }
                } else {
                    for (int a = grp; a < 21; a += 2) {
                        uint32_t word = 0;
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int v = valid[b] ? (int) smat[a * 21 + res[b]] + cb[b] : -64;
                            word |= (uint32_t) (uint8_t) (int8_t) v << (8 * b);
                        }
                        pw[a * PSTRIDE + w] = word;
                    }
                }
                pw[PK_NEUTRAL * PSTRIDE + w] = 0xC0C0C0C0u;
            }
        }
        __syncthreads();

        uint32_t H[RTP], E[RTP];
#pragma unroll
        for (int r = 0; r < RTP; r++) {
            H[r] = 0;
            E[r] = 0;
        }
        uint32_t outG = 0, outFf = 0, prevInG = 0;
        uint32_t bestcm = 0, bestcol = 0, savedcm = 0, savedcol = 0;
        uint32_t T = NEUT | (NEUT << 16);
        uint32_t pa[WORDS], pb[WORDS], T2, pa2[WORDS], pb2[WORDS];
        // the targets of the segment in flight, as this lane's group sees them
        uint64_t tOffA = 0, tOffB = 0;
        int32_t tLA = 0, tLB = 0, tStepA = 1, tStepB = 1;
        uint32_t chunk = 0, chunkNext = 0;
        auto loadChunk = [&](int c0) -> uint32_t {   // row offsets (bytes) of the residues of column c0 + l, task A | task B << 16
            const int col = c0 + l;
            uint32_t a = PK_NEUTRAL, b = PK_NEUTRAL;
            if (col < tLA) a = tRes[(int64_t) tOffA + (int64_t) col * tStepA];
            if (col < tLB) b = tRes[(int64_t) tOffB + (int64_t) col * tStepB];
            return (a * ROWB) | ((b * ROWB) << 16);
        };
        uint32_t S = 0;   // the chain's step counter at the start of the segment in flight
        // one column step (see sw_score_pk_aligned_kernel): consumes (Tc, pac, pbc), produces the next column's (Tn, pan, pbn)
        auto columnStep = [&](const int i, const uint32_t &Tc, const uint32_t (&pac)[WORDS], const uint32_t (&pbc)[WORDS], uint32_t &Tn,
                              uint32_t (&pan)[WORDS], uint32_t (&pbn)[WORDS]) {
            const int i1 = (i + 1) & (LW - 1);
            if (i1 == 0) {
                chunk = chunkNext;
                chunkNext = loadChunk(i + 1 + LW);
            }
            Tn = dppShr1(Tc);
            Tn = writeLane<0>(readLane(chunk, i1), Tn);
            Tn = writeLane<32>(readLane(chunk, 32 + i1), Tn);
            {
                const uint32_t aA = addWord0(base, Tn), aB = addWord1(base, Tn);
#pragma unroll
                for (int w = 0; w < WORDS; w++) {
                    pan[w] = ldsWord(aA, w);
                    pbn[w] = ldsWord(aB, w);
                }
            }
            uint32_t inG = dppShr1(outG), inFf = dppShr1(outFf);
            inG = writeLane<32>(0, inG);
            inFf = writeLane<32>(0, inFf);
            uint32_t h[RTP];
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                const uint32_t d0 = (w == 0) ? prevInG : H[4 * w - 1];
                addProfile4(pac[w], pbc[w], d0, H[4 * w], H[4 * w + 1], H[4 * w + 2], h[4 * w], h[4 * w + 1], h[4 * w + 2], h[4 * w + 3]);
            }
            prevInG = inG;
            uint32_t Fl = 0, Ff = inFf, cm = 0;
#pragma unroll
            for (int r = 0; r < RT; r++) {
                uint32_t hpre = pkMax(h[r], E[r]);
                if (r != 0) hpre = pkMax(hpre, Fl);   // a lane is a segment of the reference: no vertical gap of the lane structure enters row 0
                const uint32_t g = pkMax(hpre, Ff);
                const uint32_t open = pkSubSat(hpre, goP);
                E[r] = pkMax(pkSubSat(E[r], geP), open);
                if (r == 0) Fl = open;
                else if (r + 1 != RT) Fl = pkMax(pkSubSat(Fl, geP), open);
                Ff = pkMax(pkSubSat(Ff, geP), open);
                H[r] = g;
                const uint32_t code = pkRowCode(g, 31 - r);
                cm = r == 0 ? code : pkMax(cm, code);
            }
            outG = H[RT - 1];
            outFf = Ff;
            const uint32_t t = bestcm | 0x001F001Fu;
            const uint32_t m = pkAshr15(pkSub(t, cm));   // all ones where cm > t
            const uint32_t k = S + (uint32_t) i;
            const uint32_t kk = k | (k << 16);
            bestcm = bfiAsm(m, cm, bestcm);
            bestcol = bfiAsm(m, kk, bestcol);
        };
        // the boundary between two quads reaches lane i of either group at step i of the new segment
        auto boundary = [&](const int i) {
            if (l == i) {
                savedcm = bestcm;
                savedcol = bestcol;
                bestcm = 0;
                bestcol = 0;
                prevInG = 0;
#pragma unroll
                for (int r = 0; r < RTP; r++) {
                    H[r] = 0;
                    E[r] = 0;
                }
            }
        };
        // first residues of a segment into the first lane of either group, and that lane's profile words again
        auto enterSegment = [&]() {
            chunk = loadChunk(0);
            chunkNext = loadChunk(LW);
            T = writeLane<0>(readLane(chunk, 0), T);
            T = writeLane<32>(readLane(chunk, 32), T);
            const uint32_t aA = addWord0(base, T), aB = addWord1(base, T);
#pragma unroll
            for (int w = 0; w < WORDS; w++) {
                pa[w] = ldsWord(aA, w);
                pb[w] = ldsWord(aB, w);
            }
        };
        // the finished quad: its snapshot -> max value, then smallest column, then smallest row over the 32 lanes of a group
        uint32_t pSlotA = 0, pSlotB = 0, pN = 0, pS = 0;
        bool pHaveA = false, pHaveB = false, havePrev = false;
        auto storePrev = [&]() {
            unsigned long long keyA = 0, keyB = 0;
            const uint32_t vA = (savedcm & 0xFFFFu) >> 5, vB = savedcm >> 21;
            const uint32_t rowA = (uint32_t) (q0 + 31 - (int) (savedcm & 31u)), rowB = (uint32_t) (q0 + 31 - (int) ((savedcm >> 16) & 31u));
            const uint32_t colA = (savedcol & 0xFFFFu) - (uint32_t) l - pS, colB = (savedcol >> 16) - (uint32_t) l - pS;   // step - lane - segment start
            if (vA > 0) keyA = ((unsigned long long) vA << 40) | ((unsigned long long) (0xFFFFFu - colA) << 20) | (unsigned long long) (0xFFFFFu - rowA);
            if (vB > 0) keyB = ((unsigned long long) vB << 40) | ((unsigned long long) (0xFFFFFu - colB) << 20) | (unsigned long long) (0xFFFFFu - rowB);
#pragma unroll
            for (int off = LW / 2; off >= 1; off >>= 1) {
                const unsigned long long oa = __shfl_xor(keyA, off, LW), ob = __shfl_xor(keyB, off, LW);
                keyA = oa > keyA ? oa : keyA;
                keyB = ob > keyB ? ob : keyB;
            }
            if (l == 0) {
                if (pHaveA) {
                    const int v = (int) (keyA >> 40);
                    out[3 * pSlotA + 0] = v;
                    out[3 * pSlotA + 1] = v == 0 ? -1 : 0xFFFFF - (int) ((keyA >> 20) & 0xFFFFF);
                    out[3 * pSlotA + 2] = v == 0 ? (int) pN - 1 : 0xFFFFF - (int) (keyA & 0xFFFFF);
                }
                if (pHaveB) {
                    const int v = (int) (keyB >> 40);
                    out[3 * pSlotB + 0] = v;
                    out[3 * pSlotB + 1] = v == 0 ? -1 : 0xFFFFF - (int) ((keyB >> 20) & 0xFFFFF);
                    out[3 * pSlotB + 2] = v == 0 ? (int) pN - 1 : 0xFFFFF - (int) (keyB & 0xFFFFF);
                }
            }
        };

        bool more = true;
        while (more) {
            const SwTask &A = grp ? tk[2] : tk[0];
            const SwTask &B = grp ? tk[3] : tk[1];
            tOffA = A.tOff; tLA = A.n > 0 ? A.tL : 0; tStepA = A.tStep;
            tOffB = B.tOff; tLB = B.n > 0 ? B.tL : 0; tStepB = B.tStep;
            int maxTL = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) maxTL = max(maxTL, tk[x].n > 0 ? tk[x].tL : 0);
            const int L = max((maxTL + 1) & ~1, LW);
            enterSegment();
#pragma unroll 1
            for (int i = 0; i < LW; i += 2) {
                boundary(i);
                columnStep(i, T, pa, pb, T2, pa2, pb2);
                boundary(i + 1);
                columnStep(i + 1, T2, pa2, pb2, T, pa, pb);
            }
            if (havePrev) storePrev();
#pragma unroll 1
            for (int i = LW; i < L; i += 2) {
                columnStep(i, T, pa, pb, T2, pa2, pb2);
                columnStep(i + 1, T2, pa2, pb2, T, pa, pb);
            }
            pSlotA = A.slot; pSlotB = B.slot; pN = (uint32_t) Q.n; pS = S;
            pHaveA = A.n > 0; pHaveB = B.n > 0; havePrev = true;
            S += (uint32_t) L;
            // the next quad of the list continues the chain while it belongs to the same query
            qd++;
            more = false;
            if (qd < qdEnd) {
                SwTask nk[4];
                loadQuad(qd, nk);
                int nTL = 0;
#pragma unroll
                for (int x = 0; x < 4; x++) nTL = max(nTL, nk[x].n > 0 ? nk[x].tL : 0);
                if (nk[0].n == Q.n && nk[0].qOff == Q.qOff && nk[0].qStep == Q.qStep && S + (uint32_t) max(nTL + 1, LW) + LW <= 65535u) {
#pragma unroll
                    for (int x = 0; x < 4; x++) tk[x] = nk[x];
                    more = true;
                }
            }
        }
        // ---- drain: 32 steps on the neutral row carry the boundary behind the last quad down the lanes
        tLA = 0;
        tLB = 0;
        enterSegment();
#pragma unroll 1
        for (int i = 0; i < LW; i += 2) {
            boundary(i);
            columnStep(i, T, pa, pb, T2, pa2, pb2);
            boundary(i + 1);
            columnStep(i + 1, T2, pa2, pb2, T, pa, pb);
        }
        storePrev();
    }
}

}  // namespace sdpk
#endif
