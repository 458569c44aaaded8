// clusterhits on gfx950: batched agglomerative clustering of the best hits of one (query set, target set)
// pair per workgroup (R/src/util/ClusterHits.cpp:295-492).
//
// The reference keeps a dense K x K matrix D of merge scores plus a (deliberately stale) arg-max vector
// dmin.  D[a][b] is only ever rewritten when node a or node b changes, and then from their current
// contents, so D is a pure function of the current node states; only dmin (and D[i][dmin[i]], cached here)
// carries history.  The kernel therefore never materialises D (8*K^2 bytes, 72 MB at K = 3000): every
// entry is recomputed on demand from integer geometry (bounding boxes, sizes, rank-sorted member lists)
// and a logGamma table, and the reference's update rules for dmin are replayed literally:
//   init   dmin[i]  = first arg-max_j D[i][j]                                   (:377-388)
//   loop   i1 = first arg-max_i D[i][dmin[i]], i2 = dmin[i1], stop if the score is 0   (:396-410)
//          merge i2 into i1 (always, even below sMin), then test score >= sMin          (:395,412-453)
//          dmin[i1] = first arg-max_j of the new row;  for j != i1,i2:
//          dmin[j]  = D[j][i1] > D[j][dmin[j]] ? i1 : dmin[j]   (stale entries persist)   (:439-449)
// Per merge this is O(K) box tests spread over the workgroup and a few O(cluster) list merges; HBM traffic
// is the K*(4+4+1) input bytes plus K*4 output bytes per pair (algorithmic), the scratch stays in L2.
// The double-precision P-values of the emitted clusters (exp/pow/log) are evaluated on the host with the
// reference's own expressions (sd_clusterhits_batch below) so their text form matches digit for digit.
#include <memory>
#include "sd_common.h"
#include "sd_host.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

namespace {

constexpr uint32_t NIL = 0xFFFFFFFFu;
constexpr uint64_t CH_SCRATCH_WORDS = 11;   // 32-bit scratch words per hit (ChView's arrays)

struct ChPair {
    uint64_t off;   // first hit
    uint32_t K;
    uint32_t pad;
};

struct ChView {
    const uint32_t *qPos, *tPos;
    const uint8_t *strand;
    uint32_t *rank, *next, *head, *size, *iMin, *iMax, *jMin, *jMax, *dmin;
    uint32_t *tail, *mcnt;   // last member of a node's list; conserved neighbour pairs inside the node
    double *cached;
    const double *lg;
    double logq0, ln2;
    uint32_t d;
};

__device__ __forceinline__ double chScoreKSM(const ChView &v, int k, int span, int m) {
    // clusterMatchScore (:120-134) = -0.5*logClusterPval(k, span) - 0.5*logOrderingPval(k, m)
    const double logpClu = 2 * v.lg[span + 1] - 2 * v.lg[span - k + 1] - v.lg[k + 1] + k * v.logq0;
    const double logpOrd = log(1 - 1.0 * m / k) - m * v.ln2 - v.lg[m + 1];
    return -0.5 * logpClu - 0.5 * logpOrd;
}

// one step of findConservedPairs (:104-117): does `cur`, directly behind `prev` in query order, continue its direction?
__device__ __forceinline__ uint32_t chConserved(const ChView &v, uint32_t prev, uint32_t cur) {
    const uint8_t sp = v.strand[prev], sc = v.strand[cur];
    const bool prevS = ((sp & 1) != 0) == ((sp & 2) != 0), sEq = ((sc & 1) != 0) == ((sc & 2) != 0);
    const bool sameOrder = v.tPos[cur] > v.tPos[prev];
    return ((prevS == sameOrder) && (sEq == sameOrder)) ? 1u : 0u;
}

// D[a][b]: 0 if either node is empty or the boxes are incompatible (unsigned gap arithmetic of :158)
__device__ double chPairScore(const ChView &v, uint32_t a, uint32_t b) {
    const uint32_t sa = v.size[a], sb = v.size[b];
    if (a == b || sa == 0 || sb == 0) return 0.0;
    const uint32_t iMin1 = v.iMin[a], iMax1 = v.iMax[a], jMin1 = v.jMin[a], jMax1 = v.jMax[a];
    const uint32_t iMin2 = v.iMin[b], iMax2 = v.iMax[b], jMin2 = v.jMin[b], jMax2 = v.jMax[b];
    const uint32_t gj = min(jMin1 - jMax2, jMin2 - jMax1);
    const uint32_t gi = min(iMin1 - iMax2, iMin2 - iMax1);
    if (!(gj <= v.d && gi <= v.d)) return 0.0;
    const int k = (int) (sa + sb);
    const uint32_t iMax = max(iMax1, iMax2), iMin = min(iMin1, iMin2), jMax = max(jMax1, jMax2), jMin = min(jMin1, jMin2);
    const int spanI = (int) (iMax - iMin + 1), spanJ = (int) (jMax - jMin + 1);
    const int span = spanI > spanJ ? spanI : spanJ;
    // conserved neighbour pairs of the union in ascending query position (findConservedPairs, :104-117).  Whether two
    // members count depends on that pair alone (chConserved), so when one node's members all come before the other's -- the
    // usual case: a collinear cluster growing at its ends -- the union's count is the two nodes' own counts plus the one
    // pair at the seam, and the lists need not be walked (a self pair of K = 3 000 collinear hits was 2.4 s of list walks)
    {
        const uint32_t ta = v.tail[a], hb = v.head[b], tb = v.tail[b], ha = v.head[a];
        if (v.rank[ta] < v.rank[hb]) return chScoreKSM(v, k, span, (int) (v.mcnt[a] + v.mcnt[b] + chConserved(v, ta, hb)));
        if (v.rank[tb] < v.rank[ha]) return chScoreKSM(v, k, span, (int) (v.mcnt[a] + v.mcnt[b] + chConserved(v, tb, ha)));
    }
    uint32_t pa = v.head[a], pb = v.head[b];
    int m = 0;
    bool first = true, prevS = false;
    uint32_t prevT = 0;
    while (pa != NIL || pb != NIL) {
        uint32_t pick;
        if (pb == NIL || (pa != NIL && v.rank[pa] < v.rank[pb])) { pick = pa; pa = v.next[pa]; }
        else { pick = pb; pb = v.next[pb]; }
        const uint8_t st = v.strand[pick];
        const bool sEq = ((st & 1) != 0) == ((st & 2) != 0);
        const uint32_t tp = v.tPos[pick];
        if (!first) {
            const bool sameOrder = tp > prevT;
            if ((prevS == sameOrder) && (sEq == sameOrder)) m++;
        }
        prevT = tp; prevS = sEq; first = false;
    }
    return chScoreKSM(v, k, span, m);
}

// block-wide arg-max: larger value wins, ties -> smaller index.  256 threads.
__device__ void blockArgMax(double &val, uint32_t &idx, double *sVal, uint32_t *sIdx) {
    const int t = threadIdx.x;
    sVal[t] = val;
    sIdx[t] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
            const double ov = sVal[t + s];
            const uint32_t oi = sIdx[t + s];
            if (ov > sVal[t] || (ov == sVal[t] && oi < sIdx[t])) { sVal[t] = ov; sIdx[t] = oi; }
        }
        __syncthreads();
    }
    val = sVal[0];
    idx = sIdx[0];
    __syncthreads();
}

__global__ void __launch_bounds__(256)
clusterhits_kernel(const ChPair *__restrict__ pairs, uint32_t nPairs, const uint32_t *__restrict__ qPosAll,
                   const uint32_t *__restrict__ tPosAll, const uint8_t *__restrict__ strandAll,
                   const double *__restrict__ lg, double logq0, double ln2, uint32_t d, uint32_t *__restrict__ scratchU,
                   double *__restrict__ scratchD, uint32_t *__restrict__ nodeOf, uint32_t *__restrict__ mergesOut) {
    __shared__ double sVal[256];
    __shared__ uint32_t sIdx[256];
    __shared__ uint32_t sI1, sI2;
    __shared__ double sMax;
    const uint32_t p = blockIdx.x;
    if (p >= nPairs) return;
    const uint64_t off = pairs[p].off;
    const uint32_t K = pairs[p].K;
    ChView v;
    v.qPos = qPosAll + off; v.tPos = tPosAll + off; v.strand = strandAll + off;
    uint32_t *su = scratchU + off * CH_SCRATCH_WORDS;
    v.rank = su; v.next = su + K; v.head = su + 2 * (uint64_t) K; v.size = su + 3 * (uint64_t) K;
    v.iMin = su + 4 * (uint64_t) K; v.iMax = su + 5 * (uint64_t) K; v.jMin = su + 6 * (uint64_t) K; v.jMax = su + 7 * (uint64_t) K;
    v.dmin = su + 8 * (uint64_t) K;
    v.tail = su + 9 * (uint64_t) K; v.mcnt = su + 10 * (uint64_t) K;
    v.cached = scratchD + off;
    v.lg = lg; v.logq0 = logq0; v.ln2 = ln2; v.d = d;
    uint32_t *out = nodeOf + off;
    const int t = threadIdx.x;
    if (K <= 1) {   // K == 1 entries are skipped by the reference (:359-361)
        if (t == 0 && K == 1) out[0] = 0;
        if (t == 0) mergesOut[p] = 0;
        return;
    }
    // rank by (qPos, index); singleton nodes
    for (uint32_t h = t; h < K; h += 256) {
        const uint32_t q = v.qPos[h];
        uint32_t r = 0;
        for (uint32_t x = 0; x < K; x++) {
            const uint32_t qx = v.qPos[x];
            r += (qx < q) || (qx == q && x < h);
        }
        v.rank[h] = r;
        v.next[h] = NIL;
        v.head[h] = h;
        v.tail[h] = h;
        v.mcnt[h] = 0;
        v.size[h] = 1;
        v.iMin[h] = q; v.iMax[h] = q;
        v.jMin[h] = v.tPos[h]; v.jMax[h] = v.tPos[h];
    }
    __threadfence_block();
    __syncthreads();
    // init: dmin[i] = first arg-max_j D[i][j]
    for (uint32_t i = t; i < K; i += 256) {
        // nodes are singletons here, so the box test of chPairScore reduces to |dq| <= d && |dt| <= d
        const uint32_t qi = v.qPos[i], ti = v.tPos[i];
        double best = 0.0;   // D[i][0] is the start value (dmin starts at 0, :369)
        uint32_t bj = 0;
        for (uint32_t j = 0; j < K; j++) {
            const uint32_t qj = v.qPos[j], tj = v.tPos[j];
            const uint32_t gi = min(qi - qj, qj - qi), gj = min(ti - tj, tj - ti);
            double s = 0.0;
            if (j != i && gi <= d && gj <= d) s = chPairScore(v, i, j);
            if (j == 0) { best = s; bj = 0; }
            else if (s > best) { best = s; bj = j; }
        }
        v.dmin[i] = bj;
        v.cached[i] = best;
    }
    __threadfence_block();
    __syncthreads();
    const double sMin = -0.5 * (2 * lg[d + 2] - 2 * lg[d + 1 - 2 + 1] - lg[3] + 2 * logq0) - 0.5 * (log(1 - 1.0 * 1 / 2) - 1 * ln2 - lg[2]);
    uint32_t merges = 0;
    for (;;) {
        // i1 = first arg-max_i D[i][dmin[i]]
        double bv = -DBL_MAX;
        uint32_t bi = NIL;
        for (uint32_t i = t; i < K; i += 256) {
            const double c = v.cached[i];
            if (c > bv || (c == bv && i < bi)) { bv = c; bi = i; }
        }
        blockArgMax(bv, bi, sVal, sIdx);
        if (t == 0) {
            sI1 = bi;
            sI2 = v.dmin[bi];
            sMax = bv;
        }
        __syncthreads();
        const uint32_t i1 = sI1, i2 = sI2;
        const double maxScore = sMax;
        if (maxScore == 0.0) break;
        // merge i2 into i1: rank-sorted list merge, box union
        if (t == 0) {
            const uint32_t h1 = v.head[i1], t1 = v.tail[i1], h2 = v.head[i2], t2 = v.tail[i2];
            if (v.rank[t1] < v.rank[h2]) {          // i2's members all behind i1's: append
                v.next[t1] = h2;
                v.tail[i1] = t2;
                v.mcnt[i1] += v.mcnt[i2] + chConserved(v, t1, h2);
            } else if (v.rank[t2] < v.rank[h1]) {   // all in front: prepend
                v.next[t2] = h1;
                v.head[i1] = h2;
                v.mcnt[i1] += v.mcnt[i2] + chConserved(v, t2, h1);
            } else {                                // interleaved: rank-sorted merge, the count taken along the way
                uint32_t pa = h1, pb = h2, hd = NIL, tl = NIL, m = 0;
                while (pa != NIL || pb != NIL) {
                    uint32_t pick;
                    if (pb == NIL || (pa != NIL && v.rank[pa] < v.rank[pb])) { pick = pa; pa = v.next[pa]; }
                    else { pick = pb; pb = v.next[pb]; }
                    if (hd == NIL) hd = pick;
                    else {
                        v.next[tl] = pick;
                        m += chConserved(v, tl, pick);
                    }
                    tl = pick;
                }
                v.next[tl] = NIL;
                v.head[i1] = hd;
                v.tail[i1] = tl;
                v.mcnt[i1] = m;
            }
            v.head[i2] = NIL;
            v.size[i1] += v.size[i2];
            v.size[i2] = 0;
            v.iMin[i1] = min(v.iMin[i1], v.iMin[i2]); v.iMax[i1] = max(v.iMax[i1], v.iMax[i2]);
            v.jMin[i1] = min(v.jMin[i1], v.jMin[i2]); v.jMax[i1] = max(v.jMax[i1], v.jMax[i2]);
        }
        merges++;
        __threadfence_block();
        __syncthreads();
        // new row/column i1, stale-dmin update of every other row
        double rv = -DBL_MAX;
        uint32_t rj = NIL;
        for (uint32_t j = t; j < K; j += 256) {
            double s = 0.0;
            if (j != i1 && j != i2) {
                s = chPairScore(v, j, i1);
                const uint32_t dj = v.dmin[j];
                const double cur = (dj == i2) ? 0.0 : (dj == i1 ? s : v.cached[j]);
                if (s > cur) { v.dmin[j] = i1; v.cached[j] = s; }
                else v.cached[j] = cur;
            } else if (j == i2) {
                v.cached[j] = 0.0;
            }
            if (s > rv || (s == rv && j < rj)) { rv = s; rj = j; }
        }
        blockArgMax(rv, rj, sVal, sIdx);
        if (t == 0) {
            v.dmin[i1] = rj;
            v.cached[i1] = rv;
        }
        __threadfence_block();
        __syncthreads();
        if (!(maxScore >= sMin)) break;
    }
    // node of every hit
    for (uint32_t n = t; n < K; n += 256) {
        uint32_t h = v.head[n];
        if (v.size[n] == 0) continue;
        while (h != NIL) {
            out[h] = n;
            h = v.next[h];
        }
    }
    if (t == 0) mergesOut[p] = merges;
}

// ---- host side: the P-values of the final nodes are evaluated in csrc/host/sd_chpval.cpp, which g++ compiles with the
// reference's floating-point flags (-mfma, GCC's -ffp-contract=fast): `-0.5 * a - 0.5 * b` and `... + k * log(q0)` contract
// into different FMAs under clang, and the reference prints these doubles with four significant digits
using sd::ClusterHit;
}  // namespace

extern "C" int sd_clusterhits_batch(sd_ctx *ctx, const sd_ch_params *par, uint32_t nPairs, const uint64_t *hitOff,
                                    const uint32_t *qPos, const uint32_t *tPos, const uint8_t *strands, const double *pval,
                                    const uint32_t *Nq, const double *lGamma, uint32_t lGammaLen, uint32_t *clusterOfHit,
                                    uint32_t *rankInCluster, uint32_t *nClusters, double *pCO, double *pMH,
                                    uint32_t *clusterSizeOut) {
    if (!ctx || !par || !hitOff || !qPos || !tPos || !strands || !pval || !Nq || !lGamma) return SD_EINVAL;
    (void) hipSetDevice(ctx->device);
    const uint64_t total = hitOff[nPairs];
    if (total == 0 || nPairs == 0) return SD_OK;
    std::unique_ptr<HostScope> hs(new HostScope(ctx, "ch.setup"));
    std::vector<ChPair> hp(nPairs);
    uint32_t maxK = 0;
    for (uint32_t p = 0; p < nPairs; p++) {
        hp[p].off = hitOff[p];
        hp[p].K = (uint32_t) (hitOff[p + 1] - hitOff[p]);
        hp[p].pad = 0;
        maxK = std::max(maxK, hp[p].K);
    }
    // every table index used on the device is < span + 2 <= max position + 3
    uint32_t maxPos = 0;
    for (uint64_t x = 0; x < total; x++) maxPos = std::max(maxPos, std::max(qPos[x], tPos[x]));
    if ((uint64_t) maxPos + 3 > lGammaLen || (uint64_t) maxK + 2 > lGammaLen || par->maxGeneGap + 3 > lGammaLen)
        return sdFail(ctx, SD_EINVAL, "logGamma table too short: %u entries, need %llu", lGammaLen, (unsigned long long) std::max<uint64_t>(maxPos + 3, maxK + 2));
    hs.reset(new HostScope(ctx, "ch.device"));
    struct { ChPair *p; } dPairs;
    struct { uint32_t *p; } dQ, dT, dScratchU, dNode, dMerges;
    struct { uint8_t *p; } dS;
    struct { double *p; } dLg, dScratchD;
    SD_HIP(ctx, wsGet(ctx, "ch.pairs", nPairs, &dPairs.p));
    SD_HIP(ctx, wsGet(ctx, "ch.q", total, &dQ.p));
    SD_HIP(ctx, wsGet(ctx, "ch.t", total, &dT.p));
    SD_HIP(ctx, wsGet(ctx, "ch.s", total, &dS.p));
    SD_HIP(ctx, wsGet(ctx, "ch.lg", lGammaLen, &dLg.p));
    SD_HIP(ctx, wsGet(ctx, "ch.scratchU", total * CH_SCRATCH_WORDS, &dScratchU.p));
    SD_HIP(ctx, wsGet(ctx, "ch.scratchD", total, &dScratchD.p));
    SD_HIP(ctx, wsGet(ctx, "ch.node", total, &dNode.p));
    SD_HIP(ctx, wsGet(ctx, "ch.merges", nPairs, &dMerges.p));
    hs.reset(new HostScope(ctx, "ch.h2d"));
    {
        // one pinned staging block for all inputs (pageable sources make every copy a slow synchronous path)
        const size_t bPairs = nPairs * sizeof(ChPair), bPos = total * sizeof(uint32_t), bLg = lGammaLen * sizeof(double);
        auto up = [](size_t v) { return (v + 255) & ~(size_t) 255; };
        const size_t oQ = up(bPairs), oT = oQ + up(bPos), oS = oT + up(bPos), oL = oS + up(total), all = oL + up(bLg);
        uint8_t *stage = nullptr;
        SD_HIP(ctx, pinGet(ctx, "ch.stage", all, &stage));
        memcpy(stage, hp.data(), bPairs);
        memcpy(stage + oQ, qPos, bPos);
        memcpy(stage + oT, tPos, bPos);
        memcpy(stage + oS, strands, total);
        memcpy(stage + oL, lGamma, bLg);
        SD_HIP(ctx, hipMemcpyAsync(dPairs.p, stage, bPairs, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(ctx, hipMemcpyAsync(dQ.p, stage + oQ, bPos, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(ctx, hipMemcpyAsync(dT.p, stage + oT, bPos, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(ctx, hipMemcpyAsync(dS.p, stage + oS, total, hipMemcpyHostToDevice, ctx->stream));
        SD_HIP(ctx, hipMemcpyAsync(dLg.p, stage + oL, bLg, hipMemcpyHostToDevice, ctx->stream));
    }
    SD_HIP(ctx, hipMemsetAsync(dNode.p, 0xFF, total * sizeof(uint32_t), ctx->stream));
    hs.reset(new HostScope(ctx, "ch.kernel"));
    {
        const bool dbg = getenv("SD_DEBUG_TIMING") != nullptr;
        auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double a0 = now();
        if (dbg) (void) sdStreamSync(ctx);
        const double a1 = now();
        {
            ProfScope ps(ctx, "clusterhits");
            hipLaunchKernelGGL(clusterhits_kernel, dim3(nPairs), dim3(256), 0, ctx->stream, dPairs.p, nPairs, dQ.p, dT.p, dS.p, dLg.p,
                               log(0.001), log(2.0), par->maxGeneGap, dScratchU.p, dScratchD.p, dNode.p, dMerges.p);
            if (dbg) fprintf(stderr, "[clusterhits] sync-before %.1f ms, launch call %.1f ms\n", a1 - a0, now() - a1);
        }
        const double a2 = now();
        SD_HIP(ctx, hipGetLastError());
        SD_HIP(ctx, sdStreamSync(ctx));
        if (dbg) fprintf(stderr, "[clusterhits] profscope %.1f ms, sync-after %.1f ms\n", a2 - a1, now() - a2);
    }
    hs.reset(new HostScope(ctx, "ch.d2h"));
    uint32_t *node = nullptr;
    SD_HIP(ctx, pinGet(ctx, "ch.hnode", total, &node));
    SD_HIP(ctx, hipMemcpyAsync(node, dNode.p, total * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SD_HIP(ctx, sdStreamSync(ctx));
    hs.reset(new HostScope(ctx, "ch.finalise"));
    // ---- emission (:456-485): nodes in index order, size >= cls, pCO / pMH thresholds
#pragma omp parallel
    {
        std::vector<uint32_t> start, items;
        std::vector<ClusterHit> cluster;
#pragma omp for schedule(dynamic, 1)
        for (uint32_t p = 0; p < nPairs; p++) {
            const uint64_t off = hitOff[p];
            const uint32_t K = hp[p].K;
            for (uint32_t h = 0; h < K; h++) {
                clusterOfHit[off + h] = UINT32_MAX;
                rankInCluster[off + h] = 0;
            }
            nClusters[p] = 0;
            if (K <= 1) continue;
            // members of every surviving node, hits in index order (counting sort by node)
            start.assign((size_t) K + 1, 0);
            items.resize(K);
            for (uint32_t h = 0; h < K; h++) {
                const uint32_t n = node[off + h];
                if (n < K) start[n + 1]++;
            }
            for (uint32_t n = 0; n < K; n++) start[n + 1] += start[n];
            {
                std::vector<uint32_t> &fill = items;   // filled through a moving cursor kept in `cursor`
                static thread_local std::vector<uint32_t> cursor;
                cursor.assign(start.begin(), start.end() - 1);
                for (uint32_t h = 0; h < K; h++) {
                    const uint32_t n = node[off + h];
                    if (n < K) fill[cursor[n]++] = h;
                }
            }
            uint32_t nClu = 0;
            for (uint32_t n = 0; n < K; n++) {
                const uint32_t sz = start[n + 1] - start[n];
                if (sz < par->clusterSize || sz == 0) continue;
                cluster.clear();
                for (uint32_t x = start[n]; x < start[n + 1]; x++) {
                    const uint32_t h = items[x];
                    ClusterHit hh;
                    hh.pval = pval[off + h]; hh.qPos = qPos[off + h]; hh.tPos = tPos[off + h];
                    hh.qS = strands[off + h] & 1; hh.tS = (strands[off + h] >> 1) & 1; hh.idx = h;
                    cluster.push_back(hh);
                }
                const double co = sd::chClusterPval(lGamma, cluster);   // sorts `cluster` by qPos, as the reference does
                const double mh = sd::chMultihitPval(lGamma, cluster, (int) Nq[p], par->alpha);
                if (co <= par->pCluThr && mh <= par->pMHThr) {
                    pCO[off + nClu] = co;
                    pMH[off + nClu] = mh;
                    clusterSizeOut[off + nClu] = (uint32_t) cluster.size();
                    for (size_t r = 0; r < cluster.size(); r++) {
                        clusterOfHit[off + cluster[r].idx] = nClu;
                        rankInCluster[off + cluster[r].idx] = (uint32_t) r;
                    }
                    nClu++;
                }
            }
            nClusters[p] = nClu;
        }
    }
    return SD_OK;
}
