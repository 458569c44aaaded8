// P-values of an emitted cluster of hits, double precision (the reference evaluates them in clusterhits and prints them
// with %.3E: R/src/util/ClusterHits.cpp:80-134,184-213,462-464).
//
// Two null models, both over lnFact[n] = ln Gamma(n), i.e. lnFact[n + 1] = ln n!:
//   * clustering x ordering: k hits that span s consecutive genes (the larger of the query-side and the target-side extent)
//     and of whose k - 1 neighbour pairs (in query order) m keep order and strand,
//         ln P_clu = 2 ln s! - 2 ln (s - k)! - ln k! + k ln q0          (q0 = 0.001)
//         ln P_ord = ln(1 - m / k) - m ln 2 - ln m!
//         P = exp((ln P_clu + ln P_ord) / 2);
//   * multi-hit: with the per-hit threshold theta = alpha / (Nq + 1), the k' hits below it and their summed log excess
//     r = sum(ln theta - ln p), P = exp(-r) * sum_{i < k' - 1} r^i / i!.
// The printed digits depend on the last bit, so the products that the reference's AVX2 build contracts into fused
// multiply-adds are written as std::fma here (k * ln q0 onto the placement term; m * ln 2 off the tail term) -- the result
// no longer depends on the compiler's contraction mode -- and r^i / i! keeps the library calls pow and exp.  Pinned bit for
// bit against the reference's compiled functions (oracle/_ref/libsdref_ch.so): tests/test_oracle_clusterhits_ref.py (host,
// sd_host_cluster_pvalues) and tests/test_gpu_clusterhits.py (through the device path).
#include "sd_host.h"

#include <algorithm>
#include <cmath>

namespace sd {

namespace {

constexpr double kSiteProbability = 0.001;   // q0

struct ClusterShape {
    int hits = 0;        // k
    int span = 0;        // s
    int conserved = 0;   // m
};

// orders the members by gene position on the query side (the order they are printed in) and measures the cluster
ClusterShape measure(std::vector<ClusterHit> &members) {
    ClusterShape shape;
    shape.hits = (int) members.size();
    uint32_t qLo = members[0].qPos, qHi = members[0].qPos, tLo = members[0].tPos, tHi = members[0].tPos;
    for (const ClusterHit &h : members) {
        qLo = std::min(qLo, h.qPos);
        qHi = std::max(qHi, h.qPos);
        tLo = std::min(tLo, h.tPos);
        tHi = std::max(tHi, h.tPos);
    }
    shape.span = (int) std::max(qHi - qLo, tHi - tLo) + 1;
    std::sort(members.begin(), members.end(), [](const ClusterHit &a, const ClusterHit &b) {
        return a.qPos != b.qPos ? a.qPos < b.qPos : a.idx < b.idx;
    });
    for (size_t x = 1; x < members.size(); x++) {
        const ClusterHit &left = members[x - 1], &right = members[x];
        const bool ascending = right.tPos > left.tPos;
        // a neighbour pair is conserved when both hits pair equal strands exactly if the target order is ascending
        if ((left.qS == left.tS) == ascending && (right.qS == right.tS) == ascending) shape.conserved++;
    }
    return shape;
}

double logPlacement(const double *lnFact, const ClusterShape &c) {
    const double arrangements = 2 * lnFact[c.span + 1] - 2 * lnFact[c.span - c.hits + 1] - lnFact[c.hits + 1];
    return std::fma((double) c.hits, std::log(kSiteProbability), arrangements);
}

double logOrdering(const double *lnFact, const ClusterShape &c) {
    const double tail = std::log(1 - 1.0 * c.conserved / c.hits);
    return std::fma(-(double) c.conserved, std::log(2.0), tail) - lnFact[c.conserved + 1];
}

}  // namespace

double chClusterPval(const double *lnFact, std::vector<ClusterHit> &members) {
    if (members.empty()) return 1.0;
    const ClusterShape shape = measure(members);
    return std::exp(0.5 * logPlacement(lnFact, shape) + 0.5 * logOrdering(lnFact, shape));
}

double chMultihitPval(const double *lnFact, const std::vector<ClusterHit> &members, int nQuerySet, double alpha) {
    const double logTheta = std::log(alpha / (nQuerySet + 1));
    size_t strong = 0;
    double excess = 0;
    for (const ClusterHit &h : members) {
        const double lp = std::log(h.pval);
        if (lp < logTheta) {
            strong++;
            excess -= lp - logTheta;
        }
    }
    if (excess == 0) return 1.0;           // no hit below the threshold
    if (std::isinf(excess)) return 0.0;    // a P-value of zero among them
    const double decay = std::exp(-excess);
    if (decay == 0) return 0.0;
    double series = 0;
    for (size_t i = 0; i + 1 < strong; i++) series += std::pow(excess, (double) i) / std::exp(lnFact[i + 1]);
    return decay * series;
}

}  // namespace sd
