// P-values of the emitted clusters (R/src/util/ClusterHits.cpp:80-134,184-213,462-464) in double precision with the
// reference's own expressions.  This file is compiled by g++ with the reference's AVX2 build flags (-mfma and GCC's default
// -ffp-contract=fast, spacedust_amd/build.py HOST_FLAGS): which products fuse into an FMA decides the last bit of
// `-0.5 * logpClu - 0.5 * logpOrd` and of `... + k * log(q0)`, and the reference prints these values with %.3E.
// Pinned against the reference's compiled functions (oracle/_ref/libsdref_ch.so) in tests/test_gpu_clusterhits.py.
#include "sd_host.h"

#include <algorithm>
#include <climits>
#include <cmath>

namespace sd {

namespace {


double hLogClusterPval(const double *lookup, int k, int m, double q0 = 0.001) {
    return 2 * lookup[m + 1] - 2 * lookup[m - k + 1] - lookup[k + 1] + k * log(q0);
}
double hLogOrderingPval(const double *lookup, int k, int m) { return log(1 - 1.0 * m / k) - m * log(2) - lookup[m + 1]; }

double hClusterMatchScore(const double *lookup, std::vector<ClusterHit> &c) {
    if (c.size() == 0) return 0.0;
    unsigned int iMax = 0, iMin = INT_MAX, jMax = 0, jMin = INT_MAX;
    for (size_t l = 0; l < c.size(); l++) {
        iMax = (c[l].qPos > iMax) ? c[l].qPos : iMax;
        iMin = (c[l].qPos < iMin) ? c[l].qPos : iMin;
        jMax = (c[l].tPos > jMax) ? c[l].tPos : jMax;
        jMin = (c[l].tPos < jMin) ? c[l].tPos : jMin;
    }
    int spanI = iMax - iMin + 1, spanJ = jMax - jMin + 1;
    int span = (spanI > spanJ) ? spanI : spanJ;
    int k = (int) c.size();
    std::sort(c.begin(), c.end(), [](const ClusterHit &a, const ClusterHit &b) {
        if (a.qPos != b.qPos) return a.qPos < b.qPos;
        return a.idx < b.idx;
    });
    int m = 0;
    for (size_t l = 0; l + 1 < c.size(); l++) {
        bool isSameOrder = (c[l + 1].tPos > c[l].tPos);
        bool s1 = (c[l].qS == c[l].tS), s2 = (c[l + 1].qS == c[l + 1].tS);
        if ((s1 == isSameOrder) && (s2 == isSameOrder)) m++;
    }
    double logpClu = hLogClusterPval(lookup, k, span);
    double logpOrd = hLogOrderingPval(lookup, k, m);
    return -0.5 * logpClu - 0.5 * logpOrd;
}

double hMultihitPval(const double *lookup, const std::vector<ClusterHit> &cluster, int Nq, double alpha) {
    size_t k = 0;
    double r = 0;
    double pvalThreshold = alpha / (Nq + 1);
    double logPvalThr = log(pvalThreshold);
    for (size_t i = 0; i < cluster.size(); ++i) {
        double logPvalue = log(cluster[i].pval);
        if (logPvalue < logPvalThr) {
            k++;
            r -= logPvalue - logPvalThr;
        }
    }
    if (r == 0) return 1.0;
    if (std::isinf(r)) return 0.0;
    double expMinusR = exp(-r);
    if (expMinusR == 0) return 0.0;
    double sum = 0;
    for (size_t i = 0; i < k - 1; ++i) sum += pow(r, i) / exp(lookup[i + 1]);
    return expMinusR * sum;
}


}  // namespace

double chClusterPval(const double *lookup, std::vector<ClusterHit> &cluster) { return exp(-hClusterMatchScore(lookup, cluster)); }
double chMultihitPval(const double *lookup, const std::vector<ClusterHit> &cluster, int Nq, double alpha) {
    return hMultihitPval(lookup, cluster, Nq, alpha);
}

}  // namespace sd
