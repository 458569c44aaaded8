// TEST INFRASTRUCTURE ONLY (never linked into or loaded by the product).
//
// oracle/_ref/libsdref_ch.so: the reference's own clusterhits arithmetic behind flat arrays.  The translation unit
// below IS R/src/util/ClusterHits.cpp, compiled where it lies (oracle/Makefile passes -I$(REF)/src/util; nothing is copied):
// logGamma, logClusterPval, logOrderingPval, findSpan, findConservedPairs, clusterMatchScore, isCompatibleCluster,
// groupNodes and multihitPval are the reference's compiled functions.  Its monolithic `clusterhits()` entry point needs
// DBReader/DBWriter/LocalParameters (the latter unbuildable here: cmake-generated headers), so this TU is compiled with
// hidden visibility + function sections and linked with --gc-sections: the unreferenced entry point and its undefined
// externals are dropped, the helper functions stay.  What drives them here is the merge loop of
// ClusterHits.cpp:363-485 re-driven over flat arrays (same first-index arg-max, merge-then-test do/while, `j != 0` reset
// and stale dmin), one call per (query set, target set) entry.
#include "ClusterHits.cpp"

#include <cstdint>
#include <cstring>

extern "C" __attribute__((visibility("default")))
double ref_ch_loggamma(double x) { return logGamma(x); }

// Outputs per hit: clusterOf (ordinal of the emitted cluster, UINT32_MAX none), rank (position in the cluster's printed
// order = ascending query position, ClusterHits.cpp:104-106,462); per emitted cluster c: pCO[c], pMH[c], size[c].
// lg: logGamma table the entry is scored with (lgN entries), as ClusterHits.cpp:267-271 builds it.
extern "C" __attribute__((visibility("default")))
int ref_clusterhits_entry(uint32_t K, const uint32_t *qPos, const uint32_t *tPos, const uint8_t *strands, const double *pval,
                          uint32_t Nq, uint32_t d, uint32_t cls, double alpha, float pCluThr, float pMHThr, double *lg,
                          uint32_t *clusterOf, uint32_t *rank, double *pCO, double *pMH, uint32_t *size, uint32_t *nClusters) {
    *nClusters = 0;
    for (uint32_t i = 0; i < K; i++) {
        clusterOf[i] = UINT32_MAX;
        rank[i] = 0;
    }
    if (K <= 1) return 0;
    std::vector<hit> match(K);
    for (uint32_t i = 0; i < K; i++) {
        match[i].alignment = std::to_string(i);   // identifies the hit after the reference's sort by query position
        match[i].pval = pval[i];
        match[i].qPos = qPos[i];
        match[i].tPos = tPos[i];
        match[i].qStrand = (strands[i] & 1) != 0;
        match[i].tStrand = (strands[i] & 2) != 0;
    }
    std::vector<std::vector<double> > D(K, std::vector<double>(K, 0.0));
    std::vector<int> dmin(K);
    std::vector<std::vector<int> > nodes(K);
    for (uint32_t n = 0; n < K; n++) nodes[n].push_back((int) n);
    for (uint32_t i = 0; i < K; i++)
        for (uint32_t j = 0; j < K; j++) {
            if (i != j) {
                std::vector<hit> tmp = groupNodes(nodes, match, (int) i, (int) j, d);
                D[i][j] = clusterMatchScore(lg, tmp);
            }
            if (D[i][j] > D[i][dmin[i]]) dmin[i] = (int) j;
        }
    const double sMin = -0.5 * logClusterPval(lg, 2, (int) d + 1) - 0.5 * logOrderingPval(lg, 2, 1);
    double maxScore;
    do {
        size_t i1 = 0;
        for (size_t i = 0; i < K; i++)
            if (D[i][dmin[i]] > D[i1][dmin[i1]]) i1 = i;
        const size_t i2 = (size_t) dmin[i1];
        maxScore = D[i1][i2];
        if (maxScore == 0) break;
        nodes[i1].insert(nodes[i1].end(), nodes[i2].begin(), nodes[i2].end());
        nodes[i2].clear();
        for (size_t j = 0; j < K; j++) {
            if (i1 == j || i2 == j) {
                D[i1][j] = 0.0;
                D[j][i1] = 0.0;
            } else {
                std::vector<hit> tmp = groupNodes(nodes, match, (int) i1, (int) j, d);
                D[i1][j] = clusterMatchScore(lg, tmp);
                D[j][i1] = D[i1][j];
            }
            D[i2][j] = 0.0;
            D[j][i2] = 0.0;
            if (j != 0) {
                if (D[i1][j] > D[i1][dmin[i1]]) dmin[i1] = (int) j;
            } else {
                dmin[i1] = (int) j;
            }
            if (j != i1 && j != i2 && D[j][i1] > D[j][dmin[j]]) dmin[j] = (int) i1;
        }
    } while (maxScore >= sMin);
    uint32_t nc = 0;
    for (size_t i = 0; i < nodes.size(); i++) {
        if (nodes[i].size() < cls) continue;
        std::vector<hit> cluster;
        for (size_t j = 0; j < nodes[i].size(); j++) cluster.push_back(match[nodes[i][j]]);
        const double co = exp(-clusterMatchScore(lg, cluster));   // sorts `cluster` by query position
        const double mh = multihitPval(lg, cluster, (int) Nq, alpha);
        if (co <= pCluThr && mh <= pMHThr) {
            for (size_t m = 0; m < cluster.size(); m++) {
                const uint32_t h = (uint32_t) std::stoul(cluster[m].alignment);
                clusterOf[h] = nc;
                rank[h] = (uint32_t) m;
            }
            pCO[nc] = co;
            pMH[nc] = mh;
            size[nc] = (uint32_t) cluster.size();
            nc++;
        }
    }
    *nClusters = nc;
    return 0;
}
