// C-ABI wrappers of the host-side stages (sd_host_* in include/spacedust_gpu.h).
#include "sd_host.h"

#include <climits>
#include <vector>
#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include "spacedust_gpu.h"

#include <cmath>
#include <cstring>
#include <unistd.h>

struct sd_host_index {
    sd::TargetIndex idx;
};

extern "C" {

int sd_host_create(int threads, sd_host **out) {
    if (!out) return SD_EINVAL;
    sd_host *h = new sd_host();
    h->threads = threads > 0 ? threads : 1;
    sd::initSubMat(h->blosum2, sd::MAT_BLOSUM62, 2.0f, 0.0f);      // Alignment.cpp:152
    sd::initSubMat(h->ungapped2, sd::MAT_BLOSUM62, 2.0f, -0.2f);   // Prefiltering.cpp:69,991
    sd::initSubMat(h->seed8, sd::MAT_VTML80, 8.0f, -0.2f);         // Prefiltering.cpp:68,991
    *out = h;
    return SD_OK;
}

void sd_host_destroy(sd_host *h) { delete h; }

static const sd::SubMat &pick(sd_host *h, int which) {
    return which == 0 ? h->blosum2 : (which == 1 ? h->seed8 : h->ungapped2);
}

int sd_host_matrix(sd_host *h, int which, int8_t *out, double *pBack, uint8_t *aa2num) {
    const sd::SubMat &m = pick(h, which);
    for (int i = 0; i < 21; i++)
        for (int j = 0; j < 21; j++) out[i * 21 + j] = (int8_t) m.sub[i][j];
    if (pBack) memcpy(pBack, m.pBack, sizeof(double) * 21);
    if (aa2num) memcpy(aa2num, m.aa2num, 256);
    return SD_OK;
}

// the two host tables sd_target_build needs: tantan's likelihood ratios of the seed matrix (BaseMatrix.h:85-96) and the k-mer
// self scores of IndexBuilder.cpp:10-21
int sd_host_index_tables(sd_host *h, double *maskRatios /* 21 x 21 */, int8_t *selfScore /* 21 */) {
    if (!h || !maskRatios || !selfScore) return SD_EINVAL;
    sd::MaskCtx mc;
    sd::initMaskCtx(h->seed8, mc);
    for (int i = 0; i < 21; i++)
        for (int j = 0; j < 21; j++) maskRatios[i * 21 + j] = mc.lr[i][j];
    for (int a = 0; a < 21; a++) selfScore[a] = (int8_t) (char) h->seed8.sub[a][a];
    return SD_OK;
}

int sd_host_map_sequence(sd_host *h, const char *ascii, uint64_t len, uint8_t *out) {
    sd::mapSequence(h->blosum2, ascii, len, out);
    return SD_OK;
}

int sd_host_comp_bias(sd_host *h, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int kmerSize,
                      int8_t *swBias, int8_t *diagBias, int16_t *kmerBias) {
    uint8_t seedPos[8];
    const int span = sd::spacedPattern(kmerSize, seedPos);
    const double tDbg = omp_get_wtime();
#pragma omp parallel num_threads(h->threads)
    {
        std::vector<float> cbSeed;
#pragma omp for schedule(dynamic, 64)
        for (uint32_t i = 0; i < n; i++) {
            const uint8_t *s = residues + offsets[i];
            const int L = (int) (offsets[i + 1] - offsets[i]);
            if (swBias) sd::swCompBias8(h->blosum2, s, L, swBias + offsets[i]);
            if (diagBias || kmerBias) {
                // one local correction with the seed matrix serves both roundings
                cbSeed.resize(L > 0 ? L : 1);
                sd::calcLocalAaBiasCorrection(h->seed8, s, L, cbSeed.data(), 1.0f);
                if (diagBias) sd::diagCompBias8From(cbSeed.data(), L, diagBias + offsets[i]);
                if (kmerBias) {
                    for (int x = 0; x < L; x++) kmerBias[offsets[i] + x] = 0;
                    sd::kmerThrBias16From(cbSeed.data(), L, seedPos, kmerSize, span, kmerBias + offsets[i]);
                }
            }
        }
    }
    if (getenv("SD_DEBUG_TIMING")) fprintf(stderr, "[sd_host_comp_bias] %u seqs, %d threads: %.1f ms\n", n, h->threads, (omp_get_wtime() - tDbg) * 1e3);
    return SD_OK;
}

int sd_host_sw_comp_bias(sd_host *h, int which, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int8_t *swBias) {
    if (!h || !residues || !offsets || !swBias || which < 0 || which > 2) return SD_EINVAL;
    const sd::SubMat &m = pick(h, which);
#pragma omp parallel for schedule(dynamic, 64) num_threads(h->threads)
    for (uint32_t i = 0; i < n; i++)
        sd::swCompBias8(m, residues + offsets[i], (int) (offsets[i + 1] - offsets[i]), swBias + offsets[i]);
    return SD_OK;
}

int sd_host_index_build(sd_host *h, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int kmerSize,
                        int kmerThr, int mask, double maskProb, sd_host_index **out) {
    if (!out || (kmerSize != 6 && kmerSize != 7)) return SD_EINVAL;
    // positions in the index are 16 bit (IndexEntryLocal::position_j, M/src/prefiltering/IndexTable.h): sequences beyond
    // 65 535 residues (--max-seq-len) are the caller's to split, not something to wrap silently
    for (uint32_t i = 0; i < n; i++)
        if (offsets[i + 1] - offsets[i] > 65535) return SD_EINVAL;
    sd_host_index *ix = new sd_host_index();
    sd::buildTargetIndex(h->seed8, residues, offsets, n, kmerSize, kmerThr, mask != 0, maskProb, h->threads, ix->idx);
    if (ix->idx.tableSize == 0) {   // the offset table could not be allocated
        delete ix;
        return SD_ENOMEM;
    }
    *out = ix;
    return SD_OK;
}

int sd_host_auto_kmer_size(uint64_t targetResidues) { return sd::autoKmerSize(targetResidues); }

int sd_host_map_profiles(const char *profileData, const uint64_t *byteOffsets, uint32_t n, uint8_t *queryLetters,
                         uint8_t *consensus, int8_t *alnProfile, int16_t *sortedScore, uint8_t *sortedIndex,
                         uint64_t *posOffsets) {
    if (!profileData || !byteOffsets || !queryLetters || !alnProfile || !sortedScore || !sortedIndex || !posOffsets) return SD_EINVAL;
    // a profile longer than --max-seq-len (65 535) is cut there: Sequence::mapProfile reads `while (l < maxLen && l < seqLen)`
    // (M/src/commons/Sequence.cpp:247-266); the positions of the following profiles move up
    constexpr uint64_t MAX_PROFILE_LEN = 65535;
    posOffsets[0] = 0;
    for (uint32_t i = 0; i <= n; i++)
        if (byteOffsets[i] % sd::PROFILE_RECORD || (i && byteOffsets[i] < byteOffsets[i - 1])) return SD_EINVAL;
    for (uint32_t i = 0; i < n; i++)
        posOffsets[i + 1] = posOffsets[i] + std::min<uint64_t>((byteOffsets[i + 1] - byteOffsets[i]) / sd::PROFILE_RECORD, MAX_PROFILE_LEN);
#pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t i = 0; i < n; i++) {
        const uint64_t p0 = posOffsets[i];
        sd::mapProfile(profileData + byteOffsets[i], (uint32_t) (posOffsets[i + 1] - p0), queryLetters + p0,
                       consensus ? consensus + p0 : nullptr, alnProfile + p0 * 21, sortedScore + p0 * 20, sortedIndex + p0 * 20);
    }
    return SD_OK;
}

int sd_host_profile_kmer_threshold(float sensitivity, int kmerSize) { return sd::profileKmerThreshold(sensitivity, kmerSize); }

int sd_host_index_info(sd_host_index *ix, uint64_t *tableSize, uint64_t *nEntries, uint64_t *maskedResidues) {
    if (tableSize) *tableSize = ix->idx.tableSize;
    if (nEntries) *nEntries = ix->idx.entrySeq.size();
    if (maskedResidues) *maskedResidues = ix->idx.maskedResidues;
    return SD_OK;
}

int sd_host_index_arrays(sd_host_index *ix, const uint32_t **kmerOffsets, const uint32_t **entrySeq,
                         const uint16_t **entryPos, const uint8_t **maskedResidues) {
    if (kmerOffsets) *kmerOffsets = ix->idx.offsets.data();
    if (entrySeq) *entrySeq = ix->idx.entrySeq.data();
    if (entryPos) *entryPos = ix->idx.entryPos.data();
    if (maskedResidues) *maskedResidues = ix->idx.masked.data();
    return SD_OK;
}

int sd_host_index_block_base(sd_host_index *ix, const uint64_t **blockBase, uint64_t *nBlocks) {
    if (!ix) return SD_EINVAL;
    if (blockBase) *blockBase = ix->idx.blockBase.empty() ? nullptr : ix->idx.blockBase.data();
    if (nBlocks) *nBlocks = ix->idx.blockBase.size();
    return SD_OK;
}

void sd_host_index_destroy(sd_host_index *ix) { delete ix; }

int sd_host_ext_matrix(sd_host *h, int wordLen, const int16_t **score, const uint16_t **index, uint32_t *size) {
    if (wordLen == 2) {
        if (!h->haveTwo) {
            sd::buildExtMatrix(h->seed8, 2, h->two, h->threads);
            h->haveTwo = true;
        }
        *score = h->two.score.data(); *index = h->two.index.data(); *size = h->two.size;
        return SD_OK;
    }
    if (wordLen == 3) {
        if (!h->haveThree) {
            sd::buildExtMatrix(h->seed8, 3, h->three, h->threads);
            h->haveThree = true;
        }
        *score = h->three.score.data(); *index = h->three.index.data(); *size = h->three.size;
        return SD_OK;
    }
    return SD_EINVAL;
}

int sd_host_kmer_threshold(float sensitivity, int kmerSize) { return sd::kmerThreshold(sensitivity, kmerSize); }

unsigned sd_host_bin_size(uint64_t dbSize, uint64_t l2CacheSize) {
    if (l2CacheSize == 0) {
        // Util::getL2CacheSize (M/src/commons/Util.cpp:317-332)
        long v = -1;
#ifdef _SC_LEVEL2_CACHE_SIZE
        v = sysconf(_SC_LEVEL2_CACHE_SIZE);
#endif
        l2CacheSize = v > 0 ? (uint64_t) v : 262144;
    }
    return sd::diagonalBinSize(dbSize, l2CacheSize);
}

// (query, target) pairs in prefilter order from the row-per-query hit table (Alignment.cpp:346-379 reads them in
// this order); returns the number of pairs, pairQ/pairT may be NULL to only count
uint64_t sd_host_pair_list(const sd_hit *hits, const uint32_t *counts, uint32_t nQ, uint32_t rowWidth, uint32_t *pairQ,
                           uint32_t *pairT) {
    std::vector<uint64_t> start((size_t) nQ + 1, 0);
    for (uint32_t q = 0; q < nQ; q++) start[q + 1] = start[q] + counts[q];
    if (pairQ && pairT) {
#pragma omp parallel for schedule(static)
        for (uint32_t q = 0; q < nQ; q++) {
            const sd_hit *row = hits + (size_t) q * rowWidth;
            uint64_t w = start[q];
            for (uint32_t x = 0; x < counts[q]; x++, w++) {
                pairQ[w] = q;
                pairT[w] = row[x].seqId;
            }
        }
    }
    return start[nQ];
}

static double chLogGamma(double x) {
    // Lanczos approximation of R/src/util/ClusterHits.cpp:23-63
    static const double r10 = 10.900511;
    static const double dk[11] = {2.48574089138753565546e-5, 1.05142378581721974210, -3.45687097222016235469,
                                  4.51227709466894823700, -2.98285225323576655721, 1.05639711577126713077,
                                  -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
                                  4.63399473359905636708e-6, -2.71994908488607703910e-9};
    static const double gc = 2 * sqrt(exp(1.0) / M_PI);
    if (x < 0.5) return log(M_PI) - log(std::abs(sin(M_PI * x))) - chLogGamma(1 - x);
    if (x == 1) return 0.0;
    double sum = dk[0];
    sum += dk[1] / (x + 0);
    sum += dk[2] / (x + 1);
    sum += dk[3] / (x + 2);
    sum += dk[4] / (x + 3);
    sum += dk[5] / (x + 4);
    sum += dk[6] / (x + 5);
    sum += dk[7] / (x + 6);
    sum += dk[8] / (x + 7);
    sum += dk[9] / (x + 8);
    sum += dk[10] / (x + 9);
    return log(gc) + (x - 0.5) * log(x + r10 - 0.5) - (x - 0.5) + log(sum);
}

int sd_host_cluster_pvalues(uint32_t nHits, const uint32_t *qPos, const uint32_t *tPos, const uint8_t *strands, const double *pval,
                            uint32_t querySetSize, double alpha, const double *lGamma, uint32_t lGammaLen, double *pCluster,
                            double *pMultihit, uint32_t *order) {
    if (!nHits || !qPos || !tPos || !strands || !pval || !lGamma || !pCluster || !pMultihit) return SD_EINVAL;
    std::vector<sd::ClusterHit> c(nHits);
    uint32_t lo = UINT32_MAX, hi = 0;
    for (uint32_t x = 0; x < nHits; x++) {
        c[x].pval = pval[x];
        c[x].qPos = qPos[x];
        c[x].tPos = tPos[x];
        c[x].qS = strands[x] & 1;
        c[x].tS = (strands[x] >> 1) & 1;
        c[x].idx = x;
        lo = std::min(lo, std::min(qPos[x], tPos[x]));
        hi = std::max(hi, std::max(qPos[x], tPos[x]));
    }
    if ((uint64_t) (hi - lo) + 2 >= lGammaLen || (uint64_t) nHits + 1 >= lGammaLen) return SD_EINVAL;   // the table covers span + 1 and hits + 1
    *pCluster = sd::chClusterPval(lGamma, c);
    *pMultihit = sd::chMultihitPval(lGamma, c, (int) querySetSize, alpha);
    if (order)
        for (uint32_t x = 0; x < nHits; x++) order[x] = c[x].idx;
    return SD_OK;
}

int sd_host_lgamma_table(double *out, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) out[i] = chLogGamma(i * 1.0);
    return SD_OK;
}

double sd_host_evalue(uint64_t dbResidues, double score, double qLen) {
    sd::Evaluer e;
    sd::initEvaluer(e, dbResidues);
    return sd::computeEvalue(e, score, qLen);
}

int sd_host_can_be_covered(float covThr, int covMode, float queryLength, float targetLength) {
    return sd::canBeCovered(covThr, covMode, queryLength, targetLength) ? 1 : 0;
}

double sd_host_bitscore(double score) {
    sd::Evaluer e;
    sd::initEvaluer(e, 1);
    return sd::computeBitScore(e, score);
}

}  // extern "C"
