// spacedust_amd host library: the host-side (CPU, C++) support code of the
// clustersearch hot path.  Nothing here is a fallback for a GPU kernel: these are the
// stages that stay on the host by design (matrix derivation, float composition bias,
// target masking + index construction, E-values, text formatting).
//
// Reference files are cited as M/ = /root/reference/lib/mmseqs/, R/ = /root/reference/.
#ifndef SD_HOST_H
#define SD_HOST_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <cstdlib>
#include <vector>

namespace sd {

enum { ALPH = 21, X_CODE = 20 };

// ---------------------------------------------------------------------------
// Substitution matrices (M/src/commons/SubstitutionMatrix.cpp:12-58,327-418,
// M/src/commons/BaseMatrix.cpp:97-159)
// ---------------------------------------------------------------------------
struct SubMat {
    int alphabetSize;
    float bitFactor;
    double lambda;
    double pBack[ALPH];            // member pBack (file background, rescaled by 1-pX)
    double probMatrix[ALPH][ALPH]; // joint probabilities reconstructed from the scores
    short sub[ALPH][ALPH];         // integer scoring matrix
    uint8_t aa2num[256];
    char num2aa[ALPH + 1];
    const char *name;              // "blosum62.out" / "VTML80.out"
};
enum MatrixId { MAT_BLOSUM62 = 0, MAT_VTML80 = 1 };
void initSubMat(SubMat &m, MatrixId id, float bitFactor, float scoreBias);
void mapSequence(const SubMat &m, const char *seq, size_t len, uint8_t *out);

// float composition bias (M/src/commons/SubstitutionMatrix.cpp:79-109)
void calcLocalAaBiasCorrection(const SubMat &m, const uint8_t *seq, int N, float *out, float scale);
// the three integer roundings used downstream
void swCompBias8(const SubMat &blosum2, const uint8_t *seq, int N, int8_t *out);              // StripedSmithWaterman.cpp:1231-1235
void diagCompBias8(const SubMat &seed8, const uint8_t *seq, int N, int8_t *out);              // UngappedAlignment.cpp:392-396
void kmerThrBias16(const SubMat &seed8, const uint8_t *seq, int N, const uint8_t *seedPos, int k,
                   int span, int16_t *out /* N-span+1 */);
void diagCompBias8From(const float *cb, int N, int8_t *out);
void kmerThrBias16From(const float *cb, int N, const uint8_t *seedPos, int k, int span, int16_t *out);                                    // QueryMatcher.cpp:230-240

// ---------------------------------------------------------------------------
// Extended (2-mer / 3-mer) substitution tables (M/src/prefiltering/ExtendedSubstitutionMatrix.cpp:20-69)
// rows are exactly `size` long here (no SIMD padding); score desc, stable.
// ---------------------------------------------------------------------------
struct ExtMatrix {
    int wordLen;
    uint32_t size;                 // 20^wordLen
    std::vector<int16_t> score;    // [size][size]
    std::vector<uint16_t> index;   // [size][size] (word index < 8000)
};
void buildExtMatrix(const SubMat &seed8, int wordLen, ExtMatrix &out, int threads);

// the complete (residue, window length, window sum) -> correction table of calcLocalAaBiasCorrection's float tail
// (every entry evaluated with the reference's expression order): tab[(residue * 41 + windowLength) * span + (sum - lo)];
// the device-side composition bias forms the integer sums and reads the corrections from here
void biasTableFull(const SubMat &m, std::vector<float> &tab, int &lo, int &span);

// spaced seed patterns (M/src/commons/Sequence.h:22-25)
int spacedPattern(int k, uint8_t *pos /* k entries */);   // returns span

// similar k-mer enumeration, host version (M/src/prefiltering/KmerGenerator.cpp:107-216); k = 6 or 7
size_t generateKmerList(const ExtMatrix &three, const ExtMatrix &two, int k, const uint8_t *window, int thr,
                        std::vector<uint32_t> &out);

// ---------------------------------------------------------------------------
// Profile queries (M/src/commons/Sequence.cpp:241-305, Sequence.h:458-471)
// ---------------------------------------------------------------------------
constexpr int PROFILE_RECORD = 25;   // bytes per position: 20 scores, query letter, consensus letter, neff, 2 reserved
// one profile of L positions -> query letters, consensus letters (nullable), alignment profile int8 [L][21]
// (score / 4, X column 0), k-mer generator rows: scores sorted descending by the reference's sorting network and
// the amino acids in that order, [L][20] each
void mapProfile(const char *data, uint32_t L, uint8_t *letters, uint8_t *consensus, int8_t *aln, int16_t *sortedScore,
                uint8_t *sortedIndex);
// similar k-mers of one window: score[s] / index[s] = sorted row of seed position s (s < k)
size_t generateProfileKmerList(const int16_t *const *score, const uint8_t *const *index, int k, int thr,
                               std::vector<uint32_t> &out);
int profileKmerThreshold(float sensitivity, int k);   // Prefiltering.cpp:1031-1043 (no context pseudo counts)

// ---------------------------------------------------------------------------
// tantan repeat masking (M/lib/tantan/tantan.cpp, M/src/commons/Masker.cpp:15-55)
// ---------------------------------------------------------------------------
struct MaskCtx {
    double lr[ALPH][ALPH];         // likelihood ratio matrix (BaseMatrix.h:83-96)
};
void initMaskCtx(const SubMat &seed8, MaskCtx &ctx);
int tantanMask(const MaskCtx &ctx, uint8_t *seq, int L, double minMaskProb);

// ---------------------------------------------------------------------------
// Target k-mer index (M/src/prefiltering/IndexTable.h, IndexBuilder.cpp:55-239)
// Layout is ours (HBM-friendly): u32 offsets (relative to a u64 base per 65 536 k-mers once nEntries >= 2^32), SoA entries.
// ---------------------------------------------------------------------------
// calloc-backed uint32 array: a fresh 20^k table (k = 7: 5 GB) comes as untouched zero pages instead of being
// written once by a constructor; pages are first touched by the parallel passes that fill it
struct ZeroedU32 {
    uint32_t *p = nullptr;
    size_t n = 0;
    ZeroedU32() {}
    ZeroedU32(const ZeroedU32 &) = delete;
    ZeroedU32 &operator=(const ZeroedU32 &) = delete;
    ~ZeroedU32() { free(p); }
    bool reset(size_t count) {
        free(p);
        p = count ? (uint32_t *) calloc(count, sizeof(uint32_t)) : nullptr;
        n = p ? count : 0;
        return count == 0 || p != nullptr;
    }
    void shrink(size_t count) { n = count; }
    uint32_t *data() { return p; }
    const uint32_t *data() const { return p; }
    size_t size() const { return n; }
    uint32_t &operator[](size_t i) { return p[i]; }
    const uint32_t &operator[](size_t i) const { return p[i]; }
};

struct TargetIndex {
    int k, span;
    uint8_t seedPos[8];
    uint64_t tableSize;                 // 20^k
    ZeroedU32 offsets;                  // tableSize+1: list starts -- absolute while the index has < 2^32 entries
                                        // (blockBase empty), else relative to blockBase[kmer >> 16]
    std::vector<uint64_t> blockBase;    // wide indexes only: one base per 65 536 k-mers (start = base + offset)
    uint64_t nEntries;
    std::vector<uint32_t> entrySeq;     // nEntries
    std::vector<uint16_t> entryPos;     // nEntries
    std::vector<uint8_t> masked;        // concatenated masked numeric residues
    std::vector<uint64_t> seqOffsets;   // nSeq+1
    uint64_t maskedResidues;
};
// seqs: concatenated numeric (unmasked) residues
void buildTargetIndex(const SubMat &seed8, const uint8_t *seqs, const uint64_t *offsets, uint32_t nSeq, int k,
                      int kmerThr, bool mask, double maskProb, int threads, TargetIndex &out);
int kmerThreshold(float sensitivity, int k);                          // Prefiltering.cpp:1005-1065 (seq-seq rows)
int autoKmerSize(uint64_t targetResidues);                            // IndexTable.h:439-449
unsigned diagonalBinSize(uint64_t dbSize, uint64_t l2CacheSize);      // QueryMatcher.cpp:422-450

// ---------------------------------------------------------------------------
// E-values (M/src/alignment/EvalueComputation.h, M/lib/alp/sls_pvalues.cpp:366-520)
// blosum62, gap open 11 / extend 1 preset only.
// ---------------------------------------------------------------------------
struct Evaluer {
    double lambda, K, logK;
    double aI, bI, alphaI, betaI, aJ, bJ, alphaJ, betaJ, sigma, tau;
    double viThr, vjThr, cThr;
    double dbResidues;
};
void initEvaluer(Evaluer &e, uint64_t dbResidues);
double computeEvalue(const Evaluer &e, double score, double qLen);
double computeBitScore(const Evaluer &e, double score);

// ---------------------------------------------------------------------------
// Alignment post-processing / text (M/src/alignment/Matcher.cpp:100-137,166-185,280-327;
// M/src/commons/Util.cpp:222-251,477-540)
// ---------------------------------------------------------------------------
bool canBeCovered(float covThr, int covMode, float qLen, float tLen);
bool hasCoverage(float covThr, int covMode, float qCov, float tCov);
float computeCov(unsigned start, unsigned end, unsigned len);
char *u32toa(uint32_t v, char *buf);       // returns pointer past the last digit
char *i32toa(int32_t v, char *buf);
char *seqIdToBuffer(float seqId, char *buf);
std::string compressBacktrace(const char *bt, size_t n);
void compressBacktraceAppend(const char *bt, size_t n, std::string &out);

// ---------------------------------------------------------------------------
// clusterhits: P-values of an emitted cluster (R/src/util/ClusterHits.cpp:120-134,184-213,462-464), sd_chpval.cpp
// ---------------------------------------------------------------------------
struct ClusterHit {
    double pval;
    uint32_t qPos, tPos;
    bool qS, tS;
    uint32_t idx;
};
// exp(-clusterMatchScore): sorts `cluster` by query position like the reference's findConservedPairs does
double chClusterPval(const double *lgammaTable, std::vector<ClusterHit> &cluster);
double chMultihitPval(const double *lgammaTable, const std::vector<ClusterHit> &cluster, int Nq, double alpha);

}  // namespace sd

// the host-side handle behind sd_host_* (include/spacedust_gpu.h); the device code reads the matrices from it
struct sd_host {
    sd::SubMat blosum2, ungapped2, seed8;
    sd::ExtMatrix two, three;
    bool haveTwo = false, haveThree = false;
    int threads = 1;
    // composition-bias tables for the device path (built on first use)
    std::vector<float> biasTabSeed, biasTabBlosum;
    int biasLoSeed = 0, biasSpanSeed = 0, biasLoBlosum = 0, biasSpanBlosum = 0;
};
#endif
