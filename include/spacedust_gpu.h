/* spacedust_gpu.h -- C ABI of the MI355X-native clustersearch hot path (libsdgpu.so).
 *
 * One handle (sd_ctx) per GPU / per process.  The caller owns every host buffer it passes in
 * or receives; the library owns device memory behind opaque handles.  Every function returns
 * 0 on success or a negative SD_E* code and never exits or throws; sd_last_error() gives a
 * message.  There is NO CPU fallback behind any entry point: without a visible HIP device
 * sd_ctx_create fails with SD_ENODEVICE.
 *
 * Each entry point names the reference interface it replaces (M/ = lib/mmseqs/ of
 * soedinglab/spacedust, R/ = the repository root); INTEGRATION.md shows the call sites a
 * maintainer would patch.
 */
#ifndef SPACEDUST_GPU_H
#define SPACEDUST_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    SD_OK = 0,
    SD_ENODEVICE = -1,   /* no HIP device / HIP runtime error at start-up */
    SD_EHIP = -2,        /* HIP runtime error */
    SD_EINVAL = -3,      /* bad argument */
    SD_ENOMEM = -4,      /* device or host allocation failed */
    SD_EUNSUPPORTED = -5,/* a reference code path that is not implemented on the device (fails loudly) */
    SD_EMISMATCH = -6    /* forward/backward SW score mismatch (fatal in the reference too) */
};

typedef struct sd_ctx sd_ctx;
typedef struct sd_seqset sd_seqset;
typedef struct sd_target sd_target;

/* ---- context ------------------------------------------------------------------------- */
int sd_ctx_create(int device, sd_ctx **out);
/* priority < 0 / 0 / > 0: the context's stream gets the device's highest / middle / lowest stream priority (two
 * contexts on one device share it: e.g. the memory-bound prefilter ahead of the VALU-bound alignments) */
int sd_ctx_create_prio(int device, int priority, sd_ctx **out);
/* as sd_ctx_create_prio, and the context's stream may not use `reserveCUs` of the device's compute units (a CU mask with that
 * many bits cleared, spread over the XCDs): kernels of the other contexts of the device find those free.  The pipeline keeps the
 * long-lived one-wavefront workgroups of the alignment stage off a few CUs so that the prefilter's many short dependent launches
 * start at once instead of waiting for a generation of alignment wavefronts to retire. */
int sd_ctx_create_masked(int device, int priority, int reserveCUs, sd_ctx **out);
void sd_ctx_destroy(sd_ctx *ctx);
const char *sd_last_error(sd_ctx *ctx);
int sd_device_name(sd_ctx *ctx, char *buf, size_t cap);
/* free / total device memory right now (hipMemGetInfo) */
int sd_device_memory(sd_ctx *ctx, uint64_t *freeBytes, uint64_t *totalBytes);
/* A number that identifies the physical GPU behind a device ordinal across processes and nodes (hash of the host name and the
 * PCI bus id): ranks compare it to learn whether they share a device (RCCL refuses two ranks per GPU). */
int sd_device_identity(int device, uint64_t *id);
/* The context's persistent workspaces (device buffers that live from call to call, and pinned host staging): their total
 * sizes and, if buf is not NULL, a text table "key bytes\n" of the device ones, largest first (cut at cap - 1 characters). */
int sd_workspace_report(sd_ctx *ctx, uint64_t *deviceBytes, uint64_t *pinnedBytes, char *buf, size_t cap);
/* Frees the context's device workspaces (they grow again on demand): between phases of very different shape. */
int sd_workspace_release(sd_ctx *ctx);
int sd_synchronize(sd_ctx *ctx);

/* Kernel timing with HIP events on the library's own stream (bench.py roofline leg).
 * sd_profile_enable(ctx, 1) makes every launch record start/stop events; sd_profile_get returns the
 * accumulated time and launch count of the kernel family `name` ("sw_score", "prefilter_gather", ...). */
int sd_profile_enable(sd_ctx *ctx, int on);
int sd_profile_reset(sd_ctx *ctx);
int sd_profile_get(sd_ctx *ctx, const char *name, double *totalMs, uint64_t *launches);
int sd_profile_names(sd_ctx *ctx, char *buf, size_t cap); /* comma separated */

/* ---- sequence sets resident in HBM ----------------------------------------------------
 * residues: numeric amino acids (0..20, X = 20; M/src/commons/Sequence.cpp:307-324), concatenated;
 * offsets[n+1].  Replaces DBReader::getData + Sequence::mapSequence on the device side
 * (M/src/alignment/Alignment.cpp:339,362).  swCompBias (nullable): per residue int8 composition bias of
 * SmithWaterman::ssw_init (M/src/alignment/StripedSmithWaterman.cpp:1230-1235); NULL = all zero. */
int sd_seqset_create(sd_ctx *ctx, const uint8_t *residues, const uint64_t *offsets, uint32_t n,
                     const int8_t *swCompBias, sd_seqset **out);
/* Profile queries for sd_sw_align_batch*: the set a Matcher::initQuery with a profile Sequence stands for
 * (ssw_init's PROFILE branch, StripedSmithWaterman.cpp:1238-1301).  queryLetters take the residues' place (identity
 * counting, :558), alnProfile (total x 21 int8) replaces "matrix row + composition bias" in the score, start-position
 * and traceback kernels; pass the returned set as `queries`.  Identity pairs (scoreIdentical, :1675-1710) sum the
 * profile along the main diagonal. */
int sd_profileset_create(sd_ctx *ctx, const uint8_t *queryLetters, const uint64_t *offsets, uint32_t n,
                         const int8_t *alnProfile, sd_seqset **out);
/* A set is destroyed before the context it was made with: its device buffers return to that context's pool (the next set of a
 * similar size takes them over; sd_ctx_destroy / sd_workspace_release free them).  No call that uses the set may be in flight on any
 * context when it is destroyed (every sd_* call returns with its work finished, so this only concerns sets shared between threads). */
void sd_seqset_destroy(sd_seqset *s);

/* ---- Smith-Waterman (align module) ---------------------------------------------------- */
typedef struct {
    int32_t gapOpen;       /* 11 */
    int32_t gapExtend;     /* 1 */
    int8_t matrix[21 * 21];/* blosum62 at 2 bit-factor, scoreBias 0 (Alignment.cpp:152) */
    int32_t covMode;       /* Parameters::COV_MODE_* ; clustersearch: 2 (query) */
    float covThr;          /* 0.8 */
    double evalThr;        /* 10 */
    int32_t swMode;        /* Matcher::SCORE_ONLY 0 / SCORE_COV 1 / SCORE_COV_SEQID 2 */
    uint64_t dbResidues;   /* tdbr->getAminoAcidDBSize() (Alignment.cpp:263) */
} sd_sw_params;

typedef struct {
    int32_t score;         /* s_align.score1 */
    int32_t qStart, qEnd, tStart, tEnd; /* -1 where the reference leaves them unset */
    int32_t identical;     /* identicalAACnt (0 when no backtrace) */
    int32_t btLen;         /* backtrace length (alnLength) */
    int32_t flags;         /* bit0: word (int16) kernel semantics used; bit1: fwd/bwd mismatch; bits 8..: length of the run-length
                            * text at btOffset after sd_sw_set_cigar_pool(ctx, 1), 0 otherwise */
    double evalue;         /* bit-exact EvalueComputation value when <= 2*evalThr (everything the reference can report);
                            * above that the device-evaluated value (same formula, relative error < 1e-12) */
    uint64_t btOffset;     /* offset of the expanded backtrace (chars M/I/D) in the pool */
} sd_sw_result;

/* Replaces matcher.initQuery + matcher.getSWResult for a batch of (query,target) pairs
 * (M/src/alignment/Alignment.cpp:340,379 -> Matcher.cpp:60 -> StripedSmithWaterman.cpp:310-545).
 * isIdentity[i] != 0 selects scoreIdentical (StripedSmithWaterman.cpp:1675).  btPool receives the
 * backtraces (caller allocates btCap bytes; *btUsed is set; SD_ENOMEM if too small). */
int sd_sw_align_batch(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                      uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint8_t *isIdentity,
                      sd_sw_result *out, char *btPool, uint64_t btCap, uint64_t *btUsed);

/* sd_sw_align_batch for callers that only consume what Alignment::run can write (swMode 2): returns the records of the
 * identity pairs and of the pairs that were not stopped at an E-value / coverage gate -- out[x] belongs to pair
 * outIdx[x], x < *nOut, pair order preserved; every other pair fails checkCriteria (Alignment.cpp:548-567) anyway.
 * out / outIdx must hold nPairs entries. */
int sd_sw_align_batch_compact(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                              uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint8_t *isIdentity,
                              uint32_t *outIdx, sd_sw_result *out, uint32_t *nOut, char *btPool, uint64_t btCap,
                              uint64_t *btUsed);

/* sd_sw_align_batch_compact with the prefilter's diagonal of every pair (sd_hit.diagonal; Matcher::getSWResult receives it too,
 * Alignment.cpp:379).  Same results.  The diagonal lets the library skip work whose outcome is certain: a pair whose ungapped
 * score along that diagonal -- a lower bound of the byte kernel's maximum -- already saturates the byte kernel goes straight to
 * the word-kernel pass the reference would rerun it with (StripedSmithWaterman.cpp:360-368).  pairDiag may be NULL; an entry of
 * 0x8000 (no sequence pair below 32768 residues has that diagonal) means "no hint for this pair". */
int sd_sw_align_batch_compact_diag(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                                   uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint16_t *pairDiag,
                                   const uint8_t *isIdentity, uint32_t *outIdx, sd_sw_result *out, uint32_t *nOut, char *btPool,
                                   uint64_t btCap, uint64_t *btUsed);

/* Same contract and results as sd_sw_align_batch, with the gating / task building between the passes done on
 * the host (one device round trip per pass).  Kept as the A/B cross-check of the device-resident orchestration
 * (tests/test_gpu_sw.py); btOffset values may differ, the bytes they point to may not. */
/* ---- besthitbyset on the device (R/src/util/besthitbyset.cpp:41-144 with --simple-best-hit 1) ----------------------
 * sd_seqset_set_groups: the target set of every sequence of a (target) sequence set (createsetdb's set membership) and,
 * optionally, its DB key (Matcher::compareHits' last criterion; NULL: the index), resident beside the sequences.
 * sd_sw_align_batch_best_by_group = sd_sw_align_batch_compact_diag whose returned records are restricted to what besthitbyset
 * can keep: identity pairs, and per (query, target set) the ONE accepted alignment that Matcher::compareHits (Matcher.h:157-168)
 * orders first -- smallest E-value, i.e. for one query the largest score, then the shorter target, then the smaller key.
 * "Accepted" is Alignment::checkCriteria (Alignment.cpp:548-567) with the E-value, coverage mode / threshold of `par` and the
 * sequence-identity and alignment-length thresholds given here.  Every other pair is computed as before but its record and its
 * backtrace stay on the device (at proteome scale: a third of the records, more where a target set holds paralogs).  The caller's
 * own best-hit selection over what comes back gives what it gave over all records. */
int sd_seqset_set_groups(sd_seqset *s, const uint32_t *groupOf, uint32_t nGroups, const uint32_t *keys);
int sd_sw_align_batch_best_by_group(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                                    uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint16_t *pairDiag,
                                    const uint8_t *isIdentity, float seqIdThr, int32_t alnLenThr, uint32_t *outIdx, sd_sw_result *out,
                                    uint32_t *nOut, char *btPool, uint64_t btCap, uint64_t *btUsed);

/* ---- Matcher::compressAlignment on the device (M/src/alignment/Matcher.cpp:166-185) -----------------------------------------------
 * on != 0: the alignment calls of this context (sd_sw_align_batch, _compact, _compact_diag, _best_by_group) fill the pool with the
 * run-length text of every returned backtrace ("57M2I103M"; "0M" first when a backtrace does not begin with a match, as the
 * reference's state machine prints it) instead of its letters: btOffset is the start of the text, bits 8.. of `flags` its length
 * (bits 0..7 keep their meaning), btLen stays the alignment length.  A protein alignment's text is about a tenth of its letters:
 * what crosses PCIe per 12 000-query step at 1 000 proteomes falls from 780 MB to under 100 MB, and the host's compression pass
 * (sd_host_compress_backtrace per record) is not run.  The pool must hold 2 bytes per backtrace letter in the worst case
 * (SD_ENOMEM otherwise, as before).  sd_sw_download_bytes: bytes of records (+ indices) and of pool the context's alignment calls
 * have copied to the host since it was created. */
int sd_sw_set_cigar_pool(sd_ctx *ctx, int on);
int sd_sw_download_bytes(sd_ctx *ctx, uint64_t *recordBytes, uint64_t *poolBytes);

/* ---- self-test entry points of the library's own device primitives (csrc/hip/sd_scan_sort.h), for tests/ ----------------------
 * sd_selftest_sort_pairs: stable sort of (key, value) pairs by the key bits [beginBit, endBit) -- what the alignment task order and
 * the prefilter's fall-back paths use.  sd_selftest_scan: exclusive sums of n 32-bit values into 64-bit offsets (n + 1 outputs: the
 * total last) and, with runningMax != NULL, the inclusive running maximum of the same values. */
int sd_selftest_sort_pairs(sd_ctx *ctx, const uint32_t *keys, const uint32_t *vals, uint32_t n, int beginBit, int endBit, uint32_t *outKeys,
                           uint32_t *outVals);
int sd_selftest_scan(sd_ctx *ctx, const uint32_t *in, uint32_t n, uint64_t *exclusiveSum, uint32_t *runningMax);

int sd_sw_align_batch_hostpath(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                               uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, const uint8_t *isIdentity,
                               sd_sw_result *out, char *btPool, uint64_t btCap, uint64_t *btUsed);

/* The score pass alone (sw_sse2_byte / sw_sse2_word semantics, StripedSmithWaterman.cpp:639-1214):
 * lanes = 32 reproduces the AVX2 byte kernel's lane structure, 16 the word kernel's.  reverse != 0 runs the
 * start-position pass on query[0..qEnd[i]] reversed x target[0..tEnd[i]] scanned downwards.
 * out: 3 int32 per pair {max score, end_db (scan position mapped back to a target index, -1 if none), end_query}. */
int sd_sw_score_batch(sd_ctx *ctx, const sd_sw_params *par, const sd_seqset *queries, const sd_seqset *targets,
                      uint32_t nPairs, const uint32_t *pairQ, const uint32_t *pairT, int lanes, int reverse,
                      const int32_t *qEnd, const int32_t *tEnd, int32_t *out);

/* number of DP cells the last sd_sw_* call evaluated on the device, by pass */
int sd_sw_last_cells(sd_ctx *ctx, uint64_t *forwardCells, uint64_t *reverseCells, uint64_t *tracebackCells);

/* ---- prefilter ------------------------------------------------------------------------ */
typedef struct {
    int32_t kmerSize;        /* 6 or 7 */
    int32_t kmerThr;         /* Prefiltering::getKmerThreshold (Prefiltering.cpp:1005) */
    int32_t maxHitsPerQuery; /* --max-seqs */
    int32_t minDiagScore;    /* --min-ungapped-score, must be >= 1 */
    uint32_t binSize;        /* BINSIZE of CacheFriendlyOperations picked from dbSize vs host L2 (QueryMatcher.cpp:422-450) */
    int32_t covMode;         /* coverage pre-filter of Prefiltering.cpp:856-863; covThr <= 0 disables */
    float covThr;
    int8_t ungappedMatrix[21 * 21]; /* blosum62 at 2 bit-factor, scoreBias -0.2 (Prefiltering.cpp:69,991) */
} sd_prefilter_params;

typedef struct {
    uint32_t seqId;   /* hit_t (QueryMatcher.h:33-37) */
    int32_t score;
    uint16_t diagonal;
    uint16_t pad;
} sd_hit;

/* Target side: replaces Prefiltering::getIndexTable (Prefiltering.cpp:514) -- the k-mer index, the masked
 * sequence lookup and the sorted 2-mer/3-mer similarity tables are uploaded once and stay resident.
 *   kmerOffsets[20^k+1] (u32), entrySeq/entryPos: lists sorted by (seqId,pos) (IndexTable.h:182-189)
 *   maskedResidues/seqOffsets: SequenceLookup (M/src/prefiltering/SequenceLookup.h)
 *   ext2/ext3: ExtendedSubstitutionMatrix::calcScoreMatrix rows (score int16, index u16), 400x400 and 8000x8000
 * kmerOffsets, kmerBlockBase, entrySeq, entryPos and maskedResidues may be device pointers (an index that arrived by a
 * device-to-device broadcast is made resident without a host round trip); seqOffsets is read on the host. */
int sd_target_create(sd_ctx *ctx, int kmerSize, const uint32_t *kmerOffsets, const uint32_t *entrySeq,
                     const uint16_t *entryPos, uint64_t nEntries, const uint8_t *maskedResidues,
                     const uint64_t *seqOffsets, uint32_t nSeq, const int16_t *ext2Score, const uint16_t *ext2Index,
                     const int16_t *ext3Score, const uint16_t *ext3Index, sd_target **out);
/* the same for a wide index (sd_host_index_block_base); kmerBlockBase = NULL is sd_target_create */
int sd_target_create_wide(sd_ctx *ctx, int kmerSize, const uint32_t *kmerOffsets, const uint64_t *kmerBlockBase,
                          const uint32_t *entrySeq, const uint16_t *entryPos, uint64_t nEntries, const uint8_t *maskedResidues,
                          const uint64_t *seqOffsets, uint32_t nSeq, const int16_t *ext2Score, const uint16_t *ext2Index,
                          const int16_t *ext3Score, const uint16_t *ext3Index, sd_target **out);
/* The same target built ON THE DEVICE from the unmasked residues: tantan masking (M/src/commons/Masker.cpp:15-55,
 * M/lib/tantan/tantan.cpp:308-460), k-mer collection with the self-score threshold, one entry per (k-mer, target) at the
 * smallest position, lists in (target, position) order (IndexBuilder::fillDatabase, M/src/prefiltering/IndexBuilder.cpp:55-239,
 * IndexTable.h:131-189,376-392) -- replaces sd_host_build_index + sd_target_create_wide (minutes of host time for 10^4
 * proteomes).  residues: host or device pointer.  maskRatios / selfScore: sd_host_index_tables.  The result equals the
 * host-built index entry for entry (tests/test_gpu_index_build.py); the wide form is chosen from the entry count.
 * stats (nullable): [0] entries, [1] masked residues, [2] k-mer range passes, [3] k-mer records before the per-target dedupe. */
int sd_target_build(sd_ctx *ctx, int kmerSize, int kmerThr, int mask, double maskProb, const uint8_t *residues,
                    const uint64_t *seqOffsets, uint32_t nSeq, const double *maskRatios, const int8_t *selfScore,
                    const int16_t *ext2Score, const uint16_t *ext2Index, const int16_t *ext3Score, const uint16_t *ext3Index,
                    sd_target **out, uint64_t *stats);
/* what a target holds on the device (any pointer may be NULL): masked residues, absolute list starts (tableSize + 1),
 * entries as (sequence id, position) */
/* sampled check of an index too large to download: n (k-mer, sequence, position) triples = all entries of the nSample sequences
 * `sample` (ascending) as computed elsewhere; *missing = triples that are not entries of the index, *inSample = entries of the
 * index that belong to the sample sequences (== n exactly when it holds nothing else for them); masked residues
 * [resBegin, resEnd) to maskedOut (nullable) */
int sd_target_sample_check(sd_ctx *ctx, const sd_target *t, const uint32_t *kmer, const uint32_t *seq, const uint32_t *pos, uint64_t n,
                           const uint32_t *sample, uint32_t nSample, uint64_t *missing, uint64_t *inSample, uint64_t resBegin,
                           uint64_t resEnd, uint8_t *maskedOut);
int sd_target_download(sd_ctx *ctx, const sd_target *t, uint64_t *nEntries, uint64_t *tableSize, uint8_t *masked, uint64_t *starts,
                       uint32_t *entrySeq, uint16_t *entryPos);
void sd_target_destroy(sd_target *t);

/* Replaces the per-query loop body of Prefiltering::runSplit (Prefiltering.cpp:817-886), i.e.
 * QueryMatcher::matchQuery (QueryMatcher.cpp:85) + the coverage pre-filter, for nQ queries at once.
 *   qKmerBias: per window start i the rounded composition bias of QueryMatcher.cpp:230-240 (int16, same
 *              offsets as the residues; entries past L-span are ignored)
 *   qDiagBias: per residue int8 of UngappedAlignment.cpp:392-396
 *   identityId[q]: index of the query in the target DB or UINT32_MAX
 * outHits: nQ * maxHitsPerQuery slots (row q at q*maxHitsPerQuery), outCount[nQ].
 * outCount[q] == UINT32_MAX marks a query that was NOT computed: it needs a reference route the device lacks (a second
 * overflow of the hit buffer beyond 32 parts, QueryMatcher.cpp:289-303, or >= 2^32 index hits; result lists have no length
 * limit); its row is empty, the other queries of the
 * call are computed normally and sd_last_error() names the count.  Callers must treat such rows as errors, not as "no hits".
 * stats (nullable, 4*nQ u64): #similar k-mers, #index entries, #diagonals scored, sum of diagonal lengths. */
int sd_prefilter_batch(sd_ctx *ctx, const sd_target *target, const sd_prefilter_params *par, uint32_t nQ,
                       const uint8_t *qResidues, const uint64_t *qOffsets, const int16_t *qKmerBias,
                       const int8_t *qDiagBias, const uint32_t *identityId, sd_hit *outHits, uint32_t *outCount,
                       uint64_t *stats);

/* ---- Profile queries (`--num-iterations` > 1 feeds profile DBs to prefilter and align) ----
 * sd_host_map_profiles restates Sequence::mapProfile (M/src/commons/Sequence.cpp:241-292) for n profile DB entries
 * (25 bytes per position, Sequence.h:458-471; byteOffsets[n+1] into profileData): per position the query letter,
 * the consensus letter (nullable output), the int8 alignment profile [21] = score / 4 with the X column 0, and the
 * k-mer generator's row: the 20 scores sorted descending by Util::rankedDescSort20's network and the amino acids
 * in that order.  posOffsets[n+1] receives the position offsets: byteOffsets / 25 (when byteOffsets[0] = 0), except that a profile
 * longer than --max-seq-len (65 535 positions) is cut there as mapProfile does (:247-266) and the later ones move up. */
int sd_host_map_profiles(const char *profileData, const uint64_t *byteOffsets, uint32_t n, uint8_t *queryLetters,
                         uint8_t *consensus, int8_t *alnProfile /* total x 21 */, int16_t *sortedScore /* total x 20 */,
                         uint8_t *sortedIndex /* total x 20 */, uint64_t *posOffsets);
/* Prefiltering::getKmerThreshold for profile searches without context pseudo counts (Prefiltering.cpp:1031-1043) */
int sd_host_profile_kmer_threshold(float sensitivity, int kmerSize);
/* sd_prefilter_batch for profile queries: QueryMatcher::matchQuery with a DBTYPE_HMM_PROFILE Sequence --
 * k-mers from the per-position rows (Sequence::nextProfileKmer :294-305, KmerGenerator::setDivideStrategy(ScoreMatrix**)
 * KmerGenerator.cpp:30-39), windows whose query letter is X skipped, no composition bias (QueryMatcher.cpp:93-99),
 * diagonal scores from the alignment profile (UngappedAlignment::createProfile's profile branch, :398-405).
 * The target index must have been built with k-mer threshold 0 (Prefiltering.cpp:525-527). */
int sd_prefilter_profile_batch(sd_ctx *ctx, const sd_target *target, const sd_prefilter_params *par, uint32_t nQ,
                               const uint8_t *queryLetters, const uint64_t *qOffsets, const int16_t *sortedScore,
                               const uint8_t *sortedIndex, const int8_t *alnProfile,
                               const uint32_t *identityId /* as in sd_prefilter_batch; NULL = none */, sd_hit *outHits,
                               uint32_t *outCount, uint64_t *stats);

/* ---- clusterhits ---------------------------------------------------------------------- */
typedef struct {
    uint32_t maxGeneGap;   /* d, --max-gene-gap (3) */
    uint32_t clusterSize;  /* cls, --cluster-size (2) */
    double alpha;          /* 1 */
    float pCluThr, pMHThr; /* 0.01 */
} sd_ch_params;

/* Replaces the body of the omp-for in clusterhits (R/src/util/ClusterHits.cpp:295-492) for nPairs
 * (query set, target set) entries.  Hits of entry p are [hitOff[p], hitOff[p+1]).
 *   strands: bit0 query strand, bit1 target strand (ClusterHits.cpp:346-347)
 *   lGamma[lGammaLen]: logGamma(i) table (ClusterHits.cpp:267-271), generated by sd_host_lgamma_table
 * Outputs per hit: clusterOfHit (ordinal of the emitted cluster inside its entry, UINT32_MAX = none) and
 * rankInCluster (emission order inside the cluster); per entry nClusters; per emitted cluster (slot
 * hitOff[p]+ordinal) pCO, pMH, size. */
int sd_clusterhits_batch(sd_ctx *ctx, const sd_ch_params *par, uint32_t nPairs, const uint64_t *hitOff,
                         const uint32_t *qPos, const uint32_t *tPos, const uint8_t *strands, const double *pval,
                         const uint32_t *Nq, const double *lGamma, uint32_t lGammaLen, uint32_t *clusterOfHit,
                         uint32_t *rankInCluster, uint32_t *nClusters, double *pCO, double *pMH,
                         uint32_t *clusterSizeOut);

/* ---- host-side stages that stay on the CPU ------------------------------ */
typedef struct sd_host sd_host;
int sd_host_create(int threads, sd_host **out);
void sd_host_destroy(sd_host *h);
/* which: 0 blosum62@2 (SW), 1 VTML80@8 bias -0.2 (seeds), 2 blosum62@2 bias -0.2 (diagonal scoring) */
int sd_host_matrix(sd_host *h, int which, int8_t *out21x21, double *pBack21, uint8_t *aa2num256);
/* the matrix file text (.out layout: background, lambda, half-bit scores) of blosum62 (0) / VTML80 (1): what an index file's
 * SCOREMATRIXNAME entry and the reference's "NAME.out:DATA" matrix argument carry (BaseMatrix.cpp:176-214); returns the length */
int sd_host_matrix_text(int which, char *buf, size_t cap);
int sd_host_map_sequence(sd_host *h, const char *ascii, uint64_t len, uint8_t *out);
/* SubstitutionMatrix::calcLocalAaBiasCorrection + the three integer roundings, for n sequences */
int sd_host_comp_bias(sd_host *h, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int kmerSize,
                      int8_t *swBias, int8_t *diagBias, int16_t *kmerBias);
/* the Smith-Waterman composition bias alone, against matrix `which` (sd_host_matrix numbering): the --realign pass builds its
 * query profile from the score-biased matrix, and calcLocalAaBiasCorrection reads the matrix it is given
 * (Alignment.cpp:296-303 -> Matcher::initQuery -> StripedSmithWaterman.cpp:1230-1235) */
int sd_host_sw_comp_bias(sd_host *h, int which, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int8_t *swBias);
/* The three integer composition-bias arrays of sd_host_comp_bias, computed on the device (same values, bit for bit:
 * the kernel forms calcLocalAaBiasCorrection's integer window sums, SubstitutionMatrix.cpp:79-109, and reads the float
 * tail from a table the host evaluated with the reference's expression order).  For callers short of host cores. */
int sd_comp_bias_batch(sd_ctx *ctx, sd_host *h, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int kmerSize,
                       int8_t *swBias, int8_t *diagBias, int16_t *kmerBias);

/* IndexBuilder::fillDatabase (mask + count + fill), returns an index handle to read back */
typedef struct sd_host_index sd_host_index;
/* the two host tables sd_target_build takes: tantan's likelihood ratios of the seed matrix (BaseMatrix.h:85-96) and the
 * k-mer self scores (IndexBuilder.cpp:10-21) */
int sd_host_index_tables(sd_host *h, double *maskRatios /* 21 x 21 */, int8_t *selfScore /* 21 */);
int sd_host_index_build(sd_host *h, const uint8_t *residues, const uint64_t *offsets, uint32_t n, int kmerSize,
                        int kmerThr, int mask, double maskProb, sd_host_index **out);
int sd_host_index_info(sd_host_index *ix, uint64_t *tableSize, uint64_t *nEntries, uint64_t *maskedResidues);
int sd_host_index_arrays(sd_host_index *ix, const uint32_t **kmerOffsets, const uint32_t **entrySeq,
                         const uint16_t **entryPos, const uint8_t **maskedResidues);
/* Indexes of 2^32 entries and more (targets beyond ~4.4e9 residues; the reference's offsets are size_t,
 * IndexTable.h:486) are WIDE: kmerOffsets[i] is then relative to blockBase[i >> 16] (one 64-bit base per 65 536 k-mers,
 * nBlocks of them), list i starts at blockBase[i >> 16] + kmerOffsets[i].  *blockBase = NULL for an ordinary index. */
int sd_host_index_block_base(sd_host_index *ix, const uint64_t **blockBase, uint64_t *nBlocks);
void sd_host_index_destroy(sd_host_index *ix);
int sd_host_ext_matrix(sd_host *h, int wordLen, const int16_t **score, const uint16_t **index, uint32_t *size);
int sd_host_kmer_threshold(float sensitivity, int kmerSize);
/* IndexTable::computeKmerSize (M/src/prefiltering/IndexTable.h:439-449): 6 below 3.35e9 target residues, else 7 */
int sd_host_auto_kmer_size(uint64_t targetResidues);
unsigned sd_host_bin_size(uint64_t dbSize, uint64_t l2CacheSize); /* l2CacheSize 0 = sysconf of this host */
/* the (query, target) pair list Alignment::run walks (Alignment.cpp:346-379), from sd_prefilter_batch's row-per-query
 * output; returns the number of pairs (pairQ / pairT NULL: count only) */
uint64_t sd_host_pair_list(const sd_hit *hits, const uint32_t *counts, uint32_t nQ, uint32_t rowWidth, uint32_t *pairQ,
                           uint32_t *pairT);
int sd_host_lgamma_table(double *out, uint32_t n);
/* The two P-values of one cluster of hits (host; what sd_clusterhits_batch evaluates for every cluster it emits,
 * R/src/util/ClusterHits.cpp:80-134,184-213): gene positions on both sides, strands (bit 0 query, bit 1 target), per-hit
 * P-values, the size of the query set.  order[nHits] (nullable): the members by query position, the order they are printed in. */
int sd_host_cluster_pvalues(uint32_t nHits, const uint32_t *qPos, const uint32_t *tPos, const uint8_t *strands, const double *pval,
                            uint32_t querySetSize, double alpha, const double *lGamma, uint32_t lGammaLen, double *pCluster,
                            double *pMultihit, uint32_t *order);
double sd_host_evalue(uint64_t dbResidues, double score, double qLen);
double sd_host_bitscore(double score);
/* A double after one hand-off between two reference modules: printed as "%.3E" (Matcher.cpp:288, besthitbyset.cpp:129,
 * combinehits.cpp:218-221) and parsed by strtod.  text: >= 16 bytes, NUL-terminated; *back: the parsed value. */
int sd_host_quantise_3e(double v, char *text, double *back);
/* Matcher::compressAlignment (M/src/alignment/Matcher.cpp:166-185): the run-length form of a backtrace of M / I / D letters (an empty
 * one is "0M").  out == NULL: *len = length needed; SD_ENOMEM when cap is smaller. */
int sd_host_compress_backtrace(const char *bt, uint64_t n, char *out, uint64_t cap, uint64_t *len);
/* Util::canBeCovered (M/src/commons/Util.cpp:477-494): the length pre-check of Prefiltering.cpp:856-863 / Alignment.cpp:370 */
int sd_host_can_be_covered(float covThr, int covMode, float queryLength, float targetLength);

/* ---- in-process aggregation between align and clusterhits (host) ---------
 * Replaces prefixid -> besthitbyset -> mergeresultsbyset -> combinehits (R/data/clustersearch.sh:121-140) and
 * summarizeresults (:151) without the text DB round trips, keeping their %.3E re-quantisation. */
typedef struct sd_agg sd_agg;
/* qSetOf/qLen: genome set and length of every query protein (DB key order); likewise for the targets */
int sd_agg_create(const uint32_t *qSetOf, const int32_t *qLen, uint32_t nQ, const uint32_t *tSetOf, const int32_t *tLen,
                  uint32_t nT, uint32_t nQSets, uint32_t nTSets, double evalThr, int covMode, float covThr, int alnLenThr,
                  int filterSelfMatch, sd_agg **out);
void sd_agg_destroy(sd_agg *a);
/* alignment results of a batch of pairs (pairs of one query contiguous; query key = qBase + pairQ[i]):
 * checkCriteria, compareHits order, best hit per (query protein, target set), log P threshold of combinehits */
int sd_agg_add(sd_agg *a, uint32_t nPairs, uint32_t qBase, const uint32_t *pairQ, const uint32_t *pairT,
               const sd_sw_result *res, const uint8_t *isIdentity, const char *btPool);
/* optional: DB keys of the query / target proteins (default: key = index) -- the order of the hits inside an entry
 * (mergeresultsbyset walks the set's members by key) and Matcher::compareHits' last criterion use keys */
int sd_agg_set_keys(sd_agg *a, const uint32_t *qKeys, const uint32_t *tKeys);
/* on != 0: a query's records arrive in the line order of its alignment DB entry and that order is not Matcher::compareHits order -- the
 * merged result of `search --num-iterations` (mergedbs concatenates the iterations' lists, M/data/workflow/blastpgp.sh:106-117).
 * besthitbyset keeps, per target set, the first line whose %.3E text E-value is strictly smaller than the one it holds
 * (R/src/util/besthitbyset.cpp:88-101): among equal E-values the line of the earlier iteration.  Default (0): the compareHits minimum,
 * which is the same line whenever the list is one sorted list (and does not depend on the order the records arrive in). */
int sd_agg_set_list_order(sd_agg *a, int on);
/* cigarText != 0: the pool handed to sd_agg_add holds, per record, the run-length text of its backtrace (Matcher::compressAlignment's
 * output, M/src/alignment/Matcher.cpp:166-185) as the alignment calls return it after sd_sw_set_cigar_pool(ctx, 1) -- btOffset its
 * start, flags >> 8 its length -- instead of the backtrace letters; the aggregation copies it where it used to compress.  Default 0. */
int sd_agg_set_pool_form(sd_agg *a, int cigarText);
int sd_agg_finish(sd_agg *a, uint64_t *nEntries, uint64_t *nHits);
int sd_agg_stats(sd_agg *a, uint64_t *nAligned, uint64_t *nAccepted);
int sd_agg_get(sd_agg *a, uint64_t *entryOff, uint32_t *entryQSet, uint32_t *entryTSet, uint32_t *hitQ, uint32_t *hitT,
               double *pval);
int sd_agg_write_tsv(sd_agg *a, const char *path, const uint32_t *clusterOfHit, const uint32_t *rankInCluster,
                     const uint32_t *nClusters, const double *pCO, const double *pMH, const uint32_t *clusterSize,
                     const char *qNames, const uint64_t *qNameOff, const char *tNames, const uint64_t *tNameOff,
                     const char *qSources, const uint64_t *qSourceOff, const char *tSources, const uint64_t *tSourceOff,
                     int canonical, uint64_t *nClusterLines, uint64_t *nHitLines);
/* the same, appending to `path` when append != 0 and numbering the clusters from firstClusterKey (several results, one file) */
int sd_agg_write_tsv_from(sd_agg *a, const char *path, int append, uint64_t firstClusterKey, const uint32_t *clusterOfHit,
                          const uint32_t *rankInCluster, const uint32_t *nClusters, const double *pCO, const double *pMH,
                          const uint32_t *clusterSize, const char *qNames, const uint64_t *qNameOff, const char *tNames,
                          const uint64_t *tNameOff, const char *qSources, const uint64_t *qSourceOff, const char *tSources,
                          const uint64_t *tSourceOff, int canonical, uint64_t *nClusterLines, uint64_t *nHitLines);
/* The clusters of a finished aggregation as one self-contained byte string ("cluster records"): what a rank hands to
 * sd_gather_results and what the result TSV is written from (R/src/util/SummarizeResults.cpp:77-112 reads the same fields
 * from the cluster DB).  Little endian; per cluster: u32 members, u32 qSet, u32 tSet, u32 0, f64 pCO, f64 pMH; then per
 * member in cluster order: u32 q, u32 t, char pval[16], char seqId[8], char eval[16], i32 qStart, qEnd, qLen, tStart, tEnd,
 * tLen, u32 cigarLen, the cigar bytes padded to a multiple of 4.  out == NULL: *bytes = size needed. */
int sd_agg_records(sd_agg *a, const uint32_t *clusterOfHit, const uint32_t *rankInCluster, const uint32_t *nClusters,
                   const double *pCO, const double *pMH, const uint32_t *clusterSize, void *out, uint64_t cap, uint64_t *bytes);
/* Validates a record buffer that was gathered from other ranks before it is written: whole records only, every set / sequence
 * index inside [0, nQSets) / [0, nTSets) / [0, nQ) / [0, nT) -- SD_EINVAL otherwise; counts of clusters and members (nullable). */
int sd_records_check(const void *records, uint64_t bytes, uint32_t nQSets, uint32_t nTSets, uint32_t nQ, uint32_t nT, uint64_t *nClusters,
                     uint64_t *nMembers);
/* the TSV of cluster records -- of one result or of several ranks' records concatenated: clusters numbered from firstClusterKey */
int sd_records_write_tsv(const void *records, uint64_t bytes, const char *path, int append, uint64_t firstClusterKey,
                         const char *qNames, const uint64_t *qNameOff, const char *tNames, const uint64_t *tNameOff,
                         const char *qSources, const uint64_t *qSourceOff, const char *tSources, const uint64_t *tSourceOff,
                         int canonical, uint64_t *nClusterLines, uint64_t *nHitLines);


/* ---- the tail of Alignment::run for a batch: criteria, order, --realign, text (host) --------
 * What `align` does with the records of one query after the Smith-Waterman calls: Alignment::checkCriteria
 * (M/src/alignment/Alignment.cpp:389-399,548-567, with the derived fields of Matcher::getSWResult, Matcher.cpp:88-137),
 * the compareHits sort (Alignment.cpp:403-405, Matcher.h:157-168), the --realign second pass (:408-440) and
 * Matcher::resultToBuffer (Matcher.cpp:280-327) -- batched over the queries of a chunk. */
typedef struct {
    double evalThr;        /* -e */
    float seqIdThr;        /* --min-seq-id */
    int32_t alnLenThr;     /* --min-aln-len */
    int32_t covMode;       /* --cov-mode */
    float covThr;          /* -c (the realign pass's coverage when realign != 0, Alignment.cpp:49-51) */
    int32_t seqIdMode;     /* --seq-id-mode: 0 alignment length, 1 shorter, 2 longer sequence */
    int32_t swMode;        /* Matcher::SCORE_ONLY 0 / SCORE_COV 1 / SCORE_COV_SEQID 2 the records were computed with */
    int32_t addBacktrace;  /* -a */
    int32_t realign;       /* --realign: the first pass is SCORE_ONLY without a coverage threshold (Alignment.cpp:45-57) */
    int32_t realignSwMode; /* mode of the realignment records: max(alignment mode, SCORE_COV) (Alignment.cpp:46) */
    int32_t realignMaxSeqs;/* --realign-max-seqs */
    uint32_t maxAccept;    /* --max-accept */
    uint32_t maxRejected;  /* --max-rejected */
} sd_aln_criteria;

/* Records res[i] of local query resQ[i] (non-decreasing) against target id resT[i], in prefilter order.  order[] receives the
 * indices of the records that pass checkCriteria, grouped by query in compareHits order; countPerQuery[nQ] how many each
 * query has.  qLen[nQ], tLen / tKey by target id (tKey NULL: the id is the DB key). */
int sd_host_accept_sort(const sd_aln_criteria *crit, uint32_t nQ, uint32_t nRes, const uint32_t *resQ, const uint32_t *resT,
                        const sd_sw_result *res, const uint8_t *isIdentity, const int32_t *qLen, const int32_t *tLen,
                        const uint32_t *tKey, uint32_t *order, uint32_t *countPerQuery);
/* --realign (Alignment.cpp:408-440): second[s] is the realignment (score-biased matrix, E-value gate off) of the s-th accepted
 * first-pass record first[order[s]]; isIdentity is indexed like `second`.  merged[s] = second[s] carrying the first pass's
 * score and E-value; outOrder / outCount: the realigned records with coverage (or identity), at most realignMaxSeqs per
 * query, in compareHits order -- indices into merged. */
int sd_host_realign_select(const sd_aln_criteria *crit, uint32_t nQ, const uint32_t *countPerQuery, const uint32_t *order,
                           const uint32_t *resT, const sd_sw_result *first, const sd_sw_result *second,
                           const uint8_t *isIdentity, const int32_t *qLen, const int32_t *tLen, const uint32_t *tKey,
                           sd_sw_result *merged, uint32_t *outOrder, uint32_t *outCount);
typedef struct sd_alntext sd_alntext;
int sd_alntext_create(sd_alntext **out);
void sd_alntext_destroy(sd_alntext *t);
/* alignment DB entries of nQ queries: entry q = the lines of records rec[order[x]] (x in query q's slice, countPerQuery as
 * above), each `tKey bits seqId eval qStart qEnd qLen tStart tEnd tLen [cigar]` exactly as Matcher::resultToBuffer prints
 * them.  recT[i]: target id of record i; isIdentity indexed like rec. */
int sd_alntext_format(sd_alntext *t, const sd_aln_criteria *crit, uint32_t nQ, const uint32_t *countPerQuery,
                      const uint32_t *order, const uint32_t *recT, const sd_sw_result *rec, const uint8_t *isIdentity,
                      const char *btPool, const int32_t *qLen, const int32_t *tLen, const uint32_t *tKey);
/* text / entryOff[nQ+1] of the last sd_alntext_format; valid until the next call on t */
int sd_alntext_get(sd_alntext *t, const char **text, const uint64_t **entryOff);


/* ---- the whole search in one object: host driver of the streaming pipeline (C++; replaces what the reference runs as
 * `search` + prefixid/besthitbyset/mergeresultsbyset/combinehits + `clusterhits`, R/data/clustersearch.sh:110-146) ------
 * sd_search owns two device contexts on one GPU (prefilter | alignments on separate HIP streams), the resident target
 * (index + sequences) and the stage threads: composition bias of chunk i+2 | prefilter + pair list of chunk i+1 |
 * alignments of chunk i | aggregation of chunk i-1.  Query ranges are streamed back to back; every range gets its own
 * aggregation, clusterhits call and result.  Nothing here computes: every stage is one of the calls above. */
typedef struct {
    const uint8_t *residues;     /* numeric residues (profile queries: the query letters), concatenated */
    const uint64_t *offsets;     /* n + 1 */
    uint32_t n;
    const uint32_t *setId;       /* genome set of every protein (createsetdb's lookup column 3) */
    const uint32_t *posInSet;    /* gene index inside the set (lookup name, third field from the end) */
    const uint8_t *strand;       /* start < end */
    uint32_t nSets;
    const uint32_t *keys;        /* DB keys (NULL: key = index); ordering inside result entries and compareHits tie-break */
    /* profile queries (iterations >= 1 of --num-iterations): sd_host_map_profiles' arrays, NULL for sequence DBs */
    const int8_t *alnProfile;    /* total x 21 */
    const int16_t *sortedScore;  /* total x 20 */
    const uint8_t *sortedIndex;  /* total x 20 */
} sd_setdb;

typedef struct {
    float sensitivity;       /* -s (clustersearch: 5.7) */
    int32_t kmerSize;        /* -k, 0 = automatic (IndexTable.h:439-449) */
    int32_t maxSeqs;         /* --max-seqs */
    int32_t minDiagScore;    /* --min-ungapped-score (15) */
    uint32_t binSize;        /* 0 = from the DB size and this host's L2 (QueryMatcher.cpp:422-450) */
    int32_t mask;            /* --mask (1) */
    double maskProb;         /* --mask-prob (0.9) */
    int32_t compBiasCorr;    /* --comp-bias-corr (1) */
    double evalThr;          /* -e */
    int32_t covMode;         /* --cov-mode */
    float covThr;            /* -c */
    int32_t alnLenThr;       /* --min-aln-len */
    uint32_t maxGeneGap, clusterSize;   /* clusterhits */
    double alpha;
    float pCluThr, pMHThr;
    int32_t filterSelfMatch; /* --filter-self-match (reference default 0) */
    int32_t profileQueries;  /* the query side is a profile DB: own k-mer threshold table, index with threshold 0 */
    int32_t chunkQueries;    /* queries per device chunk at most; a range is cut into equal chunks (0 = by the target set: 10000, 2500 against 10^6 sequences and more) */
    int32_t deviceBias;      /* composition bias: 0 = host stage (OpenMP), anything else (1, -1 = default) = on the device */
    int32_t threads;         /* host threads of this rank (0 = all of the cgroup quota) */
    int32_t alignPriority;   /* stream priority of the alignment context (sd_ctx_create_prio) */
} sd_search_params;
void sd_search_default_params(sd_search_params *p);   /* the clustersearch workflow defaults (R/src/workflow/clustersearch.cpp:9-37) */

typedef struct sd_search sd_search;
typedef struct sd_search_result sd_search_result;
/* builds the index of `target` on the host (tantan masking + IndexBuilder::fillDatabase), uploads it and the target
 * sequences; the arrays behind `target` must stay valid until sd_search_destroy */
int sd_search_create(int device, const sd_search_params *par, const sd_setdb *target, sd_search **out);
/* the same with a target index that exists already (read from a createindex file, or built once and handed to every rank):
 * arrays as sd_host_index_arrays returns them; they may be released after the call */
typedef struct {
    int32_t kmerSize, kmerThr;
    const uint32_t *kmerOffsets;     /* 20^k + 1 */
    const uint32_t *entrySeq;
    const uint16_t *entryPos;
    uint64_t nEntries;
    const uint8_t *maskedResidues;   /* the masked target residues the diagonal scoring reads (SequenceLookup) */
    uint64_t nMaskedResidues;        /* how many residues tantan masked (statistics only) */
    const uint64_t *kmerBlockBase;   /* NULL, or the bases of a wide index (sd_host_index_block_base) */
} sd_index_view;
int sd_search_create_indexed(int device, const sd_search_params *par, const sd_setdb *target, const sd_index_view *index,
                             sd_search **out);
void sd_search_destroy(sd_search *s);
const char *sd_search_last_error(sd_search *s);
/* the device contexts of the pipeline, e.g. for sd_profile_*: 0 prefilter, 1 alignment lane 0, 2 composition bias (NULL when
 * the bias runs on the host), 3 alignment lane 1 (NULL with SD_ALIGN_LANES=1), 4 clusterhits, 5 prefilter lane 1 (NULL with SD_PF_LANES=1); NULL beyond */
sd_ctx *sd_search_ctx(sd_search *s, int which);
/* the resident target of the search (borrowed: lives as long as the search object; context = sd_search_ctx(s, 0)) */
const sd_target *sd_search_target(sd_search *s);
/* optional sinks, called from the pipeline's threads in chunk order: the prefilter rows of a chunk (after the coverage
 * pre-filter; what `prefilter` writes) and its reportable alignment records (what `align` writes after checkCriteria) */
typedef void (*sd_pref_sink)(void *user, uint32_t firstQuery, uint32_t nQ, const sd_hit *rows, const uint32_t *counts,
                             uint32_t rowWidth);
typedef void (*sd_aln_sink)(void *user, uint32_t firstQuery, uint32_t nQ, uint32_t nRes, const uint32_t *resQ /* chunk-local */,
                            const uint32_t *resT, const sd_sw_result *res, const uint8_t *isIdentity, const char *btPool);
int sd_search_set_sinks(sd_search *s, sd_pref_sink pref, sd_aln_sink aln, void *user);
/* queries per device chunk (at most; 0 = the library's choice) for the following sd_search_stream calls (results do not depend on it) */
int sd_search_set_chunk_queries(sd_search *s, int32_t chunkQueries);
/* on != 0: the following sd_search_stream calls build every range's cluster records (sd_search_result_records) when the range is
 * finished, inside the pipeline, instead of on demand afterwards -- what a rank of a multi-GPU run asks for, whose records all go into
 * the final gather (sd_gather_results) */
int sd_search_set_want_records(sd_search *s, int on);
/* With sd_search_set_want_records(s, 1): fn(user, range, records, bytes) is called on the stream's finalising thread as soon as range
 * `range` of the running sd_search_stream is finalised and its cluster records are built -- in range order, also for ranges without
 * hits (bytes = 0) -- so that a consumer (sd_gather_stream_sink: the round-by-round gather of a multi-GPU run) takes them while the
 * later ranges are still being searched.  The bytes are those sd_search_result_records returns; valid during the call.  NULL: off. */
typedef void (*sd_records_sink)(void *user, uint32_t range, const void *records, uint64_t bytes);
int sd_search_set_records_sink(sd_search *s, sd_records_sink fn, void *user);
/* query ranges [rangeBegin[i], rangeEnd[i]) of `query` (whole query sets each), streamed through one pipeline;
 * sameDb != 0: query protein i is target protein i (identity pairs, self hit first).  results[nRanges] receives one
 * handle per range (destroy each). */
int sd_search_stream(sd_search *s, const sd_setdb *query, int sameDb, uint32_t nRanges, const uint32_t *rangeBegin,
                     const uint32_t *rangeEnd, sd_search_result **results);
/* counts[8]: entries, matched hits, clusters, hits in clusters, pairs aligned, alignments accepted, prefilter hits, and the gate
 * "alignments accepted" refers to: 0 = the user's -e, 1 = combinehits' E-value bound (a stream without an alignment sink runs its
 * alignments with that bound as their gate -- sd_search_stats; entries, hits and clusters do not depend on it).  The fused path is
 * the clustersearch workflow's aggregation as the reference fixes it (clustersearch.sh:121-151: besthitbyset with
 * --simple-best-hit 1, combinehits with --aggregation-mode 0, no suboptimal hits); other aggregation settings are not a parameter
 * of sd_search and go through the glue modules (sd_agg_* / sdgpu besthitbyset ...) with an alignment sink, i.e. without the pushdown */
int sd_search_result_counts(sd_search_result *r, uint64_t *counts);
/* copies (any pointer may be NULL): entryOff[entries+1], entryQSet/entryTSet[entries], hitQ/hitT/pval[hits],
 * clusterOfHit/rankInCluster[hits], nClusters[entries], pCO/pMH/clusterSize[hits] (slot entryOff[e]+ordinal) */
int sd_search_result_arrays(sd_search_result *r, uint64_t *entryOff, uint32_t *entryQSet, uint32_t *entryTSet, uint32_t *hitQ,
                            uint32_t *hitT, double *pval, uint32_t *clusterOfHit, uint32_t *rankInCluster, uint32_t *nClusters,
                            double *pCO, double *pMH, uint32_t *clusterSize);
/* summarizeresults (R/src/util/SummarizeResults.cpp:77-112) of this result; names / sources as in sd_agg_write_tsv */
int sd_search_result_write_tsv(sd_search_result *r, const char *path, const char *qNames, const uint64_t *qNameOff,
                               const char *tNames, const uint64_t *tNameOff, const char *qSources, const uint64_t *qSourceOff,
                               const char *tSources, const uint64_t *tSourceOff, int canonical, int append,
                               uint64_t firstClusterKey, uint64_t *nClusterLines, uint64_t *nHitLines);
/* the result's cluster records (sd_agg_records of its aggregation and clusters); out == NULL: *bytes = size needed */
int sd_search_result_records(sd_search_result *r, void *out, uint64_t cap, uint64_t *bytes);
void sd_search_result_destroy(sd_search_result *r);
/* accumulated since create: stats[16] = similar k-mers, index hits, diagonals, diagonal length, prefilter hits, pairs,
 * forward / reverse / traceback cells, index entries, masked residues, k, k-mer threshold, bin size, queries not computed
 * (per-query error slots of sd_prefilter_batch; sd_search_last_error says why -- a caller must not take their empty rows for results),
 * 1 if the last stream ran its alignments with combinehits' E-value bound as their gate (a stream that writes nothing but cluster
 * records: pairs that cannot reach the cluster-hit result stop after the score pass, as pairs above -e do in the reference; the
 * `accepted` count of its results then refers to that gate; SD_EVAL_PUSHDOWN=0 switches it off);
 * seconds[16] = index build, upload, bias, prefilter, pair list, seqset, align, aggregate (waiting), aggregate (busy),
 * clusterhits, waiting for the prefilter, total of the last stream, 0... */
int sd_search_stats(sd_search *s, uint64_t *stats, double *seconds);
/* bytes the alignment lanes have copied to the host since create: records (+ pair indices), and backtrace pool -- letters, or
 * run-length text when the stream's only consumer of backtraces is the aggregation (sd_sw_set_cigar_pool; SD_CIGAR_ON_DEVICE=0: letters) */
int sd_search_download_bytes(sd_search *s, uint64_t *recordBytes, uint64_t *poolBytes);


/* ---- multi-GPU seam: query sets sharded over the ranks, one RCCL gather at the end --------
 * One process per GPU.  The path has no data-path collective: every rank searches its own query sets against its own
 * replica of the target.  The reference's counterpart is the MPI master merging per-rank result files
 * (M/src/prefiltering/Prefiltering.cpp:630-658). */
/* the ranks' share of the query sets: greedy by residue count, deterministic, mine[] ascending (must hold nSets entries) */
int sd_shard_query_sets(const uint64_t *setResidues, uint32_t nSets, uint32_t world, uint32_t rank, uint32_t *mine,
                        uint32_t *nMine);
typedef struct sd_comm sd_comm;
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to every rank (torch.distributed broadcast, a file, MPI ...) */
int sd_comm_unique_id(char *out128);
int sd_comm_init(int device, int nRanks, int rank, const char *uniqueId128, sd_comm **out);
void sd_comm_destroy(sd_comm *c);
/* A pinned host buffer owned by the communicator, with a device buffer of the same size behind it (both grow-only, freed by
 * sd_comm_destroy; a larger request replaces the buffer: take the pointer again).  which = 0: this rank's records, 1: the gathered
 * records on the root.  Records built in buffer 0 and gathered into buffer 1 cross the bus at its rate with no staging copy and no
 * allocation inside sd_gather_results; any other host memory still works (the runtime stages it). */
int sd_comm_host_buffer(sd_comm *c, int which, uint64_t bytes, void **ptr);
const char *sd_comm_last_error(sd_comm *c);
/* gatherv of byte records over RCCL: sizes[nRanks] receives every rank's byte count (on all ranks); on `root`, outOnRoot
 * (capacity outCap) receives the records concatenated in rank order and *outBytes their total.  The ranks agree on the
 * outcome of their local staging before any payload moves: when outCap is too small on the root, EVERY rank returns
 * SD_ENOMEM (*outBytes = the size needed; call again with room -- the size probe, and the ONLY meaning of SD_ENOMEM here), when a
 * rank cannot stage its records (device allocation, copy) EVERY rank returns SD_EHIP; no rank is left waiting in a send or
 * receive, and no rank repeats the collective alone. */
int sd_gather_results(sd_comm *c, const void *local, uint64_t nBytes, int root, uint64_t *sizes, void *outOnRoot, uint64_t outCap,
                      uint64_t *outBytes);
/* The same gather round by round behind a running search, instead of one blob per rank behind the last kernel (the reference's
 * merge starts when the last rank's files exist, M/src/prefiltering/Prefiltering.cpp:630-658).  The nRanges ranges of this rank's
 * sd_search_stream are grouped into nRounds rounds (roundOfRange[i], non-decreasing; e.g. the step of a range; ranks may hold
 * different numbers of ranges, also none, in a round -- nRounds must be the same on every rank).  sd_gather_stream_sink has the
 * sd_records_sink signature -- sd_search_set_records_sink(s, sd_gather_stream_sink, g) --: it appends a range's records to the communicator's
 * pinned send buffer (sd_comm_host_buffer(c, 0, ...) sizes it; records beyond it are staged in pageable memory), and when a round's
 * last range has arrived a worker thread runs that round's sd_gather_results on the communicator's stream while the search goes on.
 * On the root the rounds land back to back in outOnRoot (round 0: ranks 0 .. N-1, round 1: ...).  sd_gather_stream_end waits for the
 * last round (ranges that never arrived count as empty, so the ranks' collectives still match), fills roundOffsets[nRounds + 1] (the
 * root's byte offset of every round), sizes[nRounds * nRanks] and *totalOnRoot (each nullable), frees the object and returns the
 * first failing round's code (SD_ENOMEM: outCap too small -- on every rank, in the same round, as in sd_gather_results).  A closing
 * round tells the root whether every rank delivered all the ranges it announced: a rank whose search failed closes its stream early,
 * its missing rounds go out empty so that no rank waits, and the root's _end / _wait returns SD_EMISMATCH (the records it holds are
 * not everything). */
typedef struct sd_gather_stream sd_gather_stream;
int sd_gather_stream_begin(sd_comm *c, int root, uint32_t nRanges, const uint32_t *roundOfRange, uint32_t nRounds, void *outOnRoot,
                           uint64_t outCap, int ownBuffer, sd_gather_stream **out);
void sd_gather_stream_sink(void *gatherStream, uint32_t range, const void *records, uint64_t bytes);
int sd_gather_stream_end(sd_gather_stream *g, uint64_t *roundOffsets, uint64_t *sizes, uint64_t *totalOnRoot);
/* ownBuffer != 0 in a _begin call -- the same value on EVERY rank: it decides the rounds' protocol -- : outOnRoot / outCap are ignored,
 * the root's buffer belongs to the stream and grows round by round (over RCCL the ranks repeat a round whose records did not fit --
 * SD_ENOMEM is agreed on before any payload moves; over TCP the byte counts are gathered first).
 * sd_gather_stream_wait = _end without freeing the object: *dataOnRoot (nullable) is the root's buffer, valid until
 * sd_gather_stream_destroy.  sd_gather_stream_begin_tcp: the same stream over the TCP rendezvous (root 0; ranks that share a device,
 * which RCCL refuses -- one-GPU test rigs; `sdgpu clustersearch` picks the transport the way its final gather did). */
int sd_gather_stream_wait(sd_gather_stream *g, uint64_t *roundOffsets, uint64_t *sizes, uint64_t *totalOnRoot, const void **dataOnRoot);
void sd_gather_stream_destroy(sd_gather_stream *g);
/* Host-side rendezvous of the ranks over TCP (addr / port as a one-process-per-GPU launcher's MASTER_ADDR / MASTER_PORT; every
 * rank connects to rank 0 once, the calls are matched in program order): sd_tcp_bcast hands rank 0's buffer (the 128-byte
 * unique id) to every rank; sd_tcp_gather is the gatherv of byte records to rank 0 for ranks that share a device -- RCCL
 * refuses two ranks on one GPU, so one-GPU test rigs move the records this way.  The reference merges its MPI ranks' files
 * (M/src/prefiltering/Prefiltering.cpp:630-658). */
typedef struct sd_tcp sd_tcp;
int sd_tcp_connect(const char *addr, int port, int nRanks, int rank, sd_tcp **out);
void sd_tcp_close(sd_tcp *t);
int sd_tcp_bcast(sd_tcp *t, void *buf, uint64_t bytes);
/* sizes[nRanks] and the concatenated records on rank 0; outCap too small: SD_ENOMEM there with *outBytes = the size needed
 * (the records are consumed either way: gather the byte counts first) */
int sd_tcp_gather(sd_tcp *t, const void *local, uint64_t nBytes, uint64_t *sizes, void *outOnRoot, uint64_t outCap, uint64_t *outBytes);
int sd_gather_stream_begin_tcp(sd_tcp *t, int nRanks, int rank, uint32_t nRanges, const uint32_t *roundOfRange, uint32_t nRounds, void *outOnRoot,
                               uint64_t outCap, int ownBuffer, sd_gather_stream **out);


/* ---- result2profile: alignment DB -> profile DB between the iterations of `search --num-iterations` (SURVEY.md 8(f).3;
 * host, float).  M/src/util/result2profile.cpp:239-282: MSA from the backtraces (MultipleAlignment::computeMSA, no query
 * gaps), MsaFilter::filter, PSSMCalculator::computePSSMFromMSA, calcGlobalAaBiasCorrection, Masker::maskPssm,
 * Profile::toBuffer.  The records are what sd_host_map_profiles reads (25 bytes per position). */
typedef struct {
    int32_t filterMsa;        /* --filter-msa (1) */
    int32_t filterMinEnable;  /* --filter-min-enable (0) */
    float filterMaxSeqId;     /* --max-seq-id (0.9) */
    const char *qid;          /* --qid ("0.0"; comma separated list) */
    float qsc;                /* --qsc (-20) */
    float covMSAThr;          /* --cov (0) */
    int32_t Ndiff;            /* --diff (1000) */
    int32_t pcMode;           /* --pseudo-cnt-mode: 0 substitution score (the context specific library is not built in) */
    float pca, pcb;           /* --pca / --pcb substitution values (1.1 / 4.1) */
    int32_t wg;               /* --wg (0): global instead of position specific sequence weights */
    int32_t compBiasCorr;     /* --comp-bias-corr (1) */
    int32_t maskProfile;      /* --mask-profile (1) */
    double maskProb;          /* --mask-prob (0.9) */
} sd_r2p_params;
typedef struct sd_r2p sd_r2p;
int sd_r2p_create(sd_r2p **out);
void sd_r2p_destroy(sd_r2p *r);
/* nQ centre sequences (numeric residues; profile centres: their query letters) with, per centre, the alignments that
 * enter the profile (the caller has applied the E-value cut and dropped the self hit, result2profile.cpp:184-203):
 * edges [edgeOff[q], edgeOff[q+1]) -> target id, qStart, tStart and the expanded backtrace btPool[btOff[e] .. btOff[e+1]).
 * outProfiles: 25 bytes per centre position at qOff[q] * 25; outConsensus (nullable): the consensus residues. */
int sd_r2p_batch(sd_r2p *r, const sd_r2p_params *par, uint32_t nQ, const uint8_t *qLetters, const uint64_t *qOff,
                 const uint64_t *edgeOff, const uint32_t *edgeT, const int32_t *edgeQStart, const int32_t *edgeTStart,
                 const char *btPool, const uint64_t *btOff, const uint8_t *tResidues, const uint64_t *tOff, char *outProfiles,
                 uint8_t *outConsensus);
/* The same with the position-specific sequence weights -- PSSMCalculator::computeSequenceWeights' sub-alignment per column
 * (M/src/alignment/PSSMCalculator.cpp:394-588), O(columns^2 x rows) and the bulk of the step -- on the GPU (csrc/hip/sd_r2p.hip:
 * a workgroup per centre, every summation in the reference's order, the CPU-specific approximate reciprocal tabulated by the
 * host); alignment assembly, the greedy diversity filter, pseudo counts, scores and masking stay host stages.  Same bytes as
 * sd_r2p_batch.  (--wg: global weights, computed on the host as before.) */
int sd_r2p_batch_device(sd_ctx *ctx, sd_r2p *r, const sd_r2p_params *par, uint32_t nQ, const uint8_t *qLetters, const uint64_t *qOff,
                        const uint64_t *edgeOff, const uint32_t *edgeT, const int32_t *edgeQStart, const int32_t *edgeTStart,
                        const char *btPool, const uint64_t *btOff, const uint8_t *tResidues, const uint64_t *tOff, char *outProfiles,
                        uint8_t *outConsensus);

#ifdef __cplusplus
}
#endif
#endif
