// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" driver around the *real* reference classes, compiled by
// oracle/Makefile straight from the sources under /root/reference (no cmake, no
// generated headers, no stand-ins) into oracle/_ref/libsdref.so.  It exposes the
// reference's own
//   * SubstitutionMatrix / calcLocalAaBiasCorrection   (M/src/commons/SubstitutionMatrix.cpp:12,79)
//   * Masker + tantan                                   (M/src/commons/Masker.cpp:15, M/lib/tantan/tantan.cpp)
//   * IndexTable / SequenceLookup build                 (M/src/prefiltering/IndexTable.h:131,342; the
//     loop below follows IndexBuilder::fillDatabase, M/src/prefiltering/IndexBuilder.cpp:55-239, which itself
//     cannot be linked because it needs DBReader -> Parameters.cpp -> cmake-generated headers)
//   * ExtendedSubstitutionMatrix + KmerGenerator        (M/src/prefiltering/KmerGenerator.cpp:107)
//   * QueryMatcher::matchQuery                          (M/src/prefiltering/QueryMatcher.cpp:85)
//   * SmithWaterman::ssw_init / ssw_align               (M/src/alignment/StripedSmithWaterman.cpp:1216,238)
//   * EvalueComputation (ALP)                           (M/src/alignment/EvalueComputation.h:9)
// so that the restatement in oracle/sd_oracle.cpp and the HIP kernels can be
// diffed against the reference itself on arbitrary inputs, and so that bench.py can
// time the reference's AVX2 code as cpu_baseline.kind == "reference".
//
// M/ = /root/reference/lib/mmseqs/.  spacedust's ClusterHits.cpp is not linkable
// (DBReader/LocalParameters), see DESIGN.md.
#include "SubstitutionMatrix.h"
#include "Sequence.h"
#include "Masker.h"
#include "IndexTable.h"
#include "SequenceLookup.h"
#include "ExtendedSubstitutionMatrix.h"
#include "KmerGenerator.h"
#include "QueryMatcher.h"
#include "StripedSmithWaterman.h"
#include "EvalueComputation.h"
#include "Indexer.h"
#include "Util.h"
#include "Debug.h"

#include <cstring>
#include <string>
#include <vector>
#include <omp.h>

namespace {
struct RefCtx {
    SubstitutionMatrix *blosum2;   // blosum62, bitFactor 2, scoreBias 0     (SW; Alignment.cpp:152)
    SubstitutionMatrix *ungapped2; // blosum62, bitFactor 2, scoreBias -0.2  (diagonal scoring; Prefiltering.cpp:69,991)
    SubstitutionMatrix *seed8;     // VTML80,  bitFactor 8, scoreBias -0.2   (k-mer seeds; Prefiltering.cpp:68,991)
    ScoreMatrix three, two;
    bool haveExt;
    int kmerSize;
};
struct RefIndex {
    RefCtx *ctx;
    IndexTable *table;
    SequenceLookup *lookup;
    size_t nSeq;
    size_t maxLen;
    size_t aaSize;
    size_t masked;
};
struct RefPref {
    RefIndex *idx;
    QueryMatcher *matcher;
    Sequence *seq;
};
struct RefSW {
    RefCtx *ctx;
    SmithWaterman *sw;
    EvalueComputation *evaluer;
    Sequence *q;
    Sequence *qp = NULL;   // profile query (DBTYPE_HMM_PROFILE), created on first use
    Sequence *t;
    int8_t *tiny;
    size_t maxLen;
    int curQL = 0;
};
}

extern "C" {

void *ref_ctx_create(const char *blosumPath, const char *vtmlPath, int kmerSize) {
    Debug::setDebugLevel(1);
    RefCtx *c = new RefCtx();
    c->blosum2 = new SubstitutionMatrix(blosumPath, 2.0f, 0.0f);
    c->ungapped2 = new SubstitutionMatrix(blosumPath, 2.0f, -0.2f);
    c->seed8 = new SubstitutionMatrix(vtmlPath, 8.0f, -0.2f);
    c->haveExt = false;
    c->kmerSize = kmerSize;
    return c;
}

// which: 0 = blosum62@2 (SW), 1 = VTML80@8 (seed), 2 = blosum62@2 bias -0.2 (ungapped).  out: short[21*21], pBack double[21], aa2num uint8[256]
int ref_get_matrix(void *vc, int which, short *out, double *pback, unsigned char *aa2num) {
    RefCtx *c = (RefCtx *) vc;
    SubstitutionMatrix *m = which == 0 ? c->blosum2 : (which == 1 ? c->seed8 : c->ungapped2);
    for (int i = 0; i < m->alphabetSize; i++)
        for (int j = 0; j < m->alphabetSize; j++) out[i * m->alphabetSize + j] = m->subMatrix[i][j];
    for (int i = 0; i < m->alphabetSize; i++) pback[i] = m->pBack[i];
    for (int i = 0; i < 255; i++) aa2num[i] = m->aa2num[i];
    aa2num[255] = m->aa2num[(int) 'X'];
    return m->alphabetSize;
}

void ref_compbias(void *vc, int which, const unsigned char *num, int L, float scale, float *out) {
    RefCtx *c = (RefCtx *) vc;
    SubstitutionMatrix *m = which == 0 ? c->blosum2 : (which == 1 ? c->seed8 : c->ungapped2);
    SubstitutionMatrix::calcLocalAaBiasCorrection(m, num, L, out, scale);
}

static void ensureExt(RefCtx *c) {
    if (c->haveExt) return;
    // M/src/prefiltering/Prefiltering.cpp:209-214
    int a = c->seed8->alphabetSize;
    c->seed8->alphabetSize = a - 1;
    c->two = ExtendedSubstitutionMatrix::calcScoreMatrix(*c->seed8, 2);
    c->three = ExtendedSubstitutionMatrix::calcScoreMatrix(*c->seed8, 3);
    c->seed8->alphabetSize = a;
    c->haveExt = true;
}

// dump of the sorted 3-mer (which=3) or 2-mer (which=2) table
size_t ref_ext_matrix(void *vc, int which, short *score, unsigned int *index, size_t *rowSize) {
    RefCtx *c = (RefCtx *) vc;
    ensureExt(c);
    ScoreMatrix &s = which == 3 ? c->three : c->two;
    *rowSize = s.rowSize;
    if (score != NULL) {
        memcpy(score, s.score, s.elementSize * s.rowSize * sizeof(short));
        memcpy(index, s.index, s.elementSize * s.rowSize * sizeof(unsigned int));
    }
    return s.elementSize;
}

// similar k-mer list for one window of kmerSize numeric residues
size_t ref_kmer_list(void *vc, const unsigned char *window, int thr, size_t *out, size_t cap) {
    RefCtx *c = (RefCtx *) vc;
    ensureExt(c);
    KmerGenerator gen(c->kmerSize, c->seed8->alphabetSize - 1, (short) thr);
    gen.setDivideStrategy(&c->three, &c->two);
    std::pair<size_t *, size_t> r = gen.generateKmerList(window);
    size_t n = std::min(cap, r.second);
    memcpy(out, r.first, n * sizeof(size_t));
    return r.second;
}

// tantan-mask one numeric sequence in place (M/src/commons/Masker.cpp:15-55), returns #masked
int ref_mask(void *vc, unsigned char *num, int L, double maskProb) {
    RefCtx *c = (RefCtx *) vc;
    Sequence s(L + 2, Parameters::DBTYPE_AMINO_ACIDS, c->seed8, c->kmerSize, true, false);
    memcpy(s.numSequence, num, L);
    s.L = L;
    Masker masker(*c->seed8);
    int n = masker.maskSequence(s, true, maskProb, false, 0);
    memcpy(num, s.numSequence, L);
    return n;
}

// Build masked SequenceLookup + IndexTable for n target sequences given as ASCII
// (concatenated, offsets[n+1]).  Follows IndexBuilder::fillDatabase (non-profile branch).
void *ref_index_build(void *vc, const char *seqs, const size_t *offsets, size_t n, int kmerThr,
                      int mask, double maskProb, int threads) {
    RefCtx *c = (RefCtx *) vc;
    RefIndex *ix = new RefIndex();
    ix->ctx = c;
    ix->nSeq = n;
    size_t maxLen = 0, aa = 0;
    for (size_t i = 0; i < n; i++) {
        size_t l = offsets[i + 1] - offsets[i];
        maxLen = std::max(maxLen, l);
        aa += l;
    }
    ix->maxLen = maxLen;
    ix->aaSize = aa;
    BaseMatrix &subMat = *c->seed8;
    const int alphabetSize = subMat.alphabetSize - 1;   // Prefiltering.cpp:530-533
    ix->table = new IndexTable(alphabetSize, c->kmerSize, false);
    ix->lookup = new SequenceLookup(n, aa);
    char *idScoreLookup = new char[subMat.alphabetSize];
    for (int a = 0; a < subMat.alphabetSize; a++) idScoreLookup[a] = (char) subMat.subMatrix[a][a];
    size_t maskedResidues = 0, tableSize = 0;
    std::vector<size_t> seqOff(n + 1, 0);
    for (size_t i = 0; i < n; i++) seqOff[i + 1] = seqOff[i] + (offsets[i + 1] - offsets[i]);
#pragma omp parallel num_threads(threads)
    {
        Masker masker(subMat);
        Indexer idxer(alphabetSize, c->kmerSize);
        Sequence s(maxLen + 2, Parameters::DBTYPE_AMINO_ACIDS, &subMat, c->kmerSize, true, false);
        unsigned int *buffer = (unsigned int *) malloc((maxLen + 2) * sizeof(unsigned int));
#pragma omp for schedule(dynamic, 100) reduction(+:maskedResidues, tableSize)
        for (size_t id = 0; id < n; id++) {
            s.resetCurrPos();
            s.mapSequence(id, (unsigned int) id, seqs + offsets[id], (unsigned int) (offsets[id + 1] - offsets[id]));
            maskedResidues += masker.maskSequence(s, mask != 0, maskProb, false, 0);
            ix->lookup->addSequence(s.numSequence, s.L, id, seqOff[id]);
            ix->table->addKmerCount(&s, &idxer, buffer, kmerThr, idScoreLookup);
            if (Util::overlappingKmers(s.L, s.getEffectiveKmerSize()) > 0) tableSize += 1;
        }
        free(buffer);
    }
    ix->masked = maskedResidues;
    ix->table->initMemory(tableSize);
    ix->table->init();
#pragma omp parallel num_threads(threads)
    {
        Indexer idxer(alphabetSize, c->kmerSize);
        Sequence s(maxLen + 2, Parameters::DBTYPE_AMINO_ACIDS, &subMat, c->kmerSize, true, false);
        size_t bufferSize = maxLen + 2;
        IndexEntryLocalTmp *buffer = (IndexEntryLocalTmp *) malloc(bufferSize * sizeof(IndexEntryLocalTmp));
#pragma omp for schedule(dynamic, 100)
        for (size_t id = 0; id < n; id++) {
            s.resetCurrPos();
            s.mapSequence(id, (unsigned int) id, ix->lookup->getSequence(id));
            ix->table->addSequence(&s, &idxer, &buffer, bufferSize, kmerThr, idScoreLookup);
        }
        free(buffer);
    }
    delete[] idScoreLookup;
    ix->table->revertPointer();
    ix->table->sortDBSeqLists();
    return ix;
}

size_t ref_index_info(void *vi, size_t *tableSize, size_t *masked) {
    RefIndex *ix = (RefIndex *) vi;
    *tableSize = ix->table->getTableSize();
    *masked = ix->masked;
    return ix->table->getOffsets()[ix->table->getTableSize()];
}

// offsets: size_t[tableSize+1]; entries packed {u32 seqId; u16 pos}; lookup: masked numeric residues
void ref_index_dump(void *vi, size_t *offsets, unsigned char *entries, unsigned char *lookup) {
    RefIndex *ix = (RefIndex *) vi;
    size_t ts = ix->table->getTableSize();
    memcpy(offsets, ix->table->getOffsets(), (ts + 1) * sizeof(size_t));
    memcpy(entries, ix->table->getEntries(), offsets[ts] * sizeof(IndexEntryLocal));
    if (lookup != NULL) memcpy(lookup, ix->lookup->getData(), ix->aaSize);
}

void *ref_prefilter_create(void *vi, int kmerThr, size_t maxQueryLen, size_t maxHits, int minDiagScore,
                           int compBias) {
    RefIndex *ix = (RefIndex *) vi;
    RefCtx *c = ix->ctx;
    ensureExt(c);
    RefPref *p = new RefPref();
    p->idx = ix;
    size_t maxLen = std::max(ix->maxLen, maxQueryLen) + 2;
    p->matcher = new QueryMatcher(ix->table, ix->lookup, c->seed8, c->ungapped2, (short) kmerThr, c->kmerSize,
                                  ix->nSeq, (unsigned int) maxLen, maxHits, compBias != 0, 1.0f, true,
                                  (unsigned int) minDiagScore, false, false);
    p->matcher->setSubstitutionMatrix(&c->three, &c->two);
    p->seq = new Sequence(maxLen, Parameters::DBTYPE_AMINO_ACIDS, c->seed8, c->kmerSize, true, compBias != 0);
    return p;
}

// out: triples (seqId, score, diagonal as uint16) ; returns count.  stats[0]=kmersPerPos*L stats[1]=dbMatches
size_t ref_prefilter_query(void *vp, const char *seq, unsigned int L, unsigned int identityId,
                           unsigned int *outId, int *outScore, unsigned short *outDiag, double *stats) {
    RefPref *p = (RefPref *) vp;
    p->seq->mapSequence(0, 0, seq, L);
    std::pair<hit_t *, size_t> r = p->matcher->matchQuery(p->seq, identityId, false);
    for (size_t i = 0; i < r.second; i++) {
        outId[i] = r.first[i].seqId;
        outScore[i] = r.first[i].prefScore;
        outDiag[i] = r.first[i].diagonal;
    }
    if (stats != NULL) {
        stats[0] = p->matcher->getStatistics()->kmersPerPos * (double) L;
        stats[1] = (double) p->matcher->getStatistics()->dbMatches;
    }
    return r.second;
}

// profile queries: the matcher takes its k-mer generator rows from the Sequence (Prefiltering.cpp:790-795:
// matcher.setProfileMatrix(seq.profile_matrix)); the index must have been built with threshold 0 (:525-527)
void *ref_prefilter_create_profile(void *vi, int kmerThr, size_t maxQueryLen, size_t maxHits, int minDiagScore) {
    RefIndex *ix = (RefIndex *) vi;
    RefCtx *c = ix->ctx;
    RefPref *p = new RefPref();
    p->idx = ix;
    size_t maxLen = std::max(ix->maxLen, maxQueryLen) + 2;
    p->matcher = new QueryMatcher(ix->table, ix->lookup, c->seed8, c->ungapped2, (short) kmerThr, c->kmerSize,
                                  ix->nSeq, (unsigned int) maxLen, maxHits, true, 1.0f, true,
                                  (unsigned int) minDiagScore, false, false);
    p->seq = new Sequence(maxLen, Parameters::DBTYPE_HMM_PROFILE, c->seed8, c->kmerSize, true, true);
    p->matcher->setProfileMatrix(p->seq->profile_matrix);
    return p;
}

// similar k-mers of the window starting at position `pos` of a profile (KmerGenerator over the Sequence's rows)
size_t ref_profile_kmer_list(void *vc, const char *data, unsigned int L, unsigned int pos, int thr, size_t *out, size_t cap) {
    RefCtx *c = (RefCtx *) vc;
    Sequence s(L + 2, Parameters::DBTYPE_HMM_PROFILE, c->seed8, c->kmerSize, true, false);
    s.mapSequence(0, 0, data, L);
    KmerGenerator gen(c->kmerSize, c->seed8->alphabetSize - 1, (short) thr);
    gen.setDivideStrategy(s.profile_matrix);
    const unsigned char *kmer = NULL;
    for (unsigned int i = 0; i <= pos && s.hasNextKmer(); i++) kmer = s.nextKmer();
    std::pair<size_t *, size_t> r = gen.generateKmerList(kmer);
    for (size_t i = 0; i < r.second && i < cap; i++) out[i] = r.first[i];
    return r.second;
}

void ref_prefilter_destroy(void *vp) {
    RefPref *p = (RefPref *) vp;
    delete p->matcher;
    delete p->seq;
    delete p;
}

void ref_index_destroy(void *vi) {
    RefIndex *ix = (RefIndex *) vi;
    delete ix->table;
    delete ix->lookup;
    delete ix;
}

void *ref_sw_create(void *vc, size_t maxLen, size_t dbResidues, int compBias) {
    RefCtx *c = (RefCtx *) vc;
    RefSW *s = new RefSW();
    s->ctx = c;
    s->maxLen = maxLen + 2;
    BaseMatrix *m = c->blosum2;
    s->sw = new SmithWaterman(s->maxLen, m->alphabetSize, compBias != 0, 1.0f, Parameters::DBTYPE_AMINO_ACIDS);
    s->evaluer = new EvalueComputation(dbResidues, m, 11, 1);
    s->q = new Sequence(s->maxLen, Parameters::DBTYPE_AMINO_ACIDS, m, 0, false, compBias != 0);
    s->t = new Sequence(s->maxLen, Parameters::DBTYPE_AMINO_ACIDS, m, 0, false, compBias != 0);
    s->tiny = new int8_t[m->alphabetSize * m->alphabetSize];
    for (int i = 0; i < m->alphabetSize; i++)
        for (int j = 0; j < m->alphabetSize; j++) s->tiny[i * m->alphabetSize + j] = (int8_t) m->subMatrix[i][j];
    return s;
}

void ref_sw_set_query(void *vs, const char *seq, unsigned int L) {
    RefSW *s = (RefSW *) vs;
    s->q->mapSequence(0, 0, seq, L);
    s->sw->ssw_init(s->q, s->tiny, s->ctx->blosum2);
    s->curQL = s->q->L;
}

// profile query: data = L records of Sequence::PROFILE_READIN_SIZE bytes (Matcher::initQuery's profile branch:
// ssw_init(query, query->getAlignmentProfile(), m))
void ref_sw_set_query_profile(void *vs, const char *data, unsigned int L) {
    RefSW *s = (RefSW *) vs;
    BaseMatrix *m = s->ctx->blosum2;
    if (s->qp == NULL) s->qp = new Sequence(s->maxLen, Parameters::DBTYPE_HMM_PROFILE, m, 0, false, false);
    s->qp->mapSequence(0, 0, data, L);
    s->sw->ssw_init(s->qp, s->qp->getAlignmentProfile(), m);
    s->curQL = s->qp->L;
}

// out[0..7] = score, qStart, qEnd, tStart, tEnd, identical, cigarLen(backtrace length), 0 ; returns evalue.
// swMode: 0 score only, 1 score+cov (start positions), 2 score+cov+seqid (backtrace).
double ref_sw_align(void *vs, const char *tseq, unsigned int tL, int swMode, double evalThr, int covMode,
                    float covThr, int *out, char *backtrace, size_t btCap, int isIdentity) {
    RefSW *s = (RefSW *) vs;
    s->t->mapSequence(1, 1, tseq, tL);
    std::string bt;
    s_align a;
    if (isIdentity) {
        a = s->sw->scoreIdentical(s->t->numSequence, s->t->L, s->evaluer, swMode, bt);
    } else {
        a = s->sw->ssw_align(s->t->numSequence, s->t->numConsensusSequence, s->t->getAlignmentProfile(), s->t->L, bt,
                             11, 1, (uint8_t) swMode, evalThr, s->evaluer, covMode, covThr, 0.0f, s->curQL / 2, 1);
    }
    out[0] = (int) a.score1;
    out[1] = a.qStartPos1;
    out[2] = a.qEndPos1;
    out[3] = a.dbStartPos1;
    out[4] = a.dbEndPos1;
    out[5] = (int) a.identicalAACnt;
    out[6] = (int) bt.size();
    out[7] = 0;
    if (backtrace != NULL && btCap > 0) {
        size_t n = std::min(btCap - 1, bt.size());
        memcpy(backtrace, bt.data(), n);
        backtrace[n] = '\0';
    }
    if (isIdentity == 0) delete[] a.cigar;
    return a.evalue;
}

double ref_evalue(void *vs, double score, double qLen) {
    return ((RefSW *) vs)->evaluer->computeEvalue(score, qLen);
}
double ref_bitscore(void *vs, double score) {
    return ((RefSW *) vs)->evaluer->computeBitScore(score);
}

void ref_sw_destroy(void *vs) {
    RefSW *s = (RefSW *) vs;
    delete s->sw;
    delete s->evaluer;
    delete s->q;
    delete s->qp;
    delete s->t;
    delete[] s->tiny;
    delete s;
}

unsigned long ref_l2_cache_size() { return Util::getL2CacheSize(); }

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// cpu_baseline driver: the reference's per-query loop bodies (Prefiltering::runSplit :817-886 and
// Alignment::run :312-514) over a sample of queries with OpenMP threads, each thread owning its
// QueryMatcher / SmithWaterman exactly as the reference's threads do.  Stops at the deadline.
// out[0] = queries done, out[1] = pairs aligned, out[2] = forward DP cells, out[3] = seconds,
// out[4] = thread-seconds spent inside the Smith-Waterman calls alone (sum over the threads)
// ------------------------------------------------------------------------------------------------
#include <chrono>
// evalThr: the alignment's E-value gate (Alignment.cpp:379 hands par.evalThr to getSWResult): 10 is clustersearch's default; bench.py
// also times the reference with the bound the device pipeline pushes down from combinehits (same final cluster hits, less work)
extern "C" int ref_run_queries(void *vi, const char *seqs, const size_t *offsets, const unsigned int *sample,
                               size_t nSample, int kmerThr, size_t maxHits, int threads, double seconds,
                               size_t dbResidues, double *out, double evalThr) {
    RefIndex *ix = (RefIndex *) vi;
    RefCtx *c = ix->ctx;
    ensureExt(c);
    size_t maxLen = ix->maxLen + 2;
    std::vector<RefPref *> pf(threads);
    std::vector<RefSW *> sw(threads);
    for (int t = 0; t < threads; t++) {
        pf[t] = (RefPref *) ref_prefilter_create(ix, kmerThr, maxLen, maxHits, 15, 1);
        sw[t] = (RefSW *) ref_sw_create(c, maxLen, dbResidues, 1);
    }
    size_t done = 0, pairs = 0, cells = 0;
    double swSeconds = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    bool stop = false;
#pragma omp parallel num_threads(threads) reduction(+ : done, pairs, cells, swSeconds)
    {
        const int t = omp_get_thread_num();
        std::vector<unsigned int> ids(maxHits + 2);
        std::vector<int> sc(maxHits + 2);
        std::vector<unsigned short> dg(maxHits + 2);
        int res[8];
#pragma omp for schedule(dynamic, 1)
        for (size_t s = 0; s < nSample; s++) {
            if (stop) continue;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) {
                stop = true;
                continue;
            }
            const unsigned int q = sample[s];
            const char *qs = seqs + offsets[q];
            const unsigned int qL = (unsigned int) (offsets[q + 1] - offsets[q]);
            size_t n = ref_prefilter_query(pf[t], qs, qL, q, ids.data(), sc.data(), dg.data(), NULL);
            const auto s0 = std::chrono::steady_clock::now();
            ref_sw_set_query(sw[t], qs, qL);
            for (size_t h = 0; h < n; h++) {
                const unsigned int tid = ids[h];
                const unsigned int tL = (unsigned int) (offsets[tid + 1] - offsets[tid]);
                if (Util::canBeCovered(0.8f, Parameters::COV_MODE_QUERY, (float) qL, (float) tL) == false) continue;
                ref_sw_align(sw[t], seqs + offsets[tid], tL, 2, evalThr, Parameters::COV_MODE_QUERY, 0.8f, res, NULL, 0, tid == q);
                pairs++;
                cells += (size_t) qL * tL;
            }
            swSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count();
            done++;
        }
    }
    out[3] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out[0] = (double) done;
    out[1] = (double) pairs;
    out[2] = (double) cells;
    out[4] = swSeconds;
    for (int t = 0; t < threads; t++) {
        ref_prefilter_destroy(pf[t]);
        ref_sw_destroy(sw[t]);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Whole query sets through the reference's two loop bodies, every alignment kept: the input of an
// independent restatement of the aggregation modules (oracle/agg_restatement.py), so that the
// (query set, target set) entries of a measured run can be compared with something reference-derived
// at the run's own size.  Per aligned pair one row of 8 doubles:
//   query, target, score, E-value, bit score, qStart, qEnd, backtrace length (0: stopped at a gate)
// Returns the number of rows (rows beyond cap are counted, not written).
// ------------------------------------------------------------------------------------------------
extern "C" size_t ref_run_query_set(void *vi, const char *seqs, const size_t *offsets, const unsigned int *queries,
                                    size_t nQueries, int kmerThr, size_t maxHits, int threads, size_t dbResidues,
                                    double *rows, size_t cap) {
    RefIndex *ix = (RefIndex *) vi;
    RefCtx *c = ix->ctx;
    ensureExt(c);
    size_t maxLen = ix->maxLen + 2;
    std::vector<RefPref *> pf(threads);
    std::vector<RefSW *> sw(threads);
    for (int t = 0; t < threads; t++) {
        pf[t] = (RefPref *) ref_prefilter_create(ix, kmerThr, maxLen, maxHits, 15, 1);
        sw[t] = (RefSW *) ref_sw_create(c, maxLen, dbResidues, 1);
    }
    size_t nRows = 0;
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num();
        std::vector<unsigned int> ids(maxHits + 2);
        std::vector<int> sc(maxHits + 2);
        std::vector<unsigned short> dg(maxHits + 2);
        std::vector<double> mine;
        int res[8];
#pragma omp for schedule(dynamic, 1)
        for (size_t s = 0; s < nQueries; s++) {
            const unsigned int q = queries[s];
            const char *qs = seqs + offsets[q];
            const unsigned int qL = (unsigned int) (offsets[q + 1] - offsets[q]);
            size_t n = ref_prefilter_query(pf[t], qs, qL, q, ids.data(), sc.data(), dg.data(), NULL);
            ref_sw_set_query(sw[t], qs, qL);
            mine.clear();
            for (size_t h = 0; h < n; h++) {
                const unsigned int tid = ids[h];
                const unsigned int tL = (unsigned int) (offsets[tid + 1] - offsets[tid]);
                if (Util::canBeCovered(0.8f, Parameters::COV_MODE_QUERY, (float) qL, (float) tL) == false) continue;
                const double ev = ref_sw_align(sw[t], seqs + offsets[tid], tL, 2, 10.0, Parameters::COV_MODE_QUERY, 0.8f, res, NULL, 0, tid == q);
                const double row[8] = {(double) q, (double) tid, (double) res[0], ev, ref_bitscore(sw[t], (double) res[0]),
                                       (double) res[1], (double) res[2], (double) res[6]};
                mine.insert(mine.end(), row, row + 8);
            }
            size_t at;
#pragma omp critical
            {
                at = nRows;
                nRows += mine.size() / 8;
            }
            for (size_t r = 0; r < mine.size() / 8; r++)
                if (at + r < cap) memcpy(rows + (at + r) * 8, mine.data() + r * 8, 8 * sizeof(double));
        }
    }
    for (int t = 0; t < threads; t++) {
        ref_prefilter_destroy(pf[t]);
        ref_sw_destroy(sw[t]);
    }
    return nRows;
}
