// The tail of Alignment::run for a batch of alignment records (batched alignment text):
//   Alignment::checkCriteria                (M/src/alignment/Alignment.cpp:389-399,548-567)
//   SORT_SERIAL(swResults, compareHits)     (Alignment.cpp:403-405, Matcher.h:157-168)
//   the --realign second pass               (Alignment.cpp:408-440)
//   Matcher::resultToBuffer                 (M/src/alignment/Matcher.cpp:280-327)
// The derived fields of Matcher::getSWResult that are not in sd_sw_result (coverage, sequence identity, alignment length,
// bit score; Matcher.cpp:88-137) are formed here from the record, per alignment mode.
#include "sd_host.h"
#include "spacedust_gpu.h"

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <omp.h>

namespace {

struct Derived {
    float qcov, dbcov, seqId;
    unsigned alnLength;
    int bits;
};

// Matcher::getSWResult after ssw_align (Matcher.cpp:88-137)
inline Derived derive(const sd_sw_result &r, int swMode, int seqIdMode, int qLen, int tLen, bool identity) {
    Derived d;
    d.qcov = 0.0f;
    d.dbcov = 0.0f;
    d.seqId = 0.0f;
    const unsigned qS = (unsigned) r.qStart, qE = (unsigned) r.qEnd, tS = (unsigned) r.tStart, tE = (unsigned) r.tEnd;
    if (swMode == 1 || swMode == 2) {
        // s_align.qCov / tCov: set where the reference computes start positions (StripedSmithWaterman.cpp:483-489)
        if (r.qStart >= 0 && r.tStart >= 0) {
            d.qcov = sd::computeCov(qS, qE, (unsigned) qLen);
            d.dbcov = sd::computeCov(tS, tE, (unsigned) tLen);
        }
    }
    d.alnLength = (unsigned) (std::max(std::abs((int) qE - (int) qS), std::abs((int) tE - (int) tS)) + 1);   // Matcher::computeAlnLength
    if (swMode == 2) {
        if (r.btLen > 0) d.alnLength = (unsigned) r.btLen;
        switch (seqIdMode) {   // Util::computeSeqId (M/src/commons/Util.cpp:532-542)
            case 1: d.seqId = static_cast<float>(r.identical) / static_cast<float>(std::min(qLen, tLen)); break;
            case 2: d.seqId = static_cast<float>(r.identical) / static_cast<float>(std::max(qLen, tLen)); break;
            default: d.seqId = static_cast<float>(r.identical) / static_cast<float>(d.alnLength); break;
        }
    } else {
        unsigned qAln, tAln;
        if (swMode == 1) {
            qAln = std::max(qE - qS, 1u);
            tAln = std::max(tE - tS, 1u);
        } else {
            qAln = std::max(qE, 1u);
            tAln = std::max(tE, 1u);
        }
        // Matcher::estimateSeqIdByScorePerCol (Matcher.cpp:160-164): the score enters as uint16_t
        float e = ((uint16_t) r.score / static_cast<float>(std::max(qAln, tAln))) * 0.1656 + 0.1141;
        e = std::min(e, 1.0f);
        d.seqId = std::max(0.0f, e);
    }
    d.bits = static_cast<int>(sd_host_bitscore((double) (uint32_t) r.score) + 0.5);   // Matcher.cpp:130
    if (identity) {   // Alignment.cpp:382-387
        d.qcov = 1.0f;
        d.dbcov = 1.0f;
        d.seqId = 1.0f;
    }
    return d;
}

inline bool criteria(const sd_aln_criteria &c, const sd_sw_result &r, const Derived &d, bool identity, float covThr) {
    if (identity) return true;
    const bool evalOk = r.evalue <= c.evalThr;
    const bool seqIdOk = d.seqId >= c.seqIdThr;
    const bool covOk = sd::hasCoverage(covThr, c.covMode, d.qcov, d.dbcov);
    const bool lenOk = d.alnLength >= (unsigned) c.alnLenThr;   // Util::hasAlignmentLength
    return evalOk && seqIdOk && covOk && lenOk;
}

struct Key {
    double eval;
    int bits;
    int dbLen;
    uint32_t dbKey;
    uint32_t idx;
};
inline bool hitLess(const Key &a, const Key &b) {   // Matcher::compareHits
    if (a.eval != b.eval) return a.eval < b.eval;
    if (a.bits != b.bits) return a.bits > b.bits;
    if (a.dbLen != b.dbLen) return a.dbLen < b.dbLen;
    return a.dbKey < b.dbKey;
}

}  // namespace

struct sd_alntext {
    std::string text;
    std::vector<uint64_t> entryOff;
    std::vector<std::string> perQuery;   // formatting scratch, one string per query of the last call (capacity kept)
};

extern "C" {

int sd_host_accept_sort(const sd_aln_criteria *crit, uint32_t nQ, uint32_t nRes, const uint32_t *resQ, const uint32_t *resT,
                        const sd_sw_result *res, const uint8_t *isIdentity, const int32_t *qLen, const int32_t *tLen,
                        const uint32_t *tKey, uint32_t *order, uint32_t *countPerQuery) {
    if (!crit || !order || !countPerQuery || (nRes && (!resQ || !resT || !res || !qLen || !tLen))) return SD_EINVAL;
    for (uint32_t i = 1; i < nRes; i++)
        if (resQ[i] < resQ[i - 1]) return SD_EINVAL;
    std::vector<uint32_t> start((size_t) nQ + 1, 0);
    for (uint32_t i = 0; i < nRes; i++) {
        if (resQ[i] >= nQ) return SD_EINVAL;
        start[resQ[i] + 1]++;
    }
    for (uint32_t q = 0; q < nQ; q++) start[q + 1] += start[q];
    // in realign mode the first pass runs without a coverage threshold (Alignment.cpp:48-52)
    const float covThr = crit->realign ? 0.0f : crit->covThr;
    std::vector<uint32_t> accepted((size_t) nQ, 0);
    // every query writes its accepted indices at the front of its own segment of `order`; compacted afterwards
    std::vector<uint32_t> scratch(nRes);
#pragma omp parallel
    {
        std::vector<Key> keys;
#pragma omp for schedule(dynamic, 64)
        for (uint32_t q = 0; q < nQ; q++) {
            keys.clear();
            unsigned rejected = 0, passed = 0;
            for (uint32_t i = start[q]; i < start[q + 1]; i++) {
                if (passed >= crit->maxAccept || rejected >= crit->maxRejected) break;   // Alignment.cpp:349
                const bool ident = isIdentity && isIdentity[i];
                const int tl = tLen[resT[i]];
                const Derived d = derive(res[i], crit->swMode, crit->seqIdMode, qLen[q], tl, ident);
                if (!criteria(*crit, res[i], d, ident, covThr)) {
                    rejected++;
                    continue;
                }
                rejected = 0;
                passed++;
                Key k;
                k.eval = res[i].evalue;
                k.bits = d.bits;
                k.dbLen = tl;
                k.dbKey = tKey ? tKey[resT[i]] : resT[i];
                k.idx = i;
                keys.push_back(k);
            }
            if (keys.size() > 1) std::sort(keys.begin(), keys.end(), hitLess);
            for (size_t x = 0; x < keys.size(); x++) scratch[start[q] + x] = keys[x].idx;
            accepted[q] = (uint32_t) keys.size();
        }
    }
    uint64_t w = 0;
    for (uint32_t q = 0; q < nQ; q++) {
        countPerQuery[q] = accepted[q];
        for (uint32_t x = 0; x < accepted[q]; x++) order[w++] = scratch[start[q] + x];
    }
    return SD_OK;
}

int sd_host_realign_select(const sd_aln_criteria *crit, uint32_t nQ, const uint32_t *countPerQuery, const uint32_t *order,
                           const uint32_t *resT, const sd_sw_result *first, const sd_sw_result *second,
                           const uint8_t *isIdentity, const int32_t *qLen, const int32_t *tLen, const uint32_t *tKey,
                           sd_sw_result *merged, uint32_t *outOrder, uint32_t *outCount) {
    if (!crit || !countPerQuery || !order || !first || !second || !merged || !outOrder || !outCount) return SD_EINVAL;
    // `second` / `merged` / isIdentity are indexed like `order` (one record per accepted first-pass hit, in order)
    std::vector<uint64_t> base((size_t) nQ + 1, 0);
    for (uint32_t q = 0; q < nQ; q++) base[q + 1] = base[q] + countPerQuery[q];
    std::vector<uint32_t> scratch(std::max<uint64_t>(base[nQ], 1));   // every query's order at the front of its own segment
#pragma omp parallel
    {
        std::vector<Key> keys;
#pragma omp for schedule(dynamic, 64)
        for (uint32_t q = 0; q < nQ; q++) {
            keys.clear();
            int acceptedN = 0;
            for (uint32_t x = 0; x < countPerQuery[q] && acceptedN < crit->realignMaxSeqs; x++) {
                const uint64_t s = base[q] + x;
                const uint32_t i = order[s];
                const bool ident = isIdentity && isIdentity[s];
                const int tl = tLen[resT[i]];
                const int mode = crit->realignSwMode;
                Derived d = derive(second[s], mode, crit->seqIdMode, qLen[q], tl, false);
                const bool covOk = sd::hasCoverage(crit->covThr, crit->covMode, d.qcov, d.dbcov);
                if (!(covOk || ident)) continue;
                merged[s] = second[s];
                merged[s].score = first[i].score;      // res.score / res.eval of the first pass (Alignment.cpp:426-427)
                merged[s].evalue = first[i].evalue;
                acceptedN++;
                Key k;
                k.eval = first[i].evalue;
                k.bits = static_cast<int>(sd_host_bitscore((double) (uint32_t) first[i].score) + 0.5);
                k.dbLen = tl;
                k.dbKey = tKey ? tKey[resT[i]] : resT[i];
                k.idx = (uint32_t) s;
                keys.push_back(k);
            }
            if (keys.size() > 1) std::sort(keys.begin(), keys.end(), hitLess);
            for (size_t x = 0; x < keys.size(); x++) scratch[base[q] + x] = keys[x].idx;
            outCount[q] = (uint32_t) keys.size();
        }
    }
    uint64_t w = 0;
    for (uint32_t q = 0; q < nQ; q++)
        for (uint32_t x = 0; x < outCount[q]; x++) outOrder[w++] = scratch[base[q] + x];
    return SD_OK;
}

int sd_alntext_create(sd_alntext **out) {
    if (!out) return SD_EINVAL;
    *out = new sd_alntext();
    return SD_OK;
}

void sd_alntext_destroy(sd_alntext *t) { delete t; }

int sd_alntext_format(sd_alntext *t, const sd_aln_criteria *crit, uint32_t nQ, const uint32_t *countPerQuery,
                      const uint32_t *order, const uint32_t *recT, const sd_sw_result *rec, const uint8_t *isIdentity,
                      const char *btPool, const int32_t *qLen, const int32_t *tLen, const uint32_t *tKey) {
    if (!t || !crit || !countPerQuery || (nQ && !qLen)) return SD_EINVAL;
    std::vector<uint64_t> start((size_t) nQ + 1, 0);
    for (uint32_t q = 0; q < nQ; q++) start[q + 1] = start[q] + countPerQuery[q];
    const uint64_t n = start[nQ];
    if (n && (!order || !recT || !rec || !tLen)) return SD_EINVAL;
    // every query formats its lines into a string of its own (no worst-case slab: a backtrace of n columns compresses to anything
    // between 2 and 2 n characters), then the entries are laid out back to back
    const int mode = crit->realign ? crit->realignSwMode : crit->swMode;
    std::vector<std::string> &perQ = t->perQuery;
    if (perQ.size() < nQ) perQ.resize(nQ);
#pragma omp parallel
    {
        std::string cigar;
        char head[11 * 12 + 64];
#pragma omp for schedule(dynamic, 16)
        for (uint32_t q = 0; q < nQ; q++) {
            std::string &out = perQ[q];
            out.clear();
            for (uint64_t x = start[q]; x < start[q + 1]; x++) {
                const uint32_t i = order[x];
                const sd_sw_result &r = rec[i];
                const bool ident = isIdentity && isIdentity[i];
                const int tl = tLen[recT[i]];
                const Derived d = derive(r, mode, crit->seqIdMode, qLen[q], tl, ident);
                char *p = sd::u32toa(tKey ? tKey[recT[i]] : recT[i], head);
                *p++ = '\t';
                p = sd::i32toa(d.bits, p);
                *p++ = '\t';
                p = sd::seqIdToBuffer(d.seqId, p);
                *p++ = '\t';
                p += snprintf(p, 32, "%.3E", r.evalue);
                *p++ = '\t';
                p = sd::i32toa(r.qStart, p);
                *p++ = '\t';
                p = sd::i32toa(r.qEnd, p);
                *p++ = '\t';
                p = sd::i32toa(qLen[q], p);
                *p++ = '\t';
                p = sd::i32toa(r.tStart, p);
                *p++ = '\t';
                p = sd::i32toa(r.tEnd, p);
                *p++ = '\t';
                p = sd::i32toa(tl, p);
                if (crit->addBacktrace) *p++ = '\t';
                out.append(head, (size_t) (p - head));
                if (crit->addBacktrace) {
                    // Matcher::compressAlignment (Matcher.cpp:166-185): an empty backtrace compresses to "0M"
                    const size_t btN = (r.btLen > 0 && btPool) ? (size_t) r.btLen : 0;
                    sd::compressBacktraceAppend(btN ? btPool + r.btOffset : "", btN, out);
                }
                out.push_back('\n');
            }
        }
    }
    t->entryOff.assign((size_t) nQ + 1, 0);
    for (uint32_t q = 0; q < nQ; q++) t->entryOff[q + 1] = t->entryOff[q] + perQ[q].size();
    t->text.resize(t->entryOff[nQ]);
#pragma omp parallel for schedule(static)
    for (uint32_t q = 0; q < nQ; q++)
        if (!perQ[q].empty()) memcpy(&t->text[t->entryOff[q]], perQ[q].data(), perQ[q].size());
    return SD_OK;
}

int sd_alntext_get(sd_alntext *t, const char **text, const uint64_t **entryOff) {
    if (!t) return SD_EINVAL;
    if (text) *text = t->text.data();
    if (entryOff) *entryOff = t->entryOff.data();
    return SD_OK;
}

}  // extern "C"
