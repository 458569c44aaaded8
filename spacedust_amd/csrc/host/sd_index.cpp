// Target k-mer index construction (host).  Semantics follow
// IndexBuilder::fillDatabase (M/src/prefiltering/IndexBuilder.cpp:55-239) for an amino-acid
// target searched with sequence queries:
//   1. every target is tantan-masked to X (Masker.cpp:15-55) and stored in the lookup,
//   2. per target, the spaced k-mers without X whose self score sum(seedMat[a][a]) >= kmerThr
//      are collected (IndexTable.h:131-166), each distinct k-mer once per target at its smallest
//      position (sort by (kmer,pos), IndexTable.h:50-61,376-392),
//   3. lists are ordered by (seqId, pos) (IndexTable.h:182-189).
// The in-memory layout differs from the reference (u32 offsets, SoA entries) because it is
// the layout the HIP kernels read; tests compare it entry for entry with the reference's.
#include "sd_host.h"

#include <algorithm>
#include <cstring>
#include <omp.h>

namespace sd {

namespace {
struct Tmp {
    uint32_t kmer;
    uint16_t pos;
};
inline bool tmpLess(const Tmp &a, const Tmp &b) {
    if (a.kmer != b.kmer) return a.kmer < b.kmer;
    return a.pos < b.pos;
}

// collect (kmer,pos) of one masked sequence, sorted by (kmer,pos)
size_t collect(const uint8_t *s, int L, int k, int span, const uint8_t *seedPos, const int *selfScore, int thr,
               const uint64_t *powers, std::vector<Tmp> &buf) {
    buf.clear();
    for (int i = 0; i + span <= L; i++) {
        bool hasX = false;
        int score = 0;
        uint64_t idx = 0;
        for (int p = 0; p < k; p++) {
            uint8_t a = s[i + seedPos[p]];
            hasX |= (a == X_CODE);
            if (a < ALPH) score += selfScore[a];
            idx += (uint64_t) a * powers[p];
        }
        if (hasX) continue;
        if (thr > 0 && score < thr) continue;
        Tmp t;
        t.kmer = (uint32_t) idx;
        t.pos = (uint16_t) i;
        buf.push_back(t);
    }
    if (buf.size() > 1) std::sort(buf.begin(), buf.end(), tmpLess);
    return buf.size();
}
}  // namespace

void buildTargetIndex(const SubMat &seed8, const uint8_t *seqs, const uint64_t *offsets, uint32_t nSeq, int k,
                      int kmerThr, bool mask, double maskProb, int threads, TargetIndex &out) {
    out.k = k;
    out.span = spacedPattern(k, out.seedPos);
    uint64_t tableSize = 1;
    uint64_t powers[8];
    for (int i = 0; i < k; i++) {
        powers[i] = tableSize;
        tableSize *= (ALPH - 1);
    }
    out.tableSize = tableSize;
    out.seqOffsets.assign(offsets, offsets + nSeq + 1);
    const uint64_t total = offsets[nSeq];
    out.masked.assign(seqs, seqs + total);
    int selfScore[ALPH];
    for (int a = 0; a < ALPH; a++) selfScore[a] = (int) (char) seed8.sub[a][a];   // IndexBuilder.cpp:10-21

    MaskCtx mctx;
    initMaskCtx(seed8, mctx);
    uint64_t maskedResidues = 0;
    std::vector<uint32_t> counts(tableSize + 1, 0);
#pragma omp parallel num_threads(threads)
    {
        std::vector<Tmp> buf;
#pragma omp for schedule(dynamic, 100) reduction(+ : maskedResidues)
        for (uint32_t id = 0; id < nSeq; id++) {
            uint8_t *s = out.masked.data() + offsets[id];
            int L = (int) (offsets[id + 1] - offsets[id]);
            if (mask) maskedResidues += tantanMask(mctx, s, L, maskProb);
            collect(s, L, k, out.span, out.seedPos, selfScore, kmerThr, powers, buf);
            uint32_t prev = UINT32_MAX;
            for (size_t i = 0; i < buf.size(); i++) {
                if (buf[i].kmer != prev) {
#pragma omp atomic
                    counts[buf[i].kmer]++;
                }
                prev = buf[i].kmer;
            }
        }
    }
    out.maskedResidues = maskedResidues;
    out.offsets.assign(tableSize + 1, 0);
    uint64_t run = 0;
    for (uint64_t i = 0; i < tableSize; i++) {
        out.offsets[i] = (uint32_t) run;
        run += counts[i];
    }
    out.offsets[tableSize] = (uint32_t) run;
    out.entrySeq.assign(run, 0);
    out.entryPos.assign(run, 0);
    // fill in target order so every list comes out sorted by (seqId,pos) without a second sort:
    // targets are processed in blocks; a serial pass over per-block results keeps seqId ascending.
    std::vector<uint32_t> cursor(out.offsets.begin(), out.offsets.end() - 1);
    const uint32_t BLOCK = 4096;
    for (uint32_t b0 = 0; b0 < nSeq; b0 += BLOCK) {
        uint32_t b1 = std::min(nSeq, b0 + BLOCK);
        std::vector<std::vector<Tmp> > res(b1 - b0);
#pragma omp parallel num_threads(threads)
        {
            std::vector<Tmp> buf;
#pragma omp for schedule(dynamic, 16)
            for (uint32_t id = b0; id < b1; id++) {
                const uint8_t *s = out.masked.data() + offsets[id];
                int L = (int) (offsets[id + 1] - offsets[id]);
                collect(s, L, k, out.span, out.seedPos, selfScore, kmerThr, powers, buf);
                std::vector<Tmp> &r = res[id - b0];
                uint32_t prev = UINT32_MAX;
                for (size_t i = 0; i < buf.size(); i++) {
                    if (buf[i].kmer != prev) r.push_back(buf[i]);
                    prev = buf[i].kmer;
                }
            }
        }
        for (uint32_t id = b0; id < b1; id++) {
            const std::vector<Tmp> &r = res[id - b0];
            for (size_t i = 0; i < r.size(); i++) {
                uint32_t c = cursor[r[i].kmer]++;
                out.entrySeq[c] = id;
                out.entryPos[c] = r[i].pos;
            }
        }
    }
}

}  // namespace sd
