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
#include <cstdlib>
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
    // one table of 20^k + 2 counters serves as histogram, cursor array and final offsets (k = 7: 5 GB, so no copies):
    // list sizes are counted into A[kmer + 2]; after the inclusive prefix sum A[kmer + 1] is the start of the list, and it
    // is that slot the fill advances, which leaves A[kmer] = start of list kmer for every kmer when the fill is done.
    ZeroedU32 &A = out.offsets;
    if (!A.reset(tableSize + 2)) {
        out.tableSize = 0;
        return;
    }
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
                    A[(size_t) buf[i].kmer + 2]++;
                }
                prev = buf[i].kmer;
            }
        }
    }
    out.maskedResidues = maskedResidues;
    uint64_t run = 0;
    {
        // two-pass parallel inclusive prefix sum
        const int nb = std::max(1, threads);
        // ranges are whole 65 536-slot blocks: a block's base is set by the thread that then uses it (wide form below)
        const uint64_t n = tableSize + 2, blk = (((n + nb - 1) / nb) + 0xFFFFull) & ~0xFFFFull;
        std::vector<uint64_t> part(nb + 1, 0);
#pragma omp parallel for num_threads(threads) schedule(static, 1)
        for (int b = 0; b < nb; b++) {
            uint64_t sum = 0;
            for (uint64_t i = std::min(n, b * blk), e = std::min(n, (b + 1) * blk); i < e; i++) sum += A[i];
            part[b + 1] = sum;
        }
        for (int b = 0; b < nb; b++) part[b + 1] += part[b];
        run = part[nb];
        // 2^32 entries and more (targets beyond ~4.4e9 residues; the reference's offsets are size_t, IndexTable.h:486): the
        // 32-bit slots then hold starts relative to a 64-bit base per 65 536 slots (a block's lists are far below 2^32)
        const bool wide = run > 0xFFFFFFFFull || getenv("SD_INDEX_WIDE") != nullptr;
        out.blockBase.clear();
        if (wide) out.blockBase.assign(((tableSize + 2) >> 16) + 1, 0);
#pragma omp parallel for num_threads(threads) schedule(static, 1)
        for (int b = 0; b < nb; b++) {
            uint64_t sum = part[b];
            for (uint64_t i = std::min(n, b * blk), e = std::min(n, (b + 1) * blk); i < e; i++) {
                if (wide && (i & 0xFFFFu) == 0) out.blockBase[i >> 16] = sum;   // everything before slot i
                sum += A[i];
                A[i] = wide ? (uint32_t) (sum - out.blockBase[i >> 16]) : (uint32_t) sum;
            }
        }
    }
    out.nEntries = run;
    out.entrySeq.assign(run, 0);
    out.entryPos.assign(run, 0);
    // fill in target order so every list comes out sorted by (seqId,pos) without a second sort:
    // targets are processed in blocks; a serial pass over per-block results keeps seqId ascending.
    uint32_t *cursor = A.data() + 1;
    const uint64_t *base = out.blockBase.empty() ? nullptr : out.blockBase.data();
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
                uint64_t c = cursor[r[i].kmer]++;
                if (base) c += base[((uint64_t) r[i].kmer + 1) >> 16];   // cursor slot = kmer + 1
                out.entrySeq[c] = id;
                out.entryPos[c] = r[i].pos;
            }
        }
    }
    A.shrink(tableSize + 1);   // A[kmer] = list start, A[tableSize] = number of entries
}

}  // namespace sd
