// Profile queries, host side: the parse of a profile DB entry (Sequence::mapProfile, M/src/commons/Sequence.cpp:241-292),
// the similar-k-mer enumeration over per-position score rows (KmerGenerator with setDivideStrategy(ScoreMatrix**),
// M/src/prefiltering/KmerGenerator.cpp:30-39,107-216) and the profile k-mer threshold (Prefiltering.cpp:1019-1043).
#include "sd_host.h"

#include <algorithm>
#include <cstring>

namespace sd {

// Util::rankedDescSort20 (M/src/commons/Util.cpp:88-114): a fixed 20-input sorting network, descending; ties end up
// wherever the network leaves them, so the comparator sequence is part of the semantics.
static void rankedDescSort20(int16_t *val, uint8_t *index) {
    static const uint8_t net[][2] = {
        {0, 16}, {1, 17}, {2, 18}, {3, 19}, {4, 12}, {5, 13}, {6, 14}, {7, 15},
        {0, 8}, {1, 9}, {2, 10}, {3, 11},
        {8, 16}, {9, 17}, {10, 18}, {11, 19}, {0, 4}, {1, 5}, {2, 6}, {3, 7},
        {8, 12}, {9, 13}, {10, 14}, {11, 15}, {4, 16}, {5, 17}, {6, 18}, {7, 19}, {0, 2}, {1, 3},
        {4, 8}, {5, 9}, {6, 10}, {7, 11}, {12, 16}, {13, 17}, {14, 18}, {15, 19}, {0, 1},
        {4, 6}, {5, 7}, {8, 10}, {9, 11}, {12, 14}, {13, 15}, {16, 18}, {17, 19},
        {2, 16}, {3, 17}, {6, 12}, {7, 13}, {18, 19},
        {2, 8}, {3, 9}, {10, 16}, {11, 17},
        {2, 4}, {3, 5}, {6, 8}, {7, 9}, {10, 12}, {11, 13}, {14, 16}, {15, 17},
        {2, 3}, {4, 5}, {6, 7}, {8, 9}, {10, 11}, {12, 13}, {14, 15}, {16, 17},
        {1, 16}, {3, 18}, {5, 12}, {7, 14},
        {1, 8}, {3, 10}, {9, 16}, {11, 18},
        {1, 4}, {3, 6}, {5, 8}, {7, 10}, {9, 12}, {11, 14}, {13, 16}, {15, 18},
        {1, 2}, {3, 4}, {5, 6}, {7, 8}, {9, 10}, {11, 12}, {13, 14}, {15, 16}, {17, 18}};
    for (size_t c = 0; c < sizeof(net) / sizeof(net[0]); c++) {
        const int x = net[c][0], y = net[c][1];
        if (val[x] < val[y]) {
            std::swap(val[x], val[y]);
            std::swap(index[x], index[y]);
        }
    }
}

void mapProfile(const char *data, uint32_t L, uint8_t *letters, uint8_t *consensus, int8_t *aln, int16_t *sortedScore,
                uint8_t *sortedIndex) {
    for (uint32_t l = 0; l < L; l++) {
        const char *rec = data + (size_t) l * PROFILE_RECORD;
        int16_t *sc = sortedScore + (size_t) l * 20;
        uint8_t *ix = sortedIndex + (size_t) l * 20;
        int8_t *row = aln + (size_t) l * ALPH;
        for (int a = 0; a < 20; a++) {
            const short s = (short) rec[a];       // signed char scores (:248-249)
            sc[a] = s;
            ix[a] = (uint8_t) a;
            row[a] = (int8_t) (s / 4);            // alignment profile, truncating division (:273-275)
        }
        row[20] = 0;                              // X scores 0 (:278-280)
        letters[l] = (uint8_t) rec[20];
        if (consensus) consensus[l] = (uint8_t) rec[21];
        rankedDescSort20(sc, ix);                 // k-mer generator rows (:284-290)
    }
}

size_t generateProfileKmerList(const int16_t *const *score, const uint8_t *const *index, int k, int thr,
                               std::vector<uint32_t> &out) {
    // divide strategy 1+1+...+1: list s is the sorted row of seed position s, multiplier 20^s; the window handed to
    // the generator is all zeros, so every "row of the k-mer's own sub-word" is row 0 = the position's row itself
    short best[8], rest[8];
    uint64_t mult[8];
    uint64_t pw = 1;
    for (int s = 0; s < k; s++) {
        best[s] = score[s][0];
        mult[s] = pw;
        pw *= 20;
    }
    rest[k - 1] = 0;
    for (int s = k - 1; s >= 1; s--) rest[s - 1] = (short) (best[s] + rest[s]);
    const short threshold = (short) thr;
    const short cutoff1 = (short) (threshold - rest[0]);
    std::vector<short> sA, sB;
    std::vector<uint32_t> iA, iB;
    for (int pos = 0; pos < 20 && score[0][pos] >= cutoff1; pos++) {   // rowSize is 32 but entries 20..31 hold -SHRT_MAX
        sA.push_back(score[0][pos]);
        iA.push_back(index[0][pos]);
    }
    for (int s = 0; s < k - 1; s++) {
        const int16_t *sc = score[s + 1];
        const uint8_t *ix = index[s + 1];
        sB.clear();
        iB.clear();
        for (size_t i = 0; i < sA.size(); i++) {
            const short si = sA[i];
            const short cutoff2 = (short) (threshold - si - rest[s + 1]);
            for (int j = 0; j < 20 && sc[j] >= cutoff2; j++) {
                sB.push_back((short) (si + sc[j]));
                iB.push_back((uint32_t) (iA[i] + (uint64_t) ix[j] * mult[s + 1]));
            }
        }
        sA.swap(sB);
        iA.swap(iB);
    }
    out.swap(iA);
    return out.size();
}

int profileKmerThreshold(float sensitivity, int k) {
    float best;
    if (k == 5) {
        float base = 108.8;
        best = base - (sensitivity * 4.7);
    } else if (k == 6) {
        float base = 134.35;
        best = base - (sensitivity * 6.15);
    } else {
        float base = 149.15;
        best = base - (sensitivity * 6.85);
    }
    return static_cast<int>(best);
}

}  // namespace sd
