// Persistent target index in the reference's `createindex` file layout: NAME.idx is a DB of dbtype
// DBTYPE_INDEX_DB whose entries are addressed by the keys of M/src/prefiltering/PrefilteringIndexReader.cpp:10-34 --
//   VERSION 0, META 1, SCOREMATRIXNAME 2, SCOREMATRIX2MER 3, SCOREMATRIX3MER 4, DBR1INDEX 5, DBR1DATA 6, DBR2INDEX 7, DBR2DATA 8,
//   ENTRIES 9, ENTRIESOFFSETS 10, ENTRIESNUM 12, SEQCOUNT 13, SEQINDEXDATA 14, SEQINDEXDATASIZE 15, SEQINDEXSEQOFFSET 16,
//   HDR1INDEX 18 .. HDR2DATA 21, GENERATOR 22, SPACEDPATTERN 23
// written by createIndexFile (:52-305): every entry page aligned, '\0' terminated; ENTRIES = packed 6-byte
// (uint32 seqId, uint16 position) records sorted inside a k-mer's list (IndexTable.h:25-39), ENTRIESOFFSETS = size_t[20^k + 1],
// SEQINDEXDATA = the masked numeric sequences back to back, SEQINDEXSEQOFFSET = size_t[n + 1], META = 12 ints.
// `sdgpu createindex` builds the index once (tantan masking + IndexBuilder::fillDatabase on the host); prefilter / search /
// clustersearch find TARGET.idx next to the target DB, check META against their parameters and upload it instead of
// rebuilding -- the serial prefix of every run (2.4 s at 100 proteomes, 19 s at 1 000) becomes a file read.
#include "sd_cli.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <memory>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace sdcli {

namespace {

enum { K_VERSION = 0, K_META = 1, K_SCOREMATRIXNAME = 2, K_SCOREMATRIX2MER = 3, K_SCOREMATRIX3MER = 4, K_DBR1INDEX = 5, K_DBR1DATA = 6,
       K_DBR2INDEX = 7, K_DBR2DATA = 8, K_ENTRIES = 9, K_ENTRIESOFFSETS = 10, K_ENTRIESNUM = 12, K_SEQCOUNT = 13, K_SEQINDEXDATA = 14,
       K_SEQINDEXDATASIZE = 15, K_SEQINDEXSEQOFFSET = 16, K_HDR1INDEX = 18, K_HDR1DATA = 19, K_HDR2INDEX = 20, K_HDR2DATA = 21,
       K_GENERATOR = 22, K_SPACEDPATTERN = 23 };
const char *INDEX_VERSION = "16";   // MMSEQS_CURRENT_INDEX_VERSION of the vendored MMseqs2 (read from an index the reference wrote)

struct HostH {
    sd_host *h = nullptr;
    ~HostH() { if (h) sd_host_destroy(h); }
};
struct IndexH {
    sd_host_index *ix = nullptr;
    ~IndexH() { if (ix) sd_host_index_destroy(ix); }
};

// page aligned entries into one data file; the index file lists (key, offset, length incl. terminator)
struct IdxWriter {
    FILE *f = nullptr;
    uint64_t off = 0;
    std::vector<std::pair<uint32_t, std::pair<uint64_t, uint64_t> > > index;
    bool put(uint32_t key, const void *data, uint64_t len) {
        const uint64_t start = off;
        if (len && fwrite(data, 1, len, f) != len) return false;
        fputc('\0', f);
        off += len + 1;
        index.push_back(std::make_pair(key, std::make_pair(start, len + 1)));
        return align();
    }
    void alias(uint32_t key, uint64_t start, uint64_t len) { index.push_back(std::make_pair(key, std::make_pair(start, len))); }
    bool align() {   // DBWriter::alignToPageSize
        static const char zeros[4096] = {0};
        const uint64_t pad = (4096 - off % 4096) % 4096;
        if (pad && fwrite(zeros, 1, pad, f) != pad) return false;
        off += pad;
        return true;
    }
};

// DBReader<unsigned int>::serialize (M/src/commons/DBReader.cpp:956-973): header + Index{id, offset, length} records of 24 bytes
std::vector<char> serializeDb(const sddb::Reader &rd, int dbtype) {
    const size_t n = rd.size();
    struct Rec {
        unsigned id;
        unsigned pad0;
        size_t offset;
        unsigned length;
        unsigned pad1;
    };
    static_assert(sizeof(Rec) == 24, "DBReader::Index layout");
    std::vector<char> out(2 * sizeof(size_t) + 3 * sizeof(unsigned) + n * sizeof(Rec), 0);
    char *p = out.data();
    const size_t size = n, dataSize = rd.totalDataSize();
    // NOSORT ids are key ordered: the order the reference's index array has after DBReader::open
    unsigned lastKey = 0, maxSeqLen = 0;
    for (size_t i = 0; i < n; i++) {
        lastKey = std::max(lastKey, rd.key(i));
        maxSeqLen = std::max<unsigned>(maxSeqLen, (unsigned) rd.entryLength(i));   // the longest entry, terminators included
    }
    memcpy(p, &size, sizeof(size_t)); p += sizeof(size_t);
    memcpy(p, &dataSize, sizeof(size_t)); p += sizeof(size_t);
    memcpy(p, &lastKey, sizeof(unsigned)); p += sizeof(unsigned);
    memcpy(p, &dbtype, sizeof(int)); p += sizeof(int);
    memcpy(p, &maxSeqLen, sizeof(unsigned)); p += sizeof(unsigned);
    Rec *r = (Rec *) p;
    for (size_t i = 0; i < n; i++) {
        r[i].id = rd.key(i);
        r[i].pad0 = 0;
        r[i].offset = rd.entryOffset(i);
        r[i].length = (unsigned) rd.entryLength(i);
        r[i].pad1 = 0;
    }
    return out;
}

// ScoreMatrix::serialize (M/src/commons/ScoreMatrix.h:26-41): rows padded to (size / 64 + 1) * 64 with score -255 / index 0
// (ExtendedSubstitutionMatrix.cpp:24-26,58-61), all scores (short) first, then all indices (unsigned int)
std::vector<char> serializeExt(const int16_t *score, const uint16_t *index, uint32_t size) {
    const size_t row = ((size_t) size / 64 + 1) * 64;
    std::vector<char> out((size_t) size * row * (sizeof(short) + sizeof(unsigned)));
    short *s = (short *) out.data();
    unsigned *ix = (unsigned *) (out.data() + (size_t) size * row * sizeof(short));
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < size; i++) {
        for (uint32_t z = 0; z < size; z++) {
            s[i * row + z] = score[(size_t) i * size + z];
            ix[i * row + z] = index[(size_t) i * size + z];
        }
        for (size_t z = size; z < row; z++) {
            s[i * row + z] = -255;
            ix[i * row + z] = 0;
        }
    }
    return out;
}

}  // namespace

int createindexModule(const Args &a) {
    if (a.pos.size() != 2) return fail("usage: createindex <sequenceDB> <tmpDir> [options]");
    if (a.integer("--spaced-kmer-mode", 1) != 1 || a.has("--spaced-kmer-pattern")) return fail("only the default spaced k-mer patterns are supported");
    if (a.integer("--split", 0) > 1) return fail("--split > 1 is not supported (the whole index is resident in HBM)");
    if (a.multi("--seed-sub-mat", "aa", "VTML80.out") != "VTML80.out") return fail("--seed-sub-mat: only VTML80.out is built in");
    const int threads = threadsOf(a);
    HostH host;
    if (sd_host_create(threads, &host.h) != SD_OK) return fail("sd_host_create failed");
    std::string err;
    SeqDb db;
    if (!db.load(a.pos[0], host.h, &err)) return fail(err);
    if (db.profile) return fail("profile databases are not indexed on this path");
    int k = (int) a.integer("-k", 0);
    if (k == 0) k = sd_host_auto_kmer_size(db.totalResidues());
    if (k != 6 && k != 7) return fail("-k: k-mer sizes 6 and 7 are implemented");
    // createindex defaults to the search sensitivity 7.5 (M/src/commons/Parameters.cpp); --k-score overrides
    const float sens = (float) a.real("-s", 7.5);
    const long long kScore = strtoll(a.multi("--k-score", "seq", "2147483647").c_str(), nullptr, 10);
    const int kmerThr = kScore != 2147483647LL ? (int) kScore : sd_host_kmer_threshold(sens, k);
    const int mask = a.integer("--mask", 1) != 0 ? 1 : 0;
    IndexH index;
    int rc = sd_host_index_build(host.h, db.residues.data(), db.offsets.data(), db.n, k, kmerThr, mask, a.real("--mask-prob", 0.9), &index.ix);
    if (rc != SD_OK) return fail("sd_host_index_build failed (" + std::to_string(rc) + ")");
    uint64_t tableSize = 0, nEntries = 0, maskedRes = 0;
    sd_host_index_info(index.ix, &tableSize, &nEntries, &maskedRes);
    const uint32_t *kOff, *eSeq;
    const uint16_t *ePos;
    const uint8_t *masked;
    sd_host_index_arrays(index.ix, &kOff, &eSeq, &ePos, &masked);
    const uint64_t *kBase = nullptr;   // wide index (>= 2^32 entries): offsets relative to a base per 65 536 k-mers
    sd_host_index_block_base(index.ix, &kBase, nullptr);

    const std::string out = a.pos[0] + ".idx";
    sddb::removeDb(out);
    IdxWriter w;
    w.f = fopen(out.c_str(), "wb");
    if (!w.f) return fail("cannot create " + out);
    std::vector<char> fbuf(16u << 20);
    setvbuf(w.f, fbuf.data(), _IOFBF, fbuf.size());
    bool ok = w.put(K_VERSION, INDEX_VERSION, strlen(INDEX_VERSION));
    sddb::Reader hdr;
    const bool haveHdr = sddb::fileExists(a.pos[0] + "_h.index") &&
                         hdr.open(a.pos[0] + "_h", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err);
    const int maxSeqLen = (int) a.integer("--max-seq-len", 65535);
    int meta[12] = {maxSeqLen, k, a.integer("--comp-bias-corr", 1) != 0 ? 1 : 0, 21, mask, 1, kmerThr, db.rd.dbtype(), db.rd.dbtype(),
                    haveHdr ? 1 : 0, 0, 1};
    ok = ok && w.put(K_META, meta, sizeof(meta));
    {
        std::vector<char> txt(1 << 16);
        const int n = sd_host_matrix_text(1, txt.data(), txt.size());
        std::string named = std::string("VTML80.out:") + std::string(txt.data(), n > 0 ? (size_t) n : 0);
        ok = ok && w.put(K_SCOREMATRIXNAME, named.data(), named.size());
    }
    ok = ok && w.put(K_SPACEDPATTERN, "", 0);   // createIndexFile writes the pattern exactly when it is empty (:98-102)
    ok = ok && w.put(K_GENERATOR, "sdgpu", 5);
    // the sequence DB itself (DBR1; DBR2 = the same entries, :124-127), then its headers
    sddb::Reader plain;
    if (!plain.open(a.pos[0], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    {
        std::vector<char> ser = serializeDb(plain, plain.dbtype());
        const uint64_t offIndex = w.off;
        ok = ok && w.put(K_DBR1INDEX, ser.data(), ser.size());
        const uint64_t offData = w.off;
        std::vector<char> data(plain.totalDataSize());
        plain.copyData(data.data());
        ok = ok && w.put(K_DBR1DATA, data.data(), data.size());
        w.alias(K_DBR2INDEX, offIndex, ser.size() + 1);
        w.alias(K_DBR2DATA, offData, data.size() + 1);
    }
    if (haveHdr) {
        std::vector<char> ser = serializeDb(hdr, hdr.dbtype());
        const uint64_t offIndex = w.off;
        ok = ok && w.put(K_HDR1INDEX, ser.data(), ser.size());
        const uint64_t offData = w.off;
        std::vector<char> data(hdr.totalDataSize());
        hdr.copyData(data.data());
        ok = ok && w.put(K_HDR1DATA, data.data(), data.size());
        w.alias(K_HDR2INDEX, offIndex, ser.size() + 1);
        w.alias(K_HDR2DATA, offData, data.size() + 1);
    }
    {
        const int16_t *s2, *s3;
        const uint16_t *i2, *i3;
        uint32_t z2, z3;
        sd_host_ext_matrix(host.h, 3, &s3, &i3, &z3);
        sd_host_ext_matrix(host.h, 2, &s2, &i2, &z2);
        std::vector<char> m3 = serializeExt(s3, i3, z3);
        ok = ok && w.put(K_SCOREMATRIX3MER, m3.data(), m3.size());
        std::vector<char> m2 = serializeExt(s2, i2, z2);
        ok = ok && w.put(K_SCOREMATRIX2MER, m2.data(), m2.size());
    }
    {   // ENTRIES: packed (uint32, uint16)
        std::vector<char> ent(nEntries * 6 + 1);
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i < nEntries; i++) {
            memcpy(ent.data() + i * 6, &eSeq[i], 4);
            memcpy(ent.data() + i * 6 + 4, &ePos[i], 2);
        }
        ok = ok && w.put(K_ENTRIES, ent.data(), nEntries * 6);
    }
    {
        std::vector<size_t> off64(tableSize + 1);   // the file carries absolute size_t offsets (IndexTable.h:486)
#pragma omp parallel for schedule(static)
        for (uint64_t i = 0; i <= tableSize; i++) off64[i] = (kBase ? kBase[i >> 16] : 0) + kOff[i];
        ok = ok && w.put(K_ENTRIESOFFSETS, off64.data(), off64.size() * sizeof(size_t));
    }
    ok = ok && w.put(K_ENTRIESNUM, &nEntries, sizeof(uint64_t));
    const size_t seqCount = db.n;
    ok = ok && w.put(K_SEQCOUNT, &seqCount, sizeof(size_t));
    const int64_t dataSize = (int64_t) db.totalResidues();
    ok = ok && w.put(K_SEQINDEXDATASIZE, &dataSize, sizeof(int64_t));
    {
        std::vector<size_t> so(db.offsets.begin(), db.offsets.end());
        ok = ok && w.put(K_SEQINDEXSEQOFFSET, so.data(), so.size() * sizeof(size_t));
    }
    {   // getDataSize() + 1 bytes (:288)
        std::vector<char> sd(db.totalResidues() + 1, 0);
        memcpy(sd.data(), masked, db.totalResidues());
        ok = ok && w.put(K_SEQINDEXDATA, sd.data(), sd.size());
    }
    ok = ok && fclose(w.f) == 0;
    if (!ok) return fail("cannot write " + out);
    std::sort(w.index.begin(), w.index.end());
    FILE *ix = fopen((out + ".index").c_str(), "w");
    if (!ix) return fail("cannot create " + out + ".index");
    for (size_t i = 0; i < w.index.size(); i++)
        fprintf(ix, "%u\t%llu\t%llu\n", w.index[i].first, (unsigned long long) w.index[i].second.first, (unsigned long long) w.index[i].second.second);
    fclose(ix);
    FILE *t = fopen((out + ".dbtype").c_str(), "wb");
    const int32_t dbt = sddb::DBTYPE_INDEX_DB;
    if (t) {
        fwrite(&dbt, 4, 1, t);
        fclose(t);
    }
    info(a, "Index %s: k = %d, k-mer threshold %d, %llu entries, %llu masked residues\n", out.c_str(), k, kmerThr,
         (unsigned long long) nEntries, (unsigned long long) maskedRes);
    return 0;
}

// TARGET.idx next to the target DB -> the arrays sd_target_create / sd_search_create take.  Returns 0 when loaded, 1 when
// there is no usable index file (the caller builds the index), -1 on a broken file.
int loadTargetIndex(const std::string &targetDb, int wantK, int wantKmerThr, int wantMask, uint64_t nSeq, uint64_t residues,
                    LoadedIndex &out, std::string *why) {
    const std::string path = targetDb + ".idx";
    if (!sddb::fileExists(path + ".index")) {
        if (why) *why = "no " + path;
        return 1;
    }
    std::string err;
    std::unique_ptr<sddb::Reader> rd(new sddb::Reader());
    if (!rd->open(path, sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) {
        if (why) *why = err;
        return -1;
    }
    auto get = [&](uint32_t key, size_t *len) -> const char * {
        const size_t id = rd->idOfKey(key);
        if (id == SIZE_MAX) return nullptr;
        if (len) *len = rd->entryLength(id) - 1;
        return rd->data(id);
    };
    size_t len = 0;
    const char *v = get(K_VERSION, &len);
    if (!v || strncmp(v, INDEX_VERSION, strlen(INDEX_VERSION)) != 0) {
        if (why) *why = path + ": outdated index version";
        return 1;
    }
    const char *m = get(K_META, &len);
    if (!m || len < 48) return -1;
    int meta[12];
    memcpy(meta, m, sizeof(meta));
    if (meta[1] != wantK || meta[6] != wantKmerThr || meta[4] != wantMask || meta[5] != 1 || meta[3] != 21 || meta[11] != 1) {
        if (why)
            *why = path + " was created with k " + std::to_string(meta[1]) + " / k-mer threshold " + std::to_string(meta[6]) + " / mask " +
                   std::to_string(meta[4]) + ", this run needs " + std::to_string(wantK) + " / " + std::to_string(wantKmerThr) + " / " +
                   std::to_string(wantMask);
        return 1;
    }
    size_t lenE = 0, lenO = 0, lenS = 0, lenN = 0;
    const char *ent = get(K_ENTRIES, &lenE), *off = get(K_ENTRIESOFFSETS, &lenO), *seq = get(K_SEQINDEXDATA, &lenS), *num = get(K_ENTRIESNUM, &lenN);
    const char *cnt = get(K_SEQCOUNT, nullptr);
    if (!ent || !off || !seq || !num || !cnt) return -1;
    uint64_t nEntries = 0;
    size_t seqCount = 0;
    memcpy(&nEntries, num, 8);
    memcpy(&seqCount, cnt, sizeof(size_t));
    uint64_t tableSize = 1;
    for (int i = 0; i < wantK; i++) tableSize *= 20;
    if (seqCount != nSeq || lenS < residues || lenE < nEntries * 6 || lenO < (tableSize + 1) * sizeof(size_t)) {
        if (why) *why = path + " does not belong to this target DB (sequence count / size differ)";
        return 1;
    }
    out.k = wantK;
    out.kmerThr = wantKmerThr;
    out.nEntries = nEntries;
    out.offsets.resize(tableSize + 1);
    out.entrySeq.resize(std::max<uint64_t>(nEntries, 1));
    out.entryPos.resize(std::max<uint64_t>(nEntries, 1));
    out.masked.resize(std::max<uint64_t>(residues, 1));
    const size_t *o64 = (const size_t *) off;
    out.blockBase.clear();
    if (nEntries >= (1ull << 32) || getenv("SD_INDEX_WIDE")) {   // wide: 32-bit slots relative to a base per 65 536 k-mers
        out.blockBase.assign(((tableSize + 2) >> 16) + 1, 0);
        for (uint64_t b = 0; (b << 16) <= tableSize; b++) out.blockBase[b] = o64[b << 16];
    }
    const uint64_t *bb = out.blockBase.empty() ? nullptr : out.blockBase.data();
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i <= tableSize; i++) out.offsets[i] = (uint32_t) (o64[i] - (bb ? bb[i >> 16] : 0));
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < nEntries; i++) {
        memcpy(&out.entrySeq[i], ent + i * 6, 4);
        memcpy(&out.entryPos[i], ent + i * 6 + 4, 2);
    }
    memcpy(out.masked.data(), seq, residues);
    return 0;
}

}  // namespace sdcli
