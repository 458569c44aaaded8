// On-disk database contract of the reference's modules, read and written by the `sdgpu` multi-call binary so that
// `prefilter`, `align`, `clusterhits` (and the glue modules between them) are drop-ins under the reference's
// unmodified workflow scripts (R/data/clustersearch.sh:110-152, M/data/workflow/blastp.sh:70,85):
//   name            entry payloads, each terminated by '\0' (optionally split into name.0 .. name.N-1, concatenated
//                   logically in numeric order: M/src/commons/FileUtil.cpp:330-346)
//   name.index      lines "key \t offset \t length" (length counts the '\0'; M/src/commons/DBWriter.cpp:483-495)
//   name.dbtype     little-endian int32 (M/src/commons/Parameters.h:68-94; bit 31 = compressed, extended bits in
//                   the high half, M/src/commons/DBReader.h:370-377)
//   name.lookup     "key \t accession \t setId" (M/src/commons/DBReader.h:58), name.source "setId \t file"
// Entry ids follow DBReader::open's access modes (M/src/commons/DBReader.cpp:265-420): the index is sorted by key,
// LINEAR_ACCESS numbers the entries by data offset, NOSORT keeps key order.
#ifndef SD_DB_H
#define SD_DB_H

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace sddb {

enum DbType {
    DBTYPE_AMINO_ACIDS = 0, DBTYPE_NUCLEOTIDES = 1, DBTYPE_HMM_PROFILE = 2, DBTYPE_ALIGNMENT_RES = 5,
    DBTYPE_CLUSTER_RES = 6, DBTYPE_PREFILTER_RES = 7, DBTYPE_INDEX_DB = 9, DBTYPE_GENERIC_DB = 12, DBTYPE_OMIT_FILE = 13
};
enum { EXT_COMPRESSED = 1, EXT_INDEX_NEED_SRC = 2, EXT_CONTEXT_PSEUDO_COUNTS = 4, EXT_GPU = 8, EXT_SET = 16 };
inline int baseType(int dbtype) { return dbtype & 0xFFFF; }                                   // DBTYPE_MASK
inline unsigned extendedType(int dbtype) { return ((uint32_t) dbtype >> 16) & 0x7FFE; }
inline int withExtended(int dbtype, unsigned ext) { return dbtype | ((ext & 0x7FFE) << 16); }
inline bool isCompressed(int dbtype) { return dbtype != -1 && ((uint32_t) dbtype & (1u << 31)) != 0; }

int readDbType(const std::string &dataName);   // -1 when there is no .dbtype file
bool fileExists(const std::string &path);

struct LookupEntry {
    uint32_t key;
    std::string name;
    uint32_t fileNumber;
};

class Reader {
public:
    enum Access { NOSORT = 0, LINEAR_ACCESS = 2 };
    enum Mode { USE_INDEX = 1, USE_DATA = 2, USE_LOOKUP = 4 };

    Reader() {}
    ~Reader() { close(); }
    Reader(const Reader &) = delete;
    Reader &operator=(const Reader &) = delete;

    // returns false and fills err on failure (nothing exits)
    bool open(const std::string &dataName, int mode, int access, std::string *err);
    void close();

    size_t size() const { return key_.size(); }
    int dbtype() const { return dbtype_; }
    uint32_t key(size_t id) const { return key_[id]; }
    // payload of entry id ('\0'-terminated inside the mapping); entryLength counts the terminator
    const char *data(size_t id) const;
    size_t entryLength(size_t id) const { return length_[id]; }
    uint64_t entryOffset(size_t id) const { return offset_[id]; }
    void copyData(char *dst) const;   // the data file(s) back to back (totalDataSize bytes)
    // DBReader::getSeqLen: sequence entries end in "\n\0" (M/src/commons/DBReader.h:225-231)
    size_t seqLen(size_t id) const { return length_[id] >= 2 ? length_[id] - 2 : 0; }
    size_t idOfKey(uint32_t key) const;                     // SIZE_MAX when absent (DBReader::getId -> UINT_MAX)
    uint64_t totalDataSize() const { return totalData_; }
    uint64_t aminoAcidDbSize() const;                       // DBReader::getAminoAcidDBSize: sum of seqLen
    size_t maxSeqLen() const;
    const std::vector<LookupEntry> &lookup() const { return lookup_; }   // sorted by key (USE_LOOKUP)
    std::vector<LookupEntry> takeLookup() { return std::move(lookup_); }   // the same list handed over (the reader keeps none)
    const std::string &name() const { return name_; }

private:
    struct Map {
        char *p = nullptr;
        size_t n = 0;
        uint64_t start = 0;
    };
    std::string name_;
    std::vector<Map> maps_;
    uint64_t totalData_ = 0;
    int dbtype_ = -1;
    // id order
    std::vector<uint32_t> key_;
    std::vector<uint64_t> offset_;
    std::vector<uint64_t> length_;
    // key order -> id
    std::vector<uint32_t> sortedKey_;
    std::vector<uint32_t> sortedId_;
    std::vector<LookupEntry> lookup_;
    bool hasData_ = false;
};

// Single-file writer: entries appended in call order, the index is written sorted by key on close (what
// DBWriter::close(merge = true) leaves behind, M/src/commons/DBWriter.cpp:232-243,531-650).
class Writer {
public:
    Writer() {}
    ~Writer();
    Writer(const Writer &) = delete;
    Writer &operator=(const Writer &) = delete;

    bool open(const std::string &dataName, int dbtype, std::string *err);
    // payload without terminator; a '\0' is appended (DBWriter::writeData, M/src/commons/DBWriter.cpp:331-399)
    bool write(uint32_t key, const char *data, size_t len);
    bool close(std::string *err);
    uint64_t entries() const { return index_.size(); }

private:
    struct Idx {
        uint32_t key;
        uint64_t offset;
        uint64_t length;
    };
    std::string name_;
    FILE *f_ = nullptr;
    std::vector<char> buf_;
    uint64_t offset_ = 0;
    int dbtype_ = -1;
    std::vector<Idx> index_;
};

// removes name, name.N, name.index, name.dbtype (DBReader::removeDb)
void removeDb(const std::string &dataName);

// "setId \t file" lines of name.source -> file per set key (convertalignments.cpp:121 readSetToSource keeps the basename)
bool readSources(const std::string &dataName, std::vector<std::pair<uint32_t, std::string> > &out, std::string *err);

}  // namespace sddb
#endif
