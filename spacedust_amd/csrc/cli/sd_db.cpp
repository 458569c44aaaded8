// DB files of the reference's modules (see sd_db.h).  Restated from the format, not from the reader's code:
// mmap the data file(s), parse the text index with a hand-rolled integer scanner, order ids the way
// DBReader::open does (M/src/commons/DBReader.cpp:100-214,265-420).
#include "sd_db.h"

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <numeric>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace sddb {

bool fileExists(const std::string &path) {
    struct stat st;
    return stat(path.c_str(), &st) == 0;
}

int readDbType(const std::string &dataName) {
    FILE *f = fopen((dataName + ".dbtype").c_str(), "rb");
    if (!f) return -1;
    int32_t v = -1;
    const size_t n = fread(&v, sizeof(v), 1, f);
    fclose(f);
    return n == 1 ? (int) v : -1;
}

namespace {

struct FileMap {
    char *p = nullptr;
    size_t n = 0;
    bool map(const std::string &path, std::string *err) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) {
            if (err) *err = "cannot open " + path + ": " + strerror(errno);
            return false;
        }
        struct stat st;
        if (fstat(fd, &st) != 0) {
            ::close(fd);
            if (err) *err = "cannot stat " + path;
            return false;
        }
        n = (size_t) st.st_size;
        if (n > 0) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                ::close(fd);
                if (err) *err = "cannot mmap " + path;
                return false;
            }
            p = (char *) m;
        }
        ::close(fd);
        return true;
    }
    void unmap() {
        if (p) munmap(p, n);
        p = nullptr;
        n = 0;
    }
};

inline const char *scanU64(const char *c, const char *end, uint64_t &v) {
    v = 0;
    while (c < end && *c >= '0' && *c <= '9') v = v * 10 + (uint64_t) (*c++ - '0');
    return c;
}
inline const char *skipBlank(const char *c, const char *end) {
    while (c < end && (*c == '\t' || *c == ' ')) c++;
    return c;
}
inline const char *nextLine(const char *c, const char *end) {
    while (c < end && *c != '\n') c++;
    return c < end ? c + 1 : c;
}

}  // namespace

bool Reader::open(const std::string &dataName, int mode, int access, std::string *err) {
    close();
    name_ = dataName;
    dbtype_ = readDbType(dataName);
    if (isCompressed(dbtype_)) {
        if (err) *err = dataName + " is a compressed database (--compressed 1); not supported";
        return false;
    }
    if (mode & USE_DATA) {
        std::vector<std::string> files;
        for (size_t i = 0;; i++) {
            const std::string f = dataName + "." + std::to_string(i);
            if (!fileExists(f)) break;
            files.push_back(f);
        }
        if (files.empty() && fileExists(dataName)) files.push_back(dataName);
        if (files.empty()) {
            if (err) *err = "no data file found for " + dataName;
            return false;
        }
        for (const std::string &f : files) {
            FileMap fm;
            if (!fm.map(f, err)) return false;
            Map m;
            m.p = fm.p;
            m.n = fm.n;
            m.start = totalData_;
            totalData_ += fm.n;
            maps_.push_back(m);
        }
        hasData_ = true;
    }
    if (mode & USE_INDEX) {
        FileMap ix;
        if (!ix.map(dataName + ".index", err)) return false;
        const char *c = ix.p, *end = ix.p + ix.n;
        std::vector<uint32_t> k;
        std::vector<uint64_t> o, l;
        while (c < end) {
            if (*c == '\n') {
                c++;
                continue;
            }
            uint64_t key, off, len;
            c = scanU64(c, end, key);
            c = skipBlank(c, end);
            c = scanU64(c, end, off);
            c = skipBlank(c, end);
            c = scanU64(c, end, len);
            c = nextLine(c, end);
            k.push_back((uint32_t) key);
            o.push_back(off);
            l.push_back(len);
        }
        ix.unmap();
        const size_t n = k.size();
        // key order first (DBReader sorts the index by id unless it already is)
        std::vector<uint32_t> byKey(n);
        std::iota(byKey.begin(), byKey.end(), 0u);
        if (!std::is_sorted(k.begin(), k.end()))
            std::stable_sort(byKey.begin(), byKey.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
        // local ids: key order, or data-offset order of the key-sorted entries for LINEAR_ACCESS
        std::vector<uint32_t> order(byKey);
        if (access == LINEAR_ACCESS) {
            bool sorted = true;
            for (size_t i = 1; i < n && sorted; i++) sorted = o[order[i - 1]] <= o[order[i]];
            if (!sorted) std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return o[a] < o[b]; });
        }
        key_.resize(n);
        offset_.resize(n);
        length_.resize(n);
        std::vector<uint32_t> idOfLine(n);
        for (size_t i = 0; i < n; i++) {
            key_[i] = k[order[i]];
            offset_[i] = o[order[i]];
            length_[i] = l[order[i]];
            idOfLine[order[i]] = (uint32_t) i;
        }
        sortedKey_.resize(n);
        sortedId_.resize(n);
        for (size_t i = 0; i < n; i++) {
            sortedKey_[i] = k[byKey[i]];
            sortedId_[i] = idOfLine[byKey[i]];
        }
        if (hasData_) {
            for (size_t i = 0; i < n; i++) {
                if (offset_[i] + length_[i] > totalData_) {
                    if (err) *err = dataName + ".index points beyond the data file (entry key " + std::to_string(key_[i]) + ")";
                    return false;
                }
            }
        }
    }
    if (mode & USE_LOOKUP) {
        FileMap lk;
        if (!lk.map(dataName + ".lookup", err)) return false;
        const char *c = lk.p, *end = lk.p + lk.n;
        while (c < end) {
            if (*c == '\n') {
                c++;
                continue;
            }
            LookupEntry e;
            uint64_t key, fileNo = 0;
            c = scanU64(c, end, key);
            c = skipBlank(c, end);
            const char *s = c;
            while (c < end && *c != '\t' && *c != '\n' && *c != ' ') c++;
            e.name.assign(s, c - s);
            c = skipBlank(c, end);
            c = scanU64(c, end, fileNo);
            c = nextLine(c, end);
            e.key = (uint32_t) key;
            e.fileNumber = (uint32_t) fileNo;
            lookup_.push_back(std::move(e));
        }
        lk.unmap();
        if (!std::is_sorted(lookup_.begin(), lookup_.end(), [](const LookupEntry &a, const LookupEntry &b) { return a.key < b.key; }))
            std::stable_sort(lookup_.begin(), lookup_.end(), [](const LookupEntry &a, const LookupEntry &b) { return a.key < b.key; });
    }
    return true;
}

void Reader::close() {
    for (Map &m : maps_)
        if (m.p) munmap(m.p, m.n);
    maps_.clear();
    key_.clear();
    offset_.clear();
    length_.clear();
    sortedKey_.clear();
    sortedId_.clear();
    lookup_.clear();
    totalData_ = 0;
    hasData_ = false;
    dbtype_ = -1;
}

const char *Reader::data(size_t id) const {
    const uint64_t off = offset_[id];
    // the file holding this offset (entries never straddle files)
    size_t f = 0;
    if (maps_.size() > 1) {
        size_t lo = 0, hi = maps_.size();
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (maps_[mid].start <= off) lo = mid;
            else hi = mid;
        }
        f = lo;
    }
    return maps_[f].p + (off - maps_[f].start);
}

void Reader::copyData(char *dst) const {
    for (const Map &m : maps_) memcpy(dst + m.start, m.p, m.n);
}

size_t Reader::idOfKey(uint32_t key) const {
    if (key < sortedKey_.size() && sortedKey_[key] == key) return sortedId_[key];   // dense keys 0 .. n-1: no search
    const std::vector<uint32_t>::const_iterator it = std::lower_bound(sortedKey_.begin(), sortedKey_.end(), key);
    if (it == sortedKey_.end() || *it != key) return SIZE_MAX;
    return sortedId_[it - sortedKey_.begin()];
}

uint64_t Reader::aminoAcidDbSize() const {
    uint64_t s = 0;
    for (size_t i = 0; i < length_.size(); i++) s += seqLen(i);
    return s;
}

size_t Reader::maxSeqLen() const {
    size_t m = 0;
    for (size_t i = 0; i < length_.size(); i++) m = std::max(m, seqLen(i));
    return m;
}

Writer::~Writer() {
    if (f_) fclose(f_);
}

bool Writer::open(const std::string &dataName, int dbtype, std::string *err) {
    name_ = dataName;
    dbtype_ = dbtype;
    removeDb(dataName);
    f_ = fopen(dataName.c_str(), "wb");
    if (!f_) {
        if (err) *err = "cannot create " + dataName + ": " + strerror(errno);
        return false;
    }
    buf_.resize(8u << 20);
    setvbuf(f_, buf_.data(), _IOFBF, buf_.size());
    offset_ = 0;
    index_.clear();
    return true;
}

bool Writer::write(uint32_t key, const char *data, size_t len) {
    if (!f_) return false;
    if (len && fwrite(data, 1, len, f_) != len) return false;
    if (fputc('\0', f_) == EOF) return false;
    Idx e;
    e.key = key;
    e.offset = offset_;
    e.length = len + 1;
    index_.push_back(e);
    offset_ += len + 1;
    return true;
}

bool Writer::close(std::string *err) {
    if (!f_) return true;
    const bool okData = fclose(f_) == 0;
    f_ = nullptr;
    if (!okData) {
        if (err) *err = "cannot write " + name_;
        return false;
    }
    if (!std::is_sorted(index_.begin(), index_.end(), [](const Idx &a, const Idx &b) { return a.key < b.key; }))
        std::stable_sort(index_.begin(), index_.end(), [](const Idx &a, const Idx &b) { return a.key < b.key; });
    FILE *ix = fopen((name_ + ".index").c_str(), "wb");
    if (!ix) {
        if (err) *err = "cannot create " + name_ + ".index";
        return false;
    }
    std::vector<char> ibuf(4u << 20);
    setvbuf(ix, ibuf.data(), _IOFBF, ibuf.size());
    for (const Idx &e : index_) fprintf(ix, "%u\t%llu\t%llu\n", e.key, (unsigned long long) e.offset, (unsigned long long) e.length);
    const bool okIdx = fclose(ix) == 0;
    if (dbtype_ != DBTYPE_OMIT_FILE) {
        FILE *t = fopen((name_ + ".dbtype").c_str(), "wb");
        if (!t) {
            if (err) *err = "cannot create " + name_ + ".dbtype";
            return false;
        }
        const int32_t v = (int32_t) ((uint32_t) dbtype_ & ~(1u << 31));
        fwrite(&v, sizeof(v), 1, t);
        fclose(t);
    }
    if (!okIdx && err) *err = "cannot write " + name_ + ".index";
    return okIdx;
}

void removeDb(const std::string &dataName) {
    ::remove(dataName.c_str());
    for (size_t i = 0;; i++) {
        const std::string f = dataName + "." + std::to_string(i);
        if (!fileExists(f)) break;
        ::remove(f.c_str());
    }
    ::remove((dataName + ".index").c_str());
    ::remove((dataName + ".dbtype").c_str());
}

bool readSources(const std::string &dataName, std::vector<std::pair<uint32_t, std::string> > &out, std::string *err) {
    FileMap fm;
    if (!fm.map(dataName + ".source", err)) return false;
    const char *c = fm.p, *end = fm.p + fm.n;
    while (c < end && *c != '\0') {
        if (*c == '\n') {
            c++;
            continue;
        }
        uint64_t key;
        c = scanU64(c, end, key);
        c = skipBlank(c, end);
        const char *s = c;
        while (c < end && *c != '\n') c++;
        out.push_back(std::make_pair((uint32_t) key, std::string(s, c - s)));
        c = nextLine(c, end);
    }
    fm.unmap();
    return true;
}

}  // namespace sddb
