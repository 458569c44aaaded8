#include "sd_cli.h"

#include <chrono>
#include <sys/stat.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>

namespace sdcli {

static std::string g_module = "sdgpu";

int fail(const std::string &msg) {
    fprintf(stderr, "%s: %s\n", g_module.c_str(), msg.c_str());
    return 1;
}

int failCtx(sd_ctx *ctx, int rc, const char *what) {
    fprintf(stderr, "%s: %s failed (%d): %s\n", g_module.c_str(), what, rc, ctx ? sd_last_error(ctx) : "");
    return 1;
}

void info(const Args &a, const char *fmt, ...) {
    g_module = "sdgpu " + a.module;
    if (a.integer("-v", 3) < 3) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stdout, fmt, ap);
    va_end(ap);
    fflush(stdout);
}

static double lapNow() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
Lap::Lap(const char *m) : module(m), last(lapNow()), on(getenv("SD_DEBUG_TIMING") != nullptr) {}
void Lap::mark(const char *what) {
    if (!on) return;
    const double now = lapNow();
    fprintf(stderr, "[%s] %s %.3f s\n", module, what, now - last);
    last = now;
}

int threadsOf(const Args &a) {
    long long t = a.integer("--threads", 0);
    if (t <= 0) {
        // cgroup v2 quota (the GPU boxes expose 256 logical CPUs under a quota of 16)
        std::ifstream f("/sys/fs/cgroup/cpu.max");
        std::string q, p;
        if (f >> q >> p && q != "max") {
            const double quota = strtod(q.c_str(), nullptr), period = strtod(p.c_str(), nullptr);
            if (quota > 0 && period > 0) t = (long long) (quota / period + 0.5);
        }
        const unsigned hw = std::thread::hardware_concurrency();
        if (t <= 0 || (hw > 0 && t > (long long) hw)) t = hw > 0 ? hw : 1;
    }
    return (int) std::max<long long>(1, t);
}

Resident &resident() {
    static Resident r;
    return r;
}

sd_ctx *Resident::ctx(int device, int *rc) {
    auto it = ctxOfDevice.find(device);
    if (it != ctxOfDevice.end()) {
        if (rc) *rc = SD_OK;
        return it->second;
    }
    sd_ctx *c = nullptr;
    const int r = sd_ctx_create(device, &c);
    if (rc) *rc = r;
    if (r != SD_OK) return nullptr;
    ctxOfDevice[device] = c;
    return c;
}

sd_host *Resident::host(int threads) {
    auto it = hostOfThreads.find(threads);
    if (it != hostOfThreads.end()) return it->second;
    sd_host *h = nullptr;
    if (sd_host_create(threads, &h) != SD_OK) return nullptr;
    hostOfThreads[threads] = h;
    return h;
}

void Resident::clear() {
    for (auto &kv : hostOfThreads)
        if (kv.second) sd_host_destroy(kv.second);
    hostOfThreads.clear();
    for (auto &kv : seqSets)
        if (kv.second) sd_seqset_destroy(kv.second);
    seqSets.clear();
    for (auto &kv : targets)
        if (kv.second.t) sd_target_destroy(kv.second.t);
    targets.clear();
    for (auto &kv : ctxOfDevice)
        if (kv.second) sd_ctx_destroy(kv.second);
    ctxOfDevice.clear();
    seqDbs.clear();
    seqDbSig.clear();
    enabled = false;
}

std::shared_ptr<SeqDb> loadTargetDb(const std::string &path, sd_host *host, std::string *err) {
    Resident &R = resident();
    // a cached DB is only taken while its index file is the one that was read (size and mtime, as loadSetInfo does): a module of the
    // workflow may rewrite a DB under the same path
    struct stat st;
    const bool haveStat = ::stat((path + ".index").c_str(), &st) == 0;
    if (R.enabled) {
        auto it = R.seqDbs.find(path);
        if (it != R.seqDbs.end()) {
            const std::pair<long long, long long> &sig = R.seqDbSig[path];
            if (haveStat && sig.first == (long long) st.st_size && sig.second == (long long) st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec)
                return it->second;
            R.seqDbs.erase(it);
            for (auto ss = R.seqSets.begin(); ss != R.seqSets.end();)   // the device copy of the stale DB goes with it
                if (ss->first.compare(0, path.size() + 1, path + "|") == 0) {
                    if (ss->second) sd_seqset_destroy(ss->second);
                    ss = R.seqSets.erase(ss);
                } else {
                    ++ss;
                }
        }
    }
    std::shared_ptr<SeqDb> db(new SeqDb());
    if (!db->load(path, host, err)) return std::shared_ptr<SeqDb>();
    if (R.enabled && !db->profile && haveStat) {
        R.seqDbs[path] = db;
        R.seqDbSig[path] = std::make_pair((long long) st.st_size, (long long) st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec);
    }
    return db;
}

bool SeqDb::load(const std::string &path, sd_host *host, std::string *err) {
    if (!rd.open(path, sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, err)) return false;
    const int bt = sddb::baseType(rd.dbtype());
    if (rd.dbtype() == -1) {
        if (err) *err = path + " has no .dbtype file";
        return false;
    }
    if (bt != sddb::DBTYPE_AMINO_ACIDS && bt != sddb::DBTYPE_HMM_PROFILE) {
        if (err) *err = path + ": only amino acid and profile databases are on this path (dbtype " + std::to_string(bt) + ")";
        return false;
    }
    profile = bt == sddb::DBTYPE_HMM_PROFILE;
    n = (uint32_t) rd.size();
    keys.resize(n);
    lens.resize(n);
    offsets.assign((size_t) n + 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        keys[i] = rd.key(i);
        const size_t el = rd.entryLength(i);
        const size_t L = profile ? (std::max<size_t>(el, 1) - 1) / 25 : rd.seqLen(i);   // DBReader::getSeqLen (DBReader.h:224-231)
        lens[i] = (int32_t) L;
        offsets[i + 1] = offsets[i] + L;
    }
    const uint64_t total = offsets[n];
    residues.resize(total);
    if (!profile) {
        // one pass over the ASCII payloads (entries are contiguous in LINEAR_ACCESS order for DBs written by createdb)
        std::vector<char> ascii(total);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < n; i++) memcpy(ascii.data() + offsets[i], rd.data(i), (size_t) lens[i]);
        sd_host_map_sequence(host, ascii.data(), total, residues.data());
    } else {
        std::vector<uint64_t> byteOff((size_t) n + 1, 0);
        for (uint32_t i = 0; i < n; i++) byteOff[i + 1] = byteOff[i] + (uint64_t) lens[i] * 25;
        std::vector<char> raw(byteOff[n]);
#pragma omp parallel for schedule(static)
        for (uint32_t i = 0; i < n; i++) memcpy(raw.data() + byteOff[i], rd.data(i), (size_t) lens[i] * 25);
        consensus.resize(total);
        alnProfile.resize(total * 21);
        sortedScore.resize(total * 20);
        sortedIndex.resize(total * 20);
        std::vector<uint64_t> posOff((size_t) n + 1);
        const int rc = sd_host_map_profiles(raw.data(), byteOff.data(), n, residues.data(), consensus.data(), alnProfile.data(),
                                            sortedScore.data(), sortedIndex.data(), posOff.data());
        if (rc != SD_OK) {
            if (err) *err = "sd_host_map_profiles failed for " + path;
            return false;
        }
        // profiles beyond --max-seq-len positions were cut (Sequence::mapProfile): the position offsets are the mapper's
        for (uint32_t i = 0; i <= n; i++) offsets[i] = posOff[i];
        for (uint32_t i = 0; i < n; i++) lens[i] = (int32_t) (posOff[i + 1] - posOff[i]);
    }
    return true;
}

bool SetInfo::load(const std::string &dbPath, bool needSources, std::string *err) {
    sddb::Reader lk;
    if (!lk.open(dbPath, sddb::Reader::USE_LOOKUP, sddb::Reader::NOSORT, err)) return false;
    std::vector<sddb::LookupEntry> L = lk.takeLookup();
    uint32_t maxKey = 0, maxSet = 0;
    for (const sddb::LookupEntry &e : L) {
        maxKey = std::max(maxKey, e.key);
        maxSet = std::max(maxSet, e.fileNumber);
    }
    const size_t nk = L.empty() ? 0 : (size_t) maxKey + 1;
    setOfKey.assign(nk, 0);
    posOfKey.assign(nk, 0);
    strandOfKey.assign(nk, 0);
    nameOfKey.assign(nk, std::string());
    for (sddb::LookupEntry &e : L) {
        // name = accession_index_start_end (R/data/createsetdb.sh:128-133); fields from the end
        const std::string &s = e.name;
        size_t p3 = s.rfind('_');
        size_t p2 = p3 == std::string::npos || p3 == 0 ? std::string::npos : s.rfind('_', p3 - 1);
        size_t p1 = p2 == std::string::npos || p2 == 0 ? std::string::npos : s.rfind('_', p2 - 1);
        if (p1 == std::string::npos) {
            if (err) *err = "Invalid lookup record \"" + s + "\" in " + dbPath + ".lookup (expected accession_index_start_end)";
            return false;
        }
        const long long pos = strtoll(s.c_str() + p1 + 1, nullptr, 10);
        const long long st = strtoll(s.c_str() + p2 + 1, nullptr, 10);
        const long long en = strtoll(s.c_str() + p3 + 1, nullptr, 10);
        setOfKey[e.key] = e.fileNumber;
        posOfKey[e.key] = (uint32_t) pos;
        strandOfKey[e.key] = st < en ? 1 : 0;
        nameOfKey[e.key] = std::move(e.name);
    }
    sddb::Reader sz;
    if (!sz.open(dbPath + "_set_size", sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, err)) return false;
    uint32_t maxSetKey = maxSet;
    for (size_t i = 0; i < sz.size(); i++) maxSetKey = std::max(maxSetKey, sz.key(i));
    nSets = (L.empty() && sz.size() == 0) ? 0 : maxSetKey + 1;
    setSize.assign(nSets, 0);
    for (size_t i = 0; i < sz.size(); i++) setSize[sz.key(i)] = (uint32_t) strtoul(sz.data(i), nullptr, 10);
    if (needSources) {
        std::vector<std::pair<uint32_t, std::string> > src;
        if (!sddb::readSources(dbPath, src, err)) return false;
        sourceOfSet.assign(nSets, std::string());
        for (size_t i = 0; i < src.size(); i++)
            if (src[i].first < nSets) sourceOfSet[src[i].first] = src[i].second;
    }
    return true;
}

std::shared_ptr<const SetInfo> loadSetInfo(const std::string &dbPath, bool needSources, std::string *err) {
    struct Entry {
        std::shared_ptr<SetInfo> info;
        off_t size = 0;
        struct timespec mtime = {0, 0};
        bool sources = false;
    };
    static std::map<std::string, Entry> cache;
    struct stat st;
    const bool haveStat = ::stat((dbPath + ".lookup").c_str(), &st) == 0;
    auto it = cache.find(dbPath);
    if (it != cache.end() && haveStat && it->second.size == st.st_size && it->second.mtime.tv_sec == st.st_mtim.tv_sec &&
        it->second.mtime.tv_nsec == st.st_mtim.tv_nsec) {
        if (needSources && !it->second.sources) {
            SetInfo &si = *it->second.info;
            std::vector<std::pair<uint32_t, std::string> > src;
            if (!sddb::readSources(dbPath, src, err)) return std::shared_ptr<const SetInfo>();
            si.sourceOfSet.assign(si.nSets, std::string());
            for (size_t i = 0; i < src.size(); i++)
                if (src[i].first < si.nSets) si.sourceOfSet[src[i].first] = src[i].second;
            it->second.sources = true;
        }
        return it->second.info;
    }
    Entry e;
    e.info.reset(new SetInfo());
    if (!e.info->load(dbPath, needSources, err)) return std::shared_ptr<const SetInfo>();
    e.sources = needSources;
    if (haveStat) {
        e.size = st.st_size;
        e.mtime = st.st_mtim;
        cache[dbPath] = e;
    }
    return e.info;
}

}  // namespace sdcli
