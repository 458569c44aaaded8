// Shared pieces of the `sdgpu` multi-call binary: the host driver of the MI355X hot path behind the reference's own
// module command lines and DB files (R/src/spacedust.cpp:105-117 registers the modules; R/data/clustersearch.sh:110-152 and
// M/data/workflow/blastp.sh:70,85 are the callers).  Everything that computes goes through the C ABI of libsdgpu.so
// (include/spacedust_gpu.h); this side only reads / writes DBs and sequences the calls.
#ifndef SD_CLI_H
#define SD_CLI_H

#include "sd_args.h"
#include "sd_db.h"
#include "spacedust_gpu.h"

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace sdcli {

int fail(const std::string &msg);                       // prints "sdgpu <module>: msg", returns 1
int failCtx(sd_ctx *ctx, int rc, const char *what);     // with sd_last_error
void info(const Args &a, const char *fmt, ...);         // honours -v (>= 3 prints)
int threadsOf(const Args &a);                           // --threads, default: cgroup quota / hardware threads

// SD_DEBUG_TIMING=1: wall time between the marks of a module, to stderr
struct Lap {
    const char *module;
    double last;
    bool on;
    explicit Lap(const char *module);
    void mark(const char *what);
};

// A sequence or profile DB in the layout the device wants: ids follow DBReader::LINEAR_ACCESS order
// (Prefiltering.cpp:178, Alignment.cpp:75-90), residues are Sequence::mapSequence's numeric alphabet
// (M/src/commons/Sequence.cpp:307-324).
struct SeqDb {
    sddb::Reader rd;
    uint32_t n = 0;
    bool profile = false;
    std::vector<uint8_t> residues;      // numeric residues (profiles: the query letters)
    std::vector<uint64_t> offsets;      // n + 1
    std::vector<uint32_t> keys;         // DB key of id
    std::vector<int32_t> lens;
    // profile DBs (DBTYPE_HMM_PROFILE): Sequence::mapProfile's arrays (sd_host_map_profiles)
    std::vector<uint8_t> consensus;
    std::vector<int8_t> alnProfile;     // total x 21
    std::vector<int16_t> sortedScore;   // total x 20
    std::vector<uint8_t> sortedIndex;   // total x 20
    bool load(const std::string &path, sd_host *host, std::string *err);
    uint64_t totalResidues() const { return offsets.empty() ? 0 : offsets.back(); }
};

// set membership of a createsetdb DB (R/data/createsetdb.sh:119-180): from NAME.lookup (key, acc_idx_start_end, set) and
// NAME_set_size
struct SetInfo {
    std::vector<uint32_t> setOfKey;     // by DB key (keys are dense: the reference indexes its lookup array by key,
                                        // R/src/util/ClusterHits.cpp:328-329)
    std::vector<uint32_t> posOfKey;     // gene index inside its set (third field from the end of the lookup name)
    std::vector<uint8_t> strandOfKey;   // start < end
    std::vector<std::string> nameOfKey;
    std::vector<uint32_t> setSize;      // by set key, from NAME_set_size
    std::vector<std::string> sourceOfSet;
    uint32_t nSets = 0;
    bool load(const std::string &dbPath, bool needSources, std::string *err);
};

// SetInfo::load through a cache of this process (a workflow's clusterhits and summarizeresults read the same two set DBs: three
// million lookup lines each at 1 000 target proteomes); an entry is reused while NAME.lookup keeps its size and modification time
std::shared_ptr<const SetInfo> loadSetInfo(const std::string &dbPath, bool needSources, std::string *err);

// a target index read from TARGET.idx (sd_mod_index.cpp), in the array form sd_target_create / sd_search_create take
struct LoadedIndex {
    int k = 0, kmerThr = 0;
    uint64_t nEntries = 0;
    std::vector<uint32_t> offsets, entrySeq;
    std::vector<uint64_t> blockBase;    // non-empty: a wide index (>= 2^32 entries), offsets relative to blockBase[i >> 16]
    std::vector<uint16_t> entryPos;
    std::vector<uint8_t> masked;
};
int loadTargetIndex(const std::string &targetDb, int wantK, int wantKmerThr, int wantMask, uint64_t nSeq, uint64_t residues,
                    LoadedIndex &out, std::string *why);

// What a workflow of several modules over ONE target keeps between them (iterativeSearch: prefilter / align / result2profile
// of three iterations): the loaded target DB, one context per device, the target index built on it and the target sequence
// set.  The reference's modules are separate processes that reload everything; in this binary the workflow runs them in
// process, and at 1 000 target proteomes the reloads were ~40 % of `search --num-iterations 3`.  Off unless a workflow
// switches it on; a module that runs alone behaves as before.
struct Resident {
    bool enabled = false;
    std::map<std::string, std::shared_ptr<SeqDb> > seqDbs;          // DB path -> loaded sequences
    std::map<std::string, std::pair<long long, long long> > seqDbSig; // DB path -> (size, mtime in ns) of its .index when it was loaded
    std::map<int, sd_ctx *> ctxOfDevice;
    struct TargetEntry {
        sd_target *t = nullptr;
        uint64_t nEntries = 0, masked = 0;
    };
    std::map<std::string, TargetEntry> targets;                      // "path|k|threshold|mask|prob|device"
    std::map<std::string, sd_seqset *> seqSets;                      // "path|device"
    sd_ctx *ctx(int device, int *rc);                                // the device's context (created on first use)
    std::map<int, sd_host *> hostOfThreads;                          // the host objects (matrices, extended k-mer tables) by thread count
    sd_host *host(int threads);                                      // created on first use; NULL on failure
    void clear();                                                    // destroys everything (sequence sets and targets before their contexts)
};
Resident &resident();
// the target DB of a module: from the resident cache when it is on, else loaded for this call
std::shared_ptr<SeqDb> loadTargetDb(const std::string &path, sd_host *host, std::string *err);

// modules (each: argv after the module name -> exit code)
int createindexModule(const Args &a);
int prefilterModule(const Args &a);
int alignModule(const Args &a);
int clusterhitsModule(const Args &a);
int prefixidModule(const Args &a);
int besthitbysetModule(const Args &a);
int mergeresultsbysetModule(const Args &a);
int combinehitsModule(const Args &a);
int summarizeresultsModule(const Args &a);
int createsetdbModule(const Args &a);
int searchModule(const Args &a);
int clustersearchModule(const Args &a);
int result2profileModule(const Args &a);
int subtractdbsModule(const Args &a);
int mergedbsModule(const Args &a);

}  // namespace sdcli
#endif
