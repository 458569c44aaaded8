// The workflow modules: what the reference runs as shell scripts over several processes, here one process around the
// C++ pipeline object of libsdgpu.so (sd_search_*):
//   search         <queryDB> <targetDB> <alignmentDB> <tmpDir>   = prefilter + align (M/data/workflow/blastp.sh:59-90),
//                                                                   the DBs written from the pipeline's sinks
//   clustersearch  <querySetDB> <targetSetDB> <out.tsv> <tmpDir> = search -> prefixid -> besthitbyset -> mergeresultsbyset ->
//                                                                   combinehits -> clusterhits -> summarizeresults
//                                                                   (R/data/clustersearch.sh:110-152), fused in-process
// Several ranks (RANK / WORLD_SIZE / LOCAL_RANK in the environment, one process per GPU): whole query sets are dealt to
// the ranks, every rank's cluster records are gathered on rank 0 (RCCL: sd_gather_results), which writes the TSV (the
// reference's MPI mode merges per-rank result files on the master instead, M/src/prefiltering/Prefiltering.cpp:619-650).
#include "sd_cli.h"
#include "sd_align_core.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <omp.h>
#include <sys/stat.h>
#include <unistd.h>
#include <thread>

namespace sdcli {

namespace {

struct SearchH {
    sd_search *s = nullptr;
    ~SearchH() { if (s) sd_search_destroy(s); }
};
struct HostH {
    sd_host *h = nullptr;
    bool own = true;   // false: the resident host object of a workflow (sd_cli.h)
    ~HostH() { if (h && own) sd_host_destroy(h); }
    int open(int threads) {
        if (resident().enabled) {
            h = resident().host(threads);
            own = false;
            return h ? SD_OK : SD_ENOMEM;
        }
        return sd_host_create(threads, &h);
    }
};

int envInt(const char *name, int def) {
    const char *e = getenv(name);
    return e ? atoi(e) : def;
}

// search parameters from the command line over the workflow's defaults
void fillParams(const Args &a, sd_search_params &p, bool clustersearchDefaults) {
    sd_search_default_params(&p);
    if (!clustersearchDefaults) {   // plain `search` defaults (M/src/commons/Parameters.cpp:2351-2611)
        p.sensitivity = 5.7f;       // search workflow default (Parameters: sensitivity 5.7 for search)
        p.evalThr = 0.001;
        p.covMode = 0;
        p.covThr = 0.0f;
        p.alnLenThr = 0;
    }
    p.sensitivity = (float) a.real("-s", p.sensitivity);
    p.kmerSize = (int32_t) a.integer("-k", 0);
    p.maxSeqs = (int32_t) a.integer("--max-seqs", 300);
    p.minDiagScore = (int32_t) a.integer("--min-ungapped-score", 15);
    p.binSize = (uint32_t) a.integer("--bin-size", 0);
    p.mask = (int32_t) a.integer("--mask", 1);
    p.maskProb = a.real("--mask-prob", 0.9);
    p.compBiasCorr = (int32_t) a.integer("--comp-bias-corr", 1);
    p.evalThr = a.real("-e", p.evalThr);
    p.covMode = (int32_t) a.integer("--cov-mode", p.covMode);
    p.covThr = (float) a.real("-c", p.covThr);
    p.alnLenThr = (int32_t) a.integer("--min-aln-len", p.alnLenThr);
    p.maxGeneGap = (uint32_t) a.integer("--max-gene-gap", 3);
    p.clusterSize = (uint32_t) a.integer("--cluster-size", 2);
    p.alpha = a.real("--alpha", 1.0);
    p.pCluThr = (float) a.real("--cluster-pval", 0.01);
    p.pMHThr = (float) a.real("--multihit-pval", 0.01);
    p.filterSelfMatch = a.flag("--filter-self-match", false) ? 1 : 0;
    p.chunkQueries = (int32_t) a.integer("--chunk-queries", 0);   // 0: sd_search_stream chooses by the size of the target set
    p.threads = threadsOf(a);
}

int checkWorkflowFlags(const Args &a) {
    if (a.integer("--compressed", 0) != 0) return fail("--compressed 1 is not supported");
    if (a.integer("--search-mode", 0) != 0) return fail("--search-mode 1/2 (Foldseek) are outside this path; only --search-mode 0");
    if (a.flag("--profile-cluster-search", false)) return fail("--profile-cluster-search is outside this path");
    if (a.integer("--gpu", 0) != 0) return fail("--gpu 1 selects the reference's CUDA ungapped prefilter (a different algorithm); run without it");
    if (a.multi("--sub-mat", "aa", "blosum62.out") != "blosum62.out") return fail("--sub-mat: only blosum62.out is built into this path");
    if (a.multi("--seed-sub-mat", "aa", "VTML80.out") != "VTML80.out") return fail("--seed-sub-mat: only VTML80.out is built into this path");
    if (a.integer("--exact-kmer-matching", 0) != 0 || a.integer("--spaced-kmer-mode", 1) != 1 || !a.flag("--diag-score", true))
        return fail("--exact-kmer-matching 1 / --spaced-kmer-mode 0 / --diag-score 0 are not supported");
    if (a.integer("--alt-ali", 0) != 0 || a.flag("--realign", false)) return fail("--alt-ali / --realign: use the `align` module");
    if (a.real("--min-seq-id", 0.0) != 0.0) return fail("--min-seq-id > 0: use the module-by-module path");
    if (a.integer("--max-accept", INT_MAX) != INT_MAX || a.integer("--max-rejected", INT_MAX) != INT_MAX)
        return fail("--max-accept / --max-rejected: use the `align` module");
    if (!a.flag("--simple-best-hit", true) || a.integer("--suboptimal-hits", 0) != 0 || a.integer("--aggregation-mode", 0) != 0)
        return fail("the fused aggregation implements --simple-best-hit 1 --suboptimal-hits 0 --aggregation-mode 0; use the glue modules otherwise");
    return 0;
}

// DB writers fed by the pipeline's sinks
struct Sinks {
    const SeqDb *qdb = nullptr, *tdb = nullptr;
    sddb::Writer pref, aln;
    bool wantPref = false, wantAln = false, failed = false;
    sd_aln_criteria crit;
    sd_alntext *text = nullptr;
    std::string buf;
    std::vector<uint32_t> order, counts;
    std::vector<int32_t> qlen;
    ~Sinks() { if (text) sd_alntext_destroy(text); }

    static void onPref(void *u, uint32_t first, uint32_t nQ, const sd_hit *rows, const uint32_t *counts, uint32_t W) {
        Sinks *s = (Sinks *) u;
        if (!s->wantPref || s->failed) return;
        char line[64];
        std::string &b = s->buf;
        for (uint32_t i = 0; i < nQ; i++) {
            b.clear();
            const sd_hit *row = rows + (size_t) i * W;
            for (uint32_t x = 0; x < counts[i]; x++) {
                const int len = snprintf(line, sizeof(line), "%u\t%d\t%d\n", s->tdb->keys[row[x].seqId], row[x].score, (int) (int16_t) row[x].diagonal);
                b.append(line, (size_t) len);
            }
            if (!s->pref.write(s->qdb->keys[first + i], b.data(), b.size())) s->failed = true;
        }
    }
    static void onAln(void *u, uint32_t first, uint32_t nQ, uint32_t nRes, const uint32_t *resQ, const uint32_t *resT,
                      const sd_sw_result *res, const uint8_t *ident, const char *pool) {
        Sinks *s = (Sinks *) u;
        if (!s->wantAln || s->failed) return;
        s->qlen.resize(nQ);
        for (uint32_t i = 0; i < nQ; i++) s->qlen[i] = s->qdb->lens[first + i];
        s->order.resize(std::max<uint32_t>(nRes, 1));
        s->counts.assign(std::max<uint32_t>(nQ, 1), 0);
        if (sd_host_accept_sort(&s->crit, nQ, nRes, resQ, resT, res, ident, s->qlen.data(), s->tdb->lens.data(), s->tdb->keys.data(),
                                s->order.data(), s->counts.data()) != SD_OK ||
            sd_alntext_format(s->text, &s->crit, nQ, s->counts.data(), s->order.data(), resT, res, ident, pool, s->qlen.data(),
                              s->tdb->lens.data(), s->tdb->keys.data()) != SD_OK) {
            s->failed = true;
            return;
        }
        const char *txt;
        const uint64_t *eoff;
        sd_alntext_get(s->text, &txt, &eoff);
        for (uint32_t i = 0; i < nQ; i++)
            if (!s->aln.write(s->qdb->keys[first + i], txt + eoff[i], (size_t) (eoff[i + 1] - eoff[i]))) s->failed = true;
    }
};

// Several ranks leave NAME.<rank> + NAME.<rank>.index each; rank 0 turns them into one DB with split data files -- NAME.0 ..
// NAME.<world-1> and one NAME.index whose offsets run through their concatenation, the layout the reference's multi-threaded
// DBWriter leaves and every reader here understands (M/src/commons/DBWriter.cpp:540-600)
bool mergeRankDbs(const std::string &name, int world, int dbtype, std::string *err) {
    struct Row {
        uint32_t key;
        uint64_t off, len;
    };
    std::vector<Row> rows;
    uint64_t base = 0;
    for (int r = 0; r < world; r++) {
        const std::string part = name + "." + std::to_string(r);
        std::ifstream in(part + ".index");
        if (!in) {
            *err = "cannot read " + part + ".index";
            return false;
        }
        unsigned long long k, o, l;
        while (in >> k >> o >> l) rows.push_back({(uint32_t) k, base + o, l});
        struct stat st;
        if (stat(part.c_str(), &st) != 0) {
            *err = "cannot stat " + part;
            return false;
        }
        base += (uint64_t) st.st_size;
        remove((part + ".index").c_str());
        remove((part + ".dbtype").c_str());
    }
    std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.key < y.key; });
    FILE *f = fopen((name + ".index").c_str(), "wb");
    if (!f) {
        *err = "cannot write " + name + ".index";
        return false;
    }
    for (const Row &w : rows) fprintf(f, "%u\t%llu\t%llu\n", w.key, (unsigned long long) w.off, (unsigned long long) w.len);
    fclose(f);
    f = fopen((name + ".dbtype").c_str(), "wb");
    if (!f) {
        *err = "cannot write " + name + ".dbtype";
        return false;
    }
    const int32_t t = dbtype;
    fwrite(&t, sizeof(t), 1, f);
    fclose(f);
    return true;
}

int runSearch(const Args &a, bool withClusters) {
    if (a.pos.size() != 4)
        return fail(withClusters ? "usage: clustersearch <querySetDB> <targetSetDB> <out.tsv> <tmpDir> [options]"
                                 : "usage: search <queryDB> <targetDB> <alignmentDB> <tmpDir> [options]");
    if (int rc = checkWorkflowFlags(a)) return rc;
    if (a.integer("--num-iterations", 1) > 1) return fail("--num-iterations > 1: run the iterations with the modules (prefilter, align, result2profile)");
    const int rank = (int) a.integer("--rank", envInt("RANK", 0)), world = (int) a.integer("--world-size", envInt("WORLD_SIZE", 1));
    const int device = a.has("--device") ? (int) a.integer("--device", 0) : envInt("LOCAL_RANK", 0);
    sd_search_params par;
    fillParams(a, par, withClusters);
    const std::string tmpDir = a.pos[3];
    mkdir(tmpDir.c_str(), 0777);

    HostH host;
    if (sd_host_create(par.threads, &host.h) != SD_OK) return fail("sd_host_create failed");
    std::string err;
    const bool sameDb = a.pos[0] == a.pos[1];
    std::unique_ptr<SeqDb> tdb(new SeqDb()), qdbOwn;
    if (!tdb->load(a.pos[1], host.h, &err)) return fail(err);
    if (tdb->profile) return fail("profile target databases are not supported on this path");
    SeqDb *qdb = tdb.get();
    if (!sameDb) {
        qdbOwn.reset(new SeqDb());
        if (!qdbOwn->load(a.pos[0], host.h, &err)) return fail(err);
        qdb = qdbOwn.get();
    }
    SetInfo qs, tsOwn;
    const SetInfo *qsP = nullptr, *tsP = nullptr;
    if (withClusters) {
        if (!qs.load(a.pos[0], true, &err)) return fail(err);
        if (!sameDb && !tsOwn.load(a.pos[1], true, &err)) return fail(err);
        qsP = &qs;
        tsP = sameDb ? &qs : &tsOwn;
    }
    par.profileQueries = qdb->profile ? 1 : 0;
    SetDbArrays tv, qv;
    tv.fill(*tdb, tsP);
    qv.fill(*qdb, qsP);
    info(a, "Query database size: %u type: %s\nTarget database size: %u type: Aminoacid\n", qdb->n, qdb->profile ? "Profile" : "Aminoacid", tdb->n);

    SearchH S;
    int rc;
    {
        // TARGET.idx (createindex layout) instead of a rebuild when its parameters match this run
        const int k = par.kmerSize ? par.kmerSize : sd_host_auto_kmer_size(tdb->totalResidues());
        const int thr = par.profileQueries ? 0 : sd_host_kmer_threshold(par.sensitivity, k);
        LoadedIndex loaded;
        std::string why;
        const int got = loadTargetIndex(a.pos[1], k, thr, par.mask ? 1 : 0, tdb->n, tdb->totalResidues(), loaded, &why);
        if (got < 0) return fail(why);
        if (got == 0) {
            sd_index_view v;
            v.kmerSize = k;
            v.kmerThr = thr;
            v.kmerOffsets = loaded.offsets.data();
            v.kmerBlockBase = loaded.blockBase.empty() ? nullptr : loaded.blockBase.data();
            v.entrySeq = loaded.entrySeq.data();
            v.entryPos = loaded.entryPos.data();
            v.nEntries = loaded.nEntries;
            v.maskedResidues = loaded.masked.data();
            v.nMaskedResidues = 0;
            info(a, "Use index %s.idx\n", a.pos[1].c_str());
            rc = sd_search_create_indexed(device, &par, &tv.view, &v, &S.s);
        } else {
            if (sddb::fileExists(a.pos[1] + ".idx.index")) info(a, "Index file not used: %s\n", why.c_str());
            rc = sd_search_create(device, &par, &tv.view, &S.s);
        }
    }
    if (rc == SD_ENODEVICE) return fail("no usable HIP device (sd_search_create returned -1); this path has no CPU fallback");
    if (rc != SD_OK) return fail("sd_search_create failed (" + std::to_string(rc) + ")");
    uint64_t st[16];
    double tm[16];
    sd_search_stats(S.s, st, tm);
    info(a, "Index table: k-mer size %llu, k-mer threshold %llu, %llu entries, %llu masked residues (%.2f s host, %.2f s upload)\n",
         (unsigned long long) st[11], (unsigned long long) st[12], (unsigned long long) st[9], (unsigned long long) st[10], tm[0], tm[1]);

    // query ranges of this rank: whole query sets (contiguous id ranges), dealt greedily by residue count; a plain search
    // deals contiguous blocks of queries
    std::vector<uint32_t> rb, re;
    if (world <= 1) {
        rb.push_back(0);
        re.push_back(qdb->n);
    } else if (withClusters) {
        std::vector<uint32_t> first(qs.nSets, UINT32_MAX), last(qs.nSets, 0);
        std::vector<uint64_t> resOfSet(qs.nSets, 0);
        for (uint32_t i = 0; i < qdb->n; i++) {
            const uint32_t sId = qv.setId[i];
            first[sId] = std::min(first[sId], i);
            last[sId] = std::max(last[sId], i + 1);
            resOfSet[sId] += (uint64_t) qdb->lens[i];
        }
        for (uint32_t sId = 0; sId < qs.nSets; sId++)
            if (first[sId] != UINT32_MAX) {
                uint64_t members = 0;
                for (uint32_t i = first[sId]; i < last[sId]; i++) members += qv.setId[i] == sId;
                if (members != last[sId] - first[sId]) return fail("the members of a query set are not contiguous in the DB; multi-rank sharding needs createsetdb's layout");
            }
        std::vector<uint32_t> mine(std::max<uint32_t>(qs.nSets, 1));
        uint32_t nMine = 0;
        sd_shard_query_sets(resOfSet.data(), qs.nSets, (uint32_t) world, (uint32_t) rank, mine.data(), &nMine);
        for (uint32_t x = 0; x < nMine; x++)
            if (first[mine[x]] != UINT32_MAX) {
                // neighbouring sets of this rank merge into one range
                if (!re.empty() && re.back() == first[mine[x]]) re.back() = last[mine[x]];
                else {
                    rb.push_back(first[mine[x]]);
                    re.push_back(last[mine[x]]);
                }
            }
    } else {
        const uint64_t per = ((uint64_t) qdb->n + world - 1) / world;
        rb.push_back((uint32_t) std::min<uint64_t>(qdb->n, per * rank));
        re.push_back((uint32_t) std::min<uint64_t>(qdb->n, per * (rank + 1)));
    }

    // DB outputs: `search` always writes the alignment DB (and pref_0 under <tmpDir>); clustersearch only with --keep-dbs 1
    Sinks sinks;
    sinks.qdb = qdb;
    sinks.tdb = tdb.get();
    const bool dbs = !withClusters || a.integer("--keep-dbs", 0) != 0;
    const std::string rankSuffix = world > 1 ? "." + std::to_string(rank) : "";
    if (dbs) {
        memset(&sinks.crit, 0, sizeof(sinks.crit));
        sinks.crit.evalThr = par.evalThr;
        sinks.crit.alnLenThr = par.alnLenThr;
        sinks.crit.covMode = par.covMode;
        sinks.crit.covThr = par.covThr;
        sinks.crit.swMode = 2;
        sinks.crit.addBacktrace = 1;
        sinks.crit.realignMaxSeqs = INT_MAX;
        sinks.crit.maxAccept = (uint32_t) INT_MAX;
        sinks.crit.maxRejected = (uint32_t) INT_MAX;
        sd_alntext_create(&sinks.text);
        const std::string alnPath = (withClusters ? tmpDir + "/result" : a.pos[2]) + rankSuffix;
        if (!sinks.pref.open(tmpDir + "/pref_0" + rankSuffix, sddb::DBTYPE_PREFILTER_RES, &err)) return fail(err);
        if (!sinks.aln.open(alnPath, sddb::DBTYPE_ALIGNMENT_RES, &err)) return fail(err);
        sinks.wantPref = sinks.wantAln = true;
        sd_search_set_sinks(S.s, Sinks::onPref, Sinks::onAln, &sinks);
    }
    // the ranks meet over TCP at MASTER_ADDR:MASTER_PORT+1: the RCCL unique id (clustersearch), the merge of the per-rank DBs
    sd_tcp *tcp = nullptr;
    struct TcpClose {
        sd_tcp *&t;
        ~TcpClose() { if (t) sd_tcp_close(t); }
    } tcpClose{tcp};
    // The one exchange of the path: every rank's cluster records to rank 0, which writes the TSV from the gathered buffer -- round by
    // round behind the running search (sd_gather_stream_*: a rank's ranges in eight rounds; the records of a range leave through the
    // stream's records sink when the range is finalised).  Over RCCL (sd_gather_results per round); ranks that share a device -- RCCL
    // refuses that, a one-GPU test rig -- hand their records over the TCP rendezvous instead.
    sd_comm *comm = nullptr;
    sd_gather_stream *gstream = nullptr;
    struct GatherClose {   // (declared behind tcpClose: destroyed first -- the worker thread uses the socket / the communicator)
        sd_gather_stream *&g;
        sd_comm *&c;
        ~GatherClose() {
            if (g) sd_gather_stream_destroy(g);
            if (c) sd_comm_destroy(c);
        }
    } gatherClose{gstream, comm};
    bool sharedDevice = false;
    if (world > 1) {
        const char *addrEnv = getenv("MASTER_ADDR");
        const std::string addr = addrEnv && *addrEnv ? addrEnv : "127.0.0.1";
        const int port = (int) a.integer("--comm-port", envInt("MASTER_PORT", 29500) + 1);
        if (sd_tcp_connect(addr.c_str(), port, world, rank, &tcp) != SD_OK)
            return fail("rendezvous of the ranks at " + addr + ":" + std::to_string(port) + " failed");
    }
    if (world > 1 && withClusters) {
        // the physical GPU of every rank (host name + PCI bus id, not the local ordinal: ranks on different nodes, or ranks that
        // each see their GPU as device 0 through HIP_VISIBLE_DEVICES, are not sharing) -> does each rank have a GPU of its own?
        uint64_t gpuId = 0;
        if (sd_device_identity(device, &gpuId) != SD_OK) return fail("sd_device_identity failed");
        int32_t devs[2] = {(int32_t) (uint32_t) gpuId, (int32_t) (uint32_t) (gpuId >> 32)};
        std::vector<int32_t> allDevs((size_t) world * 2, 0);
        uint64_t got = 0;
        if (sd_tcp_gather(tcp, devs, sizeof(devs), nullptr, allDevs.data(), allDevs.size() * sizeof(int32_t), &got) != SD_OK)
            return fail("rendezvous of the ranks failed (devices)");
        int32_t shared = 0;
        if (rank == 0)
            for (int x = 0; x < world; x++)
                for (int y = 0; y < x; y++)
                    if (allDevs[(size_t) x * 2] == allDevs[(size_t) y * 2] && allDevs[(size_t) x * 2 + 1] == allDevs[(size_t) y * 2 + 1]) shared = 1;
        if (sd_tcp_bcast(tcp, &shared, sizeof(shared)) != SD_OK) return fail("rendezvous of the ranks failed");
        sharedDevice = shared != 0;
        constexpr uint32_t GATHER_ROUNDS = 8;
        std::vector<uint32_t> roundOfRange(rb.size());
        for (size_t i = 0; i < rb.size(); i++) roundOfRange[i] = (uint32_t) (i * GATHER_ROUNDS / rb.size());
        if (!sharedDevice) {
            char uid[128];
            if (rank == 0 && sd_comm_unique_id(uid) != SD_OK) return fail("sd_comm_unique_id failed (librccl not loadable?)");
            if (sd_tcp_bcast(tcp, uid, sizeof(uid)) != SD_OK) return fail("broadcast of the RCCL unique id failed");
            if (sd_comm_init(device, world, rank, uid, &comm) != SD_OK) return fail("sd_comm_init failed");
            rc = sd_gather_stream_begin(comm, 0, (uint32_t) rb.size(), roundOfRange.data(), GATHER_ROUNDS, nullptr, 0, 1, &gstream);
        } else {
            rc = sd_gather_stream_begin_tcp(tcp, world, rank, (uint32_t) rb.size(), roundOfRange.data(), GATHER_ROUNDS, nullptr, 0, 1, &gstream);
        }
        if (rc != SD_OK) return fail("sd_gather_stream_begin failed (" + std::to_string(rc) + ")");
        sd_search_set_records_sink(S.s, sd_gather_stream_sink, gstream);
    }
    std::vector<sd_search_result *> results(std::max<size_t>(rb.size(), 1), nullptr);
    if (!rb.empty()) {
        // the TSV (this rank's, or rank 0's from the gathered buffers) is written from the cluster records: the stream builds them range by
        // range while its lanes work (sd_search_result_records below is then a copy, not a second pass over every entry's hits)
        if (withClusters) sd_search_set_want_records(S.s, 1);
        rc = sd_search_stream(S.s, &qv.view, sameDb ? 1 : 0, (uint32_t) rb.size(), rb.data(), re.data(), results.data());
        sd_search_set_records_sink(S.s, nullptr, nullptr);
        if (rc != SD_OK) return fail(std::string("sd_search_stream: ") + sd_search_last_error(S.s));
    }
    struct Free {
        std::vector<sd_search_result *> &v;
        ~Free() { for (sd_search_result *r : v) if (r) sd_search_result_destroy(r); }
    } freeResults{results};
    if (dbs) {
        if (sinks.failed) return fail("writing the prefilter / alignment DB failed");
        if (!sinks.pref.close(&err) || !sinks.aln.close(&err)) return fail(err);
    }
    // (the gather's last rounds: the socket / the communicator are the worker thread's until then)
    const void *gathered = nullptr;
    uint64_t gatheredBytes = 0;
    if (gstream) {
        rc = sd_gather_stream_wait(gstream, nullptr, nullptr, &gatheredBytes, &gathered);
        if (rc == SD_EMISMATCH) return fail("a rank closed its gather before all its query sets had arrived (its search failed): no result is written");
        if (rc != SD_OK) return fail("the gather of the cluster records failed (" + std::to_string(rc) + ")" + (comm ? std::string(": ") + sd_comm_last_error(comm) : std::string()));
        info(a, "records gathered %s: %llu bytes from %d ranks\n", sharedDevice ? "over TCP (ranks share a device)" : "over RCCL", (unsigned long long) gatheredBytes, world);
    }
    if (world > 1) {
        if (dbs) {   // every rank's DB parts are closed: rank 0 makes them one DB (split data files, one index)
            uint8_t done = 1;
            std::vector<uint8_t> allDone((size_t) world, 0);
            uint64_t got = 0;
            if (sd_tcp_gather(tcp, &done, 1, nullptr, allDone.data(), allDone.size(), &got) != SD_OK) return fail("rendezvous of the ranks failed (DB parts)");
            if (rank == 0) {
                const std::string alnPath = withClusters ? tmpDir + "/result" : a.pos[2];
                if (!mergeRankDbs(tmpDir + "/pref_0", world, sddb::DBTYPE_PREFILTER_RES, &err)) return fail(err);
                if (!mergeRankDbs(alnPath, world, sddb::DBTYPE_ALIGNMENT_RES, &err)) return fail(err);
            }
        }
    }
    sd_search_stats(S.s, st, tm);
    info(a, "%llu prefilter hits, %llu pairs aligned; prefilter %.2f s | align %.2f s | aggregate %.2f s | clusterhits %.2f s | total %.2f s\n",
         (unsigned long long) st[4], (unsigned long long) st[5], tm[3], tm[6], tm[8], tm[9], tm[11]);
    const uint64_t notComputed = st[14];
    if (notComputed) fprintf(stderr, "sdgpu %s: %s\n", a.module.c_str(), sd_search_last_error(S.s));
    if (!withClusters) return notComputed ? 1 : 0;

    // summarizeresults: this rank's part, then rank 0 concatenates in rank order
    std::string qn, tn, qsrc, tsrc;
    std::vector<uint64_t> qno, tno, qso, tso;
    {
        std::vector<std::string> names(qdb->n);
        for (uint32_t i = 0; i < qdb->n; i++) names[i] = qdb->keys[i] < qs.nameOfKey.size() ? qs.nameOfKey[qdb->keys[i]] : std::string();
        packNames(names, qn, qno);
        names.assign(tdb->n, std::string());
        for (uint32_t i = 0; i < tdb->n; i++) names[i] = tdb->keys[i] < tsP->nameOfKey.size() ? tsP->nameOfKey[tdb->keys[i]] : std::string();
        packNames(names, tn, tno);
        packNames(qs.sourceOfSet, qsrc, qso);
        packNames(tsP->sourceOfSet, tsrc, tso);
    }
    // one rank: its cluster records (every range's clusters, in range order); several: what the gather brought to rank 0
    std::vector<char> rec;
    const char *toWrite = nullptr;
    uint64_t toWriteBytes = 0;
    if (world == 1) {
        for (size_t r = 0; r < rb.size(); r++) {
            uint64_t need = 0;
            rc = sd_search_result_records(results[r], nullptr, 0, &need);
            if (rc != SD_OK) return fail("sd_search_result_records failed (" + std::to_string(rc) + ")");
            const size_t at = rec.size();
            rec.resize(at + need);
            rc = sd_search_result_records(results[r], rec.data() + at, need, &need);
            if (rc != SD_OK) return fail("sd_search_result_records failed (" + std::to_string(rc) + ")");
        }
        toWrite = rec.data();
        toWriteBytes = rec.size();
    } else {
        toWrite = (const char *) gathered;
        toWriteBytes = gatheredBytes;
    }
    uint64_t nClu = 0, nHit = 0;
    if (rank == 0) {
        // (gathered records came over the wire: indices are checked against the tables they will index)
        rc = sd_records_check(toWrite, toWriteBytes, (uint32_t) (qso.size() - 1), (uint32_t) (tso.size() - 1), (uint32_t) (qno.size() - 1),
                              (uint32_t) (tno.size() - 1), nullptr, nullptr);
        if (rc != SD_OK) return fail("the gathered cluster records are truncated or index outside the name tables");
        rc = sd_records_write_tsv(toWrite, toWriteBytes, a.pos[2].c_str(), 0, 0, qn.data(), qno.data(), tn.data(), tno.data(),
                                  qsrc.data(), qso.data(), tsrc.data(), tso.data(), 0, &nClu, &nHit);
        if (rc != SD_OK) return fail("sd_records_write_tsv failed (" + std::to_string(rc) + ")");
        info(a, "%llu clusters with %llu hits written\n", (unsigned long long) nClu, (unsigned long long) nHit);
    }
    return notComputed ? 1 : 0;
}

}  // namespace

int r2pSetupFromArgs(const Args &a, R2pSetup &s) {
    if (a.integer("--compressed", 0) != 0) return fail("--compressed 1 is not supported");
    if (a.integer("--profile-output-mode", 0) != 0) return fail("--profile-output-mode 0 only");
    if (a.integer("--pseudo-cnt-mode", 0) != 0) return fail("--pseudo-cnt-mode 1 needs the context library (not built in)");
    if (a.flag("--allow-deletion", false)) return fail("--allow-deletion 1 is not supported");
    if (a.multi("--sub-mat", "aa", "blosum62.out") != "blosum62.out") return fail("--sub-mat: only blosum62.out is built into this path");
    s.qid = a.str("--qid", "0.0");
    sd_r2p_params &p = s.par;
    memset(&p, 0, sizeof(p));
    p.filterMsa = (int32_t) a.integer("--filter-msa", 1);
    p.filterMinEnable = (int32_t) a.integer("--filter-min-enable", 0);
    p.filterMaxSeqId = (float) a.real("--max-seq-id", 0.9);
    p.qid = s.qid.c_str();
    p.qsc = (float) a.real("--qsc", -20.0);
    p.covMSAThr = (float) a.real("--cov", 0.0);
    p.Ndiff = (int32_t) a.integer("--diff", 1000);
    p.pcMode = 0;
    p.pca = (float) strtod(a.multi("--pca", "substitution", "1.1").c_str(), nullptr);
    p.pcb = (float) strtod(a.multi("--pcb", "substitution", "4.1").c_str(), nullptr);
    p.wg = a.flag("--wg", false) ? 1 : 0;
    p.compBiasCorr = (int32_t) a.integer("--comp-bias-corr", 1);
    p.maskProfile = (int32_t) a.integer("--mask-profile", 1);
    p.maskProb = a.real("--mask-prob", 0.9);
    double evalThr = a.real("-e", 0.001), evalProfile = a.real("--e-profile", 0.001);
    s.evalProfile = (evalThr < evalProfile) ? evalThr : evalProfile;   // result2profile.cpp:33
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// result2profile <queryDB> <targetDB> <alignmentDB> <profileDB>   (M/src/util/result2profile.cpp:16-322): the host step
// between search iterations.  The arithmetic is sd_r2p_batch of the C ABI; this is the DB side.
int result2profileModule(const Args &a) {
    if (a.pos.size() != 4) return fail("usage: result2profile <queryDB> <targetDB> <alignmentDB> <profileDB> [options]");
    R2pSetup RS;
    if (int rcS = r2pSetupFromArgs(a, RS)) return rcS;
    const sd_r2p_params &p = RS.par;
    const double evalProfile = RS.evalProfile;
    const int threads = threadsOf(a);
    Lap lap("result2profile");
    HostH host;
    if (host.open(threads) != SD_OK) return fail("sd_host_create failed");
    std::string err;
    const bool sameDb = a.pos[0] == a.pos[1];
    std::shared_ptr<SeqDb> tdb = loadTargetDb(a.pos[1], host.h, &err);
    std::unique_ptr<SeqDb> qdbOwn;
    if (!tdb) return fail(err);
    if (tdb->profile) return fail("Only the query OR the target database can be a profile database");
    SeqDb *qdb = tdb.get();
    if (!sameDb) {
        qdbOwn.reset(new SeqDb());
        if (!qdbOwn->load(a.pos[0], host.h, &err)) return fail(err);
        qdb = qdbOwn.get();
    }
    lap.mark("load DBs");
    sddb::Reader aln;
    if (!aln.open(a.pos[2], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    sd_r2p *r2p = nullptr;
    if (sd_r2p_create(&r2p) != SD_OK) return fail("sd_r2p_create failed");
    std::unique_ptr<sd_r2p, void (*)(sd_r2p *)> guard(r2p, sd_r2p_destroy);
    // the position-specific weights -- the O(columns^2 x rows) part -- run on the GPU (sd_r2p_batch_device); --profile-weights-host 1
    // asks for the host implementation explicitly (a box without a GPU, the CPU-side tests).  No silent fallback.
    const bool hostWeights = a.integer("--profile-weights-host", 0) != 0;
    sd_ctx *r2pCtx = nullptr;
    bool ownCtx = true;
    if (!hostWeights) {
        const int device = a.has("--device") ? (int) a.integer("--device", 0) : envInt("LOCAL_RANK", 0);
        int rcCtx = SD_OK;
        if (resident().enabled) {
            r2pCtx = resident().ctx(device, &rcCtx);
            ownCtx = false;
        } else {
            rcCtx = sd_ctx_create(device, &r2pCtx);
        }
        if (rcCtx != SD_OK)
            return fail("no usable HIP device (sd_ctx_create returned " + std::to_string(rcCtx) + "); result2profile computes the sequence weights on "
                        "the GPU (--profile-weights-host 1 selects the host implementation)");
    }
    std::unique_ptr<sd_ctx, void (*)(sd_ctx *)> ctxGuard(ownCtx ? r2pCtx : nullptr, sd_ctx_destroy);
    lap.mark("r2p object + context");
    sddb::Writer out;
    if (!out.open(a.pos[3], sddb::DBTYPE_HMM_PROFILE, &err)) return fail(err);
    const size_t n = aln.size();
    const size_t chunk = 4096;
    double tParse = 0, tCompute = 0, tWrite = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tLoaded = now();
    std::vector<uint8_t> qLetters;
    std::vector<uint64_t> qOff, edgeOff, btOff;
    std::vector<uint32_t> edgeT, keys;
    std::vector<int32_t> eQ, eT;
    std::string pool;
    std::vector<char> profiles;
    struct ParsedEntry {
        bool valid = false;
        size_t qId = 0;
        std::vector<uint32_t> edgeT;
        std::vector<int32_t> eQ, eT;
        std::vector<uint64_t> btEnd;
        std::string pool, error;
    };
    std::vector<ParsedEntry> parsed;
    for (size_t c0 = 0; c0 < n; c0 += chunk) {
        const size_t c1 = std::min(n, c0 + chunk);
        const double t0 = now();
        qLetters.clear();
        qOff.assign(1, 0);
        edgeOff.assign(1, 0);
        btOff.assign(1, 0);
        edgeT.clear();
        eQ.clear();
        eT.clear();
        keys.clear();
        pool.clear();
        // the entries of the chunk parsed on all threads, each into lists of its own, then laid out back to back in entry order
        parsed.resize(c1 - c0);
#pragma omp parallel for schedule(dynamic, 16)
        for (size_t id = c0; id < c1; id++) {
            ParsedEntry &pe = parsed[id - c0];
            pe.valid = false;
            pe.error.clear();
            pe.edgeT.clear();
            pe.eQ.clear();
            pe.eT.clear();
            pe.btEnd.clear();
            pe.pool.clear();
            const uint32_t qKey = aln.key(id);
            pe.qId = qdb->rd.idOfKey(qKey);
            if (pe.qId == SIZE_MAX) continue;   // "Invalid query sequence": skipped (result2profile.cpp:173-176)
            pe.valid = true;
            const char *d = aln.data(id);
            while (*d != '\0') {
                const char *ls = d;
                while (*d != '\n' && *d != '\0') d++;
                const char *le = d;
                if (*d == '\n') d++;
                // columns: tKey bits seqId eval qStart qEnd qLen tStart tEnd tLen [cigar]
                const char *col[12];
                int nc = 0;
                col[nc++] = ls;
                for (const char *c = ls; c < le && nc < 12; c++)
                    if (*c == '\t') col[nc++] = c + 1;
                const uint32_t tKey = (uint32_t) strtoul(ls, nullptr, 10);
                if (tKey == qKey && sameDb) continue;   // the query repeated in the same database case (:186-195)
                double evalue = 0.0;
                if (nc >= 4) evalue = strtod(col[3], nullptr);
                if (!(evalue < evalProfile)) continue;
                if (nc < 11) {
                    pe.error = "alignment DB without backtraces (run align with -a); recomputing them is the align module's job";
                    break;
                }
                const size_t tId = tdb->rd.idOfKey(tKey);
                if (tId == SIZE_MAX) {
                    pe.error = "Sequence " + std::to_string(tKey) + " does not exist in target sequence database";
                    break;
                }
                pe.edgeT.push_back((uint32_t) tId);
                pe.eQ.push_back((int32_t) strtol(col[4], nullptr, 10));
                pe.eT.push_back((int32_t) strtol(col[7], nullptr, 10));
                // Matcher::uncompressAlignment (Matcher.cpp:187-201)
                size_t count = 0;
                for (const char *c = col[10]; c < le; c++) {
                    if (*c >= '0' && *c <= '9') count = count * 10 + (size_t) (*c - '0');
                    else {
                        pe.pool.append(count == 0 ? 1 : count, *c);
                        count = 0;
                    }
                }
                pe.btEnd.push_back(pe.pool.size());
            }
        }
        for (size_t id = c0; id < c1; id++) {
            const ParsedEntry &pe = parsed[id - c0];
            if (!pe.error.empty()) return fail(pe.error);
            if (!pe.valid) continue;
            keys.push_back(aln.key(id));
            qLetters.insert(qLetters.end(), qdb->residues.begin() + qdb->offsets[pe.qId], qdb->residues.begin() + qdb->offsets[pe.qId + 1]);
            qOff.push_back(qLetters.size());
            edgeT.insert(edgeT.end(), pe.edgeT.begin(), pe.edgeT.end());
            eQ.insert(eQ.end(), pe.eQ.begin(), pe.eQ.end());
            eT.insert(eT.end(), pe.eT.begin(), pe.eT.end());
            const uint64_t base = pool.size();
            pool.append(pe.pool);
            for (uint64_t e : pe.btEnd) btOff.push_back(base + e);
            edgeOff.push_back(edgeT.size());
        }
        const uint32_t nQ = (uint32_t) keys.size();
        profiles.assign(qOff.back() * 25 + 1, 0);
        pool.push_back(' ');
        qLetters.push_back(0);
        edgeT.push_back(0);
        eQ.push_back(0);
        eT.push_back(0);
        const double t1 = now();
        const int rc = r2pCtx ? sd_r2p_batch_device(r2pCtx, r2p, &p, nQ, qLetters.data(), qOff.data(), edgeOff.data(), edgeT.data(), eQ.data(),
                                                    eT.data(), pool.data(), btOff.data(), tdb->residues.data(), tdb->offsets.data(),
                                                    profiles.data(), nullptr)
                              : sd_r2p_batch(r2p, &p, nQ, qLetters.data(), qOff.data(), edgeOff.data(), edgeT.data(), eQ.data(), eT.data(),
                                             pool.data(), btOff.data(), tdb->residues.data(), tdb->offsets.data(), profiles.data(), nullptr);
        if (rc != SD_OK) return fail("sd_r2p_batch failed (" + std::to_string(rc) + ")" + (r2pCtx ? std::string(": ") + sd_last_error(r2pCtx) : std::string()));
        const double t2 = now();
        for (uint32_t q = 0; q < nQ; q++)
            if (!out.write(keys[q], profiles.data() + qOff[q] * 25, (size_t) (qOff[q + 1] - qOff[q]) * 25)) return fail("cannot write " + a.pos[3]);
        tParse += t1 - t0;
        tCompute += t2 - t1;
        tWrite += now() - t2;
    }
    lap.mark("chunks");
    info(a, "result2profile: parse %.2f s, profiles %.2f s (%d threads, weights on the %s), write %.2f s\n", tParse, tCompute, threads,
         r2pCtx ? "GPU" : "host", tWrite);
    (void) tLoaded;
    if (!out.close(&err)) return fail(err);
    // the profile DB shares the query DB's ancillary files (DBReader::softlinkDb(..., SEQUENCE_ANCILLARY), :318-320)
    for (const char *suffix : {"_h", "_h.index", "_h.dbtype", ".lookup", ".source"}) {
        const std::string src = a.pos[0] + suffix, dst = a.pos[3] + suffix;
        ::remove(dst.c_str());
        if (sddb::fileExists(src)) {
            char *real = realpath(src.c_str(), nullptr);
            if (real) {
                if (symlink(real, dst.c_str()) != 0) { /* best effort, as the reference */ }
                free(real);
            }
        }
    }
    lap.mark("close + links");
    info(a, "%zu profiles written\n", n);
    return 0;
}

// subtractdbs <resultDB A> <resultDB B> <outDB>  (M/src/util/subtractdbs.cpp:13-118): lines of A whose target is not listed
// (with E <= threshold) in B
int subtractdbsModule(const Args &a) {
    if (a.pos.size() != 3) return fail("usage: subtractdbs <resultDB> <resultDB> <outDB>");
    double evalThr = a.real("-e", 0.001), evalProfile = a.real("--e-profile", 0.001);
    evalProfile = (evalThr < evalProfile) ? evalThr : evalProfile;
    std::string err;
    sddb::Reader left, right;
    if (!left.open(a.pos[0], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::LINEAR_ACCESS, &err)) return fail(err);
    if (!right.open(a.pos[1], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    sddb::Writer out;
    if (!out.open(a.pos[2], left.dbtype(), &err)) return fail(err);
    auto evalOf = [](const char *ls, const char *le) {
        int tabs = 0;
        const char *c3 = nullptr;
        for (const char *c = ls; c < le; c++)
            if (*c == '\t' && ++tabs == 3) c3 = c + 1;
        // an alignment record has at least ten columns; anything shorter (prefilter lines) counts as E = 0
        int cols = 1;
        for (const char *c = ls; c < le; c++) cols += (*c == '\t');
        return (cols >= 10 && c3) ? strtod(c3, nullptr) : 0.0;
    };
    // a block of entries on all threads (every thread its own lookup), written in order
    std::vector<std::string> results;
    const size_t block = 8192;
    for (size_t b0 = 0; b0 < left.size(); b0 += block) {
        const size_t b1 = std::min(left.size(), b0 + block);
        results.resize(b1 - b0);
#pragma omp parallel
        {
            std::map<unsigned, bool> lookup;
#pragma omp for schedule(dynamic, 16)
            for (size_t id = b0; id < b1; id++) {
                lookup.clear();
                const char *leftData = left.data(id);
                for (const char *d = leftData; *d != '\0';) {
                    const char *ls = d;
                    while (*d != '\n' && *d != '\0') d++;
                    if (evalOf(ls, d) <= evalProfile) lookup[(unsigned) strtoul(ls, nullptr, 10)] = true;
                    if (*d == '\n') d++;
                }
                const size_t rid = right.idOfKey(left.key(id));
                if (rid != SIZE_MAX) {
                    for (const char *d = right.data(rid); *d != '\0';) {
                        const char *ls = d;
                        while (*d != '\n' && *d != '\0') d++;
                        if (evalOf(ls, d) <= evalProfile) lookup[(unsigned) strtoul(ls, nullptr, 10)] = false;
                        if (*d == '\n') d++;
                    }
                }
                std::string &result = results[id - b0];
                result.clear();
                for (const char *d = leftData; *d != '\0';) {
                    const char *ls = d;
                    while (*d != '\n' && *d != '\0') d++;
                    if (*d == '\n') d++;
                    if (lookup[(unsigned) strtoul(ls, nullptr, 10)]) result.append(ls, d - ls);
                }
            }
        }
        for (size_t id = b0; id < b1; id++)
            if (!out.write(left.key(id), results[id - b0].data(), results[id - b0].size())) return fail("cannot write " + a.pos[2]);
    }
    if (!out.close(&err)) return fail(err);
    return 0;
}

// mergedbs <keyDB> <outDB> <DB1> <DB2> ...  (M/src/util/mergedbs.cpp:8-78): per key of keyDB the concatenation of its entries
int mergedbsModule(const Args &a) {
    if (a.pos.size() < 4) return fail("usage: mergedbs <sequenceDB> <outDB> <resultDB1> <resultDB2> [...]");
    if (a.has("--prefixes") && !a.str("--prefixes", "").empty()) return fail("--prefixes is not supported");
    std::string err;
    sddb::Reader keysDb;
    if (!keysDb.open(a.pos[0], sddb::Reader::USE_INDEX, sddb::Reader::NOSORT, &err)) return fail(err);
    std::vector<std::unique_ptr<sddb::Reader> > in;
    for (size_t i = 2; i < a.pos.size(); i++) {
        in.emplace_back(new sddb::Reader());
        if (!in.back()->open(a.pos[i], sddb::Reader::USE_INDEX | sddb::Reader::USE_DATA, sddb::Reader::NOSORT, &err)) return fail(err);
    }
    sddb::Writer out;
    if (!out.open(a.pos[1], in[0]->dbtype(), &err)) return fail(err);
    std::string buf;
    for (size_t id = 0; id < keysDb.size(); id++) {
        const uint32_t key = keysDb.key(id);
        buf.clear();
        for (size_t i = 0; i < in.size(); i++) {
            const size_t e = in[i]->idOfKey(key);
            if (e == SIZE_MAX) continue;
            buf.append(in[i]->data(e), in[i]->entryLength(e) - 1);
        }
        if (!out.write(key, buf.data(), buf.size())) return fail("cannot write " + a.pos[1]);
    }
    if (!out.close(&err)) return fail(err);
    return 0;
}

namespace {

int runModule(int (*fn)(const Args &), const char *name, const std::vector<std::string> &pos, const std::vector<std::string> &flags) {
    std::vector<const char *> argv;
    for (const std::string &s : pos) argv.push_back(s.c_str());
    for (const std::string &s : flags) argv.push_back(s.c_str());
    Args sub;
    sub.module = name;
    std::string err;
    if (!sub.parse((int) argv.size(), argv.data(), &err)) return fail(std::string(name) + ": " + err);
    info(sub, "");
    omp_set_num_threads(threadsOf(sub));
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = fn(sub);
    // the reference prints "Time for processing" after every module of a workflow (M/src/commons/CommandCaller / Timer)
    info(sub, "%s: time for processing %.2f s\n", name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return rc;
}

std::vector<std::string> with(std::vector<std::string> v, std::initializer_list<std::string> more) {
    v.insert(v.end(), more.begin(), more.end());
    return v;
}

// `search --num-iterations N` (M/src/workflow/Search.cpp:476-518 builds the per-step parameter strings,
// M/data/workflow/blastpgp.sh:52-140 runs them): iteration 0 with --realign and the profile E-value, prefilter results of
// later iterations minus what is aligned already, the last alignment with the user's -e, merged into the result DB
int iterativeSearch(const Args &a, const std::string &Q, const std::string &T, const std::string &result, const std::string &tmp,
                    const std::string *inMemoryTsv = nullptr) {
    const int numIt = (int) a.integer("--num-iterations", 1);
    if (!inMemoryTsv) mkdir(tmp.c_str(), 0777);
    const std::string eUser = a.str("-e", "0.001"), eProfile = a.str("--e-profile", "0.001");
    const std::vector<std::string> common = {"--threads", std::to_string(threadsOf(a)), "-v", a.str("-v", "3")};
    std::vector<std::string> pref = with(common, {"-s", a.str("-s", "5.7"), "-k", a.str("-k", "0"), "--max-seqs", a.str("--max-seqs", "300"), "-c",
                                                  a.str("-c", "0"), "--cov-mode", a.str("--cov-mode", "0"), "--min-ungapped-score",
                                                  a.str("--min-ungapped-score", "15"), "--mask", a.str("--mask", "1"), "--mask-prob",
                                                  a.str("--mask-prob", "0.9"), "--comp-bias-corr", a.str("--comp-bias-corr", "1")});
    for (const char *f : {"--device", "--bin-size", "--l2-cache-size", "--chunk-queries"})
        if (a.has(f)) pref = with(pref, {f, a.str(f, "")});
    std::vector<std::string> aln = with(common, {"-a", "1", "--alignment-mode", a.str("--alignment-mode", "2"), "--min-aln-len", a.str("--min-aln-len", "0"),
                                                 "-c", a.str("-c", "0"), "--cov-mode", a.str("--cov-mode", "0"), "--min-seq-id",
                                                 a.str("--min-seq-id", "0"), "--comp-bias-corr", a.str("--comp-bias-corr", "1"),
                                                 "--realign-score-bias", a.str("--realign-score-bias", "-0.2")});
    if (a.has("--device")) aln = with(aln, {"--device", a.str("--device", "0")});
    std::vector<std::string> prof = with(common, {"-e", eProfile, "--e-profile", eProfile, "--mask-profile", a.str("--mask-profile", "1"),
                                                        "--comp-bias-corr", a.str("--comp-bias-corr", "1"), "--filter-msa", a.str("--filter-msa", "1"),
                                                        "--filter-min-enable", a.str("--filter-min-enable", "0"), "--max-seq-id",
                                                        a.str("--max-seq-id", "0.9"), "--qid", a.str("--qid", "0.0"), "--qsc", a.str("--qsc", "-20"),
                                                        "--cov", a.str("--cov", "0"), "--diff", a.str("--diff", "1000"), "--pca",
                                                        a.str("--pca", "substitution:1.100,context:1.400"), "--pcb",
                                                        a.str("--pcb", "substitution:4.100,context:5.800")});
    // result2profile's sequence weights run on the GPU and the module fails without one unless the user asked for the host form:
    // the workflow hands both choices on (device and --profile-weights-host) instead of deciding for the module
    if (a.has("--profile-weights-host")) prof = with(prof, {"--profile-weights-host", a.str("--profile-weights-host", "0")});
    if (a.has("--device")) prof = with(prof, {"--device", a.str("--device", "0")});
    const bool keep = a.integer("--keep-tmp", 0) != 0;   // keep the per-iteration DBs (parity checks at size read them)
    if (inMemoryTsv) {
        // clustersearch without DB files between the modules (sd_mod_iter.cpp): chunks of queries through all iterations in memory,
        // several at a time, into one aggregation.  --keep-tmp 1 wants the per-iteration DBs: the module chain below writes them
        return iterativeClusterSearchInMemory(a, Q, T, *inMemoryTsv, pref, aln, prof, eUser, eProfile);
    }
    // the modules below run in this process: the target DB, its index on the device and its sequence set stay resident between them
    // (sd_cli.h: Resident) instead of being reloaded / rebuilt by every module
    struct ResidentScope {
        ResidentScope() { resident().enabled = !(getenv("SD_RESIDENT") && atoi(getenv("SD_RESIDENT")) == 0); }
        ~ResidentScope() { resident().clear(); }
    } residentScope;
    std::string query = Q;
    for (int step = 0; step < numIt; step++) {
        const std::string s = std::to_string(step), s1 = std::to_string(step - 1);
        const bool last = step == numIt - 1;
        const std::string prefDb = tmp + (step == 0 ? "/pref_0" : "/pref_tmp_" + s);
        if (int rc = runModule(prefilterModule, "prefilter", {query, T, prefDb}, pref)) return rc;
        std::string prefForAln = prefDb;
        if (step >= 1) {
            prefForAln = tmp + "/pref_" + s;
            if (int rc = runModule(subtractdbsModule, "subtractdbs", {prefDb, tmp + "/aln_" + s1, prefForAln},
                                   with(common, {"--e-profile", eProfile, "-e", eUser})))
                return rc;
            if (!keep) sddb::removeDb(prefDb);
        }
        // the realign pass belongs to iteration 0 only; every iteration but the last aligns with the profile E-value
        // (Search.cpp:484-486,497-505)
        std::vector<std::string> alnPar = with(aln, {"-e", last ? eUser : eProfile, "--realign", step == 0 ? "1" : "0"});
        const std::string alnDb = tmp + (step == 0 ? "/aln_0" : "/aln_tmp_" + s);
        if (int rc = runModule(alignModule, "align", {query, T, prefForAln, alnDb}, alnPar)) return rc;
        std::string merged = alnDb;
        if (step > 0) {
            merged = last ? result : tmp + "/aln_" + s;
            if (int rc = runModule(mergedbsModule, "mergedbs", {query, merged, tmp + "/aln_" + s1, alnDb}, {})) return rc;
            if (!keep) {
                sddb::removeDb(tmp + "/aln_" + s1);
                sddb::removeDb(alnDb);
            }
        }
        if (!last) {
            const std::string profDb = tmp + "/profile_" + s;
            if (int rc = runModule(result2profileModule, "result2profile", {query, T, merged, profDb}, prof)) return rc;
            query = profDb;
        }
    }
    return 0;
}

}  // namespace

int searchModule(const Args &a) {
    if (a.integer("--num-iterations", 1) > 1) {
        if (a.pos.size() != 4) return fail("usage: search <queryDB> <targetDB> <alignmentDB> <tmpDir> [options]");
        if (int rc = checkWorkflowFlags(a)) return rc;
        return iterativeSearch(a, a.pos[0], a.pos[1], a.pos[2], a.pos[3]);
    }
    return runSearch(a, false);
}

int clustersearchModule(const Args &a) {
    if (a.integer("--num-iterations", 1) <= 1) return runSearch(a, true);
    // iterative profile search (BASELINE config 4): the iterations through the modules, then the module chain of
    // R/data/clustersearch.sh:121-151 on the merged alignment DB
    if (a.pos.size() != 4) return fail("usage: clustersearch <querySetDB> <targetSetDB> <out.tsv> <tmpDir> [options]");
    if (int rc = checkWorkflowFlags(a)) return rc;
    const std::string Q = a.pos[0], T = a.pos[1], tmp = a.pos[3];
    mkdir(tmp.c_str(), 0777);
    Args s = a;   // the search step runs with the clustersearch workflow's defaults (R/src/workflow/clustersearch.cpp:9-37)
    auto def = [&](const char *f, const char *v) { if (!s.has(f)) s.opt[f] = v; };
    def("-s", "5.7");
    def("--cov-mode", "2");
    def("-c", "0.8");
    def("-e", "10");
    def("--min-aln-len", "30");
    def("--alignment-mode", "2");
    for (const char *db : {"/result", "/result_prefixed", "/aggregate", "/aggregate_merged", "/matches", "/matches_h", "/clusters", "/clusters_h"})
        sddb::removeDb(tmp + db);
    // default: in memory (no DB between the modules of an iteration, no text chain behind them).  The module chain runs when its DBs
    // are wanted (--keep-tmp 1), with SD_ITER_FILES=1, and for several ranks (they shard whole query sets over the module chain's DBs)
    const bool files = a.integer("--keep-tmp", 0) != 0 || (getenv("SD_ITER_FILES") && atoi(getenv("SD_ITER_FILES")) != 0) ||
                       envInt("WORLD_SIZE", 1) > 1 || a.integer("--world-size", 1) > 1;
    if (!files) return iterativeSearch(s, Q, T, tmp + "/result", tmp + "/search", &a.pos[2]);
    if (int rc = iterativeSearch(s, Q, T, tmp + "/result", tmp + "/search")) return rc;
    const std::vector<std::string> common = {"--threads", std::to_string(threadsOf(a)), "-v", a.str("-v", "3")};
    if (int rc = runModule(prefixidModule, "prefixid", {tmp + "/result", tmp + "/result_prefixed"}, common)) return rc;
    if (int rc = runModule(besthitbysetModule, "besthitbyset", {Q, T, tmp + "/result_prefixed", tmp + "/aggregate"},
                           with(common, {"--simple-best-hit", "1", "--suboptimal-hits", "0"})))
        return rc;
    if (int rc = runModule(mergeresultsbysetModule, "mergeresultsbyset", {Q + "_set_to_member", tmp + "/aggregate", tmp + "/aggregate_merged"}, common))
        return rc;
    if (int rc = runModule(combinehitsModule, "combinehits", {Q, T, tmp + "/aggregate_merged", tmp + "/matches", tmp},
                           with(common, {"--alpha", a.str("--alpha", "1"), "--aggregation-mode", "0", "--filter-self-match",
                                         a.flag("--filter-self-match", false) ? "1" : "0"})))
        return rc;
    std::vector<std::string> ch = with(common, {"--multihit-pval", a.str("--multihit-pval", "0.01"), "--cluster-pval", a.str("--cluster-pval", "0.01"),
                                                "--max-gene-gap", a.str("--max-gene-gap", "3"), "--cluster-size", a.str("--cluster-size", "2"),
                                                "--db-output", "1", "--alpha", a.str("--alpha", "1")});
    if (a.has("--device")) ch = with(ch, {"--device", a.str("--device", "0")});
    if (int rc = runModule(clusterhitsModule, "clusterhits", {Q, T, tmp + "/matches", tmp + "/clusters"}, ch)) return rc;
    return runModule(summarizeresultsModule, "summarizeresults", {Q, T, tmp + "/clusters", a.pos[2]}, common);
}

}  // namespace sdcli
