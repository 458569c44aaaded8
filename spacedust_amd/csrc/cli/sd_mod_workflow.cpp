// The workflow modules: what the reference runs as shell scripts over several processes, here one process around the
// C++ pipeline object of libsdgpu.so (sd_search_*):
//   search         <queryDB> <targetDB> <alignmentDB> <tmpDir>   = prefilter + align (M/data/workflow/blastp.sh:59-90),
//                                                                   the DBs written from the pipeline's sinks
//   clustersearch  <querySetDB> <targetSetDB> <out.tsv> <tmpDir> = search -> prefixid -> besthitbyset -> mergeresultsbyset ->
//                                                                   combinehits -> clusterhits -> summarizeresults
//                                                                   (R/data/clustersearch.sh:110-152), fused in-process
// Several ranks (RANK / WORLD_SIZE / LOCAL_RANK in the environment, one process per GPU): whole query sets are dealt to
// the ranks, every rank writes its part of the output, rank 0 concatenates after a marker-file barrier in <tmpDir> (the
// reference's MPI mode exchanges through the shared file system the same way, M/src/prefiltering/Prefiltering.cpp:619-650).
#include "sd_cli.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sys/stat.h>
#include <thread>

namespace sdcli {

namespace {

struct SearchH {
    sd_search *s = nullptr;
    ~SearchH() { if (s) sd_search_destroy(s); }
};
struct HostH {
    sd_host *h = nullptr;
    ~HostH() { if (h) sd_host_destroy(h); }
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
    p.chunkQueries = (int32_t) a.integer("--chunk-queries", 10000);
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

// sd_setdb over a loaded DB (+ set membership by id when a lookup exists)
struct SetDbArrays {
    std::vector<uint32_t> setId, pos;
    std::vector<uint8_t> strand;
    sd_setdb view;
    void fill(const SeqDb &db, const SetInfo *sets) {
        memset(&view, 0, sizeof(view));
        view.residues = db.residues.data();
        view.offsets = db.offsets.data();
        view.n = db.n;
        view.keys = db.keys.data();
        if (sets) {
            setId.resize(db.n);
            pos.resize(db.n);
            strand.resize(db.n);
            for (uint32_t i = 0; i < db.n; i++) {
                const uint32_t k = db.keys[i];
                setId[i] = k < sets->setOfKey.size() ? sets->setOfKey[k] : 0;
                pos[i] = k < sets->posOfKey.size() ? sets->posOfKey[k] : 0;
                strand[i] = k < sets->strandOfKey.size() ? sets->strandOfKey[k] : 0;
            }
            view.setId = setId.data();
            view.posInSet = pos.data();
            view.strand = strand.data();
            view.nSets = sets->nSets;
        }
        if (db.profile) {
            view.alnProfile = db.alnProfile.data();
            view.sortedScore = db.sortedScore.data();
            view.sortedIndex = db.sortedIndex.data();
        }
    }
};

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

void packNames(const std::vector<std::string> &v, std::string &blob, std::vector<uint64_t> &off) {
    off.assign(v.size() + 1, 0);
    for (size_t i = 0; i < v.size(); i++) off[i + 1] = off[i] + v[i].size();
    blob.clear();
    blob.reserve(off.back());
    for (const std::string &s : v) blob += s;
}

bool waitForFile(const std::string &path, int seconds) {
    for (int i = 0; i < seconds * 10; i++) {
        if (sddb::fileExists(path)) return true;
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    return false;
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
    int rc = sd_search_create(device, &par, &tv.view, &S.s);
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
    std::vector<sd_search_result *> results(std::max<size_t>(rb.size(), 1), nullptr);
    if (!rb.empty()) {
        rc = sd_search_stream(S.s, &qv.view, sameDb ? 1 : 0, (uint32_t) rb.size(), rb.data(), re.data(), results.data());
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
    sd_search_stats(S.s, st, tm);
    info(a, "%llu prefilter hits, %llu pairs aligned; prefilter %.2f s | align %.2f s | aggregate %.2f s | clusterhits %.2f s | total %.2f s\n",
         (unsigned long long) st[4], (unsigned long long) st[5], tm[3], tm[6], tm[8], tm[9], tm[11]);
    if (!withClusters) return 0;

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
    const std::string part = world > 1 ? tmpDir + "/out.tsv.part" + std::to_string(rank) : a.pos[2];
    ::remove(part.c_str());
    {   // an empty result still leaves a file
        FILE *f = fopen(part.c_str(), "w");
        if (!f) return fail("cannot create " + part);
        fclose(f);
    }
    uint64_t key = 0, nClu = 0, nHit = 0;
    for (size_t r = 0; r < rb.size(); r++) {
        uint64_t nc = 0, nh = 0;
        rc = sd_search_result_write_tsv(results[r], part.c_str(), qn.data(), qno.data(), tn.data(), tno.data(), qsrc.data(), qso.data(),
                                        tsrc.data(), tso.data(), 0, 1, key, &nc, &nh);
        if (rc != SD_OK) return fail("sd_search_result_write_tsv failed (" + std::to_string(rc) + ")");
        key += nc;
        nClu += nc;
        nHit += nh;
    }
    info(a, "%llu clusters with %llu hits written%s\n", (unsigned long long) nClu, (unsigned long long) nHit,
         world > 1 ? (" by rank " + std::to_string(rank)).c_str() : "");
    if (world > 1) {
        {   // this rank is done
            FILE *f = fopen((tmpDir + "/out.tsv.done" + std::to_string(rank)).c_str(), "w");
            if (f) {
                fprintf(f, "%llu\n", (unsigned long long) nClu);
                fclose(f);
            }
        }
        if (rank == 0) {
            FILE *out = fopen(a.pos[2].c_str(), "w");
            if (!out) return fail("cannot create " + a.pos[2]);
            uint64_t base = 0;
            std::vector<char> line;
            for (int r = 0; r < world; r++) {
                if (!waitForFile(tmpDir + "/out.tsv.done" + std::to_string(r), 24 * 3600)) return fail("rank " + std::to_string(r) + " did not finish");
                // cluster keys are renumbered so that the merged file counts them like one run would
                FILE *in = fopen((tmpDir + "/out.tsv.part" + std::to_string(r)).c_str(), "r");
                if (!in) return fail("part of rank " + std::to_string(r) + " is missing");
                char *l = nullptr;
                size_t cap = 0;
                ssize_t n;
                uint64_t seen = 0;
                while ((n = getline(&l, &cap, in)) > 0) {
                    if (l[0] == '#') {
                        const char *tab = strchr(l, '\t');
                        fprintf(out, "#%llu%s", (unsigned long long) (base + seen), tab ? tab : "\n");
                        seen++;
                    } else {
                        fwrite(l, 1, (size_t) n, out);
                    }
                }
                free(l);
                fclose(in);
                base += seen;
            }
            fclose(out);
            for (int r = 0; r < world; r++) {
                ::remove((tmpDir + "/out.tsv.done" + std::to_string(r)).c_str());
                ::remove((tmpDir + "/out.tsv.part" + std::to_string(r)).c_str());
            }
        }
    }
    return 0;
}

}  // namespace

int searchModule(const Args &a) { return runSearch(a, false); }
int clustersearchModule(const Args &a) { return runSearch(a, true); }

int result2profileModule(const Args &) { return fail("not built yet"); }
int subtractdbsModule(const Args &) { return fail("not built yet"); }
int mergedbsModule(const Args &) { return fail("not built yet"); }

}  // namespace sdcli
