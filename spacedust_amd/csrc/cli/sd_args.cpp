// see sd_args.h.  The flag names are the reference's (PARAM_* tables of M/src/commons/Parameters.cpp:30-330 and
// R/src/commons/LocalParameters.h:32-150), so that any parameter string its workflows build parses here; what a module
// does not implement it rejects by value in the module, never silently.
#include "sd_args.h"

#include <cstdlib>
#include <cstring>
#include <set>

namespace sdcli {

namespace {

const char *const kBoolFlags[] = {
    "--add-orf-stop", "--add-self-matches", "--adjust-kmer-len", "--allow-deletion", "--beats-first",
    "--cluster-reassign", "--cluster-use-weight", "--db-output", "--diag-score", "--exhaustive-search",
    "--filter-hits", "--filter-self-match", "--first-seq-as-repr", "--force-reuse", "--full-header",
    "--greedy-best-hits", "--help", "--hh-format", "--ignore-multi-kmer", "--include-only-extendable",
    "--lca-search", "--merge-stop-empty", "--positive-filter", "--preserve-keys", "--profile-cluster-search",
    "--realign", "--recover-deleted", "--remove-tmp-files", "--set-mode", "--short-output", "--shuffle",
    "--simple-best-hit", "--single-step-clustering", "--skip-query", "--split-aa", "--take-larger-entry",
    "--touch-lock", "--trim-to-one-column", "--tsv", "--use-all-table-starts", "--use-fasta-header",
    "--use-header-file", "--use-seq-id", "--wg", "--wrapped-scoring", "-a", "-h",
};
const char *const kValueFlags[] = {
    "--aggregation-mode", "--alignment-mode", "--alignment-output-mode", "--alph-size", "--alpha", "--alt-ali",
    "--blacklist", "--blocklen", "--chain-alignments", "--check-compatible", "--cluster-mode", "--cluster-pval",
    "--cluster-size", "--cluster-steps", "--cluster-weight-threshold", "--column-to-take", "--comp-bias-corr",
    "--comp-bias-corr-scale", "--comparison-operator", "--comparison-value", "--compressed", "--contig-end-mode",
    "--contig-start-mode", "--corr-score-weight", "--cov", "--cov-mode", "--create-lookup", "--createdb-mode",
    "--db-load-mode", "--db-suffix-list", "--dbtype", "--diff", "--disk-space-limit", "--e-profile",
    "--exact-kmer-matching", "--exhaustive-search-filter", "--expand-filter-clusters", "--expansion-mode",
    "--extended-dbtype", "--extract-lines", "--extract-mode", "--file-exclude", "--file-include",
    "--filter-column", "--filter-expression", "--filter-file", "--filter-min-enable", "--filter-msa",
    "--filter-regex", "--foldseek-path", "--format-mode", "--format-output", "--forward-frames",
    "--fwbw-backtrace-mode", "--fwbw-gapextend", "--fwbw-gapopen", "--gap-extend", "--gap-open", "--gap-pc",
    "--gff-dir", "--gff-type", "--gpu", "--gpu-server", "--gpu-server-wait-timeout", "--hash-shift",
    "--header-type", "--headers-split-mode", "--id-list", "--id-mode", "--id-offset", "--identifier-field",
    "--idx-entry-type", "--idx-seq-src", "--index-dbsuffix", "--index-subset", "--join-db", "--k-score",
    "--kb-columns", "--kmer-per-seq", "--kmer-per-seq-scale", "--lca-mode", "--lca-ranks", "--local-tmp", "--mact",
    "--majority", "--mapping-file", "--mask", "--mask-lower-case", "--mask-n-repeat", "--mask-prob",
    "--mask-profile", "--match-mode", "--match-ratio", "--max-accept", "--max-gaps", "--max-gene-gap",
    "--max-iterations", "--max-length", "--max-rejected", "--max-seq-id", "--max-seq-len", "--max-seqs",
    "--max-sequences", "--merge-query", "--min-aln-len", "--min-length", "--min-seq-id", "--min-sequences",
    "--min-ungapped-score", "--mpi-runner", "--msa-format-mode", "--msa-type", "--multihit-pval",
    "--ncbi-tax-dump", "--neff", "--num-iterations", "--orf-filter", "--orf-filter-e", "--orf-filter-s",
    "--orf-start-mode", "--output-dbtype", "--overlap", "--pairing-dummy-mode", "--pairing-filter",
    "--pairing-mode", "--pairing-prox-dist", "--pca", "--pcb", "--pick-id-from", "--pick-n-sim-kmer",
    "--prefilter-mode", "--prefix", "--prefixes", "--profile-output-mode", "--pseudo-cnt-mode", "--qid", "--qsc",
    "--realign-max-seqs", "--realign-score-bias", "--report-mode", "--rescore-mode", "--result-direction",
    "--reverse-frames", "--score-bias", "--search-mode", "--search-type", "--seed-sub-mat", "--sens-steps",
    "--seq-id-mode", "--sequence-overlap", "--sequence-split-mode", "--similarity-type", "--sort-entries",
    "--sort-results", "--spaced-kmer-mode", "--spaced-kmer-pattern", "--split", "--split-memory-limit",
    "--split-mode", "--start-sens", "--stat", "--strand", "--sub-mat", "--subdb-mode", "--suboptimal-hits",
    "--summary-prefix", "--tar-exclude", "--tar-include", "--target-column", "--target-search-mode", "--tau",
    "--tax-db-mode", "--tax-lineage", "--tax-mapping-file", "--tax-mapping-mode", "--tax-output-mode",
    "--taxon-list", "--temperature", "--threads", "--translate", "--translation-mode", "--translation-table",
    "--unpack-name-mode", "--unpack-suffix", "--vote-mode", "--weights", "--write-lookup", "--zdrop", "-c", "-e",
    "-k", "-s", "-v",
    // this binary's own (multi-GPU launch and tuning; not reference flags)
    "--device", "--rank", "--world-size", "--chunk-queries", "--bin-size", "--l2-cache-size", "--keep-dbs", "--keep-tmp", "--comm-port",
    "--profile-weights-host",
};

const std::set<std::string> &boolFlags() {
    static const std::set<std::string> s(kBoolFlags, kBoolFlags + sizeof(kBoolFlags) / sizeof(kBoolFlags[0]));
    return s;
}
const std::set<std::string> &valueFlags() {
    static const std::set<std::string> s(kValueFlags, kValueFlags + sizeof(kValueFlags) / sizeof(kValueFlags[0]));
    return s;
}

bool parseBool(const std::string &v) {   // Parameters::parseBool: TRUE/true/1 ...
    return v == "1" || v == "true" || v == "TRUE" || v == "True" || v == "yes" || v == "on";
}

}  // namespace

bool Args::parse(int argc, const char **argv, std::string *err) {
    int i = 0;
    // positional arguments first: everything up to the first token starting with '-'
    for (; i < argc; i++) {
        if (argv[i][0] == '-' && argv[i][1] != '\0') break;
        pos.push_back(argv[i]);
    }
    for (; i < argc; i++) {
        const std::string f = argv[i];
        if (boolFlags().count(f)) {
            if (i + 1 == argc || argv[i + 1][0] == '-') {
                const bool cur = opt.count(f) ? parseBool(opt[f]) : false;
                opt[f] = cur ? "0" : "1";   // toggles the default (all of these default to false except where noted by the module)
                opt[f + "#toggled"] = "1";
            } else {
                opt[f] = parseBool(argv[i + 1]) ? "1" : "0";
                i++;
            }
        } else if (valueFlags().count(f)) {
            if (i + 1 == argc) {
                if (err) *err = "Missing argument " + f;
                return false;
            }
            opt[f] = argv[++i];
        } else {
            if (err) *err = "Unrecognized parameter \"" + f + "\"";
            return false;
        }
    }
    return true;
}

std::string Args::str(const std::string &f, const std::string &def) const {
    const std::map<std::string, std::string>::const_iterator it = opt.find(f);
    return it == opt.end() ? def : it->second;
}

long long Args::integer(const std::string &f, long long def) const {
    const std::map<std::string, std::string>::const_iterator it = opt.find(f);
    return it == opt.end() ? def : strtoll(it->second.c_str(), nullptr, 10);
}

double Args::real(const std::string &f, double def) const {
    const std::map<std::string, std::string>::const_iterator it = opt.find(f);
    return it == opt.end() ? def : strtod(it->second.c_str(), nullptr);
}

bool Args::flag(const std::string &f, bool def) const {
    const std::map<std::string, std::string>::const_iterator it = opt.find(f);
    if (it == opt.end()) return def;
    if (opt.count(f + "#toggled")) return !def;   // a bare boolean flag flips the module's default
    return parseBool(it->second);
}

std::string Args::multi(const std::string &f, const std::string &tag, const std::string &def) const {
    const std::map<std::string, std::string>::const_iterator it = opt.find(f);
    if (it == opt.end()) return def;
    const std::string &v = it->second;
    if (v.find(':') == std::string::npos) return v;
    size_t p = 0;
    while (p < v.size()) {
        size_t e = v.find(',', p);
        if (e == std::string::npos) e = v.size();
        const std::string item = v.substr(p, e - p);
        const size_t c = item.find(':');
        if (c != std::string::npos && item.substr(0, c) == tag) return item.substr(c + 1);
        p = e + 1;
    }
    return def;
}

std::vector<std::string> Args::flagsAsArgv() const {
    std::vector<std::string> out;
    for (std::map<std::string, std::string>::const_iterator it = opt.begin(); it != opt.end(); ++it) {
        if (it->first.find('#') != std::string::npos) continue;
        out.push_back(it->first);
        out.push_back(it->second);
    }
    return out;
}

}  // namespace sdcli
