// sdgpu -- multi-call host driver of the MI355X-native clustersearch hot path.  `sdgpu <module> <args>` accepts the
// command lines the reference's workflow scripts issue as `$MMSEQS <module> <args>` (R/src/spacedust.cpp:26-114 is
// the reference's command table; R/data/clustersearch.sh and M/data/workflow/blastp.sh / blastpgp.sh are the callers), so a
// wrapper that forwards the hot modules here and everything else to the reference binary runs those scripts unchanged.
#include "sd_cli.h"
#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

using namespace sdcli;

namespace {
struct Module {
    const char *name;
    int (*fn)(const Args &);
    const char *what;
};
const Module kModules[] = {
    {"prefilter", prefilterModule, "k-mer prefilter on the GPU: <queryDB> <targetDB> <prefilterDB>"},
    {"align", alignModule, "Smith-Waterman alignments on the GPU: <queryDB> <targetDB> <prefilterDB> <alignmentDB>"},
    {"clusterhits", clusterhitsModule, "agglomerative hit clustering on the GPU: <querySetDB> <targetSetDB> <matchesDB> <clustersDB>"},
    {"search", searchModule, "prefilter + align (and the --num-iterations profile loop): <queryDB> <targetDB> <alignmentDB> <tmpDir>"},
    {"clustersearch", clustersearchModule, "the whole --search-mode 0 workflow in one process: <querySetDB> <targetSetDB> <out.tsv> <tmpDir>"},
    {"prefixid", prefixidModule, "host glue: prefix every line of a result DB with its entry key"},
    {"besthitbyset", besthitbysetModule, "host glue: best hit per (query protein, target set)"},
    {"mergeresultsbyset", mergeresultsbysetModule, "host glue: concatenate the entries of a set's members"},
    {"combinehits", combinehitsModule, "host glue: per (query set, target set) hit lists + multihit E-value"},
    {"summarizeresults", summarizeresultsModule, "host glue: clusters DB -> final TSV"},
    {"result2profile", result2profileModule, "host: alignment DB -> profile DB (between search iterations)"},
    {"subtractdbs", subtractdbsModule, "host glue: remove from result DB A the targets listed in result DB B"},
    {"mergedbs", mergedbsModule, "host glue: merge the entries of several DBs by key"},
    {"createindex", createindexModule, "target k-mer index built once, in the reference's createindex file layout: <sequenceDB> <tmpDir>"},
    {"createsetdb", createsetdbModule, "FASTA files (Prodigal headers) -> setDB in the createsetdb layout"},
};
}  // namespace

int main(int argc, const char **argv) {
    // The host stages' OpenMP teams wait between parallel regions while a kernel runs or another stage has the CPUs; libgomp's default
    // is to spin there (round 6, `clustersearch --num-iterations 3` on 4 query proteomes: 98 CPU-seconds inside 6.6 s of the alignment
    // stage, 89 inside 5.8 s of result2profile -- sixteen threads spinning).  The policy is read when libgomp is loaded, i.e. before
    // main: when the caller has not chosen one, choose the passive one and start over (spacedust_amd/cpus.py does the same for Python).
    if (!getenv("OMP_WAIT_POLICY") && !getenv("GOMP_SPINCOUNT") && !getenv("SD_NO_REEXEC")) {
        setenv("OMP_WAIT_POLICY", "passive", 1);
        setenv("GOMP_SPINCOUNT", "0", 1);
        setenv("SD_NO_REEXEC", "1", 1);
        execv("/proc/self/exe", (char *const *) argv);
        // (exec failed: go on with the default policy)
    }
    if (argc < 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) {
        printf("sdgpu: MI355X-native hot path of spacedust clustersearch --search-mode 0\n\nusage: sdgpu <module> <args>\n\n");
        for (const Module &m : kModules) printf("  %-18s %s\n", m.name, m.what);
        return argc < 2 ? 1 : 0;
    }
    for (const Module &m : kModules) {
        if (strcmp(argv[1], m.name) != 0) continue;
        Args a;
        a.module = m.name;
        std::string err;
        if (!a.parse(argc - 2, argv + 2, &err)) {
            fprintf(stderr, "sdgpu %s: %s\n", m.name, err.c_str());
            return 1;
        }
        info(a, "");
        // OpenMP teams of the host stages: --threads, else the cgroup CPU quota (a 256-thread default team on a 16-CPU quota
        // spends its time being descheduled)
        omp_set_num_threads(threadsOf(a));
        return m.fn(a);
    }
    fprintf(stderr, "sdgpu: unknown module \"%s\" (sdgpu --help lists the modules)\n", argv[1]);
    return 1;
}
