"""The drop-in boundary on the GPU: the reference's own module command lines (copied from the log of
`spacedust clustersearch genome genome result.tsv tmp --filter-self-match`, i.e. what R/data/clustersearch.sh:110-152 and
M/data/workflow/blastp.sh:70,85 pass to `$MMSEQS prefilter|align|...`) run through the `sdgpu` binary on DB files, and every
DB must hash to what the reference binary's DB hashes to:
    pref_0 8109a70b... (98 957 lines), result 2e917f0e... (15 065 lines), final TSV (cut -f2-) abb28ee3... (416 lines),
    R/util/run_regression.sh:20-23: 308 hit lines, 2 clusters with P < 1E-20."""
import os
import subprocess

import pytest

from dbutil import write_db, flat_lines_from_gz, entries_by_first_column, sorted_md5, sdgpu, example_fasta, SDGPU

pytestmark = pytest.mark.gpu

PREFILTER_PAR = ("--sub-mat aa:blosum62.out,nucl:nucleotide.out --seed-sub-mat aa:VTML80.out,nucl:nucleotide.out -k 0 "
                 "--target-search-mode 0 --k-score seq:2147483647,prof:2147483647 --alph-size aa:21,nucl:5 --max-seq-len 65535 "
                 "--max-seqs 300 --split 0 --split-mode 2 --split-memory-limit 0 -c 0.8 --cov-mode 2 --comp-bias-corr 1 "
                 "--comp-bias-corr-scale 1 --diag-score 1 --exact-kmer-matching 0 --mask 1 --mask-prob 0.9 --mask-lower-case 0 "
                 "--mask-n-repeat 0 --min-ungapped-score 15 --add-self-matches 0 --spaced-kmer-mode 1 --db-load-mode 0 "
                 "--pca substitution:1.100,context:1.400 --pcb substitution:4.100,context:5.800 --threads 8 --compressed 0 -v 3 "
                 "-s 5.7").split()
ALIGN_PAR = ("--sub-mat aa:blosum62.out,nucl:nucleotide.out -a 1 --alignment-mode 2 --alignment-output-mode 0 --wrapped-scoring 0 "
             "-e 10 --min-seq-id 0 --min-aln-len 30 --seq-id-mode 0 --alt-ali 0 -c 0.8 --cov-mode 2 --max-seq-len 65535 "
             "--comp-bias-corr 1 --comp-bias-corr-scale 1 --max-rejected 2147483647 --max-accept 2147483647 --add-self-matches 0 "
             "--db-load-mode 0 --pca substitution:1.100,context:1.400 --pcb substitution:4.100,context:5.800 --score-bias 0 "
             "--realign 0 --realign-score-bias -0.2 --realign-max-seqs 2147483647 --corr-score-weight 0 --gap-open aa:11,nucl:5 "
             "--gap-extend aa:1,nucl:2 --zdrop 40 --threads 8 --compressed 0 -v 3").split()
CLUSTERHITS_PAR = ("--multihit-pval 0.01 --cluster-pval 0.01 --max-gene-gap 3 --cluster-size 2 --cluster-use-weight 0 --db-output 1 "
                   "--alpha 1 --threads 8 --compressed 0 -v 3").split()


@pytest.fixture(scope='module')
def work(tmp_path_factory):
    tmp = tmp_path_factory.mktemp('dropin')
    fa = example_fasta(tmp)
    sdgpu('createsetdb', fa[0], fa[1], tmp / 'genome', tmp / 'tmp', '-v', '0')
    return tmp


def flat(work, db):
    sdgpu('prefixid', work / db, work / (db + '.flat'), '--tsv', '--threads', '1')
    return open(work / (db + '.flat')).readlines()


def test_module_by_module_reproduces_reference_dbs(work):
    g = work / 'genome'
    sdgpu('prefilter', g, g, work / 'pref_0', *PREFILTER_PAR)
    lines = flat(work, 'pref_0')
    assert (len(lines), sorted_md5(lines)) == (98957, '8109a70bdea70ee10e0dbd27ba6b7e37')
    sdgpu('align', g, g, work / 'pref_0', work / 'result', *ALIGN_PAR)
    lines = flat(work, 'result')
    assert (len(lines), sorted_md5(lines)) == (15065, '2e917f0e9782e8a7412c7360aa7bf1b4')
    # the written entries equal the fixture from the real reference classes in DB order too (entry order, line order)
    assert lines == flat_lines_from_gz('config1_aln.tsv.gz')
    sdgpu('prefixid', work / 'result', work / 'result_prefixed', '--threads', '8', '-v', '3')
    sdgpu('besthitbyset', g, g, work / 'result_prefixed', work / 'aggregate', '--simple-best-hit', '1', '--suboptimal-hits', '0',
          '--threads', '8', '--compressed', '0', '-v', '3')
    sdgpu('mergeresultsbyset', str(g) + '_set_to_member', work / 'aggregate', work / 'aggregate_merged', '--threads', '8', '-v', '3')
    sdgpu('combinehits', g, g, work / 'aggregate_merged', work / 'matches', work / 'tmp', '--alpha', '1', '--aggregation-mode', '0',
          '--filter-self-match', '1', '--threads', '8', '--compressed', '0', '-v', '3')
    sdgpu('clusterhits', g, g, work / 'matches', work / 'clusters', *CLUSTERHITS_PAR)
    sdgpu('summarizeresults', g, g, work / 'clusters', work / 'result.tsv', '--threads', '8', '-v', '3')
    tsv = open(work / 'result.tsv').readlines()
    n_hit = sum(1 for l in tsv if l.startswith('>'))
    clu = [l for l in tsv if l.startswith('#')]
    assert (n_hit, len(clu), sum(1 for l in clu if float(l.split('\t')[3]) < 1e-20)) == (308, 108, 2)   # run_regression.sh:20-23
    assert sorted_md5(tsv, drop_first_column=True) == 'abb28ee37bc130a5f09a9f767ef00ccf'


def test_align_on_the_reference_prefilter_db(work):
    """`align` alone on the prefilter DB of the real reference classes (fixture), read from split data files"""
    pref = flat_lines_from_gz('config1_pref.tsv.gz')
    write_db(str(work / 'pref_ref'), entries_by_first_column(pref, 5898), 7, splits=8)
    g = work / 'genome'
    sdgpu('align', g, g, work / 'pref_ref', work / 'result_ref', *ALIGN_PAR)
    lines = flat(work, 'result_ref')
    assert (len(lines), sorted_md5(lines)) == (15065, '2e917f0e9782e8a7412c7360aa7bf1b4')


def test_no_device_no_result(work):
    """without a visible GPU the hot modules exit non-zero: there is no CPU path behind them"""
    import subprocess
    from dbutil import SDGPU
    env = dict(os.environ, HIP_VISIBLE_DEVICES='-1', ROCR_VISIBLE_DEVICES='-1')
    g = str(work / 'genome')
    p = subprocess.run([SDGPU, 'prefilter', g, g, str(work / 'pref_none')] + PREFILTER_PAR, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and 'no usable HIP device' in p.stderr


def test_fused_clustersearch_and_search_modules(work):
    """`sdgpu clustersearch` (the whole workflow in one process around the C++ pipeline object) writes the reference's TSV,
    and with --keep-dbs 1 the prefilter / alignment DBs its sinks write hash like the reference binary's; `sdgpu search`
    (prefilter + align, blastp.sh) writes the same alignment DB"""
    g = work / 'genome'
    sdgpu('clustersearch', g, g, work / 'fused.tsv', work / 'tmpf', '--filter-self-match', '--keep-dbs', '1', '--threads', '8')
    tsv = open(work / 'fused.tsv').readlines()
    clu = [l for l in tsv if l.startswith('#')]
    assert (sum(1 for l in tsv if l.startswith('>')), len(clu), sum(1 for l in clu if float(l.split('\t')[3]) < 1e-20)) == (308, 108, 2)
    assert sorted_md5(tsv, drop_first_column=True) == 'abb28ee37bc130a5f09a9f767ef00ccf'
    assert [int(l.split('\t')[0][1:]) for l in clu] == list(range(108))
    for db, n, md5 in (('tmpf/pref_0', 98957, '8109a70bdea70ee10e0dbd27ba6b7e37'), ('tmpf/result', 15065, '2e917f0e9782e8a7412c7360aa7bf1b4')):
        sdgpu('prefixid', work / db, work / 'x.flat', '--tsv')
        lines = open(work / 'x.flat').readlines()
        assert (len(lines), sorted_md5(lines)) == (n, md5), db
    sdgpu('search', g, g, work / 'res_search', work / 'tmps', '-a', '1', '--alignment-mode', '2', '-e', '10', '--min-aln-len', '30',
          '-c', '0.8', '--cov-mode', '2', '-s', '5.7', '--max-seqs', '300', '--threads', '8')
    sdgpu('prefixid', work / 'res_search', work / 'y.flat', '--tsv')
    lines = open(work / 'y.flat').readlines()
    assert (len(lines), sorted_md5(lines)) == (15065, '2e917f0e9782e8a7412c7360aa7bf1b4')
    # without --filter-self-match (the reference's default) the self pairs of the two genomes stay in
    sdgpu('clustersearch', g, g, work / 'noself.tsv', work / 'tmpn')
    assert sum(1 for l in open(work / 'noself.tsv') if l.startswith('#')) > 108


def test_align_realign_reproduces_reference_db(work):
    """`align --realign 1 -e 0.001` (iteration 0 of `search --num-iterations`, M/src/workflow/Search.cpp:484-486): score-only
    first pass without coverage, second pass with the score-biased matrix, scores / E-values of the first pass kept
    (Alignment.cpp:45-57,408-440).  md5 of the reference binary's own aln_0 on the regression input (12 919 lines)."""
    pref = flat_lines_from_gz('config1_pref.tsv.gz')
    if not os.path.exists(work / 'pref_ref.index'):
        write_db(str(work / 'pref_ref'), entries_by_first_column(pref, 5898), 7, splits=8)
    g = work / 'genome'
    par = [x for x in ALIGN_PAR]
    par[par.index('-e') + 1] = '0.001'
    par[par.index('--realign') + 1] = '1'
    sdgpu('align', g, g, work / 'pref_ref', work / 'aln_0', *par)
    lines = flat(work, 'aln_0')
    assert (len(lines), sorted_md5(lines)) == (12919, '5d4e3228b0afb4378a0364a0291627f4')


def _key_ordered_md5(path):
    import hashlib
    from dbutil import read_db
    d = read_db(str(path))
    h = hashlib.md5()
    for k in sorted(d):
        h.update(d[k])
    return len(d), h.hexdigest()


def _reference_profiles_on_this_box(aln_lines):
    """profile records the reference's own MultipleAlignment / MsaFilter / PSSMCalculator classes (libsdref_r2p.so) compute
    on THIS machine from an alignment DB: PSSMCalculator takes reciprocals with rcpps (PSSMCalculator.cpp:497-504), whose
    low bits are implementation-defined and differ between CPU vendors, so profile bytes are only comparable on one box"""
    import gzip
    from dbutil import GOLD
    from oracle.pyoracle import RefResult2Profile
    from spacedust_amd import api
    seqs = []
    for f in ('NC_000913.faa', 'NC_000915.faa'):
        cur = None
        for line in gzip.open(os.path.join(GOLD, 'examples', f + '.gz'), 'rt'):
            line = line.rstrip('\n')
            if line.startswith('>'):
                if cur is not None:
                    seqs.append(''.join(cur))
                cur = []
            else:
                cur.append(line)
        seqs.append(''.join(cur))
    by = {}
    for l in aln_lines:
        w = l.rstrip('\n').split('\t')
        by.setdefault(int(w[0]), []).append(w)
    ref = RefResult2Profile()
    out = {}
    for q in range(len(seqs)):
        et, eq, ets, bts = [], [], [], []
        for w in by.get(q, []):
            if int(w[1]) == q or not (float(w[4]) < 0.001):
                continue
            et.append(int(w[1]))
            eq.append(int(w[5]))
            ets.append(int(w[8]))
            bts.append(api.uncompress_cigar(w[11]))
        out[q] = ref.profile(seqs[q], [seqs[t] for t in et], eq, ets, bts)
    return out


def _example_sequences():
    import gzip
    from dbutil import GOLD
    seqs = []
    for f in ('NC_000913.faa', 'NC_000915.faa'):
        cur = None
        for line in gzip.open(os.path.join(GOLD, 'examples', f + '.gz'), 'rt'):
            line = line.rstrip('\n')
            if line.startswith('>'):
                if cur is not None:
                    seqs.append(''.join(cur))
                cur = []
            else:
                cur.append(line)
        seqs.append(''.join(cur))
    return seqs


ITER_COMMON = ['--threads', '8', '-v', '0']
ITER_PREF = ITER_COMMON + ['-s', '5.7', '-k', '0', '--max-seqs', '300', '-c', '0.8', '--cov-mode', '2', '--min-ungapped-score', '15', '--mask', '1',
                           '--mask-prob', '0.9', '--comp-bias-corr', '1']
ITER_ALN = ITER_COMMON + ['-a', '1', '--alignment-mode', '2', '--min-aln-len', '30', '-c', '0.8', '--cov-mode', '2', '--min-seq-id', '0',
                          '--comp-bias-corr', '1', '--realign-score-bias', '-0.2']
ITER_PROF = ITER_COMMON + ['-e', '0.001', '--e-profile', '0.001', '--mask-profile', '1', '--comp-bias-corr', '1', '--filter-msa', '1',
                           '--filter-min-enable', '0', '--max-seq-id', '0.9', '--qid', '0.0', '--qsc', '-20', '--cov', '0', '--diff', '1000',
                           '--pca', 'substitution:1.100,context:1.400', '--pcb', 'substitution:4.100,context:5.800']


def _compress(bt):
    """Matcher::compressAlignment (Matcher.cpp:166-185)"""
    out, state, count = [], 'M', 0
    for ch in bt:
        if ch != state:
            out.append('%d%s' % (count, state))
            state, count = ch, 1
        else:
            count += 1
    out.append('%d%s' % (count, state))
    return ''.join(out)


def _lines(db):
    return {k: [l.split('\t') for l in v.decode().split('\n') if l] for k, v in db.items()}


def test_profile_iterations_pinned_to_the_reference_classes_on_this_box(work):
    """Iterations 1 and 2 of BASELINE config 4 (M/data/workflow/blastpgp.sh:73-133) on all 5 898 queries of the regression input,
    module by module, every step against the reference's own classes run on THIS machine -- nothing here depends on the CPU model
    (PSSMCalculator's rcpps makes profile bytes vendor-specific, so recorded checksums cannot serve):
      prefilter   profile_k vs genome      == QueryMatcher driven with the DBTYPE_HMM_PROFILE Sequence (libsdref), row for row
      align       profile_k on those rows  == Matcher::getSWResult with the profile query (libsdref): targets, coordinates,
                                              backtraces, E-values
      result2profile                       == MultipleAlignment / MsaFilter / PSSMCalculator (libsdref_r2p), byte for byte
    and the DBs of the fused `clustersearch --num-iterations 3` equal the ones these steps produce."""
    import numpy as np
    from dbutil import read_db
    from oracle.pyoracle import ref_available, ref_r2p_available, Ref, RefSW, RefResult2Profile
    from spacedust_amd import api
    if not (ref_available() and ref_r2p_available()):
        pytest.skip('oracle/_ref/libsdref*.so not built (needs /root/reference at build time)')
    g = work / 'genome'
    if not os.path.exists(work / 'aln_0.index'):
        test_align_realign_reproduces_reference_db(work)
    seqs = _example_sequences()
    n = len(seqs)
    lens = np.array([len(x) for x in seqs])
    blob = ''.join(seqs).encode()
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    ref = Ref(6)
    rix = ref.index(blob, off, kmer_thr=0)                      # profile searches index every target k-mer (Prefiltering.cpp:525-527)
    host = api.Host()
    thr = host.profile_kmer_threshold(5.7, 6)
    rpf = rix.prefilter_profile(int(lens.max()) + 10, thr, max_hits=300)
    rsw = RefSW(ref, int(lens.max()) + 10, int(off[-1]))
    r2p = RefResult2Profile()
    it = work / 'it'
    os.makedirs(it, exist_ok=True)
    sdgpu('result2profile', g, g, work / 'aln_0', it / 'profile_0', *ITER_PROF)
    prev_aln = work / 'aln_0'
    n_rows = n_aln = 0
    for step in (1, 2):
        last = step == 2
        prof_db = it / ('profile_%d' % (step - 1))
        prof = read_db(str(prof_db))
        assert len(prof) == n and all(len(prof[q]) == 25 * lens[q] for q in range(n))
        # ---- prefilter with the profile
        sdgpu('prefilter', prof_db, g, it / ('pref_tmp_%d' % step), *ITER_PREF)
        got = _lines(read_db(str(it / ('pref_tmp_%d' % step))))
        for q in range(n):
            ids, sc, dg, _ = rpf.query(prof[q])
            keep = (lens[ids].astype(np.float32) / np.float32(lens[q])) >= np.float32(0.8)   # Util::canBeCovered, --cov-mode 2
            want = [(int(t), int(s), int(np.int16(np.uint16(d)))) for t, s, d in zip(ids[keep], sc[keep], dg[keep])]
            mine = [(int(w[0]), int(w[1]), int(w[2])) for w in got.get(q, [])]
            assert mine == want, (step, q, mine[:3], want[:3])
            n_rows += len(want)
        # ---- minus what is aligned already, then the profile alignments
        sdgpu('subtractdbs', it / ('pref_tmp_%d' % step), prev_aln, it / ('pref_%d' % step), *ITER_COMMON, '--e-profile', '0.001', '-e', '10')
        todo = _lines(read_db(str(it / ('pref_%d' % step))))
        sdgpu('align', prof_db, g, it / ('pref_%d' % step), it / ('aln_tmp_%d' % step), *ITER_ALN, '-e', '10' if last else '0.001', '--realign', '0')
        aln = _lines(read_db(str(it / ('aln_tmp_%d' % step))))
        for q in range(n):
            rows = todo.get(q, [])
            if not rows:
                assert not aln.get(q), (step, q)
                continue
            rsw.set_query_profile(prof[q])
            want = {}
            for w in rows:
                t = int(w[0])
                r = rsw.align(seqs[t], sw_mode=2, eval_thr=10.0 if last else 0.001, cov_mode=2, cov_thr=0.8)
                aln_len = len(r['backtrace'])
                if r['btLen'] > 0 and r['evalue'] <= (10.0 if last else 0.001) and aln_len >= 30:
                    want[t] = (r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], _compress(r['backtrace']), '%.3E' % r['evalue'])
            mine = {int(w[0]): (int(w[4]), int(w[5]), int(w[7]), int(w[8]), w[10], w[3]) for w in aln.get(q, [])}
            assert mine == want, (step, q, sorted(set(mine) ^ set(want))[:5])
            n_aln += len(want)
        # ---- merge, next profile
        merged = it / ('aln_%d' % step)
        sdgpu('mergedbs', prof_db, merged, prev_aln, it / ('aln_tmp_%d' % step))
        prev_aln = merged
        if not last:
            sdgpu('result2profile', prof_db, g, merged, it / ('profile_%d' % step), *ITER_PROF)
            nxt = read_db(str(it / ('profile_%d' % step)))
            by = _lines(read_db(str(merged)))
            bad = 0
            for q in range(n):
                et, eq, ets, bts = [], [], [], []
                for w in by.get(q, []):
                    if not (float(w[3]) < 0.001):
                        continue   # (the centre is a profile DB entry: not "the same database", its own sequence stays in)
                    et.append(int(w[0]))
                    eq.append(int(w[4]))
                    ets.append(int(w[7]))
                    bts.append(api.uncompress_cigar(w[10]))
                bad += r2p.profile(None, [seqs[t] for t in et], eq, ets, bts, centre_profile=prof[q]) != nxt[q]
            assert bad == 0, (step, bad)
    assert n_rows > 150000 and n_aln > 1000
    # the fused workflow went through the same DBs
    # (--keep-tmp 1: the module chain with its per-iteration DBs; without it the iterations run in memory, next test but one)
    sdgpu('clustersearch', g, g, work / 'iter_pinned.tsv', work / 'tmpip', '--filter-self-match', '--num-iterations', '3', '--threads', '8', '-v', '0',
          '--keep-tmp', '1')
    for name in ('profile_0', 'profile_1'):
        assert _key_ordered_md5(work / 'tmpip' / 'search' / name) == _key_ordered_md5(it / name), name
    assert _key_ordered_md5(work / 'tmpip' / 'result') == _key_ordered_md5(it / 'aln_2')


def test_iterative_profile_search_config4(work):
    """BASELINE config 4 end to end on the regression input: `clustersearch --num-iterations 3` = sequence search with --realign,
    result2profile, two profile searches (profile k-mer prefilter, profile Smith-Waterman) with subtractdbs / mergedbs
    between them (M/src/workflow/Search.cpp:476-518, M/data/workflow/blastpgp.sh:52-140).  Iteration 0 is pinned to the
    reference binary (aln_0 md5), iterations 1 and 2 to the reference's classes on this box (previous test).  Where the host CPU
    rounds rcpps like the one the reference binary ran on (profile_0 and profile_1 md5 equal), the recorded checksums of that binary's run
    hold as well: profile_1, the merged alignment DB (18 698 lines), 331 hits / 119 clusters."""
    g = work / 'genome'
    sdgpu('clustersearch', g, g, work / 'iter.tsv', work / 'tmpi', '--filter-self-match', '--num-iterations', '3', '--threads', '8', '-v', '1',
          '--keep-tmp', '1')
    n0, md5_0 = _key_ordered_md5(work / 'tmpi' / 'search' / 'profile_0')
    assert n0 == 5898
    tsv = open(work / 'iter.tsv').readlines()
    n_hit, n_clu = sum(1 for l in tsv if l.startswith('>')), sum(1 for l in tsv if l.startswith('#'))
    assert n_hit > 308 and n_clu > 108          # the profile iterations add hits to the single-pass result
    # rcpps is an approximation whose last bits differ between CPU models (the GPU boxes are not all alike): the recorded
    # checksums apply only where both profile DBs round like on the machine they were recorded on
    md5_1 = _key_ordered_md5(work / 'tmpi' / 'search' / 'profile_1')
    if md5_0 == '169a337cab4e438fdcb75be742eef3d2' and md5_1 == (5898, '0841b3aae841b086fa8850c8086c20af'):
        sdgpu('prefixid', work / 'tmpi' / 'result', work / 'iter_result.flat', '--tsv')
        lines = open(work / 'iter_result.flat').readlines()
        assert (len(lines), sorted_md5(lines)) == (18698, 'deee49195d78013868efd140ad77b913')
        assert (n_hit, n_clu) == (331, 119)
        assert sorted_md5(tsv, drop_first_column=True) == 'ca3dd1ba9c0f89b9a7cf0726a1bab2ce'


def test_iterative_search_in_memory_equals_the_module_chain(work):
    """`clustersearch --num-iterations 3` without --keep-tmp runs the iterations in memory (csrc/cli/sd_mod_iter.cpp: chunks of queries
    through prefilter -> subtract -> align -> merge -> result2profile of all iterations, several chunks at a time, one aggregation): the
    TSV is the module chain's, byte for byte -- whatever the number of workers and wherever the chunks are cut -- and no DB is left
    between the modules.  Also with two different set DBs (query != target: no identity pairs, the query's own sequence is an ordinary
    target of the profile iterations)."""
    g = work / 'genome'
    common = ['--filter-self-match', '--num-iterations', '3', '--threads', '8', '-v', '0']
    sdgpu('clustersearch', g, g, work / 'itf.tsv', work / 'tmpitf', *common, '--keep-tmp', '1')
    want = open(work / 'itf.tsv').read()
    assert want.count('\n#') > 100
    for tag, env in (('a', {}), ('b', {'SD_ITER_WORKERS': '1', 'SD_ITER_CHUNK': '5898'}), ('c', {'SD_ITER_WORKERS': '4', 'SD_ITER_CHUNK': '611'})):
        p = subprocess.run([SDGPU, 'clustersearch', str(g), str(g), str(work / ('itm_%s.tsv' % tag)), str(work / ('tmpitm_' + tag))] + common,
                           capture_output=True, text=True, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr[-600:]
        assert open(work / ('itm_%s.tsv' % tag)).read() == want, tag
        assert not os.path.exists(work / ('tmpitm_' + tag) / 'search' / 'aln_0') and not os.path.exists(work / ('tmpitm_' + tag) / 'result')
    # two and four iterations: the first / middle / last alignment parameters (Search.cpp:484-505) in their places
    for n_it in ('2', '4'):
        it = ['--filter-self-match', '--num-iterations', n_it, '--threads', '8', '-v', '0']
        sdgpu('clustersearch', g, g, work / ('itf_n%s.tsv' % n_it), work / ('tmpitf_n' + n_it), *it, '--keep-tmp', '1')
        sdgpu('clustersearch', g, g, work / ('itm_n%s.tsv' % n_it), work / ('tmpitm_n' + n_it), *it)
        assert open(work / ('itm_n%s.tsv' % n_it)).read() == open(work / ('itf_n%s.tsv' % n_it)).read(), n_it
    fa = example_fasta(work)
    q, t = work / 'q913i', work / 't915i'
    sdgpu('createsetdb', fa[0], q, work / 'tmpqi', '-v', '0')
    sdgpu('createsetdb', fa[1], t, work / 'tmpti', '-v', '0')
    two = ['--num-iterations', '3', '--threads', '8', '-v', '0']
    sdgpu('clustersearch', q, t, work / 'itf2.tsv', work / 'tmpitf2', *two, '--keep-tmp', '1')
    sdgpu('clustersearch', q, t, work / 'itm2.tsv', work / 'tmpitm2', *two)
    assert open(work / 'itm2.tsv').read() == open(work / 'itf2.tsv').read() and os.path.getsize(work / 'itf2.tsv') > 1000


def test_prefilter_and_clustersearch_use_the_index_file(work):
    """TARGET.idx (createindex layout) replaces the host index build when its META matches the run; results unchanged"""
    g = work / 'genome'
    sdgpu('createindex', g, work / 'tmpx', '-s', '5.7', '--threads', '8', '-v', '0')
    p = sdgpu('prefilter', g, g, work / 'pref_idx', *PREFILTER_PAR)
    assert 'Use index' in p.stdout
    lines = flat(work, 'pref_idx')
    assert (len(lines), sorted_md5(lines)) == (98957, '8109a70bdea70ee10e0dbd27ba6b7e37')
    p = sdgpu('clustersearch', g, g, work / 'fused_idx.tsv', work / 'tmpfx', '--filter-self-match', '--threads', '8')
    assert 'Use index' in p.stdout
    assert sorted_md5(open(work / 'fused_idx.tsv').readlines(), drop_first_column=True) == 'abb28ee37bc130a5f09a9f767ef00ccf'
    # an index built for other parameters is reported and not used
    p = sdgpu('prefilter', g, g, work / 'pref_idx2', *(PREFILTER_PAR[:-1] + ['4']))
    assert 'Index file not used' in p.stdout
    for f in ('genome.idx', 'genome.idx.index', 'genome.idx.dbtype'):
        os.remove(work / f)


def test_literal_config1_two_set_dbs(work):
    """BASELINE configs[0] as written: NC_000913 as the query set DB, NC_000915 as the target set DB (two createsetdb calls,
    query != target, no --filter-self-match): 176 hit lines in 61 clusters, one of them with P < 1E-20, and the canonical
    TSV (without the cluster-key / query columns' first field) hashes like the reference binary's"""
    fa = example_fasta(work)
    q, t = work / 'q913', work / 't915'
    sdgpu('createsetdb', fa[0], q, work / 'tmpq', '-v', '0')
    sdgpu('createsetdb', fa[1], t, work / 'tmpt', '-v', '0')
    sdgpu('clustersearch', q, t, work / 'c1.tsv', work / 'tmpc1', '--threads', '8', '-v', '0')
    tsv = open(work / 'c1.tsv').readlines()
    clu = [l for l in tsv if l.startswith('#')]
    assert (sum(1 for l in tsv if l.startswith('>')), len(clu), sum(1 for l in clu if float(l.split('\t')[3]) < 1e-20)) == (176, 61, 1)
    assert len(tsv) == 237 and sorted_md5(tsv, drop_first_column=True) == '521fe66c5fd4b93b0b3363bd149f9bd7'
