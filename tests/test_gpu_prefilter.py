"""GPU parity: prefilter HIP path (through the C ABI) vs the oracle, bit exact (ids, scores, diagonals, order)."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from spacedust_amd import api

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _run(gpu, host, res, off, identity, max_hits=300, cov_thr=0.8, k=6, kmer_thr=112):
    sw_b, dg_b, km_b = host.comp_bias(res, off, k)
    idx = host.build_index(res, off, k, kmer_thr)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, kmer_thr=kmer_thr, max_hits=max_hits, cov_thr=cov_thr, bin_size=2, k=k)
    hits, cnt, st = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, identity, want_stats=True)
    return idx, hits, cnt, st


def test_prefilter_synthetic_matches_oracle(gpu, host, oracle, small_proteomes):
    ps = small_proteomes
    ident = np.arange(ps.n, dtype=np.uint32)
    idx, hits, cnt, st = _run(gpu, host, ps.residues, ps.offsets, ident, max_hits=50, cov_thr=0.0)
    ot = oracle.target(ps.residues, ps.offsets)
    assert ot.n_entries == idx.n_entries
    for q in range(ps.n):
        seq = ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        ids, sc, dg, ost = ot.prefilter(seq, identity_id=q, max_hits=50, bin_size=2)
        n = int(cnt[q])
        assert n == len(ids), (q, n, len(ids))
        assert (hits[q, :n]['seqId'] == ids).all(), q
        assert (hits[q, :n]['score'] == sc).all(), q
        assert (hits[q, :n]['diagonal'] == dg).all(), q
        assert tuple(int(x) for x in st[q]) == tuple(int(x) for x in ost), (q, st[q], ost)


@pytest.mark.parametrize('kmer_thr', [122, 100])
def test_prefilter_k7_matches_oracle(gpu, host, oracle, small_proteomes, kmer_thr):
    """k = 7 (the automatic choice from 3.35e9 target residues, IndexTable.h:439-449): spaced seed of span 11 and the
    2+2+3 k-mer generator (KmerGenerator.cpp:41-86).  122 = kmerThreshold(5.7, 7); 100 is a far more permissive list."""
    ps = small_proteomes
    assert host.kmer_threshold(5.7, 7) == 122
    ident = np.arange(ps.n, dtype=np.uint32)
    idx, hits, cnt, st = _run(gpu, host, ps.residues, ps.offsets, ident, max_hits=50, cov_thr=0.0, k=7, kmer_thr=kmer_thr)
    ot = oracle.target(ps.residues, ps.offsets, k=7, kmer_thr=kmer_thr)
    assert ot.n_entries == idx.n_entries
    total = 0
    # all queries go through the device (at 100 the k-mer lists of the batch exceed the 2^30 budget and the batch is
    # split); the single-core oracle checks every query at 122 and every 6th at 100
    for q in range(0, ps.n, 1 if kmer_thr == 122 else 6):
        seq = ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        ids, sc, dg, ost = ot.prefilter(seq, identity_id=q, kmer_thr=kmer_thr, max_hits=50, bin_size=2)
        n = int(cnt[q])
        total += n
        assert n == len(ids), (q, n, len(ids))
        assert (hits[q, :n]['seqId'] == ids).all(), q
        assert (hits[q, :n]['score'] == sc).all(), q
        assert (hits[q, :n]['diagonal'] == dg).all(), q
        assert tuple(int(x) for x in st[q]) == tuple(int(x) for x in ost), (q, st[q], ost)
    assert total > ps.n // 6


def _read_fasta_gz(path):
    names, seqs, cur = [], [], []
    with gzip.open(path, 'rt') as f:
        for line in f:
            line = line.rstrip('\n')
            if line.startswith('>'):
                if names:
                    seqs.append(''.join(cur))
                names.append(line[1:])
                cur = []
            else:
                cur.append(line)
    seqs.append(''.join(cur))
    return names, seqs


def test_prefilter_examples_md5(gpu, host):
    """config 1 (the reference's run_regression.sh input): the flattened, sorted prefilter DB must hash to the
    md5 of the reference's own pref_0 (8109a70b..., 98 957 lines)."""
    seqs = []
    for f in ('NC_000913.faa.gz', 'NC_000915.faa.gz'):
        seqs += _read_fasta_gz(os.path.join(GOLD, 'examples', f))[1]
    res, off = host.map_sequences(seqs)
    ident = np.arange(len(seqs), dtype=np.uint32)
    idx, hits, cnt, st = _run(gpu, host, res, off, ident)
    assert idx.n_entries == 1784989 and idx.masked_residues == 11546
    lines = []
    for q in range(len(seqs)):
        for h in hits[q, :int(cnt[q])]:
            lines.append('%d\t%d\t%d\t%d\n' % (q, h['seqId'], h['score'], np.int16(np.uint16(h['diagonal']))))
    assert len(lines) == 98957
    lines.sort(key=lambda s: s.encode())
    assert hashlib.md5(''.join(lines).encode()).hexdigest() == '8109a70bdea70ee10e0dbd27ba6b7e37'


def test_prefilter_bucket_path_equals_sort_path(gpu, host, monkeypatch):
    """the bucketed LDS match (default) and the global-radix-sort fallback give identical hit tables, at a size
    where queries span many target-range buckets and long queries overflow the small bucket capacity"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=12, genes_per_proteome=900, n_families=1400, seed=91)
    ident = np.arange(ps.n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets)
    idx = host.build_index(ps.residues, ps.offsets)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.8, bin_size=None)
    nq = 3000
    res = ps.residues[:int(ps.offsets[nq])]
    off = ps.offsets[:nq + 1]
    a = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
    monkeypatch.setenv('SD_PF_SORT', '1')
    b = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2])
    for q in range(nq):
        n = int(a[1][q])
        assert np.array_equal(a[0][q, :n], b[0][q, :n]), q
    assert int(a[1].sum()) > 50000
    # the coarse split for very hit-rich queries (large target sets), forced here by a tiny per-range budget: every
    # query is cut into 2, 4, ... target ranges that go through the bucket machinery as virtual queries
    monkeypatch.delenv('SD_PF_SORT')
    for budget in ('6000', '500'):
        monkeypatch.setenv('SD_PF_COARSE', budget)
        c = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
        assert np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2]), budget
        for q in range(nq):
            n = int(a[1][q])
            assert np.array_equal(a[0][q, :n], c[0][q, :n]), (budget, q)


def test_prefilter_rescoring_path_matches_reference(gpu, host, oracle):
    """the truncated-score path on the device against rows produced by the real reference"""
    g = np.load(os.path.join(GOLD, 'rescore_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    res = np.concatenate(nums)
    n = len(nums)
    ident = np.arange(n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.0, bin_size=2)
    hits, cnt, _ = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident, want_stats=True)
    rows = g['pf_rows']
    for q in g['queries']:
        exp = rows[rows[:, 0] == q]
        m = int(cnt[q])
        assert m == len(exp), (q, m, len(exp))
        assert (hits[q, :m]['seqId'] == exp[:, 1]).all() and (hits[q, :m]['score'] == exp[:, 2]).all(), q
        assert (hits[q, :m]['diagonal'].astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), q


@pytest.mark.parametrize('kmer_thr', [122, 100])
def test_prefilter_k7_matches_reference(gpu, host, oracle, kmer_thr):
    """k = 7 on the device against rows produced by the real reference (tools/make_golden_k7.py)"""
    g = np.load(os.path.join(GOLD, 'k7_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    res = np.concatenate(nums)
    ident = np.arange(len(nums), dtype=np.uint32)
    idx, hits, cnt, _ = _run(gpu, host, res, off, ident, max_hits=300, cov_thr=0.0, k=7, kmer_thr=kmer_thr)
    rows = g['pf_rows_%d' % kmer_thr]
    for q in g['queries']:
        exp = rows[rows[:, 0] == q]
        m = int(cnt[q])
        assert m == len(exp), (q, m, len(exp))
        assert (hits[q, :m]['seqId'] == exp[:, 1]).all() and (hits[q, :m]['score'] == exp[:, 2]).all(), q
        assert (hits[q, :m]['diagonal'].astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), q


def test_prefilter_hit_buffer_overflow_matches_reference(gpu, host, oracle, monkeypatch):
    """the one-overflow path on the device (bucket path and sort fallback) against rows from the real reference"""
    g = np.load(os.path.join(GOLD, 'overflow_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    res = np.concatenate(nums)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.0, bin_size=2)
    qs = [int(q) for q in g['queries']]
    qoff = np.zeros(len(qs) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(nums[q]) for q in qs])
    qres = np.concatenate([nums[q] for q in qs])
    qkm = np.concatenate([km_b[int(off[q]):int(off[q + 1])] for q in qs])
    qdg = np.concatenate([dg_b[int(off[q]):int(off[q + 1])] for q in qs])
    rows = g['pf_rows']
    for mode in ('bucket', 'coarse', 'sort'):
        if mode == 'coarse':   # the overflowing query (2*10^6 hits) is split into target ranges first; the split position
            monkeypatch.setenv('SD_PF_COARSE', '300000')   # of the overflow is looked up through the virtual query
        if mode == 'sort':
            monkeypatch.delenv('SD_PF_COARSE')
            monkeypatch.setenv('SD_PF_SORT', '1')
        hits, cnt, st = api.prefilter(gpu, tgt, par, qres, qoff, qkm, qdg, np.array(qs, np.uint32), want_stats=True)
        assert int(st[0, 1]) == int(g['index_hits_q0'][0])
        for x, q in enumerate(qs):
            exp = rows[rows[:, 0] == q]
            m = int(cnt[x])
            assert m == len(exp), (mode, q, m, len(exp))
            assert (hits[x, :m]['seqId'] == exp[:, 1]).all() and (hits[x, :m]['score'] == exp[:, 2]).all(), (mode, q)
            assert (hits[x, :m]['diagonal'].astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), (mode, q)


@pytest.mark.parametrize('k', [6, 7])
def test_prefilter_wide_index_equals_ordinary(gpu, host, small_proteomes, monkeypatch, k):
    """the wide index form (>= 2^32 entries: 32-bit list starts relative to a 64-bit base per 65 536 k-mers, forced here by
    SD_INDEX_WIDE) through emit_kmers<WIDE> / gather_hits: same rows, same statistics as the ordinary index"""
    ps = small_proteomes
    thr = host.kmer_threshold(5.7, k)
    ident = np.arange(ps.n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets, k=k)
    out = []
    for wide in (False, True):
        if wide:
            monkeypatch.setenv('SD_INDEX_WIDE', '1')
        idx = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=thr)
        if wide:
            monkeypatch.delenv('SD_INDEX_WIDE')
        assert (idx.block_base is not None) == wide
        tgt = api.Target(gpu, host, idx)
        par = api.prefilter_params(host, idx.n, kmer_thr=thr, max_hits=300, cov_thr=0.0, k=k)
        out.append(api.prefilter(gpu, tgt, par, ps.residues, ps.offsets, km_b, dg_b, ident, want_stats=True))
    (h0, c0, s0), (h1, c1, s1) = out
    assert np.array_equal(c0, c1) and np.array_equal(s0, s1) and int(c0.sum()) > ps.n
    for q in range(ps.n):
        assert np.array_equal(h0[q, :int(c0[q])], h1[q, :int(c1[q])]), q


def test_prefilter_sequences_of_32768_residues_and_more(gpu, host, oracle):
    """with 32 768 residues or more on either side the 16-bit diagonal is ambiguous and the reference scores every real
    diagonal it can stand for (computeLongScore, UngappedAlignment.cpp:312-329): two targets of 40 000 and 34 000
    residues, fragments of them from both sides of the wrap, every sequence as a query -- rows from the real reference
    (tests/golden/long_vectors.npz, tools/make_golden_long.py)"""
    g = np.load(os.path.join(GOLD, 'long_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    n = len(off) - 1
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(n)]
    res = np.concatenate(nums)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.0)
    hits, cnt, _ = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, np.arange(n, dtype=np.uint32))
    rows = g['pf_rows']
    long_rows = 0
    for q in range(n):
        exp = rows[rows[:, 0] == q]
        m = int(cnt[q])
        assert m == len(exp), (q, m, len(exp))
        assert (hits[q, :m]['seqId'] == exp[:, 1]).all() and (hits[q, :m]['score'] == exp[:, 2]).all(), q
        assert (hits[q, :m]['diagonal'].astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), q
        long_rows += int(((exp[:, 1] < 2) | (q < 2)).sum())
    assert long_rows > 40


def test_prefilter_repeated_overflow_matches_reference(gpu, host, oracle, monkeypatch):
    """a query whose index hits overflow the reference's hit buffer two and three times (QueryMatcher.cpp:281-316 with the
    merge / score / keep-one-per-target branch :289-303) is computed on the device -- through the k-mer-major join and through
    the index-lookup path, coarse split included -- and equals rows of the real reference (tests/golden/overflow2_vectors.npz,
    tools/make_golden_overflow2.py: query 0 has 6.45 * 10^6 index hits against the 2 * 10^6-entry buffer); in the global-sort
    fallback such a query is reported through its count slot, and the others of the batch are still complete"""
    g = np.load(os.path.join(GOLD, 'overflow2_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    res = np.concatenate(nums)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.0, bin_size=2)
    qs = [int(q) for q in g['queries']]
    qoff = np.zeros(len(qs) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(nums[q]) for q in qs])
    qres = np.concatenate([nums[q] for q in qs])
    qkm = np.concatenate([km_b[int(off[q]):int(off[q + 1])] for q in qs])
    qdg = np.concatenate([dg_b[int(off[q]):int(off[q + 1])] for q in qs])
    rows = g['pf_rows']
    stats = g['stats']
    for mode in ('join', 'lookup', 'coarse', 'sort'):
        if mode == 'lookup':
            monkeypatch.setenv('SD_PF_JOIN', '0')
        if mode == 'coarse':
            monkeypatch.setenv('SD_PF_COARSE', '300000')
        if mode == 'sort':
            monkeypatch.delenv('SD_PF_COARSE')
            monkeypatch.setenv('SD_PF_SORT', '1')
        hits, cnt, st = api.prefilter(gpu, tgt, par, qres, qoff, qkm, qdg, np.array(qs, np.uint32), want_stats=True)
        for x, q in enumerate(qs):
            assert int(st[x, 1]) == int(stats[x, 2]), (mode, q, st[x], stats[x])
            exp = rows[rows[:, 0] == q]
            m = int(cnt[x])
            if mode == 'sort' and int(st[x, 1]) > 4000000:
                assert m == 0xFFFFFFFF and 'more than once' in gpu.last_error(), (mode, q, m)
                continue
            assert m == len(exp), (mode, q, m, len(exp))
            assert (hits[x, :m]['seqId'] == exp[:, 1]).all() and (hits[x, :m]['score'] == exp[:, 2]).all(), (mode, q)
            assert (hits[x, :m]['diagonal'].astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), (mode, q)


@pytest.mark.parametrize('bin_size,max_hits,min_diag', [(4, 300, 15), (64, 7, 15), (2048, 300, 40), (2, 1, 15)])
def test_prefilter_parameters_match_oracle(gpu, host, oracle, small_proteomes, bin_size, max_hits, min_diag):
    """BINSIZE (the order ties are cut in: bin-major, QueryMatcher.cpp:422-450), result list length and the minimum
    diagonal score"""
    ps = small_proteomes
    ident = np.arange(ps.n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets)
    idx = host.build_index(ps.residues, ps.offsets)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=max_hits, min_diag=min_diag, cov_thr=0.0, bin_size=bin_size)
    hits, cnt, _ = api.prefilter(gpu, tgt, par, ps.residues, ps.offsets, km_b, dg_b, ident)
    ot = oracle.target(ps.residues, ps.offsets)
    total = 0
    for q in range(0, ps.n, 2):
        seq = ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        ids, sc, dg, _ = ot.prefilter(seq, identity_id=q, max_hits=max_hits, min_diag=min_diag, bin_size=bin_size)
        n = int(cnt[q])
        total += n
        assert n == len(ids), (q, n, len(ids))
        assert (hits[q, :n]['seqId'] == ids).all() and (hits[q, :n]['score'] == sc).all() and (hits[q, :n]['diagonal'] == dg).all(), q
    assert total >= ps.n // 2


def test_prefilter_long_result_lists(gpu, host, oracle):
    """result lists beyond 2 047 hits (--max-seqs 2N past ~1 000 proteomes) use the 8 192-entry selection: one family of
    3 600 near-identical sequences, so thousands of targets pass every cut (and saturate the 8-bit score: the list is
    cut inside the rescaled tie classes)"""
    rng = np.random.default_rng(404)
    aa = 'ACDEFGHIKLMNPQRSTVWY'
    base = ''.join(rng.choice(list(aa), 220))
    seqs = [base]
    for _ in range(3600):
        sq = list(base)
        for p in np.nonzero(rng.random(len(sq)) < rng.uniform(0.02, 0.35))[0]:
            sq[p] = aa[rng.integers(20)]
        seqs.append(''.join(sq))
    seqs += [''.join(rng.choice(list(aa), int(rng.integers(100, 300)))) for _ in range(400)]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    res, off = host.map_sequences(seqs)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    tgt = api.Target(gpu, host, idx)
    ot = oracle.target(res, off)
    queries = [int(np.nonzero(order == 0)[0][0]), 5, 17, 123]
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(seqs[q]) for q in queries])
    qres = np.concatenate([res[int(off[q]):int(off[q + 1])] for q in queries])
    qkm = np.concatenate([km_b[int(off[q]):int(off[q + 1])] for q in queries])
    qdg = np.concatenate([dg_b[int(off[q]):int(off[q + 1])] for q in queries])
    for max_hits, min_diag in ((3000, 15), (2500, 1), (4000, 15)):
        par = api.prefilter_params(host, idx.n, max_hits=max_hits, min_diag=min_diag, cov_thr=0.0, bin_size=4)
        hits, cnt, _ = api.prefilter(gpu, tgt, par, qres, qoff, qkm, qdg, np.array(queries, np.uint32))
        longest = 0
        for x, q in enumerate(queries):
            ids, sc, dg, _ = ot.prefilter(res[int(off[q]):int(off[q + 1])], identity_id=q, max_hits=max_hits, min_diag=min_diag,
                                          bin_size=4)
            n = int(cnt[x])
            longest = max(longest, n)
            assert n == len(ids), (max_hits, q, n, len(ids))
            assert (hits[x, :n]['seqId'] == ids).all() and (hits[x, :n]['score'] == sc).all() and (hits[x, :n]['diagonal'] == dg).all(), q
        assert longest > 2048, (max_hits, longest)


def test_prefilter_result_lists_of_any_length(gpu, host, oracle):
    """the reference caps a result list at min(--max-seqs, dbSize) and nothing else (QueryMatcher.cpp:45-46,386); clustersearch
    asks for --max-seqs 2N.  Lists beyond 4 095 hits with more candidates at the score cut than the LDS sorter of
    select_hits holds (8 192) go through select_hits_big_kernel (radix selection + global-scratch sort): one family of 9 600
    near-identical sequences, cuts inside the saturated tie classes, at the minimum score, and no cut at all"""
    rng = np.random.default_rng(405)
    aa = 'ACDEFGHIKLMNPQRSTVWY'
    base = ''.join(rng.choice(list(aa), 200))
    seqs = [base]
    for _ in range(9600):
        sq = list(base)
        for p in np.nonzero(rng.random(len(sq)) < rng.uniform(0.02, 0.45))[0]:
            sq[p] = aa[rng.integers(20)]
        seqs.append(''.join(sq))
    seqs += [''.join(rng.choice(list(aa), int(rng.integers(100, 300)))) for _ in range(400)]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order]
    res, off = host.map_sequences(seqs)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    tgt = api.Target(gpu, host, idx)
    ot = oracle.target(res, off)
    queries = [int(np.nonzero(order == 0)[0][0]), 5, 17]
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(seqs[q]) for q in queries])
    qres = np.concatenate([res[int(off[q]):int(off[q + 1])] for q in queries])
    qkm = np.concatenate([km_b[int(off[q]):int(off[q + 1])] for q in queries])
    qdg = np.concatenate([dg_b[int(off[q]):int(off[q + 1])] for q in queries])
    for max_hits, min_diag, bin_size in ((6000, 15, 4), (9000, 1, 8), (4500, 40, 2), (20000, 15, 4)):
        par = api.prefilter_params(host, idx.n, max_hits=max_hits, min_diag=min_diag, cov_thr=0.0, bin_size=bin_size)
        for ident in (np.array(queries, np.uint32), np.full(len(queries), 0xFFFFFFFF, np.uint32)):
            hits, cnt, _ = api.prefilter(gpu, tgt, par, qres, qoff, qkm, qdg, ident)
            longest = 0
            for x, q in enumerate(queries):
                ids, sc, dg, _ = ot.prefilter(res[int(off[q]):int(off[q + 1])], identity_id=int(ident[x]), max_hits=max_hits,
                                              min_diag=min_diag, bin_size=bin_size)
                n = int(cnt[x])
                longest = max(longest, n)
                assert n == len(ids), (max_hits, q, n, len(ids))
                assert (hits[x, :n]['seqId'] == ids).all() and (hits[x, :n]['score'] == sc).all() and (hits[x, :n]['diagonal'] == dg).all(), (max_hits, q)
            assert longest >= min(max_hits, 9000) - 1 or max_hits == 20000 and longest > 9000, (max_hits, longest)


def test_prefilter_hot_filter_equals_unfiltered(gpu, host, monkeypatch):
    """hot_filter_kernel drops, in front of the bucket machinery, the hits of targets that cannot emit a candidate (no two hits
    with the same diagonal byte, no hit with diagonal byte 0): rows and statistics equal the unfiltered run -- on the join path,
    the lookup path, with the coarse split, and with every segment length filtered (SD_PF_FILTER_MIN=0)"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=12, genes_per_proteome=900, n_families=1400, seed=92)
    ident = np.arange(ps.n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets)
    idx = host.build_index(ps.residues, ps.offsets)
    tgt = api.Target(gpu, host, idx)
    nq = 4000
    res = ps.residues[:int(ps.offsets[nq])]
    off = ps.offsets[:nq + 1]
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.8, bin_size=None)
    monkeypatch.setenv('SD_PF_FILTER', '0')
    a = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
    assert int(a[1].sum()) > 30000
    monkeypatch.setenv('SD_PF_FILTER', '1')
    for env in ({}, {'SD_PF_FILTER_MIN': '0'}, {'SD_PF_JOIN': '0'}, {'SD_PF_JOIN': '0', 'SD_PF_FILTER_MIN': '0', 'SD_PF_COARSE': '3000'},
                {'SD_PF_COARSE': '2000', 'SD_PF_FILTER_MIN': '0'},
                # the join's ranged scatter (SD_JOIN_RANGES=1: an experiment, off by default) instead of the coarse split behind a per-query scatter
                {'SD_PF_COARSE': '2000', 'SD_JOIN_RANGES': '1'}, {'SD_PF_COARSE': '20000', 'SD_JOIN_RANGES': '1'}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        b = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
        for k_ in env:
            monkeypatch.delenv(k_)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), env
        for q in range(nq):
            n = int(a[1][q])
            assert np.array_equal(a[0][q, :n], b[0][q, :n]), (env, q)


def test_prefilter_join_path_equals_lookup_path(gpu, host, monkeypatch):
    """the k-mer-major join (default for k = 6) and the per-k-mer lookup path (SD_PF_JOIN=0) give identical hit tables
    and statistics -- plain, with the coarse split forced, and with the reference's hit-buffer overflow forced into
    the longest queries (the split is a k-mer ordinal on the join path, a stream position on the other)"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=12, genes_per_proteome=900, n_families=1400, seed=91)
    ident = np.arange(ps.n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets)
    idx = host.build_index(ps.residues, ps.offsets)
    tgt = api.Target(gpu, host, idx)
    nq = 5000
    res = ps.residues[:int(ps.offsets[nq])]
    off = ps.offsets[:nq + 1]

    def same(a, b, what):
        assert np.array_equal(a[1], b[1]), what
        assert np.array_equal(a[2], b[2]), what
        for q in range(nq):
            n = int(a[1][q])
            assert np.array_equal(a[0][q, :n], b[0][q, :n]), (what, q)

    for bin_size, max_hits in ((None, 300), (8, 50)):
        par = api.prefilter_params(host, idx.n, max_hits=max_hits, cov_thr=0.8, bin_size=bin_size)
        monkeypatch.setenv('SD_PF_JOIN', '0')
        a = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
        monkeypatch.setenv('SD_PF_JOIN', '1')
        b = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
        same(a, b, ('plain', bin_size))
        assert int(a[1].sum()) > 30000
        for budget in ('20000', '1500'):
            monkeypatch.setenv('SD_PF_COARSE', budget)
            c = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
            same(a, c, ('coarse', budget))
        monkeypatch.delenv('SD_PF_COARSE')
        # small sub-batches: groups of one query, several sub-batches per call
        monkeypatch.setenv('SD_PF_BATCH', '700')
        d = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident[:nq], want_stats=True)
        same(a, d, ('batch 700', bin_size))
        monkeypatch.delenv('SD_PF_BATCH')


@pytest.mark.parametrize('k,kmer_thr', [(6, 112), (6, 95), (7, 122)])
def test_prefilter_cumulative_score_table_equals_row_search(gpu, host, monkeypatch, k, kmer_thr):
    """countGE of a sorted 3-mer row as one read of the target's cumulative score table (sdBuildExt3Cum) against the 13-step search of
    the row it replaces (a target created with SD_PF_CUM=0 has no table): identical rows and statistics (k-mer and hit counts per
    query) on the join path, the lookup path and the k = 7 generator"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=6, genes_per_proteome=500, n_families=800, seed=93)
    ident = np.arange(ps.n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets, k)
    idx = host.build_index(ps.residues, ps.offsets, k, kmer_thr)
    par = api.prefilter_params(host, idx.n, kmer_thr=kmer_thr, max_hits=100, cov_thr=0.0, bin_size=2, k=k)
    monkeypatch.setenv('SD_PF_CUM', '0')
    plain = api.Target(gpu, host, idx)
    monkeypatch.delenv('SD_PF_CUM')
    table = api.Target(gpu, host, idx)
    for env in ({}, {'SD_PF_JOIN': '0'}):
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        a = api.prefilter(gpu, plain, par, ps.residues, ps.offsets, km_b, dg_b, ident, want_stats=True)
        b = api.prefilter(gpu, table, par, ps.residues, ps.offsets, km_b, dg_b, ident, want_stats=True)
        for k_ in env:
            monkeypatch.delenv(k_)
        assert int(a[1].sum()) > 1000
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), env
        for q in range(ps.n):
            n = int(a[1][q])
            assert np.array_equal(a[0][q, :n], b[0][q, :n]), (env, q)
