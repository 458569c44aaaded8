"""End-to-end on the GPU: the reference's own regression input (examples/*.faa, run_regression.sh) through
prefilter -> align -> aggregation -> clusterhits -> TSV must reproduce the reference's known answers."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from spacedust_amd.pipeline import SetDB, ClusterSearch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def load_examples(host):
    names, seqs, set_id, pos, strand = [], [], [], [], []
    sources = ['NC_000913.faa', 'NC_000915.faa']
    for si, f in enumerate(sources):
        with gzip.open(os.path.join(GOLD, 'examples', f + '.gz'), 'rt') as fh:
            cur, hdr, idx = [], None, 0
            def flush():
                nonlocal idx
                if hdr is None:
                    return
                w = hdr.replace(' ', '').split('#')          # createsetdb.sh:119-139
                acc, st, en, sd = w[0], int(w[1]), int(w[2]), w[3]
                if sd == '-1':
                    st, en = en, st
                names.append('%s_%d_%d_%d' % (acc, idx, st, en))
                seqs.append(''.join(cur))
                set_id.append(si); pos.append(idx); strand.append(1 if st < en else 0)
                idx += 1
            for line in fh:
                line = line.rstrip('\n')
                if line.startswith('>'):
                    flush()
                    hdr, cur = line[1:], []
                else:
                    cur.append(line)
            flush()
    res, off = host.map_sequences(seqs)
    return SetDB(res, off, set_id, pos, strand, 2, names=names, sources=sources)


def _subset(db, set_index):
    """one genome of the two-genome example DB as its own set DB (what a createsetdb call on that FASTA alone makes)"""
    m = np.nonzero(np.asarray(db.set_id) == set_index)[0]
    a, b = int(m[0]), int(m[-1]) + 1
    assert (m == np.arange(a, b)).all()
    off = (db.offsets[a:b + 1] - db.offsets[a]).astype(np.uint64)
    res = db.residues[int(db.offsets[a]):int(db.offsets[b])]
    return SetDB(res, off, [0] * (b - a), list(np.asarray(db.pos_in_set)[a:b]), list(np.asarray(db.strand)[a:b]), 1,
                 names=db.names[a:b], sources=[db.sources[set_index]])


def test_query_db_differs_from_target_db(gpu, host, tmp_path):
    """pipeline.search(same_db=False) -- BASELINE configs[0] as written: NC_000913 (query set DB) against NC_000915 (target
    set DB), no self-match filter: the reference binary finds 176 hit lines in 61 clusters, one with P < 1E-20, canonical
    TSV md5 521fe66c..."""
    both = load_examples(host)
    q, t = _subset(both, 0), _subset(both, 1)
    cs = ClusterSearch(gpu, host, t, max_seqs=300, filter_self_match=False)
    out = cs.search(q, same_db=False, tsv_path=str(tmp_path / 'c1.tsv'), canonical=True)
    lines = open(tmp_path / 'c1.tsv').readlines()
    n_clu = sum(1 for l in lines if l.count('\t') == 4)
    sig = sum(1 for l in lines if l.count('\t') == 4 and float(l.split('\t')[2]) < 1e-20)
    assert (len(lines) - n_clu, n_clu, sig) == (176, 61, 1)
    assert out['clusters'] == 61
    lines.sort(key=lambda s_: s_.encode())
    assert hashlib.md5(''.join(lines).encode()).hexdigest() == '521fe66c5fd4b93b0b3363bd149f9bd7'


@pytest.mark.parametrize('pushdown', ['0', '1'])
def test_examples_regression_known_answers(gpu, host, tmp_path, monkeypatch, pushdown):
    """the reference's regression run.  pushdown = 0: every pair aligned to -e 10 (15 065 accepted alignments = the lines of the
    reference's alignment DB); pushdown = 1, the default of a stream that leaves nothing but cluster records: the alignments gated
    at combinehits' E-value bound -- fewer accepted alignments, the same TSV byte for byte"""
    monkeypatch.setenv('SD_EVAL_PUSHDOWN', pushdown)
    db = load_examples(host)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
    out = cs.search(db, same_db=True, tsv_path=str(tmp_path / 'result.tsv'), canonical=True, chunk_queries=3000)
    assert cs.index_entries == 1784989 and cs.masked_residues == 11546
    assert cs.stats['prefilter_hits'] == 98957
    assert int(cs._raw_stats()[0][15]) == int(pushdown)
    if pushdown == '0':
        assert out['accepted'] == 15065
    else:
        assert 308 <= out['accepted'] < 15065
    lines = open(tmp_path / 'result.tsv').readlines()
    n_clu = sum(1 for l in lines if l.count('\t') == 4)
    n_hit = len(lines) - n_clu
    sig = sum(1 for l in lines if l.count('\t') == 4 and float(l.split('\t')[2]) < 1e-20)
    # R/util/run_regression.sh:20-23: 308 hit lines, 2 clusters with P < 1E-20 (108 clusters in total)
    assert (n_hit, sig, n_clu) == (308, 2, 108)
    lines.sort(key=lambda s: s.encode())
    assert hashlib.md5(''.join(lines).encode()).hexdigest() == 'abb28ee37bc130a5f09a9f767ef00ccf'


def test_synthetic_proteomes_match_real_reference(gpu, host):
    """BASELINE configs[1]-like data (synthetic proteomes, --max-seqs 300) at a size the suite can afford (40 proteomes,
    1.2e5 targets): device prefilter rows and alignments against the REAL reference classes run beside it
    (oracle/_ref/libsdref.so travels with the snapshot).  tools/scale_parity.py is the same check at 1 000 proteomes."""
    from oracle.pyoracle import Ref, RefSW, ref_available
    if not ref_available():
        pytest.skip('oracle/_ref/libsdref.so not present on this box')
    from spacedust_amd import api
    from spacedust_amd.synth import make_proteomes, ALPHABET
    ps = make_proteomes(40, genes_per_proteome=3000, seed=0x5ED0 + 2)
    lens = ps.lengths()
    rng = np.random.default_rng(9)
    queries = np.concatenate([np.argsort(-lens)[:6], rng.choice(ps.n, 60, replace=False)]).astype(np.int64)
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum(lens[queries])
    qres = np.concatenate([ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])] for q in queries])
    sw_b, dg_b, km_b = host.comp_bias(qres, qoff)
    idx = host.build_index(ps.residues, ps.offsets)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.0, bin_size=None)
    hits, cnt, _ = api.prefilter(gpu, tgt, par, qres, qoff, km_b, dg_b, queries.astype(np.uint32))
    blob = np.frombuffer(ALPHABET.encode(), np.uint8)[ps.residues].tobytes()
    ref = Ref(6)
    rix = ref.index(blob, ps.offsets, threads=8)
    assert rix.n_entries == idx.n_entries
    rpf = rix.prefilter(int(lens.max()) + 2, max_hits=300)
    for x, q in enumerate(queries):
        ids, sc, dg, _ = rpf.query(blob[int(ps.offsets[q]):int(ps.offsets[q + 1])], int(q))
        n = int(cnt[x])
        assert n == len(ids), (q, n, len(ids))
        assert (hits[x, :n]['seqId'] == ids).all() and (hits[x, :n]['score'] == sc).all() and (hits[x, :n]['diagonal'] == dg).all(), q
    assert int(cnt.sum()) > 3000
    mat, _, _ = host.matrix(0)
    db = int(ps.offsets[-1])
    ts = gpu.seqset(ps.residues, ps.offsets, None)
    qs = gpu.seqset(qres, qoff, sw_b)
    spar = gpu.sw_params(mat, db)
    pq = np.array([x for x in range(6, 26) for _ in range(min(25, int(cnt[x])))], np.uint32)
    pt = np.array([int(hits[x, h]['seqId']) for x in range(6, 26) for h in range(min(25, int(cnt[x])))], np.uint32)
    ident = queries[pq] == pt
    out, pool = gpu.sw_align(spar, qs, ts, pq, pt, identity=ident)
    sw = RefSW(ref, int(lens.max()) + 2, db)
    last, n_bt = -1, 0
    for i in range(len(pq)):
        if pq[i] != last:
            q = queries[pq[i]]
            sw.set_query(blob[int(ps.offsets[q]):int(ps.offsets[q + 1])])
            last = pq[i]
        t = pt[i]
        r = sw.align(blob[int(ps.offsets[t]):int(ps.offsets[t + 1])], identity=bool(ident[i]))
        o = out[i]
        assert (int(o['score']), int(o['qEnd']), int(o['tEnd']), int(o['btLen'])) == (r['score'], r['qEnd'], r['tEnd'], r['btLen']), i
        if r['btLen'] > 0:
            n_bt += 1
            bt = pool[int(o['btOffset']):int(o['btOffset']) + int(o['btLen'])].tobytes().decode()
            assert bt == r['backtrace'] and (int(o['qStart']), int(o['tStart'])) == (r['qStart'], r['tStart']), i
            assert float(o['evalue']) == r['evalue'], i
    assert n_bt > 100


def test_clusterhits_on_pipeline_entries_matches_oracle(gpu, host, oracle):
    """the (query set, target set) entries a real search produces on whole synthetic proteomes (K of a few thousand hits,
    long syntenic blocks, inversions): device clusterhits, entry by entry, against the reference's own functions
    (oracle/_ref/libsdref_ch.so: partition, sizes, P-value bit patterns, member ranks) where that library travelled, and
    against the oracle's dense restatement"""
    from oracle.pyoracle import oracle_clusterhits, ref_ch_available, RefClusterHits
    ref = RefClusterHits() if ref_ch_available() else None
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(3, genes_per_proteome=3000, seed=0x5ED0 + 2)
    db = SetDB.from_proteomes(ps)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, filter_self_match=True)
    out = cs.search(db, same_db=True, chunk_queries=4000)
    co, off = out['cluster_out'], out['entry_off']
    assert len(out['entry_q']) == 6 and int(co['n_clusters'].sum()) > 50
    for e in range(len(out['entry_q'])):
        a, b = int(off[e]), int(off[e + 1])
        assert b - a > 100
        hq, ht = out['hit_q'][a:b], out['hit_t'][a:b]
        sd = (db.strand[hq] | (db.strand[ht] << 1)).astype(np.uint8)
        cof, mo, csz, pco, pmh, nm = oracle_clusterhits(oracle, db.pos_in_set[hq], db.pos_in_set[ht], sd, out['hit_pval'][a:b],
                                                        int(db.set_size[out['entry_q'][e]]))
        n = len(csz)
        assert int(co['n_clusters'][e]) == n, (e, co['n_clusters'][e], n)
        assert (co['cluster_of'][a:b] == cof).all(), e
        assert (co['size'][a:a + n] == csz).all() and (co['pCO'][a:a + n] == pco).all() and (co['pMH'][a:a + n] == pmh).all(), e
        if ref is not None:
            rcof, rrank, rcs, rpco, rpmh = ref.entry(db.pos_in_set[hq], db.pos_in_set[ht], sd, out['hit_pval'][a:b],
                                                     int(db.set_size[out['entry_q'][e]]))
            assert len(rcs) == n and (co['cluster_of'][a:b] == rcof).all() and (co['size'][a:a + n] == rcs).all(), e
            assert co['pCO'][a:a + n].tobytes() == rpco.tobytes() and co['pMH'][a:a + n].tobytes() == rpmh.tobytes(), e
            clustered = rcof != 0xFFFFFFFF
            assert (co['rank'][a:b][clustered] == rrank[clustered]).all(), e


def test_multi_set_aggregation_matches_restatement(gpu, host, oracle, small_proteomes):
    """three query / target sets with --filter-self-match: the fused aggregation (prefixid -> besthitbyset ->
    mergeresultsbyset -> combinehits incl. the two %.3E text round trips, clustersearch.sh:121-151) against a plain Python
    restatement fed with the oracle's prefilter hits and alignments: same (query set, target set) entries, same hits in
    the same order, bit-identical P-values"""
    import math
    import sys
    ps = small_proteomes
    db = SetDB.from_proteomes(ps)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
    out = cs.search(db, same_db=True, chunk_queries=100)
    ot = oracle.target(ps.residues, ps.offsets)
    db_res = int(ps.offsets[-1])
    lens = ps.lengths()
    dbl_min = sys.float_info.min

    def logpval(ev):   # besthitbyset.cpp:49-63,129
        if ev == 0:
            return math.log(dbl_min)
        if 0 < ev < 10e-4:
            return math.log(ev)
        return math.log(1 - math.exp(-ev))

    best = {}   # (query, target set) -> (sort key, target)
    for q in range(ps.n):
        seq = ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        ids, _, _, _ = ot.prefilter(seq, identity_id=q, max_hits=300, bin_size=2)
        for t in ids:
            t = int(t)
            if not (np.float32(lens[t]) / np.float32(lens[q]) >= np.float32(0.8)):   # Util::canBeCovered, COV_MODE_QUERY
                continue
            r = oracle.sw_align(seq, ps.residues[int(ps.offsets[t]):int(ps.offsets[t + 1])], db_res, identity=(t == q))
            if t != q:
                if r['btLen'] <= 0 or r['qStart'] < 0:
                    continue
                qcov = np.float32(r['qEnd'] - r['qStart'] + 1) / np.float32(lens[q])
                if not (r['evalue'] <= 10.0 and qcov >= np.float32(0.8) and r['btLen'] >= 30):
                    continue
            key = (r['evalue'], -int(oracle.bitscore(r['score']) + 0.5), int(lens[t]), t)   # Matcher::compareHits
            ts = int(ps.set_id[t])
            if (q, ts) not in best or key < best[(q, ts)][0]:
                best[(q, ts)] = (key, t)
    expected = {}
    thr = math.log(10e-7)   # combinehits.cpp:101-103
    for (q, ts), (key, t) in sorted(best.items()):
        qs = int(ps.set_id[q])
        if qs == ts:
            continue   # --filter-self-match (combinehits.cpp:83)
        lp = float('%.3E' % logpval(float('%.3E' % key[0])))   # the alignment DB carries the E-value as %.3E text (Matcher.cpp:288)
        if lp < thr:
            expected.setdefault((qs, ts), []).append((q, t, float('%.3E' % math.exp(lp))))
    got = {}
    off = out['entry_off']
    for e in range(len(out['entry_q'])):
        a, b = int(off[e]), int(off[e + 1])
        got[(int(out['entry_q'][e]), int(out['entry_t'][e]))] = [(int(out['hit_q'][x]), int(out['hit_t'][x]), float(out['hit_pval'][x]))
                                                                  for x in range(a, b)]
    assert sorted(got) == sorted(expected)
    assert len(expected) == 6 and sum(len(v) for v in expected.values()) > 100
    for k_ in expected:
        assert got[k_] == expected[k_], k_


def test_pipeline_with_k7(gpu, host, oracle, small_proteomes, monkeypatch):
    """the search with k = 7 forced (what -k 0 selects from 3.35e9 target residues): hit and accepted-alignment counts
    against the oracle run stage by stage (the oracle's k = 7 prefilter is pinned to the real reference, k7_vectors.npz)"""
    ps = small_proteomes
    db = SetDB.from_proteomes(ps)
    monkeypatch.setenv('SD_EVAL_PUSHDOWN', '0')   # (the count of alignments accepted at -e 10 is what is compared)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, k=7, filter_self_match=True)
    assert cs.k == 7 and cs.kmer_thr == 122
    out = cs.search(db, same_db=True, chunk_queries=150)
    ot = oracle.target(ps.residues, ps.offsets, k=7, kmer_thr=122)
    lens = ps.lengths()
    db_res = int(ps.offsets[-1])
    n_hits = n_acc = 0
    for q in range(ps.n):
        seq = ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        ids, _, _, _ = ot.prefilter(seq, identity_id=q, kmer_thr=122, max_hits=300, bin_size=2)
        ids = ids[(lens[ids.astype(np.int64)].astype(np.float32) / np.float32(lens[q])) >= np.float32(0.8)]
        n_hits += len(ids)
        for t in ids:
            t = int(t)
            if t == q:
                n_acc += 1
                continue
            r = oracle.sw_align(seq, ps.residues[int(ps.offsets[t]):int(ps.offsets[t + 1])], db_res)
            if r['btLen'] > 0 and r['qStart'] >= 0 and r['evalue'] <= 10.0 and r['btLen'] >= 30 and \
                    np.float32(r['qEnd'] - r['qStart'] + 1) / np.float32(lens[q]) >= np.float32(0.8):
                n_acc += 1
    assert cs.stats['prefilter_hits'] == n_hits, (cs.stats['prefilter_hits'], n_hits)
    assert out['accepted'] == n_acc, (out['accepted'], n_acc)
    assert n_acc > ps.n


def test_best_hit_by_set_on_the_device_equals_host_selection(gpu, host, monkeypatch):
    """besthitbyset's choice left to the device (sd_sw_align_batch_best_by_group: per (query, target set) only the first accepted
    alignment in Matcher::compareHits order and the identity pair come back) against the host's selection over all accepted records:
    proteomes with few families, so that a target set holds several homologs of a query (paralogs: up to a dozen candidates per
    cell, equal scores included) -- the same entries, hits, P-value bits and clusters; fewer records cross the bus"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(5, genes_per_proteome=400, n_families=60, seed=77)
    db = SetDB.from_proteomes(ps)
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('SD_BEST_ON_DEVICE', mode)
        cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
        outs[mode] = cs.search(db, same_db=True, chunk_queries=500)
        del cs
    a, b = outs['0'], outs['1']
    assert a['matched_hits'] > 1000 and a['clusters'] > 20
    assert b['accepted'] < a['accepted']   # (what the aggregation saw)
    for k_ in ('entries', 'matched_hits', 'clusters', 'cluster_hits'):
        assert a[k_] == b[k_], k_
    for k_ in ('entry_q', 'entry_t', 'entry_off', 'hit_q', 'hit_t'):
        assert np.array_equal(a[k_], b[k_]), k_
    assert a['hit_pval'].tobytes() == b['hit_pval'].tobytes()
    for k_ in ('cluster_of', 'rank', 'n_clusters', 'size'):
        assert np.array_equal(a['cluster_out'][k_], b['cluster_out'][k_]), k_
    assert a['cluster_out']['pCO'].tobytes() == b['cluster_out']['pCO'].tobytes()
    assert a['cluster_out']['pMH'].tobytes() == b['cluster_out']['pMH'].tobytes()


def _paralog_sets(n_sets=3, n_fam=40, seed=5):
    """sets whose genes come in near-identical paralog pairs of 650 - 1 000 residues: ancestor, and a copy with zero to three
    substitutions and up to two residues trimmed -- raw scores far beyond the ~2 700 where the E-value of a pair is 0.0"""
    from spacedust_amd.synth import ProteomeSet, _BG
    rng = np.random.default_rng(seed)
    fams = [rng.choice(20, size=int(rng.integers(650, 1001)), p=_BG).astype(np.uint8) for _ in range(n_fam)]
    seqs, set_id, pos, fam = [], [], [], []
    for s in range(n_sets):
        order = rng.permutation(n_fam)
        p = 0
        for f in order:
            copies = int(rng.integers(2, 4))   # two or three paralogs of the family in this set
            for c in range(copies):
                a = fams[f].copy()
                for _ in range(int(rng.integers(0, 4)) if (c or s) else 0):
                    a[int(rng.integers(0, len(a)))] = rng.integers(0, 20)
                trim = int(rng.integers(0, 3)) if c else 0
                a = a[:len(a) - trim]
                seqs.append(a)
                set_id.append(s)
                pos.append(p)
                fam.append(f)
                p += 1
    off = np.zeros(len(seqs) + 1, np.uint64)
    np.cumsum([len(x) for x in seqs], out=off[1:])
    return ProteomeSet(np.concatenate(seqs), off, np.array(set_id, np.uint32), np.array(pos, np.uint32),
                       np.ones(len(seqs), np.uint8), np.array(fam, np.int64), n_sets)


def test_best_hit_by_set_on_the_device_with_evalues_of_zero(gpu, host, monkeypatch):
    """K m n exp(-lambda S) is exactly 0.0 beyond a raw score of ~2 700, and Matcher::compareHits (Matcher.h:157-168) then
    orders a cell's alignments by the ROUNDED bit score, the shorter target and the smaller key -- not by the raw score the device's
    table maximises.  Sets with two or three near-identical paralogs of 650 - 1 000 residues per family: the device must leave the
    choice among such pairs to the host's compareHits (every accepted pair with an E-value below BB_EVAL_EXACT comes back), so
    entries, hits, P-value bits and clusters equal the selection over all accepted records"""
    ps = _paralog_sets()
    db = SetDB.from_proteomes(ps)
    outs = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('SD_BEST_ON_DEVICE', mode)
        cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
        outs[mode] = cs.search(db, same_db=True, chunk_queries=500)
        del cs
    # (informational A/B: the round-5 device selection, raw score first -- run with `pytest -s` to see whether this data separates the two)
    monkeypatch.setenv('SD_BEST_EXACT', '0')
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
    old = cs.search(db, same_db=True, chunk_queries=500)
    del cs
    monkeypatch.delenv('SD_BEST_EXACT')
    print('raw-score selection differs from compareHits on this data: hit_t %s, accepted %d vs %d' %
          (not np.array_equal(old['hit_t'], outs['0']['hit_t']), old['accepted'], outs['1']['accepted']))
    a, b = outs['0'], outs['1']
    assert a['matched_hits'] > 100
    for k_ in ('entries', 'matched_hits', 'clusters', 'cluster_hits'):
        assert a[k_] == b[k_], k_
    for k_ in ('entry_q', 'entry_t', 'entry_off', 'hit_q', 'hit_t'):
        assert np.array_equal(a[k_], b[k_]), k_
    assert a['hit_pval'].tobytes() == b['hit_pval'].tobytes()
    for k_ in ('cluster_of', 'rank', 'n_clusters', 'size'):
        assert np.array_equal(a['cluster_out'][k_], b['cluster_out'][k_]), k_
    assert a['cluster_out']['pCO'].tobytes() == b['cluster_out']['pCO'].tobytes()
    assert a['cluster_out']['pMH'].tobytes() == b['cluster_out']['pMH'].tobytes()
