"""GPU parity for profile queries through the C ABI: sd_host_map_profiles + sd_prefilter_profile_batch
and sd_profileset_create + sd_sw_align_batch against rows / alignments produced by the REAL reference classes
(tools/make_golden_profile.py) and, on larger random cases, against the oracle."""
import os

import numpy as np
import pytest

from spacedust_amd import api

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(params=['join', 'lookup'])
def profile_path(request, monkeypatch):
    """profile queries through the k-mer-major join (default since round 6: profile_kmers_kernel writes the join's stream elements) and
    through the per-k-mer lookup path (SD_PF_JOIN_PROFILE=0): the same rows and statistics either way"""
    if request.param == 'lookup':
        monkeypatch.setenv('SD_PF_JOIN_PROFILE', '0')
    else:
        monkeypatch.delenv('SD_PF_JOIN_PROFILE', raising=False)
    return request.param


def _golden(host):
    g = np.load(os.path.join(GOLD, 'profile_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    n = len(off) - 1
    res, off2 = host.map_sequences([blob[int(off[i]):int(off[i + 1])] for i in range(n)])
    assert (off2 == off).all()
    prof = host.map_profiles(g['profiles'].tobytes(), g['poff'])
    return g, res, off, prof


def test_profile_prefilter_matches_reference(gpu, host, profile_path):
    g, res, off, prof = _golden(host)
    assert host.profile_kmer_threshold(5.7, 6) == 99
    idx = host.build_index(res, off, 6, 0)   # profile searches index every non-X target k-mer (Prefiltering.cpp:525-527)
    tgt = api.Target(gpu, host, idx)
    for thr in (99, 80):
        par = api.prefilter_params(host, idx.n, kmer_thr=thr, max_hits=300, cov_thr=0.0, bin_size=2, k=6)
        ident = np.array([4 * q if q % 2 == 0 else 0xFFFFFFFF for q in range(len(prof['offsets']) - 1)], np.uint32)
        hits, cnt, st = api.prefilter_profile(gpu, tgt, par, prof, identity_id=ident, want_stats=True)
        rows = g['pf_rows_%d' % thr]
        for q in range(len(cnt)):
            exp = rows[rows[:, 0] == q]
            m = int(cnt[q])
            assert m == len(exp), (thr, q, m, len(exp))
            assert (hits[q, :m]['seqId'] == exp[:, 1]).all() and (hits[q, :m]['score'] == exp[:, 2]).all(), (thr, q)
            assert (hits[q, :m]['diagonal'].astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), (thr, q)


def test_profile_prefilter_matches_oracle_stats(gpu, host, oracle, profile_path):
    """k-mer / index-hit / diagonal counts per query as well (the oracle is pinned to the reference for these inputs)"""
    g, res, off, prof = _golden(host)
    idx = host.build_index(res, off, 6, 0)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, kmer_thr=70, max_hits=50, cov_thr=0.0, bin_size=2, k=6)
    hits, cnt, st = api.prefilter_profile(gpu, tgt, par, prof, want_stats=True)
    ot = oracle.target(res, off, k=6, kmer_thr=0)
    po = prof['offsets']
    total = 0
    for q in range(len(cnt)):
        a, b = int(po[q]), int(po[q + 1])
        ids, sc, dg, ost = ot.prefilter_profile(prof['letters'][a:b], prof['aln'][a:b], prof['sorted_score'][a:b],
                                                prof['sorted_index'][a:b], 70, max_hits=50)
        m = int(cnt[q])
        total += m
        assert m == len(ids) and (hits[q, :m]['seqId'] == ids).all() and (hits[q, :m]['score'] == sc).all(), q
        assert (hits[q, :m]['diagonal'] == dg).all(), q
        assert tuple(int(x) for x in st[q]) == tuple(int(x) for x in ost), (q, st[q], ost)
    assert total > 100


def test_profile_alignments_match_reference(gpu, host):
    g, res, off, prof = _golden(host)
    mat, _, _ = host.matrix(0)
    db = int(off[-1])
    qs = gpu.profileset(prof['letters'], prof['offsets'], prof['aln'])
    ts = gpu.seqset(res, off, None)
    par = gpu.sw_params(mat, db)
    pq, pt = g['sw_pairs'][:, 0].astype(np.uint32), g['sw_pairs'][:, 1].astype(np.uint32)
    out, pool = gpu.sw_align(par, qs, ts, pq, pt, identity=g['sw_pairs'][:, 2].astype(np.uint8))
    assert g['sw_pairs'][:, 2].sum() >= 5   # scoreIdentical pairs (the profile's own sequence in a same-DB search)
    bts = g['sw_bt'].tobytes().decode().split('\n')
    n_bt = 0
    for x in range(len(pq)):
        r, e = out[x], g['sw_res'][x]
        assert (int(r['score']), int(r['qEnd']), int(r['tEnd'])) == (e[0], e[2], e[4]), (x, r, e)
        assert (int(r['qStart']), int(r['tStart']), int(r['btLen'])) == (e[1], e[3], e[6]), (x, r, e)
        ev = g['evalue'][x]
        if ev <= 20.0:
            assert float(r['evalue']) == ev, (x, r['evalue'], ev)
        else:
            assert abs(float(r['evalue']) - ev) <= 1e-9 * ev, (x, r['evalue'], ev)
        if e[6] > 0:
            n_bt += 1
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == bts[x], (x, bt, bts[x])
            assert int(r['identical']) == e[5], x
    assert n_bt > 50


def test_profile_alignments_match_oracle_many(gpu, host, oracle, small_proteomes):
    """a larger batch: profiles derived from proteome sequences against many targets (all kernel classes, word reruns,
    band-doubling tracebacks), checked against the oracle"""
    ps = small_proteomes
    rng = np.random.default_rng(31)
    mat, _, _ = host.matrix(0)
    m = np.array([mat[i] for i in range(441)], np.int32).reshape(21, 21)
    qids = rng.choice(ps.n, 40, replace=False)
    recs, boff = [], [0]
    for q in qids:
        s = ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        rec = np.zeros((len(s), 25), np.uint8)
        scale = int(rng.integers(2, 6))
        for i, a in enumerate(s):
            row = m[min(int(a), 19), :20] * scale + rng.integers(-5, 6, 20)
            rec[i, :20] = np.clip(row, -128, 127).astype(np.int8).view(np.uint8)
            rec[i, 20] = a
            rec[i, 21] = int(np.argmax(row))
        recs.append(rec.tobytes())
        boff.append(boff[-1] + len(recs[-1]))
    prof = host.map_profiles(b''.join(recs), np.array(boff, np.uint64))
    db = int(ps.offsets[-1])
    qs = gpu.profileset(prof['letters'], prof['offsets'], prof['aln'])
    ts = gpu.seqset(ps.residues, ps.offsets, None)
    par = gpu.sw_params(mat, db)
    pq, pt = [], []
    for x, q in enumerate(qids):
        pq += [x] * 12
        pt += [int(q)] + [int(t) for t in rng.integers(0, ps.n, 11)]   # the source sequence itself scores high (word kernel)
    pq, pt = np.array(pq, np.uint32), np.array(pt, np.uint32)
    out, pool = gpu.sw_align(par, qs, ts, pq, pt)
    po = prof['offsets']
    n_bt = n_word = 0
    for x in range(len(pq)):
        a, b = int(po[pq[x]]), int(po[pq[x] + 1])
        t = ps.residues[int(ps.offsets[pt[x]]):int(ps.offsets[pt[x] + 1])]
        o = oracle.sw_align_profile(prof['letters'][a:b], prof['aln'][a:b], t, db)
        r = out[x]
        n_word += o['flags'] & 1
        assert int(r['score']) == o['score'], (x, r, o)
        assert (int(r['qStart']), int(r['qEnd']), int(r['tStart']), int(r['tEnd'])) == \
               (o['qStart'], o['qEnd'], o['tStart'], o['tEnd']), (x, r, o)
        assert int(r['btLen']) == o['btLen'], (x, r, o)
        if o['btLen'] > 0:
            n_bt += 1
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == o['backtrace'], (x, bt, o['backtrace'])
            assert int(r['identical']) == o['identical']
    assert n_bt >= 40 and n_word >= 10, (n_bt, n_word)


def test_profile_prefilter_large_kmer_lists(gpu, host, oracle, profile_path):
    """permissive threshold: > 10^5 similar k-mers per position on average, so positions leave the 65 536-entry scratch
    tier and are redone by the large tier; counts and hits against the oracle"""
    g, res, off, _ = _golden(host)
    poff = g['poff'][:4]
    prof = host.map_profiles(g['profiles'][:int(poff[-1])].tobytes(), poff)
    idx = host.build_index(res, off, 6, 0)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, kmer_thr=45, max_hits=50, cov_thr=0.0, bin_size=2, k=6)
    hits, cnt, st = api.prefilter_profile(gpu, tgt, par, prof, want_stats=True)
    ot = oracle.target(res, off, k=6, kmer_thr=0)
    po = prof['offsets']
    for q in range(len(cnt)):
        a, b = int(po[q]), int(po[q + 1])
        ids, sc, dg, ost = ot.prefilter_profile(prof['letters'][a:b], prof['aln'][a:b], prof['sorted_score'][a:b],
                                                prof['sorted_index'][a:b], 45, max_hits=50)
        m = int(cnt[q])
        assert int(st[q][0]) > 100000 * (b - a)
        assert m == len(ids) and (hits[q, :m]['seqId'] == ids).all() and (hits[q, :m]['score'] == sc).all(), q
        assert (hits[q, :m]['diagonal'] == dg).all(), q
        assert tuple(int(x) for x in st[q]) == tuple(int(x) for x in ost), (q, st[q], ost)


def test_pipeline_with_profile_queries(gpu, host, oracle, small_proteomes, monkeypatch):
    """one search iteration with a profile query DB through the pipeline (prefilter_profile -> profile SW -> aggregation
    -> clusterhits): prefilter hit count and accepted alignment count against the oracle run stage by stage"""
    from spacedust_amd.pipeline import SetDB, ClusterSearch
    monkeypatch.setenv('SD_EVAL_PUSHDOWN', '0')   # (the count of alignments accepted at -e 10 is what is compared)
    ps = small_proteomes
    rng = np.random.default_rng(17)
    mat, _, _ = host.matrix(0)
    m = np.array([mat[i] for i in range(441)], np.int32).reshape(21, 21)
    recs, boff = [], [0]
    for q in range(ps.n):
        s = ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        rec = np.zeros((len(s), 25), np.uint8)
        for i, a in enumerate(s):
            row = m[min(int(a), 19), :20] * 3 + rng.integers(-4, 5, 20)
            rec[i, :20] = np.clip(row, -128, 127).astype(np.int8).view(np.uint8)
            rec[i, 20] = a
            rec[i, 21] = int(np.argmax(row))
        recs.append(rec.tobytes())
        boff.append(boff[-1] + len(recs[-1]))
    prof = host.map_profiles(b''.join(recs), np.array(boff, np.uint64))
    db = SetDB.from_proteomes(ps)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, profile_queries=True, filter_self_match=True)
    assert cs.kmer_thr == 99
    out = cs.search(db.with_profiles(prof), same_db=True, chunk_queries=100)
    # the same, stage by stage, on the oracle
    ot = oracle.target(ps.residues, ps.offsets, k=6, kmer_thr=0)
    po = prof['offsets']
    dbres = int(ps.offsets[-1])
    n_hits = n_acc = 0
    for q in range(ps.n):
        a, b = int(po[q]), int(po[q + 1])
        ids, sc, dg, _ = ot.prefilter_profile(prof['letters'][a:b], prof['aln'][a:b], prof['sorted_score'][a:b],
                                              prof['sorted_index'][a:b], 99, max_hits=300, identity_id=q)
        assert len(ids) >= 1 and ids[0] == q
        # Util::canBeCovered, COV_MODE_QUERY (Prefiltering.cpp:856-863): float32 ratio of the lengths
        tl = (ps.offsets[ids.astype(np.int64) + 1] - ps.offsets[ids.astype(np.int64)]).astype(np.float32)
        ids = ids[(tl / np.float32(b - a)) >= np.float32(0.8)]
        n_hits += len(ids)
        for t in ids:
            ts = ps.residues[int(ps.offsets[t]):int(ps.offsets[t + 1])]
            r = oracle.sw_align_profile(prof['letters'][a:b], prof['aln'][a:b], ts, dbres, identity=bool(t == q))
            if t == q:
                n_acc += 1
                continue
            if r['btLen'] <= 0 or r['qStart'] < 0:
                continue
            qcov = (r['qEnd'] - r['qStart'] + 1) / float(b - a)
            if r['evalue'] <= 10.0 and qcov >= 0.8 and r['btLen'] >= 30:
                n_acc += 1
    assert cs.stats['prefilter_hits'] == n_hits, (cs.stats['prefilter_hits'], n_hits)
    assert out['accepted'] == n_acc, (out['accepted'], n_acc)
    assert n_acc > ps.n


def test_result2profile_sequence_weights_on_the_device(gpu):
    """sd_r2p_batch_device (csrc/hip/sd_r2p.hip: the position-specific sequence weights of PSSMCalculator on the GPU, everything
    in the reference's summation order) gives the bytes of the host implementation for all 5 898 regression queries -- and, where
    the library travelled, of the reference's own MultipleAlignment / MsaFilter / PSSMCalculator classes run on this box
    (oracle/_ref/libsdref_r2p.so: the approximate reciprocal differs between CPU models, so the comparison is made where the
    bytes are produced) -- plus the parameter corners and alignments that are deeper than the regression input's"""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_result2profile import _load, _edges
    from oracle.pyoracle import ref_r2p_available, RefResult2Profile
    api_, seqs, res, off, aln = _load()
    queries = list(range(len(seqs)))
    edge_off, et, eq, ets, bts = _edges(api_, aln, queries)
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(seqs[q]) for q in queries])
    host_bytes = api_.result2profile(res, qoff, edge_off, et, eq, ets, bts, res, off)
    dev_bytes = api_.result2profile(res, qoff, edge_off, et, eq, ets, bts, res, off, ctx=gpu)
    bad = [x for x in range(len(queries)) if host_bytes[int(qoff[x]) * 25:int(qoff[x + 1]) * 25] != dev_bytes[int(qoff[x]) * 25:int(qoff[x + 1]) * 25]]
    assert bad == [], (len(bad), bad[:10])
    if ref_r2p_available():
        ref = RefResult2Profile()
        nbad = 0
        for x in range(0, len(queries), 7):
            a, b = edge_off[x], edge_off[x + 1]
            nbad += ref.profile(seqs[x], [seqs[t] for t in et[a:b]], eq[a:b], ets[a:b], bts[a:b]) != dev_bytes[int(qoff[x]) * 25:int(qoff[x + 1]) * 25]
        assert nbad == 0
    sample = [q for q in queries if edge_off[q + 1] - edge_off[q] >= 2][:60]
    for kw in (dict(filter_msa=0), dict(comp_bias=0, mask_profile=0), dict(max_seq_id=0.5, ndiff=3), dict(qid='0.0,0.3,0.6', cov=0.5), dict(wg=1)):
        for q in sample:
            a, b = edge_off[q], edge_off[q + 1]
            args = (res[int(off[q]):int(off[q + 1])], [0, len(seqs[q])], [0, b - a], et[a:b], eq[a:b], ets[a:b], bts[a:b], res, off)
            assert api_.result2profile(*args, **kw) == api_.result2profile(*args, ctx=gpu, **kw), (kw, q)
    # a deep alignment: one centre against 1 500 mutated copies with indels (rows start and end at many different columns)
    rng = np.random.default_rng(77)
    aa = 'ACDEFGHIKLMNPQRSTVWY'
    base = ''.join(rng.choice(list(aa), 700))
    fam = [base]
    for _ in range(1500):
        lo = int(rng.integers(0, 200))
        hi = int(rng.integers(500, 700))
        sq = list(base[lo:hi])
        for p_ in np.nonzero(rng.random(len(sq)) < rng.uniform(0.05, 0.4))[0]:
            sq[p_] = aa[rng.integers(20)]
        for _d in range(int(rng.integers(0, 4))):
            p_ = int(rng.integers(10, len(sq) - 10))
            del sq[p_:p_ + int(rng.integers(1, 6))]
        fam.append(''.join(sq))
    host = api_.Host()
    fres, foff = host.map_sequences(fam)
    sw_b, _, _ = host.comp_bias(fres, foff)
    ss = gpu.seqset(fres, foff, sw_b)
    par = gpu.sw_params(host.matrix(0)[0], int(foff[-1]))
    pq = np.zeros(1500, np.uint32)
    pt = np.arange(1, 1501, dtype=np.uint32)
    out, pool = gpu.sw_align(par, ss, ss, pq, pt)
    keep = [i for i in range(1500) if int(out[i]['btLen']) > 0]   # (fragments below the query coverage gate have no backtrace)
    assert len(keep) > 300
    bts2 = [pool[int(out[i]['btOffset']):int(out[i]['btOffset']) + int(out[i]['btLen'])].tobytes().decode() for i in keep]
    args = (fres[:int(foff[1])], [0, int(foff[1])], [0, len(keep)], [int(pt[i]) for i in keep], [int(out[i]['qStart']) for i in keep],
            [int(out[i]['tStart']) for i in keep], bts2, fres, foff)
    for kw in (dict(), dict(filter_msa=0)):
        assert api_.result2profile(*args, **kw) == api_.result2profile(*args, ctx=gpu, **kw), kw
