"""Edge cases through the C ABI against the oracle (itself equal to the real reference classes on these inputs):
queries shorter than the seed span, of exactly the span, all-X sequences, one- and two-residue sequences, a query
without any hit, an X run inside a sequence; empty batches."""
import numpy as np
import pytest

from spacedust_amd import api

pytestmark = pytest.mark.gpu
AA = 'ACDEFGHIKLMNPQRSTVWY'


def _db(oracle):
    rng = np.random.default_rng(3)
    base = ''.join(rng.choice(list(AA), 200))
    seqs = [base, base[:5], base[:10], 'X' * 40, base[20:31], 'A', 'AC', base[::-1], base[:100] + 'X' * 30 + base[130:], base,
            base[:9], 'X' + base[1:60], base[:60] + 'X']
    nums = [oracle.map_sequence(s) for s in seqs]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return seqs, nums, np.concatenate(nums), off


def test_prefilter_edge_cases(gpu, host, oracle):
    seqs, nums, res, off = _db(oracle)
    n = len(seqs)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    tgt = api.Target(gpu, host, idx)
    par = api.prefilter_params(host, idx.n, max_hits=300, cov_thr=0.0, bin_size=2)
    hits, cnt, st = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, np.arange(n, dtype=np.uint32), want_stats=True)
    ot = oracle.target(res, off)
    assert ot.n_entries == idx.n_entries
    for q in range(n):
        ids, sc, dg, ost = ot.prefilter(nums[q], identity_id=q, max_hits=300)
        m = int(cnt[q])
        assert m == len(ids), (q, m, len(ids))
        assert (hits[q, :m]['seqId'] == ids).all() and (hits[q, :m]['score'] == sc).all() and (hits[q, :m]['diagonal'] == dg).all(), q
        assert tuple(int(x) for x in st[q]) == tuple(int(x) for x in ost), q
    # the same queries without an identity target: the short / all-X ones have no hit at all
    hits2, cnt2, _ = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, np.full(n, 0xFFFFFFFF, np.uint32))
    for q in range(n):
        ids, sc, dg, _ = ot.prefilter(nums[q], max_hits=300)
        assert int(cnt2[q]) == len(ids) and (hits2[q, :len(ids)]['seqId'] == ids).all(), q
    assert int(cnt2[1]) == 0 and int(cnt2[3]) == 0 and int(cnt2[5]) == 0
    # empty batch
    e = api.prefilter(gpu, tgt, par, res[:0], off[:1], km_b[:0], dg_b[:0], np.zeros(0, np.uint32))
    assert e[0].shape[0] == 0 and len(e[1]) == 0


def test_alignment_edge_cases(gpu, host, oracle):
    seqs, nums, res, off = _db(oracle)
    sw_b, _, _ = host.comp_bias(res, off)
    mat, _, _ = host.matrix(0)
    db = int(off[-1])
    ss = gpu.seqset(res, off, sw_b)
    par = gpu.sw_params(mat, db)
    pairs = [(5, 6), (5, 0), (0, 5), (6, 6), (1, 0), (3, 0), (0, 3), (8, 0), (0, 9), (2, 4), (4, 2), (10, 0), (11, 12), (12, 11),
             (3, 3), (5, 5), (0, 0), (7, 0)]
    pq = np.array([a for a, _ in pairs], np.uint32)
    pt = np.array([b for _, b in pairs], np.uint32)
    ident = np.array([1 if (a == b and a in (0, 5)) else 0 for a, b in pairs], np.uint8)   # scoreIdentical on a 200-mer and a 1-mer
    out, pool = gpu.sw_align(par, ss, ss, pq, pt, identity=ident)
    for x, (a, b) in enumerate(pairs):
        o = oracle.sw_align(nums[a], nums[b], db, identity=bool(ident[x]))
        r = out[x]
        assert int(r['score']) == o['score'], (x, r, o)
        assert (int(r['qEnd']), int(r['tEnd'])) == (o['qEnd'], o['tEnd']), (x, r, o)
        assert (int(r['qStart']), int(r['tStart']), int(r['btLen'])) == (o['qStart'], o['tStart'], o['btLen']), (x, r, o)
        if o['btLen'] > 0:
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == o['backtrace'] and int(r['identical']) == o['identical'], (x, bt, o)
    # no pairs at all
    out0, pool0 = gpu.sw_align(par, ss, ss, np.zeros(0, np.uint32), np.zeros(0, np.uint32))
    assert len(out0) == 0


def test_device_composition_bias_equals_host(gpu, host, oracle, small_proteomes):
    """sd_comp_bias_batch (integer window sums on the device + the host's correction table) against sd_host_comp_bias,
    which is pinned bitwise to the reference's calcLocalAaBiasCorrection: all three arrays, every residue, including
    sequences shorter than the window, of exactly the window, X runs, and both seed patterns"""
    seqs, nums, res, off = _db(oracle)
    rng = np.random.default_rng(12)
    extra = [''.join(rng.choice(list(AA), L)) for L in (19, 20, 21, 39, 40, 41, 42, 400)]
    nums2 = nums + [oracle.map_sequence(s) for s in extra]
    off2 = np.zeros(len(nums2) + 1, np.uint64)
    off2[1:] = np.cumsum([len(x) for x in nums2])
    res2 = np.concatenate(nums2)
    ps = small_proteomes
    for r, o in ((res2, off2), (ps.residues, ps.offsets)):
        for k in (6, 7):
            hs, hd, hk = host.comp_bias(r, o, k)
            ds, dd, dk = gpu.comp_bias(host, r, o, k)
            assert np.array_equal(hs, ds), k
            assert np.array_equal(hd, dd), k
            assert np.array_equal(hk, dk), k
    assert np.abs(hs).max() > 0 and np.abs(hk).max() > 0


def test_stress_db_matches_reference(gpu, host, oracle):
    """the stress DB (ties from exact duplicates, masked repeats, X-rich, 1 to 3 000 residues) through the C ABI against
    rows and alignments produced by the real reference (tools/make_golden_stress.py)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from stress_db import stress_db
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'stress_vectors.npz'))
    seqs = stress_db()
    res, off = host.map_sequences(seqs)
    n = len(seqs)
    sw_b, dg_b, km_b = host.comp_bias(res, off)
    idx = host.build_index(res, off)
    assert idx.n_entries == int(g['n_entries'][0]) and idx.masked_residues == int(g['masked'][0])
    tgt = api.Target(gpu, host, idx)
    for mh in (300, 7):
        par = api.prefilter_params(host, idx.n, max_hits=mh, cov_thr=0.0, bin_size=2)
        hits, cnt, _ = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, np.arange(n, dtype=np.uint32))
        rows = g['pf_rows_%d' % mh]
        for q in range(n):
            exp = rows[rows[:, 0] == q]
            m = int(cnt[q])
            assert m == len(exp), (mh, q, m, len(exp))
            assert (hits[q, :m]['seqId'] == exp[:, 1]).all() and (hits[q, :m]['score'] == exp[:, 2]).all(), (mh, q)
            assert (hits[q, :m]['diagonal'].astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), (mh, q)
    mat, _, _ = host.matrix(0)
    db = int(off[-1])
    ss = gpu.seqset(res, off, sw_b)
    spar = gpu.sw_params(mat, db)
    pq = g['sw_pairs'][:, 0].astype(np.uint32)
    pt = g['sw_pairs'][:, 1].astype(np.uint32)
    out, pool = gpu.sw_align(spar, ss, ss, pq, pt, identity=(pq == pt))
    bts = g['sw_bt'].tobytes().decode().split('\n')
    for x in range(len(pq)):
        r, e = out[x], g['sw_res'][x]
        assert (int(r['score']), int(r['qEnd']), int(r['tEnd']), int(r['btLen'])) == (e[0], e[2], e[4], e[6]), (x, r, e)
        assert (int(r['qStart']), int(r['tStart'])) == (e[1], e[3]), (x, r, e)
        ev = g['sw_ev'][x]
        if ev <= 20.0:
            assert float(r['evalue']) == ev, (x, r['evalue'], ev)
        if e[6] > 0:
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == bts[x] and int(r['identical']) == e[5], x


def test_sequences_beyond_65535_residues_are_rejected(gpu, host):
    """index and k-mer positions are 16 bit (IndexEntryLocal::position_j): a longer sequence is SD_EINVAL at every door,
    never wrapped"""
    from spacedust_amd._lib import SdError
    res = np.random.default_rng(1).integers(0, 20, 70000).astype(np.uint8)
    off = np.array([0, 66000, 70000], np.uint64)
    with pytest.raises(SdError):
        host.build_index(res, off)
    with pytest.raises(SdError, match='65535'):
        gpu.seqset(res, off, np.zeros(len(res), np.int8))
