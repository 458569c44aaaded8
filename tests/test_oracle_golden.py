"""CPU: the oracle restatement (and the product's host stages, which share the support code) against golden
vectors produced by the REAL reference code (tools/make_golden.py -> tests/golden/reference_vectors.npz) on the
reference's own regression input, and against the reference's known answers."""
import gzip
import hashlib
import os

import numpy as np
import pytest

from oracle.pyoracle import oracle_clusterhits

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLD, 'reference_vectors.npz'))


@pytest.fixture(scope='module')
def examples(oracle):
    seqs = []
    for f in ('NC_000913.faa.gz', 'NC_000915.faa.gz'):
        cur = None
        with gzip.open(os.path.join(GOLD, 'examples', f), 'rt') as fh:
            for line in fh:
                if line.startswith('>'):
                    if cur is not None:
                        seqs.append(''.join(cur))
                    cur = []
                else:
                    cur.append(line.rstrip('\n'))
        seqs.append(''.join(cur))
    nums = [oracle.map_sequence(s) for s in seqs]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return seqs, nums, off


def test_matrices_match_reference(oracle, host, gold):
    for w, name in ((0, 'blosum62_2'), (1, 'vtml80_8_m02'), (2, 'blosum62_2_m02')):
        m, pb, a2n = oracle.matrix(w)
        assert (m == gold['mat_' + name]).all()
        assert (pb == gold['pback_' + name]).all()
        assert (a2n[:255] == gold['aa2num']).all()
        hm, hpb, _ = host.matrix(w)
        assert (hm.reshape(21, 21) == gold['mat_' + name]).all() and (hpb == gold['pback_' + name]).all()


def test_composition_bias_bitwise(oracle, gold, examples):
    _, nums, _ = examples
    for w in (0, 1):
        got = np.concatenate([oracle.compbias(w, nums[i]) for i in gold['cb_sample']])
        assert (got.view(np.uint32) == gold['cb_%d' % w].view(np.uint32)).all()


def test_similar_kmer_lists(oracle, gold):
    off = gold['kmer_list_off']
    for i in range(len(gold['kmer_thr'])):
        got = oracle.kmer_list(gold['kmer_windows'][i], int(gold['kmer_thr'][i]))
        assert (got == gold['kmer_lists'][off[i]:off[i + 1]]).all()


def test_extended_matrices(oracle, gold):
    for w in (2, 3):
        sc, ix = oracle.ext_matrix(w)
        d = hashlib.md5(sc.astype(np.int16).tobytes() + ix.astype(np.uint16).tobytes()).digest()
        assert (np.frombuffer(d, np.uint8) == gold['ext%d_md5' % w]).all()


@pytest.fixture(scope='module')
def target(oracle, examples):
    _, nums, off = examples
    return oracle.target(np.concatenate(nums), off)


def test_masking_and_index(oracle, gold, examples, target):
    _, nums, off = examples
    # reference known answers on the regression input: 1 784 989 index entries, 11 546 masked residues
    assert (target.n_entries, target.masked_residues) == (1784989, 11546)
    assert tuple(int(x) for x in gold['index_stats']) == (1784989, 11546)
    o, es, ep, mk = target.dump()
    assert (np.nonzero(mk != np.concatenate(nums))[0] == gold['masked_positions']).all()
    d = hashlib.md5(o.astype(np.uint32).tobytes() + es.tobytes() + ep.tobytes()).digest()
    assert (np.frombuffer(d, np.uint8) == gold['index_md5']).all()


def test_prefilter_hits(oracle, gold, examples, target):
    _, nums, _ = examples
    rows = gold['pf_rows']
    for q in np.unique(rows[:, 0]):
        exp = rows[rows[:, 0] == q]
        ids, sc, dg, _ = target.prefilter(nums[q], identity_id=int(q))
        assert (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all() and (dg == exp[:, 3]).all(), q


def test_smith_waterman_alignments(oracle, gold, examples):
    _, nums, off = examples
    rows, ev = gold['sw_rows'], gold['sw_eval']
    bts = gold['sw_bt'].tobytes().decode().split('\n')
    db = int(off[-1])
    for x in range(0, len(rows), 3):
        q, t, score, qs, qe, ts, te, ident, btl = (int(v) for v in rows[x])
        o = oracle.sw_align(nums[q], nums[t], db, identity=(q == t))
        assert (o['score'], o['qStart'], o['qEnd'], o['tStart'], o['tEnd'], o['btLen']) == (score, qs, qe, ts, te, btl), x
        assert o['evalue'] == ev[x]
        if btl > 0:
            assert o['backtrace'] == bts[x] and o['identical'] == ident


def test_evalue_bitscore(oracle, host, gold):
    for s, l, e in gold['evalue_samples']:
        assert oracle.evalue(1861962, s, l) == e
        assert host.evalue(1861962, s, l) == e
    for s, b in gold['bitscore_samples']:
        assert oracle.bitscore(s) == b and host.bitscore(s) == b


def test_clusterhits_regression_known_answers(oracle):
    """the two match entries of run_regression.sh (K = 732, 551): 108 clusters, 308 member hits, 2 clusters with
    P < 1E-20 (R/util/run_regression.sh:20-23), and the canonical TSV whose md5 equals the reference binary's."""
    g = np.load(os.path.join(GOLD, 'config1_matches.npz'))
    off = g['entry_off']
    names = g['names'].tobytes().decode().split('\n')
    text = g['hit_text'].tobytes().decode().split('\n')
    files = ['NC_000913.faa', 'NC_000915.faa']
    lines, n_clu, n_hit, n_sig = [], 0, 0, 0
    for e in range(len(off) - 1):
        a, b = int(off[e]), int(off[e + 1])
        cof, mo, cs, pco, pmh, _ = oracle_clusterhits(oracle, g['q_pos'][a:b], g['t_pos'][a:b], g['strands'][a:b],
                                                      g['pval'][a:b], int(g['nq'][e]))
        w = 0
        for c in range(len(cs)):
            lines.append('%s\t%s\t%.3E\t%.3E\t%d\n' % (files[g['entry_q'][e]], files[g['entry_t'][e]], pco[c], pmh[c], cs[c]))
            n_sig += pco[c] < 1e-20
            for j in range(cs[c]):
                h = a + int(mo[w + j])
                lines.append('%s\t%s\n' % (names[g['hit_t'][h]], text[h]))
                n_hit += 1
            w += cs[c]
        n_clu += len(cs)
    assert (n_hit, n_sig, n_clu) == (308, 2, 108)
    lines.sort(key=lambda s: s.encode())
    assert hashlib.md5(''.join(lines).encode()).hexdigest() == 'abb28ee37bc130a5f09a9f767ef00ccf'
    assert ''.join(lines) == open(os.path.join(GOLD, 'config1_canonical.tsv')).read()


def test_prefilter_rescoring_path(oracle):
    """more than maxHits targets saturate the 8-bit diagonal score: rescoreHits / rescaled scores
    (QueryMatcher.cpp:157-170,525-544); expected rows from the real reference (tools/make_golden_rescore.py)"""
    g = np.load(os.path.join(GOLD, 'rescore_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    tgt = oracle.target(np.concatenate(nums), off)
    rows = g['pf_rows']
    assert ((rows[:, 2] >= 255) & (rows[:, 2] < 65535)).sum() > 1000
    for q in g['queries']:
        exp = rows[rows[:, 0] == q]
        ids, sc, dg, _ = tgt.prefilter(nums[q], identity_id=int(q), max_hits=300)
        assert len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all() and (dg == exp[:, 3]).all(), q


def test_k7_kmer_lists_and_prefilter(oracle):
    """k = 7: similar k-mer lists (order included) for 12 windows and prefilter rows at two thresholds, all from the
    real reference classes (tools/make_golden_k7.py)"""
    g = np.load(os.path.join(GOLD, 'k7_vectors.npz'))
    lo = g['list_off']
    for i, (w, t) in enumerate(zip(g['windows'], g['window_thr'])):
        exp = g['lists'][int(lo[i]):int(lo[i + 1])]
        got = np.asarray(oracle.kmer_list(w, int(t), k=7), np.uint32)
        assert len(got) == len(exp) and (got == exp).all(), i
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    for thr in (122, 100):
        tgt = oracle.target(np.concatenate(nums), off, k=7, kmer_thr=thr)
        rows = g['pf_rows_%d' % thr]
        assert len(rows) > 2 * len(g['queries'])
        for q in (g['queries'] if thr == 122 else g['queries'][:3]):   # the permissive lists are slow on one core
            exp = rows[rows[:, 0] == q]
            ids, sc, dg, _ = tgt.prefilter(nums[q], identity_id=int(q), kmer_thr=thr, max_hits=300)
            assert len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all(), (thr, q)
            assert (dg.astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), (thr, q)


def test_sw_modes_and_gates(oracle):
    """alignment modes 0-2, coverage modes 0-2, E-value thresholds: rows from the real reference (tools/make_golden_modes.py)"""
    g = np.load(os.path.join(GOLD, 'sw_modes_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    db = int(off[-1])
    for mi, (sw_mode, cov_mode, cov_thr, eval_thr) in enumerate(g['modes']):
        rows, evs = g['res_%d' % mi], g['ev_%d' % mi]
        bts = g['bt_%d' % mi].tobytes().decode().split('\n')
        for x, (a, b) in enumerate(g['pairs']):
            r = oracle.sw_align(nums[a], nums[b], db, sw_mode=int(sw_mode), eval_thr=float(eval_thr), cov_mode=int(cov_mode),
                                cov_thr=float(cov_thr))
            e = rows[x]
            assert (r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['btLen']) == (e[0], e[1], e[2], e[3], e[4], e[6]), (mi, x)
            assert r['evalue'] == evs[x], (mi, x)
            if e[6] > 0:
                assert r['identical'] == e[5] and r['backtrace'] == bts[x], (mi, x)


def test_stress_db(oracle):
    """exact duplicates (ties at every cut), tandem repeats / low complexity (tantan), X-rich, 1 to 3 000 residues: index
    size and masking, prefilter rows at list lengths 300 and 7, 1 219 alignments -- rows from the real reference
    (tools/make_golden_stress.py)"""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from stress_db import stress_db
    g = np.load(os.path.join(GOLD, 'stress_vectors.npz'))
    seqs = stress_db()
    nums = [oracle.map_sequence(s) for s in seqs]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    tgt = oracle.target(np.concatenate(nums), off)
    assert tgt.n_entries == int(g['n_entries'][0]) and tgt.masked_residues == int(g['masked'][0])
    for mh in (300, 7):
        rows = g['pf_rows_%d' % mh]
        for q in range(len(seqs)):
            exp = rows[rows[:, 0] == q]
            ids, sc, dg, _ = tgt.prefilter(nums[q], identity_id=q, max_hits=mh)
            assert len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all(), (mh, q)
            assert (dg.astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), (mh, q)
    bts = g['sw_bt'].tobytes().decode().split('\n')
    db = int(off[-1])
    for x, (a, b) in enumerate(g['sw_pairs']):
        r = oracle.sw_align(nums[a], nums[b], db, identity=bool(a == b))
        e = g['sw_res'][x]
        assert (r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['btLen']) == (e[0], e[1], e[2], e[3], e[4], e[6]), x
        assert r['evalue'] == g['sw_ev'][x], x
        if e[6] > 0:
            assert r['identical'] == e[5] and r['backtrace'] == bts[x], x


def _load_profiles(oracle):
    g = np.load(os.path.join(GOLD, 'profile_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    poff = g['poff']
    profs = [oracle.map_profile(g['profiles'][int(poff[i]):int(poff[i + 1])].tobytes()) for i in range(len(poff) - 1)]
    return g, nums, profs


def test_profile_queries(oracle):
    """profile queries (a22): per-position k-mer lists, prefilter rows at two thresholds and full alignments, all from
    the real reference classes (tools/make_golden_profile.py)"""
    g, nums, profs = _load_profiles(oracle)
    seed = [0, 1, 3, 5, 8, 9]
    lo = g['list_off']
    for i, (pi, pos, thr) in enumerate(g['windows']):
        letters, cons, aln, ssc, six = profs[pi]
        rows = [pos + s for s in seed]
        got = oracle.profile_kmer_list(ssc[rows], six[rows], int(thr))
        exp = g['lists'][int(lo[i]):int(lo[i + 1])]
        assert len(got) == len(exp) and (got == exp).all(), i
    tgt = oracle.target(np.concatenate(nums), g['off'], k=6, kmer_thr=0)
    for thr in (99, 80):
        rows = g['pf_rows_%d' % thr]
        assert len(rows) > 3 * len(profs)
        for qi, (letters, cons, aln, ssc, six) in enumerate(profs):
            exp = rows[rows[:, 0] == qi]
            ids, sc, dg, _ = tgt.prefilter_profile(letters, aln, ssc, six, thr, identity_id=4 * qi if qi % 2 == 0 else 0xFFFFFFFF)
            assert len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all(), (thr, qi)
            assert (dg.astype(np.int64) == (exp[:, 3] & 0xFFFF)).all(), (thr, qi)
    bts = g['sw_bt'].tobytes().decode().split('\n')
    db_res = int(g['off'][-1])
    n_bt = 0
    for x, (qi, t, ident) in enumerate(g['sw_pairs']):
        letters, cons, aln, ssc, six = profs[qi]
        r = oracle.sw_align_profile(letters, aln, nums[t], db_res, identity=bool(ident))
        e = g['sw_res'][x]
        assert (r['score'], r['qEnd'], r['tEnd']) == (e[0], e[2], e[4]), x
        assert r['evalue'] == g['evalue'][x], x
        assert (r['qStart'], r['tStart'], r['btLen']) == (e[1], e[3], e[6]), x
        if e[6] > 0:
            assert r['identical'] == e[5] and r['backtrace'] == bts[x], x
            n_bt += 1
    assert n_bt > 50


def test_prefilter_hit_buffer_overflow(oracle):
    """a query with more index hits than the reference's hit buffer (2*max(1e6, #targets)): one overflow, two match
    parts, merged result lists (QueryMatcher.cpp:281-326); rows from the real reference (tools/make_golden_overflow.py)"""
    g = np.load(os.path.join(GOLD, 'overflow_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]
    tgt = oracle.target(np.concatenate(nums), off)
    rows = g['pf_rows']
    assert int(g['index_hits_q0'][0]) >= 2000000
    for q in g['queries']:
        exp = rows[rows[:, 0] == q]
        ids, sc, dg, st = tgt.prefilter(nums[q], identity_id=int(q), max_hits=300)
        assert len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all() and (dg == (exp[:, 3] & 0xFFFF)).all(), q


def test_long_sequences_and_long_gaps(oracle):
    """sequences of 32 768 residues and more (computeLongScore, UngappedAlignment.cpp:312-329) and alignments across gaps
    of > 1 000 residues (band doubling from |tLen - qLen| + 1, StripedSmithWaterman.cpp banded traceback); rows from the
    real reference (tools/make_golden_long.py)"""
    g = np.load(os.path.join(GOLD, 'long_vectors.npz'))
    off = g['off']
    blob = g['blob'].tobytes().decode()
    n = len(off) - 1
    nums = [oracle.map_sequence(blob[int(off[i]):int(off[i + 1])]) for i in range(n)]
    assert max(len(x) for x in nums) >= 32768
    tgt = oracle.target(np.concatenate(nums), off)
    rows = g['pf_rows']
    for q in range(n):
        exp = rows[rows[:, 0] == q]
        ids, sc, dg, st = tgt.prefilter(nums[q], identity_id=q, max_hits=300)
        assert len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all() and (dg == (exp[:, 3] & 0xFFFF)).all(), q
    for x in range(len(g['gap_q'])):
        o = oracle.sw_align(oracle.map_sequence(str(g['gap_q'][x])), oracle.map_sequence(str(g['gap_t'][x])), 10 ** 7, cov_thr=0.0)
        got = (o['score'], o['qStart'], o['qEnd'], o['tStart'], o['tEnd'], o['identical'], o['btLen'])
        assert got == tuple(int(v) for v in g['gap_res'][x]), (x, got)
        assert o['backtrace'] == str(g['gap_bt'][x])
        assert o['evalue'] == float(g['gap_eval'][x])
    # scores beyond int16: the word kernel saturates at 32 767 (simdi16_adds) and reports the first cell that reaches it
    assert len(g['sat_q']) == 3 and int(g['sat_res'][:, 0].min()) == 32767
    for x in range(len(g['sat_q'])):
        o = oracle.sw_align(oracle.map_sequence(str(g['sat_q'][x])), oracle.map_sequence(str(g['sat_t'][x])), 10 ** 7, cov_thr=0.0)
        got = (o['score'], o['qStart'], o['qEnd'], o['tStart'], o['tEnd'], o['identical'], o['btLen'])
        assert got == tuple(int(v) for v in g['sat_res'][x]), (x, got)
        assert o['backtrace'] == str(g['sat_bt'][x])
