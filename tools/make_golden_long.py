#!/usr/bin/env python3
"""Golden vectors for sequences of 32 768 residues and more: the prefilter's 16-bit diagonal is ambiguous there and the
reference scores every real diagonal it can stand for (computeLongScore, UngappedAlignment.cpp:225-232,300-329), and the
banded traceback meets bands beyond 2 047 columns (gaps of > 1 000 residues inside one alignment).  Crafted DB run through
the REAL reference classes (oracle/_ref/libsdref.so).  Dev container only:  python tools/make_golden_long.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref, RefSW  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
AA = 'ACDEFGHIKLMNPQRSTVWY'


def mutate(rng, s, rate):
    s = list(s)
    for p in np.nonzero(rng.random(len(s)) < rate)[0]:
        s[p] = AA[rng.integers(20)]
    return ''.join(s)


def crafted(seed=77):
    rng = np.random.default_rng(seed)
    rnd = lambda n: ''.join(rng.choice(list(AA), n))
    t0 = rnd(40000)
    t1 = rnd(3000) + mutate(rng, t0[20000:38000], 0.10) + rnd(13000)
    seqs = [t0, t1]
    for p in (100, 9000, 16000, 20000, 27000, 32700, 33000, 34500, 36000, 37500, 39000, 39600):
        seqs.append(mutate(rng, t0[p:p + 400], 0.15))
    # fragments whose copy sits at the far end of t1 as well: real diagonals on both sides of the 16-bit wrap
    seqs.append(rnd(200) + mutate(rng, t0[36500:37000], 0.1))
    seqs.append(mutate(rng, t1[33000:33900], 0.2))
    for _ in range(10):
        seqs.append(rnd(int(rng.integers(150, 900))))
    return seqs


def gap_pairs(seed=5):
    """alignments that span one long gap: a gap of g residues opens a band of > g columns"""
    rng = np.random.default_rng(seed)
    rnd = lambda n: ''.join(rng.choice(list(AA), n))
    out = []
    for g, side in ((1100, 'q'), (1500, 't'), (2600, 'q')):
        a, b, ins = rnd(700), rnd(800), rnd(g)
        q = rnd(30) + a + (ins if side == 'q' else '') + b + rnd(20)
        t = rnd(11) + mutate(rng, a, 0.05) + (ins if side == 't' else '') + mutate(rng, b, 0.05) + rnd(40)
        out.append((q, t))
    return out


def saturating_pairs(seed=11):
    """alignments whose score leaves int16: the reference's word kernel saturates at 32 767 (simdi16_adds), so score and
    end position are those of the first cell that reaches it"""
    rng = np.random.default_rng(seed)
    rnd = lambda n: ''.join(rng.choice(list(AA), n))
    a = rnd(9000)
    b = rnd(40) + mutate(rng, a, 0.04) + rnd(25)
    c = rnd(7000)
    return [(a, b), (b, a), (c, mutate(rng, c, 0.02)[300:])]


def main():
    ref, orc = Ref(6), Oracle(4)
    seqs = crafted()
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = ''.join(seqs).encode()
    rix = ref.index(blob, off)
    rpf = rix.prefilter(int(lens.max()), max_hits=300)
    rows = []
    for q in range(len(seqs)):
        rid, rsc, rdg, _ = rpf.query(seqs[q], q)
        rows += [(q, int(t), int(s), int(d)) for t, s, d in zip(rid, rsc, rdg)]
    rows = np.array(rows, np.int64)
    print('reference prefilter rows', rows.shape)
    # oracle against it
    nums = [orc.map_sequence(s) for s in seqs]
    tgt = orc.target(np.concatenate(nums), off)
    bad = 0
    for q in range(len(seqs)):
        ids, sc, dg, st = tgt.prefilter(nums[q], identity_id=q, max_hits=300)
        exp = rows[rows[:, 0] == q]
        ok = len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all() and (dg == (exp[:, 3] & 0xFFFF)).all()
        if not ok:
            bad += 1
            print('query', q, 'len', lens[q], 'oracle', list(zip(ids, sc, dg))[:6], 'reference', exp[:6, 1:].tolist())
    print('oracle == reference for', len(seqs) - bad, 'of', len(seqs), 'queries')
    # long-gap alignments
    pairs = gap_pairs()
    sw = RefSW(ref, max(max(len(q), len(t)) for q, t in pairs), 10 ** 7)
    aln = []
    for q, t in pairs:
        sw.set_query(q)
        r = sw.align(t, sw_mode=2, eval_thr=10.0, cov_mode=2, cov_thr=0.0)
        print('gap pair', len(q), len(t), {k: v for k, v in r.items() if k != 'backtrace'}, 'band >=', abs((r['qEnd'] - r['qStart']) - (r['tEnd'] - r['tStart'])) + 1)
        aln.append(r)
    sat = saturating_pairs()
    sw2 = RefSW(ref, max(max(len(q), len(t)) for q, t in sat), 10 ** 7)
    sat_res = []
    for q, t in sat:
        sw2.set_query(q)
        r = sw2.align(t, sw_mode=2, eval_thr=10.0, cov_mode=2, cov_thr=0.0)
        print('saturating pair', len(q), len(t), {k: v for k, v in r.items() if k != 'backtrace'})
        o = orc.sw_align(orc.map_sequence(q), orc.map_sequence(t), 10 ** 7, cov_thr=0.0)
        same = all(o[k] == r[k] for k in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'backtrace', 'evalue'))
        print('   oracle == reference:', same)
        sat_res.append(r)
    np.savez_compressed(os.path.join(GOLD, 'long_vectors.npz'),
                        sat_q=np.array([q for q, _ in sat]), sat_t=np.array([t for _, t in sat]),
                        sat_res=np.array([[r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['identical'], r['btLen']] for r in sat_res], np.int64),
                        sat_bt=np.array([r['backtrace'] for r in sat_res]), sat_eval=np.array([r['evalue'] for r in sat_res]), blob=np.frombuffer(blob, np.uint8), off=off, pf_rows=rows,
                        gap_q=np.array([q for q, _ in pairs]), gap_t=np.array([t for _, t in pairs]),
                        gap_res=np.array([[r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['identical'], r['btLen']] for r in aln], np.int64),
                        gap_bt=np.array([r['backtrace'] for r in aln]), gap_eval=np.array([r['evalue'] for r in aln]))


if __name__ == '__main__':
    main()
