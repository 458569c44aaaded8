#!/usr/bin/env python3
"""Golden vectors for the prefilter's hit-buffer overflow path (QueryMatcher.cpp:281-316): a query whose index hits
exceed maxDbMatches = 2*max(1e6, #targets), so the reference closes the buffer once, runs findDuplicates on each
part and merges the two result lists.  Crafted DB run through the REAL reference classes (oracle/_ref/libsdref.so).
Dev container only:  python tools/make_golden_overflow.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
AA = 'ACDEFGHIKLMNPQRSTVWY'


def crafted(copies, seed=123):
    rng = np.random.default_rng(seed)
    base = ''.join(rng.choice(list(AA), 300))
    other = ''.join(rng.choice(list(AA), 260))
    seqs = [base, other]
    for _ in range(copies):
        s = list(base)
        for p in np.nonzero(rng.random(len(s)) < 0.10)[0]:
            s[p] = AA[rng.integers(20)]
        seqs.append(''.join(s))
    for _ in range(60):
        s = list(other)
        for p in np.nonzero(rng.random(len(s)) < 0.2)[0]:
            s[p] = AA[rng.integers(20)]
        seqs.append(''.join(s))
    return seqs


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1800
    ref, orc = Ref(6), Oracle(4)
    seqs = crafted(copies)
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = ''.join(seqs).encode()
    nums = [orc.map_sequence(s) for s in seqs]
    tgt = orc.target(np.concatenate(nums), off)
    try:
        ids, sc, dg, st = tgt.prefilter(nums[0], identity_id=0, max_hits=300)
        print('oracle: index hits of query 0:', int(st[1]), 'kmers', int(st[0]), 'hits returned', len(ids))
    except RuntimeError as e:
        print('oracle:', e)
        return
    rix = ref.index(blob, off)
    rpf = rix.prefilter(int(lens.max()), max_hits=300)
    rows = []
    for q in (0, 1, 5):
        rid, rsc, rdg, _ = rpf.query(seqs[q], q)
        rows += [(q, int(t), int(s), int(d)) for t, s, d in zip(rid, rsc, rdg)]
    rows = np.array(rows, np.int64)
    exp = rows[rows[:, 0] == 0]
    ok = len(ids) == len(exp) and (ids == exp[:, 1]).all() and (sc == exp[:, 2]).all() and (dg == (exp[:, 3] & 0xFFFF)).all()
    print('reference rows', rows.shape, 'oracle == reference for the overflowing query:', ok)
    np.savez_compressed(os.path.join(GOLD, 'overflow_vectors.npz'), blob=np.frombuffer(blob, np.uint8), off=off,
                        queries=np.array([0, 1, 5]), pf_rows=rows, index_hits_q0=np.array([int(st[1])]))


if __name__ == '__main__':
    main()
