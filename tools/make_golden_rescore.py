#!/usr/bin/env python3
"""Golden vectors for the prefilter's rescoring path (QueryMatcher.cpp:157-170,525-544): more than maxHits targets
reach the saturated 8-bit diagonal score, so the cut threshold is 255 and the reference rescales true scores
against the query's self score.  A crafted DB (families of near-identical sequences) run through the REAL reference
classes (oracle/_ref/libsdref.so).  Dev container only:  python tools/make_golden_rescore.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
AA = 'ACDEFGHIKLMNPQRSTVWY'


def crafted(seed=77):
    rng = np.random.default_rng(seed)
    seqs = []
    bases = [''.join(rng.choice(list(AA), L)) for L in (280, 350, 190)]
    for b, copies, rate in zip(bases, (340, 330, 40), (0.04, 0.08, 0.05)):
        seqs.append(b)
        for _ in range(copies):
            s = list(b)
            for p in np.nonzero(rng.random(len(s)) < rate)[0]:
                s[p] = AA[rng.integers(20)]
            seqs.append(''.join(s))
    for _ in range(150):
        seqs.append(''.join(rng.choice(list(AA), int(rng.integers(120, 400)))))
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order], [int(np.nonzero(order == x)[0][0]) for x in (0, 341, 672)]


def main():
    ref, orc = Ref(6), Oracle(4)
    seqs, base_ids = crafted()
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = ''.join(seqs).encode()
    rix = ref.index(blob, off)
    rpf = rix.prefilter(int(lens.max()), max_hits=300)
    queries = base_ids + [5, 17, 400]
    rows = []
    for q in queries:
        ids, sc, dg, _ = rpf.query(seqs[q], q)
        rows += [(q, int(t), int(s), int(d)) for t, s, d in zip(ids, sc, dg)]
    rows = np.array(rows, np.int64)
    print('rows', rows.shape, 'max score', rows[:, 2].max(), 'scores>=255 (non-self):', int(((rows[:, 2] >= 255) & (rows[:, 2] < 65535)).sum()))
    np.savez_compressed(os.path.join(GOLD, 'rescore_vectors.npz'), blob=np.frombuffer(blob, np.uint8), off=off,
                        queries=np.array(queries), pf_rows=rows)


if __name__ == '__main__':
    main()
