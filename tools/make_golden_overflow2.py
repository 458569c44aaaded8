#!/usr/bin/env python3
"""Golden vectors for the prefilter's repeated hit-buffer overflow (QueryMatcher.cpp:281-316 with the branch :289-303): a
query whose index hits exceed maxDbMatches = 2*max(1e6, #targets) three times over, so the reference closes the buffer three
times -- after the second and the third time it merges the carried results with the new ones, scores them and keeps one
element per target -- and merges a fourth part at the end.  Crafted DB run through the REAL reference classes
(oracle/_ref/libsdref.so).  Dev container only:  python tools/make_golden_overflow2.py [copies]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from oracle.pyoracle import Ref  # noqa: E402
from make_golden_overflow import crafted, GOLD  # noqa: E402


def main():
    copies = int(sys.argv[1]) if len(sys.argv) > 1 else 26000
    ref = Ref(6)
    seqs = crafted(copies)
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = ''.join(seqs).encode()
    rix = ref.index(blob, off)
    rpf = rix.prefilter(int(lens.max()), max_hits=300)
    rows, stats = [], []
    for q in (0, 1, 5, 2 + copies // 2):
        rid, rsc, rdg, st = rpf.query(seqs[q], q)
        rows += [(q, int(t), int(s), int(d)) for t, s, d in zip(rid, rsc, rdg)]
        stats.append((q, float(st[0]), float(st[1])))
        print('query', q, 'rows', len(rid), 'stats', st)
    rows = np.array(rows, np.int64)
    np.savez_compressed(os.path.join(GOLD, 'overflow2_vectors.npz'), blob=np.frombuffer(blob, np.uint8), off=off,
                        queries=np.array([0, 1, 5, 2 + copies // 2]), pf_rows=rows, stats=np.array(stats))
    print('rows', rows.shape)


if __name__ == '__main__':
    main()
