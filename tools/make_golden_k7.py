#!/usr/bin/env python3
"""Golden vectors for k = 7 (IndexTable.h:439-449: the automatic k-mer size from 3.35e9 target residues; spaced seed
11010110011, Sequence.h:24; k-mer generator split 2+2+3, KmerGenerator.cpp:41-86), produced by the REAL reference
classes (oracle/_ref/libsdref.so: KmerGenerator, IndexTable/IndexBuilder, QueryMatcher, UngappedAlignment) on a small
crafted DB.  Dev container only:  python tools/make_golden_k7.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Ref  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
AA = 'ACDEFGHIKLMNPQRSTVWY'


def crafted(seed=7):
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(50):
        b = ''.join(rng.choice(list(AA), int(rng.integers(120, 420))))
        for _ in range(4):
            s = list(b)
            for p in np.nonzero(rng.random(len(s)) < 0.3)[0]:
                s[p] = AA[rng.integers(20)]
            if rng.random() < 0.5:   # an indel, so that hits land on more than one diagonal
                p = int(rng.integers(10, len(s) - 10))
                s[p:p] = list(rng.choice(list(AA), int(rng.integers(1, 6))))
            seqs.append(''.join(s))
    seqs[3] = seqs[3][:60] + 'X' * 3 + seqs[3][63:]          # windows containing X are skipped
    seqs[10] = seqs[10][:40] + 'A' * 30 + seqs[10][70:]      # a low-complexity stretch for tantan
    return seqs


def main():
    ref = Ref(7)
    rng = np.random.default_rng(3)
    windows = rng.integers(0, 20, (12, 7)).astype(np.uint8)
    thrs = rng.integers(100, 131, 12)
    lists = [np.asarray(ref.kmer_list(w, int(t)), np.uint32) for w, t in zip(windows, thrs)]
    seqs = crafted()
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = ''.join(seqs).encode()
    out = dict(blob=np.frombuffer(blob, np.uint8), off=off, windows=windows, window_thr=thrs,
               list_off=np.cumsum([0] + [len(x) for x in lists]).astype(np.uint64), lists=np.concatenate(lists))
    queries = np.arange(0, len(seqs), 3)
    for thr in (122, 100):
        rix = ref.index(blob, off, kmer_thr=thr)
        rpf = rix.prefilter(int(lens.max()), max_hits=300)
        rows = []
        for q in queries:
            ids, sc, dg, _ = rpf.query(seqs[q], int(q))
            rows += [(int(q), int(t), int(s), int(d)) for t, s, d in zip(ids, sc, dg)]
        out['pf_rows_%d' % thr] = np.array(rows, np.int64)
        print('thr', thr, 'rows', len(rows))
    out['queries'] = queries
    np.savez_compressed(os.path.join(GOLD, 'k7_vectors.npz'), **out)


if __name__ == '__main__':
    main()
