#!/usr/bin/env python3
"""Golden vectors for the alignment modes and gates (score only / + start positions / + backtrace; coverage modes 0-2;
E-value thresholds): SmithWaterman::ssw_align of the REAL reference classes (oracle/_ref/libsdref.so) on pairs of a
small synthetic DB.  Dev container only:  python tools/make_golden_modes.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Ref, RefSW  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
AA = 'ACDEFGHIKLMNPQRSTVWY'
MODES = [(0, 2, 0.8, 10.0), (1, 2, 0.8, 10.0), (2, 0, 0.5, 10.0), (2, 1, 0.7, 1e-3), (1, 0, 0.9, 1e-5), (2, 2, 0.0, 1e-10)]


def main():
    rng = np.random.default_rng(515)
    base = [''.join(rng.choice(list(AA), int(rng.integers(60, 380)))) for _ in range(14)]
    seqs = []
    for b in base:
        seqs.append(b)
        for rate in (0.15, 0.4, 0.6):
            s = list(b)
            for p in np.nonzero(rng.random(len(s)) < rate)[0]:
                s[p] = AA[rng.integers(20)]
            if rng.random() < 0.6:
                p = int(rng.integers(3, len(s) - 3))
                s[p:p] = list(rng.choice(list(AA), int(rng.integers(1, 25))))
            if rng.random() < 0.3:
                s = s[int(rng.integers(0, len(s) // 3)):]
            seqs.append(''.join(s))
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    db = int(off[-1])
    pairs = [(4 * f + a, 4 * f + b) for f in range(len(base)) for a in range(4) for b in range(4) if a != b][::2]
    pairs += [(int(rng.integers(len(seqs))), int(rng.integers(len(seqs)))) for _ in range(20)]
    ref = Ref(6)
    sw = RefSW(ref, 600, db)
    out = dict(blob=np.frombuffer(''.join(seqs).encode(), np.uint8), off=off, pairs=np.array(pairs), modes=np.array(MODES))
    for mi, (sw_mode, cov_mode, cov_thr, eval_thr) in enumerate(MODES):
        rows, evs, bts = [], [], []
        for a, b in pairs:
            sw.set_query(seqs[a])
            r = sw.align(seqs[b], sw_mode=sw_mode, eval_thr=eval_thr, cov_mode=cov_mode, cov_thr=cov_thr)
            has_bt = r['btLen'] > 0
            rows.append((r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['identical'] if has_bt else 0, r['btLen']))
            evs.append(r['evalue'])
            bts.append(r['backtrace'])
        out['res_%d' % mi] = np.array(rows, np.int64)
        out['ev_%d' % mi] = np.array(evs)
        out['bt_%d' % mi] = np.frombuffer('\n'.join(bts).encode(), np.uint8)
        print(MODES[mi], 'starts', int((out['res_%d' % mi][:, 1] >= 0).sum()), 'backtraces', int((out['res_%d' % mi][:, 6] > 0).sum()))
    np.savez_compressed(os.path.join(GOLD, 'sw_modes_vectors.npz'), **out)


if __name__ == '__main__':
    main()
