#!/usr/bin/env python3
"""Golden vectors for a stress DB (tests/stress_db.py: exact duplicates = ties at every cut, low-complexity runs and tandem
repeats that tantan masks, X-rich, one-residue to 3 000-residue sequences): prefilter rows at two list lengths and
alignments, all from the REAL reference classes (oracle/_ref/libsdref.so).  Dev container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle.pyoracle import Ref, RefSW  # noqa: E402
from stress_db import stress_db  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def main():
    seqs = stress_db()
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    blob = ''.join(seqs).encode()
    ref = Ref(6)
    rix = ref.index(blob, off)
    out = dict(n_entries=np.array([rix.n_entries]), masked=np.array([rix.masked_residues]))
    for mh in (300, 7):
        rpf = rix.prefilter(3100, max_hits=mh)
        rows = []
        for q in range(len(seqs)):
            ids, sc, dg, _ = rpf.query(seqs[q], q)
            rows += [(q, int(t), int(s), int(d)) for t, s, d in zip(ids, sc, dg)]
        out['pf_rows_%d' % mh] = np.array(rows, np.int64)
        print('max_hits', mh, 'rows', len(rows))
    sw = RefSW(ref, 3100, int(off[-1]))
    rows300 = out['pf_rows_300']
    pairs, res, evs, bts = [], [], [], []
    for q in range(len(seqs)):
        hits = rows300[rows300[:, 0] == q][:8]
        if len(hits) == 0:
            continue
        sw.set_query(seqs[q])
        for h in hits:
            t = int(h[1])
            r = sw.align(seqs[t], identity=(t == q))
            has_bt = r['btLen'] > 0
            pairs.append((q, t))
            res.append((r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['identical'] if has_bt else 0, r['btLen']))
            evs.append(r['evalue'])
            bts.append(r['backtrace'])
    out['sw_pairs'] = np.array(pairs)
    out['sw_res'] = np.array(res, np.int64)
    out['sw_ev'] = np.array(evs)
    out['sw_bt'] = np.frombuffer('\n'.join(bts).encode(), np.uint8)
    print('alignments', len(pairs), 'with backtrace', int((out['sw_res'][:, 6] > 0).sum()))
    np.savez_compressed(os.path.join(GOLD, 'stress_vectors.npz'), **out)


if __name__ == '__main__':
    main()
