#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the REAL reference code
(oracle/_ref/libsdref.so = the reference's classes compiled from /root/reference by oracle/Makefile) on the
reference's own regression input (examples/*.faa), plus the oracle pipeline outputs whose md5s are pinned
to the reference binary's outputs.  Run in the dev container only:  python tools/make_golden.py"""
import hashlib
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Oracle, Ref, RefSW, read_fasta, oracle_clusterhits  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
EX = '/root/reference/examples/'


def main():
    ref, orc = Ref(6), Oracle(8)
    out = {}
    for w, name in ((0, 'blosum62_2'), (1, 'vtml80_8_m02'), (2, 'blosum62_2_m02')):
        m, pb, a2n = ref.matrix(w)
        out['mat_' + name] = m
        out['pback_' + name] = pb
        out['aa2num'] = a2n[:255]
    n1, s1 = read_fasta(EX + 'NC_000913.faa')
    n2, s2 = read_fasta(EX + 'NC_000915.faa')
    seqs = s1 + s2
    nums = [orc.map_sequence(s) for s in seqs]
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = ''.join(seqs).encode()
    rng = np.random.default_rng(2024)
    # --- per-function vectors from the reference
    sample = sorted(rng.choice(len(seqs), 24, replace=False).tolist())
    out['cb_sample'] = np.array(sample)
    for w in (0, 1):
        out['cb_%d' % w] = np.concatenate([ref.compbias(w, nums[i]) for i in sample])
    masked = [ref.mask(nums[i])[0] for i in range(len(seqs))]
    mpos = np.concatenate([np.nonzero(masked[i] != nums[i])[0] + int(off[i]) for i in range(len(seqs))])
    out['masked_positions'] = mpos.astype(np.uint32)
    wins = rng.integers(0, 20, size=(40, 6)).astype(np.uint8)
    thrs = rng.integers(95, 125, size=40)
    out['kmer_windows'] = wins
    out['kmer_thr'] = thrs
    kl = [ref.kmer_list(wins[i], int(thrs[i])) for i in range(40)]
    out['kmer_list_off'] = np.cumsum([0] + [len(k) for k in kl])
    out['kmer_lists'] = np.concatenate(kl).astype(np.uint32)
    for w in (2, 3):
        sc, ix = ref.ext_matrix(w)
        out['ext%d_md5' % w] = np.frombuffer(hashlib.md5(sc.astype(np.int16).tobytes() + ix.astype(np.uint16).tobytes()).digest(), np.uint8)
    rix = ref.index(blob, off)
    ro, rs, rp, rl = rix.dump()
    out['index_stats'] = np.array([rix.n_entries, rix.masked_residues], np.uint64)
    out['index_md5'] = np.frombuffer(hashlib.md5(ro.astype(np.uint32).tobytes() + rs.tobytes() + rp.tobytes()).digest(), np.uint8)
    # --- prefilter + alignment of a query subset, by the reference
    qsub = list(range(4319, 4319 + 120)) + sorted(rng.choice(4319, 60, replace=False).tolist())
    rpf = rix.prefilter(int(lens.max()))
    rsw = RefSW(ref, int(lens.max()), int(lens.sum()))
    pf_rows, sw_rows, bts = [], [], []
    for q in qsub:
        ids, sc, dg, _ = rpf.query(seqs[q], q)
        rsw.set_query(seqs[q])
        for t, s, d in zip(ids, sc, dg):
            pf_rows.append((q, int(t), int(s), int(d)))
            if float(np.float32(lens[t]) / np.float32(lens[q])) < 0.8:
                continue
            a = rsw.align(seqs[t], identity=(t == q))
            sw_rows.append((q, int(t), a['score'], a['qStart'], a['qEnd'], a['tStart'], a['tEnd'],
                            a['identical'] if a['btLen'] > 0 else 0, a['btLen']))
            bts.append(a['backtrace'])
            out.setdefault('sw_eval', []).append(a['evalue'])
    out['pf_rows'] = np.array(pf_rows, np.int64)
    out['sw_rows'] = np.array(sw_rows, np.int64)
    out['sw_eval'] = np.array(out['sw_eval'], np.float64)
    out['sw_bt'] = np.frombuffer('\n'.join(bts).encode(), np.uint8)
    out['evalue_samples'] = np.array([[s, l, rsw.evalue(s, l)] for s in (20, 35, 60, 100, 250, 900, 1423) for l in (50, 315, 1200)])
    out['bitscore_samples'] = np.array([[s, rsw.bitscore(s)] for s in (20, 35, 60, 100, 250, 900)])
    np.savez_compressed(os.path.join(GOLD, 'reference_vectors.npz'), **out)
    print('reference_vectors.npz:', {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    main()
