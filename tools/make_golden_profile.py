#!/usr/bin/env python3
"""Golden vectors for profile queries: Sequence::mapProfile + the per-position k-mer generator,
QueryMatcher::matchQuery and SmithWaterman::ssw_align with a DBTYPE_HMM_PROFILE query, all run through the REAL
reference classes (oracle/_ref/libsdref.so) on crafted profile entries (25 bytes per position, Sequence.h:458-471).
Dev container only:  python tools/make_golden_profile.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import Ref, RefSW  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
AA = 'ACDEFGHIKLMNPQRSTVWY'
SEED6 = [0, 1, 3, 5, 8, 9]


def make_profile(m, seq, rng, scale=3, noise=6, x_rate=0.02):
    """a plausible profile entry: scores = scale x the blosum row of the letter + noise (int8), query / consensus letters"""
    L = len(seq)
    rec = np.zeros((L, 25), np.uint8)
    for i, ch in enumerate(seq):
        a = AA.index(ch)
        row = m[a, :20].astype(np.int32) * scale + rng.integers(-noise, noise + 1, 20)
        rec[i, :20] = np.clip(row, -128, 127).astype(np.int8).view(np.uint8)
        rec[i, 20] = a if rng.random() > x_rate else 20
        rec[i, 21] = int(np.argmax(row))
        rec[i, 22] = 30
    return rec.tobytes()


def mutate(b, rng, rate=0.3):
    s = list(b)
    for p in np.nonzero(rng.random(len(s)) < rate)[0]:
        s[p] = AA[rng.integers(20)]
    if rng.random() < 0.5:
        p = int(rng.integers(5, len(s) - 5))
        s[p:p] = list(rng.choice(list(AA), int(rng.integers(1, 8))))
    if rng.random() < 0.3:
        p = int(rng.integers(5, len(s) - 15))
        del s[p:p + int(rng.integers(1, 8))]
    return ''.join(s)


def main():
    rng = np.random.default_rng(2207)
    ref = Ref(6)
    m, _, _ = ref.matrix(0)
    base = [''.join(rng.choice(list(AA), int(rng.integers(90, 420)))) for _ in range(36)]
    # sequence 4*b is the base itself (the profile's own sequence in a same-DB iterative search: identity id of the
    # prefilter, scoreIdentical pair of the aligner), 4*b+1..3 are mutated descendants
    seqs = [s for b in base for s in [b] + [mutate(b, rng) for _ in range(3)]]
    lens = np.array([len(s) for s in seqs])
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    blob = ''.join(seqs).encode()
    profiles = [make_profile(m, b, rng, scale=int(rng.integers(2, 5))) for b in base[:18]]
    poff = np.zeros(len(profiles) + 1, np.uint64)
    poff[1:] = np.cumsum([len(p) for p in profiles])
    out = dict(blob=np.frombuffer(blob, np.uint8), off=off, profiles=np.frombuffer(b''.join(profiles), np.uint8), poff=poff)
    # k-mer lists of a few windows
    wins, lists = [], []
    for pi, pos, thr in [(0, 3, 70), (1, 10, 85), (2, 0, 60), (5, 20, 99), (7, 40, 75), (9, 11, 90)]:
        kl = ref.profile_kmer_list(profiles[pi], pos, thr)
        wins.append((pi, pos, thr))
        lists.append(kl.astype(np.uint32))
        print('window', pi, pos, thr, len(kl))
    out['windows'] = np.array(wins)
    out['list_off'] = np.cumsum([0] + [len(x) for x in lists]).astype(np.uint64)
    out['lists'] = np.concatenate(lists)
    # prefilter rows (target index with threshold 0, Prefiltering.cpp:525-527)
    rix = ref.index(blob, off, kmer_thr=0)
    for thr in (99, 80):
        rpf = rix.prefilter_profile(int(max(lens.max(), 450)) + 10, thr, max_hits=300)
        rows = []
        for qi, pd in enumerate(profiles):
            ids, sc, dg, _ = rpf.query(pd, identity_id=4 * qi if qi % 2 == 0 else 0xFFFFFFFF)   # every other query is "in the DB"
            rows += [(qi, int(t), int(s), int(d)) for t, s, d in zip(ids, sc, dg)]
        out['pf_rows_%d' % thr] = np.array(rows, np.int64)
        print('thr', thr, 'rows', len(rows))
    # alignments: every profile against the four descendants of its base + two unrelated targets
    sw = RefSW(ref, 2000, int(off[-1]))
    pairs, res, bts = [], [], []
    for qi, pd in enumerate(profiles):
        sw.set_query_profile(pd)
        for t, ident in [(4 * qi, qi % 2 == 0)] + [(x, False) for x in range(4 * qi + 1, 4 * qi + 4)] + \
                [(int(rng.integers(len(seqs))), False) for _ in range(2)]:
            r = sw.align(seqs[t], identity=ident)
            has_bt = r['btLen'] > 0
            pairs.append((qi, t, int(ident)))
            res.append((r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['identical'] if has_bt else 0, r['btLen']))
            bts.append(r['backtrace'])
            out.setdefault('evalue', []).append(r['evalue'])
    out['evalue'] = np.array(out['evalue'])
    out['sw_pairs'] = np.array(pairs)
    out['sw_res'] = np.array(res, np.int64)
    out['sw_bt'] = np.frombuffer('\n'.join(bts).encode(), np.uint8)
    print('alignments', len(pairs), 'with backtrace', int((out['sw_res'][:, 6] > 0).sum()))
    np.savez_compressed(os.path.join(GOLD, 'profile_vectors.npz'), **out)


if __name__ == '__main__':
    main()
