#!/usr/bin/env python3
"""Cheap reproductions of what only a 10 000-proteome target has, each against the real reference (oracle/_ref/libsdref.so):
  A  more than 2^24 target sequences (25 target bits in the hit keys)
  B  index list starts beyond 2^32 (with SD_INDEX_TEST_SHIFT=4294967296: the entries sit 2^32 slots into their buffer)
  C  more saturated homologs than the result list holds (rescoring path with a 4 000-hit list)
  D  entry arrays of more than 2^32 elements (2^32 + 5e7 padding slots in front)
  E  queries with >= 2^24 index hits (wide stream positions), one of them past a double overflow of the reference's hit buffer
  a trailing 7 (B7, C7, D7) selects k = 7.      python tools/scale_cases.py A|B|C|D|E[7]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(which, log=print):
    from spacedust_amd import api
    from spacedust_amd.synth import ALPHABET
    from oracle.pyoracle import Ref
    rng = np.random.default_rng(3)
    host, gpu = api.Host(), api.Context(0)
    copies, max_hits, nq = 300, 1000, 24
    if which == 'A':
        n_seq, L = 17500000, 48
    elif which.startswith('E'):   # queries with >= 2^24 index hits (wide stream positions) beside ordinary ones, one overflow of the hit buffer
        n_seq, L, copies, max_hits, nq = 5200000, 48, 300, 1000, 12
    elif which.startswith('C'):   # more saturated homologs than the result list holds: the rescoring path with a long list
        n_seq, L, copies, max_hits, nq = 400000, 300, 6000, 4000, 6
    else:
        n_seq, L = 400000, 120
    res = rng.integers(0, 20, n_seq * L).astype(np.uint8)
    off = (np.arange(n_seq + 1, dtype=np.uint64) * L)
    # families: query i has 300 mutated copies spread over the DB
    queries = rng.choice(n_seq, nq, replace=False)
    for qi, q in enumerate(queries):
        src = res[q * L:(q + 1) * L]
        ncop = 720000 if (which.startswith('E') and qi in (3, 7)) else copies
        for t in rng.choice(n_seq, ncop, replace=False):
            c = src.copy()
            m = rng.random(L) < 0.15
            c[m] = rng.integers(0, 20, int(m.sum()))
            res[t * L:(t + 1) * L] = c
        res[q * L:(q + 1) * L] = src
    k, thr = (7, 122) if which.endswith('7') else (6, 112)
    if which.startswith('B') or which.startswith('D'):
        os.environ['SD_INDEX_WIDE'] = '1'
    t0 = time.time()
    idx = host.build_index(res, off, k=k, kmer_thr=thr)
    log('index', round(time.time() - t0, 1), 'entries', idx.n_entries, 'wide', idx.block_base is not None)
    if which.startswith('D'):   # entry arrays of more than 2^32 elements: the real entries sit behind 2^32 + 5e7 padding slots
        pad = (1 << 32) + 50000000
        idx = api.IndexArrays(k, thr, off, idx.kmer_offsets.copy(), np.concatenate([np.zeros(pad, np.uint32), idx.entry_seq]),
                              np.concatenate([np.zeros(pad, np.uint16), idx.entry_pos]), idx.masked.copy(), 0,
                              block_base=idx.block_base + np.uint64(pad))
        log('padded entries', idx.n_entries)
    tgt = api.Target(gpu, host, idx)
    qoff = np.arange(len(queries) + 1, dtype=np.uint64) * L
    qres = np.concatenate([res[q * L:(q + 1) * L] for q in queries])
    sw_b, dg_b, km_b = host.comp_bias(qres, qoff, k=k)
    par = api.prefilter_params(host, idx.n, kmer_thr=thr, max_hits=max_hits, cov_thr=0.0, k=k)
    hits, cnt, st = api.prefilter(gpu, tgt, par, qres, qoff, km_b, dg_b, queries.astype(np.uint32), want_stats=True)
    log('device rows', int(cnt[cnt != 0xFFFFFFFF].sum()), 'index hits per query', st[:, 1].tolist(), 'not computed', int((cnt == 0xFFFFFFFF).sum()))
    lut = np.frombuffer(ALPHABET.encode(), np.uint8)
    blob = lut[res].tobytes()
    ref = Ref(k)
    rix = ref.index(blob, off, kmer_thr=thr, threads=16)
    rpf = rix.prefilter(L + 2, max_hits=max_hits)
    bad, refused = [], []
    for x, q in enumerate(queries):
        if int(cnt[x]) == 0xFFFFFFFF:
            refused.append(x)
            continue
        ids, sc, dg, _ = rpf.query(blob[int(q) * L:(int(q) + 1) * L], int(q))
        n = int(cnt[x])
        ok = n == len(ids) and (hits[x, :n]['seqId'] == ids).all() and (hits[x, :n]['score'] == sc).all() and \
            (hits[x, :n]['diagonal'] == dg).all()
        if not ok:
            bad.append(x)
            log('MISMATCH', q, 'device', n, 'ref', len(ids))
    log('queries', len(queries), 'mismatching', len(bad), 'refused (per-query slot)', refused)
    return dict(index_hits=[int(v) for v in st[:, 1]], mismatching=bad, refused=refused, rows=int(cnt[cnt != 0xFFFFFFFF].sum()),
                max_db_matches=2 * max(1000000, n_seq))


if __name__ == '__main__':
    run(sys.argv[1], log=lambda *a: print(*a, flush=True))
