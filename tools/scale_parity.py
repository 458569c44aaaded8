#!/usr/bin/env python3
"""Parity at scale, on the GPU box (needs oracle/_ref/libsdref.so, which travels with the snapshot): device prefilter
and alignments against the REAL reference classes on the P-proteome target (default 1000: 3e6 sequences, BASELINE
configs[2]'s size) -- BINSIZE as chosen from the DB size, coarse split, oversize buckets, the hit-buffer overflow of
the longest queries, --max-seqs 2P.

  python tools/scale_parity.py [P [k [n_longest [max_hits]]]]   prints the counts (profiles/r02_scale_parity_p1000.log is such a run;
                                                     P = 3730 crosses the reference's k = 7 threshold of 3.35e9 residues)
  tests/test_gpu_scale.py                   asserts run(1000) has no mismatch (-m gpu)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(P=1000, n_longest=12, n_random=120, aln_queries=30, aln_hits=40, log=print, gpu=None, host=None, k=None, sensitivity=5.7,
        max_hits=None, device_index=False):
    """returns dict(prefilter_queries, prefilter_rows, prefilter_mismatch, alignments, alignment_mismatch, bin_size,
    max_index_hits, k, not_computed).  k: None = the reference's rule (7 from 3.35e9 target residues on,
    M/src/prefiltering/Prefiltering.cpp setKmerSize / IndexTable.h:439-449), else forced."""
    from spacedust_amd import api
    from spacedust_amd.synth import make_proteomes, ALPHABET
    from oracle.pyoracle import Ref, RefSW
    t0 = time.time()
    ps = make_proteomes(P, genes_per_proteome=3000, seed=0x5ED0 + 2)
    log('generated', ps.n, round(time.time() - t0, 1))
    host = host or api.Host()
    gpu = gpu or api.Context(0)
    lens = ps.lengths()
    rng = np.random.default_rng(9)
    order = np.argsort(-lens)
    # the longest (hit-buffer overflow) + random ones
    queries = np.concatenate([order[:n_longest], rng.choice(ps.n, n_random, replace=False)]).astype(np.int64)
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum(lens[queries])
    qres = np.concatenate([ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])] for q in queries])
    if k is None:
        k = 7 if int(ps.offsets[-1]) >= 3350000000 else 6
    kmer_thr = host.kmer_threshold(sensitivity, k)
    log('k', k, 'k-mer threshold', kmer_thr, 'target residues', int(ps.offsets[-1]))
    sw_b, dg_b, km_b = host.comp_bias(qres, qoff, k=k)
    t0 = time.time()
    if device_index:   # the index the searches use: built on the GPU (sd_target_build) -- the host builder needs ten minutes at 10 000 proteomes
        tgt = api.Target.build_on_device(gpu, host, ps.residues, ps.offsets, k=k, kmer_thr=kmer_thr)
        n_entries = int(tgt.build_stats['entries'])
    else:
        idx = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=kmer_thr)
        tgt = api.Target(gpu, host, idx)
        n_entries = int(idx.n_entries)
    log('index', 'device' if device_index else 'host', round(time.time() - t0, 1), 'entries', n_entries)
    max_hits = max_hits or max(300, 2 * P)   # default: --max-seqs 2P, every target set can be reached
    par = api.prefilter_params(host, ps.n, kmer_thr=kmer_thr, max_hits=max_hits, cov_thr=0.0, bin_size=None, k=k)
    log('bin size', par.binSize)
    hits, cnt, st = api.prefilter(gpu, tgt, par, qres, qoff, km_b, dg_b, queries.astype(np.uint32), want_stats=True)
    not_computed = cnt == 0xFFFFFFFF   # per-query error slots (double overflow of the reference's hit buffer / >= 2^24 index hits)
    log('device prefilter done: hits', int(cnt[~not_computed].sum()), 'max index hits/query', int(st[:, 1].max()), 'not computed',
        int(not_computed.sum()))
    lut = np.frombuffer(ALPHABET.encode(), np.uint8)
    blob = lut[ps.residues].tobytes()
    ref = Ref(k)
    t0 = time.time()
    rix = ref.index(blob, ps.offsets, kmer_thr=kmer_thr, threads=16)
    log('ref index', round(time.time() - t0, 1))
    rpf = rix.prefilter(int(lens.max()) + 2, max_hits=max_hits)
    bad = 0
    t_ref_pf, n_ref_pf = 0.0, 0   # seconds inside the reference's QueryMatcher for the RANDOM queries (the longest ones are there for the overflow route)
    for x, q in enumerate(queries):
        if not_computed[x]:
            continue
        seq = blob[int(ps.offsets[q]):int(ps.offsets[q + 1])]
        t1 = time.time()
        ids, sc, dg, _ = rpf.query(seq, int(q))
        if x >= n_longest:
            t_ref_pf += time.time() - t1
            n_ref_pf += 1
        n = int(cnt[x])
        ok = n == len(ids) and (hits[x, :n]['seqId'] == ids).all() and (hits[x, :n]['score'] == sc).all() and \
            (hits[x, :n]['diagonal'] == dg).all()
        if not ok:
            bad += 1
            log('MISMATCH query', q, 'len', lens[q], 'device', n, 'ref', len(ids))
    cnt = np.where(not_computed, 0, cnt)
    log('prefilter queries compared', int((~not_computed).sum()), 'mismatching', bad, 'rows', int(cnt.sum()))
    # alignments of the first hits of some of the random queries
    mat, _, _ = host.matrix(0)
    db = int(ps.offsets[-1])
    ts = gpu.seqset(ps.residues, ps.offsets, None)
    qs = gpu.seqset(qres, qoff, sw_b)
    spar = gpu.sw_params(mat, db)
    pq, pt = [], []
    for x in range(n_longest, min(len(queries), n_longest + aln_queries)):
        for h in range(min(aln_hits, int(cnt[x]))):
            pq.append(x)
            pt.append(int(hits[x, h]['seqId']))
    pq = np.array(pq, np.uint32)
    pt = np.array(pt, np.uint32)
    ident = (queries[pq] == pt)
    out, pool = gpu.sw_align(spar, qs, ts, pq, pt, identity=ident)
    sw = RefSW(ref, int(lens.max()) + 2, db)
    badsw = 0
    last = -1
    t_ref_sw, cells_ref_sw = 0.0, 0
    for i in range(len(pq)):
        if pq[i] != last:
            q = queries[pq[i]]
            sw.set_query(blob[int(ps.offsets[q]):int(ps.offsets[q + 1])])
            last = pq[i]
        t = pt[i]
        t1 = time.time()
        r = sw.align(blob[int(ps.offsets[t]):int(ps.offsets[t + 1])], identity=bool(ident[i]))
        t_ref_sw += time.time() - t1
        cells_ref_sw += int(lens[queries[pq[i]]]) * int(lens[t])
        o = out[i]
        same = int(o['score']) == r['score'] and int(o['qEnd']) == r['qEnd'] and int(o['tEnd']) == r['tEnd'] and \
            int(o['btLen']) == r['btLen']
        if same and r['btLen'] > 0:
            bt = pool[int(o['btOffset']):int(o['btOffset']) + int(o['btLen'])].tobytes().decode()
            same = bt == r['backtrace'] and int(o['qStart']) == r['qStart'] and int(o['tStart']) == r['tStart'] and \
                float(o['evalue']) == r['evalue']
        if not same:
            badsw += 1
    log('alignments compared', len(pq), 'mismatching', badsw)
    # the reference's cost of one query at this size, one thread: its prefilter call, plus Smith-Waterman (score, ends, start, traceback
    # as Matcher::getSWResult runs them) for the query's result list at the measured seconds per alignment of the sampled pairs
    rows_per_query = float(cnt[n_longest:].mean()) if len(cnt) > n_longest else 0.0
    ref_cpu = None
    if n_ref_pf and len(pq):
        s_pf, s_aln = t_ref_pf / n_ref_pf, t_ref_sw / len(pq)
        s_query = s_pf + rows_per_query * s_aln
        ref_cpu = dict(kind='reference', cores=1, prefilter_s_per_query=s_pf, sw_s_per_alignment=s_aln, sw_gcups_one_core=cells_ref_sw / t_ref_sw / 1e9,
                       rows_per_query=rows_per_query, s_per_query=s_query, genome_pairs_per_s_per_core=1.0 / (s_query * ps.n / float(P * P)),
                       sample='%d random queries through QueryMatcher (%.3f s each) and %d of their alignments through Matcher::getSWResult '
                              '(%.4f s each) x %.0f result rows per query, one thread; clusterhits and the glue modules not included'
                              % (n_ref_pf, s_pf, len(pq), s_aln, rows_per_query))
        log('reference CPU at this size:', ref_cpu['sample'], '->', round(ref_cpu['genome_pairs_per_s_per_core'], 3), 'genome-pairs/s per core')
    return dict(reference_cpu=ref_cpu, proteomes=P, targets=int(ps.n), k=k, kmer_thr=int(kmer_thr), target_residues=int(ps.offsets[-1]), index_entries=n_entries, max_hits=int(max_hits), index='device' if device_index else 'host',
                not_computed=int(not_computed.sum()), prefilter_queries=int((~not_computed).sum()), prefilter_rows=int(cnt.sum()),
                prefilter_mismatch=bad, alignments=len(pq), alignment_mismatch=badsw, bin_size=int(par.binSize),
                max_index_hits=int(st[:, 1].max()))


if __name__ == '__main__':
    import json
    P_ = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    kw = {}
    if len(sys.argv) > 2:
        kw['k'] = int(sys.argv[2]) or None   # 0 = the reference's rule
    if len(sys.argv) > 3:
        kw['n_longest'] = int(sys.argv[3])
    if len(sys.argv) > 4:
        kw['max_hits'] = int(sys.argv[4])
    print(json.dumps(run(P_, log=lambda *a: print(*a, flush=True), **kw)))
