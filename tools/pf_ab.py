"""A/B of the prefilter's two front halves on one GPU: the k-mer-major join (default) against the per-k-mer lookup path
(SD_PF_JOIN=0) -- identical rows required, kernel times per stage printed.
usage: python tools/pf_ab.py [--proteomes 100] [--queries 8192] [--max-seqs 300]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--proteomes', type=int, default=100)
    ap.add_argument('--genes', type=int, default=3000)
    ap.add_argument('--queries', type=int, default=8192)
    ap.add_argument('--max-seqs', type=int, default=0)
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--modes', default='join,lookup')
    args = ap.parse_args()
    from spacedust_amd import api
    from spacedust_amd.api import Host, Context
    from spacedust_amd.synth import make_proteomes
    host, gpu = Host(), Context(0)
    t0 = time.time()
    ps = make_proteomes(args.proteomes, genes_per_proteome=args.genes, seed=0x5ED0 + 2)
    k = host.kmer_size(int(ps.offsets[-1])) if hasattr(host, 'kmer_size') else 6
    kmer_thr = host.kmer_threshold(5.7, k)
    idx = host.build_index(ps.residues, ps.offsets, k, kmer_thr)
    tgt = api.Target(gpu, host, idx)
    max_seqs = args.max_seqs or max(300, 2 * args.proteomes)
    par = api.prefilter_params(host, ps.n, kmer_thr=kmer_thr, max_hits=max_seqs, k=k)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets, k)
    nq = min(args.queries, ps.n)
    res = ps.residues[:int(ps.offsets[nq])]
    off = ps.offsets[:nq + 1]
    ident = np.arange(nq, dtype=np.uint32)
    print('setup %.1f s: %d targets, k=%d, %d queries, max_seqs %d, binSize %d' % (time.time() - t0, ps.n, k, nq, max_seqs, par.binSize), flush=True)
    out = {}
    for mode in args.modes.split(','):
        os.environ['SD_PF_JOIN'] = '1' if mode == 'join' else '0'
        api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident, want_stats=True)   # warm-up (workspace growth)
        gpu.profile(True)
        t1 = time.time()
        for _ in range(args.reps):
            r = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident, want_stats=True)
        wall = (time.time() - t1) / args.reps
        rep = gpu.profile_report()
        gpu.profile(False)
        out[mode] = r
        kern = {n: v for n, v in rep.items() if not n.startswith('host:')}
        tot = sum(v[0] for v in kern.values()) / args.reps
        print('== %s: wall %.1f ms (profiled, serialised), kernels %.1f ms, hits %d, k-mers %d' %
              (mode, wall * 1e3, tot, int(r[2][:, 1].sum()), int(r[2][:, 0].sum())))
        for n, v in sorted(kern.items(), key=lambda kv: -kv[1][0]):
            print('   %-34s %9.2f ms  %5d launches' % (n, v[0] / args.reps, v[1] // args.reps))
    modes = list(out)
    if len(modes) == 2:
        a, b = out[modes[0]], out[modes[1]]
        same = np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        bad = 0
        for q in range(nq):
            n = int(a[1][q])
            if int(b[1][q]) != n or not np.array_equal(a[0][q, :n], b[0][q, :n]):
                bad += 1
                if bad <= 8:
                    sa = set(map(tuple, a[0][q, :n].tolist()))
                    sb = set(map(tuple, b[0][q, :int(b[1][q])].tolist()))
                    print('  query %d (%d residues): counts %d / %d, stats %s / %s; only %s: %s; only %s: %s' %
                          (q, int(off[q + 1] - off[q]), n, int(b[1][q]), a[2][q].tolist(), b[2][q].tolist(), modes[0],
                           sorted(sa - sb)[:4], modes[1], sorted(sb - sa)[:4]))
        print('identical counts/stats: %s, queries with differing rows: %d of %d' % (same, bad, nq))
        sys.exit(0 if same and bad == 0 else 1)


if __name__ == '__main__':
    main()
