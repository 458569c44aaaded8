"""The prefilter stage ALONE on one GPU at a bench size, per-kernel HIP-event times, for A/B runs of environment knobs in ONE
process (same box, same index, same queries): every variant's rows must equal the first variant's.
usage: python tools/pf_iso.py [--proteomes 1000] [--queries 8192] [--reps 1] VARIANT [VARIANT ...]
  VARIANT = name[:ENV=VALUE[,ENV=VALUE...]]     e.g.  base:SD_PF_HIT_BUDGET=1073741824  new
Writes one JSON line per variant (kernel ms by name, total, queries, algorithmic bytes per query) to stdout."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--proteomes', type=int, default=1000)
    ap.add_argument('--genes', type=int, default=3000)
    ap.add_argument('--queries', type=int, default=8192)
    ap.add_argument('--max-seqs', type=int, default=0)
    ap.add_argument('--reps', type=int, default=1)
    ap.add_argument('variants', nargs='+')
    args = ap.parse_args()
    from spacedust_amd import api
    from spacedust_amd.api import Host, Context
    from spacedust_amd.cpus import effective_cpus
    from spacedust_amd.synth import make_proteomes
    host, gpu = Host(effective_cpus()), Context(0)
    t0 = time.time()
    ps = make_proteomes(args.proteomes, genes_per_proteome=args.genes, seed=0x5ED0 + 2, workers=effective_cpus() if args.proteomes >= 128 else 1)
    k = host.auto_kmer_size(int(ps.offsets[-1]))
    kmer_thr = host.kmer_threshold(5.7, k)
    tgt = api.Target.build_on_device(gpu, host, ps.residues, ps.offsets, k=k, kmer_thr=kmer_thr)
    max_seqs = args.max_seqs or max(300, 2 * args.proteomes)
    par = api.prefilter_params(host, ps.n, kmer_thr=kmer_thr, max_hits=max_seqs, k=k)   # binSize: the library's rule by target count
    nq = min(args.queries, ps.n)
    off = ps.offsets[:nq + 1].astype(np.uint64)
    res = ps.residues[:int(off[-1])]
    sw_b, dg_b, km_b = host.comp_bias(res, off, k=k)
    ident = np.arange(nq, dtype=np.uint32)
    print('setup %.1f s: %d targets, k=%d, %d queries, max_seqs %d, binSize %d' % (time.time() - t0, ps.n, k, nq, max_seqs, par.binSize), file=sys.stderr, flush=True)
    first = None
    rc = 0
    touched = set()
    for var in args.variants:
        name, _, envs = var.partition(':')
        for e in touched:
            os.environ.pop(e, None)
        touched = set()
        for kv in filter(None, envs.split(',')):
            key, _, val = kv.partition('=')
            os.environ[key] = val
            touched.add(key)
        api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident)   # warm-up: workspaces of this shape
        gpu.profile(True)
        t1 = time.time()
        for _ in range(args.reps):
            r = api.prefilter(gpu, tgt, par, res, off, km_b, dg_b, ident, want_stats=True)
        wall = (time.time() - t1) / args.reps
        rep = gpu.profile_report()
        gpu.profile(False)
        kern = {n: round(v[0] / args.reps, 3) for n, v in rep.items() if n.startswith('prefilter_')}
        launches = {n: int(v[1] // args.reps) for n, v in rep.items() if n.startswith('prefilter_')}
        st = r[2]
        same = None
        if first is None:
            first = r
        else:
            same = bool(np.array_equal(first[1], r[1]) and np.array_equal(first[2], r[2]) and
                        all(np.array_equal(first[0][q, :int(r[1][q])], r[0][q, :int(r[1][q])]) for q in range(nq) if r[1][q] != 0xFFFFFFFF))
            if not same:
                rc = 1
        mem = gpu.device_memory()
        print(json.dumps(dict(variant=name, env=envs, queries=nq, kernel_ms=round(sum(kern.values()), 2), wall_ms=round(wall * 1e3, 1),
                              by_kernel=dict(sorted(kern.items(), key=lambda kv: -kv[1])), launches=launches,
                              index_hits_per_query=float(st[:, 1].mean()), kmers_per_query=float(st[:, 0].mean()),
                              not_computed=int((r[1] == 0xFFFFFFFF).sum()), equals_first=same, resident_GB=round((mem[1] - mem[0]) / 1e9, 1))), flush=True)
    sys.exit(rc)


if __name__ == '__main__':
    main()
