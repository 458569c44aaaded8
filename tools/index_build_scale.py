#!/usr/bin/env python3
"""The target index of P synthetic proteomes built on the device (sd_target_build), timed, and checked on a sample against the
host builder (api.Target.sample_check).  On the GPU box:

  python tools/index_build_scale.py [P [k]]      P = 10000 is BASELINE configs[4]: 3 * 10^7 sequences, 9 * 10^9 residues, k = 7
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(P=1000, k=None, host_too=False, log=print):
    from spacedust_amd import api
    from spacedust_amd.synth import make_proteomes
    t0 = time.time()
    ps = make_proteomes(P, genes_per_proteome=3000, seed=0x5ED0 + 2)
    t_gen = time.time() - t0
    log('generated', ps.n, 'sequences', int(ps.offsets[-1]), 'residues in', round(t_gen, 1), 's')
    host = api.Host()
    gpu = api.Context(0)
    if k is None:
        k = host.auto_kmer_size(int(ps.offsets[-1]))
    thr = host.kmer_threshold(5.7, k)
    host.ext_matrix(3)
    gpu.profile(True)
    t0 = time.time()
    tgt = api.Target.build_on_device(gpu, host, ps.residues, ps.offsets, k=k, kmer_thr=thr)
    t_build = time.time() - t0
    prof = {n: v for n, v in gpu.profile_report().items() if n.startswith("index_")}
    log('device build', round(t_build, 2), 's', tgt.build_stats, prof)
    t0 = time.time()
    chk = tgt.sample_check(host, ps.residues, ps.offsets, thr)
    t_chk = time.time() - t0
    log('sample check', chk, round(t_chk, 1), 's')
    out = dict(proteomes=P, sequences=int(ps.n), residues=int(ps.offsets[-1]), k=k, kmer_thr=int(thr), generate_s=t_gen,
               device_build_s=t_build, kernels_ms=prof, build=tgt.build_stats, sample_check=chk, sample_check_s=t_chk)
    if host_too:
        t0 = time.time()
        h = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=thr)
        out['host_build_s'] = time.time() - t0
        out['host_threads'] = host.threads
        out['host_entries'] = int(h.n_entries)
        log('host build', round(out['host_build_s'], 1), 's', h.n_entries)
    return out


if __name__ == '__main__':
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != '-' else None
    print(json.dumps(run(P, k, host_too='--host' in sys.argv)))
