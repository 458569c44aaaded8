#!/usr/bin/env python3
"""bench.py -- clustersearch hot path (prefilter + SW align + clusterhits) on MI355X.

Workload (BASELINE.json configs[1]): P synthetic proteomes (default 100 x 3000 proteins, len ~300) searched
all-vs-all, --max-seqs max(300, 2P), --filter-self-match.  The target side (k-mer index, masked lookup,
sequences) is resident in HBM.  One *step* = clustersearch of one batch of B query proteomes (default 10) against
all P target proteomes = B*P genome pairs; K timed steps, W warm-up steps.  With N ranks every rank runs its own
K steps on different query batches against its own replica of the target (weak scaling, no data-path collective);
the only RCCL traffic is the final gather of the per-rank result summaries.

Prints ONE JSON line (rank 0).  `roofline` describes the kernel with the largest accumulated time (HIP events
recorded by libsdgpu on its own stream); `cpu_baseline` times the reference's own AVX2 code
(oracle/_ref/libsdref.so, built from /root/reference by oracle/Makefile) -- or the oracle port if that library
did not travel -- on a bounded sample of the same workload on the host cores of this box.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--proteomes', type=int, default=100)
    ap.add_argument('--genes', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=10, help='query proteomes per step')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    return ap.parse_args()


def algorithmic_bytes(stats, q_len_sum):
    """B_pref = 16 S + 6 M + C (7+8) + sum(len_c) + 21 L + 10 H ; SW: per pair qLen+tLen+21 qLen+24"""
    pref = 16 * stats['kmers'] + 6 * stats['index_hits'] + 15 * stats['diagonals'] + stats['diag_len'] + 21 * q_len_sum + \
        10 * stats['prefilter_hits']
    return pref


def cpu_baseline(ps, db, cs, args, n_threads, seconds):
    """bounded sample of the same workload on the host cores: reference QueryMatcher + SmithWaterman per query
    (kind "reference") when oracle/_ref/libsdref.so is present, else the oracle port; clusterhits by the oracle."""
    from oracle import pyoracle
    from spacedust_amd.synth import ALPHABET
    P = ps.n_sets
    lut = np.frombuffer(ALPHABET.encode(), np.uint8)
    kind = 'reference' if pyoracle.ref_available() else 'port'
    rng = np.random.default_rng(1)
    sample = rng.choice(ps.n, size=min(ps.n, 4096), replace=False)
    t_idx0 = time.time()
    if kind == 'reference':
        ref = pyoracle.Ref(6)
        blob = lut[ps.residues].tobytes()
        rix = ref.index(blob, ps.offsets, kmer_thr=cs.kmer_thr, threads=n_threads)
        max_len = int(ps.lengths().max())
    else:
        orc = pyoracle.Oracle(n_threads)
        ot = orc.target(ps.residues, ps.offsets, kmer_thr=cs.kmer_thr)
    t_index = time.time() - t_idx0
    done = [0] * n_threads
    pairs = [0] * n_threads
    cells = [0] * n_threads
    deadline = time.time() + seconds
    lens = ps.lengths()
    db_res = int(ps.offsets[-1])

    def worker(w):
        if kind == 'reference':
            pf = rix.prefilter(max_len, max_hits=cs.max_seqs)
            sw = pyoracle.RefSW(ref, max_len, db_res)
        for qi in sample[w::n_threads]:
            if time.time() > deadline:
                break
            a, b = int(ps.offsets[qi]), int(ps.offsets[qi + 1])
            if kind == 'reference':
                qs = blob[a:b]
                ids, sc, dg, _ = pf.query(qs, int(qi))
                sw.set_query(qs)
                for t in ids:
                    if float(lens[t]) / float(lens[qi]) < 0.8:
                        continue
                    sw.align(blob[int(ps.offsets[t]):int(ps.offsets[t + 1])], identity=(t == qi))
                    pairs[w] += 1
                    cells[w] += int(lens[qi]) * int(lens[t])
            else:
                ids, sc, dg, _ = ot.prefilter(ps.residues[a:b], identity_id=int(qi), max_hits=cs.max_seqs,
                                              bin_size=int(cs.pf_par.binSize), kmer_thr=cs.kmer_thr)
                for t in ids:
                    if float(lens[t]) / float(lens[qi]) < 0.8:
                        continue
                    orc.sw_align(ps.residues[a:b], ps.residues[int(ps.offsets[t]):int(ps.offsets[t + 1])], db_res,
                                 identity=bool(t == qi))
                    pairs[w] += 1
                    cells[w] += int(lens[qi]) * int(lens[t])
            done[w] += 1

    t0 = time.time()
    th = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    nq = sum(done)
    q_per_s = nq / dt if dt > 0 else 0.0
    # all-vs-all of P proteomes: P*genes queries for P*P genome pairs  ->  genes/P queries per genome pair
    queries_per_pair = ps.n / float(P * P)
    # clusterhits on the CPU: oracle restatement (the reference's clusterhits() is not linkable), one core per entry
    orc2 = pyoracle.Oracle(1)
    ch_t = 0.0
    ch_n = 0
    if cs.last_entries is not None:
        eo, qp, tp, sd, pv, nq_arr = cs.last_entries
        t1 = time.time()
        for e in range(min(len(eo) - 1, 8)):
            x0, x1 = int(eo[e]), int(eo[e + 1])
            pyoracle.oracle_clusterhits(orc2, qp[x0:x1], tp[x0:x1], sd[x0:x1], pv[x0:x1], int(nq_arr[e]))
            ch_n += 1
        ch_t = time.time() - t1
    ch_per_pair_core = (ch_t / ch_n) if ch_n else 0.0
    sec_per_pair = queries_per_pair / q_per_s + ch_per_pair_core / n_threads if q_per_s > 0 else float('inf')
    return dict(value=1.0 / sec_per_pair if sec_per_pair > 0 else 0.0, unit='genome-pairs/s', cores=n_threads, kind=kind,
                sample='%d query proteins (prefilter + SW vs the full %d-proteome target, %d threads, %.1f s) + %d clusterhits entries by the oracle; index build %.1f s not included'
                       % (nq, P, n_threads, dt, ch_n, t_index),
                queries_per_s=q_per_s, sw_gcups=sum(cells) / dt / 1e9 if dt > 0 else 0.0, sw_pairs=sum(pairs),
                clusterhits_s_per_entry_core=ch_per_pair_core)


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    from spacedust_amd.api import Host, Context
    from spacedust_amd.pipeline import SetDB, ClusterSearch
    from spacedust_amd.synth import make_proteomes

    P, B = args.proteomes, args.batch
    n_threads = max(1, (os.cpu_count() or 1) // max(1, world))
    host = Host(n_threads)
    gpu = Context(local_rank if world > 1 else 0)
    t0 = time.time()
    ps = make_proteomes(P, genes_per_proteome=args.genes, seed=0x5ED0 + 2)
    t_gen = time.time() - t0
    db = SetDB.from_proteomes(ps)
    max_seqs = max(300, 2 * P)
    cs = ClusterSearch(gpu, host, db, max_seqs=max_seqs, filter_self_match=True)
    cs.last_entries = None
    n_batches = (P + B - 1) // B
    set_start = ps.set_start

    def run_step(step_idx, keep=False):
        b = (rank * (args.steps + args.warmup) + step_idx) % n_batches
        s0, s1 = b * B, min(P, (b + 1) * B)
        out = cs.search(db, same_db=True, query_range=(int(set_start[s0]), int(set_start[s1])), chunk_queries=30000)
        if keep and out['cluster_out'] is not None:
            hq, ht = out['hit_q'], out['hit_t']
            cs.last_entries = (out['entry_off'], db.pos_in_set[hq], db.pos_in_set[ht],
                               (db.strand[hq] | (db.strand[ht] << 1)).astype(np.uint8),
                               np.zeros(len(hq)) + 1e-30, db.set_size[out['entry_q']])
        return (s1 - s0) * P, out

    for w in range(args.warmup):
        run_step(w)
    gpu.profile(True)
    for k in cs.stats:
        cs.stats[k] = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gpu.synchronize()
    t0 = time.time()
    pairs_done = 0
    q_len_sum = 0
    summary = np.zeros(4, np.int64)
    stage = {}
    for k in range(args.steps):
        n, out = run_step(args.warmup + k, keep=(k == args.steps - 1))
        pairs_done += n
        summary += np.array([out['entries'], out['matched_hits'], out['clusters'], out['cluster_hits']], np.int64)
        for s, v in out['timing'].items():
            stage[s] = stage.get(s, 0.0) + v
    gathered = None
    if dist is not None:
        # final result gather over RCCL (xGMI): per-rank result summary
        tsum = torch.from_numpy(summary).cuda()
        gathered = [torch.zeros_like(tsum) for _ in range(world)]
        dist.all_gather(gathered, tsum)
    gpu.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.time() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64).cuda()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_max = float(tmax.item())
        tot = torch.tensor([pairs_done], dtype=torch.float64).cuda()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        pairs_total = float(tot.item())
    else:
        dt_max, pairs_total = dt, float(pairs_done)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    prof = gpu.profile_report()
    kernels = {k: dict(ms=v[0], launches=int(v[1])) for k, v in prof.items()}
    st = cs.stats
    qlen_steps = 0
    b_pref = algorithmic_bytes(st, int(ps.lengths().mean() * args.steps * B * args.genes))
    pf_ms = sum(v['ms'] for k, v in kernels.items() if k.startswith('prefilter_'))
    sw_ms = kernels.get('sw_score', dict(ms=0))['ms']
    cells_sw = st['cells_fwd'] + st['cells_rev']
    b_sw = st['pairs'] * (int(ps.lengths().mean()) * 23 + 24)
    dom = max(kernels.items(), key=lambda kv: kv[1]['ms'])[0] if kernels else 'none'
    if dom == 'sw_score':
        alg, per = b_sw, kernels[dom]
    elif dom.startswith('prefilter_'):
        share = kernels[dom]['ms'] / pf_ms if pf_ms > 0 else 1.0
        alg, per = (6 * st['index_hits'] if dom in ('prefilter_gather_hits', 'prefilter_sort_hits') else b_pref * share), kernels[dom]
    else:
        alg, per = 17 * int(summary[1]) + 4 * int(summary[1]), kernels.get(dom, dict(ms=1, launches=1))
    achieved = (alg / max(per['launches'], 1)) / ((per['ms'] / max(per['launches'], 1)) * 1e-3) / 1e9 if per['ms'] > 0 else 0.0
    roofline = dict(bound='hbm', kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                    traffic=None, launches=per['launches'], avg_launch_ms=per['ms'] / max(per['launches'], 1),
                    note=('sw_score is integer-VALU bound (DP state lives in VGPR/LDS): see sw_gcups; '
                          'algorithmic bytes = residue streams only') if dom == 'sw_score' else 'algorithmic bytes per SURVEY.md 8(d)')
    res = {
        'metric': 'clustersearch throughput (genome-pairs/s; SW GCUPS alongside)',
        'value': pairs_total / dt_max,
        'unit': 'genome-pairs/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt_max / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'int32',
        'data': 'synthetic',
        'config': {'workload': '%d synthetic proteomes x %d proteins (len~300) all-vs-all, clustersearch --search-mode 0 '
                               '--filter-self-match --max-seqs %d; step = %d query proteomes vs all %d targets'
                               % (P, args.genes, max_seqs, B, P),
                   'parallelism': 'query-set sharding x%d, target index replicated, RCCL final gather' % world},
        'roofline': roofline,
        'sw_gcups': cells_sw / sw_ms / 1e6 if sw_ms > 0 else 0.0,
        'sw_cells': {'forward': st['cells_fwd'], 'reverse': st['cells_rev'], 'traceback': st['cells_tb']},
        'prefilter': {'queries': args.steps * B * args.genes, 'kernel_ms': pf_ms, 'algorithmic_bytes': b_pref,
                      'achieved_GBs': b_pref / pf_ms / 1e6 if pf_ms > 0 else 0.0, 'index_hits': st['index_hits'],
                      'kmers': st['kmers'], 'hits': st['prefilter_hits']},
        'kernels': kernels,
        'stage_wall_s': stage,
        'results': {'entries': int(summary[0]), 'matched_hits': int(summary[1]), 'clusters': int(summary[2]),
                    'cluster_hits': int(summary[3])},
        'setup_s': {'generate': t_gen, 'index_build_host': cs.timing['index_build_s'], 'upload': cs.timing['upload_s']},
        'device': gpu.device_name(),
        'host_cores': os.cpu_count(),
    }
    if not args.no_cpu:
        try:
            res['cpu_baseline'] = cpu_baseline(ps, db, cs, args, os.cpu_count() or 1, args.cpu_seconds)
        except Exception as e:   # the baseline leg must never take the measurement down
            res['cpu_baseline'] = dict(value=None, unit='genome-pairs/s', cores=0, kind='failed', sample=repr(e))
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
