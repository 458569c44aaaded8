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
import resource
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spacedust_amd.cpus import configure_openmp, effective_cpus  # noqa: E402
# one process per GPU shares the node's CPU quota: size every OpenMP team for this rank's share
configure_openmp(max(1, effective_cpus() // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--proteomes', type=int, default=100)
    ap.add_argument('--genes', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=10, help='query proteomes per step')
    ap.add_argument('--chunk', type=int, default=10000, help='queries per device chunk')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--cpu-threads', type=int, default=0)
    return ap.parse_args()


def algorithmic_bytes(stats, q_len_sum):
    """B_pref = 16 S + 6 M + C (7+8) + sum(len_c) + 21 L + 10 H ; SW: per pair qLen+tLen+21 qLen+24"""
    pref = 16 * stats['kmers'] + 6 * stats['index_hits'] + 15 * stats['diagonals'] + stats['diag_len'] + 21 * q_len_sum + \
        10 * stats['prefilter_hits']
    return pref


def cpu_baseline_subprocess(args, max_seqs, kmer_thr, bin_size, entries_path, seconds, n_threads):
    """The baseline leg runs in a child process (niced, hard timeout, a bounded number of threads) so that it can
    never take the measurement -- or the box -- down with it."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'cpu_baseline.py'), '--proteomes', str(args.proteomes), '--genes',
           str(args.genes), '--max-seqs', str(max_seqs), '--kmer-thr', str(kmer_thr), '--bin-size', str(bin_size),
           '--seconds', str(seconds), '--threads', str(n_threads), '--entries', entries_path]
    try:
        out = subprocess.run(['nice', '-n', '10'] + cmd, capture_output=True, text=True, timeout=seconds * 6 + 240)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if out.returncode != 0 or not line:
            return dict(value=None, unit='genome-pairs/s', cores=0, kind='failed', sample=(out.stderr or out.stdout)[-300:])
        return json.loads(line[-1])
    except Exception as e:
        return dict(value=None, unit='genome-pairs/s', cores=0, kind='failed', sample=repr(e))


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    dist = None
    # SD_BENCH_REHEARSAL=1: rehearse the N > 1 path on a box with one GPU -- every rank on cuda:0, gloo instead of RCCL
    rehearsal = os.environ.get('SD_BENCH_REHEARSAL') == '1'
    dev_index = 0 if rehearsal else local_rank
    to_dev = (lambda t: t) if rehearsal else (lambda t: t.cuda())
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    from spacedust_amd.api import Host, Context
    from spacedust_amd.pipeline import SetDB, ClusterSearch
    from spacedust_amd.synth import make_proteomes

    P, B = args.proteomes, args.batch
    # host stages: a bounded share of the cores (never saturate the box)
    n_threads = max(1, effective_cpus() // max(1, world))
    host = Host(n_threads)
    gpu = Context(dev_index if world > 1 else 0)
    t0 = time.time()
    ps = make_proteomes(P, genes_per_proteome=args.genes, seed=0x5ED0 + 2)
    t_gen = time.time() - t0
    db = SetDB.from_proteomes(ps)
    max_seqs = max(300, 2 * P)
    cs = ClusterSearch(gpu, host, db, max_seqs=max_seqs, filter_self_match=True)
    cs.last_entries = None
    kmer_thr_used, bin_size_used = cs.kmer_thr, int(cs.bin_size)
    n_batches = (P + B - 1) // B
    set_start = ps.set_start

    def step_range(step_idx):
        b = (rank * (args.steps + args.warmup) + step_idx) % n_batches
        s0, s1 = b * B, min(P, (b + 1) * B)
        return (int(set_start[s0]), int(set_start[s1])), (s1 - s0) * P

    def run_steps(step_ids, keep=False):
        """the given steps (one batch of query proteomes each) streamed through the pipeline: every step gets its own
        aggregation / clusterhits / result, and the prefilter of step k+1 overlaps the alignments of step k"""
        rngs = [step_range(x) for x in step_ids]
        outs = cs.search_stream(db, [r for r, _ in rngs], same_db=True, chunk_queries=args.chunk)
        if keep and outs and outs[-1]['cluster_out'] is not None:
            out = outs[-1]
            hq, ht = out['hit_q'], out['hit_t']
            cs.last_entries = (out['entry_off'], db.pos_in_set[hq], db.pos_in_set[ht],
                               (db.strand[hq] | (db.strand[ht] << 1)).astype(np.uint8),
                               np.zeros(len(hq)) + 1e-30, db.set_size[out['entry_q']])
        return sum(n for _, n in rngs), outs

    if args.warmup:
        run_steps(list(range(args.warmup)))
    gpu.profile(True)
    cs.ctx_al.profile(True)
    for k in cs.stats:
        cs.stats[k] = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gpu.synchronize()
    t0 = time.time()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    pairs_done = 0
    q_len_sum = 0
    summary = np.zeros(4, np.int64)
    stage = {}
    pairs_done, outs = run_steps([args.warmup + k for k in range(args.steps)], keep=True)
    for out in outs:
        summary += np.array([out['entries'], out['matched_hits'], out['clusters'], out['cluster_hits']], np.int64)
        for s, v in out['timing'].items():
            stage[s] = stage.get(s, 0.0) + v
    gathered = None
    if dist is not None:
        # final result gather over RCCL (xGMI): per-rank result summary
        tsum = to_dev(torch.from_numpy(summary))
        gathered = [torch.zeros_like(tsum) for _ in range(world)]
        dist.all_gather(gathered, tsum)
    gpu.synchronize()
    cs.ctx_al.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    if os.environ.get('SD_BENCH_THREADS'):   # per-thread CPU seconds since process start (debugging aid)
        rows = []
        for tid in os.listdir('/proc/self/task'):
            try:
                w = open('/proc/self/task/%s/stat' % tid).read().rsplit(')', 1)
                f = w[1].split()
                rows.append(((int(f[11]) + int(f[12])) / os.sysconf('SC_CLK_TCK'), w[0].split('(', 1)[1], tid))
            except OSError:
                pass
        rows.sort(reverse=True)
        sys.stderr.write('threads: ' + ' | '.join('%s %.1fs' % (n, c) for c, n, _ in rows[:24]) + '\n')
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    if dist is not None:
        tmax = to_dev(torch.tensor([dt], dtype=torch.float64))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_max = float(tmax.item())
        tot = to_dev(torch.tensor([pairs_done], dtype=torch.float64))
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        pairs_total = float(tot.item())
    else:
        dt_max, pairs_total = dt, float(pairs_done)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    prof = dict(gpu.profile_report())
    for k_, v_ in cs.ctx_al.profile_report().items():   # the align stage runs on its own context / stream
        a_ = prof.get(k_, (0.0, 0))
        prof[k_] = (a_[0] + v_[0], a_[1] + v_[1])
    kernels = {k: dict(ms=v[0], launches=int(v[1])) for k, v in prof.items()}
    # variants of one kernel template ("name.variant") are one kernel for the roofline
    grouped = {}
    for k, v in kernels.items():
        g = grouped.setdefault(k.split('.')[0], dict(ms=0.0, launches=0))
        g['ms'] += v['ms']
        g['launches'] += v['launches']
    st = cs.stats
    qlen_steps = 0
    b_pref = algorithmic_bytes(st, int(ps.lengths().mean() * args.steps * B * args.genes))
    pf_ms = sum(v['ms'] for k, v in kernels.items() if k.startswith('prefilter_'))
    sw_ms = sum(v['ms'] for k, v in grouped.items() if k.startswith('sw_score'))
    cells_sw = st['cells_fwd'] + st['cells_rev']
    b_sw = st['pairs'] * (int(ps.lengths().mean()) * 23 + 24)
    dev_kernels = {k: v for k, v in grouped.items() if not k.startswith('host:')}
    dom = max(dev_kernels.items(), key=lambda kv: kv[1]['ms'])[0] if dev_kernels else 'none'
    if dom.startswith('sw_score'):
        alg, per = b_sw * grouped[dom]['ms'] / max(sw_ms, 1e-9), grouped[dom]
    elif dom.startswith('prefilter_'):
        share = kernels[dom]['ms'] / pf_ms if pf_ms > 0 else 1.0
        alg, per = (6 * st['index_hits'] if dom in ('prefilter_gather_hits', 'prefilter_sort_hits') else b_pref * share), grouped[dom]
    else:
        alg, per = 17 * int(summary[1]) + 4 * int(summary[1]), grouped.get(dom, dict(ms=1, launches=1))
    achieved = (alg / max(per['launches'], 1)) / ((per['ms'] / max(per['launches'], 1)) * 1e-3) / 1e9 if per['ms'] > 0 else 0.0
    # HBM traffic per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate runs, tools/pmc_summary.py; chunk size 30000 there, so launches are ~3x larger than here)
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
        if dom in pmc:
            traffic = pmc[dom]['bytes_per_launch'] * (pmc[dom]['launches'] / max(per['launches'] / max(args.steps, 1), 1))
    except (OSError, ValueError, KeyError):
        pass
    roofline = dict(bound='hbm', kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, launches=per['launches'], avg_launch_ms=per['ms'] / max(per['launches'], 1),
                    note=('the score pass is integer-VALU bound (DP state lives in VGPR/LDS): see sw_valu; '
                          'algorithmic bytes = residue streams only') if dom.startswith('sw_score') else 'algorithmic bytes per SURVEY.md 8(d)')
    # VALU view of the score pass: lane-instructions of the inner loop per DP cell (counted in the gfx950 ISA of the
    # dominant variants: packed kernel ~11.1/2 per cell... stated per cell below) against 256 CU x 64 lanes x 2.4 GHz
    VALU_PEAK = 256 * 64 * 2.4e9
    INSTR_PER_CELL = 11.1   # sw_score_pk RT=8: 178 instructions per step of 16 cells
    sw_valu = dict(cells_per_s=cells_sw / (sw_ms * 1e-3) if sw_ms > 0 else 0.0, instr_per_cell=INSTR_PER_CELL, peak_lane_instr_per_s=VALU_PEAK)
    sw_valu['frac'] = sw_valu['cells_per_s'] * INSTR_PER_CELL / VALU_PEAK
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
        'dtype': 'int16',
        'data': 'synthetic',
        'config': {'workload': '%d synthetic proteomes x %d proteins (len~300) all-vs-all, clustersearch --search-mode 0 '
                               '--filter-self-match --max-seqs %d; step = %d query proteomes vs all %d targets'
                               % (P, args.genes, max_seqs, B, P),
                   'parallelism': 'query-set sharding x%d, target index replicated, RCCL final gather' % world},
        'roofline': roofline,
        'sw_gcups': cells_sw / sw_ms / 1e6 if sw_ms > 0 else 0.0,
        'sw_valu': sw_valu,
        'sw_cells': {'forward': st['cells_fwd'], 'reverse': st['cells_rev'], 'traceback': st['cells_tb']},
        'prefilter': {'queries': args.steps * B * args.genes, 'kernel_ms': pf_ms, 'algorithmic_bytes': b_pref,
                      'achieved_GBs': b_pref / pf_ms / 1e6 if pf_ms > 0 else 0.0, 'index_hits': st['index_hits'],
                      'kmers': st['kmers'], 'hits': st['prefilter_hits'],
                      'queries_per_s': args.steps * B * args.genes / dt if dt > 0 else 0.0},
        'kernels': kernels,
        'stage_wall_s': stage,
        'host_cpu_s_per_step': round(host_cpu_s / max(1, args.steps), 3),
        'results': {'entries': int(summary[0]), 'matched_hits': int(summary[1]), 'clusters': int(summary[2]),
                    'cluster_hits': int(summary[3])},
        'setup_s': {'generate': t_gen, 'index_build_host': cs.timing['index_build_s'], 'upload': cs.timing['upload_s']},
        'device': gpu.device_name(),
        'host_cores': os.cpu_count(),
        'host_cpu_quota': effective_cpus(),
    }
    if not args.no_cpu and world == 1:   # the CPU leg belongs to the single-GPU run only
        import tempfile
        ent = os.path.join(tempfile.gettempdir(), 'sd_bench_entries_%d.npz' % os.getpid())
        if cs.last_entries is not None:
            np.savez(ent, eo=cs.last_entries[0], qp=cs.last_entries[1], tp=cs.last_entries[2], sd=cs.last_entries[3],
                     nq=cs.last_entries[5])
        del cs, gpu
        res['cpu_baseline'] = cpu_baseline_subprocess(args, max_seqs, kmer_thr_used, bin_size_used, ent, args.cpu_seconds,
                                                      args.cpu_threads or effective_cpus())
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
