#!/usr/bin/env python3
"""bench.py -- clustersearch hot path (prefilter + SW align + clusterhits) on MI355X.

Workload (BASELINE.json configs[1]): P synthetic proteomes (default 100 x 3000 proteins, len ~300) searched all-vs-all,
--max-seqs max(300, 2P), --filter-self-match.  The target side (k-mer index, masked lookup, sequences) is resident in HBM.
One *step* = clustersearch of one batch of query proteomes against all P target proteomes; K timed steps, W warm-up steps;
`value` = genome pairs of all ranks / max-over-ranks time.

N ranks (one process per GPU, `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) run BASELINE
configs[2]'s structure: a global step holds N*B query proteomes, dealt to the ranks as whole query sets by
sd_shard_query_sets (greedy by residues; a set's hits stay on one rank); every rank searches its sets against its own replica
of the target -- the index is built ONCE on rank 0 and broadcast over RCCL -- and at the end of the timed region the real
per-entry result records are gathered to rank 0 over RCCL (sd_comm_* / sd_gather_results of the C ABI).  No collective on
the data path.  Per-GPU work is fixed as N grows: "scaling": "weak".

Prints ONE JSON line (rank 0):
  roofline      the kernel with the largest accumulated time (HIP events recorded by libsdgpu on its own streams)
  cpu_baseline  the reference's own AVX2 code (oracle/_ref/libsdref.so; kind "port" if it did not travel) on a bounded
                sample of the same workload on this box's host cores -- N = 1 only
  parity_check  untimed: device prefilter rows and alignments of sample queries against the reference rows the cpu_baseline
                leg produced for the same queries
  p1000         N = 1 only (skip with --no-p1000): a short record at BASELINE configs[2]'s size (1 000 proteomes on one
                GPU) with its own reference CPU sample, so that north_star's ">= 10x at 1 000 proteomes" is driver-timed
"""
import argparse
import json
import os
import subprocess
import sys
import resource
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from spacedust_amd.cpus import configure_openmp, effective_cpus  # noqa: E402
# one process per GPU shares the node's CPU quota: size every OpenMP team for this rank's share
configure_openmp(max(1, effective_cpus() // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))))

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)   # 10 steps x 10 query proteomes = one all-vs-all pass over the 100 proteomes
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--proteomes', type=int, default=100)
    ap.add_argument('--genes', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=10, help='query proteomes per rank and step')
    ap.add_argument('--chunk', type=int, default=10000, help='queries per device chunk')
    ap.add_argument('--max-seqs', type=int, default=0, help='result list length (default: max(300, 2 x proteomes), every target set reachable)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg (and the parity check that rides on it)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--no-p1000', action='store_true', help='skip the 1 000-proteome record')
    ap.add_argument('--p1000-steps', type=int, default=2)
    ap.add_argument('--record', action='store_true', help=argparse.SUPPRESS)   # child mode: one plain measurement, JSON out
    return ap.parse_args()


def algorithmic_bytes(stats, q_len_sum):
    """B_pref = 16 S + 6 M + C (7+8) + sum(len_c) + 21 L + 10 H"""
    return 16 * stats['kmers'] + 6 * stats['index_hits'] + 15 * stats['diagonals'] + stats['diag_len'] + 21 * q_len_sum + \
        10 * stats['prefilter_hits']


def cpu_baseline_subprocess(proteomes, genes, max_seqs, kmer_thr, bin_size, entries_path, seconds, n_threads, check=0, check_out=''):
    """The baseline leg runs in a child process (niced, hard timeout, a bounded number of threads) so that it can never
    take the measurement -- or the box -- down with it."""
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'cpu_baseline.py'), '--proteomes', str(proteomes), '--genes', str(genes),
           '--max-seqs', str(max_seqs), '--kmer-thr', str(kmer_thr), '--bin-size', str(bin_size), '--seconds', str(seconds),
           '--threads', str(n_threads), '--entries', entries_path, '--check', str(check), '--check-out', check_out]
    try:
        out = subprocess.run(['nice', '-n', '10'] + cmd, capture_output=True, text=True, timeout=seconds * 6 + 600)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if out.returncode != 0 or not line:
            return dict(value=None, unit='genome-pairs/s', cores=0, kind='failed', sample=(out.stderr or out.stdout)[-300:])
        return json.loads(line[-1])
    except Exception as e:
        return dict(value=None, unit='genome-pairs/s', cores=0, kind='failed', sample=repr(e))


def records_of(out):
    """one int64 row per (query set, target set) entry of a result: ids, #hits, #clusters, order-sensitive checksums of the
    cluster assignment and the P-values' bit patterns -- what the final gather carries"""
    if out['cluster_out'] is None:
        return np.zeros((0, 6), np.int64)
    co, off = out['cluster_out'], out['entry_off']
    rows = []
    for e in range(len(out['entry_q'])):
        a, b = int(off[e]), int(off[e + 1])
        w = np.arange(1, b - a + 1, dtype=np.int64)
        nclu = int(co['n_clusters'][e])
        pbits = np.frombuffer(np.ascontiguousarray(co['pCO'][a:a + nclu]).tobytes(), np.int64)
        rows.append([int(out['entry_q'][e]), int(out['entry_t'][e]), b - a, nclu,
                     int(((co['cluster_of'][a:b].astype(np.int64) + 2) * w).sum() % (1 << 40)), int((pbits % (1 << 40)).sum() % (1 << 40))])
    return np.array(rows, np.int64).reshape(-1, 6)


def parity_check(gpu, host, ps, index, max_seqs, bin_size, kmer_thr, check_path):
    """untimed post-check: the device's prefilter rows and alignments of the sample queries the cpu_baseline leg ran
    through the reference (libsdref), compared field by field"""
    from spacedust_amd import api
    g = np.load(check_path)
    queries = g['queries']
    if len(queries) == 0:
        return dict(queries=0)
    lens = ps.lengths()
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum(lens[queries])
    qres = np.concatenate([ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])] for q in queries])
    sw_b, dg_b, km_b = host.comp_bias(qres, qoff)
    tgt = api.Target(gpu, host, index)
    par = api.prefilter_params(host, ps.n, kmer_thr=kmer_thr, max_hits=max_seqs, bin_size=bin_size)
    hits, cnt, _ = api.prefilter(gpu, tgt, par, qres, qoff, km_b, dg_b, queries.astype(np.uint32))
    bad_rows = n_rows = 0
    for x in range(len(queries)):
        want = g['rows'][int(g['row_off'][x]):int(g['row_off'][x + 1])]
        n = int(cnt[x])
        got = np.stack([hits[x, :n]['seqId'].astype(np.int64), hits[x, :n]['score'].astype(np.int64),
                        hits[x, :n]['diagonal'].astype(np.int64)], 1)
        n_rows += len(want)
        bad_rows += 0 if (got.shape == want.shape and (got == want).all()) else 1
    alns = g['alns']
    bad_aln = 0
    if len(alns):
        qpos = {int(q): x for x, q in enumerate(queries)}
        pq = np.array([qpos[int(a[0])] for a in alns], np.uint32)
        pt = alns[:, 1].astype(np.uint32)
        ts = gpu.seqset(ps.residues, ps.offsets, None)
        qs = gpu.seqset(qres, qoff, sw_b)
        spar = gpu.sw_params(host.matrix(0)[0], int(ps.offsets[-1]))
        res, _ = gpu.sw_align(spar, qs, ts, pq, pt, identity=(alns[:, 0] == alns[:, 1]))
        for i, a in enumerate(alns):
            r = res[i]
            got = [int(r['score']), int(r['qStart']), int(r['qEnd']), int(r['tStart']), int(r['tEnd']), int(r['btLen']),
                   int(r['identical']) if int(r['btLen']) > 0 else 0]
            if int(a[7]) == 0:   # stopped at a gate in the reference: only score and end positions are defined
                bad_aln += got[0] != int(a[2]) or got[2] != int(a[4]) or got[4] != int(a[6])
            else:
                bad_aln += got != [int(v) for v in a[2:9]]
    return dict(queries=int(len(queries)), prefilter_rows=int(n_rows), prefilter_queries_mismatching=int(bad_rows),
                alignments=int(len(alns)), alignments_mismatching=int(bad_aln), against='oracle/_ref/libsdref.so (the reference classes)')


def measure(args, rank, local_rank, world, dist, torch):
    from spacedust_amd.api import Host, Context
    from spacedust_amd.pipeline import SetDB, ClusterSearch, shard_query_sets, RcclGather
    from spacedust_amd.synth import make_proteomes

    P, B = args.proteomes, args.batch
    rehearsal = os.environ.get('SD_BENCH_REHEARSAL') == '1'   # N > 1 on a one-GPU box: every rank on cuda:0, gloo
    dev_index = 0 if (rehearsal or world == 1) else local_rank
    to_dev = (lambda t: t) if (rehearsal or dist is None) else (lambda t: t.cuda())
    n_threads = max(1, effective_cpus() // max(1, world))
    host = Host(n_threads)
    gpu = Context(dev_index)
    t0 = time.time()
    ps = make_proteomes(P, genes_per_proteome=args.genes, seed=0x5ED0 + 2)
    t_gen = time.time() - t0
    db = SetDB.from_proteomes(ps)
    max_seqs = args.max_seqs if args.max_seqs > 0 else max(300, 2 * P)
    # the target index: built once (rank 0, all of the node's cores while the others wait) and broadcast
    t0 = time.time()
    index = None
    k = host.auto_kmer_size(int(ps.offsets[-1]))
    kmer_thr = host.kmer_threshold(5.7, k)
    index_how = 'built on this rank'
    if dist is None or rank == 0:
        full = Host(effective_cpus()) if world > 1 else host
        index = full.build_index(ps.residues, ps.offsets, k, kmer_thr)
        index_how = 'built once on rank 0 (%d threads)' % full.threads
    if dist is not None:
        shape = torch.zeros(2, dtype=torch.int64)
        if rank == 0:
            shape[0], shape[1] = index.table_size, index.n_entries
        shape = to_dev(shape)
        dist.broadcast(shape, 0)
        table_size, n_entries = int(shape[0].item()), int(shape[1].item())
        arrays = []
        for name, dt, n in (('kmer_offsets', np.uint32, table_size + 1), ('entry_seq', np.uint32, n_entries),
                            ('entry_pos', np.uint16, n_entries), ('masked', np.uint8, int(ps.offsets[-1]))):
            if rank == 0:
                t = torch.from_numpy(np.ascontiguousarray(getattr(index, name)).view(np.uint8).copy())
            else:
                t = torch.empty(n * np.dtype(dt).itemsize, dtype=torch.uint8)
            t = to_dev(t)
            dist.broadcast(t, 0)
            arrays.append(t.cpu().numpy().view(dt))
            del t
        index_how += ', broadcast over %s (%.2f GB)' % ('gloo' if rehearsal else 'RCCL', sum(a.nbytes for a in arrays) / 1e9)
        # a wide index (>= 2^32 entries, 10 000 proteomes) also carries its block bases
        nbb = torch.tensor([0 if (rank != 0 or index.block_base is None) else len(index.block_base)], dtype=torch.int64)
        nbb = to_dev(nbb)
        dist.broadcast(nbb, 0)
        block_base = None
        if int(nbb.item()) > 0:
            t = torch.from_numpy(np.ascontiguousarray(index.block_base).view(np.uint8).copy()) if rank == 0 else \
                torch.empty(int(nbb.item()) * 8, dtype=torch.uint8)
            t = to_dev(t)
            dist.broadcast(t, 0)
            block_base = t.cpu().numpy().view(np.uint64)
        if rank != 0:
            from spacedust_amd.api import IndexArrays
            index = IndexArrays(k, kmer_thr, ps.offsets, *arrays, block_base=block_base)
    t_index = time.time() - t0
    cs = ClusterSearch(gpu, host, db, max_seqs=max_seqs, filter_self_match=True, chunk_queries=args.chunk, index=index)
    n_global = world * B
    n_batches = (P + n_global - 1) // n_global
    set_start = ps.set_start
    set_res = [int(ps.offsets[set_start[s + 1]] - ps.offsets[set_start[s]]) for s in range(P)]

    def my_ranges(step_idx):
        """this rank's share of global step step_idx: whole query sets, neighbours merged into one range"""
        b = step_idx % n_batches
        sets = [s % P for s in range(b * n_global, (b + 1) * n_global)]
        mine = sorted(sets[i] for i in shard_query_sets([set_res[s] for s in sets], world, rank))
        ranges = []
        for s in mine:
            a, e = int(set_start[s]), int(set_start[s + 1])
            if ranges and ranges[-1][1] == a:
                ranges[-1][1] = e
            else:
                ranges.append([a, e])
        return [tuple(r) for r in ranges], len(mine) * P

    def run_steps(step_ids):
        rngs, pairs = [], 0
        for x in step_ids:
            r, n = my_ranges(x)
            rngs += r
            pairs += n
        return pairs, cs.search_stream(db, rngs, same_db=True)

    if args.warmup:
        run_steps(list(range(args.warmup)))
    for c_ in cs.contexts:
        c_.profile(True)
    for name in cs.stats:
        cs.stats[name] = 0
    comm = None
    gather_how = 'none (single rank)'
    if dist is not None and not rehearsal:
        try:   # the C ABI's RCCL seam; the communicator is set up before the timed region
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(RcclGather.unique_id()), dtype=torch.uint8).clone()
            uid = to_dev(uid)
            dist.broadcast(uid, 0)
            comm = RcclGather(dev_index, world, rank, bytes(uid.cpu().numpy().tobytes()))
            gather_how = 'sd_gather_results (RCCL, C ABI)'
        except Exception as e:   # never lose the measurement to the seam: torch.distributed carries the gather instead
            comm = None
            gather_how = 'torch.distributed all_gather (sd_comm_init failed: %s)' % str(e)[:120]
    elif dist is not None:
        gather_how = 'torch.distributed all_gather over gloo (rehearsal)'
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    cs.ctx.synchronize()
    t0 = time.time()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    pairs_done, outs = run_steps([args.warmup + x for x in range(args.steps)])
    summary = np.zeros(4, np.int64)
    stage = {}
    for out in outs:
        summary += np.array([out['entries'], out['matched_hits'], out['clusters'], out['cluster_hits']], np.int64)
        for s_, v in out['timing'].items():
            stage[s_] = stage.get(s_, 0.0) + v
    recs = np.concatenate([records_of(o) for o in outs]) if outs else np.zeros((0, 6), np.int64)
    gathered = None
    if dist is not None:
        # the one exchange of the path: every rank's result records to rank 0
        if comm is not None:
            try:
                gathered = comm.gather(recs)
            except Exception as e:
                gather_how = 'torch.distributed all_gather (sd_gather_results failed: %s)' % str(e)[:120]
                comm = None
        if comm is None:
            from spacedust_amd.pipeline import gather_results
            gathered = gather_results(recs, dist, device=None if rehearsal else torch.device('cuda', dev_index))
    for c_ in cs.contexts:
        c_.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.time() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
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
        return None
    prof = {}
    for c_ in cs.contexts:   # every stage runs on its own context / stream (two of them for the alignment lanes)
        for k_, v_ in c_.profile_report().items():
            a_ = prof.get(k_, (0.0, 0))
            prof[k_] = (a_[0] + v_[0], a_[1] + v_[1])
    kernels = {k_: dict(ms=v[0], launches=int(v[1])) for k_, v in prof.items()}
    grouped = {}   # variants of one kernel template ("name.variant") are one kernel for the roofline
    for k_, v in kernels.items():
        g = grouped.setdefault(k_.split('.')[0], dict(ms=0.0, launches=0))
        g['ms'] += v['ms']
        g['launches'] += v['launches']
    st = cs.stats
    n_queries = int(sum(b - a for x in range(args.steps) for a, b in my_ranges(args.warmup + x)[0]))
    q_len_sum = int(ps.lengths().mean() * n_queries)
    b_pref = algorithmic_bytes(st, q_len_sum)
    pf_ms = sum(v['ms'] for k_, v in kernels.items() if k_.startswith('prefilter_'))
    sw_ms = sum(v['ms'] for k_, v in grouped.items() if k_.startswith('sw_score'))
    cells_sw = st['cells_fwd'] + st['cells_rev']
    b_sw = st['pairs'] * (int(ps.lengths().mean()) * 23 + 24)
    dev_kernels = {k_: v for k_, v in grouped.items() if not k_.startswith('host:')}
    dom = max(dev_kernels.items(), key=lambda kv: kv[1]['ms'])[0] if dev_kernels else 'none'
    if dom.startswith('sw_score'):
        alg, per = b_sw * grouped[dom]['ms'] / max(sw_ms, 1e-9), grouped[dom]
    elif dom.startswith('prefilter_'):
        share = kernels[dom]['ms'] / pf_ms if pf_ms > 0 else 1.0
        alg, per = (6 * st['index_hits'] if dom in ('prefilter_gather_hits', 'prefilter_sort_hits') else b_pref * share), grouped[dom]
    else:
        alg, per = 17 * int(summary[1]) + 4 * int(summary[1]), grouped.get(dom, dict(ms=1, launches=1))
    achieved = (alg / max(per['launches'], 1)) / ((per['ms'] / max(per['launches'], 1)) * 1e-3) / 1e9 if per['ms'] > 0 else 0.0
    # HBM traffic per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate runs, tools/pmc_summary.py): the newest round's file that has the kernel
    traffic, traffic_src = None, None
    for fn in ('r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', fn)))
            if dom in pmc:
                traffic = pmc[dom]['bytes_per_launch'] * (pmc[dom]['launches'] / max(per['launches'] / max(args.steps, 1), 1))
                traffic_src = 'profiles/' + fn
                break
        except (OSError, ValueError, KeyError):
            pass
    roofline = dict(bound='hbm', kernel=dom, achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, traffic_source=traffic_src, launches=per['launches'], avg_launch_ms=per['ms'] / max(per['launches'], 1),
                    note=('the score pass is integer-VALU bound (DP state lives in VGPR/LDS): see sw_valu; algorithmic bytes = '
                          'residue streams only') if dom.startswith('sw_score') else 'algorithmic bytes per SURVEY.md 8(d)')
    # VALU view of the score pass: lane-instructions of the inner loop per DP cell against the VALU issue ceiling.  Both
    # constants come from measurements kept under profiles/ (tools/valu_peak.py: issue micro-benchmark; PMC: SQ_INSTS_VALU per
    # cell); the literals are the fallback when that file is absent
    valu = dict(instr_per_cell=11.1, peak_lane_instr_per_s=256 * 64 * 2.4e9,
                source='ISA count of sw_score_pk RT=8 (178 per 16 cells); 256 CU x 64 lanes x 2.4 GHz')
    try:
        v = json.load(open(os.path.join(ROOT, 'profiles', 'r02_valu_calibration.json')))
        valu = dict(instr_per_cell=v['instr_per_cell'], peak_lane_instr_per_s=v['peak_lane_instr_per_s'], source='profiles/r02_valu_calibration.json')
    except (OSError, ValueError, KeyError):
        pass
    sw_valu = dict(cells_per_s=cells_sw / (sw_ms * 1e-3) if sw_ms > 0 else 0.0, **valu)
    sw_valu['frac'] = sw_valu['cells_per_s'] * sw_valu['instr_per_cell'] / sw_valu['peak_lane_instr_per_s']
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
                               '--filter-self-match --max-seqs %d; step = %d query proteomes per rank vs all %d targets'
                               % (P, args.genes, max_seqs, B, P),
                   'parallelism': 'whole query sets dealt to %d rank(s) by sd_shard_query_sets, target index replicated (%s), '
                                  'final result gather: %s' % (world, index_how, gather_how)},
        'roofline': roofline,
        'sw_gcups': cells_sw / sw_ms / 1e6 if sw_ms > 0 else 0.0,
        'sw_valu': sw_valu,
        'sw_cells': {'forward': st['cells_fwd'], 'reverse': st['cells_rev'], 'traceback': st['cells_tb']},
        'prefilter': {'queries': n_queries, 'kernel_ms': pf_ms, 'algorithmic_bytes': b_pref,
                      'achieved_GBs': b_pref / pf_ms / 1e6 if pf_ms > 0 else 0.0, 'index_hits': st['index_hits'],
                      'kmers': st['kmers'], 'hits': st['prefilter_hits'], 'queries_per_s': n_queries / dt if dt > 0 else 0.0},
        'kernels': kernels,
        'stage_wall_s': stage,
        'host_cpu_s_per_step': round(host_cpu_s / max(1, args.steps), 3),
        'results': {'entries': int(summary[0]), 'matched_hits': int(summary[1]), 'clusters': int(summary[2]),
                    'cluster_hits': int(summary[3])},
        'setup_s': {'generate': t_gen, 'index': t_index, 'upload': cs.timing['upload_s']},
        'device': gpu.device_name(),
        'host_cores': os.cpu_count(),
        'host_cpu_quota': effective_cpus(),
    }
    if gathered is not None:
        rows = np.concatenate([np.asarray(g, np.int64).reshape(-1, 6) for g in gathered]) if len(gathered) else np.zeros((0, 6), np.int64)
        res['gather'] = {'how': gather_how, 'records': int(len(rows)), 'records_per_rank': [int(np.asarray(g).size // 6) for g in gathered],
                         'clusters_in_records': int(rows[:, 3].sum()) if len(rows) else 0}
        res['multi_gpu_note'] = ('weak scaling over query sets as in BASELINE configs[2]; an 8-GPU curve exists only where the driver '
                                 'ran this command with --gpus 8')
    extras = dict(ps=ps, index=index, max_seqs=max_seqs, kmer_thr=kmer_thr, bin_size=int(cs.bin_size), gpu=gpu, host=host,
                  last=outs[-1] if outs else None, db=db)
    del cs
    return res, extras


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if os.environ.get('SD_BENCH_REHEARSAL') == '1':
            torch.cuda.set_device(0)
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    out = measure(args, rank, local_rank, world, dist, torch)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    res, ex = out
    if not args.no_cpu and world == 1:   # the CPU leg (and the parity sample it produces) belongs to the single-GPU run
        tmp = tempfile.gettempdir()
        ent = os.path.join(tmp, 'sd_bench_entries_%d.npz' % os.getpid())
        chk = os.path.join(tmp, 'sd_bench_check_%d.npz' % os.getpid())
        last, db = ex['last'], ex['db']
        if last is not None and last['cluster_out'] is not None:
            hq, ht = last['hit_q'], last['hit_t']
            np.savez(ent, eo=last['entry_off'], qp=db.pos_in_set[hq], tp=db.pos_in_set[ht],
                     sd=(db.strand[hq] | (db.strand[ht] << 1)).astype(np.uint8), nq=db.set_size[last['entry_q']])
        res['cpu_baseline'] = cpu_baseline_subprocess(args.proteomes, args.genes, ex['max_seqs'], ex['kmer_thr'], ex['bin_size'], ent,
                                                      args.cpu_seconds, args.cpu_threads or effective_cpus(), check=24, check_out=chk)
        try:
            if os.path.exists(chk):
                res['parity_check'] = parity_check(ex['gpu'], ex['host'], ex['ps'], ex['index'], ex['max_seqs'], ex['bin_size'],
                                                   ex['kmer_thr'], chk)
            else:
                res['parity_check'] = dict(queries=0, note='the reference library did not travel or the CPU leg failed')
        except Exception as e:
            res['parity_check'] = dict(error=repr(e)[:300])
        for f in (ent, chk):
            if os.path.exists(f):
                os.remove(f)
    del ex
    if args.record:   # child of the p1000 leg
        print(json.dumps(res))
        return
    if world == 1 and not args.no_p1000 and args.proteomes != 1000:
        # BASELINE configs[2]'s size on this one GPU, in a child process (fresh device memory): a short run with its own
        # reference CPU sample -- north_star quotes its >= 10x target at this size
        cmd = [sys.executable, os.path.abspath(__file__), '--record', '--proteomes', '1000', '--steps', str(args.p1000_steps), '--warmup', '1',
               '--batch', str(args.batch), '--chunk', str(args.chunk), '--no-p1000'] + (['--no-cpu'] if args.no_cpu else []) + \
              ['--cpu-seconds', str(min(args.cpu_seconds, 12.0))]
        try:
            t0 = time.time()
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
            line = [l for l in p.stdout.splitlines() if l.startswith('{')]
            if p.returncode == 0 and line:
                r = json.loads(line[-1])
                res['p1000'] = {k: r.get(k) for k in ('value', 'unit', 'steps', 'ms_per_step', 'config', 'sw_gcups', 'cpu_baseline', 'parity_check',
                                                      'results', 'setup_s', 'host_cpu_s_per_step')}
                res['p1000']['wall_s'] = time.time() - t0
                cb = r.get('cpu_baseline') or {}
                if cb.get('value'):
                    res['p1000']['gpu_over_cpu'] = r['value'] / cb['value']
            else:
                res['p1000'] = dict(error=(p.stderr or p.stdout)[-300:])
        except Exception as e:
            res['p1000'] = dict(error=repr(e)[:300])
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
