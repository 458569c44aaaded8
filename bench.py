#!/usr/bin/env python3
"""bench.py -- clustersearch hot path (prefilter + SW align + clusterhits) on MI355X.

Workload (BASELINE.json: the configuration north_star quotes its target on -- 1 000 synthetic proteomes; configs[2] on one GPU):
P synthetic proteomes (default 1 000 x 3 000 proteins, len ~300) searched all-vs-all, --max-seqs max(300, 2P), --filter-self-match.
The target side (k-mer index, masked lookup, sequences) is resident in HBM.  One *step* = clustersearch of one batch of query
proteomes (default 4 per rank) against all P target proteomes; K timed steps, W warm-up steps; `value` = genome pairs of all
ranks / max-over-ranks time.

N ranks (one process per GPU, `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) run BASELINE
configs[2]'s structure: a global step holds N*B query proteomes, dealt to the ranks as whole query sets by
sd_shard_query_sets (greedy by residues; a set's hits stay on one rank); every rank searches its sets against its own replica
of the target -- every rank builds the index on its own GPU (sd_target_build) -- and at the end of the timed region the real
per-entry result records are gathered to rank 0 over RCCL (sd_comm_* / sd_gather_results of the C ABI).  No collective on
the data path.  Per-GPU work is fixed as N grows: "scaling": "weak".

Prints ONE compact JSON line (rank 0; a few KB -- everything bulky goes to the side file gpurun_out/bench_detail.json):
  roofline      the prefilter kernel with the largest accumulated time (HIP events recorded by libsdgpu on its own streams)
  cpu_baseline  the reference's own AVX2 code (oracle/_ref/libsdref.so; kind "port" if it did not travel) on a bounded
                sample of the same workload on this box's host cores -- N = 1 only
  parity_check  untimed: device prefilter rows and alignments of sample queries against the reference rows the cpu_baseline
                leg produced for the same queries; the (query set, target set) entries of one whole measured query set against
                an independent aggregation of the reference's rows and alignments; the cluster records of measured entries
                against the reference's clusterhits functions
  p100 / p10000 / iter3   N = 1 only: short child records at BASELINE configs[1] / configs[4] / configs[3]
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
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--proteomes', type=int, default=1000)   # the size north_star quotes its target on (BASELINE configs[2], one GPU)
    ap.add_argument('--genes', type=int, default=3000)
    ap.add_argument('--batch', type=int, default=0, help='query proteomes per rank and step (default: 4 at 1 000 proteomes and beyond, else 10)')
    ap.add_argument('--chunk', type=int, default=0, help='queries per device chunk at most (0: the library\'s choice by target size: 10 000, or 2 500 against 10^6 target sequences and more)')
    ap.add_argument('--max-seqs', type=int, default=0, help='result list length (default: max(300, 2 x proteomes), every target set reachable)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg (and the parity check that rides on it)')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--no-children', '--no-p1000', dest='no_children', action='store_true', help='skip the child records (p100, p10000, iter3)')
    ap.add_argument('--no-p100', action='store_true', help='skip the 100-proteome record')
    ap.add_argument('--p100-steps', type=int, default=10)
    ap.add_argument('--detail-out', default='', help='side file for the bulky parts of the record (default gpurun_out/bench_detail.json)')
    ap.add_argument('--no-p10000', action='store_true', help='skip the 10 000-proteome record')
    ap.add_argument('--p10000-steps', type=int, default=1)
    ap.add_argument('--p10000-batch', type=int, default=1, help='query proteomes per step of the 10 000-proteome record')
    ap.add_argument('--p10000-chunk', type=int, default=1000, help='queries per device chunk of the 10 000-proteome record (lists of up to 20 000 hits per query)')
    ap.add_argument('--leg-budget', type=float, default=330.0, help='a child record (p100, p10000, iter3) is started only while the run is '
                                                                     'younger than this many seconds; later ones are reported as skipped')
    ap.add_argument('--no-iter3', action='store_true', help='skip the --num-iterations 3 record (BASELINE configs[3] at 1 000 target proteomes)')
    ap.add_argument('--iter3-queries', type=int, default=20, help='query proteomes of the --num-iterations 3 record')
    ap.add_argument('--no-index-check', action='store_true', help='skip the sampled check of the device-built index against the host builder')
    ap.add_argument('--strong', action='store_true', help='strong scaling: --batch query proteomes per step in TOTAL, dealt over the ranks '
                                                          '(BASELINE configs[2] as written: one query set of proteomes split over N GPUs)')
    ap.add_argument('--record', action='store_true', help=argparse.SUPPRESS)   # child mode: one plain measurement, JSON out
    a = ap.parse_args()
    if a.batch <= 0:
        a.batch = 4 if a.proteomes >= 1000 else 10
    return a


def algorithmic_bytes(stats, q_len_sum):
    """B_pref = 16 S + 6 M + C (7+8) + sum(len_c) + 21 L + 10 H"""
    return 16 * stats['kmers'] + 6 * stats['index_hits'] + 15 * stats['diagonals'] + stats['diag_len'] + 21 * q_len_sum + \
        10 * stats['prefilter_hits']


def isolated_prefilter(cs, host, ps, k, kmer_thr, max_seqs, bin_size, n_queries=8192):
    """untimed: the prefilter of one block of queries ALONE on the device (the pipeline is idle), kernel times by HIP events --
    what the stage's kernels do when nothing shares the GPU with them, next to the in-pipeline numbers of `roofline`"""
    from spacedust_amd import api
    n = min(n_queries, ps.n)
    qoff = ps.offsets[:n + 1].astype(np.uint64)
    qres = ps.residues[:int(qoff[-1])]
    sw_b, dg_b, km_b = host.comp_bias(qres, qoff, k=k)
    par = api.prefilter_params(host, ps.n, kmer_thr=kmer_thr, max_hits=max_seqs, bin_size=bin_size, k=k)
    ids = np.arange(n, dtype=np.uint32)
    tgt = cs.target_view()
    api.prefilter(cs.ctx, tgt, par, qres, qoff, km_b, dg_b, ids)          # warm: workspaces of this shape
    cs.ctx.profile(True)
    hits, cnt, st = api.prefilter(cs.ctx, tgt, par, qres, qoff, km_b, dg_b, ids, want_stats=True)
    rep = {k_: v[0] for k_, v in cs.ctx.profile_report().items() if k_.startswith('prefilter_')}
    cs.ctx.profile(True)
    ms = sum(rep.values())
    stats = dict(kmers=int(st[:, 0].sum()), index_hits=int(st[:, 1].sum()), diagonals=int(st[:, 2].sum()), diag_len=int(st[:, 3].sum()),
                 prefilter_hits=int(cnt[cnt != 0xFFFFFFFF].sum()))
    b = algorithmic_bytes(stats, int(qoff[-1]))
    return dict(queries=int(n), kernel_ms=ms, algorithmic_bytes=int(b), stage_achieved=b / ms / 1e6 if ms > 0 else 0.0,
                stage_frac=b / ms / 1e6 / HBM_PEAK_GBS if ms > 0 else 0.0, kernel_ms_by_name={k_: round(v, 3) for k_, v in sorted(rep.items())},
                note='same kernels, same algorithmic bytes (SURVEY.md 8(d)), nothing else on the device')


def cpu_baseline_subprocess(proteomes, genes, max_seqs, kmer_thr, bin_size, entries_path, seconds, n_threads, check=0, check_out='',
                            check_sets=(), check_entries=32):
    """The baseline leg runs in a child process (niced, hard timeout, a bounded number of threads) so that it can never
    take the measurement -- or the box -- down with it."""
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'cpu_baseline.py'), '--proteomes', str(proteomes), '--genes', str(genes),
           '--max-seqs', str(max_seqs), '--kmer-thr', str(kmer_thr), '--bin-size', str(bin_size), '--seconds', str(seconds),
           '--threads', str(n_threads), '--entries', entries_path, '--check', str(check), '--check-out', check_out,
           '--check-sets', ','.join(str(int(x)) for x in check_sets), '--check-entries', str(check_entries),
           '--gate-seconds', str(max(4.0, seconds / 2))]
    try:
        out = subprocess.run(['nice', '-n', '10'] + cmd, capture_output=True, text=True, timeout=seconds * 6 + 900)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if out.returncode != 0 or not line:
            return dict(value=None, unit='genome-pairs/s', cores=0, kind='failed', sample=(out.stderr or out.stdout)[-300:])
        return json.loads(line[-1])
    except Exception as e:
        return dict(value=None, unit='genome-pairs/s', cores=0, kind='failed', sample=repr(e))


def back_half_check(g, last, db):
    """the back half of the measured run against reference-derived results (check file of tools/cpu_baseline.py):
    (a) the (query set, target set) entries of whole measured query sets -- entry keys, hits in order, P-value bits -- against
        the reference's rows and alignments of every query of those sets pushed through oracle/agg_restatement.py
        (besthitbyset.cpp:41-144, combinehits.cpp:74-234);
    (b) the cluster records of measured entries -- partition, member ranks, sizes, both P-values bit for bit -- against the
        reference's own clusterhits functions (oracle/_ref/libsdref_ch.so, ClusterHits.cpp:295-492) on the same entries"""
    res = {}
    eo = last['entry_off']
    if 'agg_keys' in g.files:
        sets = set(int(x) for x in g['agg_sets'])
        want = {}
        for i, (qs, ts) in enumerate(g['agg_keys']):
            x0, x1 = int(g['agg_off'][i]), int(g['agg_off'][i + 1])
            want[(int(qs), int(ts))] = (g['agg_q'][x0:x1], g['agg_t'][x0:x1], g['agg_p'][x0:x1])
        got = {}
        for e in range(len(last['entry_q'])):
            if int(last['entry_q'][e]) in sets:
                x0, x1 = int(eo[e]), int(eo[e + 1])
                got[(int(last['entry_q'][e]), int(last['entry_t'][e]))] = (last['hit_q'][x0:x1], last['hit_t'][x0:x1], last['hit_pval'][x0:x1])
        bad = len(set(want) ^ set(got))
        n_hits = 0
        for k_ in set(want) & set(got):
            w, h = want[k_], got[k_]
            n_hits += len(w[0])
            same = len(w[0]) == len(h[0]) and (w[0] == h[0].astype(np.int64)).all() and (w[1] == h[1].astype(np.int64)).all() and \
                np.asarray(w[2], np.float64).tobytes() == np.asarray(h[2], np.float64).tobytes()
            bad += 0 if same else 1
        res.update(query_sets=sorted(sets), entries=len(want), entries_measured=len(got), entry_hits=int(n_hits), entries_mismatching=int(bad),
                   reference_alignments=int(g['agg_alignments']), reference_seconds=round(float(g['agg_seconds']), 1))
    if 'ch_ncl' in g.files and last['cluster_out'] is not None:
        co = last['cluster_out']
        n = int(g['ch_entries'])
        bad = n_clu = 0
        c0 = 0   # cluster-wise arrays of the reference side are concatenated entry by entry
        for e in range(n):
            x0, x1 = int(eo[e]), int(eo[e + 1])
            nc = int(g['ch_ncl'][e])
            rcof, rrk = g['ch_cof'][x0:x1], g['ch_rank'][x0:x1]
            ok = int(co['n_clusters'][e]) == nc and (co['cluster_of'][x0:x1] == rcof).all()
            if ok:
                clustered = rcof != 0xFFFFFFFF
                ok = (co['rank'][x0:x1][clustered] == rrk[clustered]).all() and (co['size'][x0:x0 + nc] == g['ch_size'][c0:c0 + nc]).all() and \
                    co['pCO'][x0:x0 + nc].tobytes() == g['ch_pco'][c0:c0 + nc].tobytes() and \
                    co['pMH'][x0:x0 + nc].tobytes() == g['ch_pmh'][c0:c0 + nc].tobytes()
            bad += 0 if ok else 1
            n_clu += nc
            c0 += nc
        res.update(cluster_entries=n, clusters=int(n_clu), cluster_entries_mismatching=int(bad))
    return res


def parity_check(gpu, host, ps, k, max_seqs, bin_size, kmer_thr, check_path, last=None, db=None):
    """untimed post-check: the device's prefilter rows and alignments of the sample queries the cpu_baseline leg ran
    through the reference (libsdref), compared field by field; then the back half (back_half_check)"""
    from spacedust_amd import api
    g = np.load(check_path)
    queries = g['queries']
    if len(queries) == 0:
        return dict(queries=0)
    lens = ps.lengths()
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum(lens[queries])
    qres = np.concatenate([ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])] for q in queries])
    sw_b, dg_b, km_b = host.comp_bias(qres, qoff, k=k)
    tgt = api.Target.build_on_device(gpu, host, ps.residues, ps.offsets, k=k, kmer_thr=kmer_thr)
    par = api.prefilter_params(host, ps.n, kmer_thr=kmer_thr, max_hits=max_seqs, bin_size=bin_size, k=k)
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
    res = dict(queries=int(len(queries)), prefilter_rows=int(n_rows), prefilter_queries_mismatching=int(bad_rows),
               alignments=int(len(alns)), alignments_mismatching=int(bad_aln))
    if last is not None:
        res.update(back_half_check(g, last, db))
    res['against'] = 'oracle/_ref/libsdref.so + libsdref_ch.so (the reference classes / functions), oracle/agg_restatement.py'
    return res


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
    ps = make_proteomes(P, genes_per_proteome=args.genes, seed=0x5ED0 + 2, workers=n_threads if P >= 128 else 1)   # this rank's share of the cores (small sets: in process)
    t_gen = time.time() - t0
    db = SetDB.from_proteomes(ps)
    max_seqs = args.max_seqs if args.max_seqs > 0 else max(300, 2 * P)
    # the target index is built on every rank's own GPU from the residues (sd_target_build: masking, k-mer lists, list starts):
    # no host index, nothing to broadcast
    t0 = time.time()
    k = host.auto_kmer_size(int(ps.offsets[-1]))
    kmer_thr = host.kmer_threshold(5.7, k)
    index_how = 'built on every rank\'s GPU (sd_target_build)'
    cs = ClusterSearch(gpu, host, db, max_seqs=max_seqs, filter_self_match=True, chunk_queries=args.chunk)
    t_index = time.time() - t0   # contexts, index build on the device, target sequences
    index_check = None
    if rank == 0 and not args.no_index_check:
        # untimed: the device-built index against the host builder (IndexBuilder restatement) on 9 600 sample sequences --
        # every host entry present, nothing else for these sequences, masked residues equal
        t1 = time.time()
        index_check = cs.target_view().sample_check(host, ps.residues, ps.offsets, kmer_thr)
        index_check['seconds'] = time.time() - t1
        index_check['against'] = 'sd_host_index_build (host restatement of IndexBuilder::fillDatabase, pinned on libsdref in tests/)'
    n_global = B if args.strong else world * B
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

    rec_buf = [None]   # a rank of an N > 1 run builds its records in the communicator's pinned buffer (set after the warm-up)

    gather_stream = [None]   # N > 1 over RCCL: the records of a step are gathered while the next steps run (sd_gather_stream_*)

    def run_steps(step_ids):
        rngs, pairs, rounds = [], 0, []
        for k_, x in enumerate(step_ids):
            r, n = my_ranges(x)
            rngs += r
            rounds += [k_] * len(r)
            pairs += n
        if gather_stream[0] is not None:   # one round per step; the ranges' records leave through the stream's records sink
            gather_stream[0].stream_begin(rounds, len(step_ids), out=gather_out)
            return pairs, cs.search_stream(db, rngs, same_db=True, want_records=True, arrays='last', records_sink=gather_stream[0].stream_sink())
        # every rank builds its ranges' cluster records inside the stream; a rank of an N > 1 run also copies them out for the gather,
        # a single rank leaves them in the result handles: N = 1 and N > 1 time the same work up to the hand-over
        return pairs, cs.search_stream(db, rngs, same_db=True, want_records=True if dist is not None else ('build' if os.environ.get('SD_BENCH_RECORDS', '1') != '0' else False), arrays='last',
                                       records_buffer=rec_buf[0])   # (the parity leg reads the last range's arrays; every range's counters)

    warm_bytes = 0
    if args.warmup:
        _, w_outs = run_steps(list(range(args.warmup)))
        if dist is not None and w_outs and w_outs[-1].get('records_all') is not None:
            warm_bytes = int(len(w_outs[-1]['records_all']))
        del w_outs
    for c_ in cs.contexts:
        c_.profile(True)
    for name in cs.stats:
        cs.stats[name] = 0
    comm = None
    gather_out = None
    gather_how = 'none (single rank)'
    if dist is not None and not rehearsal:
        # the C ABI's RCCL seam; the communicator is set up before the timed region.  A failure here is a failed run.
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(RcclGather.unique_id()), dtype=torch.uint8).clone()
        uid = to_dev(uid)
        dist.broadcast(uid, 0)
        comm = RcclGather(dev_index, world, rank, bytes(uid.cpu().numpy().tobytes()))
        gather_how = 'sd_gather_results (RCCL, C ABI): the cluster records of every rank to rank 0, which writes the TSV from the gathered buffer'
        # pinned, pre-sized staging from the warm-up's record size (sd_comm_host_buffer): every rank builds its records in its send buffer,
        # the root receives into its gather buffer -- no pageable staging, no allocation, no size probe inside the timed region
        if args.warmup:   # (the same on every rank: what follows is collective)
            est = int(warm_bytes / args.warmup * args.steps * 1.25) + (64 << 20)
            szs = to_dev(torch.tensor([est], dtype=torch.int64))
            dist.all_reduce(szs, op=dist.ReduceOp.MAX)   # (every rank the same size: the root's buffer is world x that)
            est = int(szs.item())
            try:
                rec_buf[0] = comm.host_buffer(0, est)
                gather_out = comm.host_buffer(1, est * world) if rank == 0 else np.zeros(0, np.uint8)
                gather_how += '; pinned pre-sized staging (%d MB per rank)' % (est >> 20)
            except Exception as e:   # (not enough pinned memory: the pageable path)
                rec_buf[0], gather_out = None, None
                gather_how += '; pageable staging (%r)' % (e,)
            okf = to_dev(torch.tensor([1 if rec_buf[0] is not None else 0], dtype=torch.int64))
            dist.all_reduce(okf, op=dist.ReduceOp.MIN)   # all ranks or none: the gather's calls are collective
            if int(okf.item()) == 0:
                rec_buf[0], gather_out = None, None
            elif os.environ.get('SD_BENCH_GATHER_STREAM', '1') != '0':
                # round by round behind the stream (SD_BENCH_GATHER_STREAM=0: one blob per rank behind the last kernel)
                gather_stream[0] = comm
                gather_how += '; one round per step on the communicator\'s stream while the next steps are searched (sd_gather_stream_*)'
    elif dist is not None:
        gather_how = 'rehearsal on one GPU (RCCL refuses two ranks per device): the same records over torch.distributed / gloo'
        if args.warmup and os.environ.get('SD_BENCH_GATHER_STREAM', '1') != '0':
            # ... or, as `sdgpu clustersearch` does for ranks that share a device, round by round over the C ABI's TCP rendezvous: the
            # same sd_gather_stream_* round logic the RCCL path runs, with real ranks
            from spacedust_amd.pipeline import TcpGather
            est = int(warm_bytes / args.warmup * args.steps * 1.25) + (64 << 20)
            szs = torch.tensor([est], dtype=torch.int64)
            dist.all_reduce(szs, op=dist.ReduceOp.MAX)
            est = int(szs.item())
            comm = TcpGather(world, rank, os.environ.get('MASTER_ADDR', '127.0.0.1'), int(os.environ.get('MASTER_PORT', '29500')) + 7)
            gather_out = np.empty(est * world, np.uint8) if rank == 0 else np.zeros(0, np.uint8)
            gather_stream[0] = comm
            gather_how = ('rehearsal on one GPU (RCCL refuses two ranks per device): one round per step over the C ABI\'s TCP rendezvous while the '
                          'next steps are searched (sd_gather_stream_begin_tcp)')
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    cs.ctx.synchronize()
    t0 = time.time()
    if os.environ.get('SD_DEBUG_WS'):   # the library's workspace log carries the same clock
        print('[bench] t=%.3f timed region starts' % time.monotonic(), file=sys.stderr)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    thr0 = _thread_cpu_snapshot()
    dl0 = cs.download_bytes()
    pairs_done, outs = run_steps([args.warmup + x for x in range(args.steps)])
    summary = np.zeros(4, np.int64)
    stage = {}
    for out in outs:
        summary += np.array([out['entries'], out['matched_hits'], out['clusters'], out['cluster_hits']], np.int64)
        for s_, v in out['timing'].items():
            stage[s_] = stage.get(s_, 0.0) + v
    gathered, gather_sizes, tsv_info = None, None, None
    if dist is not None:
        # the one exchange of the path: every rank's cluster records to rank 0, which writes the result TSV from the gathered buffer
        recs = outs[-1]['records_all'] if outs else np.zeros(0, np.uint8)   # the ranges' records back to back in one buffer
        t_g0 = time.time()
        if gather_stream[0] is not None:   # the rounds of the earlier steps are on the root already: this waits for the last ones
            gathered, _, round_sizes = comm.stream_end()
            gather_sizes = round_sizes.sum(axis=0).astype(np.uint64)
        elif comm is not None:
            gathered, gather_sizes = comm.gather_bytes(recs, out=gather_out)
        else:
            from spacedust_amd.pipeline import gather_results
            parts = gather_results(np.frombuffer(np.concatenate([recs, np.zeros((-len(recs)) % 8, np.uint8)]).tobytes(), np.int64), dist)
            lens = gather_results(np.array([len(recs)], np.int64), dist)
            gather_sizes = np.array([int(l[0]) for l in lens], np.uint64)
            gathered = np.concatenate([np.asarray(p_, np.int64).view(np.uint8)[:int(n_)] for p_, n_ in zip(parts, gather_sizes)]) if rank == 0 else None
        gather_s = time.time() - t_g0
    for c_ in cs.contexts:
        c_.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.time() - t0
    if os.environ.get('SD_DEBUG_WS'):
        print('[bench] t=%.3f timed region ends' % time.monotonic(), file=sys.stderr)
    thr1 = _thread_cpu_snapshot()
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    dl1 = cs.download_bytes()
    host_cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    if dist is not None:
        tmax = to_dev(torch.tensor([dt], dtype=torch.float64))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_max = float(tmax.item())
        tot = to_dev(torch.tensor([pairs_done], dtype=torch.float64))
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        pairs_total = float(tot.item())
        sm = to_dev(torch.from_numpy(summary.copy()))   # the job's result counters: all ranks
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        summary = sm.cpu().numpy()
    else:
        dt_max, pairs_total = dt, float(pairs_done)
    if rank != 0:
        return None
    if gathered is not None:
        # rank 0 writes the result TSV from the gathered buffer (after the timed region: the single-rank line does not write one either)
        from spacedust_amd.pipeline import write_records_tsv
        tsv = os.path.join(tempfile.gettempdir(), 'sd_bench_%d.tsv' % os.getpid())
        t_tsv = time.time()
        n_clu, n_hit = write_records_tsv(gathered, tsv, db, db)
        tsv_info = dict(bytes=int(os.path.getsize(tsv)), clusters=n_clu, hits=n_hit, seconds=time.time() - t_tsv,
                        clusters_match_results=bool(n_clu == int(summary[2]) and n_hit == int(summary[3])))
        os.remove(tsv)
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
    pf_ms = sum(v['ms'] for k_, v in kernels.items() if k_.startswith('prefilter_'))   # ('stat.*' entries are counters, not times)
    sw_ms = sum(v['ms'] for k_, v in grouped.items() if k_.startswith('sw_score'))
    cells_sw = st['cells_fwd'] + st['cells_rev']
    b_sw = st['pairs'] * (int(ps.lengths().mean()) * 23 + 24)
    # One roofline object per stage.  Prefilter: HBM -- every kernel with its own algorithmic bytes (the bytes the stage's data
    # structures make it move once: K similar k-mers of 8 B, H index hits of 8 B, the table slices once per sub-batch) over its
    # event-timed duration; the headline `roofline` is the stage with the largest share of the kernel time and, inside it, the
    # kernel with the largest.  Alignment: the score pass is integer-VALU bound -- `roofline_sw` prices it against the measured
    # VALU issue peak (sw_valu).
    K, H, Cn = st['kmers'], st['index_hits'], st['diagonals']
    n_sub = max(kernels.get('prefilter_emit_kmers', dict(launches=1))['launches'], 1)
    tab = 4 * ((20 ** k) + 1)
    ent = 8 * cs.index_entries
    join = 'prefilter_join_scatter' in kernels
    HL = int(kernels.get('stat.prefilter_hits_left', dict(ms=H))['ms'])   # hits behind the hot-target filter (H without it)
    alg_of = {'prefilter_count_kmers': 21 * q_len_sum,
              'prefilter_emit_kmers': 8 * K if join else 16 * K + 0,
              'prefilter_kmer_partition': 24 * K,
              'prefilter_join_count': 8 * K + n_sub * tab,
              'prefilter_join_scatter': 8 * K + n_sub * (tab + ent) + 8 * H,
              'prefilter_gather_hits': 12 * K + 6 * H + 10 * H,
              'prefilter_hot_filter': 8 * H + 8 * HL,          # every hit looked at once, the survivors rewritten
              'prefilter_segment_match': 8 * HL + 8 * Cn,
              'prefilter_partition_hits': 24 * HL,
              'prefilter_coarse_split': 24 * H,
              'prefilter_bucket_match': 8 * HL + 8 * Cn,
              'prefilter_bucket_match_big': 0,
              'prefilter_score_diag': 15 * Cn + st['diag_len'],
              'prefilter_keep_max': 8 * Cn,
              'prefilter_select_hits': 12 * Cn + 10 * st['prefilter_hits']}
    pmc = {}
    pmc_src = None
    for fn in ('r06_pmc_traffic.json', 'r05p_pmc_traffic.json', 'r05i_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04d_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json'):
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', fn)))
            pmc_src = 'profiles/' + fn
            break
        except (OSError, ValueError):
            pass
    per_kernel = {}
    for k_, v in kernels.items():
        if not k_.startswith('prefilter_') or v['ms'] <= 0:
            continue
        alg = alg_of.get(k_, 0)
        e_ = dict(ms=v['ms'], launches=v['launches'], avg_launch_ms=v['ms'] / max(v['launches'], 1), algorithmic_bytes=int(alg),
                  achieved=alg / v['ms'] / 1e6, frac=alg / v['ms'] / 1e6 / HBM_PEAK_GBS)
        e_['algorithmic_per_query'] = alg / max(n_queries, 1)
        if k_ in pmc:   # bytes at the memory side (FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes of the same command)
            e_['traffic'] = pmc[k_]['bytes_per_launch']
            if pmc[k_].get('bytes_per_query'):
                # compared PER QUERY: the PMC run's launches and the timed run's launches hold different numbers of queries (the PMC file
                # carries the queries its process ran through the prefilter: bench.py's prefilter_queries_in_process of that run)
                e_['traffic_per_query'] = pmc[k_]['bytes_per_query']
                e_['traffic_over_algorithmic'] = pmc[k_]['bytes_per_query'] / max(e_['algorithmic_per_query'], 1)
            else:
                e_['traffic_over_algorithmic'] = None   # a PMC file without its query count: bytes per launch of another launch size
        per_kernel[k_] = e_
    tb_ms = sum(v['ms'] for k_, v in grouped.items() if k_.startswith('sw_traceback'))
    stage_ms = {'prefilter': pf_ms, 'sw_score': sw_ms, 'sw_traceback': tb_ms}
    dom_stage = max(stage_ms.items(), key=lambda kv: kv[1])[0]
    dom = max(per_kernel.items(), key=lambda kv: kv[1]['ms'])[0] if per_kernel else 'none'
    dk = per_kernel.get(dom, dict(achieved=0.0, launches=0, avg_launch_ms=0.0))
    pmc_tot = pmc.get('_prefilter_total') or {}
    roofline = dict(bound='hbm', stage='prefilter', kernel=dom, achieved=dk['achieved'], peak=HBM_PEAK_GBS, unit='GB/s', frac=dk['achieved'] / HBM_PEAK_GBS,
                    traffic=dk.get('traffic'), traffic_source=pmc_src if dk.get('traffic') is not None else None, launches=dk['launches'],
                    traffic_per_query=dk.get('traffic_per_query'), algorithmic_per_query=dk.get('algorithmic_per_query'),
                    # the whole stage, per query: PMC bytes of every prefilter kernel / the queries of the PMC run, against SURVEY.md 8(d)
                    stage_traffic_per_query=pmc_tot.get('bytes_per_query'), stage_algorithmic_per_query=b_pref / max(n_queries, 1),
                    stage_traffic_over_algorithmic=(pmc_tot['bytes_per_query'] / (b_pref / max(n_queries, 1))) if pmc_tot.get('bytes_per_query') and b_pref > 0 else None,
                    # SURVEY.md 8(d) bytes of the timed steps over the WALL time of the timed region (the driver's clock), next to stage_frac,
                    # which divides by the sum of the prefilter kernels' event times (four lanes overlap: that sum exceeds the wall time)
                    wall_achieved=b_pref / dt_max / 1e9 if dt_max > 0 else 0.0, wall_frac=(b_pref / dt_max / 1e9 / HBM_PEAK_GBS) if dt_max > 0 else 0.0,
                    avg_launch_ms=dk['avg_launch_ms'], stage_achieved=b_pref / pf_ms / 1e6 if pf_ms > 0 else 0.0,
                    stage_frac=(b_pref / pf_ms / 1e6 / HBM_PEAK_GBS) if pf_ms > 0 else 0.0, stage_kernel_ms=stage_ms,
                    largest_stage=dom_stage, per_kernel=per_kernel,
                    note='kernel durations are HIP-event timed inside the running pipeline (four streams share the GPU); algorithmic bytes: '
                         'K k-mers x 8 B, H hits x 8 B, table slices once per sub-batch (DESIGN.md 4.3); stage_achieved uses SURVEY.md 8(d)')
    # VALU view of the score pass: lane-instructions of the inner loop per DP cell against the VALU issue ceiling.  Both
    # constants come from measurements kept under profiles/ (tools/valu_peak.py: issue micro-benchmark; PMC: SQ_INSTS_VALU per
    # cell); the literals are the fallback when that file is absent
    valu = dict(instr_per_cell=11.1, peak_lane_instr_per_s=256 * 64 * 2.4e9,
                source='ISA count of sw_score_pk RT=8 (178 per 16 cells); 256 CU x 64 lanes x 2.4 GHz')
    try:
        fn = next(f for f in ('r05i_valu_calibration.json', 'r05_valu_calibration.json', 'r04d_valu_calibration.json', 'r04_valu_calibration.json', 'r03_valu_calibration.json', 'r02_valu_calibration.json') if os.path.exists(os.path.join(ROOT, 'profiles', f)))
        v = json.load(open(os.path.join(ROOT, 'profiles', fn)))
        valu = dict(instr_per_cell=v['instr_per_cell'], peak_lane_instr_per_s=v['peak_lane_instr_per_s'], source='profiles/' + fn)
    except (OSError, ValueError, KeyError, StopIteration):
        pass
    sw_valu = dict(cells_per_s=cells_sw / (sw_ms * 1e-3) if sw_ms > 0 else 0.0, **valu)
    sw_valu['frac'] = sw_valu['cells_per_s'] * sw_valu['instr_per_cell'] / sw_valu['peak_lane_instr_per_s']
    roofline_sw = dict(bound='valu', stage='sw_score', kernel='sw_score_pk', achieved=sw_valu['cells_per_s'] * sw_valu['instr_per_cell'] / 1e12,
                       peak=sw_valu['peak_lane_instr_per_s'] / 1e12, unit='T lane-instr/s', frac=sw_valu['frac'], kernel_ms=sw_ms,
                       hbm_achieved_GBs=b_sw / sw_ms / 1e6 if sw_ms > 0 else 0.0,
                       note='integer DP in VGPR/LDS: priced against the measured VALU issue peak (tools/valu_peak.py), not HBM')
    not_computed = int(cs._raw_stats()[0][14])   # per-query error slots of the prefilter (sd_search: counted, never silent)
    if not_computed:
        raise RuntimeError('%d queries were not computed by the prefilter: this is not a valid record' % not_computed)
    mem = gpu.device_memory()   # target index + sequences + the pipeline's workspaces, after the timed steps
    ws_rep = []   # high-water marks of the persistent workspaces, per pipeline context (the largest keys named)
    for c_ in cs.contexts:
        d_, p_, top_ = c_.workspace_report(4)
        ws_rep.append(dict(device_GB=round(d_ / 1e9, 2), pinned_GB=round(p_ / 1e9, 2), largest={k_: round(v_ / 1e9, 2) for k_, v_ in top_}))
    res = {
        'metric': 'clustersearch throughput (genome-pairs/s; SW GCUPS alongside)',
        'value': pairs_total / dt_max,
        'unit': 'genome-pairs/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': dt_max / args.steps * 1e3,
        'higher_is_better': True,
        'scaling': 'strong' if args.strong else 'weak',
        'vs_baseline': None,
        'dtype': 'int16',
        'data': 'synthetic',
        'config': {'workload_short': 'p%d: %d query proteomes per step vs %d targets, --max-seqs %d' % (P, B, P, max_seqs),
                   'workload': '%d synthetic proteomes x %d proteins (len~300) all-vs-all, clustersearch --search-mode 0 '
                               '--filter-self-match --max-seqs %d; step = %d query proteomes %s vs all %d targets; timed: search + aggregation + '
                               'clusterhits + cluster records (+ the result gather for N > 1), complete in the library\'s result handles when the clock '
                               'stops; the TSV is written after it; results.evalue_pushdown = 1: alignments gated at combinehits\' E-value bound '
                               '(1.2e-6) instead of -e 10, identical cluster hits (parity_check.entries_mismatching)'
                               % (P, args.genes, max_seqs, B, 'in total (strong scaling)' if args.strong else 'per rank', P),
                   'parallelism': 'whole query sets dealt to %d rank(s) by sd_shard_query_sets, target index replicated (%s), '
                                  'final result gather: %s' % (world, index_how, gather_how),
                   'chunk_queries': args.chunk if args.chunk else ('auto: %d' % (2500 if ps.n >= 1000000 else 10000))},
        'roofline': roofline,
        'roofline_sw': roofline_sw,
        # forward cells of the aligned pairs over the score kernels' time (which also holds the reverse pass)
        'sw_gcups': st['cells_fwd'] / sw_ms / 1e6 if sw_ms > 0 else 0.0,
        'sw_gcups_fwd_plus_rev': cells_sw / sw_ms / 1e6 if sw_ms > 0 else 0.0,
        'sw_valu': sw_valu,
        'sw_cells': {'forward': st['cells_fwd'], 'reverse': st['cells_rev'], 'traceback': st['cells_tb']},
        'prefilter': {'queries': n_queries, 'kernel_ms': pf_ms, 'algorithmic_bytes': b_pref,
                      'achieved_GBs': b_pref / pf_ms / 1e6 if pf_ms > 0 else 0.0, 'index_hits': st['index_hits'],
                      'kmers': st['kmers'], 'hits': st['prefilter_hits'], 'queries_per_s': n_queries / dt if dt > 0 else 0.0},
        'kernels': kernels,
        'stage_wall_s': {k_: v_ for k_, v_ in stage.items() if not k_.startswith('cpu_')},
        'host_cpu_s_per_step': round(host_cpu_s / max(1, args.steps), 3),
        # what the alignment lanes copy to the host per step: records + pair indices, and the backtrace pool (run-length text written on
        # the device, sd_sw_set_cigar_pool; the letters were 780 MB per step)
        'align_d2h_MB_per_step': {'records': round((dl1[0] - dl0[0]) / max(1, args.steps) / 1e6, 1), 'pool': round((dl1[1] - dl0[1]) / max(1, args.steps) / 1e6, 1)},
        # where the host CPU of a step goes: thread CPU seconds of the pipeline's stage threads (sd_search), the rest of the process's
        # CPU time (OpenMP workers of the host stages -- composition bias, accept / sort / text, aggregation --, HIP runtime threads)
        'host_cpu_by_stage': dict({k_[4:]: round(v_ / max(1, args.steps), 3) for k_, v_ in stage.items() if k_.startswith('cpu_')},
                                  other_threads=round((host_cpu_s - sum(v_ for k_, v_ in stage.items() if k_.startswith('cpu_'))) / max(1, args.steps), 3)),
        # the same by thread (per step; threads sorted by CPU, the twelve busiest): name as the kernel knows it, count of threads of that
        # name folded into one line when they are a team (OpenMP workers, runtime helpers)
        'host_cpu_threads': _thread_cpu_report(thr0, thr1, max(1, args.steps)),
        'results': {'entries': int(summary[0]), 'matched_hits': int(summary[1]), 'clusters': int(summary[2]),
                    'cluster_hits': int(summary[3]), 'queries_not_computed': not_computed,
                    # 1: the alignments ran with combinehits' E-value bound (1.2e-6) as their gate instead of -e 10 -- pairs that cannot
                    # reach the cluster-hit result stop after the score pass (sd_search.cpp; SD_EVAL_PUSHDOWN=0 switches it off)
                    'evalue_pushdown': int(cs._raw_stats()[0][15])},
        'setup_s': {'generate': t_gen, 'index': cs.timing['index_build_s'], 'index_where': 'device (sd_target_build)', 'search_create': t_index,
                    'upload': cs.timing['upload_s']},
        'device': gpu.device_name(),
        'device_memory': dict(resident_GB=(mem[1] - mem[0]) / 1e9, total_GB=mem[1] / 1e9, free_GB=mem[0] / 1e9, workspaces=ws_rep),
        'host_cores': os.cpu_count(),
        'host_cpu_quota': effective_cpus(),
    }
    if gather_sizes is not None:
        res['gather'] = {'how': gather_how, 'bytes': int(gather_sizes.sum()), 'bytes_per_rank': [int(v) for v in gather_sizes],
                         # the tail of the timed region: from the end of this rank's stream (its records are built) to the root holding all bytes
                         'tail_s': round(gather_s, 3), 'tail_frac_of_timed_region': round(gather_s / dt_max, 4) if dt_max > 0 else None,
                         'records_copy_s': round(float(outs[-1].get('records_copy_s', 0.0)), 3) if outs else None,
                         'payload': 'cluster records (sd_search_result_records): per cluster sets, P-values, members with alignment fields', 'tsv': tsv_info}
        res['multi_gpu_note'] = ('%s scaling over query sets; an 8-GPU curve exists only where the driver ran this command with --gpus 8'
                                 % ('strong (BASELINE configs[2] as written)' if args.strong else 'weak'))
    res['index_check'] = index_check
    q_warm = int(sum(b - a for x in range(args.warmup) for a, b in my_ranges(x)[0]))
    res['prefilter_queries_in_process'] = n_queries + q_warm   # + the isolated leg's two calls, below
    if rank == 0 and not args.no_index_check:
        try:
            res['roofline']['isolated'] = isolated_prefilter(cs, host, ps, k, kmer_thr, max_seqs, int(cs.bin_size),
                                                             n_queries=8192 if P < 5000 else 2048)
            res['prefilter_queries_in_process'] += 2 * int(res['roofline']['isolated']['queries'])   # (a warm call and the measured one)
        except Exception as e:   # (an extra, never the record)
            res['roofline']['isolated'] = dict(error=repr(e)[:200])
        # The dominant kernel.  Inside the pipeline a kernel's event-timed duration is mostly the wait of its workgroups for a CU the score
        # wavefronts of the other streams hold, so "largest in-pipeline time" names whichever kernel queues worst (segment_match: 0.9 ms
        # alone, 10 - 18 ms in the pipeline), not the one that does the stage's work.  The kernel named in `roofline` is therefore the
        # single kernel with the largest time when the stage runs ALONE (the leg above, same run, same device); its achieved / frac stay
        # what the contract asks for -- algorithmic bytes over the event-timed duration inside the timed region -- and its rate alone
        # stands beside them.  Scopes that time several kernels together are listed in per_kernel and not eligible.
        iso_k = (res['roofline'].get('isolated') or {}).get('kernel_ms_by_name') or {}
        pk = res['roofline'].get('per_kernel') or {}
        lumped = ('prefilter_coarse_split', 'prefilter_kmer_partition')
        cand = {k_: v_ for k_, v_ in iso_k.items() if k_ in pk and k_ not in lumped and v_ > 0}
        if cand:
            dom = max(cand.items(), key=lambda kv: kv[1])[0]
            dk = pk[dom]
            q_total = max(int(res['prefilter']['queries']), 1)
            alg_iso = dk['algorithmic_bytes'] * res['roofline']['isolated']['queries'] / q_total
            res['roofline'].update(kernel=dom, achieved=dk['achieved'], frac=dk['achieved'] / HBM_PEAK_GBS, traffic=dk.get('traffic'),
                                   traffic_per_query=dk.get('traffic_per_query'), algorithmic_per_query=dk.get('algorithmic_per_query'),
                                   traffic_over_algorithmic=dk.get('traffic_over_algorithmic'),
                                   traffic_source=res['roofline'].get('traffic_source') if dk.get('traffic') is not None else None,
                                   launches=dk['launches'], avg_launch_ms=dk['avg_launch_ms'],
                                   kernel_alone=dict(ms=cand[dom], achieved=alg_iso / cand[dom] / 1e6, frac=alg_iso / cand[dom] / 1e6 / HBM_PEAK_GBS,
                                                     note='the same kernel in roofline.isolated (nothing else on the device): algorithmic bytes '
                                                          'scaled by queries / its time there'),
                                   dominant_by='largest single kernel of the prefilter stage running alone (roofline.isolated, this run); '
                                               'the largest in-pipeline time is %s' % max(pk.items(), key=lambda kv: kv[1]['ms'])[0])
    last_ranges = my_ranges(args.warmup + args.steps - 1)[0] if args.steps > 0 else []
    extras = dict(ps=ps, k=k, max_seqs=max_seqs, kmer_thr=kmer_thr, bin_size=int(cs.bin_size), gpu=gpu, host=host,
                  last=outs[-1] if outs else None, db=db,
                  # the first query set of the last measured range: its aggregated entries are checked against the reference
                  last_first_set=int(ps.set_id[last_ranges[-1][0]]) if last_ranges else None)
    del cs
    return res, extras


def _leg_allowed(res, name, t_start, args):
    """child records run in the order of their weight for north_star (p1000, p10000, iter3) while the time budget lasts"""
    age = time.time() - t_start
    if age < args.leg_budget:
        return True
    res[name] = dict(skipped='the run was %.0f s old when this record was due (--leg-budget %.0f)' % (age, args.leg_budget))
    return False



def _thread_cpu_report(a, b, steps):
    by = {}
    for tid, (comm, cpu) in b.items():
        d = cpu - a.get(tid, (comm, 0.0))[1]
        if d <= 0:
            continue
        e = by.setdefault(comm, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += d
        e[2] = max(e[2], d)
    rows = sorted(by.items(), key=lambda kv: -kv[1][1])[:12]
    return [dict(name=k, threads=v[0], cpu_s_per_step=round(v[1] / steps, 3), busiest_thread=round(v[2] / steps, 3)) for k, v in rows]


def _thread_cpu_snapshot():
    """{tid: (comm, CPU seconds)} of this process's threads (/proc/self/task/*/stat)"""
    snap = {}
    tck = os.sysconf('SC_CLK_TCK')
    try:
        for tid in os.listdir('/proc/self/task'):
            try:
                raw = open('/proc/self/task/%s/stat' % tid).read()
            except OSError:
                continue
            comm = raw[raw.index('(') + 1:raw.rindex(')')]
            f = raw[raw.rindex(')') + 2:].split()
            snap[int(tid)] = (comm, (int(f[11]) + int(f[12])) / tck)
    except OSError:
        pass
    return snap

_REAL_STDOUT = None


def _emit(text):
    """the one line of this process on the stdout it was started with"""
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (text + '\n').encode())


def main():
    args = parse()
    t_start = time.time()
    # Native libraries print to the process's stdout too (RCCL's version banner at communicator set-up, flushed at exit -- behind the
    # JSON line): file descriptor 1 goes to stderr for the run, and the one JSON line is written to the descriptor the process got.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    dist = None
    # SD_BENCH_FORCE_DIST=1: the N > 1 code path with one rank (communicator, pinned staging, the gather, the TSV from the gathered
    # buffer) -- what a one-GPU box can exercise of it
    if world > 1 or os.environ.get('SD_BENCH_FORCE_DIST') == '1':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(29500 + os.getpid() % 2000))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
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
                     sd=(db.strand[hq] | (db.strand[ht] << 1)).astype(np.uint8), nq=db.set_size[last['entry_q']], pv=last['hit_pval'])
        res['cpu_baseline'] = cpu_baseline_subprocess(args.proteomes, args.genes, ex['max_seqs'], ex['kmer_thr'], ex['bin_size'], ent,
                                                      args.cpu_seconds, args.cpu_threads or effective_cpus(), check=24, check_out=chk,
                                                      check_sets=[ex['last_first_set']] if ex['last_first_set'] is not None else [], check_entries=32)
        try:
            if os.path.exists(chk):
                res['parity_check'] = parity_check(ex['gpu'], ex['host'], ex['ps'], ex['k'], ex['max_seqs'], ex['bin_size'],
                                                   ex['kmer_thr'], chk, last=last, db=db)
            else:
                res['parity_check'] = dict(queries=0, note='the reference library did not travel or the CPU leg failed')
        except Exception as e:
            res['parity_check'] = dict(error=repr(e)[:300])
        for f in (ent, chk):
            if os.path.exists(f):
                os.remove(f)
    del ex
    if args.record:   # child of a leg: the whole record, the parent keeps what it needs
        _emit(json.dumps(res))
        return

    def child(name, cmd, keep=None):
        """a child record in its own process (fresh device memory); its full record goes to the side file, a brief into the line"""
        try:
            t0 = time.time()
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500)
            line = [l for l in p.stdout.splitlines() if l.startswith('{')]
            if p.returncode == 0 and line:
                r = json.loads(line[-1])
                r['leg_wall_s'] = time.time() - t0
                return r
            return dict(error=(p.stderr or p.stdout)[-300:])
        except Exception as e:
            return dict(error=repr(e)[:300])

    children = {}
    kids = world == 1 and not args.no_children
    if kids and not args.no_p100 and args.proteomes != 100 and _leg_allowed(children, 'p100', t_start, args):
        # BASELINE configs[1]: 100 proteomes all-vs-all (one pass = 10 steps of 10 query proteomes), its own CPU sample and parity leg
        children['p100'] = child('p100', [sys.executable, os.path.abspath(__file__), '--record', '--proteomes', '100', '--steps', str(args.p100_steps),
                                          '--warmup', '1', '--batch', '10', '--chunk', str(args.chunk), '--no-children'] +
                                 (['--no-cpu'] if args.no_cpu else []) + ['--cpu-seconds', str(min(args.cpu_seconds, 10.0))])
    if kids and not args.no_p10000 and args.proteomes != 10000 and _leg_allowed(children, 'p10000', t_start, args):
        # BASELINE configs[4]: 10 000 proteomes (3 * 10^7 sequences, 9 * 10^9 residues, k = 7) resident on this one GPU -- generated,
        # indexed on the device, checked on a sample against the host builder, and searched for a short step; --max-seqs 2N = 20 000
        children['p10000'] = child('p10000', [sys.executable, os.path.abspath(__file__), '--record', '--proteomes', '10000', '--steps', str(args.p10000_steps),
                                              '--warmup', '1', '--batch', str(args.p10000_batch), '--chunk', str(args.p10000_chunk), '--no-children', '--no-cpu'])
        # CPU figure and parity of this configuration: the reference's index of 3 * 10^7 sequences takes minutes to build on the host, so
        # they are not re-measured inside this run -- they come from the run of tests/test_gpu_scale.py::test_config5_size_search_...
        # (the same search against the reference classes on the GPU box, 44 queries and 300 alignments, and the seconds those
        # reference calls took) kept under profiles/
        c5 = children['p10000']
        for fn in ('r06_scale_parity_p10000.json', 'r05_scale_parity_p10000.json'):
            try:
                sp = json.load(open(os.path.join(ROOT, 'profiles', fn)))
            except (OSError, ValueError):
                continue
            if isinstance(c5, dict) and 'value' in c5:
                rc_, cores = sp.get('reference_cpu'), effective_cpus()
                if rc_:
                    c5['cpu_baseline'] = dict(value=rc_['genome_pairs_per_s_per_core'] * cores, unit='genome-pairs/s', cores=cores, kind='reference',
                                              sample='profiles/%s: %s; one thread measured, scaled linearly to %d' % (fn, rc_['sample'], cores))
                c5['parity_check'] = dict(source='profiles/%s (tests/test_gpu_scale.py)' % fn,
                                          queries=sp.get('prefilter_queries'), prefilter_rows=sp.get('prefilter_rows'),
                                          prefilter_queries_mismatching=sp.get('prefilter_mismatch'), alignments=sp.get('alignments'),
                                          alignments_mismatching=sp.get('alignment_mismatch'))
            break
    if kids and not args.no_iter3 and _leg_allowed(children, 'iter3', t_start, args):
        # BASELINE configs[3]: `clustersearch --num-iterations 3` (sequence search, two profile searches) against 1 000 target
        # proteomes through the sdgpu binary: the iterations in memory (timed); the module chain over DB files on two query proteomes
        # gives the same TSV and its DBs are checked on sampled queries against the reference classes (tools/iter3_scale.py) -- a child
        children['iter3'] = child('iter3', [sys.executable, os.path.join(ROOT, 'tools', 'iter3_scale.py'), '1000', str(args.iter3_queries), '24'])
        c3, cb = children['iter3'], res.get('cpu_baseline') or {}
        if isinstance(c3, dict) and c3.get('cpu_profile_iterations') and cb.get('queries_per_s'):
            # CPU side of this configuration, reference classes throughout: iteration 0 at the main record's measured rate (queries per
            # second of prefilter + alignment on `cores` threads; its extra --realign pass is not in that figure, so this errs in the
            # CPU's favour), iterations 1 and 2 and both profile computations from the sampled queries of this leg (one thread, scaled
            # linearly to the same number of threads)
            cores = int(cb.get('cores') or 1)
            s_q = cores / cb['queries_per_s'] + c3['cpu_profile_iterations']['seconds_per_query']   # core-seconds per query
            q_per_pair = 3000.0 / 1000.0
            c3['cpu_baseline'] = dict(value=cores / (s_q * q_per_pair), unit='genome-pairs/s', cores=cores, kind='reference',
                                      sample='iteration 0: the main record\'s reference rate (%d threads); iterations 1-2 + result2profile x 2: %d sampled '
                                             'queries through the reference classes on one thread (%.3f s per query), scaled to %d threads'
                                             % (cores, c3['cpu_profile_iterations']['queries'], c3['cpu_profile_iterations']['seconds_per_query'], cores))
            c3['gpu_over_cpu'] = c3['genome_pairs_per_s'] / c3['cpu_baseline']['value']

    # ---- the printed line is compact (the driver parses it); everything bulky goes to the side file
    def _brief(r):
        if not isinstance(r, dict) or 'value' not in r:
            return r if isinstance(r, dict) and ('skipped' in r or 'error' in r) else None
        cb, pc, rf = r.get('cpu_baseline') or {}, r.get('parity_check') or {}, r.get('roofline') or {}
        b = dict(value=round(r['value'], 1), unit=r.get('unit'), ms_per_step=round(r.get('ms_per_step') or 0.0, 1), steps=r.get('steps'),
                 workload=(r.get('config') or {}).get('workload_short'), cpu_value=round(cb['value'], 2) if cb.get('value') else None,
                 cpu_cores=cb.get('cores'), cpu_kind=cb.get('kind'), gpu_over_cpu=round(r['value'] / cb['value'], 1) if cb.get('value') else None,
                 parity={k_: v_ for k_, v_ in pc.items() if k_ not in ('against', 'query_sets', 'reference_seconds')} if pc else None,
                 queries_not_computed=(r.get('results') or {}).get('queries_not_computed'),
                 roofline_frac=round(rf.get('frac') or 0.0, 4), roofline_kernel=rf.get('kernel'), stage_frac=round(rf.get('stage_frac') or 0.0, 4),
                 sw_gcups=round(r.get('sw_gcups') or 0.0), leg_wall_s=round(r.get('leg_wall_s') or 0.0, 1))
        if r.get('device_memory'):
            b['resident_GB'] = round(r['device_memory']['resident_GB'], 1)
        return b

    detail = dict(main=res, **children)
    detail_path = args.detail_out or os.path.join(ROOT, 'gpurun_out', 'bench_detail.json')
    try:
        os.makedirs(os.path.dirname(detail_path), exist_ok=True)
        with open(detail_path, 'w') as f:
            json.dump(detail, f)
    except OSError as e:
        detail_path = 'not written: %r' % (e,)
    rf = dict(res['roofline'])
    iso = rf.get('isolated') or {}
    rf.pop('per_kernel', None)
    rf['isolated'] = {k_: iso.get(k_) for k_ in ('queries', 'kernel_ms', 'stage_achieved', 'stage_frac', 'error') if k_ in iso}
    line = {k_: res[k_] for k_ in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                                   'dtype', 'data', 'config')}
    line['roofline'] = rf
    line['roofline_sw'] = res['roofline_sw']
    for k_ in ('sw_gcups', 'sw_gcups_fwd_plus_rev', 'sw_cells', 'prefilter', 'prefilter_queries_in_process', 'stage_wall_s', 'host_cpu_s_per_step', 'align_d2h_MB_per_step', 'host_cpu_by_stage', 'results',
               'setup_s', 'device', 'host_cores', 'host_cpu_quota', 'gather', 'multi_gpu_note', 'index_check', 'cpu_baseline', 'parity_check'):
        if k_ in res:
            line[k_] = res[k_]
    line['device_memory'] = {k_: round(v_, 1) for k_, v_ in res['device_memory'].items() if k_ != 'workspaces'}
    # the driver reads the tail of stdout: the line stays below 8 KB -- the long notes live in the detail file only
    if isinstance(line.get('cpu_baseline'), dict):
        line['cpu_baseline'] = {k_: v_ for k_, v_ in line['cpu_baseline'].items() if k_ not in ('evalue_gate_note', 'sw_gcups_note')}
    for k_ in ('note', 'dominant_by'):
        if isinstance(line['roofline'].get(k_), str) and len(line['roofline'][k_]) > 90:
            line['roofline'][k_] = line['roofline'][k_][:87] + '...'
    if isinstance(line['roofline'].get('kernel_alone'), dict):
        line['roofline']['kernel_alone'] = {k_: v_ for k_, v_ in line['roofline']['kernel_alone'].items() if k_ != 'note'}
    line['roofline_sw'] = {k_: v_ for k_, v_ in line['roofline_sw'].items() if k_ != 'note'}
    for k_ in ('index_check', 'parity_check'):
        if isinstance(line.get(k_), dict):
            line[k_] = {kk: vv for kk, vv in line[k_].items() if kk != 'against'}
    line['detail'] = detail_path if not os.path.isabs(detail_path) else os.path.relpath(detail_path, ROOT)
    if world == 1:
        for name in ('p100', 'p10000', 'iter3'):
            c_ = children.get(name)
            if name == 'iter3' and isinstance(c_, dict) and 'genome_pairs_per_s' in c_:
                b3 = {k_: c_.get(k_) for k_ in ('query_proteomes', 'target_proteomes', 'how', 'kernel_ms_by_stage') if k_ in c_}
                b3.update(value=round(c_['genome_pairs_per_s'], 1), unit='genome-pairs/s', wall_s=round(c_['wall_s'], 2), leg_wall_s=round(c_.get('leg_wall_s') or 0.0, 1))
                mc, cb3, rf3, pc3 = c_.get('module_chain') or {}, c_.get('cpu_baseline') or {}, c_.get('roofline') or {}, c_.get('parity_check') or {}
                b3['module_chain'] = {k_: (round(v_, 1) if isinstance(v_, float) else v_) for k_, v_ in mc.items()
                                      if k_ in ('query_proteomes', 'genome_pairs_per_s', 'tsv_lines', 'tsv_equals_head_of_in_memory_tsv')}
                if cb3:
                    b3.update(cpu_value=round(cb3['value'], 2), cpu_cores=cb3.get('cores'), cpu_kind=cb3.get('kind'), gpu_over_cpu=round(c_.get('gpu_over_cpu') or 0.0, 1))
                if rf3:
                    b3['roofline'] = dict(bound='hbm', stage='prefilter x 3', frac=round(rf3['frac'], 4), achieved=round(rf3['achieved'], 1), unit='GB/s',
                                          kernel_ms=rf3['kernel_ms'], wall_frac=round(rf3.get('wall_frac') or 0.0, 4))
                b3['parity'] = {k_: v_ for k_, v_ in pc3.items() if k_ not in ('against', 'seconds')}
                line[name] = b3
            else:
                line[name] = _brief(c_)
    if len(json.dumps(line)) > 8000:   # (last resort: the per-stage walls and setup times are in the detail file as well)
        for k_ in ('stage_wall_s', 'setup_s', 'host_cpu_by_stage', 'index_check'):
            if len(json.dumps(line)) > 8000:
                line.pop(k_, None)
    _emit(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
