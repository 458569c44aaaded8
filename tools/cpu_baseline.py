#!/usr/bin/env python3
"""cpu_baseline leg of bench.py (child process): a bounded sample of the same workload on the host cores.
Prefilter + Smith-Waterman per query protein by the reference's own AVX2 code (oracle/_ref/libsdref.so,
kind "reference") when that library travelled, else by the oracle port; clusterhits entries by the reference's own
functions (oracle/_ref/libsdref_ch.so) or, without them, the oracle's restatement.  Prints one JSON object.

Beside the timing it leaves what bench.py's parity leg compares the measured run with (--check-out, an .npz):
  queries / rows / alns   the reference's prefilter rows and alignments of --check sample queries
  agg_*                   the (query set, target set) entries of the whole query sets --check-sets, from the reference's rows
                          and alignments of EVERY query of those sets (ref_run_query_set) pushed through the independent
                          restatement of besthitbyset / combinehits (oracle/agg_restatement.py)
  ch_*                    the reference's clusterhits functions on the first --check-entries measured entries handed over in
                          --entries: partition, member ranks, sizes and the bit patterns of both P-values"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--proteomes', type=int, required=True)
    ap.add_argument('--genes', type=int, default=3000)
    ap.add_argument('--max-seqs', type=int, default=300)
    ap.add_argument('--kmer-thr', type=int, default=112)
    ap.add_argument('--bin-size', type=int, default=2)
    ap.add_argument('--seconds', type=float, default=15.0)
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--entries', default='')
    ap.add_argument('--check', type=int, default=0, help='also return the reference rows / alignments of this many sample queries')
    ap.add_argument('--check-out', default='')
    ap.add_argument('--gate-seconds', type=float, default=0.0, help='also time the reference with the pushed-down E-value gate (1.2e-6) for this long')
    ap.add_argument('--check-sets', default='', help='comma-separated query sets (proteomes) whose aggregated entries the reference side computes')
    ap.add_argument('--check-entries', type=int, default=32, help='measured entries of --entries the reference clusterhits functions are run on')
    a = ap.parse_args()
    os.environ['OMP_NUM_THREADS'] = str(a.threads)
    from oracle import pyoracle
    from spacedust_amd.synth import make_proteomes, ALPHABET
    ps = make_proteomes(a.proteomes, genes_per_proteome=a.genes, seed=0x5ED0 + 2)
    P, n_threads = ps.n_sets, a.threads
    lut = np.frombuffer(ALPHABET.encode(), np.uint8)
    kind = 'reference' if pyoracle.ref_available() else 'port'
    rng = np.random.default_rng(1)
    sample = rng.choice(ps.n, size=min(ps.n, 8192), replace=False)
    lens = ps.lengths()
    db_res = int(ps.offsets[-1])
    t0 = time.time()
    if kind == 'reference':
        ref = pyoracle.Ref(6)
        blob = lut[ps.residues].tobytes()
        rix = ref.index(blob, ps.offsets, kmer_thr=a.kmer_thr, threads=n_threads)
        max_len = int(lens.max())
    else:
        orc = pyoracle.Oracle(n_threads)
        ot = orc.target(ps.residues, ps.offsets, kmer_thr=a.kmer_thr)
    t_index = time.time() - t0
    if kind == 'reference':
        # the reference's per-query loop bodies driven by OpenMP threads inside libsdref (no Python in the loop)
        import ctypes as C
        L = ref.lib
        L.ref_run_queries.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t,
                                      C.c_int, C.c_double, C.c_size_t, C.c_void_p, C.c_double]
        out = np.zeros(8, np.float64)
        smp = np.ascontiguousarray(sample, np.uint32)
        offs = np.ascontiguousarray(ps.offsets, np.uint64)
        L.ref_run_queries(rix.h, blob, offs.ctypes.data, smp.ctypes.data, len(smp), a.kmer_thr, a.max_seqs, n_threads,
                          a.seconds, db_res, out.ctypes.data, 10.0)
        nq, npairs, ncells, dt = int(out[0]), int(out[1]), float(out[2]), float(out[3])
        sw_thread_s = float(out[4])
        if a.gate_seconds > 0:
            # the same loop with the E-value gate the device pipeline pushes down from combinehits (identical cluster hits): what the
            # reference would do if it were run with -e 1.2e-6
            out2 = np.zeros(8, np.float64)
            L.ref_run_queries(rix.h, blob, offs.ctypes.data, smp.ctypes.data, len(smp), a.kmer_thr, a.max_seqs, n_threads,
                              a.gate_seconds, db_res, out2.ctypes.data, 1.2e-6)
            q_per_s_gate = out2[0] / out2[3] if out2[3] > 0 else 0.0
    else:
        done, pairs, cells = [0] * n_threads, [0] * n_threads, [0] * n_threads
        ready = threading.Barrier(n_threads + 1)
        deadline = [0.0]

        def worker(w):
            ready.wait()
            ready.wait()
            for qi in sample[w::n_threads]:
                if time.time() > deadline[0]:
                    break
                x, y = int(ps.offsets[qi]), int(ps.offsets[qi + 1])
                ids = ot.prefilter(ps.residues[x:y], identity_id=int(qi), max_hits=a.max_seqs, bin_size=a.bin_size,
                                   kmer_thr=a.kmer_thr)[0]
                for t in ids:
                    if float(lens[t]) / float(lens[qi]) < 0.8:
                        continue
                    tx, ty = int(ps.offsets[t]), int(ps.offsets[t + 1])
                    orc.sw_align(ps.residues[x:y], ps.residues[tx:ty], db_res, identity=bool(t == qi))
                    pairs[w] += 1
                    cells[w] += int(lens[qi]) * int(lens[t])
                done[w] += 1

        th = [threading.Thread(target=worker, args=(w,)) for w in range(n_threads)]
        for t in th:
            t.start()
        ready.wait()
        t0 = time.time()
        deadline[0] = t0 + a.seconds
        ready.wait()
        for t in th:
            t.join()
        dt = time.time() - t0
        nq, npairs, ncells = sum(done), sum(pairs), float(sum(cells))
        sw_thread_s = 0.0
    q_per_s = nq / dt if dt > 0 else 0.0
    queries_per_pair = ps.n / float(P * P)   # all-vs-all: P*genes queries serve P*P genome pairs
    ch_per_entry, ch_n = 0.0, 0
    ch_out = {}
    if a.entries and os.path.exists(a.entries):
        g = np.load(a.entries)
        orc2 = pyoracle.Oracle(1)
        refch = pyoracle.RefClusterHits() if pyoracle.ref_ch_available() else None
        n_ent = min(len(g['eo']) - 1, max(a.check_entries, 0))
        pv_all = g['pv'] if 'pv' in g.files else None
        lg = refch.lgamma_table(int(max(g['qp'].max(initial=0), g['tp'].max(initial=0), g['nq'].max(initial=0))) + 4096) if refch is not None and n_ent else None
        cof_l, rk_l, sz_l, pco_l, pmh_l, ncl = [], [], [], [], [], []
        t1 = time.time()
        for e in range(n_ent):
            x0, x1 = int(g['eo'][e]), int(g['eo'][e + 1])
            pv = pv_all[x0:x1] if pv_all is not None else np.full(x1 - x0, 1e-30)
            if refch is not None:   # the reference's own clusterhits functions (oracle/_ref/libsdref_ch.so)
                if x1 - x0 + 8 > len(lg):
                    lg = refch.lgamma_table(x1 - x0 + 4096)
                cof, rk, cs_, pco, pmh = refch.entry(g['qp'][x0:x1], g['tp'][x0:x1], g['sd'][x0:x1], pv, int(g['nq'][e]), lg=lg)
                cof_l.append(cof.copy()); rk_l.append(rk.copy()); sz_l.append(cs_.copy()); pco_l.append(pco.copy()); pmh_l.append(pmh.copy())
                ncl.append(len(cs_))
            else:
                pyoracle.oracle_clusterhits(orc2, g['qp'][x0:x1], g['tp'][x0:x1], g['sd'][x0:x1], pv, int(g['nq'][e]))
            ch_n += 1
        ch_per_entry = (time.time() - t1) / max(ch_n, 1)
        if refch is not None and ch_n:
            ch_out = dict(ch_entries=np.int64(ch_n), ch_ncl=np.array(ncl, np.int64), ch_cof=np.concatenate(cof_l), ch_rank=np.concatenate(rk_l),
                          ch_size=np.concatenate(sz_l) if sz_l else np.zeros(0, np.uint32), ch_pco=np.concatenate(pco_l), ch_pmh=np.concatenate(pmh_l))
    sec_per_pair = (queries_per_pair / q_per_s if q_per_s > 0 else float('inf')) + ch_per_entry / n_threads
    value_gate = None
    if kind == 'reference' and a.gate_seconds > 0 and q_per_s_gate > 0:
        value_gate = 1.0 / (queries_per_pair / q_per_s_gate + ch_per_entry / n_threads)
    # parity sample for bench.py's post-check: the reference's prefilter rows and alignments of a few sample queries
    if a.check > 0 and a.check_out and kind == 'reference':
        rpf = rix.prefilter(max_len + 2, max_hits=a.max_seqs)
        sw = pyoracle.RefSW(ref, max_len + 2, db_res)
        qs, rows, alns = [], [], []
        for qi in sample[:a.check]:
            qi = int(qi)
            seq = blob[int(ps.offsets[qi]):int(ps.offsets[qi + 1])]
            ids, sc, dg, _ = rpf.query(seq, qi)
            keep = (lens[ids].astype(np.float32) / np.float32(lens[qi])) >= np.float32(0.8)   # the writer's coverage pre-filter
            ids, sc, dg = ids[keep], sc[keep], dg[keep]
            qs.append(qi)
            rows.append(np.stack([ids.astype(np.int64), sc.astype(np.int64), dg.astype(np.int64)], 1))
            sw.set_query(seq)
            for t in ids[:12]:
                t = int(t)
                r = sw.align(blob[int(ps.offsets[t]):int(ps.offsets[t + 1])], identity=(t == qi))
                alns.append([qi, t, r['score'], r['qStart'], r['qEnd'], r['tStart'], r['tEnd'], r['btLen'], r['identical'] if r['btLen'] > 0 else 0])
        agg_out = {}
        if a.check_sets:
            # whole query sets through the reference, then the independent restatement of the aggregation modules
            from oracle import agg_restatement
            import ctypes as C
            sets = [int(x) for x in a.check_sets.split(',') if x != '']
            qsel = np.concatenate([np.arange(int(ps.set_start[s_]), int(ps.set_start[s_ + 1])) for s_ in sets]).astype(np.uint32)
            cap = len(qsel) * (a.max_seqs + 1)
            rws = np.zeros((cap, 8), np.float64)
            L = ref.lib
            L.ref_run_query_set.restype = C.c_size_t
            L.ref_run_query_set.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_size_t,
                                            C.c_void_p, C.c_size_t]
            offs = np.ascontiguousarray(ps.offsets, np.uint64)
            t2 = time.time()
            n_rows = L.ref_run_query_set(rix.h, blob, offs.ctypes.data, qsel.ctypes.data, len(qsel), a.kmer_thr, a.max_seqs, n_threads, db_res,
                                         rws.ctypes.data, cap)
            ent = agg_restatement.aggregate(rws[:min(n_rows, cap)], lens, ps.set_id)
            keys = sorted(ent)
            agg_out = dict(agg_sets=np.array(sets, np.int64), agg_keys=np.array(keys, np.int64).reshape(-1, 2),
                           agg_off=np.cumsum([0] + [len(ent[k_]) for k_ in keys]).astype(np.int64),
                           agg_q=np.array([r[0] for k_ in keys for r in ent[k_]], np.int64),
                           agg_t=np.array([r[1] for k_ in keys for r in ent[k_]], np.int64),
                           agg_p=np.array([r[2] for k_ in keys for r in ent[k_]], np.float64),
                           agg_seconds=np.float64(time.time() - t2), agg_alignments=np.int64(n_rows))
        np.savez(a.check_out, queries=np.array(qs, np.int64), row_off=np.cumsum([0] + [len(r) for r in rows]),
                 rows=np.concatenate(rows) if rows else np.zeros((0, 3), np.int64), alns=np.array(alns, np.int64).reshape(-1, 9),
                 **agg_out, **ch_out)
    ch_kind = 'reference' if (a.entries and pyoracle.ref_ch_available()) else 'port'
    sw_wall = sw_thread_s / n_threads if sw_thread_s > 0 else dt   # the Smith-Waterman part of the wall time (thread-seconds / threads)
    print(json.dumps(dict(
        value=1.0 / sec_per_pair if sec_per_pair > 0 else 0.0, unit='genome-pairs/s', cores=n_threads, kind=kind,
        sample='%d query proteins: prefilter + SW (%s) against the full %d-proteome target on %d threads = %d of the box\'s %d logical CPUs '
               '(cgroup quota) in %.1f s, plus %d clusterhits entries (%s, 1 core each); the text glue modules between them are not '
               'run on the CPU side; reference index build %.1f s not included'
               % (nq, 'the reference classes' if kind == 'reference' else 'oracle port', P, n_threads, n_threads, os.cpu_count() or 0, dt, ch_n,
                  'the reference functions' if ch_kind == 'reference' else 'oracle restatement', t_index),
        value_with_pushed_down_evalue_gate=value_gate,
        evalue_gate_note='value: the reference as it runs, every pair aligned to clustersearch\'s -e 10; value_with_pushed_down_evalue_gate: the same '
                         'loop with -e 1.2e-6, the bound the device pipeline takes from combinehits (same cluster hits, the work the GPU side does)',
        queries_per_s=q_per_s, sw_gcups=ncells / sw_wall / 1e9 if sw_wall > 0 else 0.0,
        sw_gcups_per_core=ncells / sw_thread_s / 1e9 if sw_thread_s > 0 else None,
        sw_gcups_note='forward cells of the aligned pairs / the time inside the Smith-Waterman calls alone (thread-seconds / threads)',
        sw_share_of_wall=sw_thread_s / (dt * n_threads) if dt > 0 and sw_thread_s > 0 else None, sw_pairs=npairs,
        clusterhits_s_per_entry_core=ch_per_entry, clusterhits_kind=ch_kind)))


if __name__ == '__main__':
    main()
