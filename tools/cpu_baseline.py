#!/usr/bin/env python3
"""cpu_baseline leg of bench.py (child process): a bounded sample of the same workload on the host cores.
Prefilter + Smith-Waterman per query protein by the reference's own AVX2 code (oracle/_ref/libsdref.so,
kind "reference") when that library travelled, else by the oracle port; clusterhits entries by the oracle's
restatement (the reference's clusterhits() is not linkable).  Prints one JSON object."""
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
                                      C.c_int, C.c_double, C.c_size_t, C.c_void_p]
        out = np.zeros(4, np.float64)
        smp = np.ascontiguousarray(sample, np.uint32)
        offs = np.ascontiguousarray(ps.offsets, np.uint64)
        L.ref_run_queries(rix.h, blob, offs.ctypes.data, smp.ctypes.data, len(smp), a.kmer_thr, a.max_seqs, n_threads,
                          a.seconds, db_res, out.ctypes.data)
        nq, npairs, ncells, dt = int(out[0]), int(out[1]), float(out[2]), float(out[3])
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
    q_per_s = nq / dt if dt > 0 else 0.0
    queries_per_pair = ps.n / float(P * P)   # all-vs-all: P*genes queries serve P*P genome pairs
    ch_per_entry, ch_n = 0.0, 0
    if a.entries and os.path.exists(a.entries):
        g = np.load(a.entries)
        orc2 = pyoracle.Oracle(1)
        refch = pyoracle.RefClusterHits() if pyoracle.ref_ch_available() else None
        t1 = time.time()
        for e in range(min(len(g['eo']) - 1, 6)):
            x0, x1 = int(g['eo'][e]), int(g['eo'][e + 1])
            if refch is not None:   # the reference's own clusterhits functions (oracle/_ref/libsdref_ch.so)
                refch.entry(g['qp'][x0:x1], g['tp'][x0:x1], g['sd'][x0:x1], np.full(x1 - x0, 1e-30), int(g['nq'][e]))
            else:
                pyoracle.oracle_clusterhits(orc2, g['qp'][x0:x1], g['tp'][x0:x1], g['sd'][x0:x1], np.full(x1 - x0, 1e-30), int(g['nq'][e]))
            ch_n += 1
        ch_per_entry = (time.time() - t1) / max(ch_n, 1)
    sec_per_pair = (queries_per_pair / q_per_s if q_per_s > 0 else float('inf')) + ch_per_entry / n_threads
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
        np.savez(a.check_out, queries=np.array(qs, np.int64), row_off=np.cumsum([0] + [len(r) for r in rows]),
                 rows=np.concatenate(rows) if rows else np.zeros((0, 3), np.int64), alns=np.array(alns, np.int64).reshape(-1, 9))
    print(json.dumps(dict(
        value=1.0 / sec_per_pair if sec_per_pair > 0 else 0.0, unit='genome-pairs/s', cores=n_threads, kind=kind,
        sample='%d query proteins: prefilter + SW against the full %d-proteome target with %d threads in %.1f s, plus %d '
               'clusterhits entries (oracle restatement, 1 core each); reference index build %.1f s not included'
               % (nq, P, n_threads, dt, ch_n, t_index),
        queries_per_s=q_per_s, sw_gcups=ncells / dt / 1e9 if dt > 0 else 0.0, sw_pairs=npairs,
        clusterhits_s_per_entry_core=ch_per_entry, clusterhits_kind='reference' if (a.entries and pyoracle.ref_ch_available()) else 'port')))


if __name__ == '__main__':
    main()
