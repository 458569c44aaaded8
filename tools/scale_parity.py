#!/usr/bin/env python3
"""Parity at scale, on the GPU box (needs oracle/_ref/libsdref.so, which travels with the snapshot): device prefilter
and alignments against the REAL reference classes on the P-proteome target (default 1000: 3e6 sequences) -- BINSIZE
as chosen from the DB size, coarse split, oversize buckets, the hit-buffer overflow of the longest queries,
--max-seqs 2P.  Usage: python tools/scale_parity.py [P]   (round 1: 132 queries / 262 880 prefilter rows and 1 200
alignments, 0 mismatches)"""
import numpy as np, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacedust_amd import api
from spacedust_amd.synth import make_proteomes, ALPHABET
from oracle.pyoracle import Ref, RefSW
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.time()
ps = make_proteomes(P, genes_per_proteome=3000, seed=0x5ED0 + 2)
print('generated', ps.n, round(time.time() - t0, 1), flush=True)
host = api.Host(); gpu = api.Context(0)
lens = ps.lengths()
rng = np.random.default_rng(9)
order = np.argsort(-lens)
queries = np.concatenate([order[:12], rng.choice(ps.n, 120, replace=False)]).astype(np.int64)   # the longest (overflow) + random
qoff = np.zeros(len(queries) + 1, np.uint64); qoff[1:] = np.cumsum(lens[queries])
qres = np.concatenate([ps.residues[int(ps.offsets[q]):int(ps.offsets[q + 1])] for q in queries])
sw_b, dg_b, km_b = host.comp_bias(qres, qoff)
t0 = time.time(); idx = host.build_index(ps.residues, ps.offsets); print('index', round(time.time() - t0, 1), flush=True)
tgt = api.Target(gpu, host, idx)
max_hits = max(300, 2 * P)
par = api.prefilter_params(host, idx.n, max_hits=max_hits, cov_thr=0.0, bin_size=None)
print('bin size', par.binSize, flush=True)
hits, cnt, st = api.prefilter(gpu, tgt, par, qres, qoff, km_b, dg_b, queries.astype(np.uint32), want_stats=True)
print('device prefilter done: hits', int(cnt.sum()), 'max index hits/query', int(st[:, 1].max()), flush=True)
lut = np.frombuffer(ALPHABET.encode(), np.uint8)
blob = lut[ps.residues].tobytes()
ref = Ref(6)
t0 = time.time(); rix = ref.index(blob, ps.offsets, threads=16); print('ref index', round(time.time() - t0, 1), flush=True)
rpf = rix.prefilter(int(lens.max()) + 2, max_hits=max_hits)
bad = 0
for x, q in enumerate(queries):
    seq = blob[int(ps.offsets[q]):int(ps.offsets[q + 1])]
    ids, sc, dg, _ = rpf.query(seq, int(q))
    n = int(cnt[x])
    ok = n == len(ids) and (hits[x, :n]['seqId'] == ids).all() and (hits[x, :n]['score'] == sc).all() and (hits[x, :n]['diagonal'] == dg).all()
    if not ok:
        bad += 1
        print('MISMATCH query', q, 'len', lens[q], 'device', n, 'ref', len(ids), flush=True)
print('prefilter queries compared', len(queries), 'mismatching', bad, 'rows', int(cnt.sum()), flush=True)
# alignments of the first 40 hits of 30 queries
mat, _, _ = host.matrix(0)
db = int(ps.offsets[-1])
ts = gpu.seqset(ps.residues, ps.offsets, None)
qs = gpu.seqset(qres, qoff, sw_b)
spar = gpu.sw_params(mat, db)
pq, pt = [], []
for x in range(12, 42):
    for h in range(min(40, int(cnt[x]))):
        pq.append(x); pt.append(int(hits[x, h]['seqId']))
pq = np.array(pq, np.uint32); pt = np.array(pt, np.uint32)
ident = (queries[pq] == pt)
out, pool = gpu.sw_align(spar, qs, ts, pq, pt, identity=ident)
sw = RefSW(ref, int(lens.max()) + 2, db)
badsw = 0; last = -1
for i in range(len(pq)):
    if pq[i] != last:
        q = queries[pq[i]]; sw.set_query(blob[int(ps.offsets[q]):int(ps.offsets[q + 1])]); last = pq[i]
    t = pt[i]
    r = sw.align(blob[int(ps.offsets[t]):int(ps.offsets[t + 1])], identity=bool(ident[i]))
    o = out[i]
    same = int(o['score']) == r['score'] and int(o['qEnd']) == r['qEnd'] and int(o['tEnd']) == r['tEnd'] and int(o['btLen']) == r['btLen']
    if same and r['btLen'] > 0:
        bt = pool[int(o['btOffset']):int(o['btOffset']) + int(o['btLen'])].tobytes().decode()
        same = bt == r['backtrace'] and int(o['qStart']) == r['qStart'] and int(o['tStart']) == r['tStart'] and float(o['evalue']) == r['evalue']
    if not same:
        badsw += 1
print('alignments compared', len(pq), 'mismatching', badsw, flush=True)
