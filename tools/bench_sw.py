"""Micro-benchmark of the SW kernels (GCUPS), used while tuning; bench.py is the judged entry point."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacedust_amd.api import Host, Context
from spacedust_amd.synth import make_proteomes

np_ = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ps = make_proteomes(np_, genes_per_proteome=3000, seed=21)
host = Host(); gpu = Context(0)
print(gpu.device_name())
fam = ps.family
order = np.argsort(fam, kind='stable')
pq, pt = [], []
# all homolog pairs across proteomes
f_sorted = fam[order]
start = 0
while start < len(order):
    end = start
    while end < len(order) and f_sorted[end] == f_sorted[start]: end += 1
    if f_sorted[start] >= 0:
        m = order[start:end]
        a, b = np.meshgrid(m, m)
        keep = a != b
        pq.append(a[keep]); pt.append(b[keep])
    start = end
pq = np.concatenate(pq).astype(np.uint32); pt = np.concatenate(pt).astype(np.uint32)
print('pairs', len(pq))
sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
mat, _, _ = host.matrix(0)
ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
par = gpu.sw_params(mat, int(ps.offsets[-1]))
gpu.profile(True)
for lanes in (32, 16):
    gpu.sw_score(par, ss, ss, pq[:1000], pt[:1000], lanes=lanes)
    gpu.profile(True)
    t = time.time(); out = gpu.sw_score(par, ss, ss, pq, pt, lanes=lanes); dt = time.time() - t
    cells = gpu.sw_cells()[0]
    rep = gpu.profile_report()
    print('lanes', lanes, 'cells %.3g' % cells, 'wall %.3f s' % dt, 'kernel %.3f ms' % rep['sw_score'][0], 'GCUPS(kernel) %.1f' % (cells / rep['sw_score'][0] / 1e6), 'launches', rep['sw_score'][1], 'mean score', out[:, 0].mean())
gpu.profile(True)
t = time.time(); res, pool = gpu.sw_align(par, ss, ss, pq, pt); dt = time.time() - t
f, r, tb = gpu.sw_cells()
print('align wall %.3f s' % dt, 'cells fwd %.3g rev %.3g tb %.3g' % (f, r, tb), gpu.profile_report(), 'accepted', int((res['btLen'] > 0).sum()))
