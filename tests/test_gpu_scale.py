"""BASELINE configs[2]'s data size on one GPU: 1 000 synthetic proteomes (3e6 targets, --max-seqs 2000, BINSIZE chosen
from the DB size, the hit-buffer overflow of the longest queries, coarse split, oversize buckets).  Device prefilter rows
and alignments of a query sample against the REAL reference classes (oracle/_ref/libsdref.so, which travels with the
snapshot) run beside the device on the box's host cores.  ~100 s on an MI355X box (profiles/r02_scale_parity_p1000.log)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config3_size_prefilter_and_alignments_match_real_reference(gpu, host):
    from oracle.pyoracle import ref_available
    if not ref_available():
        pytest.skip('oracle/_ref/libsdref.so not present on this box')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import scale_parity
    lines = []
    r = scale_parity.run(1000, gpu=gpu, host=host, log=lambda *a: lines.append(' '.join(str(x) for x in a)))
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'scale_parity_p1000.json'), 'w') as f:
            json.dump(dict(r, log=lines), f, indent=1)
    except OSError:
        pass
    assert r['targets'] == 3000000
    assert r['prefilter_queries'] == 132 and r['prefilter_rows'] > 200000
    assert r['max_index_hits'] > 2 * 3000000          # the hit-buffer overflow route ran (QueryMatcher.cpp:281-326)
    assert r['prefilter_mismatch'] == 0, lines
    assert r['alignments'] >= 1000 and r['alignment_mismatch'] == 0, lines


def test_config2_size_prefilter_and_alignments_match_real_reference(gpu, host):
    """BASELINE configs[1] at its exact size (100 proteomes, 3 * 10^5 targets, --max-seqs 300, the index built on the device): the
    bench's headline workload, sampled against the real reference classes"""
    from oracle.pyoracle import ref_available
    if not ref_available():
        pytest.skip('oracle/_ref/libsdref.so not present on this box')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import scale_parity
    lines = []
    r = scale_parity.run(100, gpu=gpu, host=host, device_index=True, log=lambda *a: lines.append(' '.join(str(x) for x in a)))
    assert r['targets'] == 300000 and r['max_hits'] == 300 and r['not_computed'] == 0
    assert r['prefilter_queries'] == 132 and r['prefilter_rows'] > 10000
    assert r['prefilter_mismatch'] == 0, lines
    assert r['alignments'] >= 1000 and r['alignment_mismatch'] == 0, lines


def test_config5_size_search_with_max_seqs_2N_matches_real_reference(gpu, host):
    """BASELINE configs[4] as SURVEY 8(d) writes it: the 10 000-proteome target (3 * 10^7 sequences, 9 * 10^9 residues, k = 7, wide
    index of 8.7 * 10^9 entries built ON THE DEVICE and resident in HBM), --max-seqs 2N = 20 000 -- result lists beyond the LDS
    sorter (select_hits_big_kernel), wide stream positions, the coarse split, the hit-buffer overflow of the longest queries.
    Device prefilter rows of the 4 longest + 40 random queries and 300 alignments (coordinates, backtraces, E-values) against the
    real reference classes run on this box's host cores (its own index of the same target: ~3 min on 16 threads)."""
    import time
    from oracle.pyoracle import ref_available
    if not ref_available():
        pytest.skip('oracle/_ref/libsdref.so not present on this box')
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import scale_parity
    lines = []
    t0 = time.time()
    r = scale_parity.run(10000, n_longest=4, n_random=40, aln_queries=10, aln_hits=30, gpu=gpu, host=host, device_index=True,
                         log=lambda *a: lines.append(' '.join(str(x) for x in a)))
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'scale_parity_p10000.json'), 'w') as f:
            json.dump(dict(r, log=lines, seconds=time.time() - t0), f, indent=1)
    except OSError:
        pass
    assert r['targets'] == 30000000 and r['k'] == 7 and r['max_hits'] == 20000 and r['index'] == 'device'
    assert r['index_entries'] > (1 << 32)
    assert r['not_computed'] == 0 and r['prefilter_queries'] == 44
    assert r['max_index_hits'] >= (1 << 24)           # wide stream positions ran
    assert r['prefilter_mismatch'] == 0, lines
    assert r['alignments'] >= 250 and r['alignment_mismatch'] == 0, lines


def test_index_of_more_than_2_32_entries(gpu, host, monkeypatch):
    """the wide index with entry arrays of more than 2^32 elements (what 10 000 proteomes produce): a small index whose
    entries sit behind 2^32 + 5*10^7 padding slots -- every list start has a non-zero high half, the upload and the
    interleaving kernel handle more elements than a dispatch has work-items -- returns the rows of the ordinary index"""
    import numpy as np
    from spacedust_amd import api
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=8, genes_per_proteome=400, n_families=600, seed=12)
    k, thr = 7, host.kmer_threshold(5.7, 7)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets, k=k)
    ident = np.arange(ps.n, dtype=np.uint32)
    a = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=thr)
    par = api.prefilter_params(host, a.n, kmer_thr=thr, max_hits=300, cov_thr=0.0, k=k)
    h0, c0, _ = api.prefilter(gpu, api.Target(gpu, host, a), par, ps.residues, ps.offsets, km_b, dg_b, ident)
    monkeypatch.setenv('SD_INDEX_WIDE', '1')
    b = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=thr)
    monkeypatch.delenv('SD_INDEX_WIDE')
    pad = (1 << 32) + 50000000
    wide = api.IndexArrays(k, thr, ps.offsets, b.kmer_offsets.copy(), np.concatenate([np.zeros(pad, np.uint32), b.entry_seq]),
                           np.concatenate([np.zeros(pad, np.uint16), b.entry_pos]), b.masked.copy(), 0,
                           block_base=b.block_base + np.uint64(pad))
    assert wide.n_entries > (1 << 32)
    h1, c1, _ = api.prefilter(gpu, api.Target(gpu, host, wide), par, ps.residues, ps.offsets, km_b, dg_b, ident)
    assert np.array_equal(c0, c1) and int(c0.sum()) > 2 * ps.n
    for q in range(ps.n):
        assert np.array_equal(h0[q, :int(c0[q])], h1[q, :int(c1[q])]), q


def test_queries_with_2_24_index_hits_and_more():
    """stream positions beyond 24 bits: in a sub-batch that holds a query with >= 2^24 index hits the value word of a hit is
    the whole position and the diagonal byte travels in the key from the coarse split on (tools/scale_cases.py E: 5.2*10^6
    targets, two families of 7.2*10^5 copies).  The query with 1.8*10^7 hits -- one overflow of the reference's hit buffer
    on the way -- and the one with 2.1*10^7 -- two overflows, QueryMatcher.cpp:289-303 -- equal the real reference row for
    row, as do the ten ordinary queries of the same batch."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import scale_cases
    r = scale_cases.run('E', log=lambda *a: None)
    heavy = [x for x, h in enumerate(r['index_hits']) if h >= (1 << 24)]
    assert len(heavy) == 2 and r['mismatching'] == []
    twice = [x for x in heavy if r['index_hits'][x] >= 2 * r['max_db_matches']]
    assert len(twice) == 1 and r['refused'] == []
    assert r['rows'] == 12 * 1000


@pytest.mark.parametrize('P', [1000, 10000])
def test_index_built_on_the_device_at_scale(P):
    """BASELINE configs[2] / configs[4] target sizes: the index of 1 000 (k = 6) and of 10 000 proteomes (3 * 10^7 sequences,
    9 * 10^9 residues, k = 7, more than 2^32 entries: wide form, nine sort passes) built on the GPU by sd_target_build in
    seconds, and checked on 9 600 sample sequences against the host builder: every host entry present, no other entry for these
    sequences, masked residues equal (tools/index_build_scale.py; profiles/r03_index_build_p10000.json is such a run)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import index_build_scale
    r = index_build_scale.run(P, log=lambda *a: None)
    assert r['sequences'] == 3000 * P and r['k'] == (7 if P == 10000 else 6)
    assert r['build']['entries'] > 860000 * P and r['build']['masked_residues'] > 100 * P
    if P == 10000:
        assert r['build']['entries'] > (1 << 32) and r['build']['passes'] >= 5
    chk = r['sample_check']
    assert chk['sequences'] == 9600 and chk['entries'] > 2000000
    assert chk['missing'] == 0 and chk['extra'] == 0 and chk['masked_mismatch'] == 0, chk
    assert r['device_build_s'] < 60


def test_iterative_profile_search_on_1000_proteomes():
    """BASELINE configs[3] at its target size on one GPU: `sdgpu clustersearch Q T --num-iterations 3` with T = 1 000 synthetic
    proteomes (3 * 10^6 proteins) and Q = the first three of them, through createsetdb DBs, in memory; the module chain over DB files
    on the first two gives the same TSV (the head of the in-memory one), and its per-iteration DBs are checked
    against the reference's classes on this machine for 32 sampled queries: profile prefilter rows, profile alignments
    (coordinates, backtraces, E-values) of iterations 1 and 2, profile bytes of profile_0 and profile_1 (tools/iter3_scale.py)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import iter3_scale
    r = iter3_scale.run(1000, 3, 32, log=lambda *a: None)
    assert r['queries'] == 9000 and r['hit_lines'] > 1000 and r['cluster_lines'] > 100
    assert r['files_between_modules'] == [], r['files_between_modules']   # nothing under the in-memory run's tmp directory
    assert r['module_chain']['tsv_equals_head_of_in_memory_tsv'], r['module_chain']
    pc = r['parity_check']
    if pc.get('queries', 0) == 0:
        pytest.skip('oracle/_ref/libsdref*.so not built (needs /root/reference at build time)')
    assert pc['queries'] == 32 and pc['prefilter_rows'] > 1000 and pc['alignments'] > 30 and pc['profiles'] == 64
    assert pc['prefilter_queries_mismatching'] == 0 and pc['alignment_queries_mismatching'] == 0 and pc['profiles_mismatching'] == 0, pc
