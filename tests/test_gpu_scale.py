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
