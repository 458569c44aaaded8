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
