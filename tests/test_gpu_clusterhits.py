"""GPU parity: batched clusterhits kernel vs the oracle's dense restatement (identical partitions, P-values)."""
import numpy as np
import pytest

from spacedust_amd import api
from oracle.pyoracle import oracle_clusterhits

pytestmark = pytest.mark.gpu


from chgen import entry as _entry, many_entries


def test_clusterhits_matches_oracle(gpu, host, oracle):
    rng = np.random.default_rng(17)
    entries = [_entry(rng, K) for K in (2, 3, 7, 40, 150, 400, 1, 333, 90)]
    off = np.zeros(len(entries) + 1, np.uint64)
    off[1:] = np.cumsum([len(e[0]) for e in entries])
    qp = np.concatenate([e[0] for e in entries])
    tp = np.concatenate([e[1] for e in entries])
    sd = np.concatenate([e[2] for e in entries])
    pv = np.concatenate([e[3] for e in entries])
    nq = np.full(len(entries), 600, np.uint32)
    out = api.clusterhits(gpu, host, off, qp, tp, sd, pv, nq)
    total_clusters = 0
    for p, e in enumerate(entries):
        cof, mo, cs, pco, pmh, nm = oracle_clusterhits(oracle, e[0], e[1], e[2], e[3], 600)
        a, b = int(off[p]), int(off[p + 1])
        assert int(out['n_clusters'][p]) == len(cs), (p, out['n_clusters'][p], len(cs))
        assert (out['cluster_of'][a:b] == cof).all(), p
        n = len(cs)
        assert (out['size'][a:a + n] == cs).all()
        assert (out['pCO'][a:a + n] == pco).all(), (p, out['pCO'][a:a + n], pco)
        assert (out['pMH'][a:a + n] == pmh).all()
        # emission order inside clusters
        w = 0
        for c in range(n):
            members = mo[w:w + cs[c]]
            assert (out['rank'][a + members] == np.arange(cs[c])).all()
            w += cs[c]
        total_clusters += n
    assert total_clusters > 20


def test_clusterhits_matches_reference_functions_on_1500_entries(gpu, host):
    """the batched kernel against the reference's own clusterhits code (oracle/_ref/libsdref_ch.so: R/src/util/ClusterHits.cpp
    compiled where it lies, its merge loop re-driven over flat arrays; travels with the snapshot): partition, printed member
    order, P-value bit patterns of 1 500 synthetic (query set, target set) entries in one sd_clusterhits_batch call"""
    from oracle.pyoracle import ref_ch_available, RefClusterHits
    if not ref_ch_available():
        pytest.skip('oracle/_ref/libsdref_ch.so not present on this box')
    ref = RefClusterHits()
    entries = many_entries(2024, 1500)
    off = np.zeros(len(entries) + 1, np.uint64)
    off[1:] = np.cumsum([len(e[0]) for e in entries])
    nq = np.array([e[4] for e in entries], np.uint32)
    out = api.clusterhits(gpu, host, off, np.concatenate([e[0] for e in entries]), np.concatenate([e[1] for e in entries]),
                          np.concatenate([e[2] for e in entries]), np.concatenate([e[3] for e in entries]), nq)
    total = 0
    for p, e in enumerate(entries):
        rcof, rrank, rcs, rpco, rpmh = ref.entry(e[0], e[1], e[2], e[3], e[4])
        a, b = int(off[p]), int(off[p + 1])
        n = len(rcs)
        assert int(out['n_clusters'][p]) == n, p
        assert (out['cluster_of'][a:b] == rcof).all(), p
        assert (out['size'][a:a + n] == rcs).all()
        assert out['pCO'][a:a + n].tobytes() == rpco.tobytes() and out['pMH'][a:a + n].tobytes() == rpmh.tobytes(), p
        clustered = rcof != 0xFFFFFFFF
        assert (out['rank'][a:b][clustered] == rrank[clustered]).all(), p
        total += n
    assert total > 1000
