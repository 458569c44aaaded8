"""GPU parity: batched clusterhits kernel vs the oracle's dense restatement (identical partitions, P-values)."""
import numpy as np
import pytest

from spacedust_amd import api
from oracle.pyoracle import oracle_clusterhits

pytestmark = pytest.mark.gpu


def _entry(rng, K, genome=600, chain_frac=0.6):
    """hits with unique query positions: syntenic chains (both orientations), jitter and noise"""
    q = np.sort(rng.choice(genome, size=K, replace=False)).astype(np.uint32)
    t = np.zeros(K, np.uint32)
    i = 0
    while i < K:
        run = int(rng.integers(1, 12))
        if rng.random() < chain_frac:
            start = int(rng.integers(0, genome))
            sign = 1 if rng.random() < 0.5 else -1
            for r in range(min(run, K - i)):
                t[i + r] = (start + sign * int(q[i + r] - q[i]) + int(rng.integers(-1, 2))) % genome
        else:
            t[i:i + run] = rng.integers(0, genome, size=min(run, K - i))
        i += run
    strands = rng.integers(0, 4, size=K).astype(np.uint8)
    pval = 10.0 ** rng.uniform(-60, -6.5, size=K)
    perm = rng.permutation(K)
    return q[perm], t[perm], strands[perm], pval[perm]


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
