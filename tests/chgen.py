"""Synthetic clusterhits entries shared by the CPU and GPU clusterhits tests: hits with unique query positions, syntenic
chains in both orientations with jitter, noise, random strands, P-values over the whole range the workflow produces."""
import numpy as np


def entry(rng, K, genome=600, chain_frac=0.6):
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


def many_entries(seed, n, kmax=120):
    """n entries, mostly small (the size distribution of real genome pairs), a few large"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        if i % 97 == 0:
            K = int(rng.integers(200, 500))
        else:
            K = int(min(kmax, 2 + rng.geometric(0.08)))
        genome = int(rng.choice([150, 600, 3000]))
        out.append(entry(rng, min(K, genome), genome=genome, chain_frac=float(rng.uniform(0.2, 0.9))) + (genome,))
    return out
