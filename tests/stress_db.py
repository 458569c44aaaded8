import numpy as np
AA = 'ACDEFGHIKLMNPQRSTVWY'
def stress_db(seed=5):
    """exact duplicates (ties everywhere), low-complexity runs and tandem repeats (tantan masking), X-rich and very short
    or long sequences"""
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(24):
        b = ''.join(rng.choice(list(AA), int(rng.integers(40, 500))))
        seqs += [b] * int(rng.integers(2, 6))
        for _ in range(3):
            s = list(b)
            for p in np.nonzero(rng.random(len(s)) < 0.2)[0]:
                s[p] = AA[rng.integers(20)]
            seqs.append(''.join(s))
    unit = ''.join(rng.choice(list(AA), 7))
    seqs += [unit * 30, unit * 12 + ''.join(rng.choice(list(AA), 120)), 'A' * 200, 'AG' * 90, ''.join(rng.choice(list('AGST'), 300))]
    seqs += [''.join(rng.choice(list(AA + 'XXXX'), 250)) for _ in range(4)]
    seqs += [''.join(rng.choice(list(AA), 3000)), ''.join(rng.choice(list(AA), 12)), 'M']
    long_b = ''.join(rng.choice(list(AA), 2200))
    seqs += [long_b, long_b[:1800] + ''.join(rng.choice(list(AA), 300)), long_b[200:]]
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order]
