"""Deterministic synthetic proteome sets: a pool of ancestral protein families,
proteomes assembled from conserved gene blocks (shuffled, partly inverted) plus genome-specific
singletons, each protein mutated with a per-genome divergence (substitutions + short indels).
Everything is numpy on numeric residues (alphabet ACDEFGHIKLMNPQRSTVWY = 0..19, X = 20)."""
import numpy as np

ALPHABET = 'ACDEFGHIKLMNPQRSTVWYX'
# blosum62 background (M/data/blosum62.out header), 20 standard residues
_BG = np.array([0.07422, 0.02469, 0.05363, 0.05431, 0.04742, 0.07415, 0.02621, 0.06792, 0.05815, 0.09891, 0.02499,
                0.04465, 0.03854, 0.03426, 0.05161, 0.05723, 0.05089, 0.07292, 0.01303, 0.03228])
_BG = _BG / _BG.sum()


class ProteomeSet:
    """residues/offsets: all proteins of all proteomes, proteome-major.
    set_id[p], pos_in_set[p] (gene index along the contig), strand[p] (1 = plus), family[p] (-1 singleton)."""

    def __init__(self, residues, offsets, set_id, pos_in_set, strand, family, n_sets):
        self.residues, self.offsets = residues, offsets
        self.set_id, self.pos_in_set, self.strand, self.family = set_id, pos_in_set, strand, family
        self.n_sets = n_sets
        self.n = len(offsets) - 1
        self.set_size = np.bincount(set_id, minlength=n_sets).astype(np.uint32)
        self.set_start = np.zeros(n_sets + 1, np.int64)
        np.cumsum(self.set_size, out=self.set_start[1:])

    def lengths(self):
        return (self.offsets[1:] - self.offsets[:-1]).astype(np.int64)

    def subset(self, sets):
        """proteins of the given proteomes, renumbered as sets 0..len(sets)-1"""
        keep = np.concatenate([np.arange(self.set_start[s], self.set_start[s + 1]) for s in sets])
        lens = self.lengths()[keep]
        off = np.zeros(len(keep) + 1, np.uint64)
        np.cumsum(lens, out=off[1:])
        res = np.concatenate([self.residues[int(self.offsets[p]):int(self.offsets[p + 1])] for p in keep]) if len(keep) else np.zeros(0, np.uint8)
        remap = {int(s): i for i, s in enumerate(sets)}
        sid = np.array([remap[int(x)] for x in self.set_id[keep]], np.uint32)
        return ProteomeSet(res, off, sid, self.pos_in_set[keep], self.strand[keep], self.family[keep], len(sets))

    def ascii(self, p):
        return ''.join(ALPHABET[x] for x in self.residues[int(self.offsets[p]):int(self.offsets[p + 1])])


def _lengths(rng, n, mean_len):
    l = np.rint(rng.gamma(4.0, mean_len / 4.0, n)).astype(np.int64)
    return np.clip(l, 60, 1500)


def _families(seed, n_families, mean_len):
    rng = np.random.default_rng(seed)
    fam_len = _lengths(rng, n_families, mean_len)
    fam_off = np.zeros(n_families + 1, np.int64)
    np.cumsum(fam_len, out=fam_off[1:])
    fam_res = rng.choice(20, size=int(fam_off[-1]), p=_BG).astype(np.uint8)
    return fam_res, fam_off


def _one_proteome(g, seed, fam_res, fam_off, n_families, n_shared, n_single, mean_len, div_range, indel_rate):
    """proteome g: (residues, protein lengths, strands, families); depends on (seed, g) only"""
    grng = np.random.default_rng([seed, g])
    fams = np.sort(grng.choice(n_families, size=n_shared, replace=False))
    # conserved blocks of 8..24 consecutive (ancestral order) families
    blocks = []
    i = 0
    while i < n_shared:
        bl = int(grng.integers(8, 25))
        blocks.append(fams[i:i + bl])
        i += bl
    order = grng.permutation(len(blocks))
    gene_fam, gene_strand = [], []
    for b in order:
        blk = blocks[b]
        if grng.random() < 0.3:
            blk = blk[::-1]
            st = 0
        else:
            st = 1
        gene_fam.append(blk)
        gene_strand.append(np.full(len(blk), st, np.uint8))
    gene_fam = np.concatenate(gene_fam) if gene_fam else np.zeros(0, np.int64)
    gene_strand = np.concatenate(gene_strand) if gene_strand else np.zeros(0, np.uint8)
    # singletons are spliced in at random gene positions
    ins_at = np.sort(grng.integers(0, len(gene_fam) + 1, n_single))
    gene_fam = np.insert(gene_fam, ins_at, -1)
    gene_strand = np.insert(gene_strand, ins_at, grng.integers(0, 2, n_single).astype(np.uint8))
    # sequences: ancestors (or fresh random proteins) ...
    t_g = grng.uniform(div_range[0], div_range[1])
    src = []
    for f in gene_fam:
        if f >= 0:
            src.append(fam_res[fam_off[f]:fam_off[f + 1]])
        else:
            src.append(grng.choice(20, size=int(_lengths(grng, 1, mean_len)[0]), p=_BG).astype(np.uint8))
    lens = np.fromiter((len(s) for s in src), np.int64, len(src))
    cat = np.concatenate(src)
    is_hom = np.repeat(gene_fam >= 0, lens)
    # ... substitutions with probability t_g per site (homologs only)
    sub = (grng.random(len(cat)) < t_g) & is_hom
    cat = np.where(sub, grng.choice(20, size=len(cat), p=_BG).astype(np.uint8), cat)
    # ... indels: geometric length (mean 3), half insertions / half deletions
    ev = (grng.random(len(cat)) < indel_rate) & is_hom
    ev_pos = np.nonzero(ev)[0]
    ev_len = grng.geometric(1.0 / 3.0, len(ev_pos))
    ev_ins = grng.random(len(ev_pos)) < 0.5
    count = np.ones(len(cat), np.int64)
    prot_end = np.repeat(np.cumsum(lens), lens)
    for p, l, ins in zip(ev_pos, ev_len, ev_ins):
        if ins:
            count[p] += l
        else:
            e = min(p + l, prot_end[p] - 1)   # never delete a whole protein tail past its end
            count[p:e] = 0
    out = np.repeat(cat, count)
    # inserted copies (all but the first of each run) become random residues
    first = np.repeat(np.cumsum(count) - count, count)
    is_ins = np.arange(len(out)) != first
    out = np.where(is_ins, grng.choice(20, size=len(out), p=_BG).astype(np.uint8), out)
    new_len = np.add.reduceat(count, np.concatenate(([0], np.cumsum(lens)[:-1])))
    # guard: proteins must stay >= 30 aa
    assert new_len.min() >= 20, new_len.min()
    return out, new_len, gene_strand, gene_fam.astype(np.int64)


def _worker_main(argv):
    """python -m spacedust_amd.synth <out.npz> <seed> <g0> <g1> <genes> <n_families> <shared> <mean_len> <div0> <div1> <indel>:
    proteomes g0 .. g1-1 into one file (one worker process of make_proteomes)"""
    out, seed, g0, g1, genes, n_families = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5])
    shared, mean_len, div, indel = float(argv[6]), int(argv[7]), (float(argv[8]), float(argv[9])), float(argv[10])
    fam_res, fam_off = _families(seed, n_families, mean_len)
    n_shared = min(int(round(genes * shared)), n_families)
    parts = [_one_proteome(g, seed, fam_res, fam_off, n_families, n_shared, genes - n_shared, mean_len, div, indel) for g in range(g0, g1)]
    np.savez(out, res=np.concatenate([p[0] for p in parts]), lens=np.concatenate([p[1] for p in parts]),
             strand=np.concatenate([p[2] for p in parts]), family=np.concatenate([p[3] for p in parts]),
             count=np.array([len(p[1]) for p in parts], np.int64))


def make_proteomes(n_proteomes, genes_per_proteome=3000, n_families=6000, shared_fraction=0.8, mean_len=300,
                   seed=0x5ED0, div_range=(0.1, 0.6), indel_rate=0.01, workers=None):
    """workers: processes that generate the proteomes (every proteome depends on (seed, its number) only, so the result does not
    depend on it); None = this machine's CPU quota for 32 proteomes and more, 1 below.  The workers are separate interpreters
    (`python -m spacedust_amd.synth ...`), not forks: the caller may hold a HIP context."""
    fam_res, fam_off = _families(seed, n_families, mean_len)
    n_shared = int(round(genes_per_proteome * shared_fraction))
    n_shared = min(n_shared, n_families)
    n_single = genes_per_proteome - n_shared
    if workers is None:
        from .cpus import effective_cpus
        workers = min(effective_cpus(), 32) if n_proteomes >= 32 else 1
    workers = max(1, min(workers, n_proteomes))
    if workers > 1:
        import os
        import shutil
        import subprocess
        import sys
        import tempfile
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        shm = '/dev/shm' if os.path.isdir('/dev/shm') and shutil.disk_usage('/dev/shm').free > 400000 * genes_per_proteome * n_proteomes // 3000 * 4 else None
        tmp = tempfile.mkdtemp(prefix='sd_synth_', dir=shm)
        try:
            n_chunks = min(n_proteomes, workers * 4)
            bounds = [n_proteomes * c // n_chunks for c in range(n_chunks + 1)]
            chunks = [(bounds[c], bounds[c + 1], os.path.join(tmp, 'c%05d.npz' % c)) for c in range(n_chunks)]
            env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
            running, todo, parts = [], list(chunks), {}
            while todo or running:
                while todo and len(running) < workers:
                    g0, g1, path = todo.pop(0)
                    cmd = [sys.executable, '-m', 'spacedust_amd.synth', path, str(seed), str(g0), str(g1), str(genes_per_proteome), str(n_families),
                           repr(shared_fraction), str(mean_len), repr(div_range[0]), repr(div_range[1]), repr(indel_rate)]
                    running.append((subprocess.Popen(cmd, env=env, cwd=root), g0, path))
                p, g0, path = running.pop(0)
                if p.wait() != 0:
                    raise RuntimeError('proteome generator worker failed (%d)' % p.returncode)
                with np.load(path) as z:
                    parts[g0] = {k: z[k] for k in ('res', 'lens', 'strand', 'family', 'count')}
                os.remove(path)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        order = sorted(parts)
        lens = np.concatenate([parts[g]['lens'] for g in order])
        count = np.concatenate([parts[g]['count'] for g in order])
        set_id = np.repeat(np.arange(n_proteomes, dtype=np.uint32), count)
        pos_in_set = np.concatenate([np.arange(c, dtype=np.uint32) for c in count])
        off = np.zeros(len(lens) + 1, np.uint64)
        np.cumsum(lens, out=off[1:])
        return ProteomeSet(np.concatenate([parts[g]['res'] for g in order]), off, set_id, pos_in_set,
                           np.concatenate([parts[g]['strand'] for g in order]), np.concatenate([parts[g]['family'] for g in order]), n_proteomes)
    args = (n_shared, n_single, mean_len, div_range, indel_rate)
    parts = [_one_proteome(g, seed, fam_res, fam_off, n_families, *args) for g in range(n_proteomes)]
    all_res = [p[0] for p in parts]
    all_len = [p[1] for p in parts]
    set_id = [np.full(len(p[1]), g, np.uint32) for g, p in enumerate(parts)]
    pos_in_set = [np.arange(len(p[1]), dtype=np.uint32) for p in parts]
    strand = [p[2] for p in parts]
    family = [p[3] for p in parts]
    lens = np.concatenate(all_len)
    off = np.zeros(len(lens) + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    return ProteomeSet(np.concatenate(all_res), off, np.concatenate(set_id), np.concatenate(pos_in_set),
                       np.concatenate(strand), np.concatenate(family), n_proteomes)


if __name__ == '__main__':
    import sys
    _worker_main(sys.argv[1:])
