"""result2profile on the CPU (BASELINE config 4's inter-iteration step): the product's restatement
(sd_r2p_batch) against the reference's own MultipleAlignment / MsaFilter / PSSMCalculator / Masker classes
(oracle/_ref/libsdref_r2p.so) for every query of the regression input, and the DB side of the `result2profile`,
`subtractdbs` and `mergedbs` modules."""
import gzip
import os

import numpy as np
import pytest

from dbutil import GOLD, write_db, read_db, flat_lines_from_gz, entries_by_first_column, sdgpu, example_fasta


def _load():
    from spacedust_amd import api
    host = api.Host()
    seqs = []
    for f in ('NC_000913.faa', 'NC_000915.faa'):
        cur = []
        for line in gzip.open(os.path.join(GOLD, 'examples', f + '.gz'), 'rt'):
            line = line.rstrip('\n')
            if line.startswith('>'):
                if cur:
                    seqs.append(''.join(cur))
                cur = ['']
            else:
                cur.append(line)
        seqs.append(''.join(cur))
    res, off = host.map_sequences(seqs)
    aln = {}
    for l in flat_lines_from_gz('config1_aln.tsv.gz'):
        w = l.rstrip('\n').split('\t')
        aln.setdefault(int(w[0]), []).append(w)
    return api, seqs, res, off, aln


def _edges(api, aln, queries, same_db=True, e_profile=0.001):
    edge_off, et, eq, ets, bts = [0], [], [], [], []
    for q in queries:
        for w in aln.get(q, []):
            if (same_db and int(w[1]) == q) or not (float(w[4]) < e_profile):
                continue
            et.append(int(w[1]))
            eq.append(int(w[5]))
            ets.append(int(w[8]))
            bts.append(api.uncompress_cigar(w[11]))
        edge_off.append(len(et))
    return edge_off, et, eq, ets, bts


def test_profiles_equal_reference_classes_on_every_query():
    from oracle.pyoracle import ref_r2p_available, RefResult2Profile
    if not ref_r2p_available():
        pytest.skip('oracle/_ref/libsdref_r2p.so not built (needs /root/reference)')
    api, seqs, res, off, aln = _load()
    ref = RefResult2Profile()
    queries = list(range(len(seqs)))
    edge_off, et, eq, ets, bts = _edges(api, aln, queries)
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(seqs[q]) for q in queries])
    mine = api.result2profile(res, qoff, edge_off, et, eq, ets, bts, res, off)
    n_edges = bad = 0
    for x, q in enumerate(queries):
        a, b = edge_off[x], edge_off[x + 1]
        r = ref.profile(seqs[q], [seqs[t] for t in et[a:b]], eq[a:b], ets[a:b], bts[a:b])
        bad += r != mine[int(qoff[x]) * 25:int(qoff[x + 1]) * 25]
        n_edges += b - a
    assert (len(queries), bad) == (5898, 0) and n_edges > 5000
    # other parameter corners on a sample: global weights, no filter, no bias / mask, a profile as the centre
    sample = [q for q in queries if edge_off[q + 1] - edge_off[q] >= 2][:40]
    for kw in (dict(wg=1), dict(filter_msa=0), dict(comp_bias=0, mask_profile=0), dict(max_seq_id=0.5, ndiff=3), dict(qid='0.0,0.3,0.6', cov=0.5)):
        for q in sample:
            a, b = edge_off[q], edge_off[q + 1]
            m = api.result2profile(res[int(off[q]):int(off[q + 1])], [0, len(seqs[q])], [0, b - a], et[a:b], eq[a:b], ets[a:b], bts[a:b],
                                   res, off, **kw)
            assert m == ref.profile(seqs[q], [seqs[t] for t in et[a:b]], eq[a:b], ets[a:b], bts[a:b], **kw), (kw, q)
    for q in sample[:20]:   # iteration >= 2: the centre is the previous profile (its query letters enter the MSA)
        a, b = edge_off[q], edge_off[q + 1]
        prof = mine[int(qoff[q]) * 25:int(qoff[q + 1]) * 25]
        letters = np.frombuffer(prof, np.uint8).reshape(-1, 25)[:, 20].copy()
        m = api.result2profile(letters, [0, len(letters)], [0, b - a], et[a:b], eq[a:b], ets[a:b], bts[a:b], res, off)
        assert m == ref.profile(None, [seqs[t] for t in et[a:b]], eq[a:b], ets[a:b], bts[a:b], centre_profile=prof), q


def test_modules_result2profile_subtractdbs_mergedbs(tmp_path):
    api, seqs, res, off, aln = _load()
    fa = example_fasta(tmp_path)
    g = tmp_path / 'genome'
    sdgpu('createsetdb', fa[0], fa[1], g, tmp_path / 'tmp', '-v', '0')
    lines = flat_lines_from_gz('config1_aln.tsv.gz')
    write_db(str(tmp_path / 'aln'), entries_by_first_column(lines, 5898), 5, splits=3)
    sdgpu('result2profile', g, g, tmp_path / 'aln', tmp_path / 'profile_0', '-e', '0.001', '--e-profile', '0.001', '--pca',
          'substitution:1.100,context:1.400', '--pcb', 'substitution:4.100,context:5.800', '--threads', '4', '-v', '0',
          '--profile-weights-host', '1')   # (no GPU on this side of the suite: the host implementation, asked for explicitly)
    prof = read_db(str(tmp_path / 'profile_0'))
    assert open(tmp_path / 'profile_0.dbtype', 'rb').read() == b'\x02\x00\x00\x00' and len(prof) == 5898
    assert os.path.islink(tmp_path / 'profile_0.lookup') and os.path.islink(tmp_path / 'profile_0_h')
    queries = list(range(0, 5898, 97))
    edge_off, et, eq, ets, bts = _edges(api, aln, queries)
    qoff = np.zeros(len(queries) + 1, np.uint64)
    qoff[1:] = np.cumsum([len(seqs[q]) for q in queries])
    mine = api.result2profile(np.concatenate([res[int(off[q]):int(off[q + 1])] for q in queries]), qoff, edge_off, et, eq, ets, bts, res, off)
    for x, q in enumerate(queries):
        assert prof[q] == mine[int(qoff[x]) * 25:int(qoff[x + 1]) * 25], q
    # subtractdbs: prefilter lines whose target is aligned with E <= 0.001 already are dropped (blastpgp.sh:87)
    write_db(str(tmp_path / 'p'), [(0, b'5\t50\t0\n7\t40\t1\n9\t30\t-2\n'), (1, b'5\t50\t0\n'), (2, b'')], 7)
    write_db(str(tmp_path / 'a'), [(0, b'7\t99\t0.500\t1.000E-10\t0\t9\t10\t0\t9\t10\t10M\n9\t20\t0.300\t5.000E-01\t0\t9\t10\t0\t9\t10\t10M\n'),
                                  (2, b'1\t99\t0.500\t1.000E-10\t0\t9\t10\t0\t9\t10\t10M\n')], 5)
    sdgpu('subtractdbs', tmp_path / 'p', tmp_path / 'a', tmp_path / 's', '--e-profile', '0.001', '-e', '10')
    s = read_db(str(tmp_path / 's'))
    assert s == {0: b'5\t50\t0\n9\t30\t-2\n', 1: b'5\t50\t0\n', 2: b''}
    # mergedbs: per key of the first DB, the entries of the others concatenated in argument order (blastpgp.sh:115-118)
    write_db(str(tmp_path / 'k'), [(0, b'x\n'), (1, b'y\n'), (2, b'z\n')], 0)
    sdgpu('mergedbs', tmp_path / 'k', tmp_path / 'm', tmp_path / 'a', tmp_path / 's')
    m = read_db(str(tmp_path / 'm'))
    assert m[0] == read_db(str(tmp_path / 'a'))[0] + s[0] and m[1] == s[1] and m[2] == read_db(str(tmp_path / 'a'))[2]


def _family(rng, length, rows, div_lo, div_hi):
    """a centre and `rows` homologs made from it (substitutions at a per-row rate, 1 % target gaps, 1 % centre gaps of one to three
    letters, ragged ends, flanks), with the alignment each was made by: centre start, target start, backtrace (M, I = target gap,
    D = centre gap; first and last step a match, as a Smith-Waterman path has them)"""
    letters = list('ACDEFGHIKLMNPQRSTVWY')
    centre = ''.join(rng.choice(letters, length))
    targets, qs, ts, bts = [], [], [], []
    for _ in range(rows):
        div = rng.uniform(div_lo, div_hi)
        a = int(rng.integers(0, max(1, length // 10))) if rng.random() < 0.5 else 0
        b = length - (int(rng.integers(0, max(1, length // 10))) if rng.random() < 0.5 else 0)
        head = ''.join(rng.choice(letters, int(rng.integers(0, 6))))
        t, bt = [head], []
        p = a
        while p < b:
            u = rng.random()
            inner = p != a and p != b - 1
            if inner and u < 0.01:
                n = min(int(rng.integers(1, 4)), b - 1 - p)
                bt.append('I' * n)
                p += n
            elif inner and u < 0.02:
                n = int(rng.integers(1, 4))
                t.append(''.join(rng.choice(letters, n)))
                bt.append('D' * n)
                t.append(centre[p])
                bt.append('M')
                p += 1
            else:
                t.append(centre[p] if rng.random() > div else str(rng.choice(letters)))
                bt.append('M')
                p += 1
        t.append(''.join(rng.choice(letters, int(rng.integers(0, 6)))))
        targets.append(''.join(t))
        qs.append(a)
        ts.append(len(head))
        bts.append(''.join(bt))
    return centre, targets, qs, ts, bts


def test_deep_alignments_equal_reference_classes():
    """what an iteration of a proteome-scale search hands to result2profile -- alignments of hundreds of rows -- where the regression
    input has a handful per query: centres of 64 - 900 residues with 40 - 300 synthetic homologs from near-identical (the diversity filter
    drops almost all of them) to 90 % diverged, through the whole host path (alignment assembly, MsaFilter's identity ladder over
    25-column windows, position-specific weights, pseudo counts, masking) against the reference's classes: identical profile bytes"""
    from oracle.pyoracle import ref_r2p_available, RefResult2Profile
    if not ref_r2p_available():
        pytest.skip('oracle/_ref/libsdref_r2p.so not built (needs /root/reference)')
    from spacedust_amd import api
    host = api.Host(4)
    ref = RefResult2Profile()
    rng = np.random.default_rng(31)
    cases = [(300, 300, 0.1, 0.6), (120, 300, 0.05, 0.3), (600, 250, 0.2, 0.7), (300, 40, 0.1, 0.6), (900, 300, 0.0, 0.2), (64, 300, 0.3, 0.8),
             (300, 299, 0.0, 0.05), (450, 120, 0.4, 0.9)]
    for length, rows, d0, d1 in cases:
        centre, targets, qs, ts, bts = _family(rng, length, rows, d0, d1)
        res, off = host.map_sequences([centre] + targets)
        for kw in ({}, dict(wg=1), dict(max_seq_id=0.5, ndiff=3)) if length <= 300 else ({},):
            mine = api.result2profile(res[:length], [0, length], [0, rows], list(range(1, rows + 1)), qs, ts, bts, res, off, **kw)
            assert mine == ref.profile(centre, targets, qs, ts, bts, **kw), (length, rows, d0, d1, kw)
