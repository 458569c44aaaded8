"""sd_target_build (index construction on the device: tantan masking, k-mer records, per-target dedupe, list order) against the
host-built index (sd_host_index_build, itself pinned entry for entry on the reference's IndexTable / Masker through libsdref in
tests/test_oracle_golden.py, tests/test_oracle_ref.py) -- masked bytes, list starts and every (sequence, position) entry."""
import numpy as np
import pytest

from spacedust_amd import api

pytestmark = pytest.mark.gpu


def _starts(idx):
    st = idx.kmer_offsets.astype(np.uint64)
    if idx.block_base is not None:
        st = st + np.repeat(idx.block_base, 65536)[:len(st)]
    return st


def _same(host_idx, dev):
    assert dev['n_entries'] == host_idx.n_entries
    bad = np.nonzero(dev['masked'] != host_idx.masked)[0]
    assert len(bad) == 0, ('masked bytes differ', len(bad), bad[:5])
    assert np.array_equal(dev['starts'], _starts(host_idx))
    assert np.array_equal(dev['entry_seq'], host_idx.entry_seq)
    assert np.array_equal(dev['entry_pos'], host_idx.entry_pos)


def _repeats(rng, n):
    """sequences tantan masks: tandem repeats of period 1..60 with a few substitutions, between random flanks; lengths from
    1 residue (shorter than a k-mer) over the ramp of the first 50 positions to a few thousand"""
    seqs = []
    for i in range(n):
        L = int(rng.choice([1, 2, 5, 9, 10, 11, 17, 49, 50, 51, 52, 53, 55, 64, 100, 257, 700, 3000]))
        s = rng.integers(0, 20, L).astype(np.uint8)
        if L > 30 and i % 3:
            period = int(rng.integers(1, 61))
            a = int(rng.integers(0, max(1, L // 3)))
            b = min(L, a + int(rng.integers(period * 2, period * 12 + 2)))
            unit = rng.integers(0, 20, period).astype(np.uint8)
            rep = np.tile(unit, (b - a) // period + 1)[:b - a]
            mut = rng.random(b - a) < 0.08
            rep = np.where(mut, rng.integers(0, 20, b - a).astype(np.uint8), rep)
            s[a:b] = rep
        if i % 11 == 0 and L > 3:
            s[rng.integers(0, L, max(1, L // 20))] = 20   # X in the input
        seqs.append(s)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in seqs])
    return np.concatenate(seqs), off


@pytest.mark.parametrize('k', [6, 7])
def test_device_built_index_equals_host_built(gpu, host, small_proteomes, k):
    ps = small_proteomes
    thr = host.kmer_threshold(5.7, k)
    h = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=thr)
    t = api.Target.build_on_device(gpu, host, ps.residues, ps.offsets, k=k, kmer_thr=thr)
    assert t.build_stats['masked_residues'] == h.masked_residues
    _same(h, t.download())


def test_masking_of_repeats_and_sequence_lengths_around_the_ramp(gpu, host):
    rng = np.random.default_rng(5)
    res, off = _repeats(rng, 700)
    h = host.build_index(res, off, k=6, kmer_thr=0)
    assert h.masked_residues > 2000          # the repeats are found
    t = api.Target.build_on_device(gpu, host, res, off, k=6, kmer_thr=0)
    assert t.build_stats['masked_residues'] == h.masked_residues
    _same(h, t.download())
    # without masking, with a threshold that drops k-mers
    h = host.build_index(res, off, k=6, kmer_thr=120, mask=False)
    t = api.Target.build_on_device(gpu, host, res, off, k=6, kmer_thr=120, mask=False)
    _same(h, t.download())


def test_masking_in_slices_against_a_scratch_budget(gpu, host, monkeypatch):
    """the forward posteriors between tantan's two passes live in a scratch of bounded size: with a budget of 300 KB the 700
    sequences run as many slices of wavefronts (down to one wavefront per launch) and give the same mask and index"""
    rng = np.random.default_rng(5)
    res, off = _repeats(rng, 700)
    h = host.build_index(res, off, k=6, kmer_thr=0)
    for budget in ('300000', '1', '40000000'):
        monkeypatch.setenv('SD_INDEX_MASK_BUDGET', budget)
        t = api.Target.build_on_device(gpu, host, res, off, k=6, kmer_thr=0)
        assert t.build_stats['masked_residues'] == h.masked_residues
        _same(h, t.download())


def test_many_passes_and_the_wide_form(gpu, host, small_proteomes, monkeypatch):
    """k-mer ranges of at most 2 000 records per sort pass (dozens of passes), and 32-bit list starts relative to 64-bit block
    bases (what >= 2^32 entries need, forced by SD_INDEX_WIDE): the same index"""
    ps = small_proteomes
    h = host.build_index(ps.residues, ps.offsets, k=6, kmer_thr=112)
    monkeypatch.setenv('SD_INDEX_PASS', '2000')
    monkeypatch.setenv('SD_INDEX_WIDE', '1')
    t = api.Target.build_on_device(gpu, host, ps.residues, ps.offsets, k=6, kmer_thr=112)
    assert t.build_stats['passes'] > 20
    _same(h, t.download())


def test_prefilter_on_the_device_built_target(gpu, host, small_proteomes):
    ps = small_proteomes
    ident = np.arange(ps.n, dtype=np.uint32)
    sw_b, dg_b, km_b = host.comp_bias(ps.residues, ps.offsets)
    h = host.build_index(ps.residues, ps.offsets)
    par = api.prefilter_params(host, ps.n, max_hits=300, cov_thr=0.0)
    a = api.prefilter(gpu, api.Target(gpu, host, h), par, ps.residues, ps.offsets, km_b, dg_b, ident)
    b = api.prefilter(gpu, api.Target.build_on_device(gpu, host, ps.residues, ps.offsets), par, ps.residues, ps.offsets, km_b, dg_b, ident)
    assert np.array_equal(a[1], b[1]) and int(a[1].sum()) > ps.n
    for q in range(ps.n):
        assert np.array_equal(a[0][q, :int(a[1][q])], b[0][q, :int(b[1][q])]), q
