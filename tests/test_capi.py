"""CPU: the C-ABI library loads, exports every symbol include/spacedust_gpu.h declares, fails loudly without
a GPU, and its host stages agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from spacedust_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, 'include', 'spacedust_gpu.h')).read()
    declared = sorted(set(re.findall(r'\b(sd_[a-z0-9_]+)\s*\(', hdr)))
    L = _lib.load()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert len(declared) >= 38
    assert set(_lib.DECLARED_SYMBOLS) <= set(declared)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    L = _lib.load()
    h = C.c_void_p()
    assert L.sd_ctx_create(0, C.byref(h)) == -1      # SD_ENODEVICE
    with pytest.raises(api.SdError):
        api.Context(0)


def test_host_stages_match_oracle(host, oracle, small_proteomes):
    ps = small_proteomes
    sw, dg, km = host.comp_bias(ps.residues, ps.offsets)
    m0 = oracle.matrix(0)[0]
    for i in range(0, ps.n, 17):
        a, b = int(ps.offsets[i]), int(ps.offsets[i + 1])
        seq = ps.residues[a:b]
        cb0, cb1 = oracle.compbias(0, seq), oracle.compbias(1, seq)
        exp_sw = np.where(cb0 < 0, cb0.astype(np.float64) - 0.5, cb0.astype(np.float64) + 0.5).astype(np.int8)
        assert (sw[a:b] == exp_sw).all()
        q = (cb1 / np.float32(4)).astype(np.float32)
        exp_dg = np.where(cb1 < 0, q.astype(np.float64) - 0.5, q.astype(np.float64) + 0.5).astype(np.float32).astype(np.int8)
        assert (dg[a:b] == exp_dg).all()
    idx = host.build_index(ps.residues, ps.offsets)
    ot = oracle.target(ps.residues, ps.offsets)
    o, es, ep, mk = ot.dump()
    assert idx.n_entries == ot.n_entries and (idx.kmer_offsets == o).all() and (idx.entry_seq == es).all()
    assert (idx.entry_pos == ep).all() and (idx.masked == mk).all()
    assert host.kmer_threshold(5.7, 6) == 112 and host.bin_size(5898, 2 << 20) == 2 and host.bin_size(10 ** 7, 2 << 20) == 8
    lg = host.lgamma_table(10)
    assert abs(lg[5] - np.log(24.0)) < 1e-10 and np.isinf(lg[0])


def test_pair_list_and_cpu_quota():
    import ctypes as C
    from spacedust_amd import _lib
    from spacedust_amd._lib import ptr
    from spacedust_amd.cpus import effective_cpus
    L = _lib.load()
    rng = np.random.default_rng(2)
    nq, w = 37, 11
    hits = np.zeros((nq, w), _lib.HIT_DTYPE)
    hits['seqId'] = rng.integers(0, 1000, (nq, w))
    cnt = rng.integers(0, w + 1, nq).astype(np.uint32)
    n = int(cnt.sum())
    assert L.sd_host_pair_list(ptr(hits), ptr(cnt), nq, w, None, None) == n
    pq, pt = np.empty(n, np.uint32), np.empty(n, np.uint32)
    assert L.sd_host_pair_list(ptr(hits), ptr(cnt), nq, w, ptr(pq), ptr(pt)) == n
    exp_q = np.repeat(np.arange(nq, dtype=np.uint32), cnt)
    exp_t = np.concatenate([hits['seqId'][q, :cnt[q]] for q in range(nq)]).astype(np.uint32)
    assert np.array_equal(pq, exp_q) and np.array_equal(pt, exp_t)
    assert 1 <= effective_cpus() <= (os.cpu_count() or 1)


def test_wide_index_is_the_same_index(monkeypatch):
    """indexes of 2^32 entries and more carry 32-bit list starts relative to a 64-bit base per 65 536 k-mers
    (sd_host_index_block_base); SD_INDEX_WIDE=1 forces that form at any size: base + offset must be the ordinary
    index's absolute starts, entries identical"""
    import numpy as np
    from spacedust_amd import api
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=3, genes_per_proteome=150, n_families=200, seed=5)
    host = api.Host(threads=4)
    for k, thr in ((6, 112),):   # k = 7 (a 5-GB offset table per build) is covered on the GPU box: tests/test_gpu_prefilter.py
        a = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=thr)
        assert a.block_base is None
        monkeypatch.setenv('SD_INDEX_WIDE', '1')
        b = host.build_index(ps.residues, ps.offsets, k=k, kmer_thr=thr)
        monkeypatch.delenv('SD_INDEX_WIDE')
        assert b.block_base is not None and len(b.block_base) == ((b.table_size + 2) >> 16) + 1
        idx = np.arange(b.table_size + 1, dtype=np.int64)
        absolute = b.block_base[idx >> 16] + b.kmer_offsets.astype(np.uint64)
        assert np.array_equal(absolute, a.kmer_offsets.astype(np.uint64))
        assert np.array_equal(a.entry_seq, b.entry_seq) and np.array_equal(a.entry_pos, b.entry_pos)
        assert a.n_entries == b.n_entries == int(absolute[-1])


def test_profiles_beyond_max_seq_len_are_cut():
    """Sequence::mapProfile stops at maxLen (M/src/commons/Sequence.cpp:247-266, --max-seq-len 65 535): a longer profile entry is
    cut there, the entries behind it keep their content"""
    import numpy as np
    from spacedust_amd.api import Host
    host = Host(2)
    rng = np.random.default_rng(5)
    lens = [30, 70000, 12]
    recs = [rng.integers(-20, 40, size=(n, 25)).astype(np.int8) for n in lens]
    for r in recs:
        r[:, 20] = rng.integers(0, 20, size=len(r))
    data = b''.join(r.tobytes() for r in recs)
    bo = np.concatenate([[0], np.cumsum([25 * n for n in lens])]).astype(np.uint64)
    out = host.map_profiles(data, bo)
    assert out['offsets'].tolist() == [0, 30, 30 + 65535, 30 + 65535 + 12]
    assert len(out['letters']) == 30 + 65535 + 12
    a = int(out['offsets'][2])
    assert (out['letters'][a:a + 12] == recs[2][:, 20].astype(np.uint8)).all()
    assert (out['letters'][30:30 + 65535] == recs[1][:65535, 20].astype(np.uint8)).all()
    assert (out['aln'][a:a + 12, :20] == (recs[2][:, :20].astype(np.int32) / 4).astype(np.int8)).all()
