"""Thin Python mirror of the C ABI (include/spacedust_gpu.h): Host (CPU stages), Context (one GPU),
SeqSet / Target (data resident in HBM) and the three hot-path calls.  All compute happens in
libsdgpu.so; numpy only carries buffers."""
import ctypes as C

import numpy as np

from . import _lib
from .cpus import effective_cpus
from ._lib import SdError, ptr


def _check(ctx, rc, what):
    if rc != 0:
        msg = ''
        if ctx is not None:
            msg = _lib.load().sd_last_error(ctx).decode(errors='replace')
        raise SdError('%s failed (%d): %s' % (what, rc, msg))


class Host:
    """Host-side stages (sd_host_*): matrices, composition bias, target masking + index, E-values."""

    def __init__(self, threads=0):
        import os
        self.L = _lib.load()
        self.threads = threads or effective_cpus()
        h = C.c_void_p()
        _check(None, self.L.sd_host_create(self.threads, C.byref(h)), 'sd_host_create')
        self.h = h

    def matrix(self, which):
        m = np.zeros(441, np.int8)
        pb = np.zeros(21, np.float64)
        a2n = np.zeros(256, np.uint8)
        self.L.sd_host_matrix(self.h, which, ptr(m), ptr(pb), ptr(a2n))
        return m, pb, a2n

    def map_sequences(self, seqs):
        """list of ASCII protein strings -> (residues uint8, offsets uint64)"""
        lens = np.fromiter((len(s) for s in seqs), np.uint64, len(seqs))
        off = np.zeros(len(seqs) + 1, np.uint64)
        np.cumsum(lens, out=off[1:])
        blob = ''.join(seqs).encode()
        out = np.zeros(len(blob), np.uint8)
        self.L.sd_host_map_sequence(self.h, blob, len(blob), ptr(out))
        return out, off

    def comp_bias(self, residues, offsets, k=6):
        n = len(offsets) - 1
        sw = np.zeros(len(residues), np.int8)
        dg = np.zeros(len(residues), np.int8)
        km = np.zeros(len(residues), np.int16)
        self.L.sd_host_comp_bias(self.h, ptr(residues), ptr(offsets), n, k, ptr(sw), ptr(dg), ptr(km))
        return sw, dg, km

    def build_index(self, residues, offsets, k=6, kmer_thr=112, mask=True, mask_prob=0.9):
        return HostIndex(self, residues, offsets, k, kmer_thr, mask, mask_prob)

    def ext_matrix(self, word_len):
        sp = C.c_void_p()
        ip = C.c_void_p()
        n = C.c_uint32()
        _check(None, self.L.sd_host_ext_matrix(self.h, word_len, C.byref(sp), C.byref(ip), C.byref(n)), 'sd_host_ext_matrix')
        size = n.value
        sc = np.ctypeslib.as_array(C.cast(sp, C.POINTER(C.c_int16)), shape=(size * size,))
        ix = np.ctypeslib.as_array(C.cast(ip, C.POINTER(C.c_uint16)), shape=(size * size,))
        return sc, ix, size

    def kmer_threshold(self, sensitivity, k):
        return self.L.sd_host_kmer_threshold(sensitivity, k)

    def map_profiles(self, data, byte_offsets):
        """sd_host_map_profiles: profile DB entries (25 bytes per position) -> dict(letters, consensus, aln [P,21],
        sorted_score [P,20], sorted_index [P,20], offsets [n+1] in positions)"""
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        bo = np.ascontiguousarray(byte_offsets, np.uint64)
        n = len(bo) - 1
        total = int(bo[-1]) // 25
        out = dict(letters=np.zeros(total, np.uint8), consensus=np.zeros(total, np.uint8), aln=np.zeros((total, 21), np.int8),
                   sorted_score=np.zeros((total, 20), np.int16), sorted_index=np.zeros((total, 20), np.uint8),
                   offsets=np.zeros(n + 1, np.uint64))
        _check(None, self.L.sd_host_map_profiles(ptr(data), ptr(bo), n, ptr(out['letters']), ptr(out['consensus']),
                                                 ptr(out['aln']), ptr(out['sorted_score']), ptr(out['sorted_index']),
                                                 ptr(out['offsets'])), 'sd_host_map_profiles')
        used = int(out['offsets'][-1])   # profiles beyond --max-seq-len positions are cut (Sequence::mapProfile): trim the arrays
        if used < total:
            for k_ in ('letters', 'consensus', 'aln', 'sorted_score', 'sorted_index'):
                out[k_] = np.ascontiguousarray(out[k_][:used])
        return out

    def profile_kmer_threshold(self, sensitivity, k):
        return self.L.sd_host_profile_kmer_threshold(sensitivity, k)

    def auto_kmer_size(self, target_residues):
        return self.L.sd_host_auto_kmer_size(int(target_residues))

    def bin_size(self, db_size, l2=0):
        return self.L.sd_host_bin_size(db_size, l2)

    def lgamma_table(self, n):
        out = np.zeros(n, np.float64)
        self.L.sd_host_lgamma_table(ptr(out), n)
        return out

    def evalue(self, db_residues, score, qlen):
        return self.L.sd_host_evalue(db_residues, float(score), float(qlen))

    def bitscore(self, score):
        return self.L.sd_host_bitscore(float(score))


def uncompress_cigar(c):
    """Matcher::uncompressAlignment (Matcher.cpp:187-201)"""
    out, n = [], 0
    for ch in c:
        if '0' <= ch <= '9':
            n = n * 10 + ord(ch) - 48
        else:
            out.append(ch * (n if n else 1))
            n = 0
    return ''.join(out)


def result2profile(q_letters, q_off, edge_off, edge_t, edge_qstart, edge_tstart, backtraces, t_residues, t_off, **kw):
    """sd_r2p_batch: profiles (25 bytes per position, bytes object) of nQ centre sequences from their alignments"""
    L = _lib.load()
    h = C.c_void_p()
    _check(None, L.sd_r2p_create(C.byref(h)), 'sd_r2p_create')
    p = _lib.R2pParams()
    p.filterMsa, p.filterMinEnable, p.filterMaxSeqId = kw.get('filter_msa', 1), kw.get('filter_min_enable', 0), kw.get('max_seq_id', 0.9)
    p.qid, p.qsc, p.covMSAThr, p.Ndiff = kw.get('qid', '0.0').encode(), kw.get('qsc', -20.0), kw.get('cov', 0.0), kw.get('ndiff', 1000)
    p.pcMode, p.pca, p.pcb, p.wg = 0, kw.get('pca', 1.1), kw.get('pcb', 4.1), kw.get('wg', 0)
    p.compBiasCorr, p.maskProfile, p.maskProb = kw.get('comp_bias', 1), kw.get('mask_profile', 1), kw.get('mask_prob', 0.9)
    q_letters = np.ascontiguousarray(q_letters, np.uint8)
    q_off = np.ascontiguousarray(q_off, np.uint64)
    edge_off = np.ascontiguousarray(edge_off, np.uint64)
    edge_t = np.ascontiguousarray(edge_t, np.uint32)
    qs = np.ascontiguousarray(edge_qstart, np.int32)
    ts = np.ascontiguousarray(edge_tstart, np.int32)
    bt_off = np.zeros(len(backtraces) + 1, np.uint64)
    np.cumsum([len(b) for b in backtraces], out=bt_off[1:])
    pool = np.frombuffer((''.join(backtraces) + ' ').encode(), np.uint8).copy()
    t_residues = np.ascontiguousarray(t_residues, np.uint8)
    t_off = np.ascontiguousarray(t_off, np.uint64)
    out = np.zeros(int(q_off[-1]) * 25 + 1, np.uint8)
    ctx = kw.get('ctx')   # a Context: sd_r2p_batch_device (sequence weights on the GPU); None: the host implementation
    if ctx is not None:
        rc = L.sd_r2p_batch_device(ctx.h, h, C.byref(p), len(q_off) - 1, ptr(q_letters), ptr(q_off), ptr(edge_off), ptr(edge_t), ptr(qs),
                                   ptr(ts), ptr(pool), ptr(bt_off), ptr(t_residues), ptr(t_off), ptr(out), None)
    else:
        rc = L.sd_r2p_batch(h, C.byref(p), len(q_off) - 1, ptr(q_letters), ptr(q_off), ptr(edge_off), ptr(edge_t), ptr(qs), ptr(ts), ptr(pool),
                            ptr(bt_off), ptr(t_residues), ptr(t_off), ptr(out), None)
    L.sd_r2p_destroy(h)
    _check(ctx.h if ctx is not None else None, rc, 'sd_r2p_batch')
    return out[:-1].tobytes()


class HostIndex:
    def __init__(self, host, residues, offsets, k, kmer_thr, mask, mask_prob):
        self.host = host
        self.k = k
        self.kmer_thr = kmer_thr
        self.n = len(offsets) - 1
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        residues = np.ascontiguousarray(residues, np.uint8)
        h = C.c_void_p()
        _check(None, host.L.sd_host_index_build(host.h, ptr(residues), ptr(self.offsets), self.n, k, kmer_thr,
                                                1 if mask else 0, mask_prob, C.byref(h)), 'sd_host_index_build')
        self.h = h
        ts, ne, mk = C.c_uint64(), C.c_uint64(), C.c_uint64()
        host.L.sd_host_index_info(h, C.byref(ts), C.byref(ne), C.byref(mk))
        self.table_size, self.n_entries, self.masked_residues = ts.value, ne.value, mk.value
        po, ps, pp, pm = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        host.L.sd_host_index_arrays(h, C.byref(po), C.byref(ps), C.byref(pp), C.byref(pm))
        self.kmer_offsets = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint32)), shape=(self.table_size + 1,))
        # wide index (>= 2^32 entries): kmer_offsets[i] is relative to block_base[i >> 16]; None otherwise
        pb, nb = C.c_void_p(), C.c_uint64()
        host.L.sd_host_index_block_base(h, C.byref(pb), C.byref(nb))
        self.block_base = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint64)), shape=(nb.value,)) if pb.value else None
        ne1 = max(self.n_entries, 1)
        self.entry_seq = np.ctypeslib.as_array(C.cast(ps, C.POINTER(C.c_uint32)), shape=(ne1,))[:self.n_entries]
        self.entry_pos = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_uint16)), shape=(ne1,))[:self.n_entries]
        tot = max(int(self.offsets[-1]), 1)
        self.masked = np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint8)), shape=(tot,))[:int(self.offsets[-1])]

    def __del__(self):
        try:
            self.host.L.sd_host_index_destroy(self.h)
        except Exception:
            pass


class DeviceArray:
    """a device-resident array (a torch tensor) where the C ABI takes a pointer that may live on the device: ptr(x) yields
    its data pointer, the tensor is kept alive"""

    class _P:
        def __init__(self, t):
            self.t = t

        def data_as(self, _):
            return C.c_void_p(self.t.data_ptr())

    def __init__(self, tensor, n_elements):
        self.tensor, self.n = tensor, int(n_elements)
        self.ctypes = DeviceArray._P(tensor)
        self.nbytes = int(tensor.numel() * tensor.element_size())

    def __len__(self):
        return self.n


class IndexArrays:
    """A target index that exists already as arrays (received from another rank, read from an index file): the attributes
    of HostIndex that Target and ClusterSearch(index=...) read"""

    def __init__(self, k, kmer_thr, seq_offsets, kmer_offsets, entry_seq, entry_pos, masked, masked_residues=0, block_base=None):
        self.k, self.kmer_thr = int(k), int(kmer_thr)
        host = lambda a, dt: a if isinstance(a, DeviceArray) else np.ascontiguousarray(a, dt)   # device arrays pass through
        self.block_base = None if block_base is None else host(block_base, np.uint64)
        self.offsets = np.ascontiguousarray(seq_offsets, np.uint64)
        self.n = len(self.offsets) - 1
        self.kmer_offsets = host(kmer_offsets, np.uint32)
        self.entry_seq = host(entry_seq, np.uint32)
        self.entry_pos = host(entry_pos, np.uint16)
        self.masked = host(masked, np.uint8)
        self.table_size, self.n_entries, self.masked_residues = len(self.kmer_offsets) - 1, len(self.entry_seq), int(masked_residues)


class Context:
    """One GPU (sd_ctx).  Raises SdError when no HIP device is visible -- there is no CPU fallback."""

    def __init__(self, device=0, priority=0):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.sd_ctx_create_prio(device, priority, C.byref(h))
        if rc != 0:
            raise SdError('sd_ctx_create(%d) failed (%d): no usable HIP device; the HIP path has no CPU fallback'
                          % (device, rc))
        self.h = h
        self.device_index = device

    def device_name(self):
        b = C.create_string_buffer(256)
        self.L.sd_device_name(self.h, b, 256)
        return b.value.decode()

    def synchronize(self):
        _check(self.h, self.L.sd_synchronize(self.h), 'sd_synchronize')

    def last_error(self):
        return self.L.sd_last_error(self.h).decode(errors='replace')

    def device_memory(self):
        """(free, total) bytes of this context's device"""
        f, t = C.c_uint64(), C.c_uint64()
        _check(self.h, self.L.sd_device_memory(self.h, C.byref(f), C.byref(t)), 'sd_device_memory')
        return f.value, t.value

    def workspace_report(self, top=8):
        """(device bytes, pinned host bytes, [(key, bytes)] of the largest device workspaces) of this context"""
        d, p = C.c_uint64(), C.c_uint64()
        buf = C.create_string_buffer(1 << 16)
        _check(self.h, self.L.sd_workspace_report(self.h, C.byref(d), C.byref(p), buf, len(buf)), 'sd_workspace_report')
        rows = [ln.rsplit(' ', 1) for ln in buf.value.decode().splitlines() if ln]
        return d.value, p.value, [(k, int(v)) for k, v in rows[:top]]

    def profile(self, on=True):
        self.L.sd_profile_enable(self.h, 1 if on else 0)
        self.L.sd_profile_reset(self.h)

    def profile_report(self):
        b = C.create_string_buffer(4096)
        self.L.sd_profile_names(self.h, b, 4096)
        out = {}
        for name in filter(None, b.value.decode().split(',')):
            ms, cnt = C.c_double(), C.c_uint64()
            self.L.sd_profile_get(self.h, name.encode(), C.byref(ms), C.byref(cnt))
            out[name] = (ms.value, cnt.value)
        return out

    def seqset(self, residues, offsets, sw_bias=None):
        return SeqSet(self, residues, offsets, sw_bias)

    def comp_bias(self, host, residues, offsets, k=6):
        """sd_comp_bias_batch: the (sw int8, diag int8, kmer int16) bias arrays of Host.comp_bias, formed on the device"""
        residues = np.ascontiguousarray(residues, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        sw = np.zeros(len(residues), np.int8)
        dg = np.zeros(len(residues), np.int8)
        km = np.zeros(len(residues), np.int16)
        _check(self.h, self.L.sd_comp_bias_batch(self.h, host.h, ptr(residues), ptr(offsets), len(offsets) - 1, k, ptr(sw), ptr(dg),
                                                 ptr(km)), 'sd_comp_bias_batch')
        return sw, dg, km

    def profileset(self, letters, offsets, aln):
        """profile queries for sw_align (sd_profileset_create): query letters + int8 alignment profile [P, 21]"""
        return SeqSet(self, letters, offsets, None, aln=aln)

    def sw_params(self, matrix, db_residues, sw_mode=2, eval_thr=10.0, cov_mode=2, cov_thr=0.8, gap_open=11,
                  gap_extend=1):
        p = _lib.SwParams()
        p.gapOpen, p.gapExtend = gap_open, gap_extend
        for i in range(441):
            p.matrix[i] = int(matrix[i])
        p.covMode, p.covThr, p.evalThr, p.swMode, p.dbResidues = cov_mode, cov_thr, eval_thr, sw_mode, db_residues
        return p

    def sw_score(self, par, queries, targets, pair_q, pair_t, lanes=32, reverse=False, q_end=None, t_end=None):
        pq = np.ascontiguousarray(pair_q, np.uint32)
        pt = np.ascontiguousarray(pair_t, np.uint32)
        out = np.zeros((len(pq), 3), np.int32)
        qe = np.ascontiguousarray(q_end, np.int32) if q_end is not None else None
        te = np.ascontiguousarray(t_end, np.int32) if t_end is not None else None
        _check(self.h, self.L.sd_sw_score_batch(self.h, C.byref(par), queries.h, targets.h, len(pq), ptr(pq), ptr(pt),
                                                lanes, 1 if reverse else 0, ptr(qe), ptr(te), ptr(out)),
               'sd_sw_score_batch')
        return out

    def sw_align(self, par, queries, targets, pair_q, pair_t, identity=None, bt_cap=None, hostpath=False, reuse=False,
                 compact=False, diag=None):
        """sd_sw_align_batch.  bt_cap: capacity of the backtrace pool (default: a generous guess, grown to the exact
        bound sum(qLen + tLen) if the library reports SD_ENOMEM).  reuse=True hands out views of two alternating
        context-owned buffers instead of fresh arrays (valid until the next-but-one call).  compact=True
        (sd_sw_align_batch_compact) returns (pair indices, their records, pool) for the reportable pairs only."""
        pq = np.ascontiguousarray(pair_q, np.uint32)
        pt = np.ascontiguousarray(pair_t, np.uint32)
        n = len(pq)
        idt = np.ascontiguousarray(identity, np.uint8) if identity is not None else np.zeros(n, np.uint8)
        fn = self.L.sd_sw_align_batch_hostpath if hostpath else self.L.sd_sw_align_batch
        exact_cap = None
        if bt_cap is None:
            bt_cap = max(1 << 20, 96 * n)
        if reuse:
            # one flip per call: a retry after SD_ENOMEM must stay on this call's slot -- the other one still holds the
            # previous call's records and backtraces, which the caller may be reading (pipeline.py aggregates them
            # asynchronously)
            self._flip = 1 - getattr(self, '_flip', 0)
        while True:
            if reuse:
                bufs = self.__dict__.setdefault('_align_bufs', [None, None])
                b = bufs[self._flip]
                if b is None or len(b[0]) < n or len(b[1]) < bt_cap:
                    b = (np.empty(max(n, int(1.25 * (len(b[0]) if b else 0))), _lib.SW_RESULT_DTYPE),
                         np.empty(max(bt_cap, int(1.25 * (len(b[1]) if b else 0))), np.uint8))
                    bufs[self._flip] = b
                res, pool = b[0][:n], b[1]
            else:
                res = np.zeros(n, _lib.SW_RESULT_DTYPE)
                pool = np.zeros(bt_cap, np.uint8)
            used = C.c_uint64()
            if compact:
                if not hasattr(self, '_cidx') or len(self._cidx[0]) < n:
                    self._cidx = [np.empty(int(1.25 * n) + 1, np.uint32), np.empty(int(1.25 * n) + 1, np.uint32)]
                cidx = self._cidx[getattr(self, '_flip', 0)] if reuse else np.empty(n, np.uint32)
                n_out = C.c_uint32()
                if diag is not None:   # the prefilter's diagonals: sd_sw_align_batch_compact_diag (same results)
                    dg = np.ascontiguousarray(diag, np.uint16)
                    rc = self.L.sd_sw_align_batch_compact_diag(self.h, C.byref(par), queries.h, targets.h, n, ptr(pq), ptr(pt), ptr(dg),
                                                               ptr(idt), ptr(cidx), ptr(res), C.byref(n_out), ptr(pool), len(pool),
                                                               C.byref(used))
                else:
                    rc = self.L.sd_sw_align_batch_compact(self.h, C.byref(par), queries.h, targets.h, n, ptr(pq), ptr(pt), ptr(idt),
                                                          ptr(cidx), ptr(res), C.byref(n_out), ptr(pool), len(pool), C.byref(used))
            else:
                rc = fn(self.h, C.byref(par), queries.h, targets.h, n, ptr(pq), ptr(pt), ptr(idt), ptr(res), ptr(pool),
                        len(pool), C.byref(used))
            if rc == _lib.SD_ENOMEM and exact_cap is None:
                ql = (queries.offsets[pq + 1] - queries.offsets[pq]).astype(np.int64)
                tl = (targets.offsets[pt + 1] - targets.offsets[pt]).astype(np.int64)
                exact_cap = bt_cap = (2 if getattr(self, '_cigar_pool', False) else 1) * int((ql + tl).sum()) + 64
                continue
            _check(self.h, rc, 'sd_sw_align_batch')
            if compact:
                return cidx[:n_out.value], res[:n_out.value], pool[:used.value]
            return res, pool[:used.value]

    def set_cigar_pool(self, on=True):
        """sd_sw_set_cigar_pool: the alignment calls return run-length text (Matcher::compressAlignment's output) in the pool;
        a record's text is pool[btOffset : btOffset + (flags >> 8)]"""
        _check(self.h, self.L.sd_sw_set_cigar_pool(self.h, 1 if on else 0), 'sd_sw_set_cigar_pool')
        self._cigar_pool = bool(on)

    def sw_download_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        self.L.sd_sw_download_bytes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def sw_cells(self):
        f, r, t = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.sd_sw_last_cells(self.h, C.byref(f), C.byref(r), C.byref(t))
        return f.value, r.value, t.value

    def __del__(self):
        try:
            self.L.sd_ctx_destroy(self.h)
        except Exception:
            pass


class SeqSet:
    def __init__(self, ctx, residues, offsets, sw_bias, aln=None):
        self.ctx = ctx
        self.residues = np.ascontiguousarray(residues, np.uint8)
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        self.n = len(self.offsets) - 1
        b = np.ascontiguousarray(sw_bias, np.int8) if sw_bias is not None else None
        h = C.c_void_p()
        if aln is not None:
            a = np.ascontiguousarray(aln, np.int8)
            assert a.size == len(self.residues) * 21
            _check(ctx.h, ctx.L.sd_profileset_create(ctx.h, ptr(self.residues), ptr(self.offsets), self.n, ptr(a), C.byref(h)),
                   'sd_profileset_create')
        else:
            _check(ctx.h, ctx.L.sd_seqset_create(ctx.h, ptr(self.residues), ptr(self.offsets), self.n, ptr(b), C.byref(h)),
                   'sd_seqset_create')
        self.h = h

    def __del__(self):
        try:
            self.ctx.L.sd_seqset_destroy(self.h)
        except Exception:
            pass


class Target:
    """Target side of the prefilter resident in HBM (sd_target): k-mer index, masked lookup, 3-mer tables."""

    def __init__(self, ctx, host, index):
        self.ctx, self.host, self.index = ctx, host, index
        s2, i2, _ = host.ext_matrix(2)
        s3, i3, _ = host.ext_matrix(3)
        h = C.c_void_p()
        bb = getattr(index, 'block_base', None)
        _check(ctx.h, ctx.L.sd_target_create_wide(ctx.h, index.k, ptr(index.kmer_offsets), ptr(bb) if bb is not None else None,
                                                  ptr(index.entry_seq), ptr(index.entry_pos), index.n_entries, ptr(index.masked),
                                                  ptr(index.offsets), index.n, ptr(s2), ptr(i2), ptr(s3), ptr(i3),
                                                  C.byref(h)), 'sd_target_create')
        self.h = h
        self.n = index.n

    @classmethod
    def build_on_device(cls, ctx, host, residues, offsets, k=6, kmer_thr=112, mask=True, mask_prob=0.9):
        """sd_target_build: masking, k-mer collection and list construction on the GPU (no host index)"""
        self = cls.__new__(cls)
        self.ctx, self.host, self.index = ctx, host, None
        residues = np.ascontiguousarray(residues, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        s2, i2, _ = host.ext_matrix(2)
        s3, i3, _ = host.ext_matrix(3)
        lr = np.zeros(441, np.float64)
        self_score = np.zeros(21, np.int8)
        _check(None, host.L.sd_host_index_tables(host.h, ptr(lr), ptr(self_score)), 'sd_host_index_tables')
        h = C.c_void_p()
        stats = np.zeros(4, np.uint64)
        _check(ctx.h, ctx.L.sd_target_build(ctx.h, k, kmer_thr, 1 if mask else 0, mask_prob, ptr(residues), ptr(offsets), len(offsets) - 1,
                                            ptr(lr), ptr(self_score), ptr(s2), ptr(i2), ptr(s3), ptr(i3), C.byref(h), ptr(stats)),
               'sd_target_build')
        self.h = h
        self.n = len(offsets) - 1
        self.k = k
        self.total = int(offsets[-1])
        self.build_stats = dict(entries=int(stats[0]), masked_residues=int(stats[1]), passes=int(stats[2]), records=int(stats[3]))
        return self

    def sample_check(self, host, residues, offsets, kmer_thr, runs=24, run_len=400, seed=3, mask=True, mask_prob=0.9):
        """an index too large to download against the host builder on a sample: `runs` runs of `run_len` consecutive sequences are
        masked and indexed on the host (sd_host_index_build of the sample alone: masking and k-mer collection are per sequence);
        every host entry must be in the device index, the device index must hold nothing else for these sequences, and their
        masked residues must agree.  Returns dict(sequences, entries, missing, extra, masked_mismatch)"""
        n = len(offsets) - 1
        rng = np.random.default_rng(seed)
        run_len = min(run_len, n)
        starts = np.unique(rng.integers(0, max(1, n - run_len + 1), runs))
        ids = np.unique(np.concatenate([np.arange(s, s + run_len) for s in starts])).astype(np.uint32)
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)[ids]
        sub_off = np.zeros(len(ids) + 1, np.uint64)
        np.cumsum(lens, out=sub_off[1:])
        sub_res = np.concatenate([residues[int(offsets[i]):int(offsets[i + 1])] for i in ids])
        h = host.build_index(sub_res, sub_off, k=self.k, kmer_thr=kmer_thr, mask=mask, mask_prob=mask_prob)
        list_start = h.kmer_offsets.astype(np.uint64) if h.block_base is None else None
        assert list_start is not None
        ent = np.arange(h.n_entries, dtype=np.uint64)
        kmer = (np.searchsorted(h.kmer_offsets[:-1], ent, side='right') - 1).astype(np.uint32)
        # (empty lists share their start with the next one: side='right' lands on the last list starting at or before the entry,
        #  which is the one that holds it)
        seq = ids[h.entry_seq].astype(np.uint32)
        pos = h.entry_pos.astype(np.uint32)
        missing, in_sample = C.c_uint64(), C.c_uint64()
        _check(self.ctx.h, self.ctx.L.sd_target_sample_check(self.ctx.h, self.h, ptr(kmer), ptr(seq), ptr(pos), len(kmer), ptr(ids), len(ids),
                                                             C.byref(missing), C.byref(in_sample), 0, 0, None), 'sd_target_sample_check')
        bad = 0
        z = np.zeros(1, np.uint32)
        for s in starts:   # (runs may overlap: compare run by run against the host's masked bytes of the same sequences)
            a, b = int(offsets[s]), int(offsets[min(n, s + run_len)])
            buf = np.zeros(b - a, np.uint8)
            m2, i2 = C.c_uint64(), C.c_uint64()
            _check(self.ctx.h, self.ctx.L.sd_target_sample_check(self.ctx.h, self.h, ptr(z), ptr(z), ptr(z), 0, ptr(z), 0, C.byref(m2), C.byref(i2),
                                                                 a, b, ptr(buf)), 'sd_target_sample_check')
            first = int(np.searchsorted(ids, s))
            ha, hb = int(sub_off[first]), int(sub_off[first + min(run_len, n - int(s))])
            bad += int((buf != h.masked[ha:hb]).sum())
        return dict(sequences=int(len(ids)), entries=int(h.n_entries), missing=int(missing.value),
                    extra=int(in_sample.value) - int(h.n_entries) + int(missing.value), masked_mismatch=bad,
                    masked_residues_in_sample=int(h.masked_residues))

    def download(self, entries=True):
        """the device's copy of the index: dict(n_entries, masked, starts [tableSize + 1], entry_seq, entry_pos)"""
        ne, ts = C.c_uint64(), C.c_uint64()
        _check(self.ctx.h, self.ctx.L.sd_target_download(self.ctx.h, self.h, C.byref(ne), C.byref(ts), None, None, None, None), 'sd_target_download')
        total = self.total if getattr(self, 'total', None) is not None else int(self.index.offsets[-1])
        out = dict(n_entries=ne.value, masked=np.zeros(total, np.uint8), starts=np.zeros(ts.value + 1, np.uint64))
        es = np.zeros(ne.value if entries else 0, np.uint32)
        ep = np.zeros(ne.value if entries else 0, np.uint16)
        _check(self.ctx.h, self.ctx.L.sd_target_download(self.ctx.h, self.h, None, None, ptr(out['masked']), ptr(out['starts']),
                                                         ptr(es) if entries else None, ptr(ep) if entries else None), 'sd_target_download')
        out['entry_seq'], out['entry_pos'] = es, ep
        return out

    def __del__(self):
        try:
            self.ctx.L.sd_target_destroy(self.h)
        except Exception:
            pass


def prefilter_params(host, n_targets, kmer_thr=112, max_hits=300, min_diag=15, bin_size=None, cov_mode=2,
                     cov_thr=0.8, k=6):
    p = _lib.PrefilterParams()
    p.kmerSize, p.kmerThr, p.maxHitsPerQuery, p.minDiagScore = k, kmer_thr, max_hits, min_diag
    p.binSize = bin_size if bin_size is not None else host.bin_size(n_targets)
    p.covMode, p.covThr = cov_mode, cov_thr
    m, _, _ = host.matrix(2)
    for i in range(441):
        p.ungappedMatrix[i] = int(m[i])
    return p


def prefilter(ctx, target, par, residues, offsets, kmer_bias, diag_bias, identity_id, want_stats=False):
    """sd_prefilter_batch: returns (hits[nQ, maxHits] structured array, counts[nQ], stats[nQ,4] or None)"""
    residues = np.ascontiguousarray(residues, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    kb = np.ascontiguousarray(kmer_bias, np.int16)
    db = np.ascontiguousarray(diag_bias, np.int8)
    ident = np.ascontiguousarray(identity_id, np.uint32)
    nq = len(offsets) - 1
    hits = np.zeros((nq, par.maxHitsPerQuery), _lib.HIT_DTYPE)
    counts = np.zeros(nq, np.uint32)
    stats = np.zeros((nq, 4), np.uint64) if want_stats else None
    _check(ctx.h, ctx.L.sd_prefilter_batch(ctx.h, target.h, C.byref(par), nq, ptr(residues), ptr(offsets), ptr(kb),
                                           ptr(db), ptr(ident), ptr(hits), ptr(counts), ptr(stats)),
           'sd_prefilter_batch')
    return hits, counts, stats


def prefilter_profile(ctx, target, par, prof, identity_id=None, want_stats=False):
    """sd_prefilter_profile_batch for the profiles of Host.map_profiles (the target index built with kmer_thr=0)"""
    nq = len(prof['offsets']) - 1
    ident = np.ascontiguousarray(identity_id, np.uint32) if identity_id is not None else None
    hits = np.zeros((nq, par.maxHitsPerQuery), _lib.HIT_DTYPE)
    counts = np.zeros(nq, np.uint32)
    stats = np.zeros((nq, 4), np.uint64) if want_stats else None
    _check(ctx.h, ctx.L.sd_prefilter_profile_batch(ctx.h, target.h, C.byref(par), nq, ptr(prof['letters']), ptr(prof['offsets']),
                                                   ptr(prof['sorted_score']), ptr(prof['sorted_index']), ptr(prof['aln']),
                                                   ptr(ident), ptr(hits), ptr(counts), ptr(stats)), 'sd_prefilter_profile_batch')
    return hits, counts, stats


def clusterhits(ctx, host, hit_off, q_pos, t_pos, strands, pval, nq, max_gene_gap=3, cluster_size=2, alpha=1.0,
                p_clu_thr=0.01, p_mh_thr=0.01, lgamma=None):
    """sd_clusterhits_batch over entries [hit_off[p], hit_off[p+1]).  Returns dict of output arrays."""
    hit_off = np.ascontiguousarray(hit_off, np.uint64)
    q_pos = np.ascontiguousarray(q_pos, np.uint32)
    t_pos = np.ascontiguousarray(t_pos, np.uint32)
    strands = np.ascontiguousarray(strands, np.uint8)
    pval = np.ascontiguousarray(pval, np.float64)
    nq = np.ascontiguousarray(nq, np.uint32)
    n_pairs = len(hit_off) - 1
    total = int(hit_off[-1])
    if lgamma is None:
        need = int(max(int(q_pos.max(initial=0)), int(t_pos.max(initial=0)), int(nq.max(initial=0)), total)) + 8
        lgamma = host.lgamma_table(need)
    par = _lib.ChParams(max_gene_gap, cluster_size, alpha, p_clu_thr, p_mh_thr)
    out = dict(cluster_of=np.full(total, 0xFFFFFFFF, np.uint32), rank=np.zeros(total, np.uint32),
               n_clusters=np.zeros(n_pairs, np.uint32), pCO=np.zeros(total, np.float64),
               pMH=np.zeros(total, np.float64), size=np.zeros(total, np.uint32))
    _check(ctx.h, ctx.L.sd_clusterhits_batch(ctx.h, C.byref(par), n_pairs, ptr(hit_off), ptr(q_pos), ptr(t_pos),
                                             ptr(strands), ptr(pval), ptr(nq), ptr(lgamma), len(lgamma),
                                             ptr(out['cluster_of']), ptr(out['rank']), ptr(out['n_clusters']),
                                             ptr(out['pCO']), ptr(out['pMH']), ptr(out['size'])),
           'sd_clusterhits_batch')
    return out
