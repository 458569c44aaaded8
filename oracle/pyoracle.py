"""ctypes loaders for the two CHECKERS (test infrastructure only):

  Oracle -- oracle/liboracle.so, this repo's scalar restatement (oracle/sd_oracle.cpp)
  Ref    -- oracle/_ref/libsdref.so, the real reference classes built from /root/reference
            by oracle/Makefile (present only where that build has been run)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DATA = '/root/reference/lib/mmseqs/data/'

_vp = C.c_void_p
_u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')


def build(ref=True):
    """(re)build liboracle.so and, when the reference tree exists, _ref/libsdref.so"""
    subprocess.check_call(['make', '-s', '-C', HERE, 'liboracle.so'])
    if ref and os.path.isdir('/root/reference/lib/mmseqs/src'):
        subprocess.check_call(['make', '-s', '-j8', '-C', HERE, '_ref/libsdref.so'])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    def __init__(self, threads=8):
        path = os.path.join(HERE, 'liboracle.so')
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.or_ctx_create.restype = _vp
        L.or_ctx_create.argtypes = [C.c_int]
        L.or_target_create.restype = _vp
        L.or_target_create.argtypes = [_vp, _vp, _vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_double]
        L.or_target_info.restype = C.c_uint64
        L.or_target_info.argtypes = [_vp, _vp, _vp]
        L.or_target_dump.argtypes = [_vp, _vp, _vp, _vp, _vp]
        L.or_target_destroy.argtypes = [_vp]
        L.or_prefilter_query.restype = C.c_int64
        L.or_prefilter_query.argtypes = [_vp, _vp, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32,
                                         C.c_int, _vp, _vp, _vp, _vp]
        L.or_get_matrix.argtypes = [_vp, C.c_int, _vp, _vp, _vp]
        L.or_compbias.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_float, _vp]
        L.or_map_sequence.argtypes = [_vp, C.c_char_p, C.c_size_t, _vp]
        L.or_ext_matrix.restype = C.c_size_t
        L.or_ext_matrix.argtypes = [_vp, C.c_int, _vp, _vp]
        L.or_kmer_list.restype = C.c_size_t
        L.or_kmer_list.argtypes = [_vp, C.c_int, _vp, C.c_int, _vp, C.c_size_t]
        L.or_mask.argtypes = [_vp, _vp, C.c_int, C.c_double]
        L.or_sw_pass.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, _vp]
        L.or_sw_align.restype = C.c_double
        L.or_sw_align.argtypes = [_vp, _vp, C.c_int, _vp, C.c_int, C.c_uint64, C.c_int, C.c_double, C.c_int,
                                  C.c_float, C.c_int, C.c_int, _vp, _vp, C.c_int]
        L.or_sw_align_profile.restype = C.c_double
        L.or_sw_align_profile.argtypes = [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_uint64, C.c_int, C.c_double, C.c_int,
                                          C.c_float, C.c_int, _vp, _vp, C.c_int]
        L.or_prefilter_query_profile.restype = C.c_int64
        L.or_prefilter_query_profile.argtypes = [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                                 C.c_uint32, _vp, _vp, _vp, _vp]
        L.or_map_profile.argtypes = [_vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]
        L.or_profile_kmer_list.restype = C.c_size_t
        L.or_profile_kmer_list.argtypes = [_vp, _vp, C.c_int, C.c_int, _vp, C.c_size_t]
        L.or_banded_traceback.argtypes = [_vp, _vp, _vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int]
        L.or_diag_score.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_uint16]
        L.or_evalue.restype = C.c_double
        L.or_evalue.argtypes = [C.c_uint64, C.c_double, C.c_double]
        L.or_bitscore.restype = C.c_double
        L.or_bitscore.argtypes = [C.c_double]
        if hasattr(L, 'or_clusterhits'):
            L.or_clusterhits.restype = C.c_int
        self.ctx = L.or_ctx_create(threads)

    def matrix(self, which):
        m = np.zeros((21, 21), np.int16)
        pb = np.zeros(21, np.float64)
        a2n = np.zeros(256, np.uint8)
        self.lib.or_get_matrix(self.ctx, which, _ptr(m), _ptr(pb), _ptr(a2n))
        return m, pb, a2n

    def map_sequence(self, s):
        b = s.encode() if isinstance(s, str) else s
        out = np.zeros(len(b), np.uint8)
        self.lib.or_map_sequence(self.ctx, b, len(b), _ptr(out))
        return out

    def compbias(self, which, num, scale=1.0):
        out = np.zeros(len(num), np.float32)
        self.lib.or_compbias(self.ctx, which, _ptr(num), len(num), scale, _ptr(out))
        return out

    def ext_matrix(self, which):
        n = self.lib.or_ext_matrix(self.ctx, which, None, None)
        sc = np.zeros((n, n), np.int16)
        ix = np.zeros((n, n), np.uint16)
        self.lib.or_ext_matrix(self.ctx, which, _ptr(sc), _ptr(ix))
        return sc, ix

    def kmer_list(self, window, thr, k=6, cap=1 << 22):
        out = np.zeros(cap, np.uint32)
        w = np.ascontiguousarray(window, np.uint8)
        n = self.lib.or_kmer_list(self.ctx, k, _ptr(w), thr, _ptr(out), cap)
        return out[:n].copy()

    def map_profile(self, data):
        """Sequence::mapProfile: 25-byte records -> (letters, consensus, aln [L][21], sorted scores [L][20], order [L][20])"""
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        L = len(data) // 25
        letters, cons = np.zeros(L, np.uint8), np.zeros(L, np.uint8)
        aln = np.zeros((L, 21), np.int8)
        sc = np.zeros((L, 20), np.int16)
        ix = np.zeros((L, 20), np.uint8)
        self.lib.or_map_profile(_ptr(data), L, _ptr(letters), _ptr(cons), _ptr(aln), _ptr(sc), _ptr(ix))
        return letters, cons, aln, sc, ix

    def profile_kmer_list(self, sorted_score, sorted_index, thr, cap=1 << 22):
        sc = np.ascontiguousarray(sorted_score, np.int16)
        ix = np.ascontiguousarray(sorted_index, np.uint8)
        out = np.zeros(cap, np.uint32)
        n = self.lib.or_profile_kmer_list(_ptr(sc), _ptr(ix), sc.shape[0], thr, _ptr(out), cap)
        return out[:n].copy()

    def sw_align_profile(self, letters, aln, t, db_residues, sw_mode=2, eval_thr=10.0, cov_mode=2, cov_thr=0.8, identity=False):
        letters = np.ascontiguousarray(letters, np.uint8)
        aln = np.ascontiguousarray(aln, np.int8)
        t = np.ascontiguousarray(t, np.uint8)
        out = np.zeros(8, np.int32)
        cap = len(letters) + len(t) + 8
        bt = C.create_string_buffer(cap)
        ev = self.lib.or_sw_align_profile(self.ctx, _ptr(letters), _ptr(aln), len(letters), _ptr(t), len(t), db_residues,
                                          sw_mode, eval_thr, cov_mode, cov_thr, 1 if identity else 0, _ptr(out), bt, cap)
        return dict(score=int(out[0]), qStart=int(out[1]), qEnd=int(out[2]), tStart=int(out[3]), tEnd=int(out[4]),
                    identical=int(out[5]), btLen=int(out[6]), flags=int(out[7]), evalue=ev, backtrace=bt.value.decode())

    def mask(self, num, prob=0.9):
        a = np.array(num, np.uint8, copy=True)
        n = self.lib.or_mask(self.ctx, _ptr(a), len(a), prob)
        return a, n

    def target(self, seqs, offsets, k=6, kmer_thr=112, mask=True, mask_prob=0.9):
        return OracleTarget(self, seqs, offsets, k, kmer_thr, mask, mask_prob)

    def sw_pass(self, prof, t, lanes, direction=0, go=11, ge=1, terminate=0, bias=0, byte_mode=False):
        prof = np.ascontiguousarray(prof, np.int16)
        n = prof.shape[1]
        t = np.ascontiguousarray(t, np.uint8)
        out = np.zeros(3, np.int32)
        self.lib.or_sw_pass(_ptr(prof), n, _ptr(t), len(t), lanes, direction, go, ge, terminate, bias,
                            1 if byte_mode else 0, _ptr(out))
        return tuple(int(x) for x in out)

    def sw_align(self, q, t, db_residues, sw_mode=2, eval_thr=10.0, cov_mode=2, cov_thr=0.8, comp_bias=True,
                 identity=False):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        out = np.zeros(8, np.int32)
        cap = len(q) + len(t) + 8
        bt = C.create_string_buffer(cap)
        ev = self.lib.or_sw_align(self.ctx, _ptr(q), len(q), _ptr(t), len(t), db_residues, sw_mode, eval_thr,
                                  cov_mode, cov_thr, 1 if comp_bias else 0, 1 if identity else 0, _ptr(out), bt, cap)
        return dict(score=int(out[0]), qStart=int(out[1]), qEnd=int(out[2]), tStart=int(out[3]), tEnd=int(out[4]),
                    identical=int(out[5]), btLen=int(out[6]), flags=int(out[7]), evalue=ev,
                    backtrace=bt.value.decode())

    def evalue(self, db_residues, score, qlen):
        return self.lib.or_evalue(db_residues, float(score), float(qlen))

    def bitscore(self, score):
        return self.lib.or_bitscore(float(score))


class OracleTarget:
    def __init__(self, orc, seqs, offsets, k, kmer_thr, mask, mask_prob):
        self.orc = orc
        self.seqs = np.ascontiguousarray(seqs, np.uint8)
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        self.n = len(self.offsets) - 1
        self.h = orc.lib.or_target_create(orc.ctx, _ptr(self.seqs), _ptr(self.offsets), self.n, k, kmer_thr,
                                          1 if mask else 0, mask_prob)
        ts = C.c_uint64()
        mk = C.c_uint64()
        self.n_entries = orc.lib.or_target_info(self.h, C.byref(ts), C.byref(mk))
        self.table_size = ts.value
        self.masked_residues = mk.value

    def dump(self):
        off = np.zeros(self.table_size + 1, np.uint32)
        es = np.zeros(self.n_entries, np.uint32)
        ep = np.zeros(self.n_entries, np.uint16)
        mk = np.zeros(len(self.seqs), np.uint8)
        self.orc.lib.or_target_dump(self.h, _ptr(off), _ptr(es), _ptr(ep), _ptr(mk))
        return off, es, ep, mk

    def prefilter(self, q, identity_id=0xFFFFFFFF, kmer_thr=112, max_hits=300, min_diag=15, bin_size=2,
                  comp_bias=True):
        q = np.ascontiguousarray(q, np.uint8)
        cap = max(max_hits, 1) + 1
        ids = np.zeros(cap, np.uint32)
        sc = np.zeros(cap, np.int32)
        dg = np.zeros(cap, np.uint16)
        st = np.zeros(4, np.uint64)
        n = self.orc.lib.or_prefilter_query(self.h, _ptr(q), len(q), identity_id, kmer_thr, max_hits, min_diag,
                                            bin_size, 1 if comp_bias else 0, _ptr(ids), _ptr(sc), _ptr(dg), _ptr(st))
        if n < 0:
            raise RuntimeError('oracle prefilter path not restated: code %d' % n)
        return ids[:n].copy(), sc[:n].copy(), dg[:n].copy(), st

    def prefilter_profile(self, letters, aln, sorted_score, sorted_index, kmer_thr, max_hits=300, min_diag=15, bin_size=2,
                          identity_id=0xFFFFFFFF):
        letters = np.ascontiguousarray(letters, np.uint8)
        aln = np.ascontiguousarray(aln, np.int8)
        ssc = np.ascontiguousarray(sorted_score, np.int16)
        six = np.ascontiguousarray(sorted_index, np.uint8)
        cap = max(max_hits, 1) + 1
        ids = np.zeros(cap, np.uint32)
        sc = np.zeros(cap, np.int32)
        dg = np.zeros(cap, np.uint16)
        st = np.zeros(4, np.uint64)
        n = self.orc.lib.or_prefilter_query_profile(self.h, _ptr(letters), _ptr(aln), _ptr(ssc), _ptr(six), len(letters),
                                                    identity_id, kmer_thr, max_hits, min_diag, bin_size, _ptr(ids), _ptr(sc), _ptr(dg),
                                                    _ptr(st))
        if n < 0:
            raise RuntimeError('oracle prefilter path not restated: code %d' % n)
        return ids[:n].copy(), sc[:n].copy(), dg[:n].copy(), st

    def __del__(self):
        try:
            self.orc.lib.or_target_destroy(self.h)
        except Exception:
            pass


def ref_available():
    return os.path.exists(os.path.join(HERE, '_ref', 'libsdref.so'))


class Ref:
    """The real reference (AVX2 build flags of M/CMakeLists.txt:69) behind oracle/ref_driver.cpp."""

    def __init__(self, k=6):
        path = os.path.join(HERE, '_ref', 'libsdref.so')
        # ProfileStates (needs a cmake-generated header) is referenced but never reached on this path:
        # bind lazily so its two constructors may stay unresolved.
        L = self.lib = C.CDLL(path, mode=os.RTLD_LAZY)
        L.ref_ctx_create.restype = _vp
        L.ref_ctx_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.ref_get_matrix.argtypes = [_vp, C.c_int, _vp, _vp, _vp]
        L.ref_compbias.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_float, _vp]
        L.ref_ext_matrix.restype = C.c_size_t
        L.ref_ext_matrix.argtypes = [_vp, C.c_int, _vp, _vp, _vp]
        L.ref_kmer_list.restype = C.c_size_t
        L.ref_kmer_list.argtypes = [_vp, _vp, C.c_int, _vp, C.c_size_t]
        L.ref_mask.argtypes = [_vp, _vp, C.c_int, C.c_double]
        L.ref_index_build.restype = _vp
        L.ref_index_build.argtypes = [_vp, C.c_char_p, _vp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_int]
        L.ref_index_info.restype = C.c_size_t
        L.ref_index_info.argtypes = [_vp, _vp, _vp]
        L.ref_index_dump.argtypes = [_vp, _vp, _vp, _vp]
        L.ref_index_destroy.argtypes = [_vp]
        L.ref_prefilter_create.restype = _vp
        L.ref_prefilter_create.argtypes = [_vp, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.ref_prefilter_query.restype = C.c_size_t
        L.ref_prefilter_query.argtypes = [_vp, C.c_char_p, C.c_uint, C.c_uint, _vp, _vp, _vp, _vp]
        L.ref_prefilter_destroy.argtypes = [_vp]
        L.ref_sw_create.restype = _vp
        L.ref_sw_create.argtypes = [_vp, C.c_size_t, C.c_size_t, C.c_int]
        L.ref_sw_set_query.argtypes = [_vp, C.c_char_p, C.c_uint]
        L.ref_sw_align.restype = C.c_double
        L.ref_sw_align.argtypes = [_vp, C.c_char_p, C.c_uint, C.c_int, C.c_double, C.c_int, C.c_float, _vp,
                                   C.c_char_p, C.c_size_t, C.c_int]
        L.ref_sw_destroy.argtypes = [_vp]
        L.ref_sw_set_query_profile.argtypes = [_vp, C.c_char_p, C.c_uint]
        L.ref_prefilter_create_profile.restype = _vp
        L.ref_prefilter_create_profile.argtypes = [_vp, C.c_int, C.c_size_t, C.c_size_t, C.c_int]
        L.ref_profile_kmer_list.restype = C.c_size_t
        L.ref_profile_kmer_list.argtypes = [_vp, C.c_char_p, C.c_uint, C.c_uint, C.c_int, _vp, C.c_size_t]
        L.ref_evalue.restype = C.c_double
        L.ref_evalue.argtypes = [_vp, C.c_double, C.c_double]
        L.ref_bitscore.restype = C.c_double
        L.ref_bitscore.argtypes = [_vp, C.c_double]
        L.ref_l2_cache_size.restype = C.c_ulong
        self.k = k
        if os.path.exists(REF_DATA + 'blosum62.out'):
            a, b = (REF_DATA + 'blosum62.out').encode(), (REF_DATA + 'VTML80.out').encode()
        else:
            # no reference tree on this box: hand the matrix *contents* over in the reference's own
            # "NAME.out:DATA" form (BaseMatrix::unserialize, M/src/commons/BaseMatrix.cpp:189-214)
            H = C.CDLL(os.path.join(os.path.dirname(HERE), 'spacedust_amd', 'libsdgpu.so'))
            buf = C.create_string_buffer(1 << 16)
            H.sd_host_matrix_text.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
            H.sd_host_matrix_text(0, buf, 1 << 16)
            a = b'blosum62.out:' + buf.value
            H.sd_host_matrix_text(1, buf, 1 << 16)
            b = b'VTML80.out:' + buf.value
        self.ctx = L.ref_ctx_create(a, b, k)

    def matrix(self, which):
        m = np.zeros((21, 21), np.int16)
        pb = np.zeros(21, np.float64)
        a2n = np.zeros(256, np.uint8)
        self.lib.ref_get_matrix(self.ctx, which, _ptr(m), _ptr(pb), _ptr(a2n))
        return m, pb, a2n

    def compbias(self, which, num, scale=1.0):
        num = np.ascontiguousarray(num, np.uint8)
        out = np.zeros(len(num), np.float32)
        self.lib.ref_compbias(self.ctx, which, _ptr(num), len(num), scale, _ptr(out))
        return out

    def ext_matrix(self, which):
        rs = C.c_size_t()
        n = self.lib.ref_ext_matrix(self.ctx, which, None, None, C.byref(rs))
        sc = np.zeros((n, rs.value), np.int16)
        ix = np.zeros((n, rs.value), np.uint32)
        self.lib.ref_ext_matrix(self.ctx, which, _ptr(sc), _ptr(ix), C.byref(rs))
        return sc[:, :n], ix[:, :n]

    def kmer_list(self, window, thr, cap=1 << 22):
        out = np.zeros(cap, np.uint64)
        w = np.zeros(64, np.uint8)
        w[:len(window)] = window
        n = self.lib.ref_kmer_list(self.ctx, _ptr(w), thr, _ptr(out), cap)
        return out[:n].astype(np.uint32)

    def profile_kmer_list(self, data, pos, thr, cap=1 << 22):
        data = bytes(data)
        out = np.zeros(cap, np.uint64)
        n = self.lib.ref_profile_kmer_list(self.ctx, data, len(data) // 25, pos, thr, _ptr(out), cap)
        return out[:n].copy()

    def mask(self, num, prob=0.9):
        a = np.array(num, np.uint8, copy=True)
        n = self.lib.ref_mask(self.ctx, _ptr(a), len(a), prob)
        return a, n

    def index(self, ascii_blob, offsets, kmer_thr=112, mask=True, mask_prob=0.9, threads=8):
        return RefIndex(self, ascii_blob, offsets, kmer_thr, mask, mask_prob, threads)

    def l2_cache_size(self):
        return int(self.lib.ref_l2_cache_size())


class RefIndex:
    def __init__(self, ref, blob, offsets, kmer_thr, mask, mask_prob, threads):
        self.ref = ref
        self.blob = blob
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        self.n = len(self.offsets) - 1
        self.h = ref.lib.ref_index_build(ref.ctx, blob, _ptr(self.offsets), self.n, kmer_thr, 1 if mask else 0,
                                         mask_prob, threads)
        ts = C.c_size_t()
        mk = C.c_size_t()
        self.n_entries = ref.lib.ref_index_info(self.h, C.byref(ts), C.byref(mk))
        self.table_size = ts.value
        self.masked_residues = mk.value
        self.kmer_thr = kmer_thr

    def dump(self):
        off = np.zeros(self.table_size + 1, np.uint64)
        ent = np.zeros(self.n_entries * 6, np.uint8)
        lk = np.zeros(int(self.offsets[-1]), np.uint8)
        self.ref.lib.ref_index_dump(self.h, _ptr(off), _ptr(ent), _ptr(lk))
        e = ent.reshape(-1, 6)
        es = e[:, :4].copy().view(np.uint32).reshape(-1)
        ep = e[:, 4:6].copy().view(np.uint16).reshape(-1)
        return off, es, ep, lk

    def prefilter(self, max_query_len, max_hits=300, min_diag=15, comp_bias=True):
        return RefPrefilter(self, max_query_len, max_hits, min_diag, comp_bias)

    def prefilter_profile(self, max_query_len, kmer_thr, max_hits=300, min_diag=15):
        """profile queries against this index (build it with kmer_thr=0, Prefiltering.cpp:525-527); query() then takes
        the raw 25-byte-per-position profile entry"""
        p = RefPrefilter.__new__(RefPrefilter)
        p.idx, p.lib, p.max_hits = self, self.ref.lib, max_hits
        p.h = self.ref.lib.ref_prefilter_create_profile(self.h, kmer_thr, max_query_len, max_hits, min_diag)
        p.profile = True
        return p


class RefPrefilter:
    def __init__(self, idx, max_query_len, max_hits, min_diag, comp_bias):
        self.idx = idx
        self.lib = idx.ref.lib
        self.max_hits = max_hits
        self.h = self.lib.ref_prefilter_create(idx.h, idx.kmer_thr, max_query_len, max_hits, min_diag,
                                               1 if comp_bias else 0)

    def query(self, ascii_seq, identity_id=0xFFFFFFFF):
        cap = self.max_hits + 2
        ids = np.zeros(cap, np.uint32)
        sc = np.zeros(cap, np.int32)
        dg = np.zeros(cap, np.uint16)
        st = np.zeros(2, np.float64)
        b = ascii_seq.encode() if isinstance(ascii_seq, str) else bytes(ascii_seq)
        L = len(b) // 25 if getattr(self, 'profile', False) else len(b)
        n = self.lib.ref_prefilter_query(self.h, b, L, identity_id, _ptr(ids), _ptr(sc), _ptr(dg), _ptr(st))
        return ids[:n].copy(), sc[:n].copy(), dg[:n].copy(), st


class RefSW:
    def __init__(self, ref, max_len, db_residues, comp_bias=True):
        self.lib = ref.lib
        self.h = ref.lib.ref_sw_create(ref.ctx, max_len, db_residues, 1 if comp_bias else 0)

    def set_query(self, ascii_seq):
        b = ascii_seq.encode() if isinstance(ascii_seq, str) else ascii_seq
        self.qlen = len(b)
        self.lib.ref_sw_set_query(self.h, b, len(b))

    def set_query_profile(self, data):
        b = bytes(data)
        self.qlen = len(b) // 25
        self.lib.ref_sw_set_query_profile(self.h, b, self.qlen)

    def align(self, ascii_t, sw_mode=2, eval_thr=10.0, cov_mode=2, cov_thr=0.8, identity=False):
        b = ascii_t.encode() if isinstance(ascii_t, str) else ascii_t
        out = np.zeros(8, np.int32)
        cap = self.qlen + len(b) + 8
        bt = C.create_string_buffer(cap)
        ev = self.lib.ref_sw_align(self.h, b, len(b), sw_mode, eval_thr, cov_mode, cov_thr, _ptr(out), bt, cap,
                                   1 if identity else 0)
        return dict(score=int(out[0]), qStart=int(out[1]), qEnd=int(out[2]), tStart=int(out[3]), tEnd=int(out[4]),
                    identical=int(out[5]), btLen=int(out[6]), evalue=ev, backtrace=bt.value.decode())

    def evalue(self, score, qlen):
        return self.lib.ref_evalue(self.h, float(score), float(qlen))

    def bitscore(self, score):
        return self.lib.ref_bitscore(self.h, float(score))


def read_fasta(path):
    names, seqs, cur = [], [], []
    with open(path) as f:
        for line in f:
            line = line.rstrip('\n')
            if line.startswith('>'):
                if names:
                    seqs.append(''.join(cur))
                names.append(line[1:])
                cur = []
            else:
                cur.append(line)
    seqs.append(''.join(cur))
    return names, seqs


def oracle_clusterhits(orc, q_pos, t_pos, strands, pval, nq, d=3, cls=2, alpha=1.0, p_clu=0.01, p_mh=0.01, lg=None):
    """or_clusterhits for one entry -> (cluster_of[K], member_order[K], sizes[n], pCO[n], pMH[n], n_merges)"""
    L = orc.lib
    L.or_clusterhits.restype = C.c_int
    L.or_clusterhits.argtypes = [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                                                   C.c_float, C.c_float] + [C.c_void_p] * 8
    L.or_lgamma_table.argtypes = [C.c_void_p, C.c_uint32]
    q_pos = np.ascontiguousarray(q_pos, np.uint32)
    t_pos = np.ascontiguousarray(t_pos, np.uint32)
    strands = np.ascontiguousarray(strands, np.uint8)
    pval = np.ascontiguousarray(pval, np.float64)
    K = len(q_pos)
    if lg is None:
        n = int(max(q_pos.max(initial=0), t_pos.max(initial=0), nq, K)) + 8
        lg = np.zeros(n)
        L.or_lgamma_table(_ptr(lg), n)
    cof = np.zeros(max(K, 1), np.uint32)
    mo = np.zeros(max(K, 1), np.uint32)
    cs = np.zeros(max(K, 1), np.uint32)
    pco = np.zeros(max(K, 1))
    pmh = np.zeros(max(K, 1))
    nm = C.c_uint32()
    n = L.or_clusterhits(K, _ptr(q_pos), _ptr(t_pos), _ptr(strands), _ptr(pval), nq, d, cls, alpha, p_clu, p_mh,
                         _ptr(lg), _ptr(cof), _ptr(mo), _ptr(cs), _ptr(pco), _ptr(pmh), None, C.byref(nm))
    return cof[:K], mo, cs[:n], pco[:n], pmh[:n], nm.value


def ref_ch_available():
    return os.path.exists(os.path.join(HERE, '_ref', 'libsdref_ch.so'))


class RefClusterHits:
    """The reference's own clusterhits arithmetic (R/src/util/ClusterHits.cpp compiled where it lies: groupNodes,
    clusterMatchScore, isCompatibleCluster, multihitPval, logGamma ...) behind oracle/ref_clusterhits.cpp."""

    def __init__(self):
        self.lib = C.CDLL(os.path.join(HERE, '_ref', 'libsdref_ch.so'))
        self.lib.ref_ch_loggamma.restype = C.c_double
        self.lib.ref_ch_loggamma.argtypes = [C.c_double]
        self.lib.ref_clusterhits_entry.restype = C.c_int
        self.lib.ref_clusterhits_entry.argtypes = [C.c_uint32] + [C.c_void_p] * 4 + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_double,
                                                                                 C.c_float, C.c_float] + [C.c_void_p] * 7

    def lgamma_table(self, n):
        return np.array([self.lib.ref_ch_loggamma(float(i)) for i in range(n)])

    def entry(self, q_pos, t_pos, strands, pval, nq, d=3, cls=2, alpha=1.0, p_clu=0.01, p_mh=0.01, lg=None):
        """-> (cluster_of[K], rank[K], sizes[n], pCO[n], pMH[n])"""
        q_pos = np.ascontiguousarray(q_pos, np.uint32)
        t_pos = np.ascontiguousarray(t_pos, np.uint32)
        strands = np.ascontiguousarray(strands, np.uint8)
        pval = np.ascontiguousarray(pval, np.float64)
        K = len(q_pos)
        if lg is None:
            lg = self.lgamma_table(int(max(q_pos.max(initial=0), t_pos.max(initial=0), nq, K)) + 8)
        lg = np.ascontiguousarray(lg, np.float64)
        cof = np.zeros(max(K, 1), np.uint32)
        rk = np.zeros(max(K, 1), np.uint32)
        cs = np.zeros(max(K, 1), np.uint32)
        pco = np.zeros(max(K, 1))
        pmh = np.zeros(max(K, 1))
        n = C.c_uint32()
        self.lib.ref_clusterhits_entry(K, _ptr(q_pos), _ptr(t_pos), _ptr(strands), _ptr(pval), nq, d, cls, alpha, p_clu, p_mh,
                                       _ptr(lg), _ptr(cof), _ptr(rk), _ptr(pco), _ptr(pmh), _ptr(cs), C.byref(n))
        return cof[:K], rk[:K], cs[:n.value], pco[:n.value], pmh[:n.value]


def matrix_spec(which=0):
    """what the reference's SubstitutionMatrix constructor takes: the .out path under /root/reference, or -- on a box without
    the tree -- the matrix contents in its own "NAME.out:DATA" form (BaseMatrix.cpp:189-214)"""
    name = 'blosum62.out' if which == 0 else 'VTML80.out'
    if os.path.exists(REF_DATA + name):
        return REF_DATA + name
    H = C.CDLL(os.path.join(os.path.dirname(HERE), 'spacedust_amd', 'libsdgpu.so'))
    buf = C.create_string_buffer(1 << 16)
    H.sd_host_matrix_text.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    H.sd_host_matrix_text(which, buf, 1 << 16)
    return name + ':' + buf.value.decode()


def ref_r2p_available():
    return os.path.exists(os.path.join(HERE, '_ref', 'libsdref_r2p.so'))


class RefResult2Profile:
    """The reference's own MultipleAlignment / MsaFilter / PSSMCalculator / Masker classes driven for one centre sequence
    (oracle/ref_result2profile.cpp): alignment records in, 25-byte profile records out."""

    def __init__(self, blosum_spec=None):
        self.lib = C.CDLL(os.path.join(HERE, '_ref', 'libsdref_r2p.so'), mode=os.RTLD_LAZY)
        self.lib.ref_result2profile.restype = C.c_int
        self.blosum = (blosum_spec or matrix_spec(0)).encode()

    def profile(self, centre, edges, q_start, t_start, backtraces, centre_profile=None, pca=1.1, pcb=4.1, wg=0, filter_msa=1,
                cov=0.0, qid='0.0', qsc=-20.0, max_seq_id=0.9, ndiff=1000, filter_min_enable=0, comp_bias=1, mask_profile=1,
                mask_prob=0.9):
        """centre: ASCII sequence (or None with centre_profile = L*25 bytes); edges: list of ASCII target sequences"""
        n = len(edges)
        L = len(centre) if centre_profile is None else len(centre_profile) // 25
        eseq = (C.c_char_p * max(n, 1))(*[e.encode() for e in edges])
        elen = (C.c_uint * max(n, 1))(*[len(e) for e in edges])
        qs = (C.c_int * max(n, 1))(*[int(x) for x in q_start])
        ts = (C.c_int * max(n, 1))(*[int(x) for x in t_start])
        bts = (C.c_char_p * max(n, 1))(*[b.encode() for b in backtraces])
        out = C.create_string_buffer(L * 25 + 64)
        fl = C.c_float
        got = self.lib.ref_result2profile(self.blosum, centre.encode() if centre_profile is None else None,
                                          centre_profile if centre_profile is not None else None, C.c_uint(L), C.c_uint(n), eseq, elen,
                                          qs, ts, bts, fl(pca), fl(pcb), wg, filter_msa, fl(cov), qid.encode(), fl(qsc), fl(max_seq_id),
                                          ndiff, filter_min_enable, comp_bias, mask_profile, fl(mask_prob), out)
        return out.raw[:got]
