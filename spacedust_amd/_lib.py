"""ctypes binding of libsdgpu.so (include/spacedust_gpu.h).  The library is built in-tree by
spacedust_amd/build.py; loading fails loudly if it is missing -- there is no Python/CPU fallback."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libsdgpu.so')

_vp = C.c_void_p


SD_OK, SD_ENODEVICE, SD_EHIP, SD_EINVAL, SD_ENOMEM, SD_EUNSUPPORTED, SD_EMISMATCH = 0, -1, -2, -3, -4, -5, -6


class SdError(RuntimeError):
    pass


class SwParams(C.Structure):
    _fields_ = [('gapOpen', C.c_int32), ('gapExtend', C.c_int32), ('matrix', C.c_int8 * 441), ('covMode', C.c_int32),
                ('covThr', C.c_float), ('evalThr', C.c_double), ('swMode', C.c_int32), ('dbResidues', C.c_uint64)]


class SwResult(C.Structure):
    _fields_ = [('score', C.c_int32), ('qStart', C.c_int32), ('qEnd', C.c_int32), ('tStart', C.c_int32),
                ('tEnd', C.c_int32), ('identical', C.c_int32), ('btLen', C.c_int32), ('flags', C.c_int32),
                ('evalue', C.c_double), ('btOffset', C.c_uint64)]


SW_RESULT_DTYPE = np.dtype([('score', '<i4'), ('qStart', '<i4'), ('qEnd', '<i4'), ('tStart', '<i4'), ('tEnd', '<i4'),
                            ('identical', '<i4'), ('btLen', '<i4'), ('flags', '<i4'), ('evalue', '<f8'),
                            ('btOffset', '<u8')])
assert SW_RESULT_DTYPE.itemsize == C.sizeof(SwResult)


class PrefilterParams(C.Structure):
    _fields_ = [('kmerSize', C.c_int32), ('kmerThr', C.c_int32), ('maxHitsPerQuery', C.c_int32),
                ('minDiagScore', C.c_int32), ('binSize', C.c_uint32), ('covMode', C.c_int32), ('covThr', C.c_float),
                ('ungappedMatrix', C.c_int8 * 441)]


HIT_DTYPE = np.dtype([('seqId', '<u4'), ('score', '<i4'), ('diagonal', '<u2'), ('pad', '<u2')])


class ChParams(C.Structure):
    _fields_ = [('maxGeneGap', C.c_uint32), ('clusterSize', C.c_uint32), ('alpha', C.c_double),
                ('pCluThr', C.c_float), ('pMHThr', C.c_float)]


_lib = None


def load():
    """dlopen libsdgpu.so (raises SdError if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdError('libsdgpu.so is missing (%s): run `python -m spacedust_amd.build`; there is no CPU fallback'
                      % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    sig = {
        'sd_ctx_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
        'sd_ctx_create_prio': (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
        'sd_ctx_destroy': (None, [_vp]),
        'sd_last_error': (C.c_char_p, [_vp]),
        'sd_device_name': (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
        'sd_synchronize': (C.c_int, [_vp]),
        'sd_profile_enable': (C.c_int, [_vp, C.c_int]),
        'sd_profile_reset': (C.c_int, [_vp]),
        'sd_profile_get': (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
        'sd_profile_names': (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
        'sd_seqset_create': (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, C.POINTER(_vp)]),
        'sd_seqset_destroy': (None, [_vp]),
        'sd_sw_align_batch': (C.c_int, [_vp, C.POINTER(SwParams), _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp,
                                        C.c_uint64, C.POINTER(C.c_uint64)]),
        'sd_sw_align_batch_compact': (C.c_int, [_vp, C.POINTER(SwParams), _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp,
                                                C.POINTER(C.c_uint32), _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
        'sd_sw_align_batch_hostpath': (C.c_int, [_vp, C.POINTER(SwParams), _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp,
                                                 C.c_uint64, C.POINTER(C.c_uint64)]),
        'sd_sw_score_batch': (C.c_int, [_vp, C.POINTER(SwParams), _vp, _vp, C.c_uint32, _vp, _vp, C.c_int, C.c_int,
                                        _vp, _vp, _vp]),
        'sd_sw_last_cells': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_host_create': (C.c_int, [C.c_int, C.POINTER(_vp)]),
        'sd_host_destroy': (None, [_vp]),
        'sd_host_matrix': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
        'sd_host_map_sequence': (C.c_int, [_vp, C.c_char_p, C.c_uint64, _vp]),
        'sd_host_comp_bias': (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_int, _vp, _vp, _vp]),
        'sd_host_index_build': (C.c_int, [_vp, _vp, _vp, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_double,
                                          C.POINTER(_vp)]),
        'sd_host_index_info': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_host_index_arrays': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
        'sd_host_index_destroy': (None, [_vp]),
        'sd_host_ext_matrix': (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_uint32)]),
        'sd_host_kmer_threshold': (C.c_int, [C.c_float, C.c_int]),
        'sd_host_auto_kmer_size': (C.c_int, [C.c_uint64]),
        'sd_host_bin_size': (C.c_uint, [C.c_uint64, C.c_uint64]),
        'sd_host_pair_list': (C.c_uint64, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp]),
        'sd_host_lgamma_table': (C.c_int, [_vp, C.c_uint32]),
        'sd_host_evalue': (C.c_double, [C.c_uint64, C.c_double, C.c_double]),
        'sd_host_bitscore': (C.c_double, [C.c_double]),
        'sd_target_create': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint32, _vp, _vp, _vp,
                                       _vp, C.POINTER(_vp)]),
        'sd_target_destroy': (None, [_vp]),
        'sd_prefilter_batch': (C.c_int, [_vp, _vp, C.POINTER(PrefilterParams), C.c_uint32, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp]),
        'sd_comp_bias_batch': (C.c_int, [_vp, _vp, _vp, _vp, C.c_uint32, C.c_int, _vp, _vp, _vp]),
        'sd_prefilter_profile_batch': (C.c_int, [_vp, _vp, C.POINTER(PrefilterParams), C.c_uint32, _vp, _vp, _vp, _vp, _vp,
                                                 _vp, _vp, _vp, _vp]),
        'sd_profileset_create': (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, C.POINTER(_vp)]),
        'sd_host_map_profiles': (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
        'sd_host_profile_kmer_threshold': (C.c_int, [C.c_float, C.c_int]),
        'sd_clusterhits_batch': (C.c_int, [_vp, C.POINTER(ChParams), C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                           C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
        'sd_agg_create': (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_int,
                                    C.c_float, C.c_int, C.c_int, C.POINTER(_vp)]),
        'sd_agg_destroy': (None, [_vp]),
        'sd_agg_add': (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
        'sd_agg_finish': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_agg_stats': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_agg_get': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
        'sd_agg_write_tsv': (C.c_int, [_vp, C.c_char_p, _vp, _vp, _vp, _vp, _vp, _vp, C.c_char_p, _vp, C.c_char_p, _vp,
                                       C.c_char_p, _vp, C.c_char_p, _vp, C.c_int, C.POINTER(C.c_uint64),
                                       C.POINTER(C.c_uint64)]),
    }
    missing = []
    for name, (res, args) in sig.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        f.restype = res
        f.argtypes = args
    L._sd_missing = missing
    _lib = L
    return L


DECLARED_SYMBOLS = [
    'sd_ctx_create', 'sd_ctx_create_prio', 'sd_ctx_destroy', 'sd_last_error', 'sd_device_name', 'sd_synchronize', 'sd_profile_enable',
    'sd_profile_reset', 'sd_profile_get', 'sd_profile_names', 'sd_seqset_create', 'sd_seqset_destroy',
    'sd_sw_align_batch', 'sd_sw_align_batch_compact', 'sd_sw_align_batch_hostpath', 'sd_sw_score_batch', 'sd_sw_last_cells', 'sd_target_create', 'sd_target_destroy',
    'sd_prefilter_batch', 'sd_comp_bias_batch', 'sd_prefilter_profile_batch', 'sd_profileset_create', 'sd_host_map_profiles',
    'sd_host_profile_kmer_threshold', 'sd_clusterhits_batch', 'sd_host_create', 'sd_host_destroy', 'sd_host_matrix',
    'sd_host_map_sequence', 'sd_host_comp_bias', 'sd_host_index_build', 'sd_host_index_info', 'sd_host_index_arrays',
    'sd_host_index_destroy', 'sd_host_ext_matrix', 'sd_host_kmer_threshold', 'sd_host_auto_kmer_size', 'sd_host_bin_size', 'sd_host_pair_list',
    'sd_host_lgamma_table', 'sd_host_evalue', 'sd_host_bitscore', 'sd_agg_create', 'sd_agg_destroy', 'sd_agg_add',
    'sd_agg_finish', 'sd_agg_stats', 'sd_agg_get', 'sd_agg_write_tsv',
]


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
