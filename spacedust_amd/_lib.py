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


class SetDbView(C.Structure):   # sd_setdb
    _fields_ = [('residues', _vp), ('offsets', _vp), ('n', C.c_uint32), ('setId', _vp), ('posInSet', _vp), ('strand', _vp),
                ('nSets', C.c_uint32), ('keys', _vp), ('alnProfile', _vp), ('sortedScore', _vp), ('sortedIndex', _vp)]


class IndexView(C.Structure):   # sd_index_view
    _fields_ = [('kmerSize', C.c_int32), ('kmerThr', C.c_int32), ('kmerOffsets', _vp), ('entrySeq', _vp), ('entryPos', _vp),
                ('nEntries', C.c_uint64), ('maskedResidues', _vp), ('nMaskedResidues', C.c_uint64), ('kmerBlockBase', _vp)]


class SearchParams(C.Structure):   # sd_search_params
    _fields_ = [('sensitivity', C.c_float), ('kmerSize', C.c_int32), ('maxSeqs', C.c_int32), ('minDiagScore', C.c_int32),
                ('binSize', C.c_uint32), ('mask', C.c_int32), ('maskProb', C.c_double), ('compBiasCorr', C.c_int32),
                ('evalThr', C.c_double), ('covMode', C.c_int32), ('covThr', C.c_float), ('alnLenThr', C.c_int32),
                ('maxGeneGap', C.c_uint32), ('clusterSize', C.c_uint32), ('alpha', C.c_double), ('pCluThr', C.c_float),
                ('pMHThr', C.c_float), ('filterSelfMatch', C.c_int32), ('profileQueries', C.c_int32), ('chunkQueries', C.c_int32),
                ('deviceBias', C.c_int32), ('threads', C.c_int32), ('alignPriority', C.c_int32)]


class R2pParams(C.Structure):   # sd_r2p_params
    _fields_ = [('filterMsa', C.c_int32), ('filterMinEnable', C.c_int32), ('filterMaxSeqId', C.c_float), ('qid', C.c_char_p),
                ('qsc', C.c_float), ('covMSAThr', C.c_float), ('Ndiff', C.c_int32), ('pcMode', C.c_int32), ('pca', C.c_float),
                ('pcb', C.c_float), ('wg', C.c_int32), ('compBiasCorr', C.c_int32), ('maskProfile', C.c_int32), ('maskProb', C.c_double)]


class AlnCriteria(C.Structure):   # sd_aln_criteria
    _fields_ = [('evalThr', C.c_double), ('seqIdThr', C.c_float), ('alnLenThr', C.c_int32), ('covMode', C.c_int32),
                ('covThr', C.c_float), ('seqIdMode', C.c_int32), ('swMode', C.c_int32), ('addBacktrace', C.c_int32),
                ('realign', C.c_int32), ('realignSwMode', C.c_int32), ('realignMaxSeqs', C.c_int32), ('maxAccept', C.c_uint32),
                ('maxRejected', C.c_uint32)]


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
        'sd_ctx_create_masked': (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
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
        'sd_sw_align_batch_compact_diag': (C.c_int, [_vp, C.POINTER(SwParams), _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp,
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
        'sd_host_cluster_pvalues': (C.c_int, [C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint32, C.c_double, _vp, C.c_uint32, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), _vp]),
        'sd_host_evalue': (C.c_double, [C.c_uint64, C.c_double, C.c_double]),
        'sd_host_bitscore': (C.c_double, [C.c_double]),
        'sd_target_create': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint32, _vp, _vp, _vp,
                                       _vp, C.POINTER(_vp)]),
        'sd_target_create_wide': (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint32, _vp, _vp, _vp,
                                            _vp, C.POINTER(_vp)]),
        'sd_host_index_block_base': (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(C.c_uint64)]),
        'sd_target_destroy': (None, [_vp]),
        'sd_target_build': (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp,
                                      C.POINTER(_vp), _vp]),
        'sd_target_download': (C.c_int, [_vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _vp, _vp, _vp, _vp]),
        'sd_host_index_tables': (C.c_int, [_vp, _vp, _vp]),
        'sd_target_sample_check': (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                             C.c_uint64, C.c_uint64, _vp]),
        'sd_device_memory': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_device_identity': (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
        'sd_workspace_report': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t]),
        'sd_workspace_release': (C.c_int, [_vp]),
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
        'sd_agg_write_tsv_from': (C.c_int, [_vp, C.c_char_p, C.c_int, C.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_char_p, _vp,
                                            C.c_char_p, _vp, C.c_char_p, _vp, C.c_char_p, _vp, C.c_int, C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]),
        'sd_agg_set_keys': (C.c_int, [_vp, _vp, _vp]),
        'sd_agg_set_list_order': (C.c_int, [_vp, C.c_int]),
        'sd_agg_set_pool_form': (C.c_int, [_vp, C.c_int]),
        'sd_host_matrix_text': (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
        'sd_host_sw_comp_bias': (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint32, _vp]),
        'sd_host_can_be_covered': (C.c_int, [C.c_float, C.c_int, C.c_float, C.c_float]),
        'sd_host_quantise_3e': (C.c_int, [C.c_double, C.c_char_p, C.POINTER(C.c_double)]),
        'sd_host_compress_backtrace': (C.c_int, [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64)]),
        'sd_host_accept_sort': (C.c_int, [C.POINTER(AlnCriteria), C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
        'sd_host_realign_select': (C.c_int, [C.POINTER(AlnCriteria), C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _vp]),
        'sd_alntext_create': (C.c_int, [C.POINTER(_vp)]),
        'sd_alntext_destroy': (None, [_vp]),
        'sd_alntext_format': (C.c_int, [_vp, C.POINTER(AlnCriteria), C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
        'sd_alntext_get': (C.c_int, [_vp, C.POINTER(C.c_char_p), C.POINTER(_vp)]),
        'sd_search_default_params': (None, [C.POINTER(SearchParams)]),
        'sd_search_create': (C.c_int, [C.c_int, C.POINTER(SearchParams), C.POINTER(SetDbView), C.POINTER(_vp)]),
        'sd_search_create_indexed': (C.c_int, [C.c_int, C.POINTER(SearchParams), C.POINTER(SetDbView), C.POINTER(IndexView), C.POINTER(_vp)]),
        'sd_search_destroy': (None, [_vp]),
        'sd_search_last_error': (C.c_char_p, [_vp]),
        'sd_search_ctx': (_vp, [_vp, C.c_int]),
        'sd_search_target': (_vp, [_vp]),
        'sd_search_set_sinks': (C.c_int, [_vp, _vp, _vp, _vp]),
        'sd_search_set_chunk_queries': (C.c_int, [_vp, C.c_int32]),
        'sd_search_set_want_records': (C.c_int, [_vp, C.c_int]),
        'sd_search_stream': (C.c_int, [_vp, C.POINTER(SetDbView), C.c_int, C.c_uint32, _vp, _vp, _vp]),
        'sd_search_result_counts': (C.c_int, [_vp, _vp]),
        'sd_search_result_arrays': (C.c_int, [_vp] + [_vp] * 12),
        'sd_search_result_write_tsv': (C.c_int, [_vp, C.c_char_p, C.c_char_p, _vp, C.c_char_p, _vp, C.c_char_p, _vp, C.c_char_p, _vp,
                                                 C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_search_result_records': (C.c_int, [_vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
        'sd_agg_records': (C.c_int, [_vp] * 7 + [_vp, C.c_uint64, C.POINTER(C.c_uint64)]),
        'sd_records_check': (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_records_write_tsv': (C.c_int, [_vp, C.c_uint64, C.c_char_p, C.c_int, C.c_uint64, C.c_char_p, _vp, C.c_char_p, _vp, C.c_char_p, _vp,
                                           C.c_char_p, _vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_search_result_destroy': (None, [_vp]),
        'sd_search_stats': (C.c_int, [_vp, _vp, _vp]),
        'sd_search_download_bytes': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_sw_set_cigar_pool': (C.c_int, [_vp, C.c_int]),
        'sd_sw_download_bytes': (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        'sd_r2p_create': (C.c_int, [C.POINTER(_vp)]),
        'sd_r2p_destroy': (None, [_vp]),
        'sd_r2p_batch': (C.c_int, [_vp, C.POINTER(R2pParams), C.c_uint32] + [_vp] * 12),
        'sd_r2p_batch_device': (C.c_int, [_vp, _vp, C.POINTER(R2pParams), C.c_uint32] + [_vp] * 12),
        'sd_shard_query_sets': (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.POINTER(C.c_uint32)]),
        'sd_comm_unique_id': (C.c_int, [C.c_char_p]),
        'sd_comm_init': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(_vp)]),
        'sd_comm_destroy': (None, [_vp]),
        'sd_comm_host_buffer': (C.c_int, [_vp, C.c_int, C.c_uint64, C.POINTER(_vp)]),
        'sd_comm_last_error': (C.c_char_p, [_vp]),
        'sd_search_set_records_sink': (C.c_int, [_vp, _vp, _vp]),
        'sd_gather_stream_begin': (C.c_int, [_vp, C.c_int, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint64, C.c_int, C.POINTER(_vp)]),
        'sd_gather_stream_end': (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_uint64)]),
        'sd_gather_stream_begin_tcp': (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint32, _vp, C.c_uint32, _vp, C.c_uint64, C.c_int, C.POINTER(_vp)]),
        'sd_gather_results': (C.c_int, [_vp, _vp, C.c_uint64, C.c_int, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
        'sd_tcp_connect': (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
        'sd_tcp_close': (None, [_vp]),
        'sd_tcp_bcast': (C.c_int, [_vp, _vp, C.c_uint64]),
        'sd_tcp_gather': (C.c_int, [_vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
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


def _declared_symbols():
    """every function include/spacedust_gpu.h declares (tests/test_capi.py checks the library exports each of them)"""
    import re
    text = open(os.path.join(os.path.dirname(HERE), 'include', 'spacedust_gpu.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = []
    for m in re.finditer(r'\b(sd_[a-z0-9_]+)\s*\(', text):
        if not re.search(r'typedef[^;]*\(\s*\*\s*%s' % m.group(1), text) and m.group(1) not in names:
            names.append(m.group(1))
    return names


DECLARED_SYMBOLS = _declared_symbols()


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None
