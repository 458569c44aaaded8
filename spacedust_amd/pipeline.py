"""clustersearch --search-mode 0 on one GPU: prefilter -> align -> (host aggregation) -> clusterhits -> TSV.

Mirrors R/data/clustersearch.sh:110-152 (`search` = prefilter + align, then besthitbyset / mergeresultsbyset /
combinehits fused in sd_agg, `clusterhits`, `summarizeresults`).  The pipeline itself -- stage threads, chunking, the two
device contexts, buffers -- is C++ (csrc/host/sd_search.cpp behind sd_search_* of the C ABI); this file is its Python
view for the tests and bench.py, plus the multi-GPU sharding helpers."""
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _lib, api
from ._lib import ptr


class SetDB:
    """What createsetdb provides for one side (R/data/createsetdb.sh): numeric sequences, set membership,
    gene position / strand from the lookup names, set sizes, source names."""

    def __init__(self, residues, offsets, set_id, pos_in_set, strand, n_sets, names=None, sources=None):
        self.residues = np.ascontiguousarray(residues, np.uint8)
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        self.set_id = np.ascontiguousarray(set_id, np.uint32)
        self.pos_in_set = np.ascontiguousarray(pos_in_set, np.uint32)
        self.strand = np.ascontiguousarray(strand, np.uint8)
        self.n_sets = int(n_sets)
        self.n = len(self.offsets) - 1
        self.set_size = np.bincount(self.set_id, minlength=self.n_sets).astype(np.uint32)
        self.names = names
        self.sources = sources
        self.profile = None   # profile DB side (iterations >= 1 of --num-iterations): see with_profiles

    def with_profiles(self, prof):
        """attach Host.map_profiles output (letters / aln / sorted rows, one profile per protein, same order): the
        set then searches as a DBTYPE_HMM_PROFILE query DB; `residues` become the profiles' query letters"""
        assert len(prof['offsets']) == self.n + 1
        q = SetDB(prof['letters'], prof['offsets'], self.set_id, self.pos_in_set, self.strand, self.n_sets, self.names, self.sources)
        q.profile = prof
        return q

    @staticmethod
    def from_proteomes(ps, prefix='SYN'):
        return SetDB(ps.residues, ps.offsets, ps.set_id, ps.pos_in_set, ps.strand, ps.n_sets)

    def lengths(self):
        return (self.offsets[1:] - self.offsets[:-1]).astype(np.int64)

    def default_names(self, prefix='SYN'):
        if self.names is None:
            self.names = ['%s%05d_%d_%d_%d_%d' % (prefix, s, p + 1, p, 100 + 1000 * p if st else 999 + 1000 * p,
                                                 999 + 1000 * p if st else 100 + 1000 * p)
                          for s, p, st in zip(self.set_id, self.pos_in_set, self.strand)]
        if self.sources is None:
            self.sources = ['%s%05d.faa' % (prefix, s) for s in range(self.n_sets)]


def _pack_strings(strs):
    off = np.zeros(len(strs) + 1, np.uint64)
    np.cumsum([len(s) for s in strs], out=off[1:])
    return ''.join(strs).encode(), off


class _BorrowedContext(api.Context):
    """an sd_ctx owned by an sd_search object (sd_search_ctx): same calls, never destroyed from here"""

    def __init__(self, handle, device):
        self.L = _lib.load()
        self.h = handle
        self.device_index = device

    def __del__(self):
        pass


_STAT_NAMES = ('kmers', 'index_hits', 'diagonals', 'diag_len', 'prefilter_hits', 'pairs', 'cells_fwd', 'cells_rev', 'cells_tb',
               'index_entries', 'masked_residues', 'k', 'kmer_thr', 'bin_size')
_TIME_NAMES = ('index_build_s', 'upload_s', 'bias', 'prefilter', 'pairs', 'seqset', 'align', 'aggregate', 'aggregate_busy',
               'clusterhits', 'prefilter_wait', 'total',
               # thread CPU seconds of the stage threads themselves (OpenMP workers of the host stages are not in them)
               'cpu_bias_thread', 'cpu_prefilter_lanes', 'cpu_align_lanes', 'cpu_aggregate_and_driver')


def _setdb_struct(db, keep):
    """sd_setdb view of a SetDB; `keep` collects the arrays that must outlive the call"""
    v = _lib.SetDbView()
    v.residues, v.offsets, v.n = ptr(db.residues), ptr(db.offsets), db.n
    v.setId, v.posInSet, v.strand, v.nSets = ptr(db.set_id), ptr(db.pos_in_set), ptr(db.strand), db.n_sets
    v.keys = None
    if db.profile is not None:
        aln = np.ascontiguousarray(db.profile['aln'], np.int8)
        ss = np.ascontiguousarray(db.profile['sorted_score'], np.int16)
        si = np.ascontiguousarray(db.profile['sorted_index'], np.uint8)
        keep += [aln, ss, si]
        v.alnProfile, v.sortedScore, v.sortedIndex = ptr(aln), ptr(ss), ptr(si)
    else:
        v.alnProfile = v.sortedScore = v.sortedIndex = None
    keep.append(db)
    return v


class _BorrowedTarget:
    """the sd_target owned by an sd_search object (sd_search_target): never destroyed from here"""

    def __init__(self, ctx, handle, k):
        self.ctx, self.h, self.k = ctx, handle, k

    def sample_check(self, *a, **kw):
        return api.Target.sample_check(self, *a, **kw)


class ClusterSearch:
    """One GPU's worth of the workflow: a thin view of the C++ pipeline object sd_search (csrc/host/sd_search.cpp), which
    owns the device contexts, the resident target (index + sequences) and the stage threads.  Python only hands over the
    set DB arrays and reads the results back."""

    def __init__(self, ctx, host, target_db, sensitivity=5.7, max_seqs=300, eval_thr=10.0, cov_mode=2, cov_thr=0.8,
                 aln_len_thr=30, max_gene_gap=3, cluster_size=2, alpha=1.0, p_clu_thr=0.01, p_mh_thr=0.01,
                 filter_self_match=False, bin_size=None, verbose=False, align_ctx=None, k=None, profile_queries=False,
                 device_bias=None, chunk_queries=0, index=None):
        """ctx / host: the caller's context and host handle (used for the device index and for helper calls such as
        Host.map_profiles); the pipeline object creates its own two contexts on that device -- prefilter and alignments
        run on separate HIP streams so that the prefilter of the next chunk (HBM random-access bound) and the
        Smith-Waterman of the current one (integer-VALU bound) share the GPU instead of taking turns."""
        self.L = _lib.load()
        self.host, self.T = host, target_db
        self.verbose = verbose
        p = _lib.SearchParams()
        self.L.sd_search_default_params(C.byref(p))
        p.sensitivity, p.kmerSize, p.maxSeqs = sensitivity, int(k) if k else 0, max_seqs
        p.binSize = int(bin_size) if bin_size else 0
        p.evalThr, p.covMode, p.covThr, p.alnLenThr = eval_thr, cov_mode, cov_thr, aln_len_thr
        p.maxGeneGap, p.clusterSize, p.alpha, p.pCluThr, p.pMHThr = max_gene_gap, cluster_size, alpha, p_clu_thr, p_mh_thr
        p.filterSelfMatch = 1 if filter_self_match else 0
        p.profileQueries = 1 if profile_queries else 0
        p.chunkQueries = chunk_queries
        p.deviceBias = -1 if device_bias is None else (1 if device_bias else 0)
        p.threads = host.threads
        p.alignPriority = int(os.environ.get('SD_ALIGN_PRIO', '1'))
        self.par = p
        self._keep = []
        tv = _setdb_struct(target_db, self._keep)
        h = C.c_void_p()
        if index is not None:
            # a target index that exists already (api.HostIndex / api.IndexArrays: built once and shared, or read from a file)
            iv = _lib.IndexView()
            iv.kmerSize, iv.kmerThr = index.k, index.kmer_thr
            iv.kmerOffsets, iv.entrySeq, iv.entryPos = ptr(index.kmer_offsets), ptr(index.entry_seq), ptr(index.entry_pos)
            iv.nEntries, iv.maskedResidues, iv.nMaskedResidues = index.n_entries, ptr(index.masked), index.masked_residues
            bb = getattr(index, 'block_base', None)
            iv.kmerBlockBase = ptr(bb) if bb is not None else None
            p.kmerSize = index.k
            rc = self.L.sd_search_create_indexed(ctx.device_index, C.byref(p), C.byref(tv), C.byref(iv), C.byref(h))
        else:
            rc = self.L.sd_search_create(ctx.device_index, C.byref(p), C.byref(tv), C.byref(h))
        if rc != 0:
            raise _lib.SdError('sd_search_create failed (%d)' % rc)
        self.h = h
        self.ctx = _BorrowedContext(C.c_void_p(self.L.sd_search_ctx(h, 0)), ctx.device_index)      # prefilter
        self.ctx_al = _BorrowedContext(C.c_void_p(self.L.sd_search_ctx(h, 1)), ctx.device_index)   # alignments (lane 0)
        # every device context the pipeline runs on (prefilter, alignment lanes, composition bias, clusterhits)
        self.contexts = []
        for which in range(10):
            hp = self.L.sd_search_ctx(h, which)
            if hp:
                self.contexts.append(_BorrowedContext(C.c_void_p(hp), ctx.device_index))
        self.profile_queries = bool(profile_queries)
        self.max_seqs = max_seqs
        self.filter_self_match = filter_self_match
        self._stats0 = np.zeros(16, np.uint64)
        st, tm = self._raw_stats()
        self.k, self.kmer_thr, self.bin_size = int(st[11]), int(st[12]), int(st[13])
        self.index_entries, self.masked_residues = int(st[9]), int(st[10])
        self.timing = dict(index_build_s=float(tm[0]), upload_s=float(tm[1]))
        self.stats = _Stats(self)

    def target_view(self):
        """the search's resident target (borrowed), e.g. for sample_check against the host index builder"""
        return _BorrowedTarget(self.ctx, C.c_void_p(self.L.sd_search_target(self.h)), self.k)

    def download_bytes(self):
        """(record bytes, backtrace-pool bytes) the alignment lanes have copied to the host since create (sd_search_download_bytes)"""
        a, b = C.c_uint64(), C.c_uint64()
        self.L.sd_search_download_bytes(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def _raw_stats(self):
        st = np.zeros(16, np.uint64)
        tm = np.zeros(16, np.float64)
        self.L.sd_search_stats(self.h, ptr(st), ptr(tm))
        return st, tm

    def __del__(self):
        try:
            self.L.sd_search_destroy(self.h)
        except Exception:
            pass

    def search(self, Q, same_db=False, chunk_queries=None, tsv_path=None, canonical=True, query_range=None):
        """run the workflow for query set DB Q (optionally only proteins [a,b) = a shard of whole query sets)."""
        rng = query_range if query_range is not None else (0, Q.n)
        return self.search_stream(Q, [rng], same_db=same_db, chunk_queries=chunk_queries, tsv_paths=[tsv_path],
                                  canonical=canonical)[0]

    def search_stream(self, Q, ranges, same_db=False, chunk_queries=None, tsv_paths=None, canonical=True, want_records=False, arrays='all',
                      records_buffer=None, records_sink=None):
        """The workflow for several query ranges [a,b) of Q (whole query sets each), streamed through one pipeline
        (sd_search_stream): the prefilter of the next chunk -- of the same or of the next range -- overlaps the alignments
        of the current one.  Every range gets its own aggregation, clusterhits call and result record.  Returns the list
        of result dicts (stage timings, summed over the stream, ride on the last).
        arrays: which ranges' result arrays (entries, hits, P-values, clusters) are copied out of the library's result handles into numpy
        arrays -- 'all', or 'last' (the other ranges return their counters only: a caller that streams many ranges and reads one).
        records_sink: (function pointer, user pointer) of an sd_records_sink (sd_search_set_records_sink), e.g. RcclGather.stream_sink():
        every range's records go there the moment the range is finalised (with want_records); they are then not copied out again."""
        L = self.L
        if chunk_queries:
            L.sd_search_set_chunk_queries(self.h, int(chunk_queries))
        keep = []
        qv = _setdb_struct(Q, keep)
        n = len(ranges)
        rb = np.array([r[0] for r in ranges], np.uint32)
        re = np.array([r[1] for r in ranges], np.uint32)
        handles = (C.c_void_p * max(n, 1))()
        _, tm0 = self._raw_stats()
        # the ranges' cluster records are built inside the stream, not behind it (a multi-GPU rank's hand-over to the final gather)
        L.sd_search_set_want_records(self.h, 1 if want_records else 0)
        if records_sink is not None:
            L.sd_search_set_records_sink(self.h, records_sink[0], records_sink[1])
        try:
            rc = L.sd_search_stream(self.h, C.byref(qv), 1 if same_db else 0, n, ptr(rb), ptr(re), handles)
        finally:
            if records_sink is not None:
                L.sd_search_set_records_sink(self.h, None, None)
                if want_records is True:
                    want_records = 'build'   # (the sink has them)
        if rc != 0:
            raise _lib.SdError('sd_search_stream failed (%d): %s' % (rc, L.sd_search_last_error(self.h).decode(errors='replace')))
        _, tm1 = self._raw_stats()
        tsv_paths = tsv_paths if tsv_paths is not None else [None] * n
        results = []
        records_all, rec_at = None, None
        # want_records='build': the records are built inside the stream and stay in the result handles (a single rank has nobody to send
        # them to: it does the same work as a rank of a multi-GPU run up to the hand-over to the gather)
        if want_records == 'build':
            want_records = False
        if want_records:   # one buffer for the records of all ranges, in range order: what sd_gather_results sends, without another copy
            sizes = []
            for ri in range(n):
                need = C.c_uint64()
                api._check(None, L.sd_search_result_records(C.c_void_p(handles[ri]), None, 0, C.byref(need)), 'sd_search_result_records')
                sizes.append(int(need.value))
            rec_at = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            # records_buffer: where the caller wants them (a rank of a multi-GPU run: the communicator's pinned buffer, RcclGather.host_buffer(0, ...))
            if records_buffer is not None and records_buffer.nbytes >= int(rec_at[-1]):
                records_all = records_buffer[:int(rec_at[-1])]
            else:
                records_all = np.empty(int(rec_at[-1]), np.uint8)
        rec_copy_s = 0.0   # seconds spent copying the records out of the result handles (behind the stream: part of a rank's hand-over)
        import time as _time
        for ri in range(n):
            h = C.c_void_p(handles[ri])
            cnt = np.zeros(8, np.uint64)
            L.sd_search_result_counts(h, ptr(cnt))
            ne, nh = int(cnt[0]), int(cnt[1])
            if arrays == 'last' and ri != n - 1 and tsv_paths[ri] is None:   # counters (and records) only
                records = None
                if want_records:
                    records = records_all[int(rec_at[ri]):int(rec_at[ri + 1])]
                    if records.size:
                        need = C.c_uint64()
                        t_c = _time.time()
                        api._check(None, L.sd_search_result_records(h, ptr(records), records.nbytes, C.byref(need)), 'sd_search_result_records')
                        rec_copy_s += _time.time() - t_c
                L.sd_search_result_destroy(h)
                results.append(dict(records=records, entries=ne, matched_hits=nh, clusters=int(cnt[2]), cluster_hits=int(cnt[3]), aligned=int(cnt[4]),
                                    accepted=int(cnt[5]), prefilter_hits=int(cnt[6]), timing={}, cluster_out=None))
                continue
            eo = np.zeros(ne + 1, np.uint64)
            eq, et = np.zeros(max(ne, 1), np.uint32), np.zeros(max(ne, 1), np.uint32)
            hq, ht, pv = np.zeros(max(nh, 1), np.uint32), np.zeros(max(nh, 1), np.uint32), np.zeros(max(nh, 1), np.float64)
            out = dict(cluster_of=np.full(max(nh, 1), 0xFFFFFFFF, np.uint32), rank=np.zeros(max(nh, 1), np.uint32),
                       n_clusters=np.zeros(max(ne, 1), np.uint32), pCO=np.zeros(max(nh, 1), np.float64),
                       pMH=np.zeros(max(nh, 1), np.float64), size=np.zeros(max(nh, 1), np.uint32))
            L.sd_search_result_arrays(h, ptr(eo), ptr(eq), ptr(et), ptr(hq), ptr(ht), ptr(pv), ptr(out['cluster_of']), ptr(out['rank']),
                                      ptr(out['n_clusters']), ptr(out['pCO']), ptr(out['pMH']), ptr(out['size']))
            out = {k_: (v_[:ne] if k_ == 'n_clusters' else v_[:nh]) for k_, v_ in out.items()}
            if tsv_paths[ri] is not None:
                Q.default_names()
                self.T.default_names()
                qn, qno = _pack_strings(Q.names)
                tn, tno = _pack_strings(self.T.names)
                qs, qso = _pack_strings(Q.sources)
                ts, tso = _pack_strings(self.T.sources)
                nc, nhl = C.c_uint64(), C.c_uint64()
                api._check(None, L.sd_search_result_write_tsv(h, tsv_paths[ri].encode(), qn, ptr(qno), tn, ptr(tno), qs, ptr(qso), ts,
                                                             ptr(tso), 1 if canonical else 0, 0, 0, C.byref(nc), C.byref(nhl)),
                           'sd_search_result_write_tsv')
            records = None
            if want_records:   # the cluster records of the range: what a rank sends to the root (sd_search_result_records)
                records = records_all[int(rec_at[ri]):int(rec_at[ri + 1])]
                if records.size:
                    need = C.c_uint64()
                    t_c = _time.time()
                    api._check(None, L.sd_search_result_records(h, ptr(records), records.nbytes, C.byref(need)), 'sd_search_result_records')
                    rec_copy_s += _time.time() - t_c
            L.sd_search_result_destroy(h)
            results.append(dict(records=records, entries=ne, matched_hits=nh, clusters=int(cnt[2]), cluster_hits=int(cnt[3]), aligned=int(cnt[4]),
                                accepted=int(cnt[5]), prefilter_hits=int(cnt[6]), timing={}, entry_q=eq[:ne], entry_t=et[:ne],
                                entry_off=eo, cluster_out=out if nh > 0 else None, hit_q=hq[:nh], hit_t=ht[:nh], hit_pval=pv[:nh]))
        if results:
            d = tm1 - tm0
            results[-1]['timing'] = {name: float(d[i]) for i, name in enumerate(_TIME_NAMES) if i >= 2 and name != 'total'}
            results[-1]['timing']['total'] = float(tm1[11])
            results[-1]['records_copy_s'] = rec_copy_s
            results[-1]['records_all'] = records_all   # (want_records) the ranges' records back to back; every result's 'records' is a view of it
        return results


class _Stats(dict):
    """counters of the pipeline object since the last reset (dict view of sd_search_stats)"""

    def __init__(self, cs):
        super().__init__()
        self._cs = cs
        self._base = np.zeros(16, np.uint64)
        self._refresh()

    def _refresh(self):
        st, _ = self._cs._raw_stats()
        for i, name in enumerate(_STAT_NAMES[:9]):
            dict.__setitem__(self, name, int(st[i]) - int(self._base[i]))

    def __getitem__(self, k):
        self._refresh()
        return dict.__getitem__(self, k)

    def __setitem__(self, k, v):
        """stats[name] = 0 resets the counter"""
        st, _ = self._cs._raw_stats()
        self._base[_STAT_NAMES.index(k)] = st[_STAT_NAMES.index(k)] - np.uint64(v)
        self._refresh()

    def __iter__(self):
        return iter(_STAT_NAMES[:9])

    def keys(self):
        return list(_STAT_NAMES[:9])


def shard_query_sets(set_residues, world, rank):
    """Whole query genome sets -> ranks, greedy by residue count (largest first), so a set's hits stay on one rank
    through besthitbyset -> combinehits -> clusterhits (they group by query set).  Deterministic:
    every rank computes the same assignment and keeps its own part (sorted).  sd_shard_query_sets of the C ABI."""
    L = _lib.load()
    res = np.ascontiguousarray(set_residues, np.uint64)
    mine = np.zeros(max(len(res), 1), np.uint32)
    n = C.c_uint32()
    api._check(None, L.sd_shard_query_sets(ptr(res), len(res), world, rank, ptr(mine), C.byref(n)), 'sd_shard_query_sets')
    return [int(x) for x in mine[:n.value]]


class RcclGather:
    """The C ABI's multi-GPU seam (sd_comm_*): one RCCL communicator per rank, used for the one exchange of the path, the
    final gather of the per-rank result records to rank 0.  unique_id: the 128 bytes of sd_comm_unique_id from rank 0
    (callers broadcast them, e.g. over torch.distributed or a file)."""

    @staticmethod
    def unique_id():
        L = _lib.load()
        b = C.create_string_buffer(128)
        api._check(None, L.sd_comm_unique_id(b), 'sd_comm_unique_id')
        return b.raw

    def __init__(self, device, world, rank, unique_id):
        self.L = _lib.load()
        self.world, self.rank = world, rank
        h = C.c_void_p()
        api._check(None, self.L.sd_comm_init(device, world, rank, unique_id, C.byref(h)), 'sd_comm_init')
        self.h = h

    def host_buffer(self, which, nbytes):
        """a pinned host buffer of the communicator as a numpy uint8 view (sd_comm_host_buffer): which = 0 this rank's records (build them
        in it: search_stream(records_buffer=...)), 1 the gathered records on the root (gather_bytes(out=...)).  The view is valid until the
        next call for the same buffer with a larger size, or the communicator's end."""
        p = C.c_void_p()
        api._check(None, self.L.sd_comm_host_buffer(self.h, int(which), int(nbytes), C.byref(p)), 'sd_comm_host_buffer')
        if not nbytes or not p.value:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(p.value))

    def gather_bytes(self, local, root=0, out=None):
        """byte records of every rank -> (concatenated bytes in rank order, sizes per rank) on `root`, (None, sizes) elsewhere.
        out (root; every rank must pass one or none alike -- the calls are collective): a buffer the gathered bytes land in when it is
        large enough, e.g. host_buffer(1, ...): one exchange, no size probe, no fresh array"""
        rec = np.ascontiguousarray(local, np.uint8).reshape(-1)
        sizes = np.zeros(self.world, np.uint64)
        total = C.c_uint64()
        if out is not None:
            cap = out.nbytes if self.rank == root else 0
            rc = self.L.sd_gather_results(self.h, ptr(rec) if rec.size else None, rec.nbytes, root, ptr(sizes), ptr(out) if cap else None, cap,
                                          C.byref(total))
            if rc == 0:
                return (out[:int(total.value)] if self.rank == root else None), sizes
            if rc != _lib.SD_ENOMEM:   # (SD_ENOMEM: the root's buffer was too small -- every rank saw that, nothing was exchanged: go on below)
                raise _lib.SdError('sd_gather_results failed (%d): %s' % (rc, self.L.sd_comm_last_error(self.h).decode(errors='replace')))
            out = np.empty(int(total.value) if self.rank == root else 0, np.uint8)
            rc = self.L.sd_gather_results(self.h, ptr(rec) if rec.size else None, rec.nbytes, root, ptr(sizes), ptr(out) if out.size else None,
                                          out.nbytes, C.byref(total))
            if rc != 0:
                raise _lib.SdError('sd_gather_results failed (%d): %s' % (rc, self.L.sd_comm_last_error(self.h).decode(errors='replace')))
            return (out if self.rank == root else None), sizes
        out = np.zeros(0, np.uint8)
        rc = self.L.sd_gather_results(self.h, ptr(rec) if rec.size else None, rec.nbytes, root, ptr(sizes), None, 0, C.byref(total))
        if rc == _lib.SD_ENOMEM:   # size probe: every rank learns that the root needs room (nothing was exchanged)
            out = np.empty(int(total.value) if self.rank == root else 0, np.uint8)
            rc = self.L.sd_gather_results(self.h, ptr(rec) if rec.size else None, rec.nbytes, root, ptr(sizes), ptr(out) if out.size else None,
                                          out.nbytes, C.byref(total))
        if rc != 0:
            raise _lib.SdError('sd_gather_results failed (%d): %s' % (rc, self.L.sd_comm_last_error(self.h).decode(errors='replace')))
        return (out if self.rank == root else None), sizes

    def stream_begin(self, round_of_range, n_rounds, out=None, root=0):
        """sd_gather_stream_begin: the gather round by round behind a running search.  round_of_range[i]: the round of this rank's i-th
        range (non-decreasing); n_rounds the same on every rank; out (root): where the rounds land back to back, e.g. host_buffer(1, ...).
        Hand stream_sink() to search_stream(records_sink=...), then stream_end()."""
        rr = np.ascontiguousarray(round_of_range, np.uint32)
        g = C.c_void_p()
        cap = out.nbytes if (out is not None and self.rank == root) else 0
        self._stream_out = out
        # (out is None -- on every rank alike -- : the root's buffer belongs to the stream; this wrapper always brings one)
        api._check(None, self.L.sd_gather_stream_begin(self.h, root, len(rr), ptr(rr) if len(rr) else None, int(n_rounds), ptr(out) if cap else None, cap,
                                                       0, C.byref(g)), 'sd_gather_stream_begin')
        self._stream = (g, int(n_rounds), root)
        return g

    def stream_sink(self):
        """(function pointer, user pointer) for search_stream(records_sink=...)"""
        return C.cast(self.L.sd_gather_stream_sink, C.c_void_p), self._stream[0]

    def stream_end(self):
        """waits for the last round; -> (gathered bytes on the root / None, round offsets [n_rounds + 1], sizes [n_rounds, world])"""
        g, n_rounds, root = self._stream
        self._stream = None
        offs = np.zeros(n_rounds + 1, np.uint64)
        sizes = np.zeros((n_rounds, self.world), np.uint64)
        total = C.c_uint64()
        rc = self.L.sd_gather_stream_end(g, ptr(offs), ptr(sizes), C.byref(total))
        if rc != 0:
            raise _lib.SdError('sd_gather_stream_end failed (%d): %s' % (rc, self.L.sd_comm_last_error(self.h).decode(errors='replace')))
        out = self._stream_out
        return (out[:int(total.value)] if (self.rank == root and out is not None) else None), offs, sizes

    def gather(self, local_records, root=0):
        """variable-length int64 record arrays -> list of per-rank arrays on `root` (None elsewhere)"""
        rec = np.ascontiguousarray(local_records, np.int64).reshape(-1)
        sizes = np.zeros(self.world, np.uint64)
        total = C.c_uint64()
        out = np.zeros(0, np.uint8)
        rc = self.L.sd_gather_results(self.h, ptr(rec) if rec.size else None, rec.nbytes, root, ptr(sizes), None, 0, C.byref(total))
        if rc == _lib.SD_ENOMEM:
            pass   # size probe: the first call tells every rank how much room the root needs (nothing was exchanged)
        elif rc != 0:
            raise _lib.SdError('sd_gather_results failed (%d): %s' % (rc, self.L.sd_comm_last_error(self.h).decode(errors='replace')))
        if total.value:
            out = np.zeros(int(total.value), np.uint8)
            rc = self.L.sd_gather_results(self.h, ptr(rec) if rec.size else None, rec.nbytes, root, ptr(sizes), ptr(out), out.nbytes,
                                          C.byref(total))
            if rc != 0:
                raise _lib.SdError('sd_gather_results failed (%d): %s' % (rc, self.L.sd_comm_last_error(self.h).decode(errors='replace')))
        if self.rank != root:
            return None
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        return [out[offs[r]:offs[r + 1]].view(np.int64) for r in range(self.world)]

    def __del__(self):
        try:
            self.L.sd_comm_destroy(self.h)
        except Exception:
            pass


def write_records_tsv(records, path, Q, T, canonical=False, first_cluster_key=0):
    """the TSV of cluster records (one rank's, or the gathered buffer of all ranks): sd_records_write_tsv; returns (#clusters, #hits)"""
    L = _lib.load()
    Q.default_names()
    T.default_names()
    qn, qno = _pack_strings(Q.names)
    tn, tno = _pack_strings(T.names)
    qs, qso = _pack_strings(Q.sources)
    ts, tso = _pack_strings(T.sources)
    rec = np.ascontiguousarray(records, np.uint8).reshape(-1)
    nc, nh = C.c_uint64(), C.c_uint64()
    api._check(None, L.sd_records_check(ptr(rec) if rec.size else None, rec.nbytes, len(qso) - 1, len(tso) - 1, len(qno) - 1, len(tno) - 1, None, None),
               'sd_records_check (truncated records, or indices outside the name tables)')
    api._check(None, L.sd_records_write_tsv(ptr(rec) if rec.size else None, rec.nbytes, str(path).encode(), 0, first_cluster_key, qn, ptr(qno), tn,
                                            ptr(tno), qs, ptr(qso), ts, ptr(tso), 1 if canonical else 0, C.byref(nc), C.byref(nh)),
               'sd_records_write_tsv')
    return int(nc.value), int(nh.value)


class TcpGather:
    """The gather round by round (sd_gather_stream_*) over the C ABI's TCP rendezvous instead of RCCL: for ranks that share a device
    (RCCL refuses two ranks per GPU) -- the one-GPU rehearsal of bench.py --gpus N.  Same stream_* surface as RcclGather; rank 0 is the root."""

    def __init__(self, world, rank, addr, port):
        self.L = _lib.load()
        self.world, self.rank = world, rank
        h = C.c_void_p()
        api._check(None, self.L.sd_tcp_connect(addr.encode(), int(port), world, rank, C.byref(h)), 'sd_tcp_connect')
        self.h = h
        self._stream = None

    def stream_begin(self, round_of_range, n_rounds, out=None, root=0):
        rr = np.ascontiguousarray(round_of_range, np.uint32)
        g = C.c_void_p()
        cap = out.nbytes if (out is not None and self.rank == 0) else 0
        self._stream_out = out
        api._check(None, self.L.sd_gather_stream_begin_tcp(self.h, self.world, self.rank, len(rr), ptr(rr) if len(rr) else None, int(n_rounds),
                                                           ptr(out) if cap else None, cap, 0, C.byref(g)), 'sd_gather_stream_begin_tcp')
        self._stream = (g, int(n_rounds))
        return g

    def stream_sink(self):
        return C.cast(self.L.sd_gather_stream_sink, C.c_void_p), self._stream[0]

    def stream_end(self):
        g, n_rounds = self._stream
        self._stream = None
        offs = np.zeros(n_rounds + 1, np.uint64)
        sizes = np.zeros((n_rounds, self.world), np.uint64)
        total = C.c_uint64()
        rc = self.L.sd_gather_stream_end(g, ptr(offs), ptr(sizes), C.byref(total))
        if rc != 0:
            raise _lib.SdError('sd_gather_stream_end (TCP) failed (%d)' % rc)
        out = self._stream_out
        return (out[:int(total.value)] if (self.rank == 0 and out is not None) else None), offs, sizes

    def __del__(self):
        try:
            if self.h:
                self.L.sd_tcp_close(self.h)
        except Exception:
            pass


def gather_results(local_records, dist, device=None):
    """Final result gather (the only collective on the path): variable-length int64 record arrays from every rank
    to all ranks -- all_gather of the lengths, then of the padded payloads (RCCL on GPUs, gloo in the CPU tests)."""
    import torch
    world = dist.get_world_size()
    rec = torch.from_numpy(np.array(local_records, np.int64, copy=True).reshape(-1))
    if device is not None:
        rec = rec.to(device)
    n = torch.tensor([rec.numel()], dtype=torch.int64, device=rec.device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    m = int(max(int(x.item()) for x in lens))
    pad = torch.zeros(max(m, 1), dtype=torch.int64, device=rec.device)
    pad[:rec.numel()] = rec
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return [b[:int(l.item())].cpu().numpy() for b, l in zip(bufs, lens)]
