"""clustersearch --search-mode 0 on one GPU: prefilter -> align -> (host aggregation) -> clusterhits -> TSV.

Mirrors R/data/clustersearch.sh:110-152 (`search` = prefilter + align, then besthitbyset / mergeresultsbyset /
combinehits fused in sd_agg, `clusterhits`, `summarizeresults`).  All hot-path compute goes through the C ABI of
libsdgpu.so (HIP); this file only moves buffers and sequences the stages."""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

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


class ClusterSearch:
    """One GPU's worth of the workflow.  The target side (index + sequences) stays resident in HBM; query
    sets are processed in chunks of whole query proteins."""

    def __init__(self, ctx, host, target_db, sensitivity=5.7, max_seqs=300, eval_thr=10.0, cov_mode=2, cov_thr=0.8,
                 aln_len_thr=30, max_gene_gap=3, cluster_size=2, alpha=1.0, p_clu_thr=0.01, p_mh_thr=0.01,
                 filter_self_match=False, bin_size=None, verbose=False, align_ctx=None, k=None, profile_queries=False,
                 device_bias=None):
        """ctx runs the prefilter (and clusterhits); align_ctx -- a second context (own HIP stream and workspace)
        on the same device, created here if not given -- runs the alignments, so that the prefilter of the next
        chunk (HBM random-access bound) and the Smith-Waterman of the current one (integer-VALU bound) share the
        GPU instead of taking turns."""
        self.ctx, self.host, self.T = ctx, host, target_db
        self.ctx_al = align_ctx if align_ctx is not None else api.Context(ctx.device_index, priority=int(os.environ.get('SD_ALIGN_PRIO', '1')))
        self.verbose = verbose
        # composition bias of the queries on the device (own context and stream) when host cores are scarce (a rank's share
        # of a multi-GPU node): same values bit for bit (sd_comp_bias_batch), ~0.5 core-seconds per 30 000 queries saved
        if device_bias is None:
            from .cpus import effective_cpus
            device_bias = effective_cpus() // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))) < 8
        self.ctx_bias = api.Context(ctx.device_index) if device_bias else None
        # -k 0 semantics (IndexTable::computeKmerSize, IndexTable.h:439-441): 6 below 3.35e9 target residues, 7 from there on
        self.k = int(k) if k else host.auto_kmer_size(int(target_db.offsets[-1]))
        # profile searches: own threshold table, and the target index keeps every non-X k-mer (Prefiltering.cpp:525-527,1019-1043)
        self.profile_queries = bool(profile_queries)
        self.kmer_thr = (host.profile_kmer_threshold(sensitivity, self.k) if self.profile_queries
                         else host.kmer_threshold(sensitivity, self.k))
        self.max_seqs = max_seqs
        self.eval_thr, self.cov_mode, self.cov_thr, self.aln_len_thr = eval_thr, cov_mode, cov_thr, aln_len_thr
        self.ch = dict(max_gene_gap=max_gene_gap, cluster_size=cluster_size, alpha=alpha, p_clu_thr=p_clu_thr,
                       p_mh_thr=p_mh_thr)
        self.filter_self_match = filter_self_match
        self.timing = {}
        t0 = time.time()
        self.t_sw_bias, _, _ = host.comp_bias(target_db.residues, target_db.offsets, self.k)
        self.index = host.build_index(target_db.residues, target_db.offsets, self.k, 0 if self.profile_queries else self.kmer_thr)
        self.timing['index_build_s'] = time.time() - t0
        t0 = time.time()
        self.target = api.Target(ctx, host, self.index)
        self.t_seqs = self.ctx_al.seqset(target_db.residues, target_db.offsets, self.t_sw_bias)
        self.timing['upload_s'] = time.time() - t0
        self.pf_par = api.prefilter_params(host, target_db.n, kmer_thr=self.kmer_thr, max_hits=max_seqs, bin_size=bin_size,
                                           cov_mode=cov_mode, cov_thr=cov_thr, k=self.k)
        mat, _, _ = host.matrix(0)
        self.sw_par = self.ctx_al.sw_params(mat, int(target_db.offsets[-1]), sw_mode=2, eval_thr=eval_thr, cov_mode=cov_mode,
                                    cov_thr=cov_thr)
        self.stats = dict(prefilter_hits=0, pairs=0, cells_fwd=0, cells_rev=0, cells_tb=0, kmers=0, index_hits=0,
                          diagonals=0, diag_len=0)

    def search(self, Q, same_db=False, chunk_queries=10000, tsv_path=None, canonical=True, query_range=None):
        """run the workflow for query set DB Q (optionally only proteins [a,b) = a shard of whole query sets)."""
        rng = query_range if query_range is not None else (0, Q.n)
        return self.search_stream(Q, [rng], same_db=same_db, chunk_queries=chunk_queries, tsv_paths=[tsv_path],
                                  canonical=canonical)[0]

    def search_stream(self, Q, ranges, same_db=False, chunk_queries=10000, tsv_paths=None, canonical=True):
        """The workflow for several query ranges [a,b) of Q (whole query sets each), streamed through one pipeline:
        the prefilter of the next chunk -- of the same or of the next range -- overlaps the alignments of the current
        one.  Every range gets its own aggregation, clusterhits call and result record, exactly as separate search()
        calls would produce.  Returns the list of result dicts (stage timings, summed over the stream, ride on the last)."""
        L = self.ctx.L
        T = self.T
        t_all = time.time()
        tl = T.lengths().astype(np.int32)
        qlens = Q.lengths().astype(np.int32)
        tsv_paths = tsv_paths if tsv_paths is not None else [None] * len(ranges)
        aggs = []
        for _ in ranges:
            agg = C.c_void_p()
            api._check(None, L.sd_agg_create(ptr(Q.set_id), ptr(qlens), Q.n, ptr(T.set_id), ptr(tl), T.n, Q.n_sets, T.n_sets,
                                             self.eval_thr, self.cov_mode, self.cov_thr, self.aln_len_thr,
                                             1 if self.filter_self_match else 0, C.byref(agg)), 'sd_agg_create')
            aggs.append(agg)
        tm = dict(prefilter=0.0, align=0.0, aggregate=0.0, clusterhits=0.0, bias=0.0, aggregate_busy=0.0)
        # three threads: bias + prefilter + pair list of chunk i+1 (context A) | alignments of chunk i (context B, this
        # thread) | host aggregation of chunk i-1.  ctypes releases the GIL; one aggregation job in flight keeps the
        # order of sd_agg_add calls and bounds memory (the alignment results live in two alternating buffers).
        pool_exec = ThreadPoolExecutor(max_workers=1)
        pending = None

        def aggregate_job(agg, n_pairs, pair_q_local, pair_t, r, identity, pool, c0):
            t1 = time.time()
            idt = np.ascontiguousarray(identity, np.uint8)
            api._check(None, L.sd_agg_add(agg, n_pairs, c0, ptr(pair_q_local), ptr(pair_t), ptr(r), ptr(idt), ptr(pool)),
                       'sd_agg_add')
            return time.time() - t1

        def bias_job(c0, c1):
            """composition bias of one chunk's queries (host, float: SubstitutionMatrix.cpp:79-109); its own thread, so
            that on a box with few cores per GPU it is not serialised with the prefilter calls of the same chunk"""
            r0, r1 = int(Q.offsets[c0]), int(Q.offsets[c1])
            res = Q.residues[r0:r1]
            off = (Q.offsets[c0:c1 + 1] - Q.offsets[c0]).astype(np.uint64)
            t0 = time.time()
            if Q.profile is not None:   # no composition bias for profile queries (QueryMatcher.cpp:93-99, ssw_init :1229-1240)
                return res, off, None, None, None, 0.0
            if self.ctx_bias is not None:
                sw_b, dg_b, km_b = self.ctx_bias.comp_bias(self.host, res, off, self.k)
            else:
                sw_b, dg_b, km_b = self.host.comp_bias(res, off, self.k)
            return res, off, sw_b, dg_b, km_b, time.time() - t0

        def prefilter_job(c0, c1, bias_future):
            """prefilter + pair list of one chunk (runs on its own thread and HIP stream)"""
            t = {}
            res, off, sw_b, dg_b, km_b, t['bias'] = bias_future.result()
            ident = (np.arange(c0, c1, dtype=np.uint32) if same_db else np.full(c1 - c0, 0xFFFFFFFF, np.uint32))
            t0 = time.time()
            if Q.profile is not None:
                r0 = int(Q.offsets[c0])
                r1 = int(Q.offsets[c1])
                pslice = dict(letters=res, offsets=off, aln=Q.profile['aln'][r0:r1], sorted_score=Q.profile['sorted_score'][r0:r1],
                              sorted_index=Q.profile['sorted_index'][r0:r1])
                hits, cnt, st = api.prefilter_profile(self.ctx, self.target, self.pf_par, pslice, identity_id=ident, want_stats=True)
            else:
                hits, cnt, st = api.prefilter(self.ctx, self.target, self.pf_par, res, off, km_b, dg_b, ident, want_stats=True)
            t['prefilter'] = time.time() - t0
            # pair list in prefilter order (Alignment.cpp:346-379); Alignment::run's coverage pre-check (:370-373) is
            # the same test the prefilter applied
            t0 = time.time()
            n_pairs = int(cnt.sum())
            cnt = np.ascontiguousarray(cnt, np.uint32)
            pair_q_local = np.empty(n_pairs, np.uint32)
            pair_t = np.empty(n_pairs, np.uint32)
            if n_pairs:
                L.sd_host_pair_list(ptr(hits), ptr(cnt), c1 - c0, hits.shape[1], ptr(pair_q_local), ptr(pair_t))
            t['pairs'] = time.time() - t0
            return dict(c0=c0, c1=c1, res=res, off=off, sw_b=sw_b, st=st, n_pairs=n_pairs, pair_q_local=pair_q_local,
                        pair_t=pair_t, t=t)

        def finalize(ri):
            """aggregation result of range ri -> clusterhits -> result record (alignment thread, context B)"""
            agg = aggs[ri]
            t0 = time.time()
            ne, nh = C.c_uint64(), C.c_uint64()
            L.sd_agg_finish(agg, C.byref(ne), C.byref(nh))
            ne, nh = ne.value, nh.value
            entry_off = np.zeros(ne + 1, np.uint64)
            eq = np.zeros(max(ne, 1), np.uint32)
            et = np.zeros(max(ne, 1), np.uint32)
            hq = np.zeros(max(nh, 1), np.uint32)
            ht = np.zeros(max(nh, 1), np.uint32)
            pv = np.zeros(max(nh, 1), np.float64)
            L.sd_agg_get(agg, ptr(entry_off), ptr(eq), ptr(et), ptr(hq), ptr(ht), ptr(pv))
            hq, ht, pv, eq, et = hq[:nh], ht[:nh], pv[:nh], eq[:ne], et[:ne]
            tm['aggregate'] += time.time() - t0
            t0 = time.time()
            out = None
            n_clusters = n_cluster_hits = 0
            if nh > 0:
                qp = Q.pos_in_set[hq]
                tp = T.pos_in_set[ht]
                sd = (Q.strand[hq] | (T.strand[ht] << 1)).astype(np.uint8)
                nq = Q.set_size[eq]
                lg_n = int(max(int(Q.set_size.max()), int(T.set_size.max()), int(qp.max()), int(tp.max()))) + 8
                out = api.clusterhits(self.ctx_al, self.host, entry_off, qp, tp, sd, pv, nq, lgamma=self.host.lgamma_table(lg_n),
                                      **self.ch)
                n_clusters = int(out['n_clusters'].sum())
                n_cluster_hits = int((out['cluster_of'] != 0xFFFFFFFF).sum())
            tm['clusterhits'] += time.time() - t0
            if tsv_paths[ri] is not None and out is not None:
                Q.default_names()
                T.default_names()
                qn, qno = _pack_strings(Q.names)
                tn, tno = _pack_strings(T.names)
                qs, qso = _pack_strings(Q.sources)
                ts, tso = _pack_strings(T.sources)
                nc, nhl = C.c_uint64(), C.c_uint64()
                api._check(None, L.sd_agg_write_tsv(agg, tsv_paths[ri].encode(), ptr(out['cluster_of']), ptr(out['rank']),
                                                    ptr(out['n_clusters']), ptr(out['pCO']), ptr(out['pMH']), ptr(out['size']),
                                                    qn, ptr(qno), tn, ptr(tno), qs, ptr(qso), ts, ptr(tso),
                                                    1 if canonical else 0, C.byref(nc), C.byref(nhl)), 'sd_agg_write_tsv')
            na, nacc = C.c_uint64(), C.c_uint64()
            L.sd_agg_stats(agg, C.byref(na), C.byref(nacc))
            L.sd_agg_destroy(agg)
            return dict(entries=ne, matched_hits=nh, clusters=n_clusters, cluster_hits=n_cluster_hits, aligned=na.value,
                        accepted=nacc.value, timing={}, entry_q=eq, entry_t=et, entry_off=entry_off, cluster_out=out,
                        hit_q=hq, hit_t=ht, hit_pval=pv)

        chunks = []
        for ri, (a0, b0) in enumerate(ranges):
            c0 = a0
            while c0 < b0:
                # the very first chunk of a stream is a quarter of the others: its prefilter is the one stage nothing
                # overlaps with, so the alignment thread starts that much earlier
                step = chunk_queries if chunks else max(1, min(chunk_queries, max(1000, chunk_queries // 4)))
                chunks.append((ri, c0, min(b0, c0 + step)))
                c0 += step
        last_chunk_of = {}
        for x, (ri, _, _) in enumerate(chunks):
            last_chunk_of[ri] = x
        results = [None] * len(ranges)
        to_finalize = []   # (range, its last aggregation job)
        # stage threads: bias (chunk i+2) | prefilter (chunk i+1) | alignments (chunk i, this thread) | aggregation (chunk i-1)
        pf_exec = ThreadPoolExecutor(max_workers=1)
        bias_exec = ThreadPoolExecutor(max_workers=1)
        bias_fut = {}

        def submit_bias(x):
            if x < len(chunks) and x not in bias_fut:
                bias_fut[x] = bias_exec.submit(bias_job, chunks[x][1], chunks[x][2])

        def submit_prefilter(x):
            submit_bias(x)
            f = pf_exec.submit(prefilter_job, chunks[x][1], chunks[x][2], bias_fut.pop(x))
            submit_bias(x + 1)
            return f

        pf_next = submit_prefilter(0) if chunks else None
        for ci in range(len(chunks)):
            ri = chunks[ci][0]
            t0 = time.time()
            d = pf_next.result()
            tm['prefilter_wait'] = tm.get('prefilter_wait', 0.0) + time.time() - t0
            pf_next = submit_prefilter(ci + 1) if ci + 1 < len(chunks) else None
            for k_, v_ in d['t'].items():
                tm[k_] = tm.get(k_, 0.0) + v_
            st, n_pairs, c0, c1 = d['st'], d['n_pairs'], d['c0'], d['c1']
            self.stats['kmers'] += int(st[:, 0].sum())
            self.stats['index_hits'] += int(st[:, 1].sum())
            self.stats['diagonals'] += int(st[:, 2].sum())
            self.stats['diag_len'] += int(st[:, 3].sum())
            self.stats['prefilter_hits'] += n_pairs
            if n_pairs > 0:
                pair_q_local, pair_t = d['pair_q_local'], d['pair_t']
                t0 = time.time()
                if Q.profile is not None:
                    r0 = int(Q.offsets[c0])
                    qset = self.ctx_al.profileset(d['res'], d['off'], Q.profile['aln'][r0:r0 + len(d['res'])])
                else:
                    qset = self.ctx_al.seqset(d['res'], d['off'], d['sw_b'])
                tm['seqset'] = tm.get('seqset', 0.0) + time.time() - t0
                t0 = time.time()
                identity = (pair_q_local + np.uint32(c0) == pair_t) if same_db else np.zeros(n_pairs, bool)
                # only the reportable pairs come back (identity pairs + pairs past every gate, ~10 %): everything else
                # fails Alignment::checkCriteria and would be skipped by the aggregation anyway
                cidx, r, pool = self.ctx_al.sw_align(self.sw_par, qset, self.t_seqs, pair_q_local, pair_t, identity=identity,
                                                     reuse=True, compact=True)
                n_all = n_pairs
                pair_q_local, pair_t, identity = pair_q_local[cidx], pair_t[cidx], identity[cidx]
                n_pairs = len(cidx)
                tm['align'] += time.time() - t0
                f, rv, tb = self.ctx_al.sw_cells()
                self.stats['cells_fwd'] += f
                self.stats['cells_rev'] += rv
                self.stats['cells_tb'] += tb
                self.stats['pairs'] += n_all
                t0 = time.time()
                if pending is not None:
                    tm['aggregate_busy'] += pending.result()
                pending = pool_exec.submit(aggregate_job, aggs[ri], n_pairs, pair_q_local, pair_t, r, identity, pool, c0)
                tm['aggregate'] += time.time() - t0
                del qset
            if last_chunk_of[ri] == ci:
                to_finalize.append((ri, pending))
            # ranges whose last aggregation job has finished meanwhile
            while to_finalize and (to_finalize[0][1] is None or to_finalize[0][1].done()):
                fri, _ = to_finalize.pop(0)
                results[fri] = finalize(fri)
        pf_exec.shutdown()
        bias_exec.shutdown()
        t0 = time.time()
        if pending is not None:
            tm['aggregate_busy'] += pending.result()
        tm['aggregate'] += time.time() - t0
        pool_exec.shutdown()
        for fri, _ in to_finalize:
            results[fri] = finalize(fri)
        for ri in range(len(ranges)):   # ranges without any chunk
            if results[ri] is None:
                results[ri] = finalize(ri)
        tm['total'] = time.time() - t_all
        if results:
            results[-1]['timing'] = tm
        return results


def shard_query_sets(set_residues, world, rank):
    """Whole query genome sets -> ranks, greedy by residue count (largest first), so a set's hits stay on one rank
    through besthitbyset -> combinehits -> clusterhits (they group by query set).  Deterministic:
    every rank computes the same assignment and keeps its own part (sorted)."""
    order = sorted(range(len(set_residues)), key=lambda s: (-int(set_residues[s]), s))
    load = [0] * world
    mine = []
    for s in order:
        r = min(range(world), key=lambda x: (load[x], x))
        load[r] += int(set_residues[s])
        if r == rank:
            mine.append(s)
    return sorted(mine)


def gather_results(local_records, dist, device=None):
    """Final result gather (the only collective on the path): variable-length int64 record arrays from every rank
    to all ranks -- all_gather of the lengths, then of the padded payloads (RCCL on GPUs, gloo in the CPU tests)."""
    import torch
    world = dist.get_world_size()
    rec = torch.as_tensor(np.ascontiguousarray(local_records, np.int64).reshape(-1))
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
