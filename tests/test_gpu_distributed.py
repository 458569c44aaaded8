"""The N>1 path on real hardware as far as one GPU allows: two ranks (gloo rendezvous on 127.0.0.1) share cuda:0, each
runs the pipeline on its shard of whole query sets (shard_query_sets) against the full target DB, the per-entry
result records are gathered (gather_results) -- and must equal the records of one unsharded run: E-values and
combinehits' set-count factor depend only on the global target DB."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(out):
    """one row per (query set, target set) entry: ids, #hits, #clusters, and order-sensitive checksums of the cluster
    assignment and of the P-values' bit patterns"""
    rows = []
    if out['cluster_out'] is None:
        return np.zeros((0, 7), np.int64)
    co = out['cluster_out']
    off = out['entry_off']
    for e in range(len(out['entry_q'])):
        a, b = int(off[e]), int(off[e + 1])
        cl = co['cluster_of'][a:b].astype(np.int64)
        w = np.arange(1, b - a + 1, dtype=np.int64)
        nclu = int(co['n_clusters'][e])
        pbits = np.frombuffer(np.ascontiguousarray(co['pCO'][a:a + nclu]).tobytes(), np.int64)
        rows.append([int(out['entry_q'][e]), int(out['entry_t'][e]), b - a, nclu,
                     int(((cl + 2) * w).sum() % (1 << 40)), int((out['hit_t'][a:b].astype(np.int64) * w).sum() % (1 << 40)),
                     int((pbits % (1 << 40)).sum() % (1 << 40))])
    return np.array(rows, np.int64).reshape(-1, 7)


def _search(sets):
    from spacedust_amd.api import Host, Context
    from spacedust_amd.pipeline import SetDB, ClusterSearch
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(6, genes_per_proteome=150, n_families=220, seed=23)
    db = SetDB.from_proteomes(ps)
    host, gpu = Host(4), Context(0)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
    recs = []
    for s in sets:
        out = cs.search(db, same_db=True, query_range=(int(ps.set_start[s]), int(ps.set_start[s + 1])), chunk_queries=64)
        recs.append(_records(out))
    sizes = [int(ps.offsets[ps.set_start[s + 1]] - ps.offsets[ps.set_start[s]]) for s in range(6)]
    return (np.concatenate(recs) if recs else np.zeros((0, 7), np.int64)), sizes


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from spacedust_amd.pipeline import shard_query_sets, gather_results
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(6, genes_per_proteome=150, n_families=220, seed=23)
    sizes = [int(ps.offsets[ps.set_start[s + 1]] - ps.offsets[ps.set_start[s]]) for s in range(6)]
    mine = shard_query_sets(sizes, world, rank)
    recs, _ = _search(mine)
    got = gather_results(recs, dist)
    if rank == 0:
        q.put((mine, [g.reshape(-1, 7).tolist() for g in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank(gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mine0, gathered = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert 0 < len(mine0) < 6
    sharded = np.array(sorted(r for g in gathered for r in g), np.int64)
    whole, _ = _search(range(6))
    whole = np.array(sorted(whole.tolist()), np.int64)
    assert len(whole) > 10 and whole[:, 3].sum() > 0
    assert sharded.shape == whole.shape and (sharded == whole).all()


def test_rccl_gather_seam_single_rank(gpu):
    """sd_comm_init / sd_gather_results of the C ABI over RCCL with the one rank a one-GPU box allows: communicator from a
    fresh unique id, a variable-length record array goes through the size all_gather and comes back on the root
    (N > 1 runs the same code with grouped ncclSend / ncclRecv; the driver's multi-GPU bench exercises it)"""
    from spacedust_amd.pipeline import RcclGather
    uid = RcclGather.unique_id()
    assert len(uid) == 128 and any(uid)
    g = RcclGather(0, 1, 0, uid)
    recs = np.arange(21, dtype=np.int64).reshape(3, 7) * 1234567
    out = g.gather(recs)
    assert len(out) == 1 and (out[0].reshape(-1, 7) == recs).all()
    out = g.gather(np.zeros((0, 7), np.int64))
    assert len(out) == 1 and out[0].size == 0


def test_sdgpu_clustersearch_two_ranks_equal_one(gpu, tmp_path):
    """the binary's own multi-rank mode (RANK / WORLD_SIZE, whole query sets per rank, parts merged by rank 0): two ranks
    sharing cuda:0 must write the same clusters as one rank"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from dbutil import SDGPU, sdgpu, example_fasta, sorted_md5
    fa = example_fasta(tmp_path)
    q, t = tmp_path / 'q', tmp_path / 't'
    # query DB with two sets (so that two ranks have something each), target DB with both genomes
    sdgpu('createsetdb', fa[0], fa[1], t, tmp_path / 'tmp', '-v', '0')
    sdgpu('createsetdb', fa[1], fa[0], q, tmp_path / 'tmp', '-v', '0')
    sdgpu('clustersearch', q, t, tmp_path / 'one.tsv', tmp_path / 'tmp1', '-v', '0')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0')
        procs.append(subprocess.Popen([SDGPU, 'clustersearch', str(q), str(t), str(tmp_path / 'two.tsv'), str(tmp_path / 'tmp2'), '-v', '0'],
                                      env=env))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    one = open(tmp_path / 'one.tsv').readlines()
    two = open(tmp_path / 'two.tsv').readlines()
    assert len(one) > 100 and sum(1 for l in one if l.startswith('#')) > 20
    assert sorted_md5(one, drop_first_column=True) == sorted_md5(two, drop_first_column=True)
    # cluster keys are consecutive in the merged file
    keys = [int(l.split('\t')[0][1:]) for l in two if l.startswith('#')]
    assert keys == list(range(len(keys)))
