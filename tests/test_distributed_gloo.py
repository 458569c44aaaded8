"""CPU, world_size 2 over gloo: query-set sharding covers every set exactly once, balances residues, and the
final variable-length result gather returns every rank's records (the N>1 path of bench.py)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spacedust_amd.pipeline import shard_query_sets, gather_results


def _records_for(sets):
    # stand-in per-set result records (set id, #clusters, checksum): the gather does not care what they mean
    return np.array([[s, (s * 7) % 5, s * s + 1] for s in sets], np.int64)


def _worker(rank, world, port, sizes, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = shard_query_sets(sizes, world, rank)
    got = gather_results(_records_for(mine), dist)
    if rank == 0:
        q.put((mine, [g.tolist() for g in got]))
    else:
        q.put((mine, None))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    rng = np.random.default_rng(4)
    sizes = rng.integers(500000, 1500000, size=13).tolist()
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [o[0] for o in outs]
    allsets = sorted(s for sh in shards for s in sh)
    assert allsets == list(range(13))                      # every set exactly once
    loads = [sum(sizes[s] for s in sh) for sh in shards]
    assert max(loads) - min(loads) <= max(sizes)            # greedy LPT bound
    gathered = [o[1] for o in outs if o[1] is not None][0]
    rows = np.concatenate([np.array(g, np.int64).reshape(-1, 3) for g in gathered])
    rows = rows[np.argsort(rows[:, 0])]
    assert (rows == _records_for(range(13))).all()


def test_shard_deterministic_and_single_rank():
    sizes = [5, 9, 1, 7, 7, 3]
    assert shard_query_sets(sizes, 1, 0) == list(range(6))
    a = [shard_query_sets(sizes, 3, r) for r in range(3)]
    assert sorted(sum(a, [])) == list(range(6))
    assert a == [shard_query_sets(sizes, 3, r) for r in range(3)]
