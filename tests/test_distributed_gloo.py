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


def _tcp_worker(rank, world, port, q):
    import ctypes as C
    from spacedust_amd import _lib
    L = _lib.load()
    t = C.c_void_p()
    assert L.sd_tcp_connect(b'127.0.0.1', port, world, rank, C.byref(t)) == 0
    # broadcast of rank 0's 128 bytes (the RCCL unique id travels this way in sdgpu)
    buf = C.create_string_buffer(bytes(range(128)) if rank == 0 else bytes(128), 128)
    assert L.sd_tcp_bcast(t, buf, 128) == 0
    # gatherv of variable-length records to rank 0
    rec = np.arange(5 + 11 * rank, dtype=np.int64) * (rank + 3)
    sizes = np.zeros(world, np.uint64)
    total = C.c_uint64()
    out = np.zeros(int(sum((5 + 11 * r) * 8 for r in range(world))), np.uint8)
    rc = L.sd_tcp_gather(t, rec.ctypes.data_as(C.c_void_p), rec.nbytes, sizes.ctypes.data_as(C.c_void_p),
                         out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(total))
    assert rc == 0
    L.sd_tcp_close(t)
    q.put((rank, buf.raw, out.view(np.int64).tolist() if rank == 0 else None, sizes.tolist() if rank == 0 else None))


def test_tcp_rendezvous_world3():
    """sd_tcp_bcast / sd_tcp_gather of the C ABI (the ranks' meeting point in `sdgpu clustersearch`): three processes on 127.0.0.1"""
    world = 3
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_tcp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] == bytes(range(128)) for o in outs)
    want = np.concatenate([np.arange(5 + 11 * r, dtype=np.int64) * (r + 3) for r in range(world)])
    assert outs[0][2] == want.tolist()
    assert outs[0][3] == [(5 + 11 * r) * 8 for r in range(world)]


def _worker8(rank, world, port, sizes, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    out = []
    for step_sizes in sizes:   # several global steps, as bench.py deals them: some with fewer sets than ranks
        mine = shard_query_sets(step_sizes, world, rank)
        got = gather_results(_records_for(mine), dist)
        out.append((mine, [g.tolist() for g in got] if rank == 0 else None))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world8():
    """the layout the driver's 8-GPU run uses (one rank per GPU): 8 ranks over gloo, a step of 32 query sets (4 per rank, bench.py's
    weak-scaling step), a ragged step of 13 and one of 5 sets (three ranks with nothing to do: empty shards, empty records);
    every set lands on exactly one rank, the loads obey the greedy bound, rank 0 receives every rank's records"""
    rng = np.random.default_rng(8)
    steps = [rng.integers(800000, 1000000, size=32).tolist(), rng.integers(100000, 2000000, size=13).tolist(), [7, 7, 7, 7, 7]]
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker8, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for si, sizes in enumerate(steps):
        shards = [outs[r][si][0] for r in range(world)]
        assert sorted(s for sh in shards for s in sh) == list(range(len(sizes)))
        loads = [sum(sizes[s] for s in sh) for sh in shards]
        assert max(loads) - min(loads) <= max(sizes)
        if len(sizes) >= world:
            assert all(len(sh) > 0 for sh in shards)
        gathered = outs[0][si][1]
        assert len(gathered) == world
        for r in range(world):   # rank order is kept: part r is rank r's records
            assert np.array(gathered[r], np.int64).reshape(-1, 3).tolist() == _records_for(shards[r]).reshape(-1, 3).tolist()


def test_tcp_rendezvous_world8():
    """the C ABI's TCP rendezvous with eight ranks (what `sdgpu clustersearch` uses under an 8-process launcher)"""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_tcp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] == bytes(range(128)) for o in outs)
    want = np.concatenate([np.arange(5 + 11 * r, dtype=np.int64) * (r + 3) for r in range(world)])
    assert outs[0][2] == want.tolist()
    assert outs[0][3] == [(5 + 11 * r) * 8 for r in range(world)]


def _gather_stream_worker(rank, world, port, own, root_cap, q):
    import ctypes as C
    from spacedust_amd import _lib
    L = _lib.load()
    t = C.c_void_p()
    assert L.sd_tcp_connect(b'127.0.0.1', port, world, rank, C.byref(t)) == 0
    rng = np.random.default_rng(100 + rank)
    n_rounds = 5
    rounds = [[0, 0, 2, 4], [1, 2, 2], [], [4], [0, 1, 2, 3, 4], [3], [], [2, 2]][rank]   # rounds of this rank's ranges (rank 2: none)
    pay = [rng.integers(0, 256, int(rng.integers(0, 200000)), dtype=np.uint8) for _ in rounds]
    rr = np.array(rounds, np.uint32)
    out = np.zeros(root_cap if (rank == 0 and not own and root_cap > 0) else 0, np.uint8)
    g = C.c_void_p()
    assert L.sd_gather_stream_begin_tcp(t, world, rank, len(rr), rr.ctypes.data_as(C.c_void_p) if len(rr) else None, n_rounds,
                                        out.ctypes.data_as(C.c_void_p) if out.size else None, out.nbytes, 1 if own else 0, C.byref(g)) == 0
    sink = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64)(C.cast(L.sd_gather_stream_sink, C.c_void_p).value)
    for i, p in enumerate(pay):
        if root_cap == -1 and rank == 1 and i == len(pay) - 1:
            break   # (this rank's search "failed": its last range never arrives)
        sink(g, i, p.ctypes.data if p.size else None, p.size)
    offs = np.zeros(n_rounds + 1, np.uint64)
    sizes = np.zeros((n_rounds, world), np.uint64)
    total = C.c_uint64()
    data = C.c_void_p()
    rc = L.sd_gather_stream_wait(g, offs.ctypes.data_as(C.c_void_p), sizes.ctypes.data_as(C.c_void_p), C.byref(total), C.byref(data))
    raw = None
    if rank == 0 and rc == 0:
        raw = C.string_at(data.value, int(total.value)) if total.value else b''
        if not own:
            assert data.value == out.ctypes.data
    L.sd_gather_stream_destroy(g)
    L.sd_tcp_close(t)
    q.put((rank, rc, rounds, [p.tobytes() for p in pay], raw, offs.tolist(), sizes.tolist()))


def _run_gather_stream(world, own, root_cap, port):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_stream_worker, args=(r, world, port, own, root_cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return outs


def test_gather_round_by_round_with_real_ranks_over_tcp():
    """sd_gather_stream_* (the gather round by round behind a running search) with two, four and eight processes over the C ABI's TCP
    rendezvous -- the transport `sdgpu clustersearch` uses for ranks that share a device; the RCCL form runs the same round logic --:
    ranks with different numbers of ranges per round, ranks and rounds without any, the root's buffer owned by the stream (grown round
    by round) or brought by the caller; the rounds land back to back on the root, rank by rank inside a round.  A caller's buffer
    that is too small: the root reports SD_ENOMEM, every rank comes back (nobody waits in a send).  A rank that closes its stream before
    all its ranges have arrived (a failed search): its rounds go out empty, the closing round tells the root (SD_EMISMATCH)."""
    base = 33500 + (os.getpid() % 2000)
    for k_, (world, own, cap) in enumerate(((2, False, 8 << 20), (2, True, 0), (4, True, 0), (8, False, 16 << 20), (8, True, 0))):
        outs = _run_gather_stream(world, own, cap, base + k_)
        assert all(o[1] == 0 for o in outs)
        want, per_round = b'', []
        for rd in range(5):
            at = len(want)
            for o in outs:
                for r_, p in zip(o[2], o[3]):
                    if r_ == rd:
                        want += p
            per_round.append(len(want) - at)
        assert outs[0][4] == want and len(want) > 100000
        assert outs[0][5] == np.concatenate([[0], np.cumsum(per_round)]).tolist()
        assert [sum(row) for row in outs[0][6]] == per_round
    outs = _run_gather_stream(4, False, 50000, base + 9)   # the root's buffer is too small from the first round on
    assert outs[0][1] == -4 and all(o[1] == 0 for o in outs[1:])   # SD_ENOMEM on the root
    outs = _run_gather_stream(4, True, -1, base + 10)      # rank 1 closes its stream one range short: nobody waits, the root is told
    assert outs[0][1] == -6 and all(o[1] == 0 for o in outs[1:])   # SD_EMISMATCH on the root
