"""The N>1 path on real hardware as far as one GPU allows: two ranks (gloo rendezvous on 127.0.0.1) share cuda:0, each
runs the pipeline on its shard of whole query sets (shard_query_sets) against the full target DB, the per-entry
result records are gathered (gather_results) -- and must equal the records of one unsharded run: E-values and
combinehits' set-count factor depend only on the global target DB."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _search(sets, tsv=None):
    """cluster records (sd_search_result_records) of the given query sets, in the given order"""
    from spacedust_amd.api import Host, Context
    from spacedust_amd.pipeline import SetDB, ClusterSearch
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(6, genes_per_proteome=150, n_families=220, seed=23)
    db = SetDB.from_proteomes(ps)
    host, gpu = Host(4), Context(0)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
    ranges = [(int(ps.set_start[s]), int(ps.set_start[s + 1])) for s in sets]
    outs = cs.search_stream(db, ranges, same_db=True, chunk_queries=64, want_records=True) if ranges else []
    recs = np.concatenate([o['records'] for o in outs]) if outs else np.zeros(0, np.uint8)
    sizes = [int(ps.offsets[ps.set_start[s + 1]] - ps.offsets[ps.set_start[s]]) for s in range(6)]
    return recs, sizes, db


def _tsv_lines(records, db, path):
    from spacedust_amd.pipeline import write_records_tsv
    n_clu, n_hit = write_records_tsv(records, path, db, db)
    lines = open(path).readlines()
    assert sum(1 for l in lines if l.startswith('#')) == n_clu and len(lines) == n_clu + n_hit
    return lines


def _worker(rank, world, port, q, tmp):
    import torch.distributed as dist
    from spacedust_amd.pipeline import shard_query_sets, gather_results
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(6, genes_per_proteome=150, n_families=220, seed=23)
    sizes = [int(ps.offsets[ps.set_start[s + 1]] - ps.offsets[ps.set_start[s]]) for s in range(6)]
    mine = shard_query_sets(sizes, world, rank)
    recs, _, db = _search(mine)
    # the gather of the cluster records (RCCL refuses two ranks on one GPU: the same bytes travel over gloo here), then rank 0
    # writes the TSV from the gathered buffer
    pad = np.concatenate([recs, np.zeros((-len(recs)) % 8, np.uint8)])
    parts = gather_results(np.frombuffer(pad.tobytes(), np.int64), dist)
    lens = gather_results(np.array([len(recs)], np.int64), dist)
    if rank == 0:
        allrec = np.concatenate([np.asarray(p, np.int64).view(np.uint8)[:int(n[0])] for p, n in zip(parts, lens)])
        lines = _tsv_lines(allrec, db, os.path.join(tmp, 'two.tsv'))
        q.put((mine, lines))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank(gpu, tmp_path):
    import torch.multiprocessing as mp
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from dbutil import sorted_md5
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    mine0, two = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert 0 < len(mine0) < 6
    recs, _, db = _search(range(6))
    one = _tsv_lines(recs, db, str(tmp_path / 'one.tsv'))
    assert len(one) > 100 and sum(1 for l in one if l.startswith('#')) > 10
    # the TSV written from the gathered records is the one-rank TSV up to the order of the query sets (cluster keys count from 0 in
    # both, and are consecutive)
    assert sorted_md5(one, drop_first_column=True) == sorted_md5(two, drop_first_column=True)
    keys = [int(l.split('\t')[0][1:]) for l in two if l.startswith('#')]
    assert keys == list(range(len(keys)))


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
    raw, sizes = g.gather_bytes(np.arange(1001, dtype=np.uint8))
    assert raw.tolist() == (np.arange(1001) % 256).tolist() and sizes.tolist() == [1001]
    # pinned, pre-sized staging (sd_comm_host_buffer): the records are built in the communicator's send buffer and gathered into its
    # receive buffer -- one exchange, no size probe; a receive buffer that turns out too small falls back to the probe + fresh array
    send = g.host_buffer(0, 1 << 20)
    recv = g.host_buffer(1, 1 << 20)
    assert send.nbytes == 1 << 20 and recv.nbytes == 1 << 20
    rng = np.random.default_rng(5)
    payload = rng.integers(0, 256, 700001, dtype=np.uint8)
    send[:len(payload)] = payload
    raw, sizes = g.gather_bytes(send[:len(payload)], out=recv)
    assert sizes.tolist() == [len(payload)] and (raw == payload).all() and raw.ctypes.data == recv.ctypes.data   # landed in the pinned buffer itself
    small = g.host_buffer(1, 1000)   # (no larger request: the same buffer, the view is 1 000 bytes long)
    raw, sizes = g.gather_bytes(send[:len(payload)], out=small)
    assert sizes.tolist() == [len(payload)] and (raw == payload).all()
    big = g.host_buffer(0, 3 << 20)   # a larger request replaces the buffer
    big[:5] = [1, 2, 3, 4, 5]
    raw, _ = g.gather_bytes(big[:5], out=g.host_buffer(1, 1 << 20))
    assert raw.tolist() == [1, 2, 3, 4, 5]


def test_sdgpu_clustersearch_two_ranks_equal_one(gpu, tmp_path):
    """the binary's own multi-rank mode (RANK / WORLD_SIZE, whole query sets per rank, every rank's cluster records gathered on
    rank 0, which writes the TSV from the gathered buffer): two ranks sharing cuda:0 must write the same clusters as one rank"""
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
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(30500 + os.getpid() % 2000))
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


def test_sdgpu_search_two_ranks_leave_one_alignment_db(gpu, tmp_path):
    """plain `search` under two ranks: the per-rank parts become ONE alignment DB (split data files NAME.0 / NAME.1 under one
    NAME.index, the layout of the reference's multi-threaded DBWriter) with the entries of the one-rank DB, key for key"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from dbutil import SDGPU, sdgpu, example_fasta
    fa = example_fasta(tmp_path)
    q, t = tmp_path / 'q', tmp_path / 't'
    sdgpu('createsetdb', fa[0], fa[1], t, tmp_path / 'tmp', '-v', '0')
    sdgpu('createsetdb', fa[1], fa[0], q, tmp_path / 'tmp', '-v', '0')
    sdgpu('search', q, t, tmp_path / 'aln1', tmp_path / 'tmp1', '-v', '0')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(32500 + os.getpid() % 2000))
        procs.append(subprocess.Popen([SDGPU, 'search', str(q), str(t), str(tmp_path / 'aln2'), str(tmp_path / 'tmp2'), '-v', '0'], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    assert os.path.exists(tmp_path / 'aln2.0') and os.path.exists(tmp_path / 'aln2.1') and not os.path.exists(tmp_path / 'aln2.0.index')
    # both DBs flattened by the binary's own reader
    sdgpu('prefixid', tmp_path / 'aln1', tmp_path / 'aln1.flat', '--tsv', '--threads', '1')
    sdgpu('prefixid', tmp_path / 'aln2', tmp_path / 'aln2.flat', '--tsv', '--threads', '1')
    one, two = open(tmp_path / 'aln1.flat').readlines(), open(tmp_path / 'aln2.flat').readlines()
    assert len(one) > 10000 and sorted(one) == sorted(two)
    n1 = sum(1 for _ in open(str(tmp_path / 'aln1') + '.index'))
    n2 = sum(1 for _ in open(str(tmp_path / 'aln2') + '.index'))
    assert n1 == n2


@pytest.mark.gpu
def test_sdgpu_clustersearch_eight_ranks_equal_one(gpu, tmp_path):
    """the layout of the driver's 8-GPU run rehearsed on the one GPU a box has: eight `sdgpu clustersearch` ranks share cuda:0, each
    takes its share of 16 synthetic query proteomes (sd_shard_query_sets), searches them against all 16, and rank 0 writes the TSV
    from the eight gathered record buffers (ranks that share a device exchange over the TCP rendezvous: RCCL refuses two ranks
    per device) -- the clusters must be those of the one-rank run, the cluster keys consecutive across the eight parts"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), 'tools'))
    from dbutil import SDGPU, sdgpu, sorted_md5
    from iter3_scale import _write_fasta
    from spacedust_amd.synth import make_proteomes, ALPHABET
    ps = make_proteomes(16, genes_per_proteome=200, n_families=300, seed=0x5ED0 + 8)
    lut = np.frombuffer(ALPHABET.encode(), np.uint8)
    fa_dir = tmp_path / 'fa'
    os.makedirs(fa_dir)
    files = [_write_fasta(ps, s, str(fa_dir), lut) for s in range(16)]
    t = tmp_path / 't'
    sdgpu('createsetdb', *files, t, tmp_path / 'tmp', '-v', '0')
    sdgpu('clustersearch', t, t, tmp_path / 'one.tsv', tmp_path / 'tmp1', '-v', '0', '--filter-self-match', '1')
    procs = []
    port = str(34500 + os.getpid() % 2000)
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='8', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=port, SD_CPUS='2')
        procs.append(subprocess.Popen([SDGPU, 'clustersearch', str(t), str(t), str(tmp_path / 'eight.tsv'), str(tmp_path / 'tmp8'), '-v', '0',
                                       '--filter-self-match', '1'], env=env))
    assert [p.wait(timeout=900) for p in procs] == [0] * 8
    one = open(tmp_path / 'one.tsv').readlines()
    eight = open(tmp_path / 'eight.tsv').readlines()
    assert sum(1 for l in one if l.startswith('#')) > 10 and len(one) > 100
    assert sorted_md5(one, drop_first_column=True) == sorted_md5(eight, drop_first_column=True)
    keys = [int(l.split('\t')[0][1:]) for l in eight if l.startswith('#')]
    assert keys == list(range(len(keys)))


def test_records_built_inside_the_stream_equal_records_on_demand(gpu):
    """a rank's hand-over to the final gather: the cluster records built when a range is finalised (sd_search_set_want_records, one buffer for
    all ranges) are the bytes sd_search_result_records builds on demand from the finished results, and a stream that copies out the last
    range's arrays only (`arrays='last'`, what bench.py times) reports the same counters and the same records for every range"""
    import ctypes as C
    from spacedust_amd import _lib
    from spacedust_amd.api import Host, Context
    from spacedust_amd.pipeline import SetDB, ClusterSearch
    from spacedust_amd.synth import make_proteomes
    L = _lib.load()
    ps = make_proteomes(6, genes_per_proteome=150, n_families=220, seed=23)
    db = SetDB.from_proteomes(ps)
    host = Host(4)
    cs = ClusterSearch(gpu, host, db, max_seqs=300, bin_size=2, filter_self_match=True)
    ranges = [(int(ps.set_start[s]), int(ps.set_start[s + 1])) for s in range(6)]
    full = cs.search_stream(db, ranges, same_db=True, chunk_queries=64, want_records=True)
    lazy = cs.search_stream(db, ranges, same_db=True, chunk_queries=64, want_records=True, arrays='last')
    assert sum(o['records'].size for o in full) > 10000 and full[-1]['records_all'].size == sum(o['records'].size for o in full)
    for a, b in zip(full, lazy):
        for key in ('entries', 'matched_hits', 'clusters', 'cluster_hits', 'aligned', 'accepted', 'prefilter_hits'):
            assert a[key] == b[key], key
        assert np.array_equal(a['records'], b['records'])
    assert 'hit_q' not in lazy[0] and np.array_equal(lazy[-1]['hit_q'], full[-1]['hit_q'])
    assert np.array_equal(full[-1]['records_all'], np.concatenate([o['records'] for o in full]))
    # on demand: the flag off, the records asked of the finished result handles
    L.sd_search_set_want_records(cs.h, 0)
    keep = []
    from spacedust_amd.pipeline import _setdb_struct
    qv = _setdb_struct(db, keep)
    rb = np.array([r[0] for r in ranges], np.uint32)
    re_ = np.array([r[1] for r in ranges], np.uint32)
    handles = (C.c_void_p * len(ranges))()
    assert L.sd_search_stream(cs.h, C.byref(qv), 1, len(ranges), _lib.ptr(rb), _lib.ptr(re_), handles) == 0
    for ri in range(len(ranges)):
        h = C.c_void_p(handles[ri])
        need = C.c_uint64()
        assert L.sd_search_result_records(h, None, 0, C.byref(need)) == 0
        buf = np.zeros(int(need.value), np.uint8)
        if need.value:
            assert L.sd_search_result_records(h, _lib.ptr(buf), buf.nbytes, C.byref(need)) == 0
        L.sd_search_result_destroy(h)
        assert np.array_equal(buf, full[ri]['records']), ri


def test_gather_round_by_round_behind_the_stream_equals_the_final_gather(gpu):
    """sd_gather_stream_*: the ranges' records leave through the stream's records sink as the ranges are finalised and are gathered
    round by round on the communicator's stream while the search goes on; the rounds land back to back on the root and are, for the
    one rank a one-GPU box allows, the bytes of the records built inside the stream (= the final one-blob gather).  Rounds of one,
    several and no ranges; a send buffer smaller than a round (pageable staging of that round); a root buffer that is too small
    (SD_ENOMEM from the round that does not fit, no hang); the sink alone (plain C calls) with payloads of known bytes."""
    import ctypes as C
    from spacedust_amd import _lib
    from spacedust_amd.api import Host
    from spacedust_amd.pipeline import SetDB, ClusterSearch, RcclGather
    from spacedust_amd.synth import make_proteomes
    L = _lib.load()
    g = RcclGather(0, 1, 0, RcclGather.unique_id())
    # the sink alone
    rng = np.random.default_rng(11)
    pay = [rng.integers(0, 256, n, dtype=np.uint8) for n in (5000, 0, 70001, 123, 9, 0, 40000)]
    rounds = [0, 0, 2, 2, 2, 3, 5]   # rounds 1 and 4 hold no range
    for send_cap, recv_cap, fails in ((60000, 1 << 20, False), (1 << 20, 1 << 20, False), (1 << 20, 80000, True)):   # (the buffers only grow)
        g.host_buffer(0, send_cap)
        recv = g.host_buffer(1, recv_cap)[:recv_cap]
        gs = g.stream_begin(rounds, 6, out=recv)
        fn, user = g.stream_sink()
        sink = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64)(fn.value)
        for i, p in enumerate(pay):
            sink(user, i, p.ctypes.data if p.size else None, p.size)
        if fails:
            with pytest.raises(_lib.SdError):
                g.stream_end()
            continue
        raw, offs, sizes = g.stream_end()
        assert np.array_equal(raw, np.concatenate(pay))
        per_round = [sum(p.size for p, r in zip(pay, rounds) if r == x) for x in range(6)]
        assert sizes[:, 0].tolist() == per_round and offs.tolist() == np.concatenate([[0], np.cumsum(per_round)]).tolist()
    # the root's buffer owned by the stream and grown round by round (what `sdgpu clustersearch` asks for): over RCCL a round that does
    # not fit is agreed on by all ranks (SD_ENOMEM before any payload moves) and repeated after the root has made room
    rr = np.array(rounds, np.uint32)
    gs = C.c_void_p()
    assert L.sd_gather_stream_begin(g.h, 0, len(rr), _lib.ptr(rr), 6, None, 0, 1, C.byref(gs)) == 0
    sink = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64)(C.cast(L.sd_gather_stream_sink, C.c_void_p).value)
    for i, p in enumerate(pay):
        sink(gs, i, p.ctypes.data if p.size else None, p.size)
    total, data = C.c_uint64(), C.c_void_p()
    offs = np.zeros(7, np.uint64)
    assert L.sd_gather_stream_wait(gs, _lib.ptr(offs), None, C.byref(total), C.byref(data)) == 0
    assert C.string_at(data.value, int(total.value)) == np.concatenate(pay).tobytes() and int(offs[-1]) == int(total.value)
    L.sd_gather_stream_destroy(gs)
    # behind a running search
    ps = make_proteomes(6, genes_per_proteome=150, n_families=220, seed=23)
    db = SetDB.from_proteomes(ps)
    cs = ClusterSearch(gpu, Host(4), db, max_seqs=300, bin_size=2, filter_self_match=True)
    ranges = [(int(ps.set_start[s]), int(ps.set_start[s + 1])) for s in range(6)]
    ref = cs.search_stream(db, ranges, same_db=True, chunk_queries=64, want_records=True, arrays='last')
    want = ref[-1]['records_all']
    assert want.size > 10000
    g.host_buffer(0, 4 << 20)
    recv = g.host_buffer(1, 4 << 20)
    g.stream_begin([0, 0, 1, 2, 2, 2], 3, out=recv)
    outs = cs.search_stream(db, ranges, same_db=True, chunk_queries=64, want_records=True, arrays='last', records_sink=g.stream_sink())
    raw, offs, sizes = g.stream_end()
    assert np.array_equal(raw, want) and raw.ctypes.data == recv.ctypes.data
    assert [o['entries'] for o in outs] == [o['entries'] for o in ref] and outs[-1]['records_all'] is None
    per_range = [o['records'].size for o in ref]
    assert sizes[:, 0].tolist() == [sum(per_range[:2]), per_range[2], sum(per_range[3:])]


def test_bench_line_of_the_multi_gpu_path_is_one_json_line(gpu, tmp_path):
    """bench.py through its N > 1 code path with the one rank a one-GPU box allows (SD_BENCH_FORCE_DIST=1: RCCL communicator, pinned
    staging, the gather round by round behind the stream, the TSV from the gathered buffer) on a small workload: stdout is exactly one
    JSON line (RCCL prints its banner to the process's stdout), it carries the driver's keys, and the gathered records reproduce the
    results' cluster counters; the same with the one-blob gather at the end (SD_BENCH_GATHER_STREAM=0)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ('1', '0'):
        env = dict(os.environ, SD_BENCH_FORCE_DIST='1', SD_BENCH_GATHER_STREAM=mode, MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(29650 + int(mode)))
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--proteomes', '24',
                            '--genes', '300', '--batch', '3', '--no-cpu', '--no-children', '--no-index-check',
                            '--detail-out', str(tmp_path / ('detail%s.json' % mode))],
                           cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert r.returncode == 0, r.stderr.decode(errors='replace')[-3000:]
        out = r.stdout.decode()
        assert out.endswith('\n') and out.count('\n') == 1, out[:400]
        d = json.loads(out)
        for k_ in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                   'data', 'config', 'roofline'):
            assert k_ in d, k_
        assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0 and d['unit'] == 'genome-pairs/s'
        g = d['gather']
        assert g['bytes'] > 0 and g['tsv']['clusters_match_results'] and g['tail_frac_of_timed_region'] is not None
        assert ('sd_gather_stream' in g['how']) == (mode == '1')
        got[mode] = (g['bytes'], g['tsv']['clusters'], g['tsv']['hits'], d['results']['clusters'])
    assert got['1'] == got['0']
