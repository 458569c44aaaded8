"""GPU parity: Smith-Waterman HIP kernels (through the C ABI) vs the oracle, bit exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs(ps, rng, n_random=150):
    fam = ps.family
    pq, pt = [], []
    # homolog pairs (same family, different proteome), identity pairs, random pairs
    by_fam = {}
    for i, f in enumerate(fam):
        if f >= 0:
            by_fam.setdefault(int(f), []).append(i)
    for f, members in by_fam.items():
        for a in members:
            for b in members:
                if a != b:
                    pq.append(a)
                    pt.append(b)
    for i in range(0, ps.n, 7):
        pq.append(i)
        pt.append(i)
    for _ in range(n_random):
        pq.append(int(rng.integers(ps.n)))
        pt.append(int(rng.integers(ps.n)))
    return np.array(pq, np.uint32), np.array(pt, np.uint32)


def _seq(ps, i):
    return ps.residues[int(ps.offsets[i]):int(ps.offsets[i + 1])]


def test_sw_score_pass_lanes(gpu, host, oracle, small_proteomes):
    ps = small_proteomes
    rng = np.random.default_rng(5)
    pq, pt = _pairs(ps, rng, 60)
    pq, pt = pq[:400], pt[:400]
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    par = gpu.sw_params(mat, int(ps.offsets[-1]))
    m = mat.reshape(21, 21).astype(np.int16)
    for lanes in (32, 16):
        out = gpu.sw_score(par, ss, ss, pq, pt, lanes=lanes)
        for x in range(len(pq)):
            q, t = _seq(ps, pq[x]), _seq(ps, pt[x])
            cb = sw_bias[int(ps.offsets[pq[x]]):int(ps.offsets[pq[x] + 1])].astype(np.int16)
            prof = m[:, q] + cb[None, :]
            ref = oracle.sw_pass(prof, t, lanes)
            exp = (ref[0], ref[1] if ref[0] > 0 else -1, ref[2])
            assert tuple(int(v) for v in out[x]) == exp, (lanes, x, out[x], ref)


def test_sw_align_batch_matches_oracle(gpu, host, oracle, small_proteomes):
    ps = small_proteomes
    rng = np.random.default_rng(7)
    pq, pt = _pairs(ps, rng)
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    db = int(ps.offsets[-1])
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    par = gpu.sw_params(mat, db)
    res, pool = gpu.sw_align(par, ss, ss, pq, pt, identity=(pq == pt))
    n_bt = 0
    for x in range(len(pq)):
        o = oracle.sw_align(_seq(ps, pq[x]), _seq(ps, pt[x]), db, identity=bool(pq[x] == pt[x]))
        r = res[x]
        assert int(r['score']) == o['score'], (x, r, o)
        assert (int(r['qStart']), int(r['qEnd']), int(r['tStart']), int(r['tEnd'])) == \
               (o['qStart'], o['qEnd'], o['tStart'], o['tEnd']), (x, r, o)
        if o['evalue'] <= 20.0:   # 2 x evalThr: bit exact; far above the threshold the device value stands
            assert float(r['evalue']) == o['evalue'], (x, r['evalue'], o['evalue'])
        else:
            assert abs(float(r['evalue']) - o['evalue']) <= 1e-9 * o['evalue'], (x, r['evalue'], o['evalue'])
        assert int(r['btLen']) == o['btLen'], (x, r, o)
        if o['btLen'] > 0:
            n_bt += 1
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == o['backtrace'], (x, bt, o['backtrace'])
            assert int(r['identical']) == o['identical']
    assert n_bt > 50


def test_sw_long_query_strips(gpu, host, oracle):
    """queries longer than one 1024-row strip exercise the strip boundary hand-off"""
    rng = np.random.default_rng(3)
    bg = np.full(20, 0.05)
    base = rng.choice(20, 2300, p=bg).astype(np.uint8)
    mut = base.copy()
    flip = rng.random(len(mut)) < 0.3
    mut[flip] = rng.choice(20, int(flip.sum()), p=bg)
    seqs = [base, mut[:1900], rng.choice(20, 1500, p=bg).astype(np.uint8), base[500:1700].copy()]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    resid = np.concatenate(seqs)
    sw_bias, _, _ = host.comp_bias(resid, off)
    mat, _, _ = host.matrix(0)
    db = int(off[-1])
    ss = gpu.seqset(resid, off, sw_bias)
    par = gpu.sw_params(mat, db, cov_thr=0.0)
    pq = np.array([0, 1, 0, 2, 3, 0, 1], np.uint32)
    pt = np.array([1, 0, 2, 0, 0, 3, 3], np.uint32)
    res, pool = gpu.sw_align(par, ss, ss, pq, pt)
    for x in range(len(pq)):
        o = oracle.sw_align(seqs[pq[x]], seqs[pt[x]], db, cov_thr=0.0)
        r = res[x]
        got = (int(r['score']), int(r['qStart']), int(r['qEnd']), int(r['tStart']), int(r['tEnd']), int(r['btLen']))
        exp = (o['score'], o['qStart'], o['qEnd'], o['tStart'], o['tEnd'], o['btLen'])
        assert got == exp, (x, got, exp)
        if o['btLen'] > 0:
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == o['backtrace']


def test_sw_bands_beyond_the_lds_classes(gpu, host, oracle):
    """alignments that span one gap of 1 100 .. 2 600 residues: the banded traceback starts at |tLen - qLen| + 1 columns,
    beyond the widest LDS band class, and runs in the global-band class (device-resident and host-orchestrated task
    building); records and backtraces against rows from the real reference (tests/golden/long_vectors.npz, made by
    tools/make_golden_long.py) and against the oracle"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'long_vectors.npz'))
    seqs = []
    for q, t in zip(g['gap_q'], g['gap_t']):
        seqs += [oracle.map_sequence(str(q)), oracle.map_sequence(str(t))]
    rng = np.random.default_rng(9)
    seqs.append(rng.integers(0, 20, 700).astype(np.uint8))   # an ordinary pair next to them
    seqs.append(seqs[-1][40:640].copy())
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    resid = np.concatenate(seqs)
    sw_bias, _, _ = host.comp_bias(resid, off)
    mat, _, _ = host.matrix(0)
    db = 10 ** 7
    ss = gpu.seqset(resid, off, sw_bias)
    par = gpu.sw_params(mat, db, cov_thr=0.0)
    pq = np.array([0, 2, 4, 6, 1, 3, 5], np.uint32)
    pt = np.array([1, 3, 5, 7, 0, 2, 4], np.uint32)
    for hostpath in (False, True):
        res, pool = gpu.sw_align(par, ss, ss, pq, pt, hostpath=hostpath)
        for x in range(len(pq)):
            r = res[x]
            got = (int(r['score']), int(r['qStart']), int(r['qEnd']), int(r['tStart']), int(r['tEnd']), int(r['identical']), int(r['btLen']))
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            if x < 3:
                assert got == tuple(int(v) for v in g['gap_res'][x]), (hostpath, x, got, g['gap_res'][x])
                assert bt == str(g['gap_bt'][x]), (hostpath, x)
                assert float(r['evalue']) == float(g['gap_eval'][x])
                assert abs((got[2] - got[1]) - (got[4] - got[3])) + 1 > 1022
            o = oracle.sw_align(seqs[pq[x]], seqs[pt[x]], db, cov_thr=0.0)
            assert got == (o['score'], o['qStart'], o['qEnd'], o['tStart'], o['tEnd'], o['identical'], o['btLen']), (hostpath, x, got, o)
            assert bt == o['backtrace'], (hostpath, x)


def test_sw_scores_beyond_int16_saturate_like_the_word_kernel(gpu, host, oracle):
    """near-identical sequences of 7 000 - 9 000 residues: the exact score is ~45 000, the reference's word kernel saturates
    at 32 767 (simdi16_adds, StripedSmithWaterman.cpp:1069) and reports the first cell that reaches it -- score,
    coordinates, backtrace against the real reference (tests/golden/long_vectors.npz)"""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'long_vectors.npz'))
    seqs = []
    for q, t in zip(g['sat_q'], g['sat_t']):
        seqs += [oracle.map_sequence(str(q)), oracle.map_sequence(str(t))]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    resid = np.concatenate(seqs)
    sw_bias, _, _ = host.comp_bias(resid, off)
    mat, _, _ = host.matrix(0)
    ss = gpu.seqset(resid, off, sw_bias)
    par = gpu.sw_params(mat, 10 ** 7, cov_thr=0.0)
    pq = np.array([0, 2, 4], np.uint32)
    pt = np.array([1, 3, 5], np.uint32)
    for hostpath in (False, True):
        res, pool = gpu.sw_align(par, ss, ss, pq, pt, hostpath=hostpath)
        for x in range(3):
            r = res[x]
            got = (int(r['score']), int(r['qStart']), int(r['qEnd']), int(r['tStart']), int(r['tEnd']), int(r['identical']), int(r['btLen']))
            assert got == tuple(int(v) for v in g['sat_res'][x]), (hostpath, x, got, g['sat_res'][x])
            assert got[0] == 32767
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == str(g['sat_bt'][x]), (hostpath, x)


def test_sw_traceback_scratch_in_slices(gpu, host, monkeypatch):
    """more traceback direction bytes than the scratch budget (long result lists of homologs on very large targets): the
    rounds run the prefix of the pending tasks that fits and come back for the rest -- same records, same backtraces as
    with everything in one round (SD_TB_BUDGET forces a 1-MB budget)"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=6, genes_per_proteome=260, n_families=400, seed=23)
    rng = np.random.default_rng(11)
    pq, pt = _pairs(ps, rng, 500)
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    par = gpu.sw_params(mat, int(ps.offsets[-1]), cov_thr=0.0)
    a, pa = gpu.sw_align(par, ss, ss, pq, pt, identity=(pq == pt))
    monkeypatch.setenv('SD_TB_BUDGET', '1048576')
    b, pb = gpu.sw_align(par, ss, ss, pq, pt, identity=(pq == pt))
    monkeypatch.delenv('SD_TB_BUDGET')
    assert int((a['btLen'] > 0).sum()) > 1000
    for f in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'flags'):
        assert np.array_equal(a[f], b[f]), f
    for x in np.flatnonzero(a['btLen'] > 0):
        sa = pa[int(a['btOffset'][x]):int(a['btOffset'][x]) + int(a['btLen'][x])]
        sb = pb[int(b['btOffset'][x]):int(b['btOffset'][x]) + int(b['btLen'][x])]
        assert np.array_equal(sa, sb), x


def test_sw_device_vs_host_orchestration(gpu, host):
    """the device-resident gating / task building gives the same records as the host-side one, at a size the
    oracle would not finish in seconds (all sw modes, all coverage modes)"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=6, genes_per_proteome=260, n_families=400, seed=23)
    rng = np.random.default_rng(11)
    pq, pt = _pairs(ps, rng, 4000)
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    db = int(ps.offsets[-1])
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    for sw_mode, cov_mode, ev in ((2, 2, 10.0), (1, 0, 1e-3), (0, 1, 10.0), (2, 3, 1e-5)):
        par = gpu.sw_params(mat, db, cov_mode=cov_mode, eval_thr=ev, sw_mode=sw_mode)
        ident = (pq == pt)
        a, pa = gpu.sw_align(par, ss, ss, pq, pt, identity=ident)
        b, pb = gpu.sw_align(par, ss, ss, pq, pt, identity=ident, hostpath=True)
        for f in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'flags'):
            assert np.array_equal(a[f], b[f]), (sw_mode, cov_mode, f, np.flatnonzero(a[f] != b[f])[:5])
        near = b['evalue'] <= 2 * ev
        assert np.array_equal(a['evalue'][near], b['evalue'][near])
        assert np.all(np.abs(a['evalue'][~near] - b['evalue'][~near]) <= 1e-9 * b['evalue'][~near])
        for x in np.flatnonzero(a['btLen'] > 0)[::7]:
            sa = pa[int(a['btOffset'][x]):int(a['btOffset'][x]) + int(a['btLen'][x])]
            sb = pb[int(b['btOffset'][x]):int(b['btOffset'][x]) + int(b['btLen'][x])]
            assert np.array_equal(sa, sb), x


def test_sw_packed_kernel_equals_int32_kernel(gpu, host, monkeypatch):
    """the packed-int16 score kernels (all row classes, 32- and 64-lane variants, wide row code) against the int32
    kernel on the same pairs, through the full align call"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=5, genes_per_proteome=400, n_families=600, seed=5, mean_len=330)
    rng = np.random.default_rng(17)
    pq, pt = _pairs(ps, rng, 6000)
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    db = int(ps.offsets[-1])
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    par = gpu.sw_params(mat, db)
    ident = (pq == pt)
    a, pa = gpu.sw_align(par, ss, ss, pq, pt, identity=ident)
    monkeypatch.setenv('SD_SW_INT32', '1')
    b, pb = gpu.sw_align(par, ss, ss, pq, pt, identity=ident)
    for f in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'flags', 'evalue'):
        assert np.array_equal(a[f], b[f]), (f, np.flatnonzero(a[f] != b[f])[:5])
    lens = (ps.offsets[1:] - ps.offsets[:-1])[pq]
    assert lens.min() <= 128 and lens.max() > 768   # every row class is exercised


@pytest.mark.parametrize('chain', [2, 5, 64])
def test_sw_chained_quads_equal_one_quad_per_wavefront(gpu, host, monkeypatch, chain):
    """sw_score_pk_chain_kernel (SD_SW_CHAIN=M: a wavefront runs up to M consecutive quads of the pair list, those of one query back
    to back through one systolic pipeline) against the one-quad-per-wavefront kernel: queries with long runs of targets (chains that
    fill their M quads), runs of one to three tasks (a chain that ends after every quad, empty pairs), targets shorter than the 32
    lanes, every aligned row class"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=5, genes_per_proteome=400, n_families=600, seed=5, mean_len=330)
    rng = np.random.default_rng(23)
    # the proteomes plus 40 fragments of 3 .. 40 residues (targets shorter than the 32 lanes: a segment is padded to 32 steps)
    frag_len = rng.integers(3, 41, size=40)
    frag = [ps.residues[int(o):int(o) + int(n)] for o, n in zip(ps.offsets[rng.integers(ps.n, size=40)], frag_len)]
    residues = np.concatenate([ps.residues] + frag)
    offsets = np.concatenate([ps.offsets, ps.offsets[-1] + np.cumsum(frag_len)]).astype(ps.offsets.dtype)
    n_seq = len(offsets) - 1
    short = np.arange(ps.n, n_seq)
    lens = offsets[1:] - offsets[:-1]
    pq, pt = [], []
    for q in rng.choice(ps.n, 120, replace=False):   # long runs: 5 .. 90 targets of one query, homologs and fragments among them
        n = int(rng.integers(5, 90))
        t = rng.integers(ps.n, size=n)
        t[:4] = rng.choice(short, 4)
        same = np.flatnonzero(ps.family == ps.family[q]) if ps.family[q] >= 0 else np.zeros(0, np.int64)
        k = min(len(same), n - 4)
        t[4:4 + k] = same[:k]
        pq += [int(q)] * n
        pt += [int(x) for x in t]
    q2, t2 = _pairs(ps, rng, 3000)   # short runs
    pq = np.concatenate([np.array(pq, np.uint32), q2])
    pt = np.concatenate([np.array(pt, np.uint32), t2])
    sw_bias, _, _ = host.comp_bias(residues, offsets)
    mat, _, _ = host.matrix(0)
    ss = gpu.seqset(residues, offsets, sw_bias)
    par = gpu.sw_params(mat, int(offsets[-1]))
    ident = (pq == pt)
    monkeypatch.delenv('SD_SW_CHAIN', raising=False)
    a, pa = gpu.sw_align(par, ss, ss, pq, pt, identity=ident)
    monkeypatch.setenv('SD_SW_CHAIN', str(chain))
    b, pb = gpu.sw_align(par, ss, ss, pq, pt, identity=ident)
    for f in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'flags', 'evalue'):
        assert np.array_equal(a[f], b[f]), (f, np.flatnonzero(a[f] != b[f])[:5])
    assert np.array_equal(pa, pb)
    assert lens[pq].min() <= 160 and lens[pq].max() > 768 and lens[pt].min() < 32   # the aligned row classes, targets below 32 residues
    assert int((a['score'] > 0).sum()) > len(pq) // 2


def test_sw_align_compact_equals_full(gpu, host):
    """sd_sw_align_batch_compact returns exactly the reportable records of sd_sw_align_batch, in pair order"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=4, genes_per_proteome=300, n_families=500, seed=8)
    rng = np.random.default_rng(4)
    pq, pt = _pairs(ps, rng, 3000)
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    par = gpu.sw_params(mat, int(ps.offsets[-1]))
    ident = (pq == pt)
    full, pool_f = gpu.sw_align(par, ss, ss, pq, pt, identity=ident)
    idx, comp, pool_c = gpu.sw_align(par, ss, ss, pq, pt, identity=ident, compact=True)
    keep = np.flatnonzero(ident | ((full['btLen'] > 0) & (full['qStart'] >= 0)))
    assert np.array_equal(idx, keep.astype(np.uint32))
    for f in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'flags', 'evalue'):
        assert np.array_equal(comp[f], full[f][keep]), f
    for x in range(0, len(keep), 11):
        a = pool_c[int(comp['btOffset'][x]):int(comp['btOffset'][x]) + int(comp['btLen'][x])]
        b = pool_f[int(full['btOffset'][keep[x]]):int(full['btOffset'][keep[x]]) + int(full['btLen'][keep[x]])]
        assert np.array_equal(a, b), x
    assert 100 < len(keep) < len(pq)


def _compress_alignment(bt):
    """Matcher::compressAlignment (M/src/alignment/Matcher.cpp:166-185), letter by letter"""
    out, state, count = [], 'M', 0
    for ch in bt:
        if ch != state:
            out.append('%d%s' % (count, state))
            state, count = ch, 1
        else:
            count += 1
    out.append('%d%s' % (count, state))
    return ''.join(out)


def test_sw_run_length_text_from_the_device_equals_compress_alignment(gpu, host):
    """sd_sw_set_cigar_pool: the pool of every alignment call holds Matcher::compressAlignment of the backtrace letters the same
    call returns without it -- full, compact and with diagonals; identity pairs; alignments of under 64 to over a thousand letters,
    runs that cross the 64-letter steps of the kernel, gap-rich pairs; every other field unchanged"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=5, genes_per_proteome=320, n_families=300, seed=41, mean_len=420)
    rng = np.random.default_rng(9)
    pq, pt = _pairs(ps, rng, 5000)
    pq[:300] = pt[:300]
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    ident = (pq == pt)
    ident[:150] = False            # self pairs through the kernels: one run of several hundred letters
    try:
        for go, ge in ((11, 1), (3, 1)):   # cheap gaps: many short runs
            par = gpu.sw_params(mat, int(ps.offsets[-1]), gap_open=go, gap_extend=ge)
            for kw in (dict(), dict(compact=True), dict(compact=True, diag=np.zeros(len(pq), np.uint16))):
                gpu.set_cigar_pool(False)
                a = gpu.sw_align(par, ss, ss, pq, pt, identity=ident, **kw)
                gpu.set_cigar_pool(True)
                b = gpu.sw_align(par, ss, ss, pq, pt, identity=ident, **kw)
                if kw:
                    assert np.array_equal(a[0], b[0])
                    a, b = a[1:], b[1:]
                (ra, pa), (rb, pb) = a, b
                for f in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'evalue'):
                    assert np.array_equal(ra[f], rb[f]), f
                assert np.array_equal(ra['flags'], rb['flags'] & 0xFF)
                assert int((ra['flags'] >> 8).max()) == 0
                n_text = (rb['flags'].astype(np.int64) >> 8)
                assert np.array_equal(n_text > 0, ra['btLen'] > 0)
                lens, runs = [], 0
                for x in np.flatnonzero(ra['btLen'] > 0):
                    bt = pa[int(ra['btOffset'][x]):int(ra['btOffset'][x]) + int(ra['btLen'][x])].tobytes().decode()
                    txt = pb[int(rb['btOffset'][x]):int(rb['btOffset'][x]) + int(n_text[x])].tobytes().decode()
                    assert txt == _compress_alignment(bt), (x, bt, txt)
                    lens.append(len(bt))
                    runs += sum(c in 'MID' for c in txt)
                assert len(pb) < (len(pa) // 3 if go == 11 else len(pa))
                assert min(lens) < 64 and max(lens) > 700 and runs > 3 * len(lens), (min(lens), max(lens), runs, len(lens))
    finally:
        gpu.set_cigar_pool(False)


def test_sw_align_with_prefilter_diagonals_equals_without(gpu, host):
    """sd_sw_align_batch_compact_diag: the diagonal only lets pairs whose byte-range score saturates for certain go
    straight to the 16-bit pass; true, random and absurd diagonals all give the records of the call without them"""
    from spacedust_amd.synth import make_proteomes
    ps = make_proteomes(n_proteomes=5, genes_per_proteome=320, n_families=300, seed=41, mean_len=420)
    rng = np.random.default_rng(9)
    pq, pt = _pairs(ps, rng, 5000)
    pq[:600] = pt[:600]                                     # self pairs: score far above 255, diagonal 0
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    ident = (pq == pt)
    for sw_mode, cov_mode in ((2, 2), (2, 0), (2, 1)):   # (the compact variants are swMode 2 only)
        par = gpu.sw_params(mat, int(ps.offsets[-1]), sw_mode=sw_mode, cov_mode=cov_mode)
        idx0, c0, pool0 = gpu.sw_align(par, ss, ss, pq, pt, identity=ident, compact=True)
        true_diag = np.zeros(len(pq), np.uint16)
        for dg in (true_diag, rng.integers(0, 65536, len(pq)).astype(np.uint16), np.full(len(pq), 65535, np.uint16),
                   np.full(len(pq), 3, np.uint16), np.where(np.arange(len(pq)) % 2 == 0, 0, 0x8000).astype(np.uint16)):
            idx1, c1, pool1 = gpu.sw_align(par, ss, ss, pq, pt, identity=ident, compact=True, diag=dg)
            assert np.array_equal(idx0, idx1)
            for f in ('score', 'qStart', 'qEnd', 'tStart', 'tEnd', 'identical', 'btLen', 'flags', 'evalue'):
                assert np.array_equal(c0[f], c1[f]), (sw_mode, f, np.flatnonzero(c0[f] != c1[f])[:5])
            for x in range(0, len(idx0), 13):
                a = pool0[int(c0['btOffset'][x]):int(c0['btOffset'][x]) + int(c0['btLen'][x])]
                b = pool1[int(c1['btOffset'][x]):int(c1['btOffset'][x]) + int(c1['btLen'][x])]
                assert np.array_equal(a, b), x
        assert int((c0['score'][:50] > 255).sum()) > 0 or len(idx0) > 0


@pytest.mark.parametrize('sw_mode,cov_mode,cov_thr,eval_thr', [(0, 2, 0.8, 10.0), (1, 2, 0.8, 10.0), (2, 0, 0.5, 10.0),
                                                               (2, 1, 0.7, 1e-3), (1, 0, 0.9, 1e-5), (2, 2, 0.0, 1e-10)])
def test_sw_modes_and_gates(gpu, host, oracle, small_proteomes, sw_mode, cov_mode, cov_thr, eval_thr):
    """alignment modes (score only / + start positions / + backtrace), coverage modes 0-2 and E-value thresholds: which
    gate stops a pair decides which fields are filled (StripedSmithWaterman.cpp:389-398,483-489)"""
    ps = small_proteomes
    rng = np.random.default_rng(101 + sw_mode * 7 + cov_mode)
    pq, pt = _pairs(ps, rng, n_random=60)
    pq, pt = pq[::3], pt[::3]
    sw_bias, _, _ = host.comp_bias(ps.residues, ps.offsets)
    mat, _, _ = host.matrix(0)
    db = int(ps.offsets[-1])
    ss = gpu.seqset(ps.residues, ps.offsets, sw_bias)
    par = gpu.sw_params(mat, db, sw_mode=sw_mode, eval_thr=eval_thr, cov_mode=cov_mode, cov_thr=cov_thr)
    res, pool = gpu.sw_align(par, ss, ss, pq, pt, identity=(pq == pt))
    n_start = n_bt = 0
    for x in range(len(pq)):
        o = oracle.sw_align(_seq(ps, pq[x]), _seq(ps, pt[x]), db, sw_mode=sw_mode, eval_thr=eval_thr, cov_mode=cov_mode,
                            cov_thr=cov_thr, identity=bool(pq[x] == pt[x]))
        r = res[x]
        assert int(r['score']) == o['score'], (x, r, o)
        assert (int(r['qStart']), int(r['qEnd']), int(r['tStart']), int(r['tEnd'])) == \
               (o['qStart'], o['qEnd'], o['tStart'], o['tEnd']), (x, r, o)
        assert int(r['btLen']) == o['btLen'], (x, r, o)
        if o['evalue'] <= 2.0 * eval_thr:
            assert float(r['evalue']) == o['evalue'], (x, r['evalue'], o['evalue'])
        n_start += o['qStart'] >= 0
        if o['btLen'] > 0 and sw_mode == 2:
            n_bt += 1
            bt = pool[int(r['btOffset']):int(r['btOffset']) + int(r['btLen'])].tobytes().decode()
            assert bt == o['backtrace'] and int(r['identical']) == o['identical'], x
    if sw_mode >= 1:
        assert n_start > 10
    if sw_mode == 2:
        assert n_bt > 10
