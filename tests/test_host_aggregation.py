"""CPU: the host stage between `align` and `clusterhits` (csrc/host/sd_glue.cpp: prefixid -> besthitbyset -> mergeresultsbyset ->
combinehits fused, R/data/clustersearch.sh:121-151) on synthetic alignment records -- no GPU involved: the stage takes plain arrays.

* the two text conversions every matched hit goes through -- "%.3E" + strtod (Matcher.cpp:288, besthitbyset.cpp:129,
  combinehits.cpp:218-221) and Matcher::compressAlignment (Matcher.cpp:166-185) -- against Python's own printf / float and a
  letter-by-letter run-length loop, on values around every rounding boundary and on backtraces of every length around the 32-letter steps;
* sd_agg_add / finish / get / records against the independent numpy restatement oracle/agg_restatement.py: entries, hits in order,
  P-value bit patterns, and every text field of the cluster records."""
import ctypes as C
import math
import os
import struct
import sys

import numpy as np
import pytest

from spacedust_amd import _lib, api
from spacedust_amd._lib import ptr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))


def _q3e(L, v):
    text = C.create_string_buffer(32)
    back = C.c_double()
    assert L.sd_host_quantise_3e(float(v), text, C.byref(back)) == 0
    return text.value.decode(), back.value


def test_percent_3e_round_trip_equals_printf_and_strtod():
    L = _lib.load()
    rng = np.random.default_rng(5)
    values = []
    # random significands over the whole exponent range (E-values, log P and P-values live between 1e-300 and 1e3)
    mant = 1.0 + rng.random(200000)
    expo = rng.integers(-1020, 1020, 200000)
    values += list(np.ldexp(mant, expo) * rng.choice([-1.0, 1.0], 200000))
    # around every rounding boundary (m + 0.5) * 10^k and every m * 10^k of a sample of (m, k), three doubles either side
    for k in range(-310, 300, 3):
        for m in rng.integers(1000, 10000, 12):
            for half in ('', '.5'):
                v = float('%d%se%d' % (m, half, k))
                lo = hi = v
                for _ in range(3):
                    values += [lo, hi, -lo]
                    lo, hi = np.nextafter(lo, 0.0), np.nextafter(hi, np.inf)
    values += [0.0, 1.0, 10.0, 1000.5, 0.5, 9999.5, 99995.0, 999.95, 9.9995, 1e22, 1e23, 1e-22, 1e-23, 5e-324, 1e-310, sys.float_info.min,
               1.0005, 2.5e-7, 1.1e-6, 10e-7, math.log(10e-7), math.log(sys.float_info.min), 9.9995e-5, 9.9994999e-5]
    bad = []
    for v in values:
        v = float(v)
        text, back = _q3e(L, v)
        want = '%.3E' % v
        if text != want or struct.pack('<d', back) != struct.pack('<d', float(want)):
            bad.append((v, text, back, want))
    assert not bad, bad[:5]
    assert len(values) > 220000


def _rle(bt):
    # Matcher::compressAlignment, letter by letter
    out, state, count = [], 'M', 0
    for c in bt:
        if c != state:
            out.append('%d%s' % (count, state))
            state, count = c, 1
        else:
            count += 1
    out.append('%d%s' % (count, state))
    return ''.join(out)


def _compress(L, bt):
    raw = bt.encode()
    n = C.c_uint64()
    assert L.sd_host_compress_backtrace(raw, len(raw), None, 0, C.byref(n)) == 0
    out = C.create_string_buffer(int(n.value) + 1)
    assert L.sd_host_compress_backtrace(raw, len(raw), out, n.value, C.byref(n)) == 0
    return out.raw[:n.value].decode()


def test_backtrace_compression_equals_letter_by_letter_loop():
    L = _lib.load()
    rng = np.random.default_rng(9)
    cases = ['', 'M', 'I', 'D', 'MI', 'IM', 'M' * 31, 'M' * 32, 'M' * 33, 'M' * 64 + 'I', 'I' * 65, 'M' * 1200, 'D' * 1000 + 'M',
             'MID' * 50, 'M' * 999 + 'I' * 1000 + 'D' * 1001]
    for n in list(range(0, 140)) + [255, 256, 257, 511, 700, 4097]:
        for mode in range(3):
            if mode == 0:       # a protein alignment: match runs of dozens of letters, short gaps
                letters = rng.choice(['M', 'I', 'D'], n, p=[0.96, 0.02, 0.02])
            elif mode == 1:     # every letter as likely as the others: runs of one or two
                letters = rng.choice(['M', 'I', 'D'], n)
            else:               # long runs of every letter (run lengths of three and four digits)
                letters = np.repeat(rng.choice(['M', 'I', 'D'], n // 40 + 1), rng.integers(1, 120, n // 40 + 1))[:n]
            cases.append(''.join(letters))
    for bt in cases:
        assert _compress(L, bt) == _rle(bt), bt
    assert _compress(L, '') == '0M' and _compress(L, 'IIM') == '0M2I1M'
    assert L.sd_host_compress_backtrace(b'MMMM', 4, C.create_string_buffer(1), 1, C.byref(C.c_uint64())) == -4   # SD_ENOMEM


class _SwResult(C.Structure):
    _fields_ = [('score', C.c_int32), ('qStart', C.c_int32), ('qEnd', C.c_int32), ('tStart', C.c_int32), ('tEnd', C.c_int32),
                ('identical', C.c_int32), ('btLen', C.c_int32), ('flags', C.c_int32), ('evalue', C.c_double), ('btOffset', C.c_uint64)]


def _seq_id_text(identical, bt_len, identity):
    # Util::fastSeqIdToBuffer as the alignment DB shows it (csrc/host/sd_evalue.cpp: seqIdToBuffer): truncation, "1.00" for one
    if identity:
        return '1.00'
    s = np.float32(identical) / np.float32(bt_len)
    if s == 1.0:
        return '1.00'
    t = '0.'
    if s < 0.10:
        t += '0'
    if s < 0.01:
        t += '0'
    return t + str(int(np.float32(s) * np.float32(1000)))


@pytest.mark.parametrize('threads,pool_form', [(1, 0), (5, 0), (5, 1)])
def test_aggregation_on_synthetic_records_equals_restatement(threads, pool_form, monkeypatch):
    """3 query sets x 40 target sets, several accepted candidates per (query, target set) cell incl. equal scores (the compareHits tie
    breaks: target length, then key), E-values either side of combinehits' bound, coverage and length either side of their thresholds,
    the records handed over in chunks like the pipeline does.  Entries, hit order, P-value bits against oracle/agg_restatement.py;
    the pval / seqId / eval texts, the coordinates and the CIGAR of every member of the cluster records against Python's formatting.
    pool_form 1 (sd_agg_set_pool_form): the pool holds the run-length text of every backtrace -- what the alignment calls return after
    sd_sw_set_cigar_pool -- with its length in flags >> 8; same entries, same records."""
    import agg_restatement
    L = _lib.load()
    host = api.Host(threads)
    monkeypatch.setenv('OMP_NUM_THREADS', str(threads))
    rng = np.random.default_rng(17)
    n_sets, per_set, n_qsets = 40, 50, 3
    n_t, n_q = n_sets * per_set, n_qsets * per_set
    lengths = rng.integers(40, 600, n_t).astype(np.int32)
    set_of = (np.arange(n_t) // per_set).astype(np.uint32)
    db_res = int(lengths.sum()) * 2000
    pq, pt, recs, ident, pool, rows, bts = [], [], [], [], [], [], []
    off = 0
    for q in range(n_q):
        n_pairs = int(rng.integers(0, 90))
        targets = rng.choice(n_t, n_pairs, replace=False)
        if q % 3 == 0 and q not in targets:
            targets = np.append(targets, q)   # the identity pair
        # some queries see the same scores over and over: ties down to target length and key
        scores = rng.integers(25, 60, len(targets)) if q % 7 == 0 else rng.integers(20, 900, len(targets))
        for t, sc in zip(targets, scores):
            t, sc = int(t), int(sc)
            ql = int(lengths[q])
            is_id = t == q
            ev = host.evalue(db_res, sc, ql)
            r = _SwResult()
            r.score, r.evalue, r.flags = sc, ev, 0
            kind = rng.integers(0, 12)
            if kind == 0 and not is_id:      # stopped at a gate: no coordinates, no backtrace
                r.qStart = r.tStart = -1
                r.qEnd = r.tEnd = r.identical = r.btLen = 0
                bt = ''
            else:
                cov = rng.uniform(0.7, 1.0) if kind < 4 else rng.uniform(0.82, 1.0)
                span = max(1, min(ql, int(round(cov * ql))))
                r.qStart = int(rng.integers(0, ql - span + 1))
                r.qEnd = r.qStart + span - 1
                letters = rng.choice(['M', 'I', 'D'], span + int(rng.integers(0, 6)), p=[0.95, 0.025, 0.025])
                if kind == 1:
                    letters = letters[:int(rng.integers(20, 40))]   # either side of --min-aln-len 30
                bt = ''.join(letters)
                r.btLen = len(bt)
                r.identical = int(rng.integers(0, bt.count('M') + 1)) if rng.integers(0, 6) else bt.count('M')
                r.tStart = int(rng.integers(0, 5))
                r.tEnd = r.tStart + span - 1
            r.btOffset = off
            bts.append(bt)
            if pool_form == 1:
                txt = _rle(bt) if bt else ''
                r.flags = len(txt) << 8 | int(rng.integers(0, 2))
                pool.append(txt)
                off += len(txt)
            else:
                pool.append(bt)
                off += len(bt)
            pq.append(q)
            pt.append(t)
            recs.append(r)
            ident.append(1 if is_id else 0)
            rows.append((q, t, sc, ev, host.bitscore(sc), r.qStart, r.qEnd, r.btLen))
    pool = ''.join(pool).encode() + b' '
    n = len(pq)
    assert n > 5000
    want = agg_restatement.aggregate(np.array(rows, np.float64), lengths, set_of.astype(np.int64))
    assert len(want) >= 100 and sum(len(v) for v in want.values()) > 1000

    agg = C.c_void_p()
    q_len = np.ascontiguousarray(lengths[:n_q])
    q_set = np.ascontiguousarray(set_of[:n_q])
    assert L.sd_agg_create(ptr(q_set), ptr(q_len), n_q, ptr(set_of), ptr(lengths), n_t, n_qsets, n_sets, 10.0, 2, 0.8, 30, 1, C.byref(agg)) == 0
    assert L.sd_agg_set_pool_form(agg, pool_form) == 0
    pq, pt, ident = np.array(pq, np.uint32), np.array(pt, np.uint32), np.array(ident, np.uint8)
    rec_arr = (_SwResult * n)(*recs)
    rec_bytes = np.frombuffer(rec_arr, np.uint8)
    step = 37   # queries per hand-over
    for c0 in range(0, n_q, step):
        sel = np.nonzero((pq >= c0) & (pq < c0 + step))[0]
        if len(sel) == 0:
            continue
        a, b = int(sel[0]), int(sel[-1]) + 1
        local_q = np.ascontiguousarray(pq[a:b] - c0)
        part = np.ascontiguousarray(rec_bytes[a * C.sizeof(_SwResult):b * C.sizeof(_SwResult)])
        assert L.sd_agg_add(agg, b - a, c0, ptr(local_q), ptr(np.ascontiguousarray(pt[a:b])), ptr(part), ptr(np.ascontiguousarray(ident[a:b])),
                            C.c_char_p(pool)) == 0
    ne, nh = C.c_uint64(), C.c_uint64()
    assert L.sd_agg_finish(agg, C.byref(ne), C.byref(nh)) == 0
    ne, nh = int(ne.value), int(nh.value)
    e_off, e_q, e_t = np.zeros(ne + 1, np.uint64), np.zeros(ne, np.uint32), np.zeros(ne, np.uint32)
    h_q, h_t, h_p = np.zeros(nh, np.uint32), np.zeros(nh, np.uint32), np.zeros(nh, np.float64)
    assert L.sd_agg_get(agg, ptr(e_off), ptr(e_q), ptr(e_t), ptr(h_q), ptr(h_t), ptr(h_p)) == 0
    got = {}
    for e in range(ne):
        a, b = int(e_off[e]), int(e_off[e + 1])
        got[(int(e_q[e]), int(e_t[e]))] = [(int(h_q[x]), int(h_t[x]), float(h_p[x])) for x in range(a, b)]
    assert sorted(got) == sorted(want)
    for k in want:
        assert [(q, t, struct.pack('<d', p)) for q, t, p in got[k]] == [(q, t, struct.pack('<d', p)) for q, t, p in want[k]], k

    # the cluster records: every entry as one cluster of all its hits
    cl_of = np.zeros(nh, np.uint32)
    rank = np.zeros(nh, np.uint32)
    n_cl = np.ones(ne, np.uint32)
    size = np.zeros(nh, np.uint32)
    pco, pmh = np.zeros(nh, np.float64), np.zeros(nh, np.float64)
    for e in range(ne):
        a, b = int(e_off[e]), int(e_off[e + 1])
        rank[a:b] = np.arange(b - a)
        size[a] = b - a
    nbytes = C.c_uint64()
    assert L.sd_agg_records(agg, ptr(cl_of), ptr(rank), ptr(n_cl), ptr(pco), ptr(pmh), ptr(size), None, 0, C.byref(nbytes)) == 0
    buf = np.zeros(int(nbytes.value), np.uint8)
    assert L.sd_agg_records(agg, ptr(cl_of), ptr(rank), ptr(n_cl), ptr(pco), ptr(pmh), ptr(size), ptr(buf), nbytes.value, C.byref(nbytes)) == 0
    raw = buf.tobytes()
    by_pair = {(int(pq[i]), int(pt[i])): i for i in range(n)}
    p = 0
    members = 0
    for e in range(ne):
        m, qs, ts, _, _, _ = struct.unpack_from('<IIIIdd', raw, p)
        p += 32
        assert (qs, ts) == (int(e_q[e]), int(e_t[e])) and m == int(e_off[e + 1] - e_off[e])
        for x in range(m):
            q, t, pval, seq_id, ev, q_start, q_end, q_l, t_start, t_end, t_l, c_len = struct.unpack_from('<II16s8s16siiiiiiI', raw, p)
            p += 76
            cigar = raw[p:p + c_len].decode()
            p += (c_len + 3) & ~3
            i = by_pair[(q, t)]
            r = recs[i]
            h = int(e_off[e]) + x
            assert (q, t) == (int(h_q[h]), int(h_t[h]))
            assert ev == ('%.3E' % r.evalue).encode().ljust(16, b'\0')   # zero behind the text: the record bytes are the same every run
            assert pval == ('%.3E' % h_p[h]).encode().ljust(16, b'\0')
            assert seq_id == _seq_id_text(r.identical, r.btLen, ident[i]).encode().ljust(8, b'\0')
            assert (q_start, q_end, q_l, t_start, t_end, t_l) == (r.qStart, r.qEnd, int(lengths[q]), r.tStart, r.tEnd, int(lengths[t]))
            bt = bts[i]
            assert len(bt) == r.btLen and cigar == _rle(bt)
            if bt.startswith('M'):   # (a backtrace that begins with a gap begins "0M", which Matcher::uncompressAlignment reads as one M)
                assert api.uncompress_cigar(cigar) == bt
            members += 1
    assert p == len(raw) and members == nh
    L.sd_agg_destroy(agg)


def test_aggregation_list_order_keeps_the_first_of_equal_evalues():
    """sd_agg_set_list_order: the records of a query are the lines of a MERGED alignment DB (`search --num-iterations`: mergedbs
    concatenates the iterations' sorted lists), and besthitbyset keeps the first line with a strictly smaller %.3E text E-value
    (R/src/util/besthitbyset.cpp:88-101).  Two targets of one set with the same raw score have the same E-value: the default rule
    (one sorted list, any arrival order) takes the compareHits minimum -- the shorter target --, the list-order rule the earlier line;
    a strictly smaller E-value later in the list wins under both."""
    L = _lib.load()
    host = api.Host(1)
    lengths = np.array([300, 400, 100, 250, 260], np.int32)   # query 0 is sequence 0; targets 1, 2, 3 in set 1, target 4 in set 2
    set_of = np.array([0, 1, 1, 1, 2], np.uint32)
    db_res = 10 ** 9

    def rec(score, off):
        r = _SwResult()
        r.score, r.evalue, r.flags = score, host.evalue(db_res, score, 300), 0
        r.qStart, r.qEnd, r.tStart, r.tEnd, r.btLen, r.identical, r.btOffset = 0, 279, 0, 279, 280, 150, off
        return r

    pool = b'M' * 280 + b' '
    for targets, scores, want_default, want_list in (([1, 2], [200, 200], 2, 1),          # equal E-values: shorter target | earlier line
                                                     ([1, 2, 3], [200, 200, 260], 3, 3),   # a strictly smaller E-value later in the list
                                                     ([2, 1], [200, 200], 2, 2)):          # the earlier line is also the compareHits minimum
        for list_order, want in ((0, want_default), (1, want_list)):
            agg = C.c_void_p()
            assert L.sd_agg_create(ptr(set_of[:1]), ptr(lengths[:1]), 1, ptr(set_of), ptr(lengths), 5, 1, 3, 10.0, 2, 0.8, 30, 1, C.byref(agg)) == 0
            assert L.sd_agg_set_list_order(agg, list_order) == 0
            n = len(targets)
            recs = (_SwResult * n)(*[rec(s, 0) for s in scores])
            pq, pt, ident = np.zeros(n, np.uint32), np.array(targets, np.uint32), np.zeros(n, np.uint8)
            assert L.sd_agg_add(agg, n, 0, ptr(pq), ptr(pt), ptr(np.frombuffer(recs, np.uint8)), ptr(ident), C.c_char_p(pool)) == 0
            ne, nh = C.c_uint64(), C.c_uint64()
            assert L.sd_agg_finish(agg, C.byref(ne), C.byref(nh)) == 0
            assert (ne.value, nh.value) == (1, 1)
            e_off, e_q, e_t = np.zeros(2, np.uint64), np.zeros(1, np.uint32), np.zeros(1, np.uint32)
            h_q, h_t, h_p = np.zeros(1, np.uint32), np.zeros(1, np.uint32), np.zeros(1, np.float64)
            assert L.sd_agg_get(agg, ptr(e_off), ptr(e_q), ptr(e_t), ptr(h_q), ptr(h_t), ptr(h_p)) == 0
            assert (int(e_q[0]), int(e_t[0]), int(h_t[0])) == (0, 1, want), (targets, scores, list_order)
            L.sd_agg_destroy(agg)
