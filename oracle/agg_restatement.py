"""oracle/agg_restatement.py -- TEST INFRASTRUCTURE ONLY (tests/, bench.py's parity leg, tools/cpu_baseline.py).

A plain numpy restatement of what lies between the alignment results and `clusterhits` in R/data/clustersearch.sh:121-151,
written from the reference's module sources and independent of csrc/host/sd_glue.cpp (the product's fused form):

  Alignment::run's accept rule          M/src/alignment/Alignment.cpp:380-470 (E-value, query coverage, alignment length;
                                        the identity hit is always kept), result order Matcher::compareHits
  besthitbyset --simple-best-hit 1      R/src/util/besthitbyset.cpp:41-144: per (query, target set) the hit with the smallest
                                        E-value as the alignment DB prints it (%.3E, Matcher.cpp:288), first in result order on
                                        ties; column 2 becomes log(P) through ComputelogPval (:49-63), printed %.3E
  combinehits (multihit mode)           R/src/util/combinehits.cpp:74-234: --filter-self-match drops qset == tset (:83), keeps
                                        hits with log P < log(10e-7) (:101-113), prints exp(log P) as %.3E (:213-217)

Input: the rows of oracle/ref_driver.cpp:ref_run_query_set (the reference's own prefilter + Smith-Waterman on whole query
sets): query, target, score, E-value, bit score, qStart, qEnd, backtrace length.
Output: {(query set, target set): [(query, target, P-value), ...]} in the order combinehits writes them (ascending query
inside an entry: mergeresultsbyset concatenates the per-query lines in key order)."""
import math
import sys

import numpy as np


def _e3(x):
    """the value a double has after a `%.3E` print and a strtod (every hand-off between the modules is text)"""
    return float('%.3E' % x)


def _logpval(ev):
    """ComputelogPval(eval, log(1)), besthitbyset.cpp:49-63"""
    if ev == 0:
        return math.log(sys.float_info.min)
    if 0 < ev < 10e-4:
        return math.log(ev)
    return math.log(1 - math.exp(-ev))


def aggregate(rows, lengths, set_id, eval_thr=10.0, cov_thr=0.8, aln_len_thr=30, filter_self_match=True):
    rows = np.asarray(rows, np.float64).reshape(-1, 8)
    q = rows[:, 0].astype(np.int64)
    t = rows[:, 1].astype(np.int64)
    ev, bits = rows[:, 3], rows[:, 4]
    q_start, q_end, bt_len = rows[:, 5], rows[:, 6], rows[:, 7]
    lengths = np.asarray(lengths, np.int64)
    # accept rule (the identity pair is kept whatever its fields)
    qcov = (q_end - q_start + 1).astype(np.float32) / lengths[q].astype(np.float32)
    ok = (bt_len > 0) & (q_start >= 0) & (ev <= eval_thr) & (qcov >= np.float32(cov_thr)) & (bt_len >= aln_len_thr)
    ok |= q == t
    q, t, ev, bits = q[ok], t[ok], ev[ok], bits[ok]
    # Matcher::compareHits: E-value ascending, bit score (rounded) descending, target length ascending, target id
    order = np.lexsort((t, lengths[t], -np.floor(bits + 0.5), ev))
    q, t, ev = q[order], t[order], ev[order]
    ts = np.asarray(set_id, np.int64)[t]
    # best hit per (query, target set): the first in result order
    key = q * (int(np.max(set_id)) + 1) + ts
    _, first = np.unique(key, return_index=True)   # index of the first occurrence of every key in the sorted order
    out = {}
    thr = math.log(10e-7)
    qs_of = np.asarray(set_id, np.int64)
    # a loose bound before the text round trips: log P < log(1e-6) needs E < ~1e-6
    for i in first[ev[first] < 2e-6]:
        qi, ti = int(q[i]), int(t[i])
        qs, tset = int(qs_of[qi]), int(ts[i])
        if filter_self_match and qs == tset:
            continue
        lp = _e3(_logpval(_e3(float(ev[i]))))
        if lp < thr:
            out.setdefault((qs, tset), []).append((qi, ti, _e3(math.exp(lp))))
    for k in out:
        out[k].sort(key=lambda r: r[0])
    return out
