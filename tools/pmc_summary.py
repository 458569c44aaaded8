#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the PMC slot
budget requires).  Counter values are KiB per dispatch; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950 (128-B requests tallied at 64 B).  That factor is calibrated for this repo's access patterns
(tools/fetch_calib.sh -> profiles/r05_fetch_calibration.*): on gfx950 EVERY read request of the L2 is a 128-B one -- also for
one random 8-B read per lane -- so 2 x FETCH_SIZE = 128 x TCC_EA0_RDREQ_128B exactly in all ten patterns, and WRITE_SIZE is
exact (32-B and 64-B write requests are tallied apart).  With a third CSV (a pass of TCC_EA0_RDREQ_32B_sum / _64B_sum /
_128B_sum) the exact byte count is printed beside 2 x FETCH_SIZE per kernel.
With the bench line of the PMC run (bench.py prints prefilter_queries_in_process: every query its process ran through the prefilter,
warm-up, timed steps and both calls of the isolated leg) the JSON also carries bytes PER QUERY for every kernel group and for the
prefilter stage as a whole (_prefilter_total) -- the figure to hold against SURVEY.md 8(d)'s bytes per query; bytes per launch of a
PMC run cannot be compared with another run's launches (they hold different numbers of queries).
Usage: pmc_summary.py <fetch counter_collection.csv> <write ...csv> [out.json [rdreq ...csv | - [bench line .json of the PMC run]]]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([A-Za-z0-9_:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:70]


def load(path, counter=None, scale=1.0):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if counter is not None and r.get('Counter_Name') != counter:
            continue
        a = acc[short(r['Kernel_Name'])]
        a[0] += 1
        a[1] += float(r['Counter_Value']) * scale
    return acc


def load_exact(path):
    """bytes per kernel from the per-size read request counters"""
    tot = defaultdict(float)
    for name, size in (('TCC_EA0_RDREQ_32B_sum', 32), ('TCC_EA0_RDREQ_64B_sum', 64), ('TCC_EA0_RDREQ_128B_sum', 128)):
        for k, v in load(path, name, size).items():
            tot[k] += v[1]
    return tot


def main(fetch_csv, write_csv, exact_csv=None):
    f, w = load(fetch_csv), load(write_csv)
    ex = load_exact(exact_csv) if exact_csv else {}
    rows = []
    for k in set(f) | set(w):
        n = max(f.get(k, [0, 0])[0], w.get(k, [0, 0])[0])
        fb = 2.0 * f.get(k, [0, 0.0])[1] * 1024
        wb = w.get(k, [0, 0.0])[1] * 1024
        rows.append((fb + wb, k, n, fb, wb))
    rows.sort(reverse=True)
    print('%-72s %7s %14s %14s %16s%s' % ('kernel', 'calls', 'fetch_GB(x2)', 'write_GB', 'bytes_per_call', '   fetch_GB(exact: 32/64/128-B requests)' if ex else ''))
    for tot, k, n, fb, wb in rows[:40]:
        print('%-72s %7d %14.3f %14.3f %16.0f%s' % (k, n, fb / 1e9, wb / 1e9, tot / max(n, 1), '   %14.3f' % (ex.get(k, 0.0) / 1e9) if ex else ''))


# first match wins: the oversize-bucket launch of bucket_match is its own group (bench.py times it as prefilter_bucket_match_big)
GROUPS = [('sdpk::sw_score_pk', 'sw_score_pk'), ('sw_score_kernel', 'sw_score'), ('sw_traceback', 'sw_traceback'),
          ('emit_kmers', 'prefilter_emit_kmers'), ('count_kmers', 'prefilter_count_kmers'), ('gather_hits', 'prefilter_gather_hits'),
          ('kp_hist', 'prefilter_kmer_partition'), ('kp_scatter', 'prefilter_kmer_partition'), ('join_count', 'prefilter_join_count'),
          ('join_scatter', 'prefilter_join_scatter'), ('hot_filter', 'prefilter_hot_filter'), ('segment_match', 'prefilter_segment_match'),
          ('partition_hits', 'prefilter_partition_hits'), ('bucket_match_kernel<256', 'prefilter_bucket_match_big'),
          ('bucket_match', 'prefilter_bucket_match'), ('coarse_', 'prefilter_coarse_split'),
          ('score_diag', 'prefilter_score_diag'), ('select_hits_big', 'prefilter_select_hits_big'), ('select_hits', 'prefilter_select_hits'),
          ('clusterhits', 'clusterhits')]


def to_json(fetch_csv, write_csv, out_path, queries=0):
    """the same numbers grouped the way bench.py names its kernel groups (variants of one template summed)"""
    import json
    f, w = load(fetch_csv), load(write_csv)
    out = {}
    for k in set(f) | set(w):
        g = next((name for key, name in GROUPS if key in k), None)
        if g is None:
            continue
        o = out.setdefault(g, dict(launches=0, fetch=0.0, write=0.0))
        if not any(x in k for x in ('kp_hist', 'coarse_count', 'coarse_offsets')):   # bench.py times kp_hist + kp_scatter (coarse count + offsets + scatter) as one launch
            o['launches'] += max(f.get(k, [0, 0])[0], w.get(k, [0, 0])[0])
        o['fetch'] += 2.0 * f.get(k, [0, 0.0])[1] * 1024
        o['write'] += w.get(k, [0, 0.0])[1] * 1024
    res = {g: dict(launches=o['launches'], fetch_bytes_per_launch=o['fetch'] / max(o['launches'], 1),
                   write_bytes_per_launch=o['write'] / max(o['launches'], 1),
                   bytes_per_launch=(o['fetch'] + o['write']) / max(o['launches'], 1)) for g, o in out.items()}
    if queries > 0:
        tot = dict(fetch=0.0, write=0.0)
        for g, o in out.items():
            res[g].update(queries=queries, fetch_bytes_per_query=o['fetch'] / queries, write_bytes_per_query=o['write'] / queries,
                          bytes_per_query=(o['fetch'] + o['write']) / queries)
            if g.startswith('prefilter_'):
                tot['fetch'] += o['fetch']
                tot['write'] += o['write']
        res['_prefilter_total'] = dict(queries=queries, fetch_bytes_per_query=tot['fetch'] / queries, write_bytes_per_query=tot['write'] / queries,
                                       bytes_per_query=(tot['fetch'] + tot['write']) / queries,
                                       note='every prefilter_* kernel group of the PMC run: 2 x FETCH_SIZE + WRITE_SIZE over the queries its process ran '
                                            'through the prefilter (bench.py: prefilter_queries_in_process)')
        print('prefilter stage: %.1f MB per query at the memory side (%.1f fetched, %.1f written) over %d queries'
              % (res['_prefilter_total']['bytes_per_query'] / 1e6, tot['fetch'] / queries / 1e6, tot['write'] / queries / 1e6, queries))
        for g in sorted((g for g in res if g.startswith('prefilter_')), key=lambda g: -res[g]['bytes_per_query']):
            print('  %-32s %8.2f MB per query (fetch %.2f, write %.2f)' % (g, res[g]['bytes_per_query'] / 1e6, res[g]['fetch_bytes_per_query'] / 1e6,
                                                                            res[g]['write_bytes_per_query'] / 1e6))
    json.dump(res, open(out_path, 'w'), indent=1)


def queries_of(bench_json):
    import json
    for line in open(bench_json):
        if line.startswith('{'):
            return int(json.loads(line).get('prefilter_queries_in_process') or 0)
    return 0


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] != '-' else None)
    if len(sys.argv) > 3:
        to_json(sys.argv[1], sys.argv[2], sys.argv[3], queries_of(sys.argv[5]) if len(sys.argv) > 5 else 0)
