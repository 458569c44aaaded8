#!/usr/bin/env python3
"""Reads the counter CSVs of tools/fetch_calib.sh and prints, per access pattern, the requested bytes next to what the counters
say: FETCH_SIZE / WRITE_SIZE as rocprofv3 derives them, and the exact byte count from the per-size request counters of gfx950
(32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B; 32 x (WRREQ - WRREQ_64B) + 64 x WRREQ_64B).  The JSON it writes
({pattern: {requested, fetch_size, fetch_exact, fetch_factor, write_size, write_exact, ...}}) is what tools/pmc_summary.py applies
instead of a blanket x2.  usage: fetch_calib.py DIR OUT.json"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d):
    """{kernel: {counter: mean value per launch of the SECOND launch onwards}}"""
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(d, 'pass*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r['Kernel_Name'].split('(')[0].split()[-1]
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in cs.items()} for k, cs in acc.items()}


def main(d, out_path):
    req = json.load(open(os.path.join(d, 'requested.json')))
    cnt = load(d)
    res = {}
    print('%-20s %12s %12s %12s %8s %12s %12s %8s   %s' % ('pattern', 'requested_MB', 'FETCH_SIZE', 'fetch_exact', 'factor', 'WRITE_SIZE', 'write_exact',
                                                           'factor', 'requests 32B / 64B / 128B | wr 32B / 64B'))
    for k, rq in req.items():
        c = cnt.get(k, {})
        fs = c.get('FETCH_SIZE', 0.0) * 1024
        ws = c.get('WRITE_SIZE', 0.0) * 1024
        r32, r64, r128 = c.get('TCC_EA0_RDREQ_32B_sum', 0.0), c.get('TCC_EA0_RDREQ_64B_sum', 0.0), c.get('TCC_EA0_RDREQ_128B_sum', 0.0)
        w, w64 = c.get('TCC_EA0_WRREQ_sum', 0.0), c.get('TCC_EA0_WRREQ_64B_sum', 0.0)
        fe = 32 * r32 + 64 * r64 + 128 * r128
        we = 32 * (w - w64) + 64 * w64
        res[k] = dict(requested=rq, fetch_size=fs, fetch_exact=fe, fetch_factor=fe / fs if fs > 0 else None, write_size=ws, write_exact=we,
                      write_factor=we / ws if ws > 0 else None, rdreq_32b=r32, rdreq_64b=r64, rdreq_128b=r128, rdreq=c.get('TCC_EA0_RDREQ_sum'),
                      bubble=c.get('TCC_BUBBLE_sum'), wrreq=w, wrreq_64b=w64)
        print('%-20s %12.1f %12.1f %12.1f %8.2f %12.1f %12.1f %8.2f   %.3g / %.3g / %.3g | %.3g / %.3g'
              % (k, rq / 1e6, fs / 1e6, fe / 1e6, fe / fs if fs > 0 else 0.0, ws / 1e6, we / 1e6, we / ws if ws > 0 else 0.0, r32, r64, r128, w - w64, w64))
    json.dump(res, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
