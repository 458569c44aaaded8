#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results.db (rocpd sqlite) as a per-kernel stats table (the same numbers
`rocprofv3 --stats` prints): calls, total / average / min / max duration, share of GPU time, resources."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([A-Za-z0-9_:<>, ]+?)\(', name)
    n = m.group(1) if m else name
    return n if len(n) < 90 else n[:87] + '...'


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
                      'max(vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) from kernels group by name '
                      'order by sum(duration) desc').fetchall()
    tot = sum(r[2] for r in rows) or 1
    print('%-92s %8s %12s %12s %10s %10s %6s %5s %5s %7s %5s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us',
                                                                  'pct', 'vgpr', 'sgpr', 'lds_B', 'wg'))
    for r in rows:
        print('%-92s %8d %12.3f %12.1f %10.1f %10.1f %6.2f %5d %5d %7d %5d' % (short(r[0]), r[1], r[2] / 1e6, r[3] / 1e3,
                                                                                r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
                                                                                r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))
    print('total GPU kernel time: %.3f ms over %d dispatches' % (tot / 1e6, sum(r[1] for r in rows)))


def dispatches(path, pattern):
    """every dispatch of the kernels whose name contains `pattern`: grid, workgroup, LDS, duration"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    grid = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else 'workgroup_x')
    for r in db.execute('select name, %s, workgroup_x, lds_size, duration from kernels where name like ? order by start' % grid,
                        ('%' + pattern + '%',)):
        print('%-60s grid %9d wg %5d lds %7d  %10.1f us' % (short(r[0])[:60], r[1], r[2], r[3] or 0, r[4] / 1e3))


if __name__ == '__main__':
    if len(sys.argv) > 3 and sys.argv[2] == '--dispatches':
        dispatches(sys.argv[1], sys.argv[3])
    else:
        main(sys.argv[1])
