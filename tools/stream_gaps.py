#!/usr/bin/env python3
"""Idle gaps per HIP stream from a rocprofv3 --kernel-trace results.db: for every stream (queue) the kernels in start order, the time
between the end of one and the start of the next, summed by the kernel that follows the gap.  Shows where a lane's stream waits for
its host thread (sync round trips, uploads) rather than for the device.  usage: stream_gaps.py results.db [min_gap_us]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([A-Za-z0-9_:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:48]


def main(path, min_gap_us=5.0):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    print('columns:', cols)
    sid = next((c for c in ('stream_id', 'stream', 'queue_id', 'queue') if c in cols), None)
    if sid is None:
        print('no stream / queue column'); return
    rows = db.execute('select %s, name, start, end from kernels order by start' % sid).fetchall()
    by = defaultdict(list)
    for s, n, a, b in rows:
        by[s].append((a, b, short(n)))
    t0 = min(r[2] for r in rows); t1 = max(r[3] for r in rows)
    print('trace span %.1f ms, %d kernels, %d streams' % ((t1 - t0) / 1e6, len(rows), len(by)))
    for s, ks in sorted(by.items(), key=lambda kv: -len(kv[1])):
        if len(ks) < 50:
            continue
        busy = sum(b - a for a, b, _ in ks)
        span = ks[-1][1] - ks[0][0]
        gaps = defaultdict(lambda: [0, 0.0])
        small = 0.0
        for (a0, b0, n0), (a1, b1, n1) in zip(ks, ks[1:]):
            g = (a1 - b0) / 1e3
            if g >= min_gap_us:
                e = gaps[(n0, n1)]
                e[0] += 1; e[1] += g
            elif g > 0:
                small += g
        tot_gap = sum(v[1] for v in gaps.values())
        print('\nstream %s: %d kernels, span %.1f ms, busy %.1f ms (%.0f %%), gaps >= %.0f us: %.1f ms, smaller: %.1f ms' %
              (s, len(ks), span / 1e6, busy / 1e6, 100.0 * busy / span, min_gap_us, tot_gap / 1e3, small / 1e3))
        for (n0, n1), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
            print('   %8.1f ms in %5d gaps (avg %7.0f us)  %s -> %s' % (g / 1e3, c, g / c, n0, n1))


def union_busy(path):
    """how much of the time between the first and the last kernel of the busiest half of the trace at least one kernel runs, and how
    many run at once on average -- the device's idle time (no kernel of any stream) is what more overlap could still fill"""
    db = sqlite3.connect(path)
    rows = db.execute('select start, end from kernels order by start').fetchall()
    if not rows:
        return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # the window: the second half of the trace (the timed steps of bench.py; generation and index build come first)
    w0 = t0 + (t1 - t0) // 2
    ev = []
    for a, b in rows:
        a, b = max(a, w0), min(b, t1)
        if b > a:
            ev.append((a, 1))
            ev.append((b, -1))
    ev.sort()
    cur, last, busy, area = 0, w0, 0, 0
    for t, d in ev:
        if cur > 0:
            busy += t - last
        area += cur * (t - last)
        last = t
        cur += d
    span = t1 - w0
    print('second half of the trace (%.1f ms): at least one kernel running %.1f %% of the time, %.2f kernels at once on average'
          % (span / 1e6, 100.0 * busy / span, area / span))


def idle_report(path, top=24):
    """the periods of the second half of the trace in which NO kernel of any stream runs: total, by length class, and summed by
    (kernel that ended last -> kernel that starts next, with their streams) -- which host round trip the whole device waits for"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    sid = next((c for c in ('stream_id', 'stream', 'queue_id', 'queue') if c in cols), None)
    rows = db.execute('select start, end, name, %s from kernels order by start' % sid).fetchall()
    if not rows:
        return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    w0 = t0 + (t1 - t0) // 2
    rows = [r for r in rows if r[1] > w0]
    frontier, last = None, None   # latest end seen so far and the kernel that has it
    gaps = defaultdict(lambda: [0, 0.0])
    classes = [(50, 0, 0.0), (200, 0, 0.0), (1000, 0, 0.0), (5000, 0, 0.0), (1e12, 0, 0.0)]
    total = 0.0
    for a, b, n, s_ in rows:
        if frontier is not None and a > frontier:
            g = (a - frontier) / 1e3
            total += g
            e = gaps[('%s [s%s]' % (last[0], last[1]), '%s [s%s]' % (short(n), s_))]
            e[0] += 1; e[1] += g
            for i, (lim, c, t) in enumerate(classes):
                if g < lim:
                    classes[i] = (lim, c + 1, t + g)
                    break
        if frontier is None or b > frontier:
            frontier, last = b, (short(n), s_)
    print('\nidle device (no kernel on any stream) in the second half of the trace: %.1f ms of %.1f ms' % (total / 1e3, (t1 - w0) / 1e6))
    lo = 0
    for lim, c, t in classes:
        print('   gaps of %5s - %5s us: %6d, %8.1f ms' % (lo, lim if lim < 1e12 else 'inf', c, t / 1e3))
        lo = lim
    for (n0, n1), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:top]:
        print('   %8.1f ms in %5d gaps (avg %7.0f us)  %s -> %s' % (g / 1e3, c, g / c, n0, n1))


def timeline(path, stream, first, count):
    """kernels first .. first + count of one stream in start order: gap before, duration, name, grid"""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    grid = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else 'workgroup_x')
    sid = next((c for c in ('stream_id', 'stream', 'queue_id', 'queue') if c in cols), None)
    rows = db.execute('select name, start, end, %s from kernels where %s = ? order by start' % (grid, sid), (stream,)).fetchall()
    if first < 0:   # count kernels from the middle of the stream
        first = max(0, len(rows) // 2 - count // 2)
    prev = None
    for n, a, b, g in rows[first:first + count]:
        print('%9.3f ms  gap %8.1f us  dur %8.1f us  grid %9d  %s' % ((a - rows[0][1]) / 1e6, (a - prev) / 1e3 if prev else 0.0, (b - a) / 1e3, g, short(n)))
        prev = b


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[2] == '--timeline':
        timeline(sys.argv[1], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    else:
        main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0)
        union_busy(sys.argv[1])
        idle_report(sys.argv[1])
