#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 rocpd sqlite (``--kernel-trace``): every kernel between
two optimizer launches in start order, with its queue / stream, start offset, duration and the idle gap
before it on its own queue.  Shows what the step's critical path is made of.

    python scripts/step_timeline.py <results.db> [--step N] [--min-us 20]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--step', type=int, default=-2)
    ap.add_argument('--min-us', type=float, default=20.0)
    ap.add_argument('--marker', default='adam_flat_kernel|multi_tensor_apply', help="'|'-separated kernel name parts that end a step")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    cols = [d[1] for d in cur.execute('pragma table_info(kernels)')]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    sel = f'select name, start, end, {qcol or 0}, grid_x from kernels order by start'
    rows = cur.execute(sel).fetchall()
    marks = [i for i, r in enumerate(rows) if any(m in r[0] for m in args.marker.split('|'))]
    # consecutive optimizer launches belong to one step: keep the last of each burst
    bursts = [m for k, m in enumerate(marks) if k + 1 == len(marks) or rows[marks[k + 1]][1] - rows[m][2] > 2e6]
    a, b = bursts[args.step - 1], bursts[args.step]
    step = rows[a + 1:b + 1]
    t0 = step[0][1]
    print(f'# step of {(step[-1][2] - t0) / 1e6:.3f} ms, {len(step)} kernels, columns from table kernels: {qcol}')
    last_end = {}
    busy = {}
    small = {}
    for name, s, e, q, grid in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(e, last_end.get(q, 0))
        busy[q] = busy.get(q, 0) + (e - s)
        d = (e - s) / 1e3
        if d < args.min_us and gap < args.min_us:
            small[q] = small.get(q, 0) + d
            continue
        short = name.replace('void ', '').replace('ptmi::', '')[:70]
        print(f'q{q:<3} t={(s - t0) / 1e3:9.1f} us  dur={d:8.1f}  gap={gap:7.1f}  grid={grid:<9d} {short}')
    for q in busy:
        print(f'# queue {q}: busy {busy[q] / 1e6:.3f} ms (of which kernels < {args.min_us} us not listed: {small.get(q, 0) / 1e3:.3f} ms)')


if __name__ == '__main__':
    main()
