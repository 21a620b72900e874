#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite (``--kernel-trace --stats`` or ``--pmc``) into a small text summary.

    python scripts/profile_summary.py <results.db> [--pmc] [--top N]
"""
import argparse
import sqlite3
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--pmc', action='store_true')
    ap.add_argument('--top', type=int, default=25)
    ap.add_argument('--filter', default='')
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    if not args.pmc:
        rows = cur.execute(
            'select name, count(*), sum(duration), avg(duration), min(duration), max(duration), '
            'max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(scratch_size) '
            'from kernels group by name order by sum(duration) desc').fetchall()
        total = sum(r[2] for r in rows)
        print(f'# kernel trace: {sum(r[1] for r in rows)} dispatches, {total / 1e6:.3f} ms GPU kernel time')
        print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>9} {"min_us":>9} {"max_us":>9} {"%":>6} '
              f'{"grid":>9} {"wg":>5} {"lds":>6} {"vgpr":>5} {"scr":>4}  name')
        for r in rows[:args.top]:
            if args.filter and args.filter not in r[0]:
                continue
            print(f'{r[1]:7d} {r[2] / 1e6:10.3f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {r[5] / 1e3:9.2f} '
                  f'{100 * r[2] / total:6.2f} {r[6]:9d} {r[7]:5d} {r[8]:6d} {r[9]:5d} {r[10]:4d}  {r[0][:110]}')
        return
    cols = [d[1] for d in cur.execute('pragma table_info(counters_collection)')]
    ik, ic, iv, idp = (cols.index(c) for c in ('kernel_name', 'counter_name', 'value', 'dispatch_id'))
    acc, nd = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for r in cur.execute('select * from counters_collection'):
        if args.filter and args.filter not in r[ik]:
            continue
        acc[r[ik]][r[ic]] += r[iv]
        nd[r[ik]].add(r[idp])
    for k, d in acc.items():
        print(f'{k[:110]}  (dispatches: {len(nd[k])}; mean per dispatch)')
        for c, v in sorted(d.items()):
            print(f'    {c:28s} {v / len(nd[k]):18.1f}')


if __name__ == '__main__':
    main()
