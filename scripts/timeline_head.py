#!/usr/bin/env python3
"""Head of a captured step from a rocprofv3 rocpd sqlite taken with ``--kernel-trace --memory-copy-trace``: kernels AND memory copies
between the last kernel of one step (adam_flat) and the first recurrence launch of the next.

    python scripts/timeline_head.py <results.db>
"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view')")]
    ks = cur.execute('select name, start, end from kernels order by start').fetchall()
    copies = []
    for cand in ('memory_copies', 'memory_copy'):
        if cand in names:
            cols = [c[1] for c in cur.execute(f'pragma table_info({cand})')]
            print('#', cand, cols)
            sel = [c for c in ('name', 'start', 'end', 'size', 'src_agent_abs_index', 'dst_agent_abs_index') if c in cols]
            copies = cur.execute(f'select {", ".join(sel)} from {cand} order by start').fetchall()
            copies = [dict(zip(sel, r)) for r in copies]
            break
    else:
        print('# no memory-copy table among', [n for n in names if 'mem' in n.lower()])
    adam = [i for i, k in enumerate(ks) if 'adam_flat' in k[0]]
    if len(adam) < 4:
        print('no steps found')
        return
    a = adam[len(adam) // 2]
    t0 = ks[a][2]
    ev = []
    for name, s, e in ks[a:a + 40]:
        ev.append((s, e, 'K', name[:90]))
    t1 = max(e for s, e, _, _ in ev)
    for c in copies:
        if t0 - 200e3 <= c['start'] <= t1:
            ev.append((c['start'], c['end'], 'C', f"{c.get('name', '')} {c.get('size', '')} B  {c.get('src_agent_abs_index', '')}->{c.get('dst_agent_abs_index', '')}"))
    ev.sort()
    print('# us relative to the END of adam_flat of the previous step')
    for s, e, kind, what in ev:
        print(f'{kind} t={(s - t0) / 1e3:9.1f}  dur={(e - s) / 1e3:8.1f}  {what}')
        if 'lstm_fwd' in what:
            break


if __name__ == '__main__':
    main()
