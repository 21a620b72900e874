#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (rocpd sqlite): per kernel name, mean counter value per dispatch."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else 'ptmi'
cols = [d[1] for d in cur.execute('pragma table_info(counters_collection)')]
rows = cur.execute('select * from counters_collection').fetchall()
ik, ic, iv = cols.index('kernel_name'), cols.index('counter_name'), cols.index('value')
idp = cols.index('dispatch_id')
acc = defaultdict(lambda: defaultdict(float))
nd = defaultdict(set)
for r in rows:
    if flt not in r[ik]:
        continue
    acc[r[ik]][r[ic]] += r[iv]
    nd[r[ik]].add(r[idp])
for k, d in acc.items():
    n = len(nd[k])
    print(k[:90], 'dispatches', n)
    for c, v in sorted(d.items()):
        print(f'    {c:28s} {v / n:18.1f}')
