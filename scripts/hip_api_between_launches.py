#!/usr/bin/env python3
"""HIP runtime calls between consecutive kernel launches of one training step (rocprofv3 --kernel-trace --hip-runtime-trace ... -> rocpd
database): every hipEventRecord / hipStreamWaitEvent is a marker or barrier packet of ~4 us in the queue it goes to, and torch's caching
allocator issues one hipEventRecord per recorded stream when a tensor marked with record_stream() is freed.  Lists the launches that are
preceded by more than `--min` such calls (usage: python scripts/hip_api_between_launches.py <db> [--min 3])."""
import sqlite3
import sys
from collections import Counter

db = sys.argv[1]
minimum = int(sys.argv[sys.argv.index('--min') + 1]) if '--min' in sys.argv else 3
con = sqlite3.connect(db)
feat = list(con.execute("select stack_id, start from kernels where name like '%pit_features_kernel%' order by start"))
assert len(feat) >= 3, 'need a trace of at least three steps'
(s0, _), (s1, _) = feat[-2], feat[-1]
r0 = list(con.execute('select start, tid from regions where stack_id=?', (s0,)))[0]
r1 = list(con.execute('select start, tid from regions where stack_id=?', (s1,)))[0]
tid = r0[1]
rows = list(con.execute('select name, start, stack_id from regions where tid=? and start>=? and start<? order by start', (tid, r0[0], r1[0])))
kern = {sid: name for sid, name in con.execute('select stack_id, name from kernels')}
quiet = ('hipGetDevice', 'hipSetDevice', 'hipGetLastError', 'hipDeviceGetStreamPriorityRange', 'hipStreamIsCapturing', 'hipStreamGetCaptureInfo',
         'hipPeekAtLastError')
acc, total = Counter(), Counter()
for name, start, sid in rows:
    if name in quiet:
        continue
    if 'LaunchKernel' in name:
        n = acc['hipEventRecord'] + acc['hipStreamWaitEvent']
        if n >= minimum:
            k = kern.get(sid, '?').replace('void ', '').replace('ptmi::', '')[:60]
            print(f'{(start - r0[0]) / 1e3:9.1f} us (host)  {dict(acc)}  before  {k}')
        acc = Counter()
    else:
        acc[name] += 1
        total[name] += 1
print('per step:', dict(total))
