#!/bin/bash
# full kernel names (+ call counts) of a command: usage kernel_names.sh <tag> -- <cmd...>
tag=$1; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kn_$tag
timeout 300 rocprofv3 --kernel-trace -d /tmp/kn_$tag -o p -- "$@" > /tmp/kn_$tag.log 2>&1 </dev/null
db=$(find /tmp/kn_$tag -name "*.db" 2>/dev/null | head -1)
mkdir -p /root/repo/gpurun_out/p
if [ -n "$db" ]; then python - "$db" > /root/repo/gpurun_out/p/kn_$tag.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for name, n, tot in cur.execute('select name, count(*), sum(duration) from kernels group by name order by sum(duration) desc'):
    if 'Cijk' in name or 'ptmi' in name:
        print(n, round(tot / 1e6, 2), name)
PY
else tail -3 /tmp/kn_$tag.log > /root/repo/gpurun_out/p/kn_$tag.txt; fi
grep -c . /root/repo/gpurun_out/p/kn_$tag.txt
