#!/usr/bin/env python3
"""profiles/<tag>_pmc_fetch_size.txt + <tag>_pmc_write_size.txt -> profiles/pmc_traffic.json.

HBM bytes per launch = 2 x FETCH_SIZE (the gfx950 correction of MI355X_MICROARCH.md; calibrated here
on pit_pairwise: 58.26 MB algorithmic vs 2 x FETCH = 58.7 MB) + WRITE_SIZE, both in KiB per dispatch.
bench.py reads the result as `roofline.traffic`.
"""
import json
import re
import sys
from pathlib import Path

prof = Path(__file__).resolve().parent.parent / 'profiles'
tag = sys.argv[1] if len(sys.argv) > 1 else 'r3'
# kernel name fragment -> key = the timer name bench.py files the kernel family under (_lib.timed)
KEYS = {'pit_features_kernel': 'pit_features', 'pit_pairwise_kernel': 'pit_pairwise_sse',
        'pit_backward_kernel': 'pit_backward', 'lstm_fwd_split_kernel': 'lstm_forward', 'lstm_fwd_daf_kernel': 'lstm_forward',
        'lstm_bwd_split_kernel': 'lstm_backward', 'lstm_fwd_persistent_kernel': 'lstm_forward',
        'lstm_bwd_persistent_kernel': 'lstm_backward', 'gemm_planes_big_kernel': 'gemm_planes',
        'gemm_planes_kernel': 'gemm_planes', 'stft_fwd_kernel': 'stft_fwd',
        'istft_kernel': 'istft'}


def parse(path, counter):
    tot, cnt, cur, cur_line, dispatches = {}, {}, None, None, {}
    for line in path.read_text().splitlines():
        if not line.startswith(' '):
            cur = next((v for k, v in KEYS.items() if k in line), None)
            cur_line = line
            m = re.search(r'dispatches: (\d+)', line)
            dispatches[cur_line] = int(m.group(1)) if m else 1
        elif cur and counter in line:
            # several kernels may map to one family (template instances of the GEMM): dispatch-weighted mean
            n = dispatches.get(cur_line, 1)
            tot[cur] = tot.get(cur, 0.) + float(line.split()[-1]) * n
            cnt[cur] = cnt.get(cur, 0) + n
    return {k: tot[k] / cnt[k] for k in tot}


fetch = parse(prof / f'{tag}_pmc_fetch_size.txt', 'FETCH_SIZE')
write = parse(prof / f'{tag}_pmc_write_size.txt', 'WRITE_SIZE')
note = ('2 x FETCH_SIZE (gfx950 correction; calibrated here on pit_pairwise: 58.26 MB algorithmic vs '
        '2 x FETCH = 58.7 MB) + WRITE_SIZE; KB = 1024 B')
res = {k: dict(fetch_size_kb=fetch[k], write_size_kb=write.get(k, 0.0),
               hbm_bytes_per_launch=int((2 * fetch[k] + write.get(k, 0.0)) * 1024), note=note)
       for k in fetch}
(prof / 'pmc_traffic.json').write_text(json.dumps(res, indent=1))
print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in res.items()}))
