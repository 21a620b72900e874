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
tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
KEYS = {'pit_features_kernel': 'pit_features', 'pit_pairwise_kernel': 'pit_pairwise_sse',
        'pit_backward_kernel': 'pit_backward', 'lstm_fwd_persistent_kernel': 'lstm_fwd_persistent',
        'lstm_bwd_persistent_kernel': 'lstm_bwd_persistent', 'stft_fwd_kernel': 'stft_fwd',
        'istft_kernel': 'istft'}


def parse(path, counter):
    out, cur = {}, None
    for line in path.read_text().splitlines():
        if not line.startswith(' '):
            cur = next((v for k, v in KEYS.items() if k in line), None)
        elif cur and counter in line:
            out[cur] = float(line.split()[-1])
    return out


fetch = parse(prof / f'{tag}_pmc_fetch_size.txt', 'FETCH_SIZE')
write = parse(prof / f'{tag}_pmc_write_size.txt', 'WRITE_SIZE')
note = ('2 x FETCH_SIZE (gfx950 correction; calibrated here on pit_pairwise: 58.26 MB algorithmic vs '
        '2 x FETCH = 58.7 MB) + WRITE_SIZE; KB = 1024 B')
res = {k: dict(fetch_size_kb=fetch[k], write_size_kb=write.get(k, 0.0),
               hbm_bytes_per_launch=int((2 * fetch[k] + write.get(k, 0.0)) * 1024), note=note)
       for k in fetch}
(prof / 'pmc_traffic.json').write_text(json.dumps(res, indent=1))
print(json.dumps({k: v['hbm_bytes_per_launch'] for k, v in res.items()}))
