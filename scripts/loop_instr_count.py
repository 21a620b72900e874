#!/usr/bin/env python3
"""Static instruction mix of a kernel's outermost loop from the device assembly (no GPU needed):
    python scripts/loop_instr_count.py <file.s> <mangled-name fragment> ...
A wave64 VALU instruction occupies its SIMD for 4 cycles (16 lanes per cycle), a quarter-rate one (exp, rcp, rsq, sqrt, log) for 16."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read().split('\n')
for frag in sys.argv[2:]:
    a = next(i for i, l in enumerate(s) if l.startswith('_ZN4ptmi') and frag in l.split(':')[0])
    e = next(i for i in range(a, len(s)) if s[i].startswith('.Lfunc_end'))
    k = s[a:e]
    labels = {}
    for i, l in enumerate(k):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(k):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = i - labels[m.group(1)]
            if best is None or span > best[0]:
                best = (span, labels[m.group(1)], i)
    _, a0, b0 = best
    body = [l.strip() for l in k[a0:b0 + 1] if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter()
    quarter = ('v_exp_f32', 'v_rcp_f32', 'v_log_f32', 'v_rsq_f32', 'v_sqrt_f32', 'v_rcp_iflag_f32')
    for l in body:
        op = l.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op in quarter:
            c['valu quarter rate'] += 1
        elif op.startswith('v_'):
            c['valu'] += 1
        elif op.startswith('s_'):
            c['salu / smem / branch'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_')):
            c['vmem'] += 1
        else:
            c['other'] += 1
    cyc = 4 * c['valu'] + 16 * c['valu quarter rate']
    print(f'{frag[:60]}: outermost loop {len(body)} instructions (static, inner poll loops counted once) {dict(c)}')
    print(f'    VALU issue cycles per wavefront and step (static upper bound: both sides of wave-uniform branches counted): {cyc}')
