#!/usr/bin/env python3
"""Compile every csrc/*.hip to gfx950 assembly and flag kernels in which the compiler has serialised loads:
runs of `load; s_waitcnt vmcnt(<=1)` (each load waited for before the next one is issued).  A branch
around a load, or a conditional assignment of a loop-carried register, is what usually causes it
(merged wait counts at control-flow joins); seen in the LSTM backward, the deep-clustering and the
unit-norm kernels, where it cost 2-3x.

    python scripts/scan_serialized_loads.py [min_run=4]
"""
import re
import subprocess
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent
min_run = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for src in sorted((root / 'padertorch_amd' / 'csrc').glob('*.hip')):
    asm = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '-o', '-', str(src),
                          '--cuda-device-only', f'-I{root / "include"}'], capture_output=True, text=True).stdout
    name, toks = None, {}
    for line in asm.splitlines():
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name = m.group(1)
            toks[name] = []
            continue
        if name is None:
            continue
        t = line.strip()
        if re.match(r'(global_load|buffer_load|flat_load)', t):
            toks[name].append('L')
        elif t.startswith('s_waitcnt'):
            w = re.search(r'vmcnt\((\d+)\)', t)
            if w and int(w.group(1)) <= 1:
                toks[name].append('W')
        elif t.startswith('s_endpgm'):
            name = None
    for k, v in toks.items():
        runs = [len(r) // 2 for r in re.findall(r'(?:LW){%d,}' % min_run, ''.join(v))]
        if runs:
            try:
                demangled = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip() or k
            except OSError:
                demangled = k
            print(f'{src.name}: {demangled[:100]}: serialised runs {runs}')
