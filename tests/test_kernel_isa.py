"""Static checks on the device assembly of the recurrence kernels (hipcc cross-compiles without a GPU).

A FLAT access (a load through a pointer whose address space the compiler cannot see, e.g. `cond ? kernel_argument : table[t]`)
inside the time loop of the persistent LSTM kernels makes every later wait `s_waitcnt vmcnt(0)`: the operand passes then wait for
stores and cold loads they do not depend on ("the compiler's wait counts").  This test keeps FLAT accesses
out of the data-as-flag kernels and of the optimizer kernel."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def _assembly(stem, tmp_path):
    out = tmp_path / f'{stem}.s'
    cmd = [HIPCC, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-gpu-rdc', '--cuda-device-only', '-S', '-o', str(out),
           f'-I{ROOT / "include"}', str(ROOT / 'padertorch_amd' / 'csrc' / f'{stem}.hip')]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out.read_text()


def _kernels(asm):
    """{mangled name: body} of every kernel function in the listing."""
    bodies, name, lines = {}, None, []
    for line in asm.splitlines():
        m = re.match(r'^(_ZN4ptmi\w+):', line)
        if m:
            name, lines = m.group(1), []
        elif name is not None:
            if line.startswith('.Lfunc_end'):
                bodies[name] = '\n'.join(lines)
                name = None
            else:
                lines.append(line)
    return bodies


@pytest.mark.skipif(not Path(HIPCC).exists(), reason='hipcc not found')
@pytest.mark.parametrize('stem, patterns', [
    ('lstm_split', ('lstm_fwd_daf_kernel', 'lstm_bwd_split_kernel')),
    ('optim', ('adam_flat_kernel',)),
])
def test_no_flat_accesses_in_the_hot_kernels(stem, patterns, tmp_path):
    kernels = _kernels(_assembly(stem, tmp_path))
    checked = 0
    for name, body in kernels.items():
        if not any(p in name for p in patterns):
            continue
        # the data-as-flag instantiations of the backward kernel: template arguments ... UNI, DAF = true, TP, MSK
        if 'lstm_bwd_split_kernel' in name and not re.search(r'Lb[01]ELb1ELb[01]ELb[01]EEE', name):
            continue
        checked += 1
        flat = [l.strip() for l in body.splitlines() if re.match(r'^\s*flat_(load|store|atomic)', l)]
        assert not flat, f'{name}: FLAT accesses {flat[:3]}'
    assert checked >= len(patterns), (checked, sorted(kernels)[:10])
