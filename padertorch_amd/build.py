"""Build libptmi.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m padertorch_amd.build            # rebuild if sources are newer
    python -m padertorch_amd.build --force

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
``padertorch_amd/libptmi.so`` travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / 'csrc'
LIB = PKG / 'libptmi.so'
#: test hooks (csrc/testhooks/*.hip: kernels only tests and bench.py's measurement brackets launch) - a library of their own, so that
#: libptmi.so exports exactly the hot path's C ABI (include/ptmi.h)
HOOKS = PKG / 'libptmi_testhooks.so'
ARCH = 'gfx950'


def sources():
    return sorted(CSRC.glob('*.hip'))


def _stale():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob('*.hip')) + list(CSRC.glob('*.h')) + [PKG.parent / 'include' / 'ptmi.h']
    return any(p.stat().st_mtime > t for p in deps)


def hipcc_path():
    return shutil.which('hipcc') or '/opt/rocm/bin/hipcc'


def build(force=False, verbose=False):
    if not force and not _stale():
        build_hooks()
        return LIB
    objdir = PKG / 'csrc' / '_obj'
    objdir.mkdir(exist_ok=True)
    flags = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-fno-gpu-rdc',
             '-Wall', '-Wno-unused-function']
    procs = []
    objs = []
    for src in sources():
        obj = objdir / (src.stem + '.o')
        objs.append(obj)
        hdr_t = max(p.stat().st_mtime for p in list(CSRC.glob('*.h')) + [PKG.parent / 'include' / 'ptmi.h'])
        if not force and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_t):
            continue
        cmd = [hipcc_path(), *flags, '-c', str(src), '-o', str(obj)]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'--- hipcc failed on {src.name}\n{out}\n')
        elif out.strip() and verbose:
            print(out)
    if failed:
        raise RuntimeError('hipcc failed')
    tmp = LIB.with_suffix('.so.tmp')
    cmd = [hipcc_path(), '-shared', '-fPIC', f'--offload-arch={ARCH}', *map(str, objs), '-o', str(tmp)]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    build_hooks(force=True, verbose=verbose)
    return LIB


def build_hooks(force=False, verbose=False):
    srcs = sorted((CSRC / 'testhooks').glob('*.hip'))
    if not force and HOOKS.exists() and all(p.stat().st_mtime <= HOOKS.stat().st_mtime for p in srcs):
        return HOOKS
    tmp = HOOKS.with_suffix('.so.tmp')
    cmd = [hipcc_path(), '-O3', '-std=c++17', '-shared', '-fPIC', f'--offload-arch={ARCH}', *map(str, srcs), '-o', str(tmp)]
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(tmp, HOOKS)
    return HOOKS


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
