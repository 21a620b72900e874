"""A/B tooling, NOT part of the package (moved out of padertorch_amd/ in round 4): GEMM solution selection for the round-1 route of
the dense layers through the BLAS library (``bench.py --library-gemms``, ``padertorch_amd.ops.gemm.ENABLED = False``).

The input-projection / linear / weight-gradient GEMMs run in rocBLAS / hipBLASLt through torch
(DESIGN.md section 3.5).  Their default heuristics pick fp32 kernels that reach 65-70 % of the fp32 MFMA
peak at the PIT shapes; PyTorch's TunableOp finds solutions at 85-91 % (one-off search of ~40 s).
This module applies a committed result file for the benchmark shapes and offers the online search
for other shapes.  Nothing here changes what is computed - only which library kernel computes it.

    from scripts import library_gemm_tuning as tuning
    tuning.use_tuned_gemms()                 # committed selections for the PIT shapes (ignored, with
                                             # a warning from TunableOp, when the library versions differ)
    tuning.use_tuned_gemms(search=True)      # additionally search new shapes on first use
"""
import os
import shutil
import tempfile
from pathlib import Path

import torch

TUNED_DIR = Path(__file__).resolve().parent / 'tuned'
DEFAULT_FILE = TUNED_DIR / 'gfx950_pit_dc.csv'   # PIT (B = 4 / 32 / 64) and DC (B = 32 / 64) shapes, scripts/tune_gemms.py


def use_tuned_gemms(results_file=None, search=False, device=None):
    """Enable TunableOp with the selections of ``results_file`` (default: the committed file).

    Every process works on its own copy (TunableOp rewrites the file when ``search`` adds entries),
    so concurrent ranks never touch the same file.  Returns the path in use.
    """
    if not torch.cuda.is_available():
        raise RuntimeError('use_tuned_gemms needs the GPU (there is no CPU path)')
    import torch.cuda.tunable as tunable
    src = Path(results_file) if results_file is not None else DEFAULT_FILE
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    work = Path(tempfile.mkdtemp(prefix='ptmi_tunable_')) / f'results_dev{dev}_pid{os.getpid()}.csv'
    if src.exists():
        shutil.copyfile(src, work)
    tunable.enable(True)
    tunable.tuning_enable(bool(search))
    tunable.set_filename(str(work), insert_device_ordinal=False)
    if src.exists():
        tunable.read_file(str(work))
    return work


def disable():
    import torch.cuda.tunable as tunable
    tunable.enable(False)
