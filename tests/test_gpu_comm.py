"""``ptmi_comm_*`` / ``ptmi_allreduce_sum``: the C ABI's data-parallel gradient exchange (RCCL; reference
``padertorch/train/trainer.py:396-442``, gradients SUMMED over the replicas at ``:426-428``).  A 1-GPU box can only form a
world of one rank (RCCL refuses two ranks on one device): the entry points, the rendezvous id, in-place semantics ("sum over one
rank = identity, no division by the world size"), stream ordering and the error paths are what is checked here; the N > 1
behaviour is the ``torch.distributed`` path's (same RCCL) and the driver's SCALE run."""
import ctypes

import numpy as np
import pytest
import torch

from padertorch_amd import _lib

pytestmark = pytest.mark.gpu


def test_allreduce_sum_world_of_one():
    lib = _lib.load()
    assert int(lib.ptmi_comm_rccl_version()) > 20000          # librccl was found and answers
    ident = (ctypes.c_uint8 * 128)()
    _lib.check(lib.ptmi_comm_unique_id(ident), 'ptmi_comm_unique_id')
    assert any(ident)
    comm = ctypes.c_void_p()
    torch.cuda.set_device(0)
    _lib.check(lib.ptmi_comm_create(ctypes.byref(comm), 1, 0, ident), 'ptmi_comm_create')
    try:
        g = torch.Generator(device='cuda:0').manual_seed(3)
        flat = torch.randn(23_500_000, device='cuda:0', generator=g)          # the PIT model's flat gradient bucket size
        want = flat.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            flat.mul_(2.)                                                      # ordered in front of the collective on its stream
            _lib.check(lib.ptmi_allreduce_sum(comm, flat.data_ptr(), flat.numel(), _lib.stream(flat.device)), 'ptmi_allreduce_sum')
            flat.mul_(.5)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(flat, want)                                         # a sum over one rank, not a mean: bit-identical
        _lib.check(lib.ptmi_allreduce_sum(comm, None, 0, None), 'empty')
        assert lib.ptmi_allreduce_sum(comm, None, 4, None) != 0                # invalid arguments are refused
        assert lib.ptmi_allreduce_sum(None, flat.data_ptr(), 4, None) != 0
    finally:
        _lib.check(lib.ptmi_comm_destroy(comm), 'ptmi_comm_destroy')
    bad = ctypes.c_void_p()
    assert lib.ptmi_comm_create(ctypes.byref(bad), 2, 2, ident) != 0           # rank outside the world
    np.testing.assert_equal(bad.value, None)
