import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = Path(__file__).resolve().parent / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun)')
    config.addinivalue_line('markers', 'library_path: the test drives a torch / library path on purpose (A/B against the HIP kernels)')


@pytest.fixture(autouse=True)
def _native_path_only(request):
    """GPU tests run STRICT: a model or op that would leave the hand-written HIP path (an LSTM the recurrence kernels do not
    cover, a dense layer on the BLAS library, ...) raises instead of falling back (padertorch_amd._lib.leaving_native_path) -
    unless the test says it compares against such a path on purpose (``@pytest.mark.library_path``)."""
    if 'gpu' not in request.keywords or 'library_path' in request.keywords:
        yield
        return
    from padertorch_amd import _lib
    before = _lib.STRICT
    _lib.STRICT = True
    try:
        yield
    finally:
        _lib.STRICT = before


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU: they are skipped there only when
    the run did not ask for them (``-m gpu`` on a GPU-less box fails loudly in the test body)."""
    import torch
    if torch.cuda.is_available():
        return
    selected = config.getoption('-m') or ''
    if 'gpu' in selected and 'not gpu' not in selected:
        return
    skip = pytest.mark.skip(reason='no GPU in this container (use gpurun)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def g1():
    return json.loads((GOLDEN / 'g1_known_answers.json').read_text())


def _npz(name):
    return dict(np.load(GOLDEN / name, allow_pickle=False))


@pytest.fixture(scope='session')
def g2():
    d = _npz('g2_stft.npz')
    d['cases'] = json.loads(str(d['cases']))
    return d


@pytest.fixture(scope='session')
def g3():
    return _npz('g3_features.npz')


@pytest.fixture(scope='session')
def g4():
    d = _npz('g4_pit.npz')
    d['names'] = json.loads(str(d['names']))
    return d


@pytest.fixture(scope='session')
def g5():
    return _npz('g5_dc.npz')


@pytest.fixture(scope='session')
def g6():
    d = _npz('g6_models.npz')
    d['train_example_indices'] = json.loads(str(d['train_example_indices']))
    return d


@pytest.fixture(scope='session')
def g7():
    d = _npz('g7_td_losses.npz')
    d['cases'] = json.loads(str(d['cases']))
    d['names'] = json.loads(str(d['names']))
    return d


@pytest.fixture(scope='session')
def g8():
    d = _npz('g8_logmel.npz')
    d['configs'] = json.loads(str(d['configs']))
    return d


@pytest.fixture(scope='session')
def g9():
    d = _npz('g9_norm.npz')
    d['cases'] = json.loads(str(d['cases']))
    d['modules'] = json.loads(str(d['modules']))
    return d
