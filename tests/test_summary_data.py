"""Host-side helpers around the step: summary images (``padertorch/summary/tbx_utils.py:61-157,219-271``), ``Sorter``
(``data/batch.py:133-158``), ``collate_fn`` (``data/utils.py:21-69``), frame bookkeeping - against the golden g10
produced by the real reference and the literal answers of the reference's doctests."""
import dataclasses
import json
import warnings

import numpy as np
import pytest
import torch

from padertorch_amd.data import Sorter, collate_fn
from padertorch_amd.summary import mask_to_image, spectrogram_to_image, stft_to_image


@pytest.fixture(scope='module')
def g10():
    from conftest import GOLDEN
    d = dict(np.load(GOLDEN / 'g10_summary_data.npz', allow_pickle=False))
    d['cases'] = json.loads(str(d['cases']))
    d['structural'] = json.loads(str(d['structural']))
    return d


def test_images_match_reference_over_all_options(g10):
    mask, spec = g10['mask'], g10['spec']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for c in g10['cases']:
            bf, color, origin, key = c['batch_first'], c['color'], c['origin'], c['key']
            m = mask.transpose(1, 0, 2) if bf else mask
            z = spec.transpose(1, 0, 2) if bf else spec
            got = {
                'mask3': mask_to_image(torch.from_numpy(np.ascontiguousarray(m)), bf, color, origin),
                'mask2': mask_to_image(mask[:, 1], bf, color, origin),
                'stft3': stft_to_image(torch.from_numpy(np.ascontiguousarray(z)), bf, color, origin),
                'stft2_60': stft_to_image(spec[:, 2], bf, color, origin, 60),
                'abs3': stft_to_image(np.abs(z), bf, color, origin),
                'pow_lin': spectrogram_to_image(np.abs(z) ** 2, bf, color, origin, log=False),
            }
            for name, img in got.items():
                ref = g10[f'{key}/{name}']
                assert img.shape == ref.shape and img.dtype == ref.dtype, (key, name, img.shape, ref.shape, img.dtype)
                np.testing.assert_array_equal(img, ref, err_msg=f'{key}/{name}')


def test_image_doctest_answers_and_signature():
    data = np.array([1, 0.004, 0.003, 0.001_05, 0.001])[:, None]           # tbx_utils.py:140-146
    np.testing.assert_array_equal(np.squeeze(stft_to_image(data, color=None)), [255, 10, 0, 0, 0])
    np.testing.assert_array_equal(np.squeeze(stft_to_image(data, color=None, visible_dB=60)), [255, 51, 40, 1, 0])
    x = torch.rand(6, 2, 4)                                                  # (frames, batch, features)
    assert mask_to_image(x).shape == (1, 4, 6)                               # grayscale by default
    assert mask_to_image(x, True).shape == (1, 4, 2)                         # second positional argument = batch_first: x[0]
    assert stft_to_image(x).shape == (4, 4, 6)                               # viridis RGBA by default
    with pytest.raises(ValueError):
        mask_to_image(x, None)
    with pytest.warns(UserWarning):
        mask_to_image(x * 3)


def test_mask_estimator_images_only_with_snapshot():
    """The review renders images (device -> host copies) only when create_snapshot is set; the images are the FIRST
    example, features on the y axis (reference add_images calls the helpers with batch_first=True)."""
    from padertorch_amd.contrib.examples.speech_enhancement.mask_estimator.model import SimpleMaskEstimator
    out = dict(speech_mask_prediction=torch.rand(3, 11, 17), noise_mask_prediction=torch.rand(3, 11, 17))
    batch = dict(observation_abs=torch.rand(3, 11, 17))
    images = SimpleMaskEstimator.add_images(batch, out)
    assert set(images) == {'speech_mask', 'observed_stft', 'noise_mask'}
    assert images['speech_mask'].shape == (1, 17, 11) and images['observed_stft'].shape == (4, 17, 11)
    np.testing.assert_array_equal(images['speech_mask'], mask_to_image(out['speech_mask_prediction'][0]))


def test_sorter_and_collate_match_reference(g10):
    st = g10['structural']
    batch = [{'value': x, 'num_samples': n} for x, n in [(5, 10), (1, 30), (3, 20), (2, 20)]]
    assert Sorter('value')([{'value': x} for x in [5, 1, 3, 2]]) == ({'value': 5}, {'value': 3}, {'value': 2}, {'value': 1})
    assert list(Sorter('value')(batch)) == st['sorter_value']
    assert list(Sorter()(batch)) == st['sorter_default']                       # by num_samples, descending, stable
    assert list(Sorter('value', reverse=False)(batch)) == st['sorter_ascending']
    assert isinstance(Sorter()(batch), tuple)
    assert collate_fn([{'a': 1}, {'a': 2}]) == st['collate_flat'] == {'a': [1, 2]}
    assert collate_fn(({'a': 1}, {'a': 2})) == {'a': (1, 2)} and st['collate_tuple_is_tuple']
    assert collate_fn([{'a': {'b': [1, 2]}}, {'a': {'b': [3, 4]}}]) == st['collate_nested']
    Point = dataclasses.make_dataclass('Point', ['x', 'y'])                    # data/utils.py:42-49
    assert collate_fn([Point(1, 2), Point(3, 4)]) == Point([1, 3], [2, 4])
    assert collate_fn((Point(1, 2), Point(3, 4))) == Point((1, 3), (2, 4))
    with pytest.raises(AssertionError):
        collate_fn([{'a': 1}, {'b': 2}])


def test_sample_index_to_frame_index():
    from padertorch_amd.ops import STFT
    st = STFT(8, 1, fading=None)
    assert [st.sample_index_to_frame_index(i) for i in range(12)] == [0, 0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7]
    st = STFT(512, 128, fading='full')
    f = st.sample_index_to_frame_index(np.array([0, 255, 256, 1000]))
    np.testing.assert_array_equal(f, np.array([0, 0, 0, 5]) + 3)
    # consistent with the frame count: the last sample of a signal maps to an existing frame
    for n in (600, 4000, 32000):
        assert st.sample_index_to_frame_index(n - 1) < st.samples_to_frames(n)


def test_device_prefetcher_yields_what_example_to_device_would():
    """``data.DevicePrefetcher`` (CPU: no copy stream, same values, same order, nested structure kept, nothing dropped)."""
    import numpy as np
    import torch
    from padertorch_amd.data import DevicePrefetcher, example_to_device
    batches = [dict(y=np.full((2, 3), i, dtype=np.float32), meta=dict(n=[3, 3], t=torch.tensor([i]))) for i in range(4)]
    got = list(DevicePrefetcher(batches, 'cpu'))
    assert len(got) == 4 and len(DevicePrefetcher(batches, 'cpu')) == 4
    for i, g in enumerate(got):
        want = example_to_device(batches[i], torch.device('cpu'))
        assert torch.equal(g['y'], want['y']) and g['meta']['n'] == [3, 3] and int(g['meta']['t']) == i
    assert list(DevicePrefetcher([], 'cpu')) == []


def test_row_slot_batches_groups_and_sorts():
    """data.row_slot_batches: round(fill * row_slots) examples per batch, sorted by descending length, collated; the tail batch
    takes what is left; a stream (generator) works."""
    from padertorch_amd.data import row_slot_batches
    exs = [dict(num_samples=n, tag=i) for i, n in enumerate((5, 9, 2, 7, 3, 8, 1))]
    got = list(row_slot_batches(iter(exs), row_slots=2, fill=1.5))
    assert [b['num_samples'] for b in got] == [[9, 5, 2], [8, 7, 3], [1]]
    assert got[0]['tag'] == [1, 0, 2]
    raw = list(row_slot_batches(exs, row_slots=4, fill=1.0, key=lambda e: -e['tag'], collate=False))
    assert [[e['tag'] for e in b] for b in raw] == [[0, 1, 2, 3], [4, 5, 6]]


def test_device_prefetcher_release_mode_follows_what_to_device_does():
    """A ``to_device`` that computes (features on the copy stream) gets ``release='mark'`` by default and is refused with an explicit
    ``'record_stream'``: marks do not reach what such a function allocates or the buffers it writes; the plain mover keeps
    ``'record_stream'``.  The tensors a ``PaddedList`` carries beside its entries are found."""
    import pytest
    import torch
    from padertorch_amd.data import DevicePrefetcher, example_to_device
    from padertorch_amd.data.prefetch import _tensors
    from padertorch_amd.ops.sequence.pack_module import PaddedList
    assert DevicePrefetcher([], 'cpu').release == 'record_stream'
    assert DevicePrefetcher([], 'cpu', to_device=example_to_device).release == 'record_stream'
    custom = lambda ex, dev: example_to_device(ex, dev)      # noqa: E731
    assert DevicePrefetcher([], 'cpu', to_device=custom).release == 'mark'
    with pytest.raises(ValueError, match='mark'):
        DevicePrefetcher([], 'cpu', to_device=custom, release='record_stream')
    pad = torch.zeros(2, 5, 3)
    pl = PaddedList(pad, [5, 4], True, torch.tensor([5, 4], dtype=torch.int32))

    class Rec:
        data = torch.ones(9, 3)
    pl.packed_log1p = Rec()
    found = list(_tensors(dict(Y_abs=pl, n=[5, 4])))
    assert any(t is pad for t in found) and any(t is pl.lengths_dev for t in found) and any(t is Rec.data for t in found)


def test_static_slot_batcher_makes_device_data_examples_or_hands_the_batch_back():
    """``data.StaticSlotBatcher`` (host side; the kernels that read its tables: tests/test_gpu_ragged_graph.py): a batch that fits becomes
    padded waveforms + sample counts + a ``StaticSlots`` layout whose tables describe exactly that batch; one that does not fit (too many
    frames for the grid, another number of examples) comes back untouched and is counted."""
    import numpy as np
    import torch
    from padertorch_amd.data import StaticSlotBatcher, row_slot_batches
    from padertorch_amd.ops.sequence import SlotLayout, StaticSlots
    rng = np.random.RandomState(0)
    stream = [dict(y=rng.randn(n).astype(np.float32), s=rng.randn(2, n).astype(np.float32), num_samples=int(n), example_id=f'u{i}')
              for i, n in enumerate(rng.randint(2400, 4801, 20))]
    batcher = StaticSlotBatcher(examples=8, slots=4, max_samples=4800, device='cpu', steps=80)
    outs = [batcher(b) for b in row_slot_batches(stream, row_slots=4, fill=2.0)]
    assert [isinstance(o.get('slots'), StaticSlots) for o in outs] == [True, True, False] and batcher.refused == 1
    assert outs[2]['num_samples'] == sorted(outs[2]['num_samples'], reverse=True) and len(outs[2]['y']) == 4      # handed back as it came
    for o in outs[:2]:
        ns = o['num_samples'].tolist()
        assert ns == sorted(ns, reverse=True) and o['y'].shape == (8, 4800) and o['s'].shape == (8, 2, 4800)
        for i, n in enumerate(ns):
            assert float(o['y'][i, n:].abs().sum()) == 0. and float(o['s'][i, :, n:].abs().sum()) == 0. and float(o['y'][i, :n].abs().sum()) > 0
        st, frames = o['slots'], batcher.frames_of(ns)
        assert st.frames.tolist() == frames and len(o['example_id']) == 8
        lay = SlotLayout(frames, 4)
        rows = st.grid_of_flat.view(8, st.padded_time)
        for b, t in enumerate(frames):                       # frame t of example b sits where the host-side layout puts it; padding -> the zero row
            assert rows[b, :t].tolist() == [(lay.t0[b] + k) * 4 + lay.slot[b] for k in range(t)]
            assert (rows[b, t:] == st.steps * 4).all()
    assert outs[0]['slots'] is not outs[1]['slots']          # two layouts in turn
    tight = StaticSlotBatcher(examples=8, slots=4, max_samples=4800, device='cpu', steps=41)
    assert not isinstance(tight(next(row_slot_batches(stream, row_slots=4, fill=2.0))).get('slots'), StaticSlots) and tight.refused == 1
