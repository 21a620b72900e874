"""ops.sequence.pack_module: same results as the torch rnn utilities the reference wraps
(padertorch/ops/sequence/pack_module.py:14-34), incl. the copy-free paths for equal lengths."""
import numpy as np
import torch
from torch.nn.utils.rnn import pack_sequence as torch_pack, pad_packed_sequence


def test_unpack_equal_lengths_is_a_view_and_matches_torch():
    from padertorch_amd.ops.sequence import pack_module as pm
    xs = [torch.randn(7, 3, 5, requires_grad=True) for _ in range(4)]
    packed = torch_pack(xs)
    out = pm.unpack_sequence(packed)
    ref, lengths = pad_packed_sequence(packed)
    assert out.lengths == lengths.tolist() and not out.ragged and not out.batch_first
    assert out.padded.data_ptr() == packed.data.data_ptr()          # no copy
    np.testing.assert_array_equal(out.padded.detach().numpy(), ref.detach().numpy())
    for a, b in zip(out, xs):
        np.testing.assert_array_equal(a.detach().numpy(), b.detach().numpy())
    out.padded.square().sum().backward()                            # gradients flow through the view
    np.testing.assert_allclose(xs[1].grad.numpy(), 2 * xs[1].detach().numpy(), rtol=1e-6)
    # and back: packing the time-major PaddedList again is a view too
    again = pm.pack_sequence(out)
    assert again.data.data_ptr() == packed.data.data_ptr()
    np.testing.assert_array_equal(again.batch_sizes.numpy(), packed.batch_sizes.numpy())


def test_unpack_ragged_matches_torch():
    from padertorch_amd.ops.sequence import pack_module as pm
    xs = [torch.randn(l, 6) for l in (9, 7, 7, 2)]
    packed = torch_pack(xs)
    out = pm.unpack_sequence(packed)
    ref, lengths = pad_packed_sequence(packed)
    assert out.lengths == lengths.tolist() and out.ragged
    np.testing.assert_array_equal(out.padded.numpy(), ref.numpy())
    for a, b in zip(out, xs):
        np.testing.assert_array_equal(a.numpy(), b.numpy())
    again = pm.pack_sequence(out)
    np.testing.assert_array_equal(again.data.numpy(), packed.data.numpy())


def test_pad_direction_blocks_matches_the_hand_off_plane_layout():
    """``ops.gemm.pad_direction_blocks``: every direction's H input columns followed by zero columns up to the width of the
    forward recurrence's hand-off planes (the k layout the next projection's weights are packed with)."""
    import torch
    from padertorch_amd.ops.gemm import pad_direction_blocks
    w = torch.arange(3 * 2 * 5, dtype=torch.float32).view(3, 10)          # 3 outputs, 2 directions x H = 5
    p = pad_direction_blocks(w, 2, 5, 8)
    assert p.shape == (3, 16)
    assert torch.equal(p[:, 0:5], w[:, 0:5]) and torch.equal(p[:, 8:13], w[:, 5:10])
    assert float(p[:, 5:8].abs().sum()) == 0 and float(p[:, 13:16].abs().sum()) == 0
    x = torch.randint(-4, 5, (4, 10)).to(torch.float32)                  # integer-valued: both products are exact in fp32
    xp = pad_direction_blocks(x, 2, 5, 8)
    assert torch.equal(xp @ p.t(), x @ w.t())                             # padded operands: same product


def test_pack_meta_index_tables_match_the_per_step_definition():
    """``ops.lstm._PackMeta`` (vectorised): the predecessor row of every packed row per direction, the h0 variant, the rows of every
    sequence's first / last processed step and the packed -> padded row map equal their per-time-step definitions."""
    import random
    import numpy as np
    from padertorch_amd.ops.lstm import _PackMeta
    rnd = random.Random(7)
    for _ in range(25):
        B = rnd.randint(1, 9)
        lens = sorted((rnd.randint(1, 30) for _ in range(B)), reverse=True)
        T = lens[0]
        bs = [sum(1 for n in lens if n > t) for t in range(T)]
        m = _PackMeta(tuple(bs), torch.device('cpu'))
        offs = np.concatenate([[0], np.cumsum(bs)])
        rows = int(offs[-1])
        prev = np.full((2, rows), rows, dtype=np.int64)
        prev_h0 = prev.copy()
        for t in range(T):
            for b in range(bs[t]):
                r = offs[t] + b
                prev[0, r] = offs[t - 1] + b if t > 0 else rows
                prev[1, r] = offs[t + 1] + b if (t + 1 < T and b < bs[t + 1]) else rows
                for d in range(2):
                    prev_h0[d, r] = prev[d, r] if prev[d, r] != rows else rows + 1 + b
        assert np.array_equal(m.prev_dev.numpy(), prev) and np.array_equal(m.prev_h0_dev.numpy(), prev_h0)
        assert m.padded_rows.tolist() == [t * B + b for t in range(T) for b in range(bs[t])]
        last = [offs[n - 1] + b for b, n in enumerate(lens)]
        assert m.first_rows.tolist() == [list(range(B)), last] and m.last_rows.tolist() == [last, list(range(B))]
        assert m.rows == rows and m.T == T and m.max_batch == B and m.equal_lengths == (len(set(lens)) == 1)


def test_padded_list_notices_replaced_entries():
    """A PaddedList is a plain ``list`` to its callers (the reference's batch contract): once an entry has been replaced or
    dropped, consumers must read the entries, not the padded buffer (``intact``); in-place edits go through to the buffer."""
    from padertorch_amd.ops.sequence.pack_module import PaddedList, as_padded, pack_sequence
    for bf in (True, False):
        lens = [5, 4, 2]
        pad = torch.arange(3 * 5 * 2, dtype=torch.float32).view(3, 5, 2) if bf else \
            torch.arange(5 * 3 * 2, dtype=torch.float32).view(5, 3, 2)
        pl = PaddedList(pad.clone(), lens, batch_first=bf)
        assert pl.intact()
        pl[1].mul_(2.)                                  # in place: the buffer changes with it
        assert pl.intact()
        assert torch.equal(pack_sequence(pl).data, torch.nn.utils.rnn.pack_sequence(list(pl)).data)
        pl[2] = pl[2] + 100.                            # replaced: the list is the data now
        assert not pl.intact()
        want = torch.nn.utils.rnn.pack_sequence(list(pl))
        assert torch.equal(pack_sequence(pl).data, want.data)
        padded, lengths, _ = as_padded(pl, batch_first=True)
        assert lengths == lens and torch.equal(padded[2, :2], pl[2])
        del pl[2]
        assert not pl.intact()


def test_slot_layout_places_sequences_end_to_end():
    """ops.sequence.SlotLayout (host logic): longest-first packing into the emptiest slot, masks, predecessor tables, and the
    scatter / gather pair round-trips a batch-major padded tensor (idle rows zero)."""
    import numpy as np
    from padertorch_amd.ops.sequence import SlotLayout
    lengths = [9, 7, 7, 5, 4, 3, 3, 2]
    L = SlotLayout(lengths, slots=4)
    assert L.T == 11 and sorted(np.bincount(L.slot)) == [2, 2, 2, 2]
    total = np.zeros(4, int)
    for n, s in zip(lengths, L.slot):
        total[s] += n
    assert total.max() == L.T and abs(L.occupancy - sum(lengths) / 44.) < 1e-12
    for b, (n, s, t0) in enumerate(zip(lengths, L.slot, L.t0)):      # every sequence: contiguous, starts / ends flagged, nothing overlaps
        assert L.alive[t0:t0 + n, s].all() and L.first[t0, s] and L.last[t0 + n - 1, s]
        assert L.first[t0:t0 + n, s].sum() == 1 and L.last[t0:t0 + n, s].sum() == 1
    assert int(L.alive.sum()) == sum(lengths) and len(set(L.rows_host.tolist())) == sum(lengths)
    m = L.meta
    masks = m.masks_dev.view(-1, 3).numpy().view(np.uint64)
    for t in range(L.T):
        for k, arr in enumerate((L.alive, L.first, L.last)):
            assert int(masks[t, k]) == sum(1 << s for s in range(4) if arr[t, s])
    prev = m.prev_dev.numpy()
    for t in range(L.T):
        for s in range(4):
            r = t * 4 + s
            assert prev[0, r] == (r - 4 if L.alive[t, s] and not L.first[t, s] else m.rows)
            assert prev[1, r] == (r + 4 if L.alive[t, s] and not L.last[t, s] else m.rows)
    x = torch.arange(8 * 10 * 3, dtype=torch.float32).view(8, 10, 3) + 1.
    for b, n in enumerate(lengths):
        x[b, n:] = 0
    g = L.scatter_rows(x)
    assert g.shape == (44, 3) and int((g.abs().sum(1) > 0).sum()) == sum(lengths)
    assert torch.equal(L.gather_rows(g, 10), x)
    xr = x.clone().requires_grad_()
    (L.gather_rows(L.scatter_rows(xr) * 2., 10) * x).sum().backward()
    assert torch.equal(xr.grad, 2. * x)
    one = SlotLayout([5, 3, 2], slots=1)               # a single slot: plain concatenation
    assert one.T == 10 and one.t0 == [0, 5, 8]
    assert SlotLayout.cached((5, 3, 2), 1, 'cpu') is SlotLayout.cached([5, 3, 2], 1, torch.device('cpu'))
