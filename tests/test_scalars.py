"""ops.scalars: ``pick`` is ``vector[i]`` (value and gradient), ``Trainer.backward`` is ``loss.backward()``, and the summary's values
are staged without a concatenation when they already lie in one tensor (CPU; the launch counts are a GPU matter:
scripts/dbg_ops_between.py)."""
import torch

import padertorch_amd as pt
from padertorch_amd.ops import scalars


def test_pick_is_indexing_for_values_and_gradients():
    torch.manual_seed(0)
    x = torch.randn(5, requires_grad=True)
    v = x * 2.
    a, b = scalars.pick(v, 1), scalars.pick(v, 3)
    assert float(a) == float(v[1]) and float(b) == float(v[3])
    assert scalars.picked_from(a)[0] is v and scalars.picked_from(b)[1] == 3
    (0.5 * a + 3. * b).backward()
    x2 = x.detach().clone().requires_grad_()
    v2 = x2 * 2.
    (0.5 * v2[1] + 3. * v2[3]).backward()
    assert torch.equal(x.grad, x2.grad)
    # the cached one-hot vectors are constants: a second backward pass finds them unchanged
    x.grad = None
    v = x * 2.
    (scalars.pick(v, 1) + scalars.pick(v, 1) + scalars.pick(v, 3)).backward()
    assert x.grad.tolist() == [0., 4., 0., 2., 0.]


def test_pick_of_a_vector_without_graph_is_a_plain_view():
    v = torch.arange(3.)
    assert float(scalars.pick(v, 2)) == 2. and scalars.picked_from(scalars.pick(v, 2))[1] == 2
    assert scalars.picked_from(v[2]) is None


def test_trainer_backward_is_loss_backward(tmp_path):
    w = torch.nn.Parameter(torch.tensor([1., -2.]))
    loss = (w ** 2).sum()
    pt.Trainer.backward(loss)
    assert w.grad.tolist() == [2., -4.]
    assert scalars.unit_grad(loss) is scalars.unit_grad(loss.detach() * 2)        # one cached 1. per device and dtype


def test_loss_values_without_a_concatenation():
    vec = torch.tensor([3., 5.], requires_grad=True) * 1.
    losses = {'a': scalars.pick(vec, 0), 'b': scalars.pick(vec, 1)}
    # weights a = 0, b = 1: the weighted sum IS losses['b'] (the Trainer's trivial-factor rule) and it is the vector's last element
    vals, index = pt.Trainer._loss_values(losses, ['a', 'b'], losses['b'])
    assert vals.data_ptr() == vec.data_ptr() and index == [0, 1, 1]
    # weights a = 1, b = 0: the sum is not the last staged value - the general path, sum last
    vals, index = pt.Trainer._loss_values(losses, ['a', 'b'], losses['a'])
    assert vals.tolist() == [3., 5., 3.] and index == [0, 1, 2]
    # a single loss
    one = vec.sum()
    vals, index = pt.Trainer._loss_values({'l': one}, ['l'], one)
    assert vals.tolist() == [8.] and index == [0, 0]
    # a weighted sum that is a new tensor
    total = 0.5 * losses['a'] + losses['b']
    vals, index = pt.Trainer._loss_values(losses, ['a', 'b'], total)
    assert vals.tolist() == [3., 5., 6.5] and index == [0, 1, 2]


def test_capture_zero_block_hands_out_its_words_once_and_survives_the_capture():
    """ops.capture.zero_block: the ONE zero-filled block at the head of a captured step serves every accumulation word of the step, each
    word once; it stays referenced when the capture ends (its memory belongs to the graph).  (Host logic only: no stream capture on CPU.)"""
    from padertorch_amd.ops import capture
    with capture.capture_mode():
        capture.zero_block(torch.device('cpu'), words=8)
        words = [capture.zero_word(torch.device('cpu')) for _ in range(8)]
        block = capture._shared[0]
        assert all(w.numel() == 1 and int(w) == 0 for w in words)
        assert sorted(w.data_ptr() for w in words) == [block.data_ptr() + 4 * i for i in range(8)]
        assert capture._shared[1] == 8
    assert not capture._shared and any(e[0] is block for e in capture._zero_blocks)
    assert not capture.ACTIVE
