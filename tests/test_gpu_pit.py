"""HIP PIT loss vs the oracle / reference goldens (GPU, via the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import losses_np

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_toys_and_doctests(g1):
    from padertorch_amd.ops import pit_loss
    for toy in g1['pit_toys']:
        got = pit_loss(torch.tensor(toy['estimate'], dtype=torch.float32, device=DEV),
                       torch.tensor(toy['target'], dtype=torch.float32, device=DEV), axis=-2)
        np.testing.assert_allclose(got.item(), toy['loss'], rtol=1e-4)
    for d in g1['pit_doctests']:
        got = pit_loss(torch.ones(d['est_shape'], device=DEV), torch.zeros(d['est_shape'], device=DEV), d['axis'])
        assert got.item() == d['loss']
    est = torch.stack([torch.ones(5, 4), torch.zeros(5, 4)]).to(DEV)
    loss, perm = pit_loss(est, est[(1, 0), :, :], axis=0, return_permutation=True)
    assert loss.item() == 0. and perm == (1, 0)
    # cross-entropy keeps the reference's generic brute-force path (source_separation.py:70-73)
    got = pit_loss(torch.ones(4, 2, 5, device=DEV), torch.zeros(4, 5, dtype=torch.int64, device=DEV), 1,
                   loss_fn=torch.nn.functional.cross_entropy)
    np.testing.assert_allclose(got.item(), 0.6931, atol=1e-4)


def test_vs_reference_goldens(g4):
    from padertorch_amd.ops import pit_loss
    for name in g4['names']:
        axis = int(g4[f'{name}_axis'])
        est = torch.from_numpy(g4[f'{name}_est']).to(DEV)
        tgt = torch.from_numpy(g4[f'{name}_tgt']).to(DEV)
        loss, perm = pit_loss(est, tgt, axis, return_permutation=True)
        np.testing.assert_allclose(loss.item(), g4[f'{name}_loss'], rtol=1e-5)
        assert list(perm) == list(g4[f'{name}_perm']), name      # permutation choice: exact
    est = torch.from_numpy(g4['b_est']).to(DEV).requires_grad_(True)
    pit_loss(est, torch.from_numpy(g4['b_tgt']).to(DEV), axis=-2).backward()
    np.testing.assert_allclose(est.grad.cpu().numpy(), g4['b_grad'], atol=1e-6)


@pytest.mark.parametrize('K,F,lens', [(2, 33, [19, 17, 17, 4]), (3, 33, [19, 17, 17, 4]), (4, 33, [19, 17, 17, 4]), (5, 33, [19, 17, 17, 4])] + [
    (int(r.randint(2, 6)), int(r.choice([1, 7, 64, 257, 300])), sorted((int(x) for x in r.randint(1, 60, int(r.randint(1, 12)))), reverse=True))
    for r in (np.random.RandomState(900 + i) for i in range(14))])
def test_fused_review_vs_oracle(K, F, lens):
    """pit/model.py:117-140 fused over a ragged batch, batch- and time-major masks, + gradients (fixed cases and random K / F / lengths)."""
    from padertorch_amd.ops.losses import pit_mse_ips_losses
    rng = np.random.RandomState(K)
    B, T = len(lens), max(lens)
    mask = np.abs(rng.standard_normal((B, T, K, F))).astype(np.float32)
    Y = np.abs(rng.standard_normal((B, T, F))).astype(np.float32)
    X = np.abs(rng.standard_normal((B, T, K, F))).astype(np.float32)
    C = np.cos(rng.uniform(-3, 3, (B, T, K, F))).astype(np.float32)
    ref_mse, ref_ips, pm, pi = losses_np.pit_review_losses(
        [mask[b, :l] for b, l in enumerate(lens)], [Y[b, :l] for b, l in enumerate(lens)],
        [X[b, :l] for b, l in enumerate(lens)], [C[b, :l] for b, l in enumerate(lens)])
    ld = torch.tensor(lens, dtype=torch.int32, device=DEV)
    d = lambda a: torch.from_numpy(a).to(DEV)
    for batch_first in [True, False]:
        m = d(mask) if batch_first else d(mask).transpose(0, 1).contiguous()
        m.requires_grad_(True)
        loss, perm, ex = pit_mse_ips_losses(m, d(Y), d(X), d(C), ld, mask_batch_first=batch_first)
        np.testing.assert_allclose(loss[0].item(), ref_mse, rtol=2e-6)
        np.testing.assert_allclose(loss[1].item(), ref_ips, rtol=2e-6)
        assert perm[:, 0].tolist() == [list(p) for p in pm]
        assert perm[:, 1].tolist() == [list(p) for p in pi]
        (0.25 * loss[0] + 1.0 * loss[1]).backward()
        # torch reference gradient
        mt = torch.from_numpy(mask).double().requires_grad_(True)
        tot = 0
        for b, l in enumerate(lens):
            est = mt[b, :l] * torch.from_numpy(Y[b, :l]).double()[:, None, :]
            tg = torch.from_numpy(X[b, :l]).double()
            tc = tg * torch.from_numpy(C[b, :l]).double()
            tot = tot + 0.25 * ((est[:, list(pm[b])] - tg) ** 2).mean() / B \
                + ((est[:, list(pi[b])] - tc) ** 2).mean() / B
        tot.backward()
        got = m.grad if batch_first else m.grad.transpose(0, 1)
        np.testing.assert_allclose(got.cpu().numpy(), mt.grad.numpy(), atol=1e-7, rtol=1e-4)


def test_full_size_property():
    """BASELINE-size batch (B=32, T=253, F=257, K=2): permuting the estimate's speakers permutes the
    answer, the loss of est == target is 0, and minibatch loss == mean of single-example losses."""
    from padertorch_amd.ops.losses import pit_mse_ips_losses
    g = torch.Generator().manual_seed(0)
    B, T, K, F = 32, 253, 2, 257
    mask = torch.rand(B, T, K, F, generator=g).to(DEV)
    Y = torch.rand(B, T, F, generator=g).to(DEV)
    X = torch.rand(B, T, K, F, generator=g).to(DEV)
    C = (2 * torch.rand(B, T, K, F, generator=g) - 1).to(DEV)
    loss, perm, ex = pit_mse_ips_losses(mask, Y, X, C)
    loss2, perm2, ex2 = pit_mse_ips_losses(mask.flip(2).contiguous(), Y, X, C)
    assert torch.equal(ex, ex2) or torch.allclose(ex, ex2, rtol=1e-6)
    assert torch.equal(perm2, 1 - perm)
    np.testing.assert_allclose(loss.cpu().numpy(), ex.mean(0).cpu().numpy(), rtol=1e-6)
    singles = torch.stack([pit_mse_ips_losses(mask[b:b + 1], Y[b:b + 1], X[b:b + 1], C[b:b + 1])[0]
                           for b in range(0, B, 8)])
    np.testing.assert_allclose(singles.cpu().numpy(), ex[::8].cpu().numpy(), rtol=1e-6)
    ones = torch.ones(B, T, F, device=DEV)
    l0, _, _ = pit_mse_ips_losses(X, ones, X, torch.ones_like(X))
    assert l0.abs().max().item() == 0.
    # runs are bitwise reproducible (fixed reduction order)
    assert torch.equal(pit_mse_ips_losses(mask, Y, X, C)[2], ex)


def test_pairwise_and_hungarian_vs_reference(g1, g4):
    """compute_pairwise_losses + pit_loss_from_loss_matrix (source_separation.py:127-312)."""
    from padertorch_amd.ops.losses.source_separation import compute_pairwise_losses, pit_loss_from_loss_matrix
    for name in g4['names']:
        if f'{name}_pairwise' not in g4:
            continue
        axis = int(g4[f'{name}_axis'])
        pw = compute_pairwise_losses(torch.from_numpy(g4[f'{name}_est']).to(DEV),
                                     torch.from_numpy(g4[f'{name}_tgt']).to(DEV), axis)
        np.testing.assert_allclose(pw.cpu().numpy(), g4[f'{name}_pairwise'], rtol=1e-5)
        loss, col = pit_loss_from_loss_matrix(pw, return_permutation=True)
        np.testing.assert_allclose(loss.item(), g4[f'{name}_hungarian_loss'], rtol=1e-5)
        assert list(col) == list(g4[f'{name}_hungarian_col'])
    h = g1['hungarian']
    got = pit_loss_from_loss_matrix(-torch.tensor(h['score'], device=DEV), reduction='sum')
    assert got.item() == h['loss_sum']
    # cross-entropy pairwise branch (source_separation.py:172-175 doctest: 0.6931 with 'sum')
    est, tgt = torch.ones(4, 2, 5, device=DEV), torch.zeros(4, 5, dtype=torch.int64, device=DEV)
    pw = compute_pairwise_losses(est, tgt, 1, loss_fn=torch.nn.functional.cross_entropy)
    np.testing.assert_allclose(pit_loss_from_loss_matrix(pw, reduction='sum').item(), 0.6931, atol=1e-4)
