"""Pin oracle/losses_np.py + oracle/features_np.py against G1/G3/G4/G5."""
import numpy as np

from oracle import features_np, losses_np


def test_pit_toys(g1):
    for toy in g1['pit_toys']:
        got = losses_np.pit_loss(np.array(toy['estimate'], float), np.array(toy['target'], float), -2)
        np.testing.assert_allclose(got, toy['loss'], rtol=1e-4)


def test_pit_doctests(g1):
    for d in g1['pit_doctests']:
        got = losses_np.pit_loss(np.ones(d['est_shape']), np.zeros(d['est_shape']), d['axis'])
        assert got == d['loss']
    est = np.stack([np.ones((5, 4)), np.zeros((5, 4))])
    loss, perm = losses_np.pit_loss(est, est[[1, 0]], axis=0, return_permutation=True)
    assert loss == 0. and perm == (1, 0)          # source_separation.py:80-83


def test_dc_toys(g1):
    for toy in g1['dc_toys']:
        got = losses_np.deep_clustering_loss(np.array(toy['embedding'], float), np.array(toy['target'], float))
        np.testing.assert_allclose(got, toy['loss'], atol=1e-12)


def test_hungarian(g1):
    h = g1['hungarian']
    assert losses_np.pit_loss_from_loss_matrix(-np.array(h['score']), reduction='sum') == h['loss_sum']


def test_pit_vs_reference(g4):
    for name in g4['names']:
        axis = int(g4[f'{name}_axis'])
        loss, perm = losses_np.pit_loss(g4[f'{name}_est'], g4[f'{name}_tgt'], axis, return_permutation=True)
        np.testing.assert_allclose(loss, g4[f'{name}_loss'], rtol=1e-5)
        assert list(perm) == list(g4[f'{name}_perm']), name
        if f'{name}_pairwise' in g4:
            pw = losses_np.pairwise_losses(g4[f'{name}_est'], g4[f'{name}_tgt'], axis)
            np.testing.assert_allclose(pw, g4[f'{name}_pairwise'], rtol=1e-5)
            hl, col = losses_np.pit_loss_from_loss_matrix(pw, return_permutation=True)
            np.testing.assert_allclose(hl, g4[f'{name}_hungarian_loss'], rtol=1e-5)
            assert list(col) == list(g4[f'{name}_hungarian_col'])
            # for MSE the Hungarian 'mean' value equals the brute-force value
            np.testing.assert_allclose(hl, loss, rtol=1e-6)
            # inverse permutation convention (appendix B.6)
            assert list(np.argsort(col)) == list(perm)
    assert list(g4['tie_perm']) == [0, 1, 2]


def test_dc_vs_reference(g5):
    got = losses_np.deep_clustering_loss(g5['x'], g5['t'])
    np.testing.assert_allclose(got, g5['loss64'], rtol=1e-10)
    np.testing.assert_allclose(got, g5['loss'], atol=1e-4)


def test_features_vs_reference(g3):
    f = features_np.pre_batch_transform(g3['s'], g3['y'])
    assert f['num_frames'] == int(g3['num_frames']) == 15
    for k in ['X_abs', 'Y_abs', 'cos_phase_difference']:
        assert f[k].dtype == np.float32 and f[k].shape == g3[k].shape
        np.testing.assert_array_equal(f[k], g3[k])
    np.testing.assert_array_equal(f['Y'], g3['Y'])


def test_pit_loss_from_loss_matrix_greedy_doctest():
    """The reference doctest of ``pit_loss_from_loss_matrix`` (``source_separation.py:258-271``): optimal -26, greedy -21 with
    per-source losses [-11, -10, -0]."""
    import numpy as np
    import torch
    from padertorch_amd.ops.losses.source_separation import pit_loss_from_loss_matrix
    score = np.array([[11., 10, 0], [4, 5, 10], [6, 0, 5]])
    m = torch.tensor(-score)
    assert float(pit_loss_from_loss_matrix(m, reduction='sum', algorithm='optimal')) == -26.
    assert float(pit_loss_from_loss_matrix(m, reduction='sum', algorithm='brute_force')) == -26.
    assert float(pit_loss_from_loss_matrix(m, reduction='sum', algorithm='greedy')) == -21.
    per, perm = pit_loss_from_loss_matrix(m, reduction=None, algorithm='greedy', return_permutation=True)
    assert per.tolist() == [-11., -10., -0.] and list(perm) == [0, 2, 1]
