"""Pin the oracle on a TRAINED network: both restatements under oracle/ against vectors the reference produced on weights
it was trained to (tests/golden/make_golden_trained.py: no layer scaled, the reference's own training-loop body on
synthetic push episodes).  CPU only."""
import numpy as np
import pytest

from oracle import propnet_dense as od
from oracle import propnet_sparse as osp
from dyn_res_pile_manip_amd import synthetic as syn

SIZES = ['n20', 'n50', 'n100', 'n300']


def disp_rel(out, ref, s_cur):
    return np.abs(out - ref).max() / max(np.abs(ref - s_cur).max(), 1e-12)


@pytest.fixture(scope='module')
def W(golden):
    return od.load_weights(golden.weights_trained), osp.weights_np(golden.weights_trained)


def test_the_generator_reproduces_the_batches_the_reference_was_trained_on(golden):
    """The training fixtures carry checksums, not data: the tests regenerate the episodes (synthetic.push_batch, seeded
    numpy) -- the same floats the reference's loop consumed."""
    g = golden.train_curve
    lr, beta1, B, T = g['hyper']
    for it in (0, 1, 17, 100, len(g['batch_sums']) - 1):
        states, sdelta, attrs, pnums, dens = syn.push_batch(it, int(B), int(T))
        s = float(states.astype(np.float64).sum() + sdelta.astype(np.float64).sum() + dens.astype(np.float64).sum())
        assert s == g['batch_sums'][it], it
        assert states.shape[1] == int(T) + 1 and (attrs == 0).all() and pnums.max() == states.shape[2]


def test_the_trained_weights_moved_away_from_their_initialisation(golden):
    """No layer is scaled and the network has learnt: the loss fell by more than an order of magnitude, the predictor's
    last layer is not the x 0.02 stand-in of the other fixtures, pushed particles move by a push's length."""
    g = golden.train_curve
    assert g['losses'][:5].mean() > 10 * g['losses'][-20:].mean()
    w = golden.weights_trained
    k = 'w/model.particle_predictor.linear_1.weight'
    assert np.abs(w[k] - g['init/' + k[2:]]).max() > 1e-2
    t = golden.trained
    for n in SIZES:
        d = np.abs(t['one_step/%s/s_pred' % n] - t['one_step/%s/s_cur' % n]).max()
        assert 0.02 < d < 0.6, (n, d)


@pytest.mark.parametrize('case', SIZES)
def test_one_step_dense_and_sparse(golden, W, case):
    g = golden.trained
    p = 'one_step/' + case + '/'
    a, s, sd, d = g[p + 'attr'], g[p + 's_cur'], g[p + 's_delta'], g[p + 'dens']
    ref = g[p + 's_pred']
    out = od.predict_one_step(W[0], a, s, sd, d).numpy()
    assert disp_rel(out, ref, s) < 1e-5
    idx, cnt = osp.build_neighbours(s, sd)
    np.testing.assert_array_equal(cnt, g[p + 'nbr_cnt'].astype(np.int32))
    np.testing.assert_array_equal(idx, g[p + 'nbr_idx'].astype(np.int32))
    out2 = osp.predict_one_step(W[1], a, s, sd, d)
    assert disp_rel(out2, ref, s) < 1e-4
    assert np.abs(out2 - ref).max() < 2e-6
    # the impulse of the fixture is the reference's gen_s_delta of the recorded pushes
    M = osp.world2cam_affine(syn.demo_cam_extrinsics(), 24)
    np.testing.assert_allclose(osp.gen_s_delta(s, g[p + 'action'], M, 24), sd, rtol=0, atol=2e-7)


@pytest.mark.parametrize('case', SIZES)
def test_free_running_rollout(golden, W, case):
    g = golden.trained
    p = 'rollout/' + case + '/'
    ref = g[p + 'state_pred']
    ext = syn.demo_cam_extrinsics()
    out = od.rollout(W[0], g[p + 's_cur'], g[p + 'dens'], g[p + 'attr'], g[p + 'act_seqs'], ext, 24).numpy()
    assert np.abs(out - ref).max() < 5e-6
    M = osp.world2cam_affine(ext, 24)
    taps = {}
    out2 = osp.rollout(W[1], g[p + 's_cur'], g[p + 'dens'], g[p + 'attr'], g[p + 'act_seqs'], M, 24, taps=taps)
    nb = g[p + 's_cur'].shape[0]
    prev = np.tile(g[p + 's_cur'], (ref.shape[0] // nb, 1, 1))
    for t in range(ref.shape[1]):
        # the lists the oracle's own trajectory induces are the reference trajectory's
        idx_r, cnt_r = osp.build_neighbours(prev, osp.gen_s_delta(prev, g[p + 'act_seqs'][:, t], M, 24))
        np.testing.assert_array_equal(taps['nbr_cnt'][t], cnt_r, err_msg='step %d' % t)
        np.testing.assert_array_equal(taps['nbr_idx'][t], idx_r, err_msg='step %d' % t)
        assert disp_rel(out2[:, t], ref[:, t], prev) < 1e-3, t
        prev = ref[:, t]
    assert np.abs(out2 - ref).max() < 1e-5
    # all-step rewards of the reference's evaluate_traj on its own trajectory
    G = syn.goal_field(syn.goal_distance_image(syn.goal_mask('I')))
    B, H, N, _ = ref.shape
    r = osp.reward(ref.reshape(B * H, N, 3), G, syn.demo_cam_params(), g[p + 'goal_coor']).reshape(B, H, 1)
    np.testing.assert_allclose(r, g[p + 'next_r'], rtol=1e-5)


@pytest.mark.parametrize('case', ['n20_h1', 'n20_h2', 'n50_h1', 'n50_h2', 'n100_h1', 'n100_h2', 'n300_h1'])
def test_gradients_of_the_planner_loss(golden, W, case):
    g = golden.trained
    p = 'grad/' + case + '/'
    G = syn.goal_field(syn.goal_distance_image(syn.goal_mask('I')))
    rew, ga, _ = od.gd_loss_and_grads(W[0], g[p + 's_cur'], g[p + 'dens'], g[p + 'attr'], g[p + 'act_seqs'], G,
                                   syn.demo_cam_params(), g[p + 'goal_coor'], syn.demo_cam_extrinsics(), 24)
    ref = g[p + 'grad_act']
    assert np.abs(ref).max() > 0
    np.testing.assert_allclose(np.asarray(rew).reshape(-1), g[p + 'reward'].reshape(-1), rtol=1e-5)
    assert np.abs(np.asarray(ga) - ref).max() < 1e-6 * np.abs(ref).max()       # observed: bit-identical (round 5 asserted 1e-4)


def test_the_training_loop_body_follows_the_reference(golden):
    """oracle.propnet_dense.train_loss_and_grads + adam_steps (the CPU restatement of train/train_gnn_dyn.py:159-210) from
    the recorded initial weights over the regenerated batches: the reference's first losses."""
    g = golden.train_curve
    lr, beta1, B, T = g['hyper']
    W0 = {k[5:]: g[k] for k in g.files if k.startswith('init/')}
    n = 6
    state = {'it': 0}

    def grads_fn(Wc):
        batch = syn.push_batch(state['it'], int(B), int(T))
        state['it'] += 1
        return od.train_loss_and_grads(Wc, *batch)
    losses, _ = od.adam_steps(W0, grads_fn, n, lr=float(lr), beta1=float(beta1))
    np.testing.assert_allclose(losses, g['losses'][:n], rtol=2e-4)
