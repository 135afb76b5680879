"""GPU: the HIP path on a TRAINED network (tests/golden/make_golden_trained.py: the reference's model trained by the
reference's own loop body, no layer scaled), through the C ABI, against what the REFERENCE computed on those weights:
one step on every engine, 10-step free-running rollouts with per-step edge-set equality at the flat 1e-4, the GD planner's
gradients and its whole returned dict, and the device trainer against the reference's loss curve over 240 iterations."""
import numpy as np
import pytest
import torch

from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd import train_gnn_dyn as T
from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel
from dyn_res_pile_manip_amd.planners import PlannerGD
from oracle import propnet_sparse as osp
from test_gpu_parity import check_rollout, disp_rel

pytestmark = pytest.mark.gpu
SIZES = ['n20', 'n50', 'n100', 'n300']
ENGINES = ['valu', 'mfma', 'split', 'fused']
# max|grad - reference's autograd| / max|reference's|, by the engine that wrote the tape: 5 x the worst OBSERVED over the
# sixteen cases (fused 8.0e-6 at n100_h2, fp32 tape 3.5e-6 at n50_h1; round 6) -- rounds 2 - 5 asserted 2e-3
GRAD_BOUND = {'fused': 4e-5, 'mfma': 2e-5}


@pytest.fixture(scope='module')
def eng(golden):
    from dyn_res_pile_manip_amd.engine import Engine
    e = Engine(0)
    e.load_weights(weights.blob_from_state_dict(golden.weights_trained), 0.08)
    e.M34 = osp.world2cam_affine(syn.demo_cam_extrinsics(), 24)
    e.set_camera(e.M34, 24.0, syn.demo_cam_params())
    e.W = osp.weights_np(golden.weights_trained)
    yield e
    e.close()


def test_the_fused_engine_serves_the_trained_weights(eng, golden):
    """The split-fp16 relation encoder's range shift 2^k and its proven activation bound on TRAINED matrices: inside fp16
    with room (the bound times 2^k stays below 65 504), so the default engine takes these weights without the fp32 fallback."""
    info = eng.range_info()
    g = golden.trained
    seen = max(float(g['one_step/%s/max_relation_hidden' % n]) for n in SIZES)
    print('\n[trained] range shift k = %d, proven bound %.4g (x 2^k = %.4g of 65 504), largest |w| %.4g; relation-encoder '
          'activations the reference saw: %.4g' % (info['shift'], info['bound'], info['bound'] * 2.0 ** info['shift'],
                                                   info['wmax'], seen))
    assert info['ok']
    assert info['bound'] * 2.0 ** info['shift'] <= 65504.0
    assert seen <= info['bound']


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('case', SIZES)
def test_one_step(eng, golden, case, engine):
    eng.set_engine(_lib.ENGINES[engine])
    g = golden.trained
    p = 'one_step/' + case + '/'
    a, s, sd, d = g[p + 'attr'], g[p + 's_cur'], g[p + 's_delta'], g[p + 'dens']
    idx, cnt = eng.build_graph(s, sd)
    np.testing.assert_array_equal(cnt, g[p + 'nbr_cnt'])
    np.testing.assert_array_equal(idx, g[p + 'nbr_idx'])
    np.testing.assert_allclose(eng.gen_s_delta(s, g[p + 'action']), sd, rtol=0, atol=3e-7)
    out = eng.step(a, s, sd, d)
    ref = g[p + 's_pred']
    assert disp_rel(out, ref, s) < 1e-4
    assert np.abs(out - ref).max() < 1e-5


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('case', SIZES)
def test_free_running_rollout(eng, golden, case, engine, exact_goal_transform):
    eng.set_engine(_lib.ENGINES[engine])
    g = golden.trained
    p = 'rollout/' + case + '/'
    ref = g[p + 'state_pred']
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    eng.set_goal(syn.goal_field(obs_goal), g[p + 'goal_coor'])
    states, rew = eng.rollout(g[p + 's_cur'], g[p + 'attr'], g[p + 'dens'], g[p + 'act_seqs'], want_reward=True)
    check_rollout(eng, None, g[p + 's_cur'], g[p + 'attr'], g[p + 'dens'], g[p + 'act_seqs'], ref, states)
    assert np.abs(states - ref).max() < 2e-5
    np.testing.assert_allclose(rew, g[p + 'next_r'][:, :, 0], rtol=1e-4)


def test_free_running_rollout_in_a_chip_filling_batch(eng, golden):
    """The rollout cases replicated to 1 024 rows (BASELINE configs[1]'s batch: the whole-sample kernels with several samples
    per workgroup, the edge-chain cache where it is on): every replica gives the small batch's trajectory, which the test
    above holds against the reference."""
    eng.set_engine(_lib.ENGINE_FUSED)
    g = golden.trained
    for case in SIZES:
        p = 'rollout/' + case + '/'
        acts = g[p + 'act_seqs']
        reps = 1024 // acts.shape[0]
        small, _ = eng.rollout(g[p + 's_cur'], g[p + 'attr'], g[p + 'dens'], acts)
        big, _ = eng.rollout(g[p + 's_cur'], g[p + 'attr'], g[p + 'dens'], np.tile(acts, (reps, 1, 1)))
        ref = np.tile(g[p + 'state_pred'], (reps, 1, 1, 1))
        check_rollout(eng, None, g[p + 's_cur'], g[p + 'attr'], g[p + 'dens'], np.tile(acts, (reps, 1, 1)), ref, big)
        assert np.abs(big - np.tile(small, (reps, 1, 1, 1))).max() < 2e-6, case


@pytest.mark.parametrize('tape', ['fused', 'mfma'])
@pytest.mark.parametrize('case', ['n20_h1', 'n20_h2', 'n50_h1', 'n50_h2', 'n100_h1', 'n100_h2', 'n300_h1', 'n300_h2'])
def test_gradients_match_the_reference(eng, golden, case, tape, exact_goal_transform):
    """Reverse mode through reward, predictor, three propagation steps, encoders and gen_s_delta against the reference's
    autograd on the trained weights -- with the tape written by the fused engine (km_prop<., TAPE>) and by the fp32 matrix
    engine (k_aggregate_tape: what the planner falls back to when the split-fp16 encoder refuses weights or inputs)."""
    eng.set_engine(_lib.ENGINES[tape])
    g = golden.trained
    p = 'grad/' + case + '/'
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    eng.set_goal(syn.goal_field(obs_goal), g[p + 'goal_coor'])
    lo, hi = syn.action_limits()
    ref_ga = g[p + 'grad_act']
    N = g[p + 's_cur'].shape[1]
    # the small batch (launch-per-stage kernels) and the same case replicated to a chip-filling one
    for reps in (1, -(-1100 // (ref_ga.shape[0] * ((N + 31) // 32)))):
        acts = np.tile(g[p + 'act_seqs'], (reps, 1, 1))
        eng.gd_begin(g[p + 's_cur'], g[p + 'attr'], g[p + 'dens'], acts, 0.05, lo, hi)
        r, ga, _ = eng.gd_grad()
        want = np.tile(ref_ga, (reps, 1, 1))
        np.testing.assert_allclose(r, np.tile(g[p + 'reward'][:, 0], reps), rtol=2e-5)
        err = float(np.abs(ga - want).max() / np.abs(ref_ga).max())
        print('[grad-err] trained %s tape=%s reps=%d: max|ga - ref| / max|ref| = %.3e' % (case, tape, reps, err))
        assert err < GRAD_BOUND[tape], (reps, err)
        np.testing.assert_array_equal(np.abs(ga).sum((1, 2)) == 0, np.abs(want).sum((1, 2)) == 0)
    eng.set_engine(_lib.ENGINE_FUSED)


def _model(golden, w=None):
    config = syn.default_config()
    config['mpc']['mpc_type'] = 'GD'
    model = PropNetDiffDenModel(config, True)
    w = golden.weights_trained if w is None else w
    model.load_state_dict({k[2:]: torch.from_numpy(np.asarray(w[k])) for k in w.files if k.startswith('w/')}, strict=False)
    return config, model


@pytest.mark.parametrize('case', ['n20', 'n50', 'n100'])
def test_the_gd_planner_on_trained_weights_matches_the_reference(golden, case, exact_goal_transform):
    """visualize_mpc.py:36-41's path: a trained checkpoint through trajectory_optimization_ptcl_multi_traj with
    mpc_type 'GD', called as env/flex_env.py:1048-1065 calls it -- the reference's own run on the same weights."""
    config, model = _model(golden)
    planner = PlannerGD(config, syn.SyntheticEnv(config))
    g = golden.trained
    p = 'gd/' + case + '/'
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    lo, hi = syn.action_limits()
    traj = g[p + 'act_seq'].shape[1]
    res = planner.trajectory_optimization_ptcl_multi_traj(
        g[p + 's_cur'], g[p + 'dens'], g[p + 'attr'], obs_goal, model, g[p + 'act_seq'], np.zeros(1), n_sample=traj,
        n_look_ahead=1, n_update_iter=int(g[p + 'n_update_iter']), action_lower_lim=lo, action_upper_lim=hi, use_gpu=True,
        time_lim=1e9)
    np.testing.assert_allclose(res['rew_mean'], g[p + 'out/rew_mean'], rtol=1e-4)
    np.testing.assert_allclose(res['rew_std'], g[p + 'out/rew_std'], rtol=2e-3)
    np.testing.assert_allclose(res['action_full'], g[p + 'out/action_full'], atol=2e-3)
    np.testing.assert_allclose(res['reward_full'], g[p + 'out/reward_full'], rtol=1e-4)
    np.testing.assert_allclose(res['action_sequence'], g[p + 'out/action_sequence'], atol=2e-3)
    np.testing.assert_allclose(res['observation_sequence'], g[p + 'out/observation_sequence'], atol=2e-5)
    np.testing.assert_allclose(res['reward'], g[p + 'out/reward'], rtol=1e-4)
    np.testing.assert_allclose(res['next_r'], g[p + 'out/next_r'], rtol=1e-4)
    assert res['iter_num'] == int(g[p + 'out/iter_num'])


def test_the_device_trainer_follows_the_reference_loss_curve(golden):
    """Row f4 beyond three iterations: `drp_train_step` started from the reference run's initial weights, fed the same
    240 batches (regenerated: synthetic.push_batch; checksums in the fixture), with the reference's optimiser settings.
    The first iterations agree to rounding; later ones to what two fp32 Adam runs of this loss keep in common -- measured,
    not assumed: the fixture carries the reference's OWN second run started one ulp away (`losses_twin`), and the device
    run's deviation is held to a small multiple of that one's."""
    g = golden.train_curve
    lr, beta1, B, Tn = g['hyper']

    class _Init(object):
        files = ['w/' + k[5:] for k in g.files if k.startswith('init/')]

        def __getitem__(self, k):
            return g['init/' + k[2:]]
    config, model = _model(golden, _Init())
    opt = T.DeviceAdam(model, float(lr), betas=(float(beta1), 0.999), n_rollout=int(Tn))
    n_it = len(g['losses'])
    losses = np.zeros(n_it)
    for it in range(n_it):
        batch = syn.push_batch(it, int(B), int(Tn))
        s = float(batch[0].astype(np.float64).sum() + batch[1].astype(np.float64).sum() + batch[4].astype(np.float64).sum())
        assert s == g['batch_sums'][it], it
        losses[it] = T.run_batch(model, opt, batch + (None,), 'train', int(Tn))
    ref, twin = g['losses'], g['losses_twin']
    win = 20

    def windows(x):
        return np.array([x[i:i + win].mean() for i in range(0, n_it, win)])
    rel, rel_twin = np.abs(losses / ref - 1), np.abs(twin / ref - 1)
    wdev, wdev_twin = np.abs(windows(losses) / windows(ref) - 1), np.abs(windows(twin) / windows(ref) - 1)
    print('\n[train curve] loss deviation from the reference run -- device: first 10 max %.2e, median %.2e, max %.2e, windowed '
          'max %.2e | the reference started one ulp away: first 10 max %.2e, median %.2e, max %.2e, windowed max %.2e; last '
          'window %.4e (device) %.4e (reference)' % (rel[:10].max(), np.median(rel), rel.max(), wdev.max(), rel_twin[:10].max(),
                                                     np.median(rel_twin), rel_twin.max(), wdev_twin.max(),
                                                     windows(losses)[-1], windows(ref)[-1]))
    # the first iterations agree to rounding; then the yardstick is what the reference does to itself: two fp32 runs of this
    # loop one ulp apart drift by `rel_twin` -- the device run may drift by a small multiple of that, no more
    assert rel[:10].max() < 2e-3
    assert np.median(rel) < 4 * np.median(rel_twin) + 1e-3
    assert wdev.max() < 4 * wdev_twin.max() + 1e-2
    assert losses[-win:].mean() < 0.1 * losses[:5].mean()
    # the weights after the run: the same network up to that divergence -- its loss on a held-out batch equals the
    # reference's end point's (through the device, eval mode) within 5 %
    held = syn.push_batch(100000, int(B), int(Tn))
    mine = model.engine.train_step(*held, mode='eval')[0]

    class _After(object):
        files = ['w/' + k[6:] for k in g.files if k.startswith('after/')]

        def __getitem__(self, k):
            return g['after/' + k[2:]]
    _, ref_model = _model(golden, _After())
    ref_model.engine.train_begin(int(Tn), float(lr), float(beta1))
    theirs = ref_model.engine.train_step(*held, mode='eval')[0]
    print('[train curve] held-out loss: device-trained %.5e, reference-trained %.5e' % (mine, theirs))
    assert abs(mine / theirs - 1) < 4 * wdev_twin.max() + 1e-2


@pytest.mark.parametrize('N,ns,H', [(300, 1024, 10), (1200, 128, 6), (50, 1024, 10), (20, 4096, 5)])
def test_baseline_sizes_on_the_trained_weights(eng, N, ns, H):
    """BASELINE configs[1] (1024 x 300 x 10), a slice of configs[4]'s pile (1 200 particles) and the small-pile shapes at full
    batch, on the TRAINED network: pushed particles move by a push's length every step (the seed-0 fixtures: 2 %).  Teacher-
    forced against the oracle on 24 rows spread over the batch -- every step starts from the device's previous state with the
    device's impulses, so one flipped near-tie cannot cascade --, at the flat 1e-4 of the step's displacement."""
    eng.set_engine(_lib.ENGINE_FUSED)
    s0, dens, attr = syn.make_pile(N, 1, seed=N + 1, kind='blob' if N <= 50 else 'uniform')
    acts = np.stack([syn.pushes_through(np.tile(s0, (ns, 1, 1)), seed=7 * t + N) for t in range(H)], 1)
    states, _ = eng.rollout(s0, attr, dens, acts)
    assert np.isfinite(states).all()
    rows = np.unique(np.linspace(0, ns - 1, 24).astype(int))
    prev = np.repeat(s0[:1], len(rows), 0)
    at, de = np.repeat(attr[:1], len(rows), 0), np.repeat(dens[:1], len(rows))
    worst, moved = 0.0, 0.0
    for t in range(H):
        sd = eng.gen_s_delta(prev, acts[rows, t])
        ref = osp.predict_one_step(eng.W, at, prev, sd, de)
        out = states[rows, t]
        disp = np.abs(ref - prev).reshape(len(rows), -1).max(1)
        worst = max(worst, float((np.abs(out - ref).reshape(len(rows), -1).max(1) / np.maximum(disp, 1e-12)).max()))
        moved = max(moved, float(disp.max()))
        prev = out
    print('\n[trained, %d x %d x %d] worst step error %.2e of the displacement; largest displacement %.3f' % (ns, N, H, worst, moved))
    assert worst < 1e-4 and moved > 0.02
