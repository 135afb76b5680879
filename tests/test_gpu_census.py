"""GPU: the parity CENSUS on the trained network -- every push sequence, not the ones that cannot diverge.

tests/golden/census.npz (make_golden_census.py) holds, per pile size (20 / 50 / 100 / 300 / 600 particles), 64 UNFILTERED 10-step
push sequences and one 1 024-row MPPI population (600 particles: 32 and 128) as the REFERENCE rolled them out on weights_trained.npz: trajectory, all-step rewards, a hash of every
receiver's sender list taken from the reference's own Rr / Rs, the distance of every step's graph from a decision changing
(`margin`), and the same rows through the reference twice more, started one ulp up and one ulp down (the `twin_*` arrays): how
far the reference drifts from ITSELF.  A free-running rollout is a chaotic map with discontinuities (model/gnn_dyn.py:231-237:
`topk` and a threshold on fp32 squared distances): one ulp at the start is 1e-5 after ten steps without any list changing, and
1e-4 after one changes.  What "results identical to the reference's" can mean for a second fp32 implementation, and what is
asserted here (tests/_census.py: TAU = 5e-8, a hundred ulps of adj_thresh^2):

 (a) up to a row's first step NEAR A TIE (margin < TAU) the device's neighbour lists ARE the reference's on every row and step,
     the first step holds the flat 1e-4 of the displacement, and the accumulated deviation stays within 4 x the twins';
 (b) from that step on lists may differ (the twins' do too): the count of rows where they do is held to 4 x the twins' (+ 2),
     the state deviation to 4 x the twins', the final-reward deviation of the 1 024-row population to 4 x the twins' at the
     50th / 90th / 99th percentile; the largest single reward deviation is printed, not bounded by the twins' (a heavy tail:
     which particle a flipped edge moved);
 (c) one MPPI iteration on 1 024 rows: the softmax-mean update (planners.py:549-561) agrees with the reference's to 4 x the
     twins' own spread, the arg-max row is the reference's.
The printed census is DESIGN.md section 2's table."""
import numpy as np
import pytest
from scipy.special import softmax

import _census as C
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd.planners import world2cam_affine

pytestmark = pytest.mark.gpu
ENGINES = ['fused', 'mfma', 'split', 'valu']
K = 4.0            # "a small multiple of the reference's deviation from itself": the rule of test_the_device_trainer_follows...


@pytest.fixture(scope='module')
def eng(golden):
    from dyn_res_pile_manip_amd.engine import Engine
    e = Engine(0)
    e.load_weights(weights.blob_from_state_dict(golden.weights_trained), 0.08)
    e.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    e.obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    yield e
    e.close()


def fmt(v):
    return np.array2string(np.asarray(v), formatter={'float_kind': lambda x: '%.1e' % x}, max_line_width=200)


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('case', C.SIZES)
def test_census_rows(eng, golden, case, engine):
    eng.set_engine(_lib.ENGINES[engine])
    g = golden.census
    p = 'census/' + case + '/'
    eng.set_goal(syn.goal_field(eng.obs_goal), g[p + 'goal_coor'])
    states, rew, flips, dev = C.device_rows(eng, g, p)
    margin, disp = g[p + 'margin'], C.displacement(g, p)
    tw_dev, tw_flips = g[p + 'twin_dev'], g[p + 'twin_flips']                # [2,B,H]
    B, H = margin.shape
    first = C.first_true(C.near_tie(g, p))                                  # the row's first step near a tie (H: none)
    pre = np.arange(H)[None, :] < first[:, None]
    disp_b = disp.max(0)                                                    # the batch's displacement per step

    # ---- (a) before the first near-tie ---------------------------------------------------------------------------------------
    assert (flips[pre] == 0).all(), 'a list differs from the reference\'s %d steps before any near-tie' % (flips[pre] > 0).sum()
    assert dev[:, 0].max() < 1e-4 * disp_b[0]                               # one step, no history: the flat 1e-4
    pre_dev = np.array([dev[pre[:, t], t].max() if pre[:, t].any() else 0.0 for t in range(H)])
    pre_tw = np.array([tw_dev[:, pre[:, t], t].max() if pre[:, t].any() else 0.0 for t in range(H)])
    assert (pre_dev <= np.maximum(1e-4 * disp_b, K * pre_tw)).all(), (fmt(pre_dev), fmt(pre_tw))

    # ---- (b) from the first near-tie on --------------------------------------------------------------------------------------
    crossed = first < H
    dev_rows = (flips > 0).any(1)
    tw_rows = (tw_flips > 0).any(2)                                         # [2,B]
    assert not (dev_rows & ~crossed).any()
    assert dev_rows.sum() <= K * tw_rows.sum(1).max() + 2
    assert dev.max() <= K * max(tw_dev.max(), 1e-6), (dev.max(), tw_dev.max())
    still = int(sum((flips[b, first[b]:] == 0).all() for b in range(B) if crossed[b]))
    still_tw = [int(sum((tw_flips[q, b, first[b]:] == 0).all() for b in range(B) if crossed[b])) for q in range(2)]
    ff = C.first_true(flips > 0)
    m_at = [margin[b, ff[b]] for b in range(B) if ff[b] < H]
    post = np.array([dev[b, ff[b]:].max() for b in range(B) if ff[b] < H])
    tw_post = np.array([tw_dev[q, b, C.first_true(tw_flips[q] > 0)[b]:].max() for q in range(2) for b in range(B) if tw_rows[q, b]])
    d_r, tw_r = np.abs(rew[:, -1] - g[p + 'next_r'][:, -1]), np.abs(g[p + 'twin_next_r'][:, :, -1] - g[p + 'next_r'][None, :, -1])
    print('\n[census %s %s] %d rows x %d steps; rows reaching a near-tie (margin < %.0e): %d, first such step %s' %
          (case, engine, B, H, C.TAU, crossed.sum(), np.bincount(first, minlength=H + 1)[:H]))
    print('   before it: %d row-steps, lists equal in all; deviation / displacement per step %s (twins %s)' %
          (pre.sum(), fmt(pre_dev / disp_b), fmt(pre_tw / disp_b)))
    print('   after it: lists still the reference\'s in %d of %d rows (twins: %d, %d); rows with a flipped list: device %d, twins %d / %d; '
          'margin at the device\'s first flips %s' % (still, crossed.sum(), still_tw[0], still_tw[1], dev_rows.sum(), tw_rows[0].sum(),
                                                      tw_rows[1].sum(), fmt(np.sort(m_at))))
    print('   deviation after a row\'s first flip: device median %.1e max %.1e (twins median %.1e max %.1e); over all rows: device %.1e, twins %.1e' %
          (np.median(post) if post.size else 0, post.max() if post.size else 0, np.median(tw_post) if tw_post.size else 0,
           tw_post.max() if tw_post.size else 0, dev.max(), tw_dev.max()))
    print('   |d final reward| (|r| ~ %.0f): device median %.1e max %.1e; twins median %.1e max %.1e' %
          (np.abs(g[p + 'next_r'][:, -1]).mean(), np.median(d_r), d_r.max(), np.median(tw_r), tw_r.max()))
    eng.set_engine(_lib.ENGINE_FUSED)


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('case', C.SIZES)
def test_census_mppi_iteration(eng, golden, case, engine):
    """1 024 rows the reference's own sampler drew around census row 0 (planners.py:69-190), rolled out, scored and
    combined by `optimize_action` (planners.py:549-561) -- by the reference, by its twins, and here."""
    eng.set_engine(_lib.ENGINES[engine])
    g = golden.census
    p, m = 'census/' + case + '/', 'mppi/' + case + '/'
    eng.set_goal(syn.goal_field(eng.obs_goal), g[p + 'goal_coor'])
    acts = g[m + 'act_seqs']
    _, rew, flips, _ = C.device_rows(eng, g, p, acts, g[m + 'row_hash'])
    r_dev, r_ref, r_tw = rew[:, -1], g[m + 'reward'], g[m + 'twin_reward']
    meets = (g[m + 'min_margin'] < C.TAU) | (g[m + 'min_mask_margin'] < C.TAU_MASK)
    first = C.first_true(np.repeat(meets[:, None], acts.shape[1], 1))       # rows that meet a near-tie at all
    crossed = first < acts.shape[1]
    dev_rows, tw_rows = (flips > 0).any(1), g[m + 'twin_flip_steps'].any(2)
    # (b) on 1 024 rows: who flips, and by how much the final reward moves
    assert not (dev_rows & ~crossed).any(), 'a row that never comes near a tie has a list that differs'
    # how many: the twins' count times K -- or, where the twins' is a handful (one ulp at the start reaches few near-ties; the
    # device's per-step rounding, 3e-5 of a displacement, reaches more), a tenth of the rows that meet a near-tie (observed: <= 6 %)
    assert dev_rows.sum() <= max(K * tw_rows.sum(1).max() + 2, 0.1 * crossed.sum()), (dev_rows.sum(), tw_rows.sum(1), crossed.sum())
    d_dev, d_tw = np.abs(r_dev - r_ref), np.abs(r_tw - r_ref[None])
    q = [0.5, 0.9, 0.99]
    q_dev, q_tw = np.quantile(d_dev, q), np.quantile(d_tw, q, axis=1).max(1)
    floor = 1e-5 * np.abs(r_ref).mean()                                     # the reward's own rtol (tests/test_gpu_parity.py)
    assert (q_dev <= K * np.maximum(q_tw, floor)).all(), (q_dev, q_tw)
    # (c) the planner's quantities
    a64 = acts.astype(np.float64)
    upd = (softmax(0.1 * r_dev.astype(np.float64))[:, None, None] * a64).sum(0)
    d_upd = np.abs(upd - g[m + 'update']).max()
    tw_upd = np.abs(g[m + 'twin_update'] - g[m + 'update'][None]).max()
    assert d_upd <= K * max(tw_upd, 1e-7), (d_upd, tw_upd)
    assert int(r_dev.argmax()) == int(r_ref.argmax())
    # the same iteration through the product's own update kernel (drp_mpc_update_device)
    lo, hi = syn.action_limits()
    eng.mpc_begin(g[p + 's_cur'], g[p + 'attr'], g[p + 'dens'], a64[0], n_sample=acts.shape[0], sigma=0.6, beta_filter=0.7,
                  reward_weight=0.1, act_lo=lo, act_hi=hi)
    eng.mpc_set_actions(acts)
    eng.mpc_rollout()
    eng.mpc_update_device()
    nominal = eng.mpc_get(nominal=True)['nominal']
    assert np.abs(nominal - g[m + 'update']).max() <= K * max(tw_upd, 1e-7)
    assert eng.mpc_stats()['argmax'] == int(r_ref.argmax())
    print('\n[census mppi %s %s] smallest margin of the rows whose lists differ: device %s' % (case, engine, fmt(np.sort(g[m + 'min_margin'][dev_rows])[-6:])))
    print('[census mppi %s %s] %d rows, %d meet a near-tie; rows with a flipped list: device %d, twins %d / %d; |d final reward| '
          '50 / 90 / 99 %% / max: device %s  twins %s; |d update| device %.2e (kernel %.2e), twins %.2e; arg-max row %d = the reference\'s; '
          'rewards span %.1f' % (case, engine, acts.shape[0], crossed.sum(), dev_rows.sum(), tw_rows[0].sum(), tw_rows[1].sum(),
                                 fmt(np.append(q_dev, d_dev.max())), fmt(np.append(q_tw, d_tw.max())), d_upd,
                                 np.abs(nominal - g[m + 'update']).max(), tw_upd, r_dev.argmax(), r_ref.max() - r_ref.min()))
    eng.set_engine(_lib.ENGINE_FUSED)


@pytest.mark.parametrize('case', ['n20', 'n50', 'n100'])
def test_census_gd_planner_at_a_ten_step_horizon(golden, case, exact_goal_transform):
    """Does the planner's CHOICE survive free-running rollouts?  The reference's live planner (`mpc_type 'GD'`,
    planners.py:661-871) at n_look_ahead = 10: five Adam iterations whose gradients run back through ten free-running steps, the
    per-column best (:721-727) and the final vote (:773-781), on 6 trajectories x 3 batch columns of the trained network -- as the
    reference ran it, as its two one-ulp twins ran it, and here.  Held to K x the twins' own spread (floors: the horizon-1
    tolerances of tests/test_gpu_trained.py)."""
    import torch
    from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel
    from dyn_res_pile_manip_amd.planners import PlannerGD
    g = golden.census
    p = 'gdplan/' + case + '/'
    config = syn.default_config()
    config['mpc']['mpc_type'] = 'GD'
    model = PropNetDiffDenModel(config, True)
    w = golden.weights_trained
    model.load_state_dict({k[2:]: torch.from_numpy(np.asarray(w[k])) for k in w.files if k.startswith('w/')}, strict=False)
    planner = PlannerGD(config, syn.SyntheticEnv(config))
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    lo, hi = syn.action_limits()
    act_seq = g[p + 'act_seq']
    H, traj = act_seq.shape[:2]
    res = planner.trajectory_optimization_ptcl_multi_traj(
        g[p + 's_cur'], g[p + 'dens'], g[p + 'attr'], obs_goal, model, act_seq.copy(), np.zeros(H), n_sample=traj, n_look_ahead=H,
        n_update_iter=int(g[p + 'n_update_iter']), action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=1e9)
    model.engine.close()

    def spread(key):
        ref = np.asarray(g[p + 'out/' + key], np.float64)
        dev = np.abs(np.asarray(res[key], np.float64).reshape(ref.shape) - ref).max()
        tw = max(np.abs(np.asarray(g[p + 'twin%d/' % q + key], np.float64) - ref).max() for q in (1, 2))
        return dev, tw, np.abs(ref).max()
    line = []
    for key, floor_abs, floor_rel in (('action_sequence', 2e-3, 0.0), ('action_full', 2e-3, 0.0), ('reward', 0.0, 1e-4),
                                      ('reward_full', 0.0, 1e-4), ('rew_mean', 0.0, 1e-4), ('next_r', 0.0, 1e-4),
                                      ('observation_sequence', 5e-6, 0.0)):
        dev, tw, scale = spread(key)
        line.append('%s %.1e (twins %.1e)' % (key, dev, tw))
        assert dev <= K * max(tw, floor_abs, floor_rel * scale), (key, dev, tw)
    assert int(res['iter_num']) == int(g[p + 'out/iter_num'])
    # the choice itself: the voted trajectory's pushes are the reference's (which row won shows in action_sequence: two
    # candidates differ by whole push lengths, the bound above is 8e-3 of a workspace of +-5)
    print('\n[census gd planner %s, horizon %d] %s' % (case, H, '; '.join(line)))
