"""GPU: the reference's Python call surface (model/gnn_dyn.py, planners.py,
env/flex_rewards.py) served by the HIP engine -- written like the reference's own
call sites (visualize_mpc.py:36-84, env/flex_env.py:1048-1065)."""
import numpy as np
import pytest
import torch

from dyn_res_pile_manip_amd import flex_rewards, synthetic as syn, weights
from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel
from dyn_res_pile_manip_amd.planners import PlannerGD

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures('exact_goal_transform')]


@pytest.fixture(scope='module')
def stack(golden):
    config = syn.default_config()
    env = syn.SyntheticEnv(config)
    model = PropNetDiffDenModel(config, True)
    sd = {k[2:]: torch.from_numpy(golden.weights_seed0[k]) for k in golden.weights_seed0.files
          if k.startswith('w/')}
    model.load_state_dict(sd, strict=False)
    model.cuda().eval()
    planner = PlannerGD(config, env)
    yield config, env, model, planner
    model.engine.close()


def test_predict_one_step_torch_in_torch_out(stack, golden):
    _, _, model, _ = stack
    g = golden.one_step
    args = [torch.from_numpy(g['n64/' + k]) for k in ('attr', 's_cur', 's_delta', 'dens')]
    out = model.predict_one_step(*args)
    assert isinstance(out, torch.Tensor) and out.dtype == torch.float32
    assert np.abs(out.numpy() - g['n64/s_pred']).max() < 2e-6
    with pytest.raises(AssertionError):
        model.predict_one_step(args[0][:, :10], args[1], args[2], args[3])


def test_forward_with_dense_onehot_relations(stack, golden):
    _, _, model, _ = stack
    g = golden.one_step
    idx, cnt = g['n8/nbr_idx'], g['n8/nbr_cnt']
    B, N = cnt.shape
    E = int(cnt.reshape(B, -1).sum(1).max())
    Rr = np.zeros((B, E, N), np.float32)
    Rs = np.zeros((B, E, N), np.float32)
    for b in range(B):
        e = 0
        for i in range(N):
            for k in range(cnt[b, i]):
                Rr[b, e, i] = 1
                Rs[b, e, idx[b, i, k]] = 1
                e += 1
    out = model.model.forward(g['n8/attr'], g['n8/s_cur'], g['n8/s_delta'], Rr, Rs, g['n8/dens'])
    assert np.abs(out - g['n8/s_pred']).max() < 2e-6


def test_particle_nums_masks_padded_particles(stack, golden):
    _, _, model, _ = stack
    g = golden.one_step
    a, s, sd, d = (g['n64/' + k] for k in ('attr', 's_cur', 's_delta', 'dens'))
    nums = np.array([64, 40, 64, 10])
    out = model.predict_one_step(a, s, sd, d, particle_nums=nums)
    from oracle import propnet_dense as od
    W = od.load_weights(golden.weights_seed0)
    adj, _ = od.adjacency(torch.from_numpy(s), torch.from_numpy(sd), 0.08)
    for b in range(4):                                   # model/gnn_dyn.py:238-241
        adj[b, nums[b]:, :] = 0
        adj[b, :, nums[b]:] = 0
    Rr, Rs = od.onehot_relations(adj)
    ref = od.forward_dense(W, torch.from_numpy(a), torch.from_numpy(s), torch.from_numpy(sd), Rr, Rs,
                           torch.from_numpy(d)).numpy()
    assert np.abs(out - ref).max() < 2e-6


def test_rollout_and_evaluate_like_the_planner_does(stack, golden):
    _, _, model, planner = stack
    g = golden.rollout
    out = planner.ptcl_model_rollout(torch.from_numpy(g['c1_nb2/s_cur']), torch.from_numpy(g['c1_nb2/dens']),
                                     torch.from_numpy(g['c1_nb2/attr']), model,
                                     torch.from_numpy(g['c1_nb2/act_seqs']))
    sp = out['model_rollout']['state_pred']
    assert sp.shape == (16, 5, 64, 3) and out['rollout_time'] > 0
    assert np.abs(sp.numpy() - g['c1_nb2/state_pred']).max() < 5e-6
    r = golden.reward
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    planner.particle_num = 64
    planner.ptcl_model_rollout(g['c1/s_cur'], g['c1/dens'], g['c1/attr'], model, g['c1/act_seqs'])
    obs = torch.from_numpy(g['c1/state_pred']).reshape(16, 5, 1, 64, 3)
    rs, nr = planner.ptcl_evaluate_traj(obs, torch.from_numpy(obs_goal), torch.from_numpy(r['I/goal_coor']))
    assert rs.shape == (16, 1) and nr.shape == (16, 5, 1)
    np.testing.assert_allclose(rs.numpy(), r['I/eval_reward_seqs'], rtol=2e-5)
    np.testing.assert_allclose(nr.numpy(), r['I/eval_next_r'], rtol=2e-5)
    sd = planner.gen_s_delta(golden.s_delta['s_cur'], golden.s_delta['action'])
    np.testing.assert_allclose(sd, golden.s_delta['s_delta'], atol=3e-7)
    w2c = planner.world2cam(golden.s_delta['world2cam_in'])
    np.testing.assert_allclose(w2c, golden.s_delta['world2cam_out'], atol=1e-7)


def test_trajectory_optimization_contract(stack, golden):
    """Same call as env/flex_env.py:1048-1065; same dict as planners.py:858-871."""
    _, env, model, planner = stack
    g = golden.gd_planner
    s, dens, attr, act_seq = g['s_cur'], g['dens'], g['attr'], g['act_seq']
    nb, N, _ = s.shape
    traj = act_seq.shape[1]
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    lo, hi = syn.action_limits()
    np.random.seed(0)
    res = planner.trajectory_optimization_ptcl_multi_traj(
        s, dens, attr, obs_goal, model, act_seq, np.zeros(1), n_sample=64, n_look_ahead=1,
        n_update_iter=6, action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=1e9)
    for k in ('action_sequence', 'action_full', 'reward_full', 'observation_sequence', 'reward',
              'next_r', 'rew_mean', 'rew_std', 'times', 'iter_num'):
        assert k in res, k
    assert res['action_sequence'].shape == g['out/action_sequence'].shape == (1, 4)
    assert res['observation_sequence'].shape == g['out/observation_sequence'].shape
    assert res['reward'].shape == g['out/reward'].shape
    assert res['next_r'].shape == g['out/next_r'].shape
    assert res['rew_mean'].shape == (1, 6) and res['rew_std'].shape == (1, 6)
    assert res['action_full'].shape == (64 * nb, 4) and res['reward_full'].shape == (64,)
    lo_, hi_ = planner._clip_box()
    assert (res['action_sequence'] >= lo_ - 1e-5).all() and (res['action_sequence'] <= hi_ + 1e-5).all()
    # iteration 0 scores exactly the reference's candidate set: same mean reward as the
    # reference's first GD iteration (before any Adam step), column 0
    np.testing.assert_allclose(res['rew_mean'][0, 0], g['out/rew_mean'][0, 0], rtol=2e-5)
    np.testing.assert_allclose(res['rew_std'][0, 0], g['out/rew_std'][0, 0], rtol=2e-4)
    # the planner must not do worse than the best initial candidate
    cand_best = res['rew_mean'][0, 0]
    assert res['reward'][0] >= cand_best - 1e-3


def test_planner_with_elite_update(stack, golden):
    """mpc_type 'CEM' (an extension: the sampling planner's update is the mean of the n_elite best sequences
    instead of the softmax mean): same call, same dict; not worse than the best initial candidate."""
    config, env, model, planner = stack
    g = golden.gd_planner
    s, dens, attr, act_seq = g['s_cur'], g['dens'], g['attr'], g['act_seq']
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    lo, hi = syn.action_limits()
    old = dict(config['mpc'])
    config['mpc']['mpc_type'] = 'CEM'
    config['mpc']['cem'] = {'n_elite': 8}
    try:
        np.random.seed(0)
        res = planner.trajectory_optimization_ptcl_multi_traj(
            s, dens, attr, obs_goal, model, act_seq, np.zeros(1), n_sample=64, n_look_ahead=1,
            n_update_iter=6, action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=1e9)
    finally:
        config['mpc'].clear()
        config['mpc'].update(old)
    assert res['action_sequence'].shape == (1, 4) and res['rew_mean'].shape == (1, 6)
    assert np.isfinite(res['reward']).all()
    assert res['reward'][0] >= res['rew_mean'][0, 0] - 1e-3


def test_device_fps_equals_fps_np(stack):
    """utils.py:451-466 as used at planners.py:620-624: same points in the same order."""
    from oracle.particles import fps_np
    _, _, model, _ = stack
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    rc = np.argwhere(obs_goal < 0.5)
    cr = rc[:, ::-1].astype(np.float32)
    for k in (7, 320, 1500):
        want, want_md = fps_np(cr, k, 0)
        got, got_md, idx = model.engine.fps(cr, k, 0)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_allclose(got_md, want_md, rtol=1e-6)
        assert len(set(idx.tolist())) == k
    pts3 = np.random.default_rng(0).uniform(-0.2, 0.2, (5000, 3)).astype(np.float32)
    want, _ = fps_np(pts3, 100, 17)
    got, _, _ = model.engine.fps(pts3, 100, 17)
    np.testing.assert_array_equal(got, want)
    # The register kernel keeps SQUARED distances and takes numpy's first maximum of their square roots in two steps (max,
    # then the smallest index whose root rounds to the same float: csrc/k_fps.h).  Point sets made for that: a coarse lattice
    # with duplicates (exact ties by the hundred), and radii one float apart around a centre (different squares, one root).
    rng = np.random.default_rng(5)
    lattice = (rng.integers(0, 40, (6000, 2)) * np.float32(0.37)).astype(np.float32)
    r = np.float32(3.0) * (np.float32(1.0) + np.arange(64, dtype=np.float32) * np.float32(2.0 ** -23))
    ang = rng.uniform(0, 2 * np.pi, 64)
    ring = np.stack([r * np.cos(ang).astype(np.float32), r * np.sin(ang).astype(np.float32)], 1).astype(np.float32)
    near = np.concatenate([np.zeros((1, 2), np.float32), ring, ring[::-1] * np.float32(0.5)])
    for pts, k, init in ((lattice, 1500, 11), (near, 60, 0), (np.concatenate([near, near + np.float32(1e-3)]), 100, 0)):
        want, want_md = fps_np(pts, k, init)
        got, got_md, idx = model.engine.fps(pts, k, init)
        np.testing.assert_array_equal(got, want)
        assert got_md == np.float32(want_md)
    # 6 000 of 21 680 goal pixels (a 1 200-particle plan's subsample, planners.py:621): 27.6 ms in round 4
    big = np.argwhere(syn.goal_distance_image(syn.goal_mask('disc')) < 30.0)[:21680, ::-1].astype(np.float32)
    import time
    model.engine.fps(big, 6000, 0)
    t0 = time.perf_counter()
    got, _, _ = model.engine.fps(big, 6000, 0)
    ms = (time.perf_counter() - t0) * 1e3
    want, _ = fps_np(big, 6000, 0)
    np.testing.assert_array_equal(got, want)
    print('\n[fps] 6 000 of %d points: %.2f ms (upload + kernel + download)' % (big.shape[0], ms))
    assert ms < 16.0                      # measured 11.8 ms (1.96 us per pick on ONE workgroup, the chip otherwise idle); round 4: 27.6


def test_a_named_goal_is_installed_once_and_gives_the_same_plan(stack, golden):
    """goal_key (not in the reference): env/flex_env.py:1048 hands the planner the SAME subgoal image on each of its 20 MPC
    steps; with the caller's name for it the installed field and goal pixels are re-used without copying or hashing the
    2 MB image, a new name installs again, and the plan is the one the content digest gives."""
    config, env, model, _ = stack
    config = dict(config)
    config['mpc'] = dict(config['mpc'], mpc_type='GD')
    g = golden.gd_planner
    goal_i = syn.goal_distance_image(syn.goal_mask('I'))
    goal_d = syn.goal_distance_image(syn.goal_mask('disc'))
    lo, hi = syn.action_limits()

    def call(planner, goal, **kw):
        return planner.trajectory_optimization_ptcl_multi_traj(
            g['s_cur'], g['dens'], g['attr'], goal, model, g['act_seq'], np.zeros(1), n_sample=10, n_look_ahead=1,
            n_update_iter=3, action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=1e9, **kw)
    ref_i, ref_d = call(PlannerGD(config, env), goal_i), call(PlannerGD(config, env), goal_d)
    assert not np.array_equal(ref_i['action_full'], ref_d['action_full'])
    planner = PlannerGD(config, env)
    first = call(planner, goal_i, goal_key='I')
    assert first['times']['goal_cached'] is False
    again = call(planner, goal_i, goal_key='I')
    assert again['times']['goal_cached'] is True and again['times']['goal_time'] < 2e-4
    other = call(planner, goal_d, goal_key='disc')                  # a new name: installed again
    assert other['times']['goal_cached'] is False
    for got, want in ((first, ref_i), (again, ref_i), (other, ref_d)):
        np.testing.assert_array_equal(got['action_full'], want['action_full'])
        np.testing.assert_array_equal(got['reward_full'], want['reward_full'])
    # the name is trusted: the same name with another image keeps the installed goal (the caller's contract)
    stale = call(planner, goal_i, goal_key='disc')
    np.testing.assert_array_equal(stale['action_full'], ref_d['action_full'])


def test_a_model_without_weights_does_not_compute_with_anothers(stack, golden):
    """Models share the process's context (engine.default_engine): one that never loaded a state_dict must fail loudly, not
    answer with whichever other model's weights are resident."""
    config, _, model, _ = stack
    g = golden.one_step
    args = [g['n64/' + k] for k in ('attr', 's_cur', 's_delta', 'dens')]
    model.predict_one_step(*args)                                   # the fixture's model owns the shared context now
    empty = PropNetDiffDenModel(config, True, engine=model.engine)
    with pytest.raises(RuntimeError, match='no weights'):
        empty.predict_one_step(*args)
    np.testing.assert_array_equal(model.predict_one_step(*args), model.predict_one_step(*args))


def test_one_mpc_step_end_to_end(stack):
    """Observation -> particles -> planner -> push, chained as env/flex_env.py:1016-1065 chains them
    (tools/mpc_step_demo.py), every piece on the device."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import mpc_step_demo as demo
    from dyn_res_pile_manip_amd import utils as dev
    config, env, model, _ = stack
    config = dict(config)
    config['mpc'] = dict(config['mpc'], mpc_type='GD')
    planner = PlannerGD(config, env)
    assert dev.get_engine() is model.engine             # utils.py's helpers and the model share the process's context
    obs = syn.render_depth(1500, seed=3, kind='uniform')
    subgoal = syn.goal_distance_image(syn.goal_mask('I'))
    act_seq = np.stack([syn.nominal_pushes(1, seed=20 + i) for i in range(6)], axis=1)
    out, t = demo.mpc_step(obs, subgoal, model, planner, 30, act_seq, syn.demo_cam_params(), 24.0, n_update_iter=4)
    assert out['action_sequence'].shape == (1, 4) and np.isfinite(out['action_sequence']).all()
    assert out['observation_sequence'].shape == (1, 30, 3)
    lo, hi = syn.action_limits()
    assert (out['action_sequence'][0] >= lo - 1e-6).all() and (out['action_sequence'][0] <= hi + 1e-6).all()
    assert out['iter_num'] == 3 and np.isfinite(out['reward']).all()


@pytest.mark.parametrize('mpc_type', ['MPPI', 'CEM', 'GD'])
def test_pipelined_loop_returns_what_the_blocking_loop_returns(stack, golden, mpc_type):
    """The planner enqueues iteration i + 1 before it waits for iteration i's pushes and rewards (drp_mpc_fetch_async /
    drp_gd_step_async); with the opt-in wall-clock break (a limit that never binds here) it runs the blocking loop.
    Same seed, same dict, bit for bit."""
    import copy
    _, env, model, _ = stack
    g = golden.gd_planner
    s, dens, attr, act_seq = g['s_cur'], g['dens'], g['attr'], g['act_seq']
    config = copy.deepcopy(syn.default_config())
    config['mpc']['mpc_type'] = mpc_type
    planner = PlannerGD(config, env)
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    lo, hi = syn.action_limits()
    kw = dict(n_sample=act_seq.shape[1] if mpc_type == 'GD' else 96, n_look_ahead=1, n_update_iter=5,
              action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=1e9, seed=77)
    res = [planner.trajectory_optimization_ptcl_multi_traj(s, dens, attr, obs_goal, model, act_seq, np.zeros(1),
                                                           wallclock_limit=w, **kw) for w in (False, True)]
    assert res[0]['iter_num'] == res[1]['iter_num']
    for k in ('action_sequence', 'action_full', 'reward_full', 'observation_sequence', 'reward', 'next_r', 'rew_mean', 'rew_std'):
        np.testing.assert_array_equal(res[0][k], res[1][k], err_msg=k)


def test_config_reward_ptcl_with_the_references_own_arguments(stack, golden):
    """env/flex_env.py:1032-1036,1102 calls `config_reward_ptcl(state, goal, cam_params=..., goal_coor=..., normalize=True)`:
    no engine argument.  The mirror runs it on the process's one context -- the model's, the planner's and utils.py's --
    and installs `cam_params` on every call."""
    _, env, model, planner = stack
    from dyn_res_pile_manip_amd import flex_rewards, utils
    from dyn_res_pile_manip_amd.engine import default_engine
    assert utils.get_engine() is model.engine is default_engine()          # one context per process, not two
    g = golden.reward
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    state, goal_coor = torch.from_numpy(g['I/state']), torch.from_numpy(g['I/goal_coor'])
    r = flex_rewards.config_reward_ptcl(state, torch.from_numpy(obs_goal), cam_params=env.get_cam_params(),
                                        goal_coor=goal_coor, normalize=True)
    assert isinstance(r, torch.Tensor)
    np.testing.assert_allclose(r.numpy(), g['I/reward'], rtol=1e-5)
    # other intrinsics give another reward, and the reference's give the reference's again: cam_params is applied, every call
    fx, fy, cx, cy = env.get_cam_params()
    r2 = flex_rewards.config_reward_ptcl(state, obs_goal, cam_params=[0.9 * fx, 0.9 * fy, cx, cy], goal_coor=goal_coor, normalize=True)
    assert np.abs(r2.numpy() - g['I/reward']).max() > 1e-3
    r3 = flex_rewards.config_reward_ptcl(state, obs_goal, env.get_cam_params(), goal_coor)
    np.testing.assert_allclose(r3.numpy(), g['I/reward'], rtol=1e-5)
    # the planner still finds its own camera after that (it installs it whenever it binds a model)
    out = planner.ptcl_model_rollout(golden.rollout['c1/s_cur'], golden.rollout['c1/dens'], golden.rollout['c1/attr'], model,
                                     golden.rollout['c1/act_seqs'])
    assert np.abs(out['model_rollout']['state_pred'] - golden.rollout['c1/state_pred']).max() < 5e-6


def _scaled_relation_encoder(golden, f2, f4):
    """seed-0 weights with the relation encoder's second layer (weight and bias) times f2 and its third layer's weight times
    f4: for f2 * f4 = 1 the same function (ReLU is positively homogeneous) up to fp32 rounding."""
    sd = {k[2:]: np.array(golden.weights_seed0[k]) for k in golden.weights_seed0.files if k.startswith('w/')}
    sd['model.relation_encoder.model.2.weight'] = sd['model.relation_encoder.model.2.weight'] * np.float32(f2)
    sd['model.relation_encoder.model.2.bias'] = sd['model.relation_encoder.model.2.bias'] * np.float32(f2)
    sd['model.relation_encoder.model.4.weight'] = sd['model.relation_encoder.model.4.weight'] * np.float32(f4)
    return sd


def test_weights_the_fused_engine_cannot_serve_fall_back_by_themselves(golden):
    """A checkpoint with a matrix entry beyond fp16 (the split-fp16 relation encoder packs W2, W3, W_e as they are): the
    fused engine refuses every call with DRP_ERANGE.  The host mirror -- model and planner with NO engine argument, as
    visualize_mpc.py:36-41,70-84 build them -- switches to the fp32 matrix engine by itself, warns once and keeps working;
    the results are the oracle's for those weights.  (Weights merely far from a fresh network's, x 50 in every layer of
    the relation encoder, stay on the fused engine: its range shift covers them.)  The gradient-descent planner -- the only
    one env/flex_env.py:973-976 accepts -- and the trainer do not stop either: their tape is written on the fp32 matrix
    engine, gradients and Adam iterates against the oracle's autograd."""
    from oracle import propnet_sparse as osp
    from dyn_res_pile_manip_amd import _lib
    from dyn_res_pile_manip_amd.engine import set_default_engine
    set_default_engine(None)                                   # a fresh process-wide context for this test
    config = syn.default_config()
    env = syn.SyntheticEnv(config)
    g = golden.one_step
    a, s, sdl, d = (g['n64/' + k] for k in ('attr', 's_cur', 's_delta', 'dens'))
    try:
        # (1) every layer of the relation encoder x 50 (edge effects 125 000 times a fresh network's): inside the fused
        # engine's reach -- its range shift follows the weights --, no warning, the oracle's answer
        sd50 = {k[2:]: np.array(golden.weights_seed0[k]) for k in golden.weights_seed0.files if k.startswith('w/')}
        for lyr in ('0', '2', '4'):
            sd50['model.relation_encoder.model.%s.weight' % lyr] *= np.float32(50.0)
            sd50['model.relation_encoder.model.%s.bias' % lyr] *= np.float32(50.0)
        model = PropNetDiffDenModel(config, True)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd50.items()}, strict=False)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('error')
            out = model.predict_one_step(a, s, sdl, d)
        assert model.engine.engine_id == _lib.ENGINE_FUSED
        ref = osp.predict_one_step(osp.weights_np(sd50), a, s, sdl, d)
        assert np.abs(out - ref).max() < 1e-4 * np.abs(ref - s).max()
        # (2) an entry beyond fp16: DRP_ERANGE -> the fp32 matrix engine, one warning, the oracle's answer
        sdx = _scaled_relation_encoder(golden, 1e6, 1e-6)
        assert np.abs(sdx['model.relation_encoder.model.2.weight']).max() > 65504.0
        model2 = PropNetDiffDenModel(config, True)
        model2.load_state_dict({k: torch.from_numpy(v) for k, v in sdx.items()}, strict=False)
        with pytest.warns(RuntimeWarning, match='fp32 matrix engine'):
            out2 = model2.predict_one_step(torch.from_numpy(a), torch.from_numpy(s), torch.from_numpy(sdl), torch.from_numpy(d))
        assert model2.engine.engine_id == _lib.ENGINE_MFMA
        ref2 = osp.predict_one_step(osp.weights_np(sdx), a, s, sdl, d)
        assert np.abs(out2.numpy() - ref2).max() < 2e-6
        # the sampling planner on that model, no engine anywhere in sight: rollouts and a whole planner call
        planner = PlannerGD(config, env)
        ro = golden.rollout
        sp = planner.ptcl_model_rollout(ro['c1/s_cur'], ro['c1/dens'], ro['c1/attr'], model2, ro['c1/act_seqs'])['model_rollout']['state_pred']
        refr = osp.rollout(osp.weights_np(sdx), ro['c1/s_cur'], ro['c1/dens'], ro['c1/attr'], ro['c1/act_seqs'],
                           osp.world2cam_affine(syn.demo_cam_extrinsics(), 24), 24.0)
        assert np.abs(sp - refr).max() < 5e-6
        gp = golden.gd_planner
        lo, hi = syn.action_limits()
        np.random.seed(0)
        res = planner.trajectory_optimization_ptcl_multi_traj(
            gp['s_cur'], gp['dens'], gp['attr'], syn.goal_distance_image(syn.goal_mask('I')), model2, gp['act_seq'], np.zeros(1),
            n_sample=32, n_look_ahead=1, n_update_iter=3, action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=1e9)
        assert np.isfinite(res['reward']).all() and res['action_sequence'].shape == (1, 4)
        # the model of part (1) takes the shared context back with ITS weights -- and gets the fused engine back with them
        # (the refusal belonged to the other weights)
        out_again = model.predict_one_step(a, s, sdl, d)
        assert np.abs(out_again - ref).max() < 1e-4 * np.abs(ref - s).max()
        assert model.engine.engine_id == _lib.ENGINE_FUSED
        # (3) the reference's LIVE planner is the gradient-descent one (env/flex_env.py:973-976 accepts nothing else): on such
        # weights it must not stop -- its forward pass writes the tape on the fp32 matrix engine (k_aggregate_tape) instead.
        # Gradients, three Adam iterations and the returned pushes against the oracle's autograd on those weights.
        from oracle import propnet_dense as od
        Wt = od.load_weights({'w/' + k: v for k, v in sdx.items()})
        obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
        traj, nb = gp['act_seq'].shape[1], gp['s_cur'].shape[0]
        cfg_gd = syn.default_config()
        cfg_gd['mpc']['mpc_type'] = 'GD'
        planner_gd = PlannerGD(cfg_gd, env)
        eng = model2.engine
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)              # the winner's re-rollout falls back with its warning
            eng.dispatch_reset()
            res = planner_gd.trajectory_optimization_ptcl_multi_traj(
                gp['s_cur'], gp['dens'], gp['attr'], obs_goal, model2, gp['act_seq'], np.zeros(1),
                n_sample=traj, n_look_ahead=1, n_update_iter=3, action_lower_lim=lo, action_upper_lim=hi,
                use_gpu=True, time_lim=1e9)
        assert 'k_aggregate_tape' in eng.last_dispatch()
        G, goal_coor = eng.set_goal_image(obs_goal, 5 * gp['s_cur'].shape[1], fps_init=0, mode=flex_rewards.DIST_TRANSFORM, want=True)
        acts = torch.tensor(np.repeat(gp['act_seq'].transpose(1, 0, 2), nb, axis=0).astype(np.float32), requires_grad=True)
        opt = torch.optim.Adam([acts], lr=cfg_gd['mpc']['gd']['lr'], betas=(0.9, 0.999))
        lo_t, hi_t = torch.tensor(lo, dtype=torch.float32), torch.tensor(hi, dtype=torch.float32)
        means = []
        for it in range(3):
            r_it, g_it, _ = od.gd_loss_and_grads(Wt, gp['s_cur'], gp['dens'], gp['attr'], acts.detach().numpy(), G,
                                                 syn.demo_cam_params(), goal_coor, syn.demo_cam_extrinsics(), 24)
            means.append(np.asarray(r_it).reshape(traj, nb)[:, 0].mean())
            opt.zero_grad()
            acts.grad = torch.from_numpy(np.ascontiguousarray(g_it))
            opt.step()
            with torch.no_grad():
                acts.copy_(torch.minimum(torch.maximum(acts, lo_t), hi_t))
        np.testing.assert_allclose(res['rew_mean'][0, :3], means, rtol=1e-4)
        np.testing.assert_allclose(res['action_full'], acts.detach().numpy()[:, 0, :], atol=2e-3)
        # ... and the trainer (train/train_gnn_dyn.py:159-210 through run_batch): loss and gradients of those weights
        from dyn_res_pile_manip_amd import train_gnn_dyn as T
        batch = syn.push_batch(3, 4, 2, sizes=(10, 20, 30))
        opt_dev = T.DeviceAdam(model2, 1e-4, betas=(0.9, 0.999), n_rollout=2)
        eng.dispatch_reset()
        loss_eval = T.run_batch(model2, opt_dev, batch + (None,), 'valid', 2)
        loss_g, grad = eng.train_step(*batch, mode='grad', want_grad=True)
        assert 'k_aggregate_tape' in eng.last_dispatch()
        ref_loss, ref_grads = od.train_loss_and_grads(sdx, *batch)
        assert abs(loss_eval - ref_loss) < 2e-4 * abs(ref_loss) and abs(loss_g - ref_loss) < 2e-4 * abs(ref_loss)
        got = weights.state_dict_from_blob(grad)
        for k, _ in weights.STATE_DICT_KEYS:
            scale = max(np.abs(ref_grads[k]).max(), 1e-12)
            assert np.abs(np.asarray(got[k]).reshape(ref_grads[k].shape) - ref_grads[k]).max() < 2e-3 * scale, k
        loss_tr = T.run_batch(model2, opt_dev, batch + (None,), 'train', 2)          # one Adam step on the device
        assert abs(loss_tr - ref_loss) < 2e-4 * abs(ref_loss)
        assert np.isfinite(model2.engine.get_weights()).all()
    finally:
        from dyn_res_pile_manip_amd.engine import default_engine
        default_engine().close()
        set_default_engine(None)
