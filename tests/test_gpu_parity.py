"""GPU parity: the HIP path, through the C ABI, against the golden vectors captured
from the reference and against the oracle on seeded inputs.  Tolerances: fp32, 1e-4
relative to the step displacement (BASELINE.json north_star; SURVEY.md 7, hard part 2);
neighbour lists bit-exact."""
import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn, weights
from oracle import propnet_sparse as osp

pytestmark = pytest.mark.gpu

ONE_STEP = ['n64', 'n50', 'n150', 'n300', 'n600', 'n8', 'blob150', 'n1200']
ENGINES = ['valu', 'mfma', 'split', 'fused']


def disp_rel(out, ref, s_cur):
    return float(np.abs(out - ref).max() / max(np.abs(ref - s_cur).max(), 1e-12))


@pytest.fixture(scope='module')
def ctx(golden):
    from dyn_res_pile_manip_amd.engine import Engine
    eng = Engine(0)
    blob = weights.blob_from_state_dict(golden.weights_seed0)
    eng.load_weights(blob, 0.08)
    M34 = osp.world2cam_affine(syn.demo_cam_extrinsics(), 24)
    eng.set_camera(M34, 24.0, syn.demo_cam_params())
    eng.W = osp.weights_np(golden.weights_seed0)
    eng.M34 = M34
    yield eng
    eng.close()


def set_engine(ctx, name):
    from dyn_res_pile_manip_amd import _lib
    ctx.set_engine(_lib.ENGINES[name])


def test_gen_s_delta(ctx, golden):
    g = golden.s_delta
    out = ctx.gen_s_delta(g['s_cur'], g['action'])
    np.testing.assert_allclose(out, g['s_delta'], rtol=0, atol=3e-7)
    # masks are decided identically: same support
    np.testing.assert_array_equal(out != 0, g['s_delta'] != 0)


@pytest.mark.parametrize('case', ONE_STEP)
def test_graph_bit_exact(ctx, golden, case):
    g = golden.one_step
    idx, cnt = ctx.build_graph(g[case + '/s_cur'], g[case + '/s_delta'])
    np.testing.assert_array_equal(cnt, g[case + '/nbr_cnt'])
    np.testing.assert_array_equal(idx, g[case + '/nbr_idx'])


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('case', ONE_STEP)
def test_one_step(ctx, golden, case, engine):
    set_engine(ctx, engine)
    g = golden.one_step
    a, s, sd, d = g[case + '/attr'], g[case + '/s_cur'], g[case + '/s_delta'], g[case + '/dens']
    out = ctx.step(a, s, sd, d)
    ref = g[case + '/s_pred']
    assert disp_rel(out, ref, s) < 1e-4
    assert np.abs(out - ref).max() < 2e-6
    # stage-by-stage against the sparse oracle (same formulation as the kernels)
    B, N = a.shape
    taps = {}
    osp.predict_one_step(ctx.W, a, s, sd, d, taps=taps)
    eff = ctx.debug_fetch('effect', (B, N, 64))
    np.testing.assert_allclose(eff, taps['particle_effect_2'], rtol=0, atol=1e-5)
    c_node = ctx.debug_fetch('c_node', (B, N, 64))
    np.testing.assert_allclose(c_node, taps['c_node'], rtol=0, atol=5e-6)
    if engine != 'fused':          # the fused engine never materialises c_edge
        c_edge = ctx.debug_fetch('c_edge', (B, N, 10, 64))
        valid = np.arange(10)[None, None, :] < taps['nbr_cnt'][:, :, None]
        np.testing.assert_allclose(c_edge[valid], taps['c_edge'][valid], rtol=0, atol=5e-6)
        agg = ctx.debug_fetch('agg', (B, N, 64))     # (in the fused engine agg stays in registers)
        np.testing.assert_allclose(agg, taps['effect_rel_2'].sum(2), rtol=0, atol=2e-5)


@pytest.mark.parametrize('engine', ENGINES)
def test_forward_with_given_relations(ctx, golden, engine):
    set_engine(ctx, engine)
    g = golden.one_step
    case = 'n64'
    out = ctx.forward(g[case + '/attr'], g[case + '/s_cur'], g[case + '/s_delta'], g[case + '/dens'],
                      g[case + '/nbr_idx'], g[case + '/nbr_cnt'])
    assert disp_rel(out, g[case + '/s_pred'], g[case + '/s_cur']) < 1e-4
    # drop every edge: the model must still run (isolated particles)
    idx = -np.ones_like(g[case + '/nbr_idx'])
    cnt = np.zeros_like(g[case + '/nbr_cnt'])
    out0 = ctx.forward(g[case + '/attr'], g[case + '/s_cur'], g[case + '/s_delta'], g[case + '/dens'], idx, cnt)
    ref0 = osp.forward_sparse(ctx.W, g[case + '/attr'], g[case + '/s_cur'], g[case + '/s_delta'],
                              g[case + '/dens'], idx.astype(np.int32), cnt.astype(np.int32))
    assert np.abs(out0 - ref0).max() < 2e-6


def check_rollout(ctx, W, s0, attr, dens, acts, ref, states, tol=1e-4):
    """Free-running rollout against the reference's: at every step, first the EDGE SETS the two
    trajectories induce must be equal (hard part 1: neighbour lists from the device's own
    previous state vs from the reference's, both through the pinned oracle); while they are, the step must
    stay within a FLAT `tol` of its displacement -- the bound does not grow with the step index."""
    B, H, N, _ = ref.shape
    nb = s0.shape[0]
    prev_ref = np.tile(s0, (B // nb, 1, 1))
    prev_dev = prev_ref
    for t in range(H):
        idx_r, cnt_r = osp.build_neighbours(prev_ref, osp.gen_s_delta(prev_ref, acts[:, t], ctx.M34, 24.0))
        idx_d, cnt_d = osp.build_neighbours(prev_dev, osp.gen_s_delta(prev_dev, acts[:, t], ctx.M34, 24.0))
        np.testing.assert_array_equal(cnt_d, cnt_r, err_msg='in-degrees differ at step %d' % t)
        np.testing.assert_array_equal(idx_d, idx_r, err_msg='edge sets differ at step %d' % t)
        assert disp_rel(states[:, t], ref[:, t], prev_ref) < tol, t
        prev_ref, prev_dev = ref[:, t], states[:, t]


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('case', ['c1', 'c1_nb2', 'n150', 'n300', 'n50', 'n600', 'n1200'])
def test_rollout_vs_reference(ctx, golden, case, engine):
    set_engine(ctx, engine)
    g = golden.rollout
    ref = g[case + '/state_pred']
    states, _ = ctx.rollout(g[case + '/s_cur'], g[case + '/attr'], g[case + '/dens'], g[case + '/act_seqs'])
    check_rollout(ctx, ctx.W, g[case + '/s_cur'], g[case + '/attr'], g[case + '/dens'], g[case + '/act_seqs'], ref, states)
    assert np.abs(states - ref).max() < 5e-6


@pytest.mark.parametrize('goal', ['I', 'disc'])
def test_reward(ctx, golden, goal, exact_goal_transform):
    g = golden.reward
    obs_goal = syn.goal_distance_image(syn.goal_mask(goal))
    ctx.set_goal(syn.goal_field(obs_goal), g[goal + '/goal_coor'])
    r = ctx.reward(g[goal + '/state'], normalize=True)
    np.testing.assert_allclose(r, g[goal + '/reward'], rtol=1e-5)
    r_un = ctx.reward(g[goal + '/state'], normalize=False)
    np.testing.assert_allclose(r_un, g[goal + '/reward_unnorm'], rtol=1e-5)
    # the reference-named entry point (env/flex_rewards.py:156): goal IMAGE in, the field built on the device by the
    # transform the fixture was captured with (exact Euclidean: make_golden.py's cv2 stub), torch tensors in and out
    import torch
    from dyn_res_pile_manip_amd import flex_rewards
    flex_rewards.DIST_TRANSFORM = 'exact'           # the exact_goal_transform fixture restores it
    r_t = flex_rewards.config_reward_ptcl(torch.from_numpy(g[goal + '/state']), torch.from_numpy(obs_goal), syn.demo_cam_params(),
                                          torch.from_numpy(g[goal + '/goal_coor']), normalize=True, offset=(0, 0), engine=ctx)
    assert isinstance(r_t, torch.Tensor) and r_t.shape == (g[goal + '/state'].shape[0],)
    np.testing.assert_allclose(r_t.numpy(), g[goal + '/reward'], rtol=1e-5)
    r_u = flex_rewards.config_reward_ptcl(g[goal + '/state'], obs_goal, syn.demo_cam_params(), g[goal + '/goal_coor'],
                                          normalize=False, engine=ctx)
    np.testing.assert_allclose(r_u, g[goal + '/reward_unnorm'], rtol=1e-5)
    with pytest.raises(NotImplementedError):
        flex_rewards.config_reward_ptcl(g[goal + '/state'], obs_goal, syn.demo_cam_params(), g[goal + '/goal_coor'],
                                        offset=(3, 0), engine=ctx)
    # rollout + reward of every step == ptcl_evaluate_traj's next_r
    ctx.set_goal(syn.goal_field(obs_goal), g[goal + '/goal_coor'])
    ro = golden.rollout
    _, rew = ctx.rollout(ro['c1/s_cur'], ro['c1/attr'], ro['c1/dens'], ro['c1/act_seqs'],
                         want_states=False, want_reward=True)
    np.testing.assert_allclose(rew, g[goal + '/eval_next_r'][:, :, 0], rtol=2e-5)


def test_mppi_update_matches_reference(ctx, golden):
    g = golden.mppi
    acts = g['opt_act_seqs'][:, :, 0, :]            # [64,5,4]
    rew = g['opt_reward'][:, 0]
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    N = 16
    ctx.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
    s0, dens, attr = syn.make_pile(N, 1, seed=0)
    lo, hi = syn.action_limits()
    ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=64, sigma=0.6, beta_filter=0.7,
                  reward_weight=0.1, act_lo=lo, act_hi=hi)
    # the update is linear in the actions and depends on rewards only through softmax:
    # feed the reference's action set, roll out, then compare with optimize_action applied
    # to the rewards the device produced
    ctx.mpc_set_actions(acts.astype(np.float32))
    ctx.mpc_rollout()
    part = ctx.mpc_partials()
    got = ctx.mpc_get(rewards=True, actions=True)
    from oracle import propnet_dense as od
    expect = od.optimize_action(got['actions'].astype(np.float64), got['rewards'], 0.1)
    nominal = ctx.mpc_update(part)
    np.testing.assert_allclose(nominal, expect, rtol=1e-9, atol=1e-9)
    # two-shard combine == single shard
    m, Z, A = osp.mppi_partials(0.1, got['rewards'], got['actions'])
    np.testing.assert_allclose(part[0], m, rtol=1e-12)
    np.testing.assert_allclose(part[1], Z, rtol=1e-10)
    np.testing.assert_allclose(part[2:2 + 20].reshape(5, 4), A, rtol=1e-10)
    st = ctx.mpc_stats()
    np.testing.assert_allclose(st['mean'], got['rewards'].astype(np.float64).mean(), rtol=1e-10)
    np.testing.assert_allclose(st['std'], got['rewards'].astype(np.float64).std(ddof=1), rtol=1e-7)
    assert st['argmax'] == int(np.argmax(got['rewards']))
    # same path fully on the device
    ctx.mpc_set_actions(acts.astype(np.float32))
    ctx.mpc_rollout()
    ctx.mpc_update_device()
    np.testing.assert_allclose(ctx.mpc_get(nominal=True)['nominal'], expect, rtol=1e-9, atol=1e-9)


def test_mppi_sampler(ctx, golden):
    g = golden.mppi
    from oracle import propnet_dense as od
    N = 16
    s0, dens, attr = syn.make_pile(N, 2, seed=0)
    lo, hi = syn.action_limits()
    ns, H = 4096, 5
    ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=ns, sigma=0.6, beta_filter=0.7,
                  reward_weight=0.1, act_lo=lo, act_hi=hi, seed=42)
    # host-provided normals: exact filter/clip parity with the reference's sampler
    rng = np.random.default_rng(5)
    z = rng.standard_normal((ns, H, 4)).astype(np.float32)
    ctx.mpc_sample(0, noise=z)
    a = ctx.mpc_get(actions=True)['actions'].reshape(ns, 2, H, 4)
    np.testing.assert_array_equal(a[:, 0], a[:, 1])       # every column of a sample shares the push

    class FakeRng(object):
        def __init__(self):
            self.t = 0

        def normal(self, mu, sigma, shape):
            out = sigma * z[:, self.t].astype(np.float64)
            self.t += 1
            return out
    expect = od.sample_action_sequences(g['nominal'], ns, 0.6, 0.7, lo, hi, FakeRng())
    np.testing.assert_allclose(a[:, 0], expect, rtol=0, atol=1e-6)
    # device Philox normals: distribution parity with the reference's statistics
    ctx.mpc_sample(1)
    a = ctx.mpc_get(actions=True)['actions'].reshape(ns, 2, H, 4)[:, 0].astype(np.float64)
    np.testing.assert_allclose(a.mean(0), g['sample_mean'][:, 0], atol=0.05)
    np.testing.assert_allclose(a.std(0), g['sample_std'][:, 0], rtol=0.08)
    assert (a.min(0) >= lo - 1e-6).all() and (a.max(0) <= hi + 1e-6).all()
    resid = a - g['nominal'][None]
    corr = [np.corrcoef(resid[:, t, 0], resid[:, t + 1, 0])[0, 1] for t in range(H - 1)]
    np.testing.assert_allclose(corr, g['resid_lag1_corr'], atol=0.06)
    ctx.mpc_sample(2)
    b = ctx.mpc_get(actions=True)['actions'].reshape(ns, 2, H, 4)[:, 0]
    assert np.abs(b - a).max() > 0.1                        # a new iteration draws new noise


@pytest.mark.parametrize('noise_type', ['uniform', 'total_rand'])
def test_device_sampler_noise_types(ctx, golden, noise_type):
    """drp_mpc_params.noise_type (planners.py:123-135,169-175): host-fed draws reproduce the host mirror's
    arithmetic exactly; Philox draws match the statistics of the reference's own sampler (mppi_noise.npz)."""
    g = golden.mppi_noise
    N = 16
    s0, dens, attr = syn.make_pile(N, 1, seed=0)
    lo, hi = syn.action_limits()
    ns, H = 4096, 5
    nom = g['nominal']
    sigma = 2.0 * 24 / 12.0 if noise_type == 'uniform' else 0.6
    ctx.mpc_begin(s0, attr, dens, nom, n_sample=ns, sigma=sigma, beta_filter=0.7, reward_weight=0.1, act_lo=lo,
                  act_hi=hi, seed=11, noise_type=noise_type)
    rng = np.random.default_rng(6)
    u = (rng.random((ns, H, 4)) * (2.0 if noise_type == 'uniform' else 1.0) - (1.0 if noise_type == 'uniform' else 0.0)).astype(np.float32)
    ctx.mpc_sample(0, noise=u)
    a = ctx.mpc_get(actions=True)['actions'].astype(np.float64)
    if noise_type == 'uniform':
        resid = np.zeros((ns, 4))
        expect = np.empty((ns, H, 4))
        for t in range(H):
            resid = 0.7 * (sigma * u[:, t].astype(np.float64)) + resid * (1.0 - 0.7)
            expect[:, t] = np.clip(nom[t] + resid, lo, hi)
    else:
        expect = lo + u.astype(np.float64) * (np.asarray(hi) - np.asarray(lo))
    np.testing.assert_allclose(a, expect, rtol=0, atol=2e-6)
    ctx.mpc_sample(1)
    a = ctx.mpc_get(actions=True)['actions'].astype(np.float64)
    np.testing.assert_allclose(a.mean(0), g[noise_type + '/mean'], atol=0.15)
    np.testing.assert_allclose(a.std(0), g[noise_type + '/std'], rtol=0.08, atol=0.02)
    assert (a.min(0) >= np.asarray(lo) - 1e-6).all() and (a.max(0) <= np.asarray(hi) + 1e-6).all()
    resid = a - nom[None]
    corr = [np.corrcoef(resid[:, t, 0], resid[:, t + 1, 0])[0, 1] for t in range(H - 1)]
    np.testing.assert_allclose(corr, g[noise_type + '/resid_lag1_corr'], atol=0.08)


def test_mppi_two_shards_combine_on_device(ctx, golden):
    """Two ranks' records (sample_offset 0 and 32) combined by the device kernel == one rank
    with all 64 samples == the host mirror in sharding.py."""
    from dyn_res_pile_manip_amd import sharding
    g = golden.mppi
    acts = g['opt_act_seqs'][:, :, 0, :].astype(np.float32)
    obs_goal = syn.goal_distance_image(syn.goal_mask('disc'))
    N = 16
    ctx.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
    s0, dens, attr = syn.make_pile(N, 1, seed=1)
    lo, hi = syn.action_limits()
    kw = dict(sigma=0.6, beta_filter=0.7, reward_weight=0.1, act_lo=lo, act_hi=hi)
    ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=64, **kw)
    ctx.mpc_set_actions(acts)
    ctx.mpc_rollout()
    full = ctx.mpc_partials()
    r_full = ctx.mpc_get(rewards=True)['rewards']
    want = ctx.mpc_update(full)
    recs = []
    for rank in range(2):
        ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=32, sample_offset=32 * rank, **kw)
        ctx.mpc_set_actions(acts[32 * rank:32 * rank + 32])
        ctx.mpc_rollout()
        rec = ctx.mpc_partials()
        host = sharding.make_record(0.1, r_full[32 * rank:32 * rank + 32], acts[32 * rank:32 * rank + 32], 32 * rank)
        np.testing.assert_allclose(rec, host, rtol=1e-9, atol=1e-9)
        recs.append(rec)
    got = ctx.mpc_update(np.stack(recs))
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)
    nominal, stats = sharding.combine_records(np.stack(recs), 64)
    np.testing.assert_allclose(nominal, want, rtol=1e-10, atol=1e-10)
    assert stats['argmax'] == int(np.argmax(r_full))


def test_rccl_communicator_single_rank(ctx, golden):
    """RCCL path with a one-rank communicator: unique id, init, the all-gather inside
    drp_mpc_update_device and the combine -- same answer as the communicator-free path."""
    g = golden.mppi
    acts = g['opt_act_seqs'][:, :, 0, :].astype(np.float32)
    obs_goal = syn.goal_distance_image(syn.goal_mask('disc'))
    N = 16
    ctx.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
    s0, dens, attr = syn.make_pile(N, 1, seed=1)
    lo, hi = syn.action_limits()
    ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=64, sigma=0.6, beta_filter=0.7,
                  reward_weight=0.1, act_lo=lo, act_hi=hi)
    ctx.mpc_set_actions(acts)
    ctx.mpc_rollout()
    ctx.mpc_update_device()
    want = ctx.mpc_get(nominal=True)['nominal']
    uid = ctx.comm_unique_id()
    assert len(uid) == 128
    ctx.comm_init(uid, 0, 1)
    try:
        ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=64, sigma=0.6, beta_filter=0.7,
                      reward_weight=0.1, act_lo=lo, act_hi=hi)
        ctx.mpc_set_actions(acts)
        ctx.mpc_rollout()
        ctx.mpc_update_device()
        got = ctx.mpc_get(nominal=True)['nominal']
    finally:
        ctx._ck(ctx.lib.drp_comm_destroy(ctx.h))
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize('attr_kind', ['zero', 'uniform', 'per_sample', 'per_particle'])
def test_fused_self_edge_constant(ctx, attr_kind):
    """km_prop replaces the self loop's encoder chain by a per-sample constant when a sample's
    attributes are all equal (k_cself); otherwise the self slot runs the chain like any other.
    Both against the ORACLE's free-running rollout: the same edge sets at every step, every step within a
    flat 1e-4 of its displacement."""
    nb, N, ns, H = 2, 150, 6, 4
    s0, dens, attr = syn.make_pile(N, n_batch=nb, seed=17)
    rng = np.random.default_rng(5)
    if attr_kind == 'uniform':
        attr = np.full_like(attr, 0.37)
    elif attr_kind == 'per_sample':
        attr = np.repeat(np.array([[0.2], [-0.6]], np.float32), N, axis=1)
    elif attr_kind == 'per_particle':
        attr = rng.uniform(-1, 1, attr.shape).astype(np.float32)      # no constant: generic path
    acts = syn.sample_pushes(ns * nb, H, seed=3)
    ref = osp.rollout(ctx.W, s0, dens, attr, acts, ctx.M34, 24.0)
    set_engine(ctx, 'fused')
    out, _ = ctx.rollout(s0, attr, dens, acts)
    check_rollout(ctx, ctx.W, s0, attr, dens, acts, ref, out)
    idx = ctx.debug_fetch('nbr_idx', (ns * nb, N, 10), np.int16)
    assert (idx[..., 0] == np.arange(N)[None, :]).all()              # self loop in slot 0 on this engine


def test_elite_update_two_shards_and_device_path(ctx, golden):
    """Elite (CEM-style) update -- an extension, not in the reference (include/drp.h): the nominal sequence
    becomes the mean of the k best samples' sequences.  One rank with all 64 samples == two ranks' records
    combined (on the device and by the host mirror in sharding.py) == the communicator-free and the
    one-rank-communicator device paths; a short rank pads; ties go to the lower global index."""
    from dyn_res_pile_manip_amd import sharding
    g = golden.mppi
    acts = g['opt_act_seqs'][:, :, 0, :].astype(np.float32)
    obs_goal = syn.goal_distance_image(syn.goal_mask('disc'))
    N, k = 16, 12
    ctx.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
    s0, dens, attr = syn.make_pile(N, 1, seed=1)
    lo, hi = syn.action_limits()
    kw = dict(sigma=0.6, beta_filter=0.7, reward_weight=0.1, act_lo=lo, act_hi=hi)
    ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=64, **kw)
    ctx.mpc_set_actions(acts)
    ctx.mpc_rollout()
    r_full = ctx.mpc_get(rewards=True)['rewards']
    full = ctx.mpc_elite(k)
    host_full = sharding.make_elite_records(r_full, acts, k)
    np.testing.assert_array_equal(full, host_full)
    assert (np.diff(full[:, 0]) <= 0).all()
    want = ctx.mpc_update_elite(full, k)
    order = np.lexsort((np.arange(64), -r_full.astype(np.float64)))[:k]
    np.testing.assert_allclose(want, acts[order].astype(np.float64).mean(0), rtol=1e-12, atol=1e-12)
    # the device path (statistics + elite), without and with a one-rank communicator
    ctx.mpc_update_elite_device(k)
    np.testing.assert_array_equal(ctx.mpc_get(nominal=True)['nominal'], want)
    assert ctx.mpc_stats()['argmax'] == int(np.argmax(r_full))
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    try:
        ctx.mpc_update_elite_device(k)
        np.testing.assert_array_equal(ctx.mpc_get(nominal=True)['nominal'], want)
    finally:
        ctx._ck(ctx.lib.drp_comm_destroy(ctx.h))
    # two shards of 40 + 24 samples, the second shorter than... no: a third "rank" of 5 samples pads its records
    recs = []
    for lo_s, hi_s in ((0, 40), (40, 59), (59, 64)):
        ctx.mpc_begin(s0, attr, dens, g['nominal'], n_sample=hi_s - lo_s, sample_offset=lo_s, **kw)
        ctx.mpc_set_actions(acts[lo_s:hi_s])
        ctx.mpc_rollout()
        rec = ctx.mpc_elite(k)
        np.testing.assert_array_equal(rec, sharding.make_elite_records(r_full[lo_s:hi_s], acts[lo_s:hi_s], k, lo_s))
        recs.append(rec)
    assert recs[2][5:, 1].max() == -1.0 and np.isneginf(recs[2][5:, 0]).all()
    got = ctx.mpc_update_elite(np.stack(recs), k)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    nominal, n_el, worst = sharding.combine_elite_records(np.stack(recs), k)
    np.testing.assert_allclose(nominal, want, rtol=1e-12, atol=1e-12)
    assert n_el == k and worst == float(r_full[order[-1]])


def test_elite_of_many_samples_takes_the_round_path(ctx):
    """5 000 samples on one rank: their keys no longer fit the LDS sort, the k-rounds selection runs instead --
    the same records as the host mirror, and the same nominal."""
    from dyn_res_pile_manip_amd import sharding
    N, ns, H, k = 8, 5000, 2, 40
    obs_goal = syn.goal_distance_image(syn.goal_mask('disc'))
    ctx.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
    s0, dens, attr = syn.make_pile(N, 1, seed=4)
    lo, hi = syn.action_limits()
    acts = syn.sample_pushes(ns, H, seed=4)
    acts[100] = acts[7]                      # an exact tie between two samples: the lower index wins
    ctx.mpc_begin(s0, attr, dens, syn.nominal_pushes(H, seed=4), n_sample=ns, sigma=0.6, beta_filter=0.7,
                  reward_weight=0.1, act_lo=lo, act_hi=hi)
    ctx.mpc_set_actions(acts)
    ctx.mpc_rollout()
    r = ctx.mpc_get(rewards=True)['rewards']
    assert r[100] == r[7]
    rec = ctx.mpc_elite(k)
    np.testing.assert_array_equal(rec, sharding.make_elite_records(r, acts, k))
    nominal = ctx.mpc_update_elite(rec, k)
    want, n_el, _ = sharding.combine_elite_records(rec[None], k)
    np.testing.assert_allclose(nominal, want, rtol=1e-12, atol=1e-12)
    assert n_el == k
