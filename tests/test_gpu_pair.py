"""Tiles of 16 receivers x two slots (PAIR, k_mlp_split.h prop_tiles) against tiles of 32 receivers: the same bits.

Small workgroups (up to 128 rows of the whole-sample kernels; batches of at most two tiles per CU of the per-step kernel)
run the propagation steps on tiles of 16 receivers whose 32 item columns are 16 receivers x two consecutive slots; the
receiver's lane adds its own column's term and then its partner's, slot k before slot k + 1 -- the order of the unpaired
loop (model/gnn_dyn.py:159-166: the segmented sum is the reference's Rr^T bmm, whose order the oracle fixes per
receiver).  Which of the two runs is the host's choice (rows per workgroup; above 64 rows the mean in-degree the last
lists of the shape had, read from memory the device writes): only ever a question of speed."""
import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.planners import world2cam_affine
from oracle import propnet_sparse as osp

pytestmark = pytest.mark.gpu

PAIR_ENVS = ('DRP_PROP_PAIR_ROWS', 'DRP_PROP_PAIR_ALWAYS', 'DRP_PROP_PAIR_DEG10', 'DRP_NO_ROLLOUT_FUSED', 'DRP_NO_PROP3')


def _engine(monkeypatch, env):
    from dyn_res_pile_manip_amd.engine import Engine
    for k in PAIR_ENVS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(seed=0)), 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    return eng


@pytest.mark.parametrize('path', ['rollout', 'prop3', 'steps'])
@pytest.mark.parametrize('N,ns,nb', [(20, 1024, 1),     # 80 rows per workgroup: five tiles of 16 where three of 32 were
                                     (32, 1024, 1),     # 128 rows: eight tiles, two to a SIMD
                                     (7, 300, 3),       # a ragged last tile, several batch columns
                                     (100, 24, 1),      # one sample per workgroup, 100 rows, a saturated pile
                                     (1, 40, 1),        # a single particle
                                     (300, 6, 1)])      # a handful of large samples: the per-step kernel, one tile per CU
def test_paired_tiles_give_the_unpaired_bits(monkeypatch, path, N, ns, nb):
    H = 3
    s0, dens, attr = syn.make_pile(N, nb, seed=N + 1)
    acts = syn.sample_pushes(ns * nb, H, seed=N)
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    base = {'rollout': {}, 'prop3': {'DRP_NO_ROLLOUT_FUSED': '1'}, 'steps': {'DRP_NO_PROP3': '1'}}[path]
    out = {}
    for tag, env in (('paired', {'DRP_PROP_PAIR_ALWAYS': '128'}), ('unpaired', {'DRP_PROP_PAIR_ROWS': '0'}),
                     ('by in-degree', {'DRP_PROP_PAIR_ALWAYS': '0', 'DRP_PROP_PAIR_DEG10': '50'})):
        eng = _engine(monkeypatch, dict(base, **env))
        eng.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
        res = []
        for at in (attr, ((np.arange(nb * N, dtype=np.float32).reshape(nb, N) + 1) % 3) * 0.5):
            for _ in range(2 if tag == 'by in-degree' else 1):     # the second call reads the first one's in-degrees
                st, rw = eng.rollout(s0, at.astype(np.float32), dens, acts, want_states=True, want_reward=True)
            res += [st, rw]
        out[tag] = res
        eng.close()
    assert np.isfinite(out['paired'][0]).all()
    for tag in ('unpaired', 'by in-degree'):
        for a, b in zip(out['paired'], out[tag]):
            assert np.array_equal(a, b), tag
    assert not np.array_equal(out['paired'][0], out['paired'][2])       # the per-particle attributes did something


@pytest.mark.parametrize('N,traj,nb,H', [(20, 40, 5, 2), (10, 50, 30, 1), (60, 3, 2, 3), (300, 2, 2, 2)])
def test_paired_tiles_leave_the_tape_and_the_gradients_unchanged(monkeypatch, N, traj, nb, H):
    """The GD planner's forward pass (km_prop3<TAPE> / km_prop<., TAPE>) writes ReLU masks per edge slot: a paired column
    writes the mask of ITS slot.  Rewards, push gradients and position gradients bit for bit."""
    s0, dens, attr = syn.make_pile(N, nb, seed=N)
    acts = np.repeat(np.stack([syn.nominal_pushes(H, seed=100 + i) for i in range(traj)]), nb, axis=0).astype(np.float32)
    lo, hi = syn.action_limits()
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    out = []
    for env in ({'DRP_PROP_PAIR_ALWAYS': '128'}, {'DRP_PROP_PAIR_ROWS': '0'}):
        eng = _engine(monkeypatch, env)
        eng.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
        eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
        r, g, gs = eng.gd_grad(want_state_grad=True)
        r2 = eng.gd_step()
        out.append((r, g, gs, r2))
        eng.close()
    assert np.isfinite(out[0][1]).all() and np.abs(out[0][1]).max() > 0
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('N,ns', [(20, 64), (32, 1024), (50, 8)])
def test_paired_rollout_against_the_oracle(monkeypatch, N, ns):
    """A free-running rollout through the paired kernels against the sparse oracle: per step the edge sets the two
    trajectories induce are equal and the step stays within a flat 1e-4 of its displacement."""
    H = 4
    s0, dens, attr = syn.make_pile(N, 1, seed=7)
    acts = syn.sample_pushes(ns, H, seed=8)
    M34 = world2cam_affine(syn.demo_cam_extrinsics())
    eng = _engine(monkeypatch, {'DRP_PROP_PAIR_ALWAYS': '128'})
    dev, _ = eng.rollout(s0, attr, dens, acts)
    eng.close()
    rows = np.unique(np.linspace(0, ns - 1, 16).astype(int))
    W = osp.weights_np(weights.random_state_dict(seed=0))
    ref = osp.rollout(W, s0, dens, attr, acts[rows], M34, 24.0)
    prev_ref = np.tile(s0, (len(rows), 1, 1))
    prev_dev = prev_ref
    for t in range(H):
        idx_r, cnt_r = osp.build_neighbours(prev_ref, osp.gen_s_delta(prev_ref, acts[rows, t], M34, 24.0))
        idx_d, cnt_d = osp.build_neighbours(prev_dev, osp.gen_s_delta(prev_dev, acts[rows, t], M34, 24.0))
        np.testing.assert_array_equal(cnt_d, cnt_r)
        np.testing.assert_array_equal(idx_d, idx_r)
        scale = max(np.abs(ref[:, t] - prev_ref).max(), 1e-6)
        assert np.abs(dev[rows, t] - ref[:, t]).max() / scale < 1e-4, t
        prev_ref, prev_dev = ref[:, t], dev[rows, t]
