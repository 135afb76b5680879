"""Random shapes through the round-3 paths, each against the path it replaces (bit for bit where the arithmetic is the same):
  km_rollout (whole rollout in one launch)      vs  the step-by-step pipeline (DRP_NO_ROLLOUT_FUSED=1)      -- same bits
  km_prop with few tiles dealt one per workgroup vs  eight to a workgroup (DRP_NO_PROP_SPREAD=1)              -- same bits
  kmb_edge_encode (relation encoder backward, MFMA) vs kb_edge_encode (DRP_NO_BWD_EDGE_MFMA=1)               -- fp32 rounding
  tiles of 16 receivers x two slots (small workgroups) vs tiles of 32 receivers (DRP_PROP_PAIR_ROWS=0)        -- same bits
usage: python tools/fuzz_round3.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, '.')
import numpy as np

from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
blob = weights.blob_from_state_dict(weights.random_state_dict(seed=0))
M34 = world2cam_affine(syn.demo_cam_extrinsics())
obs_goal = syn.goal_distance_image(syn.goal_mask('I'))


def engine(env):
    for k in ('DRP_NO_ROLLOUT_FUSED', 'DRP_NO_PROP_SPREAD', 'DRP_NO_BWD_EDGE_MFMA', 'DRP_ROLLOUT_MAX_N', 'DRP_PROP_PAIR_ROWS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(0)
    e.load_weights(blob, 0.08)
    e.set_camera(M34, 24.0, syn.demo_cam_params())
    return e


bad = 0
for c in range(cases):
    kind = c % 4
    if kind == 0:
        # km_rollout: any sample size whose workgroup holds up to 3072 rows
        N = int(rng.integers(1, 257))
        nb = int(rng.choice([1, 1, 2, 3, 30]))
        ns = int(rng.integers(1, 1400))
        H = int(rng.integers(1, 6))
        B = ns * nb
        if -(-B // 256) * N > 3072 or B * H * N > 6_000_000:
            continue
        s0, dens, attr = syn.make_pile(N, nb, seed=c)
        if rng.random() < 0.3:
            attr = rng.uniform(-1, 1, attr.shape).astype(np.float32)
        acts = syn.sample_pushes(B, H, seed=c)
        out = []
        for env in ({'DRP_ROLLOUT_MAX_N': '256'}, {'DRP_NO_ROLLOUT_FUSED': '1'}):
            e = engine(env)
            e.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
            e.probe_begin('prop')
            st, rw = e.rollout(s0, attr, dens, acts, want_states=True, want_reward=True)
            _, nl = e.probe_read()
            out.append((st, rw, nl, e.debug_fetch('nbr_idx', (B, N, 10), np.int16)))
            e.close()
        ok = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][3], out[1][3])
        print('rollout  N=%3d nb=%2d ns=%4d H=%d  launches %d / %d  %s' % (N, nb, ns, H, out[0][2], out[1][2], 'same bits' if ok else 'DIFFERENT'))
    elif kind == 1:
        # km_prop spread: large samples, small batches (per-step kernels)
        N = int(rng.integers(257, 900))
        ns = int(rng.integers(1, 60))
        H = 2
        s0, dens, attr = syn.make_pile(N, 1, seed=c)
        acts = syn.sample_pushes(ns, H, seed=c)
        out = []
        for env in ({}, {'DRP_NO_PROP_SPREAD': '1'}, {'DRP_PROP_PAIR_ROWS': '0'}):
            e = engine(env)
            out.append(e.rollout(s0, attr, dens, acts)[0])
            e.close()
        ok = np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
        print('spread   N=%3d ns=%3d  %s' % (N, ns, 'same bits' if ok else 'DIFFERENT'))
    elif kind == 3:
        # paired tiles: workgroups of up to 128 rows (whole-rollout launch, step-by-step whole-sample kernels, tape of the GD planner)
        N = int(rng.integers(1, 129))
        nb = int(rng.choice([1, 1, 2, 3]))
        per_wg = max(1, 128 // N)
        ns = int(rng.integers(1, 256 * per_wg + 1)) if rng.random() < 0.7 else int(rng.integers(1, 40))
        H = int(rng.integers(1, 4))
        B = ns * nb
        s0, dens, attr = syn.make_pile(N, nb, seed=c)
        if rng.random() < 0.3:
            attr = rng.uniform(-1, 1, attr.shape).astype(np.float32)
        acts = syn.sample_pushes(B, H, seed=c)
        fused = rng.random() < 0.5
        out = []
        for env in ({}, {'DRP_PROP_PAIR_ROWS': '0'}):
            if not fused:
                env = dict(env, DRP_NO_ROLLOUT_FUSED='1')
            e = engine(env)
            e.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
            st, rw = e.rollout(s0, attr, dens, acts, want_states=True, want_reward=True)
            lo, hi = syn.action_limits()
            e.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
            r, g = e.gd_grad()[:2]
            out.append((st, rw, r, g))
            e.close()
        ok = all(np.array_equal(a, b_) for a, b_ in zip(out[0], out[1]))
        print('pair     N=%3d nb=%d ns=%4d H=%d %s  %s' % (N, nb, ns, H, 'one launch' if fused else 'per step  ', 'same bits' if ok else 'DIFFERENT'))
    else:
        # relation encoder backward on the matrix cores: GD gradients at horizon 2 and 3
        N = int(rng.integers(5, 200))
        nb = int(rng.choice([1, 2, 5]))
        traj = int(rng.integers(1, 40))
        H = int(rng.integers(2, 4))
        s0, dens, attr = syn.make_pile(N, nb, seed=c)
        acts = np.repeat(np.stack([syn.nominal_pushes(H, seed=c * 100 + i) for i in range(traj)]), nb, axis=0).astype(np.float32)
        lo, hi = syn.action_limits()
        out = []
        for env in ({}, {'DRP_NO_BWD_EDGE_MFMA': '1'}):
            e = engine(env)
            e.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
            e.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
            r, g, gs = e.gd_grad(want_state_grad=True)
            out.append((r, g, gs))
            e.close()
        scale = max(np.abs(out[1][1]).max(), 1e-12)
        err = np.abs(out[0][1] - out[1][1]).max() / scale
        ok = np.array_equal(out[0][0], out[1][0]) and err < 2e-5
        print('edge bwd N=%3d nb=%d traj=%2d H=%d  max |dg| / max |g| = %.2e  %s' % (N, nb, traj, H, err, 'ok' if ok else 'OFF'))
    bad += 0 if ok else 1
print('%d case(s) off' % bad)
sys.exit(1 if bad else 0)
