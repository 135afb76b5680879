import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine
free0 = torch.cuda.mem_get_info()[0]
for rep in range(12):
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(rep)), 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    N = [50, 300, 150, 600][rep % 4]
    eng.set_goal_image(obs_goal, 5 * N)
    s0, dens, attr = syn.make_pile(N, 1, seed=rep)
    lo, hi = syn.action_limits()
    eng.mpc_begin(s0, attr, dens, syn.nominal_pushes(5, seed=rep), n_sample=256, sigma=0.6, beta_filter=0.7, reward_weight=0.1,
                  act_lo=lo, act_hi=hi, seed=rep, sample_offset=0)
    for it in range(30):
        eng.mpc_sample(it); eng.mpc_rollout(False); eng.mpc_update_device()
    st = eng.mpc_stats()
    assert np.isfinite(st['mean'])
    acts = np.repeat(np.stack([syn.nominal_pushes(1, seed=i) for i in range(20)]), 1, axis=0).astype(np.float32)
    eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
    for _ in range(20):
        eng.gd_step()
    eng.sync()
    eng.close()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print('12 create/run/destroy cycles ok; free HBM before %.1f MB after %.1f MB' % (free0 / 1e6, free1 / 1e6))
