import sys, time
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine
N = 20
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(0)), 0.08)
eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
lo, hi = syn.action_limits()
eng.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
s0, dens, attr = syn.make_pile(N, 30, seed=N)
acts = np.repeat(np.stack([syn.nominal_pushes(1, seed=i) for i in range(50)]), 30, axis=0).astype(np.float32)
eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
for _ in range(5): eng.gd_step()
eng.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(8): eng.gd_step_async(i)
    t1 = time.perf_counter()
    for i in range(8): eng.gd_wait(i)
    t2 = time.perf_counter()
    print('enqueue %.1f us per iteration; 8 waits %.1f us total' % ((t1 - t0) / 8 * 1e6, (t2 - t1) * 1e6))
