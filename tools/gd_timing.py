"""Time one gradient-descent planner iteration at the reference's demo shape
(50 trajectories x 30 particle re-samplings = 1500 rows, horizon 1)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine, particle_num_to_iter_time
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(0)), 0.08)
eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
lo, hi = syn.action_limits()
for N in ([int(a) for a in sys.argv[1:]] or (20, 50, 100, 300)):
    eng.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
    s0, dens, attr = syn.make_pile(N, 30, seed=N)
    acts = np.repeat(np.stack([syn.nominal_pushes(1, seed=i) for i in range(50)]), 30, axis=0).astype(np.float32)
    eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
    for _ in range(30):
        eng.lib.drp_gd_step(eng.h, None)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        eng.lib.drp_gd_step(eng.h, None)
    eng.sync()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    print('N=%3d B=1500: %.3f ms per GD iteration (rollout+reward+backward+Adam+clip); reference time model '
          '(its GPU, batch 300): %d ms' % (N, ms, particle_num_to_iter_time(N)))
