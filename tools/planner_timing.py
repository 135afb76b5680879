"""One MPC step of the reference's live planner at its demo shape (config/mpc/config.yaml:
mpc_type GD, n_sample 50 trajectories x 30 particle re-samplings, n_look_ahead 1,
n_update_iter 200, time_lim 2000 ms), called as env/flex_env.py:1048-1065 calls it, plus the
same call with the sampling planner (MPPI, 1024 samples per iteration, horizon 10)."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel
from dyn_res_pile_manip_amd.planners import PlannerGD

obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
lo, hi = syn.action_limits()
for mpc_type, N, n_batch, traj, H, iters in (('GD', 20, 30, 50, 1, 200), ('GD', 50, 30, 50, 1, 200),
                                             ('GD', 100, 30, 50, 1, 200), ('MPPI', 300, 1, 8, 10, 20)):
    config = syn.default_config()
    config['mpc']['mpc_type'] = mpc_type
    env = syn.SyntheticEnv(config)
    model = PropNetDiffDenModel(config, True)
    model.load_state_dict(weights.random_state_dict(0), strict=False)
    planner = PlannerGD(config, env)
    s, dens, attr = syn.make_pile(N, n_batch=n_batch, seed=N)
    act_seq = np.stack([syn.nominal_pushes(H, seed=100 + i) for i in range(traj)], axis=1)   # [H, traj, 4]
    kw = dict(n_sample=traj if mpc_type == 'GD' else 1024, n_look_ahead=H, n_update_iter=iters,
              action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=1e9)
    planner.trajectory_optimization_ptcl_multi_traj(s, dens, attr, obs_goal, model, act_seq, np.zeros(H), **kw)
    t0 = time.perf_counter()
    res = planner.trajectory_optimization_ptcl_multi_traj(s, dens, attr, obs_goal, model, act_seq, np.zeros(H), **kw)
    ms = (time.perf_counter() - t0) * 1e3
    print('%-4s N=%3d n_batch=%2d candidates=%2d horizon=%2d: %d iterations in %.0f ms (%.2f ms each, rollout %.0f ms, '
          'optimiser %.0f ms); reward %.3f -> %.3f' % (mpc_type, N, n_batch, traj, H, res['iter_num'], ms, ms / max(res['iter_num'], 1),
                                                      res['times']['rollout_time'], res['times']['optim_time'],
                                                      res['rew_mean'][0, 0], res['reward'][0]), flush=True)
    if mpc_type == 'GD':
        # the shipped budget (config/mpc/config.yaml:40-43: time_lim 2000 ms, n_update_iter 200): the reference's
        # iteration count min(200, int(2000 / particle_num_to_iter_time(N)))
        kw2 = dict(kw, time_lim=2000.0)
        t0 = time.perf_counter()
        res = planner.trajectory_optimization_ptcl_multi_traj(s, dens, attr, obs_goal, model, act_seq, np.zeros(H), **kw2)
        ms = (time.perf_counter() - t0) * 1e3
        print('     with the shipped time_lim = 2000 ms: %d iterations (the reference\'s count), whole planner call %.1f ms' % (res['iter_num'] + 1, ms), flush=True)
    model.engine.close()
