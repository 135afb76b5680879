"""One MPC step of the reference's loop (env/flex_env.py:1016-1110) with every piece on the device:
observation (synthetic depth image) -> particles (obs2ptcl_fixed_num_batch) -> density -> goal ->
planner (mpc_type GD, as the reference's config) -> push.  The simulator step itself is out of scope."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

from dyn_res_pile_manip_amd import synthetic as syn, utils as dev, weights
from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel
from dyn_res_pile_manip_amd.planners import PlannerGD


def mpc_step(obs, subgoal, model, planner, particle_num, act_seq, cam, global_scale, n_update_iter=200, seed=0):
    t = {}
    t0 = time.perf_counter()
    np.random.seed(seed)
    obs_cur, particle_r = dev.obs2ptcl_fixed_num_batch(obs, particle_num, 30, cam, global_scale)      # :1020
    particle_den = 1.0 / (particle_r * particle_r)                                                    # :1022
    t['particles'] = time.perf_counter() - t0
    attr_cur = np.zeros((obs_cur.shape[0], particle_num), np.float32)                                 # :1044
    lo, hi = syn.action_limits()
    t0 = time.perf_counter()
    out = planner.trajectory_optimization_ptcl_multi_traj(                                           # :1048-1065
        obs_cur.astype(np.float32), particle_den.astype(np.float32), attr_cur, subgoal, model, act_seq,
        np.zeros(act_seq.shape[0]), n_sample=act_seq.shape[1], n_look_ahead=act_seq.shape[0],
        n_update_iter=n_update_iter, action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=2000.0)
    t['planner'] = time.perf_counter() - t0
    return out, t


if __name__ == '__main__':
    config = syn.default_config()
    config['mpc']['mpc_type'] = 'GD'
    env = syn.SyntheticEnv(config)
    model = PropNetDiffDenModel(config, True)
    model.load_state_dict(weights.random_state_dict(0), strict=False)
    dev.set_engine(model.engine)
    planner = PlannerGD(config, env)
    subgoal = syn.goal_distance_image(syn.goal_mask('I'))
    obs = syn.render_depth(4000, seed=1, kind='uniform')
    cam = syn.demo_cam_params()
    for particle_num in (20, 50, 100):
        act_seq = np.stack([syn.nominal_pushes(1, seed=10 + i) for i in range(50)], axis=1)           # [1, 50, 4]
        mpc_step(obs, subgoal, model, planner, particle_num, act_seq, cam, 24.0, n_update_iter=5)     # warm-up
        out, t = mpc_step(obs, subgoal, model, planner, particle_num, act_seq, cam, 24.0)
        print('particle_num %3d: particles %.1f ms, planner %.0f ms (%d iterations), push %s, predicted reward %.3f' %
              (particle_num, t['particles'] * 1e3, t['planner'] * 1e3, out['iter_num'],
               np.round(out['action_sequence'][0], 2).tolist(), float(out['reward'][0])))
