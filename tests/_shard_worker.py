"""Worker of tests/test_gpu_sharded_planner.py: one rank of a world of processes.  Transport 'gloo' (default): the
ranks share GPU 0 and exchange through torch.distributed (comm=TorchComm); 'rccl': one GPU per rank, the product
transport (comm=RcclComm, the ncclUniqueId broadcast over the gloo group).  Writes the planner's result dict."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_planner(mpc_type, comm, n_sample=48, n_update_iter=4, traj=6, nb=2, N=40, H=2, seed=77, device=0):
    from dyn_res_pile_manip_amd import synthetic as syn, weights, flex_rewards
    from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel
    from dyn_res_pile_manip_amd.planners import PlannerGD
    old_dt, flex_rewards.DIST_TRANSFORM = flex_rewards.DIST_TRANSFORM, 'exact'     # 0.2 ms instead of 2 ms per goal
    config = syn.default_config()
    config['mpc']['mpc_type'] = mpc_type
    config['mpc']['cem'] = {'n_elite': 5}
    env = syn.SyntheticEnv(config)
    model = PropNetDiffDenModel(config, True, device=device)
    model.load_state_dict(weights.random_state_dict(seed=0), strict=False)
    planner = PlannerGD(config, env)
    s, dens, attr = syn.make_pile(N, n_batch=nb, seed=3)
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    act_seq = np.stack([syn.nominal_pushes(H, seed=40 + i) for i in range(traj)], axis=1)     # [H,traj,4]
    lo, hi = syn.action_limits()
    res = planner.trajectory_optimization_ptcl_multi_traj(
        s, dens, attr, obs_goal, model, act_seq, np.zeros(H), n_sample=traj if mpc_type == 'GD' else n_sample,
        n_look_ahead=H, n_update_iter=n_update_iter, action_lower_lim=lo, action_upper_lim=hi, use_gpu=True,
        time_lim=1e9, seed=seed, comm=comm)
    model.engine.close()
    flex_rewards.DIST_TRANSFORM = old_dt
    return {k: np.asarray(res[k]) for k in ('action_sequence', 'observation_sequence', 'reward', 'next_r', 'rew_mean',
                                            'rew_std', 'action_full', 'reward_full', 'iter_num')}


if __name__ == '__main__':
    import torch.distributed as dist
    out_dir, mpc_type = sys.argv[1], sys.argv[2]
    transport = sys.argv[3] if len(sys.argv) > 3 else 'gloo'
    dist.init_process_group('gloo')
    from dyn_res_pile_manip_amd.sharding import RcclComm, TorchComm
    if transport == 'rccl':
        from dyn_res_pile_manip_amd.engine import Engine
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = [None]
        if rank == 0:
            probe = Engine(0)
            uid[0] = probe.comm_unique_id()
            probe.close()
        dist.broadcast_object_list(uid, src=0)
        res = run_planner(mpc_type, RcclComm(uid[0], rank, world), device=rank)
    else:
        res = run_planner(mpc_type, TorchComm())
    np.savez(os.path.join(out_dir, '%s_rank%d.npz' % (mpc_type, dist.get_rank())), **res)
    dist.destroy_process_group()
