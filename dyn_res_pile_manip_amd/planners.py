"""Host-side mirror of the reference's `planners.py` call surface (PlannerGD), driving
the HIP engine through the C ABI.

Same method names, argument orders and return structures as the reference
(SURVEY.md 8b); tensors may be torch tensors or numpy arrays.
`trajectory_optimization_ptcl_multi_traj` runs, by `config['mpc']['mpc_type']`:
  'GD'   the reference's live loop (planners.py:661-764): rollout, reward, reverse mode, Adam and
         the clip box on the device (drp_gd_step), for exactly the iteration count the reference
         runs (planners.py:679-682);
  'MPPI' the sampling planner built from the reference's own (dead-code) sampler and update
         (planners.py:69-190, :549-561): forward-only, the form that shards over GPUs;
  'CEM'  elite selection instead of the softmax update (not in the reference).
"""
import time

import numpy as np

from .gnn_dyn import PropNetDiffDenModel, _like, _to_np

DEBUG = False


def particle_num_to_iter_time(particle_num):
    """planners.py:25-28: the reference's fitted ms per GD iteration at batch 300 on its
    (unstated) GPU.  The GD planner divides its time budget by it (gd_iteration_count)."""
    t = (2969.3971 - 69.923244 * particle_num + 1.8509846 * particle_num ** 2) / 200.
    return max(int(t), 1)


GD_SLOTS = 8        # include/drp.h DRP_GD_SLOTS: result slots of drp_gd_step_async
GD_AHEAD = 3        # iterations enqueued ahead of the one the host waits for (< GD_SLOTS)


def gd_iteration_count(n_update_iter, time_lim_ms, particle_num):
    """planners.py:590,679-682: the GD loop runs min(n_update_iter, int(time_lim / model)) iterations,
    the budget in ms divided by the fitted per-iteration time -- a function of the arguments only (the
    reference's wall-clock check is commented out, planners.py:765-767).  The arithmetic is the
    reference's, (time_lim / 1000) * 1000 / model, so the truncation falls where it does there.
    time_lim = inf (the signature's default) overflows int() in the reference; here it means no bound."""
    if np.isinf(time_lim_ms):
        return int(n_update_iter)
    bound = int((time_lim_ms / 1000.0) * 1000.0 / particle_num_to_iter_time(particle_num))
    return min(int(n_update_iter), bound)


def world2cam_affine(cam_extrinsic):
    """planners.py:197-203: rows 0..2 of inv(inv(cam_ext) diag(1,-1,-1,1)), built in
    float64 and cast to fp32 as the reference does."""
    gl = np.diag([1.0, -1.0, -1.0, 1.0])
    m = np.linalg.inv(np.matmul(np.linalg.inv(np.asarray(cam_extrinsic, dtype=np.float64)), gl))
    return np.ascontiguousarray(m[:3, :4], dtype=np.float32)


class Planner(object):
    def __init__(self, config, env):
        self.config = config
        self.action_dim = 4
        self.global_scale = config['dataset']['global_scale']
        self.img_ch = 1
        self.n_his = config['train']['n_history']
        self.env = env
        self.cam_params = self.env.get_cam_params()
        self.is_real = self.env.is_real
        if not self.is_real:
            self.cam_extrinsic = self.env.get_cam_extrinsics()
        else:
            raise NotImplementedError('real-robot path (gen_s_delta_irl) is out of scope')
        self.screenHeight = self.env.screenHeight
        self.screenWidth = self.env.screenWidth

    def trajectory_optimization(self, state_cur, obs_goal, model_dy, act_seq, n_sample,
                                n_look_ahead, n_update_iter, action_lower_lim, action_upper_lim,
                                use_gpu):
        pass   # empty in the reference too (planners.py:55-62)


class PlannerGD(Planner):
    def __init__(self, config, env):
        super(PlannerGD, self).__init__(config, env)
        self._m34 = world2cam_affine(self.cam_extrinsic)
        self._goal_key = None

    # ---- engine plumbing -----------------------------------------------------------
    def _bind(self, model_dy):
        if not isinstance(model_dy, PropNetDiffDenModel):
            raise NotImplementedError      # planners.py:355
        model_dy._claim()                  # models share the process's context: this one's weights in
        eng = model_dy.engine
        eng.set_camera(self._m34, float(self.global_scale), self.cam_params)
        return eng

    def _set_goal(self, eng, obs_goal, goal_coor=None, max_goal_pts=None, goal_key=None):
        """Install the reward's constants on the device, once per (goal image, goal pixels):
        goal_coor given -> field from the image + the caller's pixels; None -> the image also
        yields the farthest-point subsample of its goal pixels (planners.py:620-624).  The cache key is
        a digest of the arrays' CONTENT (two goals of equal shape and sum are different goals) and
        the engine object itself, kept alive by the key -- or, when the caller names its goal (`goal_key`: any hashable that
        changes whenever the goal image does; env/flex_env.py:1048 passes the SAME subgoal for all 20 MPC steps), that name:
        the 2 MB image is then neither copied nor hashed on a hit (1.5 - 2.2 ms per planner call otherwise)."""
        import hashlib
        from . import flex_rewards
        if goal_key is not None and goal_coor is None and self._goal_key is not None:
            named = (eng, flex_rewards.DIST_TRANSFORM, ('named', goal_key), None, int(max_goal_pts))
            if named[1:] == self._goal_key[1:] and named[0] is self._goal_key[0]:
                return
        g, _ = _to_np(obs_goal)
        g = np.ascontiguousarray(g, dtype=np.float32)
        mode = flex_rewards.DIST_TRANSFORM
        if goal_coor is None and goal_key is not None:
            eng.set_goal_image(g, max_goal_pts, 0, mode)
            self._goal_key = (eng, mode, ('named', goal_key), None, int(max_goal_pts))
            return
        dig = hashlib.blake2b(g.tobytes(), digest_size=16)
        if goal_coor is None:
            key = (eng, mode, g.shape, dig.digest(), int(max_goal_pts))
            if self._goal_key is None or key[1:] != self._goal_key[1:] or key[0] is not self._goal_key[0]:
                eng.set_goal_image(g, max_goal_pts, 0, mode)
                self._goal_key = key
            return
        gc, _ = _to_np(goal_coor)
        gc = np.ascontiguousarray(gc, dtype=np.float32)
        dig.update(gc.tobytes())
        key = (eng, mode, g.shape, dig.digest(), gc.shape)
        if self._goal_key is None or key[1:] != self._goal_key[1:] or key[0] is not self._goal_key[0]:
            eng.set_goal(flex_rewards.goal_field(g, eng), gc)
            self._goal_key = key

    def _clip_box(self, cvx_l=0):
        """planners.py:151-167: the clip box of convex region `cvx_l`."""
        r = self.env.cvx_region
        xd, yd = r[cvx_l, 1] - r[cvx_l, 0], r[cvx_l, 3] - r[cvx_l, 2]
        lo = np.array([r[cvx_l, 0], r[cvx_l, 2], r[cvx_l, 0] + xd * 0.15, r[cvx_l, 2] + yd * 0.15])
        hi = np.array([r[cvx_l, 1], r[cvx_l, 3], r[cvx_l, 1] - xd * 0.15, r[cvx_l, 3] - yd * 0.15])
        return lo, hi

    # ---- planners.py:69-190 ---------------------------------------------------------
    def sample_action_sequences(self, init_act_seq, init_act_label_seq, n_sample, action_lower_lim,
                                action_upper_lim, noise_type='normal'):
        """Host (numpy, global np.random) version with the reference's semantics; the MPC
        loop uses the device sampler (Philox) with the same filter and clip."""
        beta = self.config['mpc']['mppi']['beta_filter']
        init_act_seq = np.asarray(init_act_seq, dtype=np.float64)
        dim3 = init_act_seq.ndim == 3
        act_seqs = np.stack([init_act_seq] * n_sample)
        resid = np.zeros((n_sample,) + init_act_seq.shape[1:])
        lo, hi = self._clip_box(0)
        for i in range(self.n_his - 1, init_act_seq.shape[0]):
            if noise_type == 'normal':
                sigma = self.config['mpc']['sigma'] * self.global_scale / 12.0
                noise = np.random.normal(0, sigma, resid.shape)
            elif noise_type == 'uniform':
                sigma = 2.0 * self.global_scale / 12.0
                noise = np.random.uniform(-sigma, sigma, resid.shape)
            elif noise_type == 'total_rand':
                noise = np.zeros(resid.shape)
            else:
                raise ValueError('unknown noise type: %s' % noise_type)
            resid = beta * noise + resid * (1. - beta)
            act_seqs[:, i] += resid
            if dim3:
                act_seqs[:, i, 0] = np.clip(act_seqs[:, i, 0], lo, hi)       # region 0 (planners.py:161)
            else:
                # the step's label selects the convex region (planners.py:152)
                lo_l, hi_l = self._clip_box(int(init_act_label_seq[i]))
                act_seqs[:, i] = np.clip(act_seqs[:, i], lo_l, hi_l)
            if noise_type == 'total_rand':
                act_seqs[:, i, 0] = np.random.uniform(lo, hi, (n_sample, self.action_dim))
        return act_seqs

    # ---- planners.py:192-209 --------------------------------------------------------
    def world2cam(self, world_pts):
        p, proto = _to_np(world_pts)
        hom = np.concatenate([p, np.ones((p.shape[0], 1), np.float32)], 1)
        cam = (hom @ self._m34.T) / np.float32(self.global_scale)
        return _like(cam.astype(np.float32), proto)

    # ---- planners.py:211-257 --------------------------------------------------------
    def gen_s_delta(self, s_cur, action, model_dy=None):
        s, proto = _to_np(s_cur)
        a, _ = _to_np(action)
        assert s.shape[0] == a.shape[0]
        eng = self._engine(model_dy)
        return _like(eng.gen_s_delta(s, a), proto)

    def _engine(self, model_dy=None):
        if model_dy is not None:
            self._eng = self._bind(model_dy)
        if getattr(self, '_eng', None) is None:
            raise RuntimeError('no engine bound yet: pass model_dy once')
        return self._eng

    # ---- planners.py:302-370 --------------------------------------------------------
    def ptcl_model_rollout(self, s_cur_tensor, s_param_tensor, a_cur_tensor, model_dy, act_seqs,
                           enable_grad=True):
        s, proto = _to_np(s_cur_tensor)
        d, _ = _to_np(s_param_tensor)
        a, _ = _to_np(a_cur_tensor)
        acts, _ = _to_np(act_seqs)
        assert s.shape[2] == 3 and a.shape[1] == s.shape[1]
        self.particle_num = s.shape[1]
        eng = self._engine(model_dy)
        t0 = time.perf_counter()
        states, _ = eng.rollout(s, a, d, acts, want_states=True, want_reward=False)
        ms = (time.perf_counter() - t0) * 1e3
        return {'model_rollout': {'state_pred': _like(states, proto)}, 'rollout_time': ms}

    # ---- planners.py:372-452 --------------------------------------------------------
    def ptcl_evaluate_traj(self, obs_seqs, obs_goal, obs_goal_coor_tensor, debug=False,
                           funnel_dist=None, distractor_df_fn=None, act_seqs_tensor=None,
                           normalize_rew=True):
        if distractor_df_fn is not None:
            raise NotImplementedError('distractor rewards are unused on the live path')
        obs, proto = _to_np(obs_seqs)
        assert obs.ndim == 5 and obs.shape[4] == 3
        g, _ = _to_np(obs_goal)
        assert g.shape == (self.screenHeight, self.screenWidth)
        ns, H, cvx, N, _ = obs.shape
        eng = self._engine()
        self._set_goal(eng, g, obs_goal_coor_tensor)
        r = eng.reward(obs.reshape(ns * H * cvx, N, 3), normalize=normalize_rew)
        next_r = r.reshape(ns, H, cvx)
        reward_seqs = next_r[:, -1].copy()
        return _like(reward_seqs, proto), _like(next_r, proto)

    # ---- planners.py:549-561 --------------------------------------------------------
    def optimize_action(self, act_seqs, reward_seqs):
        lam = self.config['mpc']['mppi']['reward_weight']
        act_seqs = np.asarray(act_seqs, dtype=np.float64)
        reward_seqs = np.asarray(reward_seqs, dtype=np.float64)
        assert act_seqs.ndim == 4
        ns, H, cvx, ad = act_seqs.shape
        out = np.zeros((H, cvx, ad))
        for i in range(cvx):
            z = lam * reward_seqs[:, i]
            w = np.exp(z - z.max())
            w /= w.sum()
            out[:, i, :] = (w.reshape(-1, 1, 1) * act_seqs[:, :, i, :]).sum(0)
        return out

    # ---- planners.py:563-871 --------------------------------------------------------
    def trajectory_optimization_ptcl_multi_traj(self, state_cur_np, state_param, attr_cur_np,
                                                obs_goal, model_dy, act_seq, act_label_seq, n_sample,
                                                n_look_ahead, n_update_iter, action_lower_lim,
                                                action_upper_lim, use_gpu=True,
                                                rollout_best_action_sequence=True, reward_params=None,
                                                funnel_dist=None, distractor_df_fn=None, gd_loop=1,
                                                time_lim=float('inf'), goal_coor=None, seed=None,
                                                comm=None, noise_type='normal', wallclock_limit=False, goal_key=None):
        """Same contract as the reference (arguments, returned dict keys and shapes, voting rule).

        mpc_type 'GD' (the reference's live path): the traj_num x n_batch pushes of `act_seq` are
        independent Adam problems; the loop runs gd_iteration_count(n_update_iter, time_lim, N)
        iterations -- the reference's count, a function of the arguments only (planners.py:679-682);
        gd_loop only sizes rew_mean / rew_std (planners.py:647-648).
        mpc_type 'MPPI' / 'CEM': `act_seq` [n_look_ahead, traj_num, 4] seeds iteration 0 (its
        candidates are scored, the best becomes the nominal sequence), iterations 1..n_update_iter-1
        perturb the nominal with `n_sample` filtered-noise samples (`noise_type` as in
        sample_action_sequences) and update it.

        Extra keyword arguments (not in the reference):
          goal_coor        goal pixels given by the caller (skips the farthest-point subsample);
          goal_key         the caller's name for `obs_goal` (any hashable that changes when the image does): the installed goal
                           is then re-used without copying or hashing the image -- the default identifies it by a digest of
                           its content on every call;
          seed             key of the device sampler (default: drawn from numpy's global generator);
          comm             shard the sample axis (MPPI / CEM) or the trajectories (GD) over ranks:
                           a sharding.RcclComm / sharding.TorchComm, or the tuple (rank, n_ranks, uid).
                           Every rank gets the same action_sequence / reward / rew_mean / rew_std;
                           action_full and reward_full are the rank's own shard;
          wallclock_limit  True: additionally stop when the measured time exceeds time_lim (the check the
                           reference has commented out, planners.py:765-767); single rank only."""
        from . import sharding
        assert type(state_cur_np) == np.ndarray and state_cur_np.ndim == 3
        assert state_cur_np.shape[0] == state_param.shape[0] and state_cur_np.shape[2] == 3
        assert type(obs_goal) == np.ndarray and obs_goal.ndim == 2
        assert type(act_seq) == np.ndarray and act_seq.ndim == 3
        assert act_seq.shape[0] == act_label_seq.shape[0] and act_label_seq.ndim == 1
        assert act_seq.shape[0] == n_look_ahead
        if distractor_df_fn is not None:
            raise NotImplementedError('distractor rewards are unused on the live path')
        comm = sharding.as_comm(comm)
        if comm is not None and wallclock_limit:
            raise ValueError('wallclock_limit would stop the ranks at different iterations')
        start = time.time()
        self.particle_num = N = state_cur_np.shape[1]
        n_batch = state_cur_np.shape[0]
        H = n_look_ahead
        traj_num = int(act_seq.shape[1])
        eng = self._bind(model_dy)
        self._eng = eng
        rank, n_ranks = (comm.rank, comm.n_ranks) if comm is not None else (0, 1)
        mpc_type = self.config['mpc'].get('mpc_type', 'MPPI')
        # what cannot be sharded is refused on EVERY rank, from the global arguments, before the first collective or
        # rollout: a rank that raised alone would leave the others waiting in the exchange
        n_units = traj_num if mpc_type == 'GD' else int(n_sample)
        if n_ranks > n_units:
            raise ValueError('%d %s cannot be sharded over %d ranks' %
                             (n_units, 'trajectories' if mpc_type == 'GD' else 'samples', n_ranks))
        if comm is not None:
            comm.attach(eng)

        # planners.py:620-624 + env/flex_rewards.py:172-177: goal pixels (col,row), their
        # farthest-point subsample to 5N and the distance field, all on the device
        t_goal = time.time()
        goal_was = self._goal_key
        self._set_goal(eng, obs_goal, goal_coor, max_goal_pts=N * 5, goal_key=goal_key)
        goal_cached = self._goal_key is goal_was
        if not goal_cached:
            eng.sync()                                           # the install's kernels, so that its time is its own
        goal_time = time.time() - t_goal

        lo, hi = self._clip_box(0)
        cfg = self.config['mpc']
        if noise_type == 'uniform':
            sigma = 2.0 * self.global_scale / 12.0                  # planners.py:124
        else:
            sigma = cfg['sigma'] * self.global_scale / 12.0         # planners.py:116
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
            if comm is not None:                                     # every rank samples with rank 0's key
                seed = int(comm.allgather(np.array([seed], dtype=np.int64))[0, 0])

        max_reward = -np.inf * np.ones(n_batch, dtype=np.float32)
        max_reward_traj_idx = np.zeros(n_batch, dtype=np.int64)
        best_actions_of_samples = np.zeros((n_batch, H, self.action_dim), dtype=np.float32)
        rew_mean = np.zeros((1, int(n_update_iter) * int(gd_loop)), dtype=np.float32)
        rew_std = np.zeros((1, int(n_update_iter) * int(gd_loop)), dtype=np.float32)
        rollout_time = 0.0
        optim_time = 0.0
        time_lim_s = time_lim / 1000.0

        # Sharded over ranks, every rank keeps this bookkeeping for its own rows and the call ends with ONE exchange
        # (sharding.make_run_record / combine_run_records): no host collective per iteration.
        sharded = comm is not None and n_ranks > 1
        n_it_cap = rew_mean.shape[1]
        it_sums = np.zeros((n_it_cap, 3), dtype=np.float64)
        it_repl = np.zeros(n_it_cap, dtype=bool)
        best_iter = np.zeros(n_batch, dtype=np.int64)

        def aggregate(it, rewards, actions, ns, index_offset=0, replicated=False):
            """planners.py:721-727,736-738: per-column running max / argmax / best pushes, and the
            iteration's reward mean / std over column 0.  replicated: every rank runs these same rows."""
            r = rewards.reshape(ns, n_batch)
            cur_max, idx = r.max(0), r.argmax(0)                         # first maximum, as torch.max
            acts = actions.reshape(ns, n_batch, H, self.action_dim)[idx, np.arange(n_batch)]
            idx = idx + index_offset
            better = np.asarray(cur_max) > max_reward                     # strictly, per column (planners.py:724)
            max_reward[better] = np.asarray(cur_max)[better]
            max_reward_traj_idx[better] = np.asarray(idx)[better]
            best_actions_of_samples[better] = np.asarray(acts)[better]
            best_iter[better] = it
            if it < n_it_cap:
                if sharded:
                    c0 = r[:, 0].astype(np.float64)
                    it_sums[it] = (ns, c0.sum(), (c0 * c0).sum())
                    it_repl[it] = replicated
                else:
                    rew_mean[0, it] = r[:, 0].mean()
                    rew_std[0, it] = r[:, 0].std(ddof=1) if ns > 1 else 0.0

        i = 0
        if mpc_type == 'GD':
            # the reference's live loop (planners.py:661-764): every trajectory x batch column is an
            # independent Adam problem on its own push; rollout, reward, backward, Adam and the clip all
            # run on the device
            assert n_sample == traj_num, 'GD optimises the traj_num given trajectories (n_sample == traj_num)'
            n_iter = gd_iteration_count(n_update_iter, time_lim, N)
            if n_iter < 1:
                # the reference reaches its return statement with the loop variable unbound (planners.py:870)
                raise ValueError('time_lim %.3g ms admits no iteration at %d particles (%d ms each, planners.py:25-28)'
                                 % (time_lim, N, particle_num_to_iter_time(N)))
            t_lo, t_hi = sharding.shard_range(traj_num, rank, n_ranks)      # never empty: n_ranks <= traj_num (above)
            cand = np.repeat(act_seq[:, t_lo:t_hi].transpose(1, 0, 2), n_batch, axis=0).astype(np.float32)
            eng.gd_begin(state_cur_np, attr_cur_np, state_param, cand, cfg['gd']['lr'], lo, hi)   # [traj*nb,H,4]
            reward_seqs = np.zeros((cand.shape[0],), np.float32)
            act_seqs_last = cand
            # Iterations i + 1 .. i + GD_AHEAD are enqueued before the host waits for iteration i (the iteration's own kernels
            # write its rewards and updated pushes to pinned memory): the bookkeeping below runs beside the device, and a slow
            # turn of it does not leave the device idle.  With the opt-in wall-clock break the loop may stop after any
            # iteration, so nothing is enqueued ahead there.
            ahead = not wallclock_limit
            t0 = time.perf_counter()
            enqueued = 0
            for i in range(n_iter):
                before = act_seqs_last
                if ahead:
                    while enqueued < n_iter and enqueued <= i + GD_AHEAD:
                        eng.gd_step_async(enqueued % GD_SLOTS)
                        enqueued += 1
                    reward_seqs, act_seqs_last = eng.gd_wait(i % GD_SLOTS)
                else:
                    reward_seqs = eng.gd_step()
                    act_seqs_last = eng.gd_actions()
                # the rewards belong to the pushes before the update
                aggregate(i, reward_seqs, before, t_hi - t_lo, index_offset=t_lo)
                if wallclock_limit and (time.time() - start) > time_lim_s:
                    break
            optim_time += (time.perf_counter() - t0) * 1e3
            nominal = None
        else:
            n_iter = int(n_update_iter)
            mp = dict(sigma=sigma, beta_filter=cfg['mppi']['beta_filter'], reward_weight=cfg['mppi']['reward_weight'],
                      act_lo=lo, act_hi=hi, seed=seed, noise_type=noise_type)
            # iteration 0: every rank scores all traj_num candidates (identical everywhere, no exchange)
            cand = np.repeat(act_seq.transpose(1, 0, 2), n_batch, axis=0).astype(np.float32)  # [traj*nb,H,4]
            eng.mpc_begin(state_cur_np, attr_cur_np, state_param, act_seq[:, 0, :], n_sample=traj_num, **mp)
            eng.mpc_set_actions(cand)
            t0 = time.perf_counter()
            eng.mpc_rollout(False)
            got = eng.mpc_get(rewards=True)
            rollout_time += (time.perf_counter() - t0) * 1e3
            aggregate(0, got['rewards'], cand, traj_num, replicated=True)
            reward_seqs = got['rewards'].copy()
            act_seqs_last = cand
            r0 = reward_seqs.reshape(traj_num, n_batch)
            nominal = act_seq[:, int(np.argmax(r0.mean(1))), :].astype(np.float64)
            s_lo, s_hi = sharding.shard_range(n_sample, rank, n_ranks)
            ns_loc = s_hi - s_lo                                           # > 0: n_ranks <= n_sample (above)
            if n_iter > 1:
                eng.mpc_begin(state_cur_np, attr_cur_np, state_param, nominal, n_sample=ns_loc, sample_offset=s_lo, **mp)
            k_elite = int(cfg.get('cem', {}).get('n_elite', max(1, n_sample // 10)))
            def enqueue(it):
                """One iteration on the stream: sample, rollout, update, and its results on their way to the host."""
                eng.mpc_sample(it)
                eng.mpc_rollout(False)
                if comm is None or comm.device_update:
                    # partials -> [one RCCL all-gather] -> combine, on the stream
                    if mpc_type == 'CEM':
                        eng.mpc_update_elite_device(k_elite)   # elite update (not in the reference)
                    else:
                        eng.mpc_update_device()
                elif mpc_type == 'CEM':
                    eng.mpc_update_elite(comm.allgather(eng.mpc_elite(k_elite)), k_elite)
                else:
                    eng.mpc_update(comm.allgather(eng.mpc_partials()))
                eng.mpc_fetch_async(it & 1)

            # iteration i + 1 is enqueued before the host waits for iteration i: the bookkeeping runs beside the device
            # (not with the opt-in wall-clock break, after which nothing may have run, nor with a host-transport update,
            # whose exchange blocks anyway)
            ahead = not wallclock_limit and (comm is None or comm.device_update)
            t0 = time.perf_counter()
            if ahead and n_iter > 1:
                enqueue(1)
            for i in range(1, n_iter):
                if ahead:
                    if i + 1 < n_iter:
                        enqueue(i + 1)
                else:
                    enqueue(i)
                got = eng.mpc_wait(i & 1)
                aggregate(i, got['rewards'], got['actions'], ns_loc, index_offset=s_lo)
                reward_seqs, act_seqs_last = got['rewards'], got['actions']
                if wallclock_limit and (time.time() - start) > time_lim_s:
                    break
            optim_time += (time.perf_counter() - t0) * 1e3
            if n_iter > 1:
                nominal = eng.mpc_get(nominal=True)['nominal']

        if sharded:
            rec = sharding.make_run_record(it_sums, it_repl, max_reward, max_reward_traj_idx, best_iter,
                                           best_actions_of_samples)
            mean, std, ran, mr, mi, acts = sharding.combine_run_records(comm.allgather(rec), n_it_cap, n_batch)
            rew_mean[0, ran], rew_std[0, ran] = mean[ran], std[ran]
            max_reward[:] = mr
            max_reward_traj_idx[:] = mi
            best_actions_of_samples[:] = acts.reshape(n_batch, H, self.action_dim)

        # planners.py:773-781: vote = most frequent best-trajectory index over the columns,
        # ties -> the column with the highest reward
        counts = np.bincount(max_reward_traj_idx)
        idx_best_act = int(np.argmax(counts))
        idx_best_sample, best_r = -1, -np.inf
        for j in range(n_batch):
            if idx_best_act == max_reward_traj_idx[j] and max_reward[j] > best_r:
                idx_best_sample, best_r = j, max_reward[j]
        best_seq = best_actions_of_samples[idx_best_sample][None]       # [1,H,4]

        obs_seq_best, reward_best, next_r = None, None, None
        t_best = time.perf_counter()
        if rollout_best_action_sequence:
            # planners.py:821-851: B=1 re-rollout of the winner on column 0 + all-step reward
            states, rew = eng.rollout(state_cur_np[0:1], attr_cur_np[0:1], state_param[0:1], best_seq,
                                      want_states=True, want_reward=True)
            obs_seq_best = states[0]
            next_r = rew[0][:, None]                                    # [H,1]
            reward_best = rew[0, -1:].copy()                            # [1]
        ns_last = reward_seqs.shape[0] // n_batch
        return {'action_sequence': best_seq[0],
                'action_full': act_seqs_last[:, 0, :],
                'reward_full': reward_seqs.reshape(ns_last, n_batch)[:, 0],
                'observation_sequence': obs_seq_best,
                'observation_distractor_sequence': None,
                'reward': reward_best,
                'next_r': next_r,
                'rew_mean': rew_mean,
                'rew_std': rew_std,
                'nominal_sequence': nominal,
                'times': {'total_time': time.time() - start, 'rollout_time': rollout_time,
                          'optim_time': optim_time,
                          # this build's own: goal install (s; cached per goal image), the winner's re-rollout + reward (ms)
                          'goal_time': goal_time, 'goal_cached': goal_cached,
                          'best_rollout_time': (time.perf_counter() - t_best) * 1e3},
                'iter_num': i}


def fps_np(pcd, particle_num, init_idx=-1):
    """utils.py:451-466: farthest-point subsample, on the device (see `utils.fps_np`)."""
    from . import utils
    return utils.fps_np(pcd, particle_num, init_idx)
