"""numpy-facing wrapper of one C-ABI context (one GPU, one HIP stream)."""
import ctypes

import numpy as np

from . import _lib as L


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _fp(a):
    return a.ctypes.data_as(L.c_float_p)


def _dp(a):
    return a.ctypes.data_as(L.c_double_p)


_DEFAULT = {}


def default_engine(device=0):
    """The process's ONE context of a device, created on first use: what the reference-named mirrors (the model of
    gnn_dyn.py, the helpers of utils.py, config_reward_ptcl of flex_rewards.py) run on unless they are handed an Engine --
    one stream, one set of workspaces, one installed camera, as the reference's modules share one CUDA device.  It
    serves what its fused engine refuses (DRP_ERANGE) on the fp32 matrix engine by itself (`auto_engine`)."""
    device = int(device)
    eng = _DEFAULT.get(device)
    if eng is None or getattr(eng, 'h', None) is None:
        eng = _DEFAULT[device] = Engine(device, auto_engine=True)
    return eng


def set_default_engine(engine):
    """Make `engine` the context default_engine(engine.device) returns (None: forget it)."""
    if engine is None:
        _DEFAULT.clear()
    else:
        _DEFAULT[int(engine.device)] = engine


class Engine(object):
    """Owns a `drp_ctx`.  Every method takes/returns numpy arrays (fp32).

    auto_engine: a call the fused (or split) engine refuses with DRP_ERANGE -- weights with an entry beyond fp16, inputs
    whose proven activation bound leaves it: include/drp.h -- is repeated on the fp32 matrix engine (`mfma`), which has no
    such limit; the engine stays switched until other weights are loaded, and a warning is issued.  (The gradient-descent
    planner and the trainer choose their tape's engine by themselves -- include/drp.h, drp_gd_begin -- and never raise it.)"""

    def __init__(self, device=0, engine=None, auto_engine=False):
        self.lib = L.load()
        h = ctypes.c_void_p()
        rc = self.lib.drp_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise L.DrpError('drp_create(device=%d) failed (%d): %s' %
                             (device, rc, self.lib.drp_last_error(None).decode()))
        self.h = h
        self.device = int(device)
        self.H = 0
        self.auto_engine = bool(auto_engine)
        self.engine_id = L.ENGINE_FUSED              # drp_create's choice
        if engine is not None:
            self.set_engine(engine)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.drp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc < 0:
            msg = 'drp error %d: %s' % (rc, self.lib.drp_last_error(self.h).decode())
            raise (L.DrpRangeError if rc == L.DRP_ERANGE else L.DrpError)(msg)
        return rc

    def _ranged(self, call):
        """`call()` -> return code of an entry point that checks the split engine's range."""
        try:
            return self._ck(call())
        except L.DrpRangeError as e:
            if not self.auto_engine or self.engine_id not in (L.ENGINE_FUSED, L.ENGINE_SPLIT):
                raise
            import warnings
            warnings.warn('the split-fp16 engine refused the call (%s): continuing on the fp32 matrix engine' % e,
                          RuntimeWarning, stacklevel=3)
            self._engine_before_auto = self.engine_id   # restored by the next load_weights: the refusal belongs to these weights / inputs
            self.set_engine(L.ENGINE_MFMA)
            self._auto_switched = True
            return self._ck(call())

    # ---- constants ----------------------------------------------------------------
    def set_engine(self, engine):
        self._ck(self.lib.drp_set_engine(self.h, int(engine)))
        self.engine_id = int(engine)

    def device_info(self):
        name = ctypes.create_string_buffer(256)
        ncu = ctypes.c_int()
        mem = ctypes.c_size_t()
        self._ck(self.lib.drp_device_info(self.h, name, 256, ctypes.byref(ncu), ctypes.byref(mem)))
        return {'name': name.value.decode(), 'n_cu': ncu.value, 'hbm_bytes': mem.value}

    def load_weights(self, blob, adj_thresh=0.08):
        blob = _f32(blob).ravel()
        self._weights_owner = None                    # whoever believed its weights were resident no longer is right
        self._ck(self.lib.drp_load_weights(self.h, _fp(blob), blob.size, float(adj_thresh)))
        if self.auto_engine and self.engine_id == L.ENGINE_MFMA and getattr(self, '_auto_switched', False):
            # the fallback was for the OTHER weights: these get the chance of the engine the caller had chosen again
            self.set_engine(getattr(self, '_engine_before_auto', L.ENGINE_FUSED))
            self._auto_switched = False

    def set_camera(self, m34, global_scale, intr):
        m34 = _f32(m34).ravel()
        intr = _f32(intr).ravel()
        assert m34.size == 12 and intr.size == 4
        self._cam = (m34.copy(), float(global_scale))
        self._ck(self.lib.drp_set_camera(self.h, _fp(m34), float(global_scale), _fp(intr)))

    def set_camera_intrinsics(self, intr):
        """Change [fx,fy,cx,cy] only (the reward's projection), keeping the extrinsic map."""
        m34, gs = getattr(self, '_cam', (np.eye(3, 4, dtype=np.float32).ravel(), 1.0))
        self.set_camera(m34, gs, intr)

    def set_goal(self, field, goal_coor):
        field = _f32(field)
        goal_coor = _f32(goal_coor)
        assert field.ndim == 2 and goal_coor.ndim == 2 and goal_coor.shape[1] == 2
        self._ck(self.lib.drp_set_goal(self.h, _fp(field), field.shape[0], field.shape[1],
                                       _fp(goal_coor), goal_coor.shape[0]))

    def distance_transform(self, src, mode='cv5'):
        """cv2.distanceTransform(src, cv2.DIST_L2, 5) on the device ('cv5': OpenCV's 5x5 chamfer;
        'exact': Euclidean) -> [h,w] float32."""
        src = np.ascontiguousarray(np.asarray(src) != 0, dtype=np.uint8)
        out = np.empty(src.shape, np.float32)
        self._ck(self.lib.drp_distance_transform(self.h, src.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                                                 src.shape[0], src.shape[1], L.DIST_TRANSFORMS[mode], _fp(out)))
        return out

    def set_goal_image(self, obs_goal, max_goal_pts, fps_init=0, mode='cv5', want=False):
        """Goal field (env/flex_rewards.py:172-177) and goal pixel subsample (planners.py:620-624) from
        the goal distance image, computed and kept on the device.  want=True also returns
        (field [h,w], goal_coor [m,2])."""
        g = _f32(obs_goal)
        assert g.ndim == 2
        m = ctypes.c_int()
        if not want:
            self._ck(self.lib.drp_set_goal_image(self.h, _fp(g), g.shape[0], g.shape[1], L.DIST_TRANSFORMS[mode],
                                                 int(max_goal_pts), int(fps_init), None, None, ctypes.byref(m)))
            return m.value
        field = np.empty(g.shape, np.float32)
        coor = np.empty((int(max_goal_pts), 2), np.float32)
        self._ck(self.lib.drp_set_goal_image(self.h, _fp(g), g.shape[0], g.shape[1], L.DIST_TRANSFORMS[mode],
                                             int(max_goal_pts), int(fps_init), _fp(field), _fp(coor),
                                             ctypes.byref(m)))
        return field, coor[:m.value].copy()

    # ---- single operations --------------------------------------------------------
    def gen_s_delta(self, s_cur, action):
        s_cur, action = _f32(s_cur), _f32(action)
        B, N, _ = s_cur.shape
        assert action.shape == (B, 4)
        out = np.empty((B, N, 3), dtype=np.float32)
        self._ck(self.lib.drp_gen_s_delta(self.h, _fp(s_cur), _fp(action), B, N, _fp(out)))
        return out

    def build_graph(self, s_cur, s_delta):
        s_cur, s_delta = _f32(s_cur), _f32(s_delta)
        B, N, _ = s_cur.shape
        idx = np.empty((B, N, L.DRP_K), dtype=np.int16)
        cnt = np.empty((B, N), dtype=np.uint8)
        self._ck(self.lib.drp_build_graph(self.h, _fp(s_cur), _fp(s_delta), B, N,
                                          idx.ctypes.data_as(L.c_int16_p),
                                          cnt.ctypes.data_as(L.c_uint8_p)))
        return idx, cnt

    def step(self, a_cur, s_cur, s_delta, dens):
        a_cur, s_cur, s_delta, dens = _f32(a_cur), _f32(s_cur), _f32(s_delta), _f32(dens)
        B, N, _ = s_cur.shape
        assert a_cur.shape == (B, N) and s_delta.shape == (B, N, 3) and dens.shape == (B,)
        out = np.empty((B, N, 3), dtype=np.float32)
        self._ranged(lambda: self.lib.drp_step(self.h, _fp(a_cur), _fp(s_cur), _fp(s_delta), _fp(dens), B, N, _fp(out)))
        return out

    def forward(self, a_cur, s_cur, s_delta, dens, nbr_idx, nbr_cnt):
        a_cur, s_cur, s_delta, dens = _f32(a_cur), _f32(s_cur), _f32(s_delta), _f32(dens)
        nbr_idx = np.ascontiguousarray(nbr_idx, dtype=np.int16)
        nbr_cnt = np.ascontiguousarray(nbr_cnt, dtype=np.uint8)
        B, N, _ = s_cur.shape
        assert nbr_idx.shape == (B, N, L.DRP_K) and nbr_cnt.shape == (B, N)
        out = np.empty((B, N, 3), dtype=np.float32)
        self._ranged(lambda: self.lib.drp_forward(self.h, _fp(a_cur), _fp(s_cur), _fp(s_delta), _fp(dens),
                                                  nbr_idx.ctypes.data_as(L.c_int16_p),
                                                  nbr_cnt.ctypes.data_as(L.c_uint8_p), B, N, _fp(out)))
        return out

    def rollout(self, s0, attr, dens, actions, want_states=True, want_reward=False):
        s0, attr, dens, actions = _f32(s0), _f32(attr), _f32(dens), _f32(actions)
        nb, N, _ = s0.shape
        B, H, _ = actions.shape
        states = np.empty((B, H, N, 3), dtype=np.float32) if want_states else None
        rew = np.empty((B, H), dtype=np.float32) if want_reward else None
        self._ranged(lambda: self.lib.drp_rollout(self.h, _fp(s0), _fp(attr), _fp(dens), nb, N, _fp(actions), B, H,
                                                  _fp(states) if want_states else None,
                                                  _fp(rew) if want_reward else None))
        return states, rew

    def reward(self, state, normalize=True):
        state = _f32(state)
        Bp, N, _ = state.shape
        out = np.empty((Bp,), dtype=np.float32)
        self._ck(self.lib.drp_reward(self.h, _fp(state), Bp, N, int(bool(normalize)), _fp(out)))
        return out

    # ---- device-resident MPPI -----------------------------------------------------
    def mpc_begin(self, s0, attr, dens, nominal, n_sample, sigma, beta_filter, reward_weight,
                  act_lo, act_hi, seed=0, sample_offset=0, noise_type='normal'):
        s0, attr, dens = _f32(s0), _f32(attr), _f32(dens)
        nominal = np.ascontiguousarray(nominal, dtype=np.float64)
        nb, N, _ = s0.shape
        H = nominal.shape[0]
        p = L.MpcParams()
        p.n_batch, p.n_particles, p.n_sample, p.n_look_ahead = nb, N, int(n_sample), H
        p.sigma, p.beta_filter, p.reward_weight = float(sigma), float(beta_filter), float(reward_weight)
        for i in range(4):
            p.act_lo[i] = float(act_lo[i])
            p.act_hi[i] = float(act_hi[i])
        p.seed, p.sample_offset = int(seed), int(sample_offset)
        p.noise_type, p.reserved = L.NOISE_TYPES[noise_type], 0
        self._ranged(lambda: self.lib.drp_mpc_begin(self.h, ctypes.byref(p), _fp(s0), _fp(attr), _fp(dens), _dp(nominal)))
        self.H, self.nb, self.N, self.ns = H, nb, N, int(n_sample)

    def mpc_sample(self, iteration, noise=None):
        if noise is not None:
            noise = _f32(noise)
            assert noise.shape == (self.ns, self.H, 4)
        self._ranged(lambda: self.lib.drp_mpc_sample(self.h, _fp(noise) if noise is not None else None, int(iteration)))

    def mpc_set_actions(self, actions):
        actions = _f32(actions)
        assert actions.shape == (self.ns * self.nb, self.H, 4)
        self._ranged(lambda: self.lib.drp_mpc_set_actions(self.h, _fp(actions)))

    def mpc_rollout(self, reward_all_steps=False):
        self._ck(self.lib.drp_mpc_rollout(self.h, int(bool(reward_all_steps))))

    def mpc_partials(self, fetch=True):
        out = np.empty((6 + 4 * self.H,), dtype=np.float64) if fetch else None
        self._ck(self.lib.drp_mpc_partials(self.h, _dp(out) if fetch else None))
        return out

    def mpc_update(self, partials):
        partials = np.ascontiguousarray(partials, dtype=np.float64).reshape(-1, 6 + 4 * self.H)
        nominal = np.empty((self.H, 4), dtype=np.float64)
        self._ck(self.lib.drp_mpc_update(self.h, _dp(partials), partials.shape[0], _dp(nominal)))
        return nominal

    def mpc_update_device(self):
        self._ck(self.lib.drp_mpc_update_device(self.h))

    # elite (CEM-style) update: nominal = mean of the k best sequences (not in the reference; include/drp.h)
    def mpc_elite(self, k, fetch=True):
        out = np.empty((int(k), 2 + 4 * self.H), dtype=np.float64) if fetch else None
        self._ck(self.lib.drp_mpc_elite(self.h, int(k), _dp(out) if fetch else None))
        return out

    def mpc_update_elite(self, records, k):
        records = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, int(k), 2 + 4 * self.H)
        nominal = np.empty((self.H, 4), dtype=np.float64)
        self._ck(self.lib.drp_mpc_update_elite(self.h, _dp(records), records.shape[0], int(k), _dp(nominal)))
        return nominal

    def mpc_update_elite_device(self, k):
        self._ck(self.lib.drp_mpc_update_elite_device(self.h, int(k)))

    def mpc_get(self, actions=False, rewards=False, rewards_all=False, states=False, nominal=False):
        B = self.ns * self.nb
        a = np.empty((B, self.H, 4), np.float32) if actions else None
        r = np.empty((B,), np.float32) if rewards else None
        ra = np.empty((B, self.H), np.float32) if rewards_all else None
        s = np.empty((B, self.H, self.N, 3), np.float32) if states else None
        n = np.empty((self.H, 4), np.float64) if nominal else None
        self._ck(self.lib.drp_mpc_get(self.h, _fp(a) if actions else None, _fp(r) if rewards else None,
                                      _fp(ra) if rewards_all else None, _fp(s) if states else None,
                                      _dp(n) if nominal else None))
        return {'actions': a, 'rewards': r, 'rewards_all': ra, 'states': s, 'nominal': n}

    def mpc_fetch_async(self, slot):
        """The enqueued iteration's pushes and final rewards go to pinned memory behind its kernels (mpc_wait)."""
        self._ck(self.lib.drp_mpc_fetch_async(self.h, int(slot)))

    def mpc_wait(self, slot):
        B = self.ns * self.nb
        a = np.empty((B, self.H, 4), np.float32)
        r = np.empty((B,), np.float32)
        self._ck(self.lib.drp_mpc_wait(self.h, int(slot), _fp(a), _fp(r)))
        return {'actions': a, 'rewards': r}

    def mpc_stats(self):
        out = np.empty((8,), dtype=np.float64)
        self._ck(self.lib.drp_debug_fetch(self.h, b'stats', out.ctypes.data_as(ctypes.c_void_p), out.nbytes))
        return {'mean': out[0], 'std': out[1], 'max': out[2], 'argmax': int(out[3]), 'Z': out[4], 'm': out[5]}

    def fps(self, pts, k, init_idx=0):
        """utils.fps_np on the device: returns (pts[chosen], max distance, chosen indices)."""
        pts = _f32(pts)
        n, dim = pts.shape
        idx = np.empty((k,), np.int32)
        md = ctypes.c_float()
        self._ck(self.lib.drp_fps(self.h, _fp(pts), n, dim, int(k), int(init_idx),
                                  idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(md)))
        return pts[idx], md.value, idx

    # ---- training (row f4) -------------------------------------------------
    def train_begin(self, n_rollout, lr=1e-3, beta1=0.9):
        self._ck(self.lib.drp_train_begin(self.h, int(n_rollout), float(lr), float(beta1)))
        self._n_rollout = int(n_rollout)

    def train_step(self, states, states_delta, attrs, particle_nums, particle_dens, mode='update', want_grad=False):
        """One body of the loop at train/train_gnn_dyn.py:159-210 -> (loss, gradient blob or None)."""
        states, states_delta, attrs = _f32(states), _f32(states_delta), _f32(attrs)
        dens = _f32(particle_dens)
        nums = np.ascontiguousarray(particle_nums, dtype=np.int32)
        B, T1, N, _ = states.shape
        assert T1 == self._n_rollout + 1 and states_delta.shape == (B, T1 - 1, N, 3)
        assert attrs.shape == (B, T1, N) and nums.shape == (B,) and dens.shape == (B,)
        loss = ctypes.c_double()
        grad = np.empty((38403,), np.float32) if (want_grad and mode != 'eval') else None
        self._ck(self.lib.drp_train_step(self.h, _fp(states), _fp(states_delta), _fp(attrs),
                                         nums.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _fp(dens), B, N,
                                         L.TRAIN_MODES[mode], ctypes.byref(loss), None if grad is None else _fp(grad)))
        return loss.value, grad

    def train_set_lr(self, lr):
        self._ck(self.lib.drp_train_set_lr(self.h, float(lr)))

    def get_weights(self):
        blob = np.empty((38403,), np.float32)
        self._ck(self.lib.drp_get_weights(self.h, _fp(blob), blob.size))
        return blob

    # ---- particle extraction (row f2) -------------------------------------
    def depth2fgpcd(self, depth, mask, cam_params):
        """utils.depth2fgpcd on the device -> [n,3] float64."""
        depth = _f32(depth)
        h, w = depth.shape
        cam = _f64(cam_params)
        m8 = None if mask is None else np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        mp = None if m8 is None else m8.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        n = ctypes.c_int()
        self._ck(self.lib.drp_depth2fgpcd(self.h, _fp(depth), mp, h, w, _dp(cam), None, 0, ctypes.byref(n)))
        out = np.empty((n.value, 3), np.float64)
        if n.value:
            self._ck(self.lib.drp_depth2fgpcd(self.h, _fp(depth), mp, h, w, _dp(cam), _dp(out), n.value,
                                              ctypes.byref(n)))
        return out

    def downsample_pcd(self, pcd, voxel_size):
        """utils.downsample_pcd (open3d voxel_down_sample) on the device -> [m,3] float64."""
        pcd = _f64(pcd)
        n = pcd.shape[0]
        out = np.empty((n, 3), np.float64)
        m = ctypes.c_int()
        self._ck(self.lib.drp_downsample_pcd(self.h, _dp(pcd), n, float(voxel_size), _dp(out), n, ctypes.byref(m)))
        return out[:m.value].copy()

    def fps_pcd(self, pcd, particle_num, init_idx=None, batch=None, seed=0):
        """utils.fps for a batch of starts -> (pts [batch,N,3] float32, particle_r [batch] float64)."""
        pcd = _f64(pcd)
        if init_idx is not None:
            init = np.ascontiguousarray(np.atleast_1d(init_idx), dtype=np.int32)
            batch = init.shape[0]
            ip = init.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        else:
            batch = int(batch or 1)
            ip = None
        pts = np.empty((batch, int(particle_num), 3), np.float32)
        r = np.empty((batch,), np.float64)
        self._ck(self.lib.drp_fps_pcd(self.h, _dp(pcd), pcd.shape[0], int(particle_num), batch, ip, int(seed),
                                      _fp(pts), _dp(r)))
        return pts, r

    def fps_rad(self, pcd, radius, init_idx, cap=None):
        """utils.fps_rad on the device -> (pcd[chosen] float64, chosen indices)."""
        pcd = _f64(pcd)
        n = pcd.shape[0]
        cap = int(cap or n)
        idx = np.empty((cap,), np.int32)
        cnt = ctypes.c_int()
        self._ck(self.lib.drp_fps_rad(self.h, _dp(pcd), n, float(radius), int(init_idx), cap,
                                      idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(cnt)))
        idx = idx[:cnt.value].copy()
        return pcd[idx], idx

    def recenter(self, pcd, sampled, r):
        """utils.recenter for a batch: sampled [batch,N,3] float32, r [batch] -> [batch,N,3] float32."""
        pcd, sampled = _f64(pcd), _f32(sampled)
        batch, npts, _ = sampled.shape
        rr = _f64(np.broadcast_to(np.asarray(r, dtype=np.float64), (batch,)))
        out = np.empty_like(sampled)
        self._ck(self.lib.drp_recenter(self.h, _dp(pcd), pcd.shape[0], _fp(sampled), npts, batch, _dp(rr), _fp(out)))
        return out

    def obs2ptcl(self, depth_raw, global_scale, cam_params, particle_num, batch, init_idx=None, seed=0):
        """FlexEnv.obs2ptcl_fixed_num_batch on the device -> (ptcl [batch,N,3] f64, particle_r [batch],
        (#foreground points, #voxels))."""
        depth_raw = _f32(depth_raw)
        h, w = depth_raw.shape
        cam = _f64(cam_params)
        ip = None
        if init_idx is not None:
            init = np.ascontiguousarray(init_idx, dtype=np.int32)
            assert init.shape == (batch,)
            ip = init.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        out = np.empty((int(batch), int(particle_num), 3), np.float64)
        r = np.empty((int(batch),), np.float64)
        nfg, nd = ctypes.c_int(), ctypes.c_int()
        self._ck(self.lib.drp_obs2ptcl(self.h, _fp(depth_raw), h, w, float(global_scale), _dp(cam), int(particle_num),
                                       int(batch), ip, int(seed), _dp(out), _dp(r), ctypes.byref(nfg),
                                       ctypes.byref(nd)))
        return out, r, (nfg.value, nd.value)

    # ---- gradient-descent planner ---------------------------------------
    def gd_begin(self, s0, attr, dens, actions, lr, act_lo, act_hi):
        s0, attr, dens, actions = _f32(s0), _f32(attr), _f32(dens), _f32(actions)
        nb, N, _ = s0.shape
        B, H, _ = actions.shape
        lo, hi = _f32(act_lo), _f32(act_hi)
        self._ck(self.lib.drp_gd_begin(self.h, _fp(s0), _fp(attr), _fp(dens), nb, N, _fp(actions), B, H,
                                       float(lr), _fp(lo), _fp(hi)))
        self._gd = (B, H, N)

    def gd_grad(self, want_state_grad=False):
        B, H, N = self._gd
        r = np.empty((B,), np.float32)
        g = np.empty((B, H, 4), np.float32)
        gs = np.empty((B, H, N, 3), np.float32) if want_state_grad else None
        self._ck(self.lib.drp_gd_grad(self.h, _fp(r), _fp(g), _fp(gs) if want_state_grad else None))
        return r, g, gs

    def gd_step(self):
        B, H, N = self._gd
        r = np.empty((B,), np.float32)
        self._ck(self.lib.drp_gd_step(self.h, _fp(r)))
        return r

    def gd_step_async(self, slot):
        """Enqueue one iteration; its rewards and updated pushes go to pinned memory behind it (gd_wait)."""
        self._ck(self.lib.drp_gd_step_async(self.h, int(slot)))

    def gd_wait(self, slot):
        """(rewards [B] of the iterate before the update, pushes [B,H,4] after it) of the iteration in `slot`."""
        B, H, N = self._gd
        r = np.empty((B,), np.float32)
        a = np.empty((B, H, 4), np.float32)
        self._ck(self.lib.drp_gd_wait(self.h, int(slot), _fp(r), _fp(a)))
        return r, a

    def gd_actions(self):
        B, H, N = self._gd
        a = np.empty((B, H, 4), np.float32)
        self._ck(self.lib.drp_gd_get(self.h, _fp(a)))
        return a

    # ---- multi-GPU ------------------------------------------------------------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.drp_comm_unique_id(buf)
        if rc != 0:
            raise L.DrpError('drp_comm_unique_id failed: %s' % self.lib.drp_last_error(None).decode())
        return buf.raw

    def comm_init(self, uid, rank, n_ranks):
        self._ck(self.lib.drp_comm_init(self.h, uid, int(rank), int(n_ranks)))
        self._n_ranks = int(n_ranks)

    def comm_destroy(self):
        self._ck(self.lib.drp_comm_destroy(self.h))
        self._n_ranks = 1

    def comm_info(self):
        """{'n_ranks': ncclCommCount (0 without a communicator), 'rank', 'version', 'path'} of the RCCL this process bound."""
        n, r, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        path = ctypes.create_string_buffer(1024)
        self._ck(self.lib.drp_comm_info(self.h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(v), path, 1024))
        ver = v.value
        return {'n_ranks': n.value, 'rank': r.value, 'version': ver, 'path': path.value.decode(),
                'version_str': '%d.%d.%d' % (ver // 10000, (ver // 100) % 100, ver % 100) if ver >= 10000 else str(ver)}

    def debug_stall(self, ms):
        self._ck(self.lib.drp_debug_stall(self.h, int(ms)))

    def comm_allgather(self, arr):
        """All-gather one host array per rank over the context's communicator -> [n_ranks, *arr.shape]."""
        arr = np.ascontiguousarray(arr)
        # the communicator's own count, not a cached one: an aborted communicator is gone (and the call below says so)
        out = np.empty((max(1, self.comm_info()['n_ranks']),) + arr.shape, arr.dtype)
        self._ck(self.lib.drp_comm_allgather(self.h, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes,
                                             out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # ---- measurement ------------------------------------------------------------------
    def sync(self):
        self._ck(self.lib.drp_sync(self.h))

    def probe_begin(self, kernel_class):
        self._ck(self.lib.drp_probe_begin(self.h, kernel_class.encode() if kernel_class else None))

    def probe_read(self):
        ms = ctypes.c_double()
        n = ctypes.c_long()
        self._ck(self.lib.drp_probe_read(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def probe_work(self):
        """What the propagation kernels executed since probe_begin('prop'), counted by the kernels (include/drp.h)."""
        out = (ctypes.c_ulonglong * 8)()
        self._ck(self.lib.drp_probe_work(self.h, out))
        keys = ('chain_slots', 'cached_slots', 'tiles', 'tiles_last', 'encoder_tiles', 'mfmas', 'clk_cycles', 'clk_ticks')
        return {k: int(out[i]) for i, k in enumerate(keys)}

    def dispatch_reset(self):
        self._ck(self.lib.drp_dispatch_reset(self.h))

    def last_dispatch(self):
        """Names of the kernel variants launched since dispatch_reset() (include/drp.h drp_last_dispatch)."""
        buf = ctypes.create_string_buffer(8192)
        self._ck(self.lib.drp_last_dispatch(self.h, buf, len(buf)))
        return [s for s in buf.value.decode().split(';') if s]

    def dispatch_variants(self, default_only=True):
        buf = ctypes.create_string_buffer(8192)
        self.lib.drp_dispatch_variants(int(bool(default_only)), buf, len(buf))
        return [s for s in buf.value.decode().split(';') if s]

    def range_info(self):
        """{'shift', 'bound', 'wmax', 'ok'} of the split-fp16 relation encoder for the loaded weights (drp_range_info)."""
        k, ok = ctypes.c_int(), ctypes.c_int()
        bound, wmax = ctypes.c_double(), ctypes.c_double()
        self._ck(self.lib.drp_range_info(self.h, ctypes.byref(k), ctypes.byref(bound), ctypes.byref(wmax), ctypes.byref(ok)))
        return {'shift': k.value, 'bound': bound.value, 'wmax': wmax.value, 'ok': bool(ok.value)}

    def debug_fetch(self, name, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        self._ck(self.lib.drp_debug_fetch(self.h, name.encode(), out.ctypes.data_as(ctypes.c_void_p),
                                          out.nbytes))
        return out
