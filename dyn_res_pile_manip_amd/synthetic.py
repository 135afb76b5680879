"""Synthetic stand-ins for what the planner reads from the simulator side.

The reference planner (planners.py:40-45,152-155) reads six things from its
`env` object: `is_real`, `get_cam_params()`, `get_cam_extrinsics()`,
`screenHeight`, `screenWidth` and `cvx_region`.  The real `FlexEnv`
(env/flex_env.py) needs the closed-source FleX simulator, so the hot path is
driven here by `SyntheticEnv`, which reproduces the demo camera
(env/flex_env.py:194-200: position (0, 18, 0), pitch -90 deg, fov 45 deg,
720x720) and the +-5 workspace (env/flex_env.py:454-458).

Also here: seeded synthetic piles, goal fields and nominal pushes with the
shapes SURVEY.md section 8(d) prescribes.  numpy/scipy only.
"""
import numpy as np

SCREEN = 720
GLOBAL_SCALE = 24.0
WKSPC_W = 5.0


def default_config():
    """The config keys the hot path reads (config/mpc/config.yaml)."""
    return {
        'dataset': {'global_scale': 24, 'wkspc_w': 5.0},
        'mpc': {
            'sigma': 0.3,
            'mppi': {'beta_filter': 0.7, 'reward_weight': 0.1},
            'gd': {'beta_filter': 0.7, 'lr': 0.05},
            'n_look_ahead': 1, 'n_sample': 50, 'n_update_iter': 200,
            'mpc_type': 'MPPI',
        },
        'train': {
            'n_history': 1,
            'particle': {'nf_effect': 64, 'adj_thresh': 0.08, 'add_delta': False},
        },
    }


def demo_cam_params():
    """[fx, fy, cx, cy] of the 45-degree-fov 720x720 demo camera."""
    f = (SCREEN / 2.0) / np.tan(np.deg2rad(45.0) / 2.0)
    return [float(f), float(f), SCREEN / 2.0, SCREEN / 2.0]


def demo_cam_extrinsics():
    """OpenGL view matrix of a camera at (0,18,0) looking straight down.

    With it planners.py:192-209 maps world (x,y,z) to camera
    (x/24, z/24, (18-y)/24): the table plane y=0 sits at z_cam = 0.75.
    """
    return np.array([[1.0, 0.0, 0.0, 0.0],
                     [0.0, 0.0, -1.0, 0.0],
                     [0.0, 1.0, 0.0, -18.0],
                     [0.0, 0.0, 0.0, 1.0]], dtype=np.float64)


class SyntheticEnv(object):
    """The six attributes `PlannerGD` reads from `FlexEnv`."""

    def __init__(self, config=None):
        config = config or default_config()
        self.is_real = False
        self.screenHeight = SCREEN
        self.screenWidth = SCREEN
        self.wkspc_w = float(config['dataset']['wkspc_w'])
        self.global_scale = float(config['dataset']['global_scale'])
        w = self.wkspc_w
        self.cvx_region = np.array([[-w, w, -w, w]], dtype=np.float64)

    def get_cam_params(self):
        return demo_cam_params()

    def get_cam_extrinsics(self):
        return demo_cam_extrinsics()


def make_pile(n_particles, n_batch=1, seed=0, kind='uniform'):
    """Camera-frame pile: jittered (never a lattice, top-k ties are
    unspecified in the reference), on the table plane z ~= 0.75.

    Returns s_cur [n_batch,N,3] f32, dens [n_batch] f32, attr [n_batch,N] f32.
    """
    rng = np.random.default_rng(seed)
    s = np.empty((n_batch, n_particles, 3), dtype=np.float64)
    if kind == 'uniform':
        s[..., 0] = rng.uniform(-0.2, 0.2, (n_batch, n_particles))
        s[..., 1] = rng.uniform(-0.2, 0.2, (n_batch, n_particles))
    elif kind == 'blob':
        r = 0.12 * np.sqrt(rng.uniform(0, 1, (n_batch, n_particles)))
        th = rng.uniform(0, 2 * np.pi, (n_batch, n_particles))
        s[..., 0] = r * np.cos(th)
        s[..., 1] = r * np.sin(th)
    else:
        raise ValueError('unknown pile kind: %s' % kind)
    s[..., 2] = 0.75 - rng.uniform(0, 0.01, (n_batch, n_particles))
    dens = np.full((n_batch,), n_particles / 0.16, dtype=np.float32)
    attr = np.zeros((n_batch, n_particles), dtype=np.float32)
    return s.astype(np.float32), dens, attr


def nominal_pushes(n_look_ahead, seed=0):
    """[H,4] pushes (sx,sy,ex,ey) in world units, in the style of the
    reference's init_action sets: start near the workspace rim, end inside
    the +-3.5 clip box."""
    rng = np.random.default_rng(1000 + seed)
    acts = np.empty((n_look_ahead, 4), dtype=np.float64)
    for t in range(n_look_ahead):
        ang = rng.uniform(0, 2 * np.pi)
        start = 4.5 * np.array([np.cos(ang), np.sin(ang)])
        end = rng.uniform(-2.0, 2.0, 2)
        acts[t, :2] = np.clip(start, -WKSPC_W, WKSPC_W)
        acts[t, 2:] = np.clip(end, -0.7 * WKSPC_W, 0.7 * WKSPC_W)
    return acts


def sample_pushes(n_sample, n_look_ahead, seed=0):
    """[n_sample,H,4] pushes: nominal + N(0, 0.6) noise, clipped to the
    reference's action box (planners.py:151-167)."""
    rng = np.random.default_rng(2000 + seed)
    nom = nominal_pushes(n_look_ahead, seed)
    acts = nom[None] + rng.normal(0, 0.6, (n_sample, n_look_ahead, 4))
    lo, hi = action_limits()
    return np.clip(acts, lo, hi).astype(np.float32)


def action_limits(wkspc_w=WKSPC_W):
    """Clip box of planners.py:152-155 for cvx_region [-w,w,-w,w]."""
    w = wkspc_w
    d = 2 * w * 0.15
    lo = np.array([-w, -w, -w + d, -w + d])
    hi = np.array([w, w, w - d, w - d])
    return lo, hi


def goal_mask(kind='I', size=SCREEN):
    """0/1 uint8 image, 1 on the goal shape."""
    m = np.zeros((size, size), dtype=np.uint8)
    c = size // 2
    if kind == 'I':
        m[c - 170:c + 170, c - 22:c + 22] = 1      # stem
        m[c - 170:c - 135, c - 70:c + 70] = 1      # top serif
        m[c + 135:c + 170, c - 70:c + 70] = 1      # bottom serif
    elif kind == 'disc':
        yy, xx = np.mgrid[0:size, 0:size]
        m[(yy - c) ** 2 + (xx - c) ** 2 < 75 ** 2] = 1
    else:
        raise ValueError('unknown goal kind: %s' % kind)
    return m


def goal_distance_image(mask):
    """What the reference passes as `obs_goal` (utils.py:566-579): distance
    of every pixel to the goal shape, 0 on it.  The reference uses OpenCV's
    5x5-mask approximation; here the exact Euclidean transform."""
    from scipy import ndimage
    return np.minimum(ndimage.distance_transform_edt(1 - mask), 1e4).astype(np.float32)


def goal_field(obs_goal):
    """The shifted signed field `G` sampled by the reward
    (env/flex_rewards.py:172-177), from an `obs_goal` image."""
    from scipy import ndimage
    seg = (obs_goal < 0.5)
    neg = ndimage.distance_transform_edt(seg.astype(np.uint8)).astype(np.float32)
    g = obs_goal.astype(np.float32) - neg
    return (g - g.min()).astype(np.float32)


def goal_coor_strided(obs_goal, m):
    """[M,2] (col,row) goal pixels: every k-th goal pixel (a cheap stand-in
    for the farthest-point subsample of planners.py:620-624)."""
    rc = np.argwhere(obs_goal < 0.5)
    cr = rc[:, ::-1].astype(np.float32)
    m = min(m, cr.shape[0])
    idx = np.linspace(0, cr.shape[0] - 1, m).astype(np.int64)
    return np.ascontiguousarray(cr[idx])


def render_depth(n_granules=1500, seed=0, kind='blob', size=SCREEN, grain=0.006, global_scale=GLOBAL_SCALE,
                 cam_params=None):
    """A stand-in for `pyflex.render(render_depth=True)` (env/flex_env.py:874-885): obs
    [size,size,5] float32, channels 0-2 colour in [0,255], 3 alpha, 4 depth in world units
    (camera z x global_scale).  Granules are spheres of radius `grain` (camera units) lying
    on the table plane z = 0.75, about a fifth of them stacked on a second layer."""
    rng = np.random.default_rng(seed)
    s, _, _ = make_pile(n_granules, 1, seed=seed, kind=kind)
    cx_w, cy_w = s[0, :, 0].astype(np.float64), s[0, :, 1].astype(np.float64)
    layer = (rng.uniform(0, 1, n_granules) < 0.2).astype(np.float64)
    zc = 0.75 - grain - layer * 1.6 * grain
    fx, fy, cx, cy = cam_params if cam_params is not None else demo_cam_params()
    depth = np.full((size, size), 0.75, dtype=np.float64)
    pr = int(np.ceil(fx * grain / 0.7)) + 1
    for g in range(n_granules):
        u0 = int(round(cx_w[g] * fx / 0.75 + cx))
        v0 = int(round(cy_w[g] * fy / 0.75 + cy))
        ua, ub = max(u0 - pr, 0), min(u0 + pr + 1, size)
        va, vb = max(v0 - pr, 0), min(v0 + pr + 1, size)
        if ua >= ub or va >= vb:
            continue
        uu, vv = np.meshgrid(np.arange(ua, ub), np.arange(va, vb))
        x = (uu - cx) * 0.75 / fx - cx_w[g]
        y = (vv - cy) * 0.75 / fy - cy_w[g]
        rr = grain * grain - x * x - y * y
        z = np.where(rr > 0, zc[g] - np.sqrt(np.maximum(rr, 0)), 0.75)
        depth[va:vb, ua:ub] = np.minimum(depth[va:vb, ua:ub], z)
    obs = np.zeros((size, size, 5), dtype=np.float32)
    fg = depth < 0.7499
    obs[..., 0] = np.where(fg, 200.0, 255.0)
    obs[..., 1] = np.where(fg, 120.0, 255.0)
    obs[..., 2] = np.where(fg, 40.0, 255.0)
    obs[..., 3] = 255.0
    obs[..., 4] = (depth * global_scale).astype(np.float32)
    return obs


def _relax_overlaps(xy, r_min, iters=4):
    """Jacobi passes that push apart pairs closer than r_min (each takes half of the overlap)."""
    for _ in range(iters):
        d = xy[:, None, :] - xy[None, :, :]
        dist = np.sqrt((d * d).sum(-1))
        np.fill_diagonal(dist, np.inf)
        over = np.maximum(r_min - dist, 0.0)
        if not over.any():
            break
        unit = d / np.maximum(dist, 1e-9)[..., None]
        xy = xy + 0.5 * (unit * over[..., None]).sum(1)
    return xy


def push_episode(n_particles, n_rollout, seed, kind=None):
    """A synthetic episode shaped like `ParticleDataset.__getitem__`'s return (dataset/dataset_gnn_dyn.py:
    states [T+1,n,3], states_delta [T,n,3], attrs [T+1,n], particle_num, particle_den, None), with an ANALYTIC
    stand-in for the simulator (which is closed source): the particles inside the pusher's swept band advance to the
    push end -- `states_delta`, by the reference's own formula (dataset/dataset_gnn_dyn.py:136-194, the numpy twin of
    planners.py:211-257) on the demo camera -- and then pairs left closer than 0.55 / sqrt(density) are pushed
    apart (the pile spreads ahead of the pusher), with a small positional jitter.  numpy float64 -> float32, seeded:
    the generator feeds the reference's training loop in tests/golden/make_golden_trained.py AND the device trainer
    in the tests with bit-identical batches."""
    rng = np.random.default_rng(7000 + seed)
    n = int(n_particles)
    kind = kind or ('blob' if rng.uniform() < 0.5 else 'uniform')
    s, _, _ = make_pile(n, 1, seed=7000 + seed, kind=kind)
    cur = s[0].astype(np.float64)
    area = (np.pi * 0.12 ** 2) if kind == 'blob' else 0.16
    den = float(np.clip(n / area * rng.uniform(0.7, 1.3), 15.0, 6500.0))
    r_min = 0.55 / np.sqrt(n / area)
    states = np.zeros((n_rollout + 1, n, 3))
    sdelta = np.zeros((n_rollout, n, 3))
    states[0] = cur
    w = 0.8 / 24.0
    for t in range(n_rollout):
        ang = rng.uniform(0, 2 * np.pi)
        aim = cur[rng.integers(n), :2] * [GLOBAL_SCALE, -GLOBAL_SCALE] + rng.normal(0, 0.3, 2)   # through the pile, push units
        start = aim + 3.0 * np.array([np.cos(ang), np.sin(ang)])
        end = aim - rng.uniform(0.3, 2.0) * np.array([np.cos(ang), np.sin(ang)])
        start = np.clip(start, -WKSPC_W, WKSPC_W)
        end = np.clip(end, -0.7 * WKSPC_W, 0.7 * WKSPC_W)
        # demo camera: push point (x, y) = world (x, 0, -y) -> camera (x / 24, -y / 24, 0.75)  (planners.py:192-209,231-234)
        sc = np.array([start[0], -start[1], 18.0]) / GLOBAL_SCALE
        ec = np.array([end[0], -end[1], 18.0]) / GLOBAL_SCALE
        dv = ec - sc
        length = np.linalg.norm(dv)
        dirn = dv / length
        ortho = np.array([-dirn[1], dirn[0], 0.0])
        rel = cur - sc[None]
        u, v = rel @ dirn, rel @ ortho
        hard = ((u < length) & (u > 0.0)).astype(np.float64)
        soft = np.exp(-np.maximum(np.maximum(-w - v, 0.0), np.maximum(v - w, 0.0)) / 0.01)
        to_end = (ec[None] - cur) @ dirn
        sd = to_end[:, None] * dirn[None] * hard[:, None] * soft[:, None]
        sdelta[t] = sd
        nxt = cur + sd
        nxt[:, :2] = _relax_overlaps(nxt[:, :2], r_min)
        nxt[:, :2] += rng.normal(0, 0.0005, (n, 2))
        cur = nxt
        states[t + 1] = cur
    return (states.astype(np.float32), sdelta.astype(np.float32), np.zeros((n_rollout + 1, n), np.float32), n,
            np.float32(den), None)


PUSH_BATCH_SIZES = (10, 20, 30, 50, 80, 100, 150, 200, 300)


def push_batch(iteration, batch_size=4, n_rollout=5, sizes=PUSH_BATCH_SIZES):
    """Training batch number `iteration`: `batch_size` episodes of push_episode with particle counts drawn from `sizes`,
    zero-padded to the largest as train/train_gnn_dyn.py:20-43 (`collate_fn`) pads ->
    (states [B,T+1,n_max,3], states_delta [B,T,n_max,3], attrs [B,T+1,n_max], particle_nums [B] int32, particle_dens [B])."""
    eps = []
    for j in range(batch_size):
        k = iteration * batch_size + j
        n = int(sizes[np.random.default_rng(9000 + k).integers(len(sizes))])
        eps.append(push_episode(n, n_rollout, k))
    n_max = max(e[3] for e in eps)
    states = np.zeros((batch_size, n_rollout + 1, n_max, 3), np.float32)
    sdelta = np.zeros((batch_size, n_rollout, n_max, 3), np.float32)
    attrs = np.zeros((batch_size, n_rollout + 1, n_max), np.float32)
    for j, e in enumerate(eps):
        states[j, :, :e[3]] = e[0]
        sdelta[j, :, :e[3]] = e[1]
    return (states, sdelta, attrs, np.array([e[3] for e in eps], np.int32), np.array([e[4] for e in eps], np.float32))


def pushes_through(s, seed=0):
    """[B,4] float32 pushes (sx,sy,ex,ey) that cross the piles s [B,N,3] (camera frame, demo camera): from 3 world units
    before a random particle to 0.3 ... 2 units past it, clipped to the reference's action box (planners.py:151-167)."""
    rng = np.random.default_rng(4000 + seed)
    B, N, _ = s.shape
    lo, hi = action_limits()
    acts = np.empty((B, 4), np.float64)
    for b in range(B):
        ang = rng.uniform(0, 2 * np.pi)
        d = np.array([np.cos(ang), np.sin(ang)])
        aim = s[b, rng.integers(N), :2].astype(np.float64) * [GLOBAL_SCALE, -GLOBAL_SCALE] + rng.normal(0, 0.3, 2)
        acts[b, :2] = aim + 3.0 * d
        acts[b, 2:] = aim - rng.uniform(0.3, 2.0) * d
    return np.clip(acts, lo, hi).astype(np.float32)
