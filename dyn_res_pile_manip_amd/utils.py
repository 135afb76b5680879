"""Host mirror of the reference's particle-extraction helpers (`utils.py`, and the two
`FlexEnv.obs2ptcl_fixed_num*` methods of `env/flex_env.py:917-951`), same names and argument
order, computed on the MI355X through the C ABI (include/drp.h, row f2).  Nothing here
computes on the host: without libdrp.so / a GPU every call raises.

The reference draws the sampler's start from a global generator (dgl: torch's, fps_np:
numpy's).  Here `init_idx=-1` draws from `numpy.random` too, so seeding numpy makes a run
reproducible; pass explicit indices for bit-for-bit comparisons.
"""
import numpy as np

from .engine import default_engine, set_default_engine

FG_DEPTH = 0.599 / 0.8      # env/flex_env.py:945
VOXEL = 0.01                # env/flex_env.py:947
RECENTER_R = 0.02           # env/flex_env.py:949


def get_engine(device=0):
    """The process's one context (engine.default_engine): the model, the planner bound to it and these helpers share it."""
    return default_engine(device)


def set_engine(engine):
    set_default_engine(engine)


def depth2fgpcd(depth, mask, cam_params):
    """utils.py:491-506 -> [n,3] float64."""
    return get_engine().depth2fgpcd(depth, mask, cam_params)


def downsample_pcd(pcd, voxel_size):
    """utils.py:533-544 (open3d voxel_down_sample; voxels in ascending index order)."""
    return get_engine().downsample_pcd(pcd, voxel_size)


def fps(pcd, particle_num, init_idx=-1):
    """utils.py:423-436 -> (pcd_fps [N,3] float32, particle_r)."""
    pcd = np.asarray(pcd, dtype=np.float64)
    if init_idx == -1:
        init_idx = np.random.randint(pcd.shape[0])
    pts, r = get_engine().fps_pcd(pcd, particle_num, [init_idx])
    return pts[0], float(r[0])


def fps_rad(pcd, radius, init_idx=-1):
    """utils.py:438-449 -> pcd_fps [m,3] float64 (m decided by the radius)."""
    pcd = np.asarray(pcd, dtype=np.float64)
    if init_idx == -1:
        init_idx = np.random.randint(pcd.shape[0])
    return get_engine().fps_rad(pcd, radius, init_idx)[0]


def fps_np(pcd, particle_num, init_idx=-1):
    """utils.py:451-466 for 2-D / 3-D float32 point lists -> (pcd[chosen], dist.max())."""
    pcd = np.asarray(pcd)
    if init_idx == -1:
        init_idx = np.random.randint(pcd.shape[0])
    pts, md, _ = get_engine().fps(pcd, particle_num, init_idx)
    return pts, md


def recenter(pcd, sampled_pcd, r=0.02):
    """utils.py:468-477 -> [N,3] float32."""
    s = np.asarray(sampled_pcd, dtype=np.float32)
    return get_engine().recenter(pcd, s[None], [r])[0]


def _channel_extrema(obs):
    """Per-channel maximum and minimum of an [h, w, c] image in two passes over the CONTIGUOUS array: rows of 64 pixels are
    reduced along axis 0 (a long contiguous inner dimension: vectorised), the 64 x c remainders on their own -- eight times
    faster than one reduction per strided channel view."""
    c = obs.shape[-1]
    a = np.ascontiguousarray(obs).reshape(-1, c)
    n = a.shape[0] // 64 * 64
    mx, mn = a[n:].max(0, initial=-np.inf), a[n:].min(0, initial=np.inf)
    if n:
        w = a[:n].reshape(-1, 64 * c)
        mx = np.maximum(mx, w.max(0).reshape(64, c).max(0))
        mn = np.minimum(mn, w.min(0).reshape(64, c).min(0))
    return mx, mn


def obs2ptcl_fixed_num_batch(obs, particle_num, batch_size, cam_params, global_scale, init_idx=None):
    """env/flex_env.py:933-951 with `self.get_cam_params()` / `self.global_scale` as arguments
    -> (batch_sampled_ptcl [batch,N,3] float64, batch_particle_r [batch])."""
    assert type(obs) == np.ndarray
    assert obs.shape[-1] == 5
    # the reference's range checks (env/flex_env.py:903-909), from ONE maximum and ONE minimum pass over the contiguous image
    # instead of five reductions over strided channel views (tools/particles_timing.py)
    ch_max, ch_min = _channel_extrema(obs)
    assert ch_max[:3].max() <= 255.0
    assert ch_min[:3].min() >= 0.0
    assert ch_max[:3].max() >= 1.0
    assert ch_max[4] >= 0.7 * global_scale
    assert ch_max[4] <= 0.8 * global_scale
    eng = get_engine()
    depth_raw = np.ascontiguousarray(obs[..., -1], dtype=np.float32)
    if init_idx is None:
        # the cloud size is only known on the device: draw the starts there from a numpy-drawn seed
        ptcl, r, _ = eng.obs2ptcl(depth_raw, global_scale, cam_params, particle_num, batch_size,
                                  seed=int(np.random.randint(0, 2 ** 31 - 1)))
    else:
        ptcl, r, _ = eng.obs2ptcl(depth_raw, global_scale, cam_params, particle_num, batch_size, init_idx=init_idx)
    return ptcl, r


def obs2ptcl_fixed_num(obs, particle_num, cam_params, global_scale, init_idx=None):
    """env/flex_env.py:917-931 -> (sampled_ptcl [N,3] float32, particle_r)."""
    ptcl, r = obs2ptcl_fixed_num_batch(obs, particle_num, 1, cam_params, global_scale,
                                       None if init_idx is None else [init_idx])
    return ptcl[0].astype(np.float32), float(r[0])
