"""Where an obs2ptcl_fixed_num_batch call (env/flex_env.py:933-951 through dyn_res_pile_manip_amd/utils.py) spends its time:
the reference's own asserts on the host, the contiguous copy of the depth channel, the device chain (drp_obs2ptcl: upload,
foreground compaction, voxel down-sample, 30 farthest-point samplings, recentering, download).  usage: python tools/particles_timing.py"""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

from dyn_res_pile_manip_amd import synthetic as syn, utils as dev
from dyn_res_pile_manip_amd.engine import Engine

eng = Engine(0)
dev.set_engine(eng)
cam = syn.demo_cam_params()
obs = syn.render_depth(4000, seed=1, kind='uniform')


def timeit(fn, n=20):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t))


def asserts_reference_style():
    assert obs[..., :3].max() <= 255.0 and obs[..., :3].min() >= 0.0 and obs[..., :3].max() >= 1.0
    assert obs[..., -1].max() >= 0.7 * 24 and obs[..., -1].max() <= 0.8 * 24


def asserts_one_pass():
    mx, mn = dev._channel_extrema(obs)
    assert mx[:3].max() <= 255.0 and mn[:3].min() >= 0.0 and mx[:3].max() >= 1.0 and 0.7 * 24 <= mx[4] <= 0.8 * 24


print('asserts as the reference writes them (5 strided reductions): %.2f ms' % timeit(asserts_reference_style))
print('asserts in two passes over the contiguous image:             %.2f ms' % timeit(asserts_one_pass))
print('contiguous float32 copy of the depth channel:                %.2f ms' % timeit(lambda: np.ascontiguousarray(obs[..., -1], dtype=np.float32)))
depth = np.ascontiguousarray(obs[..., -1], dtype=np.float32)
for N in (20, 50, 100, 300):
    starts = np.arange(30) * 7
    print('N = %3d: drp_obs2ptcl (upload + chain + download) %.2f ms; the mirror utils.obs2ptcl_fixed_num_batch %.2f ms' % (
        N, timeit(lambda: eng.obs2ptcl(depth, 24.0, cam, N, 30, init_idx=starts)),
        timeit(lambda: dev.obs2ptcl_fixed_num_batch(obs, N, 30, cam, 24.0, init_idx=starts))))
