"""Time the per-observation and per-goal host work of the reference moved to the device
(rows f2 and f3), next to the numpy oracle on the host cores:
  obs2ptcl_fixed_num_batch (env/flex_env.py:933-951): 720x720 depth -> 30 x N particles
  goal field + goal pixel subsample (env/flex_rewards.py:172-177, planners.py:620-624)."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

from dyn_res_pile_manip_amd import synthetic as syn
from dyn_res_pile_manip_amd.engine import Engine
from oracle import goal as og
from oracle import particles as op

eng = Engine(0)
cam = syn.demo_cam_params()
obs = syn.render_depth(5000, seed=2, kind='uniform')
depth_raw = np.ascontiguousarray(obs[..., -1])


def timeit(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


for N in (50, 150, 300, 600):
    starts = np.arange(30) * 7
    ms = timeit(lambda: eng.obs2ptcl(depth_raw, 24.0, cam, N, 30, init_idx=starts), 10)
    t0 = time.perf_counter()
    op.obs2ptcl_fixed_num_batch(depth_raw, 24.0, cam, N, starts[:3])
    cpu = (time.perf_counter() - t0) * 1e3
    print('obs2ptcl N=%3d batch 30: device %.2f ms (upload + chain + download); numpy oracle %.0f ms for 3 of 30 '
          '(clouds computed once; the reference recomputes them 30 times)' % (N, ms, cpu))

obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
for mode in ('cv5', 'exact'):
    ms = timeit(lambda: eng.set_goal_image(obs_goal, 1500, 0, mode), 10)
    t0 = time.perf_counter()
    og.goal_field(obs_goal, mode)
    px = og.goal_pixels(obs_goal)
    t1 = time.perf_counter()
    op.fps_np(px, 1500, 0)
    t2 = time.perf_counter()
    print('goal image 720x720 -> field (%s) + 1500 of %d goal pixels: device %.2f ms; numpy oracle field %.0f ms + '
          'fps_np %.0f ms' % (mode, px.shape[0], ms, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
src = (1 - syn.goal_mask('I')).astype(np.uint8)
for mode in ('cv5', 'exact'):
    print('distance transform alone (%s): %.2f ms' % (mode, timeit(lambda: eng.distance_transform(src, mode), 10)))
