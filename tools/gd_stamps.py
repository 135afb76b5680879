"""Where the GD planner's forward kernel (km_prop3 with the tape) spends a rollout step, from a diagnostic build:
  hipcc ... -DROLLOUT_STAMPS -o ab/libdrp_rstamps.so ;  DRP_LIB=ab/libdrp_rstamps.so python tools/gd_stamps.py N [rows per column]
100 MHz wall stamps of wave 0 of every 32nd workgroup (the phases of prop3_step)."""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(0)), 0.08)
eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
lo, hi = syn.action_limits()
eng.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
s0, dens, attr = syn.make_pile(N, 30, seed=N)
acts = np.repeat(np.stack([syn.nominal_pushes(1, seed=i) for i in range(50)]), 30, axis=0).astype(np.float32)
eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
fn = _lib.load().drp_debug_roll_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
iters = 30
for it in range(5 + iters):
    if it == 5:
        eng.sync()
        fn(out, 1)
        eng.probe_begin('prop')
    eng.lib.drp_gd_step(eng.h, None)
eng.sync()
ms, n = eng.probe_read()
fn(out, 0)
names = ['-', '-', '-', 'encoder weights -> LDS', 'encoder tiles', 'wait after encoder', 'edge weights -> LDS + row order',
         'propagation tiles (x3)', 'wait at propagation step end (x2)']
wgs = len(range(0, -(-1500 // (-(-1500 // 256))), 32))
steps = float(iters * wgs)
print('%d particles x 1500 rows: prop class %d launches, %.1f us each' % (N, n, ms / n * 1e3))
tot = 0.0
for q, nm in enumerate(names):
    if nm == '-':
        continue
    us = float(out[q]) * 0.01 / steps
    tot += us
    print('  %-36s %7.2f us per launch' % (nm, us))
print('  %-36s %7.2f us (launch %.1f us: the rest is the resident fill and the launch itself)' % ('sum', tot, ms / n * 1e3))
print('  wave 0 tiles by propagation step     %.2f / %.2f / %.2f us' % tuple(float(out[11 + q]) * 0.01 / steps for q in range(3)))
