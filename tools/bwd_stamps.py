"""Where kmb_rows_bwd (the GD planner's backward pass, piles of up to 256 particles) spends its time, from a diagnostic build:
  hipcc ... -DROLLOUT_STAMPS -o ab/libdrp_rstamps.so ;  DRP_LIB=ab/libdrp_rstamps.so python tools/bwd_stamps.py N [rows]
100 MHz wall stamps of wave 0 of every 32nd workgroup, per group of samples."""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(0)), 0.08)
eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
lo, hi = syn.action_limits()
eng.set_goal(syn.goal_field(obs_goal), syn.goal_coor_strided(obs_goal, 5 * N))
s0, dens, attr = syn.make_pile(N, 30, seed=N)
acts = np.repeat(np.stack([syn.nominal_pushes(1, seed=i) for i in range(50)]), 30, axis=0).astype(np.float32)
eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
fn = _lib.load().drp_debug_bwd_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
for it in range(25):
    if it == 5:
        fn(out, 1)
    eng.gd_step()
eng.sync()
fn(out, 0)
names = ['wait for the group before + predictor matrices -> LDS', 'phase P (predictor backward, update of step 2)',
         'barrier, g_agg rows -> LDS, barrier', 'x3: loads, receiver term, W_r^T', 'x3: sender term (reversed-list gather)',
         'x3: W_s^T, update, W_agg^T', 'x3: barrier, rows -> LDS / encoder matrices, barrier', 'particle encoder backward']
groups = float(out[15])
print('%d particles x 1500 rows: %.0f stamped groups (20 iterations)' % (N, groups))
tot = 0.0
for q, nm in enumerate(names):
    us = float(out[q]) * 0.01 / groups
    tot += us
    print('  %-56s %7.2f us per group' % (nm, us))
print('  %-56s %7.2f us' % ('sum', tot))
