"""Where a rollout step of the whole-sample kernels spends its time, from a diagnostic build:
  hipcc ... -DROLLOUT_STAMPS -o ab/libdrp_rstamps.so ;  DRP_LIB=ab/libdrp_rstamps.so python tools/rollout_stamps.py N [samples]
100 MHz wall stamps of wave 0 of every 32nd workgroup (km_rollout; with DRP_NO_ROLLOUT_FUSED=1: km_prop3's phases only)."""
import ctypes
import sys
sys.path.insert(0, '.')
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine

N, ns, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 50), (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 10
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(seed=0)), 0.08)
eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
eng.set_goal_image(syn.goal_distance_image(syn.goal_mask('I')), 5 * N, fps_init=0, mode='cv5', want=False)
s0, dens, attr = syn.make_pile(N, 1, seed=0)
lo, hi = syn.action_limits()
eng.mpc_begin(s0, attr, dens, syn.nominal_pushes(H, seed=0), n_sample=ns, sigma=0.6, beta_filter=0.7,
              reward_weight=0.1, act_lo=lo, act_hi=hi, seed=1234, sample_offset=0)
fn = _lib.load().drp_debug_roll_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
iters = 30
for it in range(10 + iters):
    if it == 10:
        fn(out, 1)
        eng.probe_begin('prop')
    eng.mpc_sample(it); eng.mpc_rollout(False); eng.mpc_update_device()
eng.sync()
ms, n = eng.probe_read()
fn(out, 0)
names = ['wait at step end', 'impulses + positions', 'neighbour lists', 'encoder weights -> LDS', 'encoder tiles', 'wait after encoder',
         'edge weights -> LDS + row order', 'propagation tiles (x3)', 'wait at propagation step end (x2)']
wgs = len(range(0, (ns + (-(-ns // 256)) - 1) // (-(-ns // 256)), 32))
steps = float(out[15]) if out[15] else float(iters * H * wgs)
print('%d particles x %d samples: prop class %d launches, %.1f us each; %.0f stamped steps' % (N, ns, n, ms / n * 1e3, steps))
tot = 0.0
for q, nm in enumerate(names):
    us = float(out[q]) * 0.01 / steps
    tot += us
    print('  %-36s %7.2f us per rollout step' % (nm, us))
print('  %-36s %7.2f us' % ('sum', tot))
print('  of the encoder tiles: wave 0\'s share of the lists %.2f us' % (float(out[14]) * 0.01 / steps))
print('  wave 0 tiles by propagation step     %.2f / %.2f / %.2f us' % tuple(float(out[11 + q]) * 0.01 / steps for q in range(3)))
if out[10]:
    print('  in-kernel shader clock %.3f GHz' % (float(out[9]) / (float(out[10]) * 10.0)))
