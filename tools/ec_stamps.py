"""Slot loops with and without the edge-chain cache (prop_tiles, EC), from a diagnostic build:
  hipcc ... -DPROP_STAMPS -o ab/libdrp_stamps.so ;  DRP_LIB=ab/libdrp_stamps.so DRP_NO_ROLLOUT_FUSED=1 python tools/ec_stamps.py N [samples]
(km_prop3 keeps the stamps; DRP_ECACHE_MAX_MB=0 for the recomputing kernels)."""
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
lib = _lib.load()
fn = lib.drp_debug_prop_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
for it in range(40):
    if it == 10:
        fn(None, out, 1)
        eng.probe_begin('prop')
    eng.mpc_sample(it); eng.mpc_rollout(False); eng.mpc_update_device()
eng.sync()
ms, n = eng.probe_read()
fn(None, out, 0)
c_slot, slots, c_node, c_tile, ticks, tiles, c_slot2, c_all = [float(out[i]) for i in range(8)]
ghz = c_tile / (ticks * 10.0)
print('%d x %d: %d launches of the prop class, %.1f us each; clock %.2f GHz' % (ns, N, n, ms / n * 1e3, ghz))
print('  tiles %.0f, slot iterations %.2f per tile' % (tiles, slots / tiles))
if c_slot2 > 0:
    print('  chain slots   : %.0f cycles = %.2f us each (a third of the slots)' % (c_slot / (slots / 3), c_slot / (slots / 3) / ghz * 1e-3))
    print('  cached slots  : %.0f cycles = %.2f us each' % (c_slot2 / (slots * 2 / 3), c_slot2 / (slots * 2 / 3) / ghz * 1e-3))
else:
    print('  chain slots   : %.0f cycles = %.2f us each' % (c_slot / slots, c_slot / slots / ghz * 1e-3))
print('  node part     : %.0f cycles = %.2f us per tile' % (c_node / tiles, c_node / tiles / ghz * 1e-3))
print('  whole tile    : %.0f cycles = %.2f us (head, slots, node part)' % (c_tile / tiles, c_tile / tiles / ghz * 1e-3))
