"""In-kernel clock and cycles per slot iteration of km_prop, from a diagnostic build:
  hipcc ... -DPROP_STAMPS -o ab/libdrp_stamps.so ;  DRP_LIB=ab/libdrp_stamps.so python tools/prop_stamps.py
(the stamps cost a few per cent themselves; the product build has none)."""
import ctypes
import os
import sys
os.environ.setdefault('DRP_NO_PROP3', '1')     # the budget of ONE propagation step per launch (km_prop); unset it by DRP_NO_PROP3= for km_prop3
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine

N, ns, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 300), (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 10
if os.environ.get('DRP_NO_PROP3') == '':
    del os.environ['DRP_NO_PROP3']
eng = Engine(0)
eng.set_engine(_lib.ENGINES['fused'])
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
print('km_prop: %d launches, %.1f us each (HIP events)' % (n, ms / n * 1e3))
c_slot, slots, c_node, c_tile, ticks, tiles = [float(out[i]) for i in range(6)]
ghz = c_tile / (ticks * 10.0)
print('tiles %.0f  slot iterations %.0f (%.2f per tile)' % (tiles, slots, slots / tiles))
print('in-kernel clock %.2f GHz' % ghz)
print('per wave: %.0f shader cycles per slot iteration, %.0f per node part, %.0f per tile' % (c_slot / slots, c_node / tiles, c_tile / tiles))
print('sum of tile cycles / 2048 waves / launch = %.0f cycles = %.1f us at that clock' % (c_tile / 2048 / n, c_tile / 2048 / n / ghz * 1e-3))
print('per wave and launch: entry -> weights in LDS %.0f cycles (%.1f us), tile loop %.0f cycles (%.1f us)' % (float(out[6]) / 2048 / n, float(out[6]) / 2048 / n / ghz * 1e-3, float(out[7]) / 2048 / n, float(out[7]) / 2048 / n / ghz * 1e-3))
print('per SIMD (two waves): %.0f cycles = %.2f us per slot iteration; MFMA pipe 78 x 32 = 2496' % (c_slot / slots / 2, c_slot / slots / 2 / ghz * 1e-3))

sp = (ctypes.c_ulonglong * 4096)()
f2 = lib.drp_debug_prop_span
f2.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
f2(None, sp, 4096)
a = np.array(sp[:], dtype=np.float64).reshape(2048, 2) * 0.01     # us
t0 = a[:, 0].min()
st, en = a[:, 0] - t0, a[:, 1] - t0
print('last launch: waves start %.1f .. %.1f us, end %.1f (first) / %.1f (median) / %.1f (last) us; mean lifetime %.1f us'
      % (st.min(), st.max(), en.min(), np.median(en), en.max(), (en - st).mean()))
wg_end = en.reshape(256, 8).max(1)
print('workgroup ends: min %.1f  median %.1f  max %.1f us;  by blockIdx %% 8: %s' % (wg_end.min(), np.median(wg_end), wg_end.max(),
      ' '.join('%.0f' % wg_end[g::8].mean() for g in range(8))))
