"""Where the trainer's one-launch backward pass (kmb_step_bwd<dump, coop>) spends its time, from a diagnostic build:
  python __graft_entry__.py --lib tools/bin/libdrp_ts.so -DROLLOUT_STAMPS ;  DRP_LIB=tools/bin/libdrp_ts.so python tools/train_stamps.py [shape]   (tools/bin/ is git-ignored; this name travels to the GPU box)
100 MHz wall stamps of thread 0 of every workgroup, per launch."""
import ctypes
import sys
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd.engine import Engine

SHAPES = ((4, [300, 240, 150, 280]), (4, [1000, 800, 900, 600]), (32, [300] * 32))
B, nums = SHAPES[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(0)), 0.08)
rng = np.random.default_rng(0)
N, H = max(nums), 5
states = np.zeros((B, H + 1, N, 3), np.float32); sdelta = np.zeros((B, H, N, 3), np.float32)
attrs = np.zeros((B, H + 1, N), np.float32); dens = np.zeros((B,), np.float32)
for b, n in enumerate(nums):
    s, d, _ = syn.make_pile(n, 1, seed=b); dens[b] = d[0]
    for t in range(H + 1): states[b, t, :n] = s[0] + 0.003 * t * rng.standard_normal((n, 3)).astype(np.float32)
    sdelta[b, :, :n] = 0.004 * rng.standard_normal((H, n, 3)).astype(np.float32)
pn = np.asarray(nums, np.int32)
eng.train_begin(H, 1e-3, 0.9)
fn = _lib.load().drp_debug_bwd_stamps
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
for it in range(25):
    if it == 5:
        fn(out, 1)
    eng.train_step(states, sdelta, attrs, pn, dens, mode='update')
eng.sync()
fn(out, 0)
names = ['matrices -> LDS', 'phase P (predictor backward, update of step 2)', 'x3: (sum of the three barrier parts below)',
         'x3: edge terms by the workgroup -> LDS', 'x3: matrix chain of the tile (thread 0 waits for it at the next barrier)',
         'x3: own fence + __syncthreads', 'x3: arrive, wait for the other workgroups', '-']
wgs = float(out[15])
print('B=%d N<=%d: %.0f stamped workgroups (20 iterations x %d launches)' % (B, N, wgs, H))
for q, nm in enumerate(names[:7]):
    print('  %-74s %7.2f us per launch' % (nm, float(out[q]) * 0.01 / wgs))
