import sys, time
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(0)), 0.08)
rng = np.random.default_rng(0)
B, nums = 4, [300, 240, 150, 280]
N, H = max(nums), 5
states = np.zeros((B, H + 1, N, 3), np.float32); sdelta = np.zeros((B, H, N, 3), np.float32)
attrs = np.zeros((B, H + 1, N), np.float32); dens = np.zeros((B,), np.float32)
for b, n in enumerate(nums):
    s, d, _ = syn.make_pile(n, 1, seed=b); dens[b] = d[0]
    for t in range(H + 1): states[b, t, :n] = s[0] + 0.003 * t * rng.standard_normal((n, 3)).astype(np.float32)
    sdelta[b, :, :n] = 0.004 * rng.standard_normal((H, n, 3)).astype(np.float32)
pn = np.asarray(nums, np.int32)
eng.train_begin(H, 1e-3, 0.9)
for mode in ('eval', 'grad', 'update'):
    for _ in range(3): eng.train_step(states, sdelta, attrs, pn, dens, mode=mode)
    eng.sync(); t0 = time.perf_counter()
    for _ in range(20): eng.train_step(states, sdelta, attrs, pn, dens, mode=mode)
    eng.sync(); print(mode, '%.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
