"""Time one training iteration (train/train_gnn_dyn.py:159-210; batch_size 4, n_rollout 5 as
config/train/gnn_dyn.yaml) on the device, next to the dense torch-autograd oracle on the host."""
import sys
import time

sys.path.insert(0, '.')
import numpy as np

from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
from oracle import propnet_dense as od

eng = Engine(0)
sd = weights.random_state_dict(0)
eng.load_weights(weights.blob_from_state_dict(sd), 0.08)
W = {k: np.asarray(v) for k, v in sd.items()}
rng = np.random.default_rng(0)
SHAPES = ((4, [300, 240, 150, 280]), (4, [1000, 800, 900, 600]), (32, [300] * 32))
if len(sys.argv) > 1:                      # one shape only (profiling): its index
    SHAPES = (SHAPES[int(sys.argv[1])],)
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for B, nums in SHAPES:
    N, H = max(nums), 5
    states = np.zeros((B, H + 1, N, 3), np.float32)
    sdelta = np.zeros((B, H, N, 3), np.float32)
    attrs = np.zeros((B, H + 1, N), np.float32)
    dens = np.zeros((B,), np.float32)
    for b, n in enumerate(nums):
        s, d, _ = syn.make_pile(n, 1, seed=b)
        dens[b] = d[0]
        for t in range(H + 1):
            states[b, t, :n] = s[0] + 0.003 * t * rng.standard_normal((n, 3)).astype(np.float32)
        sdelta[b, :, :n] = 0.004 * rng.standard_normal((H, n, 3)).astype(np.float32)
    pn = np.asarray(nums, np.int32)
    eng.train_begin(H, 1e-3, 0.9)
    for _ in range(2):
        eng.train_step(states, sdelta, attrs, pn, dens, mode='update')
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(ITERS):
        loss, _ = eng.train_step(states, sdelta, attrs, pn, dens, mode='update')
    eng.sync()
    ms = (time.perf_counter() - t0) / ITERS * 1e3
    for _ in range(2):
        eng.train_step(states, sdelta, attrs, pn, dens, mode='eval')
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.train_step(states, sdelta, attrs, pn, dens, mode='eval')
    eng.sync()
    ms_eval = (time.perf_counter() - t0) / 10 * 1e3
    line = 'B=%d N<=%d n_rollout=%d: %.2f ms per training iteration (upload, 5 steps forward, backward, weight gradients, Adam, re-pack); %.2f ms forward-only' % (B, N, H, ms, ms_eval)
    if B * N <= 1200 and len(sys.argv) <= 1:
        t0 = time.perf_counter()
        od.train_loss_and_grads(W, states, sdelta, attrs, pn, dens)
        line += '; dense torch autograd on the host: %.0f ms' % ((time.perf_counter() - t0) * 1e3)
    print(line, flush=True)
