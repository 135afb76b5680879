"""Soak of the trainer: thousands of update iterations over batches of every size the one-launch backward pass and the stage
kernels take (1 ... 8 samples of 10 ... 300 particles, rollouts of 1 ... 5 steps), contexts created and destroyed in between.
Every loss finite, the loss of a fixed held-out batch not above where it started, free device memory flat from context to context."""
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine

free0 = torch.cuda.mem_get_info()[0]
t0 = time.time()
total = 0
for rep in range(8):
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(rep)), 0.08)
    H = [5, 3, 1, 5][rep % 4]
    rng = np.random.default_rng(rep)
    batches = []
    for k in range(24):
        B = int(rng.choice([1, 2, 4, 8]))
        batches.append(syn.push_batch(100 * rep + k, B, H))
    held = syn.push_batch(9999, 4, H)
    eng.train_begin(H, 1e-3, 0.9)
    first, _ = eng.train_step(*held, mode='eval')
    for it in range(750):
        loss, _ = eng.train_step(*batches[it % len(batches)], mode='update')
        assert np.isfinite(loss), (rep, it, loss)
        total += 1
    last, _ = eng.train_step(*held, mode='eval')
    ran = eng.last_dispatch()
    print('H=%d: held-out loss %.4e -> %.4e; variants seen: %s' % (H, first, last, [v for v in ran if v.startswith('train:')]), flush=True)
    assert np.isfinite(last) and last < 1.2 * first          # (750 iterations at 1e-3: the five-step loss has barely begun to fall)
    eng.close()
    torch.cuda.synchronize()
    print('   free HBM %.1f MB' % (torch.cuda.mem_get_info()[0] / 1e6), flush=True)
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print('%d update iterations in %.1f s; free HBM before %.1f MB after %.1f MB' % (total, time.time() - t0, free0 / 1e6, free1 / 1e6))
# the first contexts leave the HIP runtime's own pools behind (kernel scratch, staging): flat from then on
