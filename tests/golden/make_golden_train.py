#!/usr/bin/env python3
"""Golden vectors for the training row, captured with the reference's model.

The reference's training step is not a function: it is the body of the loop in
`train/train_gnn_dyn.py:159-210`.  This script runs that body's statements --
`model.predict_one_step`, `F.mse_loss` over the real particles of every sample, division by
`n_rollout * B`, `loss.backward()`, `torch.optim.Adam(lr, betas=(adam_beta1, 0.999)).step()` --
on the REFERENCE's `PropNetDiffDenModel` (imported from /root/reference) and a synthetic
collated batch laid out as `collate_fn` (:20-45) produces it.  Usage:
    python tests/golden/make_golden_train.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402


def make_batch(syn, planner, torch, nums, n_rollout, seed):
    """What collate_fn returns: zero-padded to max(nums); impulses from random pushes through
    the reference's gen_s_delta; targets = state + impulse + jitter."""
    rng = np.random.default_rng(seed)
    B, N = len(nums), max(nums)
    states = np.zeros((B, n_rollout + 1, N, 3), np.float32)
    sdelta = np.zeros((B, n_rollout, N, 3), np.float32)
    attrs = np.zeros((B, n_rollout + 1, N), np.float32)
    dens = np.zeros((B,), np.float32)
    for b, n in enumerate(nums):
        s, d, _ = syn.make_pile(n, 1, seed=seed * 10 + b, kind='blob' if b % 2 else 'uniform')
        dens[b] = d[0] * rng.uniform(0.6, 1.4)
        cur = s[0]
        states[b, 0, :n] = cur
        pushes = syn.sample_pushes(1, n_rollout, seed=seed * 10 + b)[0]
        for t in range(n_rollout):
            planner.particle_num = n
            with torch.no_grad():
                sd = planner.gen_s_delta(torch.from_numpy(cur[None]), torch.from_numpy(pushes[t][None].astype(np.float32)))[0].numpy()
            sd = sd + 0.002 * rng.standard_normal(sd.shape).astype(np.float32)
            sdelta[b, t, :n] = sd
            cur = (cur + 0.7 * sd + 0.0015 * rng.standard_normal(cur.shape)).astype(np.float32)
            states[b, t + 1, :n] = cur
    return states, sdelta, attrs, np.asarray(nums, np.int32), dens


def main():
    from dyn_res_pile_manip_amd import synthetic as syn
    torch, PropNetDiffDenModel, ref_planners, _ = mg.load_reference()
    import torch.nn.functional as F
    torch.set_num_threads(8)
    config = syn.default_config()
    env = syn.SyntheticEnv(config)
    planner = ref_planners.PlannerGD(config, env)
    out = {}
    for name, nums, n_rollout, seed in [('b4_r3', [40, 64, 25, 64], 3, 1), ('b2_r5', [30, 12], 5, 2)]:
        model = mg.make_model(torch, PropNetDiffDenModel, config, seed=0)
        model.train(True)
        states, sdelta, attrs, pnums, dens = make_batch(syn, planner, torch, nums, n_rollout, seed)
        lr, beta1 = 1e-3, 0.9
        optimizer = torch.optim.Adam(model.parameters(), lr=lr, betas=(beta1, 0.999))
        losses = []
        for it in range(3):
            st, sd, at = torch.from_numpy(states), torch.from_numpy(sdelta), torch.from_numpy(attrs)
            pd = torch.from_numpy(dens)
            B = st.shape[0]
            # ---- train/train_gnn_dyn.py:167-203 ----
            loss = 0.
            s_cur = st[:, 0]
            a_cur = at[:, 0]
            for idx_step in range(n_rollout):
                s_nxt = st[:, idx_step + 1]
                s_delta = sd[:, idx_step]
                s_pred = model.predict_one_step(a_cur, s_cur, s_delta, pd)
                for j in range(B):
                    loss += F.mse_loss(s_pred[j, :pnums[j]], s_nxt[j, :pnums[j]])
                s_cur = s_pred
            loss = loss / (n_rollout * B)
            # ---- :206-209 ----
            optimizer.zero_grad()
            loss.backward()
            if it == 0:
                for k, v in model.named_parameters():
                    out[name + '/grad/' + k] = v.grad.detach().numpy().copy()
            optimizer.step()
            losses.append(loss.item())
        out[name + '/states'] = states
        out[name + '/states_delta'] = sdelta
        out[name + '/attrs'] = attrs
        out[name + '/particle_nums'] = pnums
        out[name + '/particle_dens'] = dens
        out[name + '/losses'] = np.asarray(losses, np.float64)
        out[name + '/lr_beta1'] = np.array([lr, beta1])
        for k, v in model.state_dict().items():
            out[name + '/after3/' + k] = v.detach().numpy().copy()
        print(name, 'losses', losses)
    np.savez_compressed(os.path.join(HERE, 'train.npz'), **out)
    print('train.npz %.1f KB' % (os.path.getsize(os.path.join(HERE, 'train.npz')) / 1024.0))


if __name__ == '__main__':
    main()
