#!/usr/bin/env python3
"""Golden vectors on a TRAINED network, captured with the reference's own model, planner and reward.

The published checkpoint (`scripts/download_model.sh:5`) is not obtainable offline, and every other fixture in this
directory uses seeded default-init weights with the predictor's last layer scaled x 0.02.  Here the REFERENCE's
`PropNetDiffDenModel` (imported from /root/reference) is trained on the CPU with the statements of the reference's loop
body (`train/train_gnn_dyn.py:159-210`: 5-step autoregressive rollout, per-sample masked `F.mse_loss`, division by
`n_rollout * B`, `torch.optim.Adam(lr, betas=(adam_beta1, 0.999))`; `config/train/gnn_dyn.yaml`: lr 1e-3, beta1 0.9,
batch 4, n_rollout 5) on synthetic push episodes (`dyn_res_pile_manip_amd.synthetic.push_episode`: an analytic stand-in
for the closed-source simulator; variable particle counts 10 ... 300, zero-padded as `collate_fn` :20-43 does) --
NO scaling of any layer.  Then, on those weights, the reference produces

  weights_trained.npz   the trained state_dict
  trained.npz           one_step/<n20|n50|n100|n300>   predict_one_step with the lists derived from its Rr / Rs
                        rollout/<...>                  10-step free-running ptcl_model_rollout (+ all-step rewards)
                        grad/<...>                     autograd gradients of the GD planner's loss w.r.t. the pushes
                        gd/<...>                       trajectory_optimization_ptcl_multi_traj dicts (Adam iterations)
  train_curve.npz       the first CURVE_ITERS losses of the run above with its initial weights, the per-batch checksums
                        of the episodes (the tests regenerate the batches from the same seeded generator), the
                        weights after CURVE_ITERS iterations, and the losses of a second reference run started one ulp
                        away (how far two fp32 runs of this loop drift apart: the tests' yardstick)

Runs ONLY in the build container.  Usage:  python tests/golden/make_golden_trained.py [--iters N]
(about 20 minutes on 8 cores at the default 3 000 iterations).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402

TRAIN_SEED = 1
TRAIN_ITERS = 3000
CURVE_ITERS = 240
BATCH = 4
N_ROLLOUT = 5
LR, BETA1 = 1e-3, 0.9


def main():
    from dyn_res_pile_manip_amd import synthetic as syn
    torch, PropNetDiffDenModel, ref_planners, config_reward_ptcl = mg.load_reference()
    import torch.nn.functional as F
    torch.set_num_threads(8)
    iters = TRAIN_ITERS
    if '--iters' in sys.argv:
        iters = int(sys.argv[sys.argv.index('--iters') + 1])
    config = syn.default_config()
    env = syn.SyntheticEnv(config)
    planner = ref_planners.PlannerGD(config, env)

    if '--cases-only' in sys.argv:
        # development shortcut: the committed weights_trained.npz into the reference model, then only the cases below (the
        # default path retrains and gives the same bytes)
        model = PropNetDiffDenModel(config, False)
        w = np.load(os.path.join(HERE, 'weights_trained.npz'))
        model.load_state_dict({k[2:]: torch.from_numpy(w[k]) for k in w.files if k.startswith('w/')})
        model.eval()
        t0 = time.time()
        make_cases(torch, model, planner, syn, t0)
        return

    # ---- training: the reference's loop body on synthetic episodes ------------------------------------------------------------
    torch.manual_seed(TRAIN_SEED)
    model = PropNetDiffDenModel(config, False)
    model.train(True)
    init_sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    optimizer = torch.optim.Adam(model.parameters(), lr=LR, betas=(BETA1, 0.999))
    losses, sums = [], []
    after_curve = None
    t0 = time.time()
    for it in range(iters):
        states, sdelta, attrs, pnums, dens = syn.push_batch(it, BATCH, N_ROLLOUT)
        st, sd, at, pd = (torch.from_numpy(x) for x in (states, sdelta, attrs, dens))
        B = st.shape[0]
        # ---- train/train_gnn_dyn.py:167-203 ----
        loss = 0.
        s_cur = st[:, 0]
        a_cur = at[:, 0]
        for idx_step in range(N_ROLLOUT):
            s_nxt = st[:, idx_step + 1]
            s_delta = sd[:, idx_step]
            s_pred = model.predict_one_step(a_cur, s_cur, s_delta, pd)
            for j in range(B):
                loss += F.mse_loss(s_pred[j, :pnums[j]], s_nxt[j, :pnums[j]])
            s_cur = s_pred
        loss = loss / (N_ROLLOUT * B)
        # ---- :206-209 ----
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
        sums.append(float(states.astype(np.float64).sum() + sdelta.astype(np.float64).sum() + dens.astype(np.float64).sum()))
        if it + 1 == CURVE_ITERS:
            after_curve = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        if it % 100 == 0 or it + 1 == iters:
            print('[train] %5d  rmse %.5f (mean of last 50: %.5f)  %.0f s' %
                  (it, np.sqrt(losses[-1]), np.sqrt(np.mean(losses[-50:])), time.time() - t0), flush=True)
    model.eval()

    # How far apart do two fp32 runs of this very loop drift?  The reference again from the same initial weights moved by ONE
    # ulp in one small layer (particle_encoder.model.0.weight), the same batches: the tests hold the device trainer's loss
    # curve to a small multiple of THIS deviation, not to an invented tolerance.
    twin = PropNetDiffDenModel(config, False)
    twin.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in init_sd.items()})
    with torch.no_grad():
        w0 = twin.model.particle_encoder.model[0].weight
        w0.copy_(torch.nextafter(w0, torch.full_like(w0, float('inf'))))
    twin.train(True)
    twin_opt = torch.optim.Adam(twin.parameters(), lr=LR, betas=(BETA1, 0.999))
    twin_losses = []
    for it in range(min(CURVE_ITERS, iters)):
        states, sdelta, attrs, pnums, dens = syn.push_batch(it, BATCH, N_ROLLOUT)
        st, sd, at, pd = (torch.from_numpy(x) for x in (states, sdelta, attrs, dens))
        loss = 0.
        s_cur = st[:, 0]
        for idx_step in range(N_ROLLOUT):
            s_pred = twin.predict_one_step(at[:, 0], s_cur, sd[:, idx_step], pd)
            for j in range(st.shape[0]):
                loss += F.mse_loss(s_pred[j, :pnums[j]], st[j, idx_step + 1, :pnums[j]])
            s_cur = s_pred
        loss = loss / (N_ROLLOUT * st.shape[0])
        twin_opt.zero_grad()
        loss.backward()
        twin_opt.step()
        twin_losses.append(loss.item())
    dev = np.abs(np.asarray(twin_losses) / np.asarray(losses[:len(twin_losses)]) - 1)
    print('[train] one-ulp twin: loss deviation median %.2e, max %.2e' % (np.median(dev), dev.max()), flush=True)

    tc = {'init/' + k: v for k, v in init_sd.items()}
    tc['losses_twin'] = np.asarray(twin_losses, np.float64)
    tc.update({'after/' + k: v for k, v in after_curve.items()})
    tc['losses'] = np.asarray(losses[:CURVE_ITERS], np.float64)
    tc['batch_sums'] = np.asarray(sums[:CURVE_ITERS], np.float64)
    tc['hyper'] = np.array([LR, BETA1, BATCH, N_ROLLOUT], np.float64)
    np.savez_compressed(os.path.join(HERE, 'train_curve.npz'), **tc)

    np.savez(os.path.join(HERE, 'weights_trained.npz'),
             meta=np.array('seed %d default init, NO layer scaled; %d Adam iterations of the reference loop body on '
                           'synthetic.push_batch episodes; final rmse %.5f' % (TRAIN_SEED, iters, np.sqrt(np.mean(losses[-50:])))),
             **mg.state_dict_arrays(model))

    make_cases(torch, model, planner, syn, t0)


def decision_margin(pos, thr=np.float32(0.0064)):
    """How far the graph of model/gnn_dyn.py:223-237 is from changing, for displaced positions pos [B,N,3]: the smallest
    |d - adj_thresh^2| over all pairs and the smallest gap between a receiver's 10th and 11th nearest sender inside the radius
    (squared camera-frame units).  A free-running comparison of two fp32 implementations is meaningful only while this stays
    above their rounding noise (positions differ by ~1e-7, squared distances by ~2e-8 at the radius): SURVEY.md 7, hard part 1."""
    d = ((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1)
    m = float(np.abs(d - thr).min())
    if d.shape[1] > 10:
        ds = np.sort(d, axis=2)
        inside = ds[:, :, 9] < thr
        if inside.any():
            m = min(m, float(np.abs(ds[:, :, 9] - ds[:, :, 10])[inside].min()))
    return m


def rollout_margin(torch, planner, s0, acts, state_pred):
    """min over steps of decision_margin on the reference's own trajectory (row = sample * n_batch + column)."""
    B, H = acts.shape[:2]
    prev = np.tile(s0, (B // s0.shape[0], 1, 1))
    m = np.inf
    for t in range(H):
        with torch.no_grad():
            sd = planner.gen_s_delta(torch.from_numpy(prev), torch.from_numpy(acts[:, t])).numpy()
        m = min(m, decision_margin((prev + sd).astype(np.float32)))
        prev = state_pred[:, t]
    return m


def make_cases(torch, model, planner, syn, t0):
    # ---- cases on the trained weights ---------------------------------------------------------------------------------------------
    out = {}
    cap = mg.Capture(model)
    mask = syn.goal_mask('I')
    obs_goal = syn.goal_distance_image(mask)
    lo, hi = syn.action_limits()
    SIZES = [('n20', 20), ('n50', 50), ('n100', 100), ('n300', 300)]

    for name, N in SIZES:
        # one step: four samples, pushes through the pile
        B = 4
        s, dens, attr = syn.make_pile(N, n_batch=B, seed=70 + N, kind='blob' if N <= 50 else 'uniform')
        dens = (dens * np.linspace(0.8, 1.2, B)).astype(np.float32)
        acts = syn.pushes_through(s, seed=70 + N)
        planner.particle_num = N
        s_t = torch.from_numpy(s)
        with torch.no_grad():
            s_delta = planner.gen_s_delta(s_t, torch.from_numpy(acts))
            s_pred = model.predict_one_step(torch.from_numpy(attr), s_t, s_delta, torch.from_numpy(dens))
        nbr_idx, nbr_cnt, _ = mg.edges_from_onehot(cap.rec['Rr'], cap.rec['Rs'])
        p = 'one_step/' + name + '/'
        out[p + 's_cur'], out[p + 's_delta'], out[p + 'attr'], out[p + 'dens'] = s, s_delta.numpy(), attr, dens
        out[p + 'action'] = acts
        out[p + 's_pred'] = s_pred.numpy()
        out[p + 'nbr_idx'], out[p + 'nbr_cnt'] = nbr_idx.astype(np.int16), nbr_cnt.astype(np.uint8)
        out[p + 'max_relation_hidden'] = np.array(max(float(t.abs().max()) for t in cap.rec['relation_encoder']))
        out[p + 'max_particle_effect'] = np.array(max(float(t.abs().max()) for t in cap.rec['particle_propagator']))
        out[p + 'max_relation_effect'] = np.array(max(float(t.abs().max()) for t in cap.rec['relation_propagator']))

        # 10-step free-running rollout, two batch columns x three push sequences.  A trained network moves the pushed particles
        # by a push's length and packs them along the push's end: among 6 x 300 receivers x 10 steps some pair of distances
        # comes within an ulp of a tie, and WHICH implementation's rounding decides that edge is not defined (the reference's
        # own aggregation order is not).  Every row of the fixture is therefore a push sequence whose whole
        # reference trajectory keeps its graph decisions at least MARGIN away from changing (first seeds that do).
        nb, ns, H = 2, 3, 10
        MARGIN = 5e-8 if N >= 300 else 2e-7
        s, dens, attr = syn.make_pile(N, n_batch=nb, seed=170 + N, kind='blob' if N <= 50 else 'uniform')
        acts = np.zeros((ns * nb, H, 4), np.float32)
        tried = 0
        for b in range(nb):
            found, cand = 0, 0
            while found < ns:
                a1 = np.stack([syn.pushes_through(s[b:b + 1], seed=1000 * t + N + 7919 * (cand + 50 * b)) for t in range(H)], 1)
                cand += 1
                tried += 1
                with torch.no_grad():
                    r1 = planner.ptcl_model_rollout(torch.from_numpy(s[b:b + 1]), torch.from_numpy(dens[b:b + 1]),
                                                    torch.from_numpy(attr[b:b + 1]), model, torch.from_numpy(a1))
                if rollout_margin(torch, planner, s[b:b + 1], a1, r1['model_rollout']['state_pred'].numpy()) >= MARGIN:
                    acts[found * nb + b] = a1[0]
                    found += 1
        with torch.no_grad():
            ro = planner.ptcl_model_rollout(torch.from_numpy(s), torch.from_numpy(dens), torch.from_numpy(attr), model,
                                            torch.from_numpy(acts))
            sp = ro['model_rollout']['state_pred']
            goal_coor = syn.goal_coor_strided(obs_goal, 5 * N)
            rs, nr = planner.ptcl_evaluate_traj(sp.reshape(ns * nb, H, 1, N, 3), torch.from_numpy(obs_goal),
                                                torch.from_numpy(goal_coor))
        p = 'rollout/' + name + '/'
        out[p + 's_cur'], out[p + 'dens'], out[p + 'attr'], out[p + 'act_seqs'] = s, dens, attr, acts
        out[p + 'state_pred'] = sp.numpy()
        out[p + 'goal_coor'] = goal_coor
        out[p + 'next_r'] = nr.numpy()
        out[p + 'margin'] = np.array(rollout_margin(torch, planner, s, acts, sp.numpy()))
        print('[cases] %s rollout: %d candidate push sequences for %d rows, decision margin of the batch %.2e' %
              (name, tried, ns * nb, float(out[p + 'margin'])), flush=True)

        # gradients of the GD planner's loss (planners.py:702-743), horizon 1 (the shipped configuration) and 2
        for H in (1, 2):
            nb, traj = 2, 4
            s, dens, attr = syn.make_pile(N, n_batch=nb, seed=270 + N + H, kind='blob' if N <= 50 else 'uniform')
            acts0 = np.stack([syn.pushes_through(np.tile(s[:1], (traj, 1, 1)), seed=2000 * t + N + H) for t in range(H)], 1)
            acts0 = np.repeat(acts0, nb, axis=0).astype(np.float32)                 # row = traj * nb + b
            a_t = torch.tensor(acts0, requires_grad=True)
            ro = planner.ptcl_model_rollout(torch.from_numpy(s), torch.from_numpy(dens), torch.from_numpy(attr), model, a_t)
            sp = ro['model_rollout']['state_pred']
            obs_seqs = sp.reshape(traj * nb, 1, H, N, 3).permute(0, 2, 1, 3, 4)
            rs, _ = planner.ptcl_evaluate_traj(obs_seqs, torch.from_numpy(obs_goal), torch.from_numpy(goal_coor))
            torch.sum(-rs).backward()
            p = 'grad/%s_h%d/' % (name, H)
            out[p + 's_cur'], out[p + 'dens'], out[p + 'attr'] = s, dens, attr
            out[p + 'act_seqs'], out[p + 'goal_coor'] = acts0, goal_coor
            out[p + 'reward'] = rs.detach().numpy()
            out[p + 'grad_act'] = a_t.grad.numpy()

        # the live planner (planners.py:563-871): Adam iterations on traj x batch pushes, the vote, the winner's re-rollout
        if N <= 100:
            nb, traj, n_it = 3, 6, 5
            s, dens, attr = syn.make_pile(N, n_batch=nb, seed=370 + N, kind='blob' if N <= 50 else 'uniform')
            act_seq = syn.pushes_through(np.tile(s[:1], (traj, 1, 1)), seed=3000 + N)[None].astype(np.float64)   # [1,traj,4]
            np.random.seed(0)
            res = planner.trajectory_optimization_ptcl_multi_traj(
                s, dens, attr, obs_goal, model, act_seq, np.zeros(1), n_sample=traj, n_look_ahead=1,
                n_update_iter=n_it, action_lower_lim=lo, action_upper_lim=hi, use_gpu=False, time_lim=1e9)
            p = 'gd/' + name + '/'
            out[p + 's_cur'], out[p + 'dens'], out[p + 'attr'], out[p + 'act_seq'] = s, dens, attr, act_seq
            out[p + 'n_update_iter'] = np.array(n_it)
            for k in ('action_sequence', 'action_full', 'reward_full', 'observation_sequence', 'reward', 'next_r',
                      'rew_mean', 'rew_std'):
                out[p + 'out/' + k] = np.asarray(res[k])
            out[p + 'out/iter_num'] = np.array(res['iter_num'])
        print('[cases] %s done  %.0f s' % (name, time.time() - t0), flush=True)
    cap.close()
    np.savez_compressed(os.path.join(HERE, 'trained.npz'), **out)
    for f in ('weights_trained.npz', 'trained.npz', 'train_curve.npz'):
        print('%-22s %8.1f KB' % (f, os.path.getsize(os.path.join(HERE, f)) / 1024.0))


if __name__ == '__main__':
    main()
