"""ORACLE (test infrastructure, not product code): CPU restatement of the
reference's hot path in the reference's own DENSE formulation, PyTorch fp32.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.  The product path (`dyn_res_pile_manip_amd`) never does.

Parity pin: the reference has no tests or golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against outputs of the
reference itself, captured in this container by `tests/golden/make_golden.py`
and committed as `tests/golden/*.npz` (`tests/test_oracle_golden.py`).  The one
third-party call that cannot be pinned is `cv2.distanceTransform`
(env/flex_rewards.py:174): the reward takes the already-built field `G` as an
input, so nothing here depends on it.

Each function cites the reference lines it restates (paths relative to the
reference repo).  Weights come as a dict of numpy arrays keyed like the
reference's `state_dict`.
"""
import numpy as np
import torch
import torch.nn.functional as F

PSTEP = 3            # model/gnn_dyn.py:160
MAX_REL = 10         # model/gnn_dyn.py:231
DENS_SCALE = 5000.0  # model/gnn_dyn.py:158
PUSHER_W = 0.8 / 24.0  # planners.py:228
SOFT_MASK_SCALE = 0.01  # planners.py:251


def _t(x):
    if isinstance(x, torch.Tensor):
        return x.float()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))


def load_weights(npz_or_dict):
    """`w/model.<...>` arrays -> dict of torch tensors keyed without the prefix."""
    out = {}
    for k in (npz_or_dict.files if hasattr(npz_or_dict, 'files') else npz_or_dict.keys()):
        if k.startswith('w/'):
            out[k[2:]] = _t(npz_or_dict[k])
        elif k.startswith('model.'):
            out[k] = _t(npz_or_dict[k])
    return out


def _lin(W, name, x):
    return torch.addmm(W[name + '.bias'], x, W[name + '.weight'].t())


# --- model/gnn_dyn.py:35-59 (ParticleEncoder), dims :127-128 -----------------
def particle_encoder(W, x):
    B, N, D = x.shape
    h = torch.relu(_lin(W, 'model.particle_encoder.model.0', x.reshape(B * N, D)))
    h = torch.relu(_lin(W, 'model.particle_encoder.model.2', h))
    return h.reshape(B, N, -1)


# --- model/gnn_dyn.py:6-32 (RelationEncoder), dims :132-133 ------------------
def relation_encoder(W, x):
    B, E, D = x.shape
    h = torch.relu(_lin(W, 'model.relation_encoder.model.0', x.reshape(B * E, D)))
    h = torch.relu(_lin(W, 'model.relation_encoder.model.2', h))
    h = torch.relu(_lin(W, 'model.relation_encoder.model.4', h))
    return h.reshape(B, E, -1)


# --- model/gnn_dyn.py:62-87 (Propagator) -------------------------------------
def propagator(W, name, x, residual=None):
    B, R, D = x.shape
    h = _lin(W, name + '.linear', x.reshape(B * R, D))
    if residual is not None:
        h = h + residual.reshape(B * R, -1)
    return torch.relu(h).reshape(B, R, -1)


# --- model/gnn_dyn.py:89-111 (ParticlePredictor): no final activation ---------
def particle_predictor(W, x):
    B, N, D = x.shape
    h = torch.relu(_lin(W, 'model.particle_predictor.linear_0', x.reshape(B * N, D)))
    return _lin(W, 'model.particle_predictor.linear_1', h).reshape(B, N, 3)


def adjacency(s_cur, s_delta, adj_thresh):
    """model/gnn_dyn.py:223-237: radius AND top-10 mask on p = s_cur + s_delta.
    adj[b,i,j] = 1 iff sender j is among receiver i's 10 nearest and
    |p_j - p_i|^2 - thr < 0 (self-loops included)."""
    B, N, _ = s_cur.shape
    p = s_cur + s_delta
    recv = p[:, :, None, :].repeat(1, 1, N, 1)
    send = p[:, None, :, :].repeat(1, N, 1, 1)
    thr = adj_thresh * adj_thresh
    dis = torch.sum((send - recv) ** 2, -1)
    k = min(MAX_REL, N)
    idx = torch.topk(dis, k=k, dim=2, largest=False).indices
    topk_mask = torch.zeros_like(dis)
    topk_mask.scatter_(2, idx, 1)
    adj = ((dis - thr) < 0).float() * topk_mask
    return adj, dis


def onehot_relations(adj):
    """model/gnn_dyn.py:242-251: edges in (b, receiver, sender) lexicographic
    order, numbered per sample, as dense one-hot Rr/Rs [B, max_E, N]."""
    B, N, _ = adj.shape
    n_rels = adj.sum(dim=(1, 2)).long()
    n_rel = int(n_rels.max().item())
    rels = adj.nonzero()
    within = torch.cat([torch.arange(int(n)) for n in n_rels]) if rels.shape[0] else rels[:, 0]
    Rr = torch.zeros((B, n_rel, N))
    Rs = torch.zeros((B, n_rel, N))
    Rr[rels[:, 0], within, rels[:, 1]] = 1
    Rs[rels[:, 0], within, rels[:, 2]] = 1
    return Rr, Rs


def forward_dense(W, a_cur, s_cur, s_delta, Rr, Rs, dens, taps=None):
    """model/gnn_dyn.py:147-198 (PropModuleDiffDen.forward)."""
    B, N = a_cur.shape
    E = Rr.shape[1]
    d = dens / DENS_SCALE
    Rr_t = Rr.transpose(1, 2)
    a_r = Rr.bmm(a_cur[..., None])
    a_s = Rs.bmm(a_cur[..., None])
    s_r = Rr.bmm(s_cur)                         # s_cur, NOT s_cur + s_delta (:170-171)
    s_s = Rs.bmm(s_cur)
    dn = d[:, None, None].repeat(1, N, 1)
    de = d[:, None, None].repeat(1, E, 1)
    pe = particle_encoder(W, torch.cat([s_delta, a_cur[:, :, None], dn], 2))
    re = relation_encoder(W, torch.cat([a_r, a_s, s_r - s_s, de], 2))
    effect = pe
    if taps is not None:
        taps['particle_encode'] = pe
        taps['relation_encode'] = re
    for p in range(PSTEP):
        e_r = Rr.bmm(effect)
        e_s = Rs.bmm(effect)
        e_rel = propagator(W, 'model.relation_propagator', torch.cat([re, e_r, e_s, de], 2))
        agg = Rr_t.bmm(e_rel)
        effect = propagator(W, 'model.particle_propagator', torch.cat([pe, agg, dn], 2),
                            residual=effect)
        if taps is not None:
            taps['effect_rel_%d' % p] = e_rel
            taps['particle_effect_%d' % p] = effect
    pred = particle_predictor(W, effect)
    if taps is not None:
        taps['particle_pred'] = pred
    return pred + s_cur                         # s_delta is NOT added (:198)


def predict_one_step(W, a_cur, s_cur, s_delta, dens, adj_thresh=0.08, taps=None):
    """model/gnn_dyn.py:209-254."""
    a_cur, s_cur, s_delta, dens = _t(a_cur), _t(s_cur), _t(s_delta), _t(dens)
    adj, _ = adjacency(s_cur, s_delta, adj_thresh)
    Rr, Rs = onehot_relations(adj)
    if taps is not None:
        taps['adj'] = adj
        taps['Rr'] = Rr
        taps['Rs'] = Rs
    return forward_dense(W, a_cur, s_cur, s_delta, Rr, Rs, dens, taps)


def world2cam_matrix(cam_extrinsic, global_scale):
    """planners.py:192-209: the constant 3x4 map cam = (M [p;1])[:3] / gs,
    M = inv(inv(cam_ext) diag(1,-1,-1,1)); float64 on the host, cast to fp32."""
    gl = np.diag([1.0, -1.0, -1.0, 1.0])
    m = np.linalg.inv(np.matmul(np.linalg.inv(cam_extrinsic), gl))
    return torch.tensor(m).float(), float(global_scale)


def world2cam(pts, cam_extrinsic, global_scale):
    m, gs = world2cam_matrix(cam_extrinsic, global_scale)
    pts = _t(pts)
    ones = torch.ones((pts.shape[0], 1))
    return torch.matmul(m, torch.cat([pts, ones], 1).T).T[:, :3] / gs


def gen_s_delta(s_cur, action, cam_extrinsic, global_scale):
    """planners.py:211-257: push (sx,sy,ex,ey) -> per-particle impulse."""
    s_cur, action = _t(s_cur), _t(action)
    B, N, _ = s_cur.shape
    zero = torch.zeros((B, 1))
    s3 = torch.cat([action[:, 0:1], zero, -action[:, 1:2]], 1)
    e3 = torch.cat([action[:, 2:3], zero, -action[:, 3:4]], 1)
    sc = world2cam(s3, cam_extrinsic, global_scale)
    ec = world2cam(e3, cam_extrinsic, global_scale)
    dvec = ec - sc
    length = torch.linalg.norm(dvec, dim=1)
    dirn = dvec / torch.linalg.norm(dvec, dim=1, keepdim=True)
    ortho = torch.cat([-dirn[:, 1:2], dirn[:, 0:1], torch.zeros((B, 1))], 1)
    rel = s_cur - sc[:, None, :]
    v = (rel * ortho[:, None, :].repeat(1, N, 1)).sum(-1)
    u = (rel * dirn[:, None, :].repeat(1, N, 1)).sum(-1)
    hard = ((u < length[:, None]) & (u > 0.0)).float()
    soft = torch.maximum(torch.clamp(-PUSHER_W - v, min=0.0), torch.clamp(v - PUSHER_W, min=0.0))
    soft = torch.exp(-soft / SOFT_MASK_SCALE)
    to_end = ((ec[:, None, :] - s_cur) * dirn[:, None, :].repeat(1, N, 1)).sum(-1)
    return to_end[..., None] * dirn[:, None, :] * hard[..., None] * soft[..., None]


def rollout(W, s0, dens, attr, act_seqs, cam_extrinsic, global_scale, adj_thresh=0.08):
    """planners.py:302-370: row = sample * n_batch + batch; returns [B,H,N,3]."""
    s0, dens, attr, act_seqs = _t(s0), _t(dens), _t(attr), _t(act_seqs)
    B, H, _ = act_seqs.shape
    nb, N, _ = s0.shape
    ns = B // nb
    s = s0.repeat(ns, 1, 1)
    d = dens.repeat(ns)
    a = attr.repeat(ns, 1)
    out = torch.zeros((B, H, N, 3))
    for t in range(H):
        sd = gen_s_delta(s, act_seqs[:, t], cam_extrinsic, global_scale)
        s = predict_one_step(W, a, s, sd, d, adj_thresh)
        out[:, t] = s
    return out


def reward_field_bilinear(G, px, py):
    """F.grid_sample(padding_mode='border', align_corners=False) on one channel,
    written out: env/flex_rewards.py:197-199 with norm = 2 p / H - 1."""
    Hh, Ww = G.shape
    ix = torch.clamp(((2 * px / Hh - 1) + 1) * Ww / 2 - 0.5, 0, Ww - 1)
    iy = torch.clamp(((2 * py / Hh - 1) + 1) * Hh / 2 - 0.5, 0, Hh - 1)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx = ix - x0
    fy = iy - y0
    x0 = x0.long()
    y0 = y0.long()
    x1 = torch.clamp(x0 + 1, max=Ww - 1)
    y1 = torch.clamp(y0 + 1, max=Hh - 1)
    return (G[y0, x0] * (1 - fx) * (1 - fy) + G[y0, x1] * fx * (1 - fy)
            + G[y1, x0] * (1 - fx) * fy + G[y1, x1] * fx * fy)


def config_reward_ptcl(state, G, cam_params, goal_coor, normalize=True, offset=(0.0, 0.0)):
    """env/flex_rewards.py:156-214 downstream of the distance transform: `G` is
    the shifted field of :172-177 (an input here)."""
    state, G, goal_coor = _t(state), _t(G), _t(goal_coor)
    B, N, _ = state.shape
    Hh, Ww = G.shape
    fx, fy, cx, cy = cam_params
    pix = torch.zeros((B, N, 2))
    pix[:, :, 0] = state[:, :, 0] * fx / state[:, :, 2] + cx
    pix[:, :, 1] = state[:, :, 1] * fy / state[:, :, 2] + cy
    pix[:, :, 0] += offset[0]
    pix[:, :, 1] += offset[1]
    grid = (pix / Hh * 2 - 1).unsqueeze(1)
    r1 = F.grid_sample(G[None, None].expand(B, 1, Hh, Ww), grid, padding_mode='border',
                       align_corners=False).squeeze(1).squeeze(1).sum(1)
    dist = torch.norm(goal_coor[None, :, None, :] - pix[:, None, :, :], dim=3)
    r2 = dist.min(dim=2).values.sum(1)
    r = r1 + r2
    if normalize:
        r = r / N
    return -r


def evaluate_traj(obs_seqs, G, cam_params, goal_coor):
    """planners.py:372-452 with distractor_df_fn=None: reward of every step,
    `reward_seqs` = the last step's."""
    obs_seqs = _t(obs_seqs)
    ns, H, cvx, N, _ = obs_seqs.shape
    r = config_reward_ptcl(obs_seqs.reshape(ns * H * cvx, N, 3), G, cam_params, goal_coor)
    next_r = r.reshape(ns, H, cvx)
    return next_r[:, -1], next_r


def sample_action_sequences(init_act_seq, n_sample, sigma, beta, lo, hi, rng, n_his=1):
    """planners.py:69-190, noise_type='normal': filtered Gaussian perturbation of
    the nominal sequence, clipped.  init_act_seq [H,4] float64; returns
    [n_sample,H,4].  `rng` is a numpy Generator (the reference uses the global
    np.random state, so parity is distributional)."""
    H = init_act_seq.shape[0]
    acts = np.stack([init_act_seq] * n_sample).astype(np.float64)
    resid = np.zeros((n_sample, 4))
    for t in range(n_his - 1, H):
        noise = rng.normal(0, sigma, (n_sample, 4))
        resid = beta * noise + resid * (1.0 - beta)
        acts[:, t] += resid
        acts[:, t] = np.clip(acts[:, t], lo, hi)
    return acts


def optimize_action(act_seqs, reward_seqs, reward_weight):
    """planners.py:549-561: softmax(reward_weight * reward) weighted mean.
    act_seqs [ns,H,4], reward_seqs [ns] -> [H,4] (float64)."""
    z = reward_weight * np.asarray(reward_seqs, dtype=np.float64)
    w = np.exp(z - z.max())
    w = w / w.sum()
    return (w[:, None, None] * np.asarray(act_seqs, dtype=np.float64)).sum(0)


def gd_loss_and_grads(W, s0, dens, attr, act_seqs, G, cam_params, goal_coor, cam_extrinsic, global_scale):
    """planners.py:685-745 for one iteration: rollout, final-step reward, loss = -sum(reward),
    autograd.  Returns (reward [B], d loss / d act_seqs [B,H,4], d loss / d state_pred [B,H,N,3])."""
    acts = torch.tensor(np.asarray(act_seqs, dtype=np.float32), requires_grad=True)
    st = rollout(W, s0, dens, attr, acts, cam_extrinsic, global_scale)
    st.retain_grad()
    r = config_reward_ptcl(st[:, -1], G, cam_params, goal_coor)
    loss = torch.sum(-r)
    loss.backward()
    return r.detach().numpy(), acts.grad.numpy(), st.grad.numpy()


def train_loss_and_grads(W, states, states_delta, attrs, particle_nums, particle_dens, adj_thresh=0.08):
    """train/train_gnn_dyn.py:159-210 for one collated batch: n_rollout autoregressive steps from
    states[:, 0] with the given impulses, per-sample MSE over the real (unpadded) particles,
    loss / (n_rollout * B), autograd.  W: dict of weight arrays (state_dict keys).
    Returns (loss float, {key: gradient array})."""
    Wt = {k: torch.tensor(np.asarray(v, dtype=np.float32), requires_grad=True) for k, v in W.items()}
    states = _t(states)
    states_delta = _t(states_delta)
    attrs = _t(attrs)
    dens = _t(particle_dens)
    B, T1, N, _ = states.shape
    n_rollout = T1 - 1
    loss = 0.0
    s_cur = states[:, 0]
    a_cur = attrs[:, 0]
    for t in range(n_rollout):
        s_pred = predict_one_step(Wt, a_cur, s_cur, states_delta[:, t], dens, adj_thresh)
        for j in range(B):
            n = int(particle_nums[j])
            loss = loss + torch.nn.functional.mse_loss(s_pred[j, :n], states[j, t + 1, :n])
        s_cur = s_pred
    loss = loss / (n_rollout * B)
    loss.backward()
    return float(loss.item()), {k: v.grad.numpy() for k, v in Wt.items()}


def adam_steps(W, grads_fn, n_steps, lr=1e-3, beta1=0.9):
    """torch.optim.Adam(params, lr, betas=(beta1, 0.999)) for n_steps on a fixed batch
    (train/train_gnn_dyn.py:128-131, :206-209).  grads_fn(W) -> (loss, grads).  Returns
    (losses, final weights)."""
    params = {k: torch.tensor(np.asarray(v, dtype=np.float32), requires_grad=True) for k, v in W.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr, betas=(beta1, 0.999))
    losses = []
    for _ in range(n_steps):
        loss, grads = grads_fn({k: v.detach().numpy() for k, v in params.items()})
        losses.append(loss)
        opt.zero_grad()
        for k, v in params.items():
            v.grad = torch.from_numpy(np.ascontiguousarray(grads[k]))
        opt.step()
    return losses, {k: v.detach().numpy().copy() for k, v in params.items()}
