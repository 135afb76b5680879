"""ORACLE (test infrastructure, not product code): the same hot path restated in
the SPARSE, FACTORED form the HIP kernels compute, numpy fp32.

It is the executable specification of the kernels' intermediate buffers
(neighbour lists, per-slot edge constants, per-node constants), so a failing
GPU parity test can be bisected stage by stage.  It is itself checked against
the golden fixtures captured from the reference (`tests/test_oracle_golden.py`).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.

Formulation (reference lines in brackets):
  * receiver-major fixed-K neighbour lists instead of dense one-hot Rr/Rs
    [model/gnn_dyn.py:223-251]: for receiver i the <=10 senders j with
    |p_j-p_i|^2 - thr < 0 among i's 10 nearest, ascending j (the order
    `nonzero()` enumerates them, :247).
  * relation propagator weight [64,193] = [W_e | W_r | W_s | w_d] over
    cat[relation_encode, effect_r, effect_s, dens] [:140-141,:186-187]:
      c_edge = W_e re + w_d d + b           (constant over the 3 steps)
      erel   = relu(c_edge + (W_r eff)[recv] + (W_s eff)[send])
  * particle propagator weight [64,129] = [W_pe | W_agg | w_d] over
    cat[particle_encode, agg, dens] with the residual inside the ReLU
    [:136-137,:191-193,:82-85]:
      c_node = W_pe pe + w_d d + b          (constant over the 3 steps)
      eff    = relu(c_node + W_agg agg + eff)
"""
import numpy as np

K = 10
F = 64
PSTEP = 3
DENS_SCALE = np.float32(5000.0)


def f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def weights_np(npz_or_dict):
    out = {}
    keys = npz_or_dict.files if hasattr(npz_or_dict, 'files') else npz_or_dict.keys()
    for k in keys:
        kk = k[2:] if k.startswith('w/') else k
        if kk.startswith('model.'):
            out[kk] = f32(np.asarray(npz_or_dict[k]))
    return out


def build_neighbours(s_cur, s_delta, adj_thresh=0.08):
    """[model/gnn_dyn.py:223-237] -> nbr_idx [B,N,K] int32 (-1 padded), nbr_cnt [B,N].
    Distances exactly as torch evaluates them: ((dx*dx + dy*dy) + dz*dz) in fp32 on
    p = s_cur + s_delta, threshold compared as (dis - fp32(thr)) < 0."""
    s_cur, s_delta = f32(s_cur), f32(s_delta)
    B, N, _ = s_cur.shape
    p = s_cur + s_delta
    thr = np.float32(adj_thresh * adj_thresh)
    nbr_idx = -np.ones((B, N, K), dtype=np.int32)
    nbr_cnt = np.zeros((B, N), dtype=np.int32)
    k = min(K, N)
    col = np.arange(N, dtype=np.uint64)[None, :]
    for b in range(B):
        d = p[b][None, :, :] - p[b][:, None, :]          # [recv i, send j, 3] = p_j - p_i
        sq = d * d
        dis = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
        # topk(k, smallest) with ties to the lower sender index (what a stable sort by distance gives): the k smallest
        # of the keys (distance bits, sender) -- a non-negative fp32's bit pattern orders like its value
        key = (dis.view(np.uint32).astype(np.uint64) << np.uint64(32)) | col
        js = (np.partition(key, k - 1, axis=1)[:, :k] & np.uint64(0xffffffff)).astype(np.int64)
        keep = (np.take_along_axis(dis, js, 1) - thr) < 0
        js = np.sort(np.where(keep, js, N + 1), axis=1)   # ascending sender, the dropped ones last
        nbr_idx[b, :, :k] = np.where(js < N, js, -1)
        nbr_cnt[b] = keep.sum(1)
    return nbr_idx, nbr_cnt


def _dense(x, Wt, b):
    return x @ Wt.T + b


def forward_sparse(W, a_cur, s_cur, s_delta, dens, nbr_idx, nbr_cnt, taps=None):
    """[model/gnn_dyn.py:147-198] on neighbour lists.  Returns s_pred [B,N,3]."""
    a_cur, s_cur, s_delta, dens = f32(a_cur), f32(s_cur), f32(s_delta), f32(dens)
    B, N = a_cur.shape
    d = dens / DENS_SCALE                                   # [:158]
    relu = lambda x: np.maximum(x, np.float32(0))
    Wrp = W['model.relation_propagator.linear.weight']      # [64,193]
    brp = W['model.relation_propagator.linear.bias']
    W_e, W_r, W_s, w_d = Wrp[:, :F], Wrp[:, F:2 * F], Wrp[:, 2 * F:3 * F], Wrp[:, 3 * F]
    Wpp = W['model.particle_propagator.linear.weight']      # [64,129]
    bpp = W['model.particle_propagator.linear.bias']
    W_pe, W_agg, w_d2 = Wpp[:, :F], Wpp[:, F:2 * F], Wpp[:, 2 * F]

    valid = np.arange(K)[None, None, :] < nbr_cnt[:, :, None]            # [B,N,K]
    send = np.where(valid, nbr_idx, 0)
    bidx = np.arange(B)[:, None, None]

    # particle encoder [:174-175]
    x_n = np.concatenate([s_delta, a_cur[..., None], np.broadcast_to(d[:, None, None], (B, N, 1))], 2)
    h = relu(_dense(x_n, W['model.particle_encoder.model.0.weight'], W['model.particle_encoder.model.0.bias']))
    pe = relu(_dense(h, W['model.particle_encoder.model.2.weight'], W['model.particle_encoder.model.2.bias']))
    c_node = _dense(pe, W_pe, bpp) + d[:, None, None] * w_d2[None, None, :]

    # relation encoder on slots [:166-171,:179-180]: inputs [a_r, a_s, s_r - s_s, d]
    a_r = np.broadcast_to(a_cur[:, :, None], (B, N, K))
    a_s = a_cur[bidx, send]
    ds = s_cur[:, :, None, :] - s_cur[bidx, send]                        # receiver - sender
    x_e = np.concatenate([a_r[..., None], a_s[..., None], ds,
                          np.broadcast_to(d[:, None, None, None], (B, N, K, 1))], 3).astype(np.float32)
    h = relu(_dense(x_e, W['model.relation_encoder.model.0.weight'], W['model.relation_encoder.model.0.bias']))
    h = relu(_dense(h, W['model.relation_encoder.model.2.weight'], W['model.relation_encoder.model.2.bias']))
    re = relu(_dense(h, W['model.relation_encoder.model.4.weight'], W['model.relation_encoder.model.4.bias']))
    c_edge = _dense(re, W_e, brp) + d[:, None, None, None] * w_d[None, None, None, :]

    eff = pe
    if taps is not None:
        taps.update(particle_encode=pe, c_node=c_node, relation_encode=re, c_edge=c_edge)
    for p in range(PSTEP):
        P_r = eff @ W_r.T                                                # [B,N,64]
        P_s = eff @ W_s.T
        erel = relu(c_edge + P_r[:, :, None, :] + P_s[bidx, send])       # [B,N,K,64]
        erel = np.where(valid[..., None], erel, np.float32(0))
        agg = erel.sum(2)                                                # [:189]
        eff = relu(c_node + agg @ W_agg.T + eff)                         # [:191-193]
        if taps is not None:
            taps['effect_rel_%d' % p] = erel
            taps['particle_effect_%d' % p] = eff
    h = relu(_dense(eff, W['model.particle_predictor.linear_0.weight'], W['model.particle_predictor.linear_0.bias']))
    pred = _dense(h, W['model.particle_predictor.linear_1.weight'], W['model.particle_predictor.linear_1.bias'])
    if taps is not None:
        taps['particle_pred'] = pred
    return (pred + s_cur).astype(np.float32)                             # [:198]


def predict_one_step(W, a_cur, s_cur, s_delta, dens, adj_thresh=0.08, taps=None):
    nbr_idx, nbr_cnt = build_neighbours(s_cur, s_delta, adj_thresh)
    if taps is not None:
        taps.update(nbr_idx=nbr_idx, nbr_cnt=nbr_cnt)
    return forward_sparse(W, a_cur, s_cur, s_delta, dens, nbr_idx, nbr_cnt, taps)


def world2cam_affine(cam_extrinsic, global_scale):
    """[planners.py:192-209] -> the 12 fp32 constants (3x4, row-major) of
    cam = (M [p;1])[:3], NOT yet divided by global_scale (the reference divides
    after the fp32 matmul, so the kernels do the same)."""
    gl = np.diag([1.0, -1.0, -1.0, 1.0])
    m = np.linalg.inv(np.matmul(np.linalg.inv(np.asarray(cam_extrinsic, dtype=np.float64)), gl))
    return f32(m[:3, :4])


def gen_s_delta(s_cur, action, M34, global_scale):
    """[planners.py:211-257] elementwise, in the operation order the kernels use."""
    s_cur, action = f32(s_cur), f32(action)
    gs = np.float32(global_scale)
    B, N, _ = s_cur.shape

    def to_cam(x, y, z):
        p = np.stack([x, y, z, np.ones_like(x)], 1)
        return (p @ M34.T) / gs

    zero = np.zeros((B,), dtype=np.float32)
    sc = to_cam(action[:, 0], zero, -action[:, 1])
    ec = to_cam(action[:, 2], zero, -action[:, 3])
    dv = ec - sc
    length = np.sqrt((dv * dv).sum(1, dtype=np.float32)).astype(np.float32)
    dirn = dv / length[:, None]
    ortho = np.stack([-dirn[:, 1], dirn[:, 0], np.zeros_like(length)], 1)
    rel = s_cur - sc[:, None, :]
    v = (rel * ortho[:, None, :]).sum(-1, dtype=np.float32)
    u = (rel * dirn[:, None, :]).sum(-1, dtype=np.float32)
    hard = ((u < length[:, None]) & (u > 0)).astype(np.float32)
    w = np.float32(0.8 / 24.0)
    soft = np.maximum(np.maximum(-w - v, 0), np.maximum(v - w, 0)).astype(np.float32)
    soft = np.exp(-soft / np.float32(0.01)).astype(np.float32)
    to_end = ((ec[:, None, :] - s_cur) * dirn[:, None, :]).sum(-1, dtype=np.float32)
    return (to_end[..., None] * dirn[:, None, :] * hard[..., None] * soft[..., None]).astype(np.float32)


def rollout(W, s0, dens, attr, act_seqs, M34, global_scale, adj_thresh=0.08, taps=None):
    """[planners.py:302-370]; row = sample * n_batch + batch."""
    s0, dens, attr, act_seqs = f32(s0), f32(dens), f32(attr), f32(act_seqs)
    B, H, _ = act_seqs.shape
    nb, N, _ = s0.shape
    ns = B // nb
    s = np.tile(s0, (ns, 1, 1))
    d = np.tile(dens, ns)
    a = np.tile(attr, (ns, 1))
    out = np.zeros((B, H, N, 3), dtype=np.float32)
    for t in range(H):
        sd = gen_s_delta(s, act_seqs[:, t], M34, global_scale)
        step_taps = {} if taps is not None else None
        s = predict_one_step(W, a, s, sd, d, adj_thresh, step_taps)
        if taps is not None:
            taps.setdefault('nbr_idx', []).append(step_taps['nbr_idx'])
            taps.setdefault('nbr_cnt', []).append(step_taps['nbr_cnt'])
        out[:, t] = s
    return out


def reward(state, G, cam_params, goal_coor, normalize=True):
    """[env/flex_rewards.py:189-214] with the bilinear sample written out
    (grid_sample, padding 'border', align_corners False)."""
    state, G, goal_coor = f32(state), f32(G), f32(goal_coor)
    Bn, N, _ = state.shape
    Hh, Ww = G.shape
    fx, fy, cx, cy = [np.float32(v) for v in cam_params]
    px = state[..., 0] * fx / state[..., 2] + cx
    py = state[..., 1] * fy / state[..., 2] + cy
    nx = px / np.float32(Hh) * np.float32(2) - np.float32(1)
    ny = py / np.float32(Hh) * np.float32(2) - np.float32(1)
    ix = np.clip(((nx + 1) * np.float32(Ww) - 1) / 2, 0, Ww - 1).astype(np.float32)
    iy = np.clip(((ny + 1) * np.float32(Hh) - 1) / 2, 0, Hh - 1).astype(np.float32)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    tx = ix - x0
    ty = iy - y0
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    x1 = np.minimum(x0 + 1, Ww - 1)
    y1 = np.minimum(y0 + 1, Hh - 1)
    val = (G[y0, x0] * (1 - tx) * (1 - ty) + G[y0, x1] * tx * (1 - ty)
           + G[y1, x0] * (1 - tx) * ty + G[y1, x1] * tx * ty)
    r1 = val.sum(1, dtype=np.float32)
    r2 = np.zeros((Bn,), dtype=np.float32)
    for b in range(Bn):
        dx = goal_coor[:, 0][:, None] - px[b][None, :]
        dy = goal_coor[:, 1][:, None] - py[b][None, :]
        r2[b] = np.sqrt(dx * dx + dy * dy).min(1).sum(dtype=np.float32)
    r = r1 + r2
    if normalize:
        r = r / np.float32(N)
    return (-r).astype(np.float32)


def mppi_partials(reward_weight, rewards, act_seqs):
    """The per-shard pieces of [planners.py:549-561] that a rank contributes:
    m = max(lambda r), Z = sum exp(lambda r - m), A = sum exp(lambda r - m) act.
    Combining shards: rescale by exp(m_g - m_all) and add."""
    z = np.float64(reward_weight) * np.asarray(rewards, dtype=np.float64)
    m = z.max()
    w = np.exp(z - m)
    return m, w.sum(), (w[:, None, None] * np.asarray(act_seqs, dtype=np.float64)).sum(0)
