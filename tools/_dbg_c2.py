import sys
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine
from oracle import propnet_sparse as osp
eng = Engine(0)
sd = weights.random_state_dict(seed=0)
eng.load_weights(weights.blob_from_state_dict(sd), 0.08)
M34 = world2cam_affine(syn.demo_cam_extrinsics())
eng.set_camera(M34, 24.0, syn.demo_cam_params())
W = osp.weights_np(sd)
N, ns, H = 300, 1024, 10
s0, dens, attr = syn.make_pile(N, 1, seed=0)
acts = syn.sample_pushes(ns, H, seed=0)
out, _ = eng.rollout(s0, attr, dens, acts)
rows = np.unique(np.linspace(0, ns - 1, 32).astype(int))
prev = np.repeat(s0[:1], len(rows), 0)
at, de = np.repeat(attr[:1], len(rows), 0), np.repeat(dens[:1], len(rows))
for t in range(H):
    sdl = osp.gen_s_delta(prev, acts[rows, t], M34, 24.0)
    ref = osp.predict_one_step(W, at, prev, sdl, de)
    o = out[rows, t]
    err = np.abs(o - ref).reshape(len(rows), -1).max(1) / np.maximum(np.abs(ref - prev).reshape(len(rows), -1).max(1), 1e-12)
    bad = np.nonzero(err > 1e-4)[0]
    for b in bad:
        r = rows[b]
        sd_dev = eng.gen_s_delta(prev[b:b + 1], acts[r:r + 1, t])
        dsd = np.abs(sd_dev[0] - sdl[b])
        i = int(np.abs(o[b] - ref[b]).max(1).argmax())
        print('step', t, 'row', r, 'err %.2e' % err[b], 'particle', i, 'abs err', np.abs(o[b] - ref[b]).max(), 'disp', np.abs(ref[b] - prev[b]).max(),
              '| s_delta device vs oracle max %.2e at particle %d (support equal: %s)' % (dsd.max(), int(dsd.max(1).argmax()), np.array_equal(sd_dev[0] != 0, sdl[b] != 0)))
        idx_d, cnt_d = eng.build_graph(prev[b:b + 1], sd_dev)
        idx_o, cnt_o = osp.build_neighbours(prev[b:b + 1], sdl[b:b + 1])
        print('   lists equal:', np.array_equal(idx_d, idx_o), np.array_equal(cnt_d, cnt_o))
        one = eng.step(at[b:b + 1], prev[b:b + 1], sd_dev, de[b:b + 1])
        print('   one-step API vs oracle: %.2e ; vs rollout %.2e' % (np.abs(one[0] - ref[b]).max(), np.abs(one[0] - o[b]).max()))
    prev = o
print('done')
