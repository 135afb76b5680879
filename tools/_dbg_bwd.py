import os, sys
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine
from oracle import propnet_dense as od
N, B, H = 256, 40, 1
s0, dens, attr = syn.make_pile(N, 1, seed=N)
acts = syn.sample_pushes(B, H, seed=B)
obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
lo, hi = syn.action_limits()
sd = weights.random_state_dict(seed=0)
G = syn.goal_field(obs_goal); gc = syn.goal_coor_strided(obs_goal, 5 * N)
res = {}
for tapeN in ('40', '256'):
    for rows in (True, False):
        os.environ['DRP_ECACHE_TAPE_MAX_N'] = tapeN
        if rows: os.environ.pop('DRP_NO_BWD_ROWS', None)
        else: os.environ['DRP_NO_BWD_ROWS'] = '1'
        eng = Engine(0)
        eng.load_weights(weights.blob_from_state_dict(sd), 0.08)
        eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
        eng.set_goal(G, gc)
        eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
        eng.dispatch_reset()
        res[tapeN, rows] = eng.gd_grad(want_state_grad=True)
        print(tapeN, rows, [v for v in eng.last_dispatch() if 'prop' in v or 'bwd' in v])
        eng.close()
W = od.load_weights({'w/' + k: np.asarray(v) for k, v in sd.items()})
r, g, gs = od.gd_loss_and_grads(W, s0, dens, attr, acts[:8], G, syn.demo_cam_params(), gc, syn.demo_cam_extrinsics(), 24)
scale = np.abs(g).max()
for k, v in res.items():
    print(k, 'grad_act vs oracle (8 rows): %.2e of scale %.3g; state grad %.2e' % (np.abs(v[1][:8] - g).max() / scale, scale, np.abs(v[2][:8] - gs).max() / np.abs(gs).max()))
for a in ('40', '256'):
    print('rows vs step, tape ec', a, ': %.2e' % (np.abs(res[a, True][1] - res[a, False][1]).max() / np.abs(res[a, False][1]).max()))
print('ec on vs off (rows): %.2e' % (np.abs(res['40', True][1] - res['256', True][1]).max() / scale))
