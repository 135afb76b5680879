mkdir -p gpurun_out
python tools/_dbg_c2.py > gpurun_out/dbg_c2.txt 2>&1
python - > gpurun_out/dbg_bwd2.txt 2>&1 <<'PY'
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
for rows in (True, False):
    if rows: os.environ.pop('DRP_NO_BWD_ROWS', None)
    else: os.environ['DRP_NO_BWD_ROWS'] = '1'
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(sd), 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    eng.set_goal(G, gc)
    eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)
    res[rows] = eng.gd_grad(want_state_grad=True)
    eng.close()
W = od.load_weights({'w/' + k: np.asarray(v) for k, v in sd.items()})
r, g, gs = od.gd_loss_and_grads(W, s0, dens, attr, acts, G, syn.demo_cam_params(), gc, syn.demo_cam_extrinsics(), 24)
scale = np.abs(g).max()
for k, v in res.items():
    e = np.abs(v[1] - g).reshape(B, -1).max(1) / scale
    print('rows' if k else 'step', 'grad_act vs oracle per row:', np.round(e * 1e6).astype(int), '(1e-6 of scale)')
PY
python -m pytest tests/test_gpu_planner.py -q -s -p no:cacheprovider -k "fps" 2>&1 | grep "fps\]\|passed\|failed" > gpurun_out/fps3.txt
