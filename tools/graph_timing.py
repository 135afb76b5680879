import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine
eng=Engine(0); eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(0)),0.08)
eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()),24.0,syn.demo_cam_params())
B,N=1024,300
s0,dens,attr=syn.make_pile(N,1,seed=0)
s=np.tile(s0,(B,1,1))
for label,sd in [('zero delta',np.zeros((B,N,3),np.float32)),('small noise',0.002*np.random.default_rng(0).standard_normal((B,N,3)).astype(np.float32))]:
    ts=[]
    for r in range(6):
        eng.probe_begin('graph'); eng.build_graph(s,sd); ts.append(eng.probe_read()[0])
    print(label, ['%.3f'%t for t in ts])
for seed in range(5):
    acts=syn.sample_pushes(B,1,seed=seed)[:,0]
    sd=eng.gen_s_delta(s,acts)
    eng.probe_begin('graph'); idx,cnt=eng.build_graph(s,sd); t=eng.probe_read()[0]
    print('push seed',seed,'%.3f ms'%t,'moved frac %.3f'%(np.abs(sd).sum(2)>0).mean(), 'max|sd| %.3f'%np.abs(sd).max())
