"""Arithmetic accuracy of the engines on the golden one-step case: the final particle effect
(64 features after three propagation steps, before the predictor) against the reference's."""
import sys
sys.path.insert(0, '.')
import numpy as np
from dyn_res_pile_manip_amd import weights, _lib
from dyn_res_pile_manip_amd.engine import Engine
g = np.load('tests/golden/one_step.npz')
w = np.load('tests/golden/weights_seed0.npz')
eng = Engine(0)
eng.load_weights(weights.blob_from_state_dict(w), 0.08)
for case in ('n64', 'n8'):
    a, s, sd, d = [g[case + '/' + k] for k in ('attr', 's_cur', 's_delta', 'dens')]
    ref = g[case + '/particle_effect_2'].reshape(a.shape[0], a.shape[1], 64)
    for name in ('valu', 'mfma', 'split', 'fused'):
        eng.set_engine(_lib.ENGINES[name])
        eng.step(a, s, sd, d)
        eff = eng.debug_fetch('effect', ref.shape)
        print('%-4s %-5s final effect: max |err| / max |ref| = %.2e' % (case, name, np.abs(eff - ref).max() / np.abs(ref).max()))
