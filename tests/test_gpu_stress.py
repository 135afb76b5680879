"""GPU: the split arithmetic beyond the one weight set of the other fixtures (tests/golden/stress.npz,
captured from the reference): a second and third seed, weights scaled so the hidden activations reach
1e2..1e3 or shrink to 1e-2, per-particle attributes, densities at both ends of the training range.
Every engine -- the default `fused` one included -- against the REFERENCE's outputs, not against another
engine."""
import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from oracle import propnet_sparse as osp
from test_oracle_golden import STRESS, stress_weights
from test_gpu_parity import check_rollout, disp_rel

pytestmark = pytest.mark.gpu
ENGINES = ['valu', 'mfma', 'split', 'fused']


@pytest.fixture(scope='module')
def eng():
    from dyn_res_pile_manip_amd.engine import Engine
    e = Engine(0)
    e.M34 = osp.world2cam_affine(syn.demo_cam_extrinsics(), 24)
    e.set_camera(e.M34, 24.0, syn.demo_cam_params())
    yield e
    e.close()


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('case', STRESS)
def test_stress_case_matches_the_reference(eng, golden, case, engine):
    g = golden.stress
    w = stress_weights(g, case)
    eng.load_weights(weights.blob_from_state_dict(w), 0.08)
    eng.set_engine(_lib.ENGINES[engine])
    s, sd, a, d = g[case + '/s_cur'], g[case + '/s_delta'], g[case + '/attr'], g[case + '/dens']
    out = eng.step(a, s, sd, d)
    assert np.isfinite(out).all()
    assert disp_rel(out, g[case + '/s_pred'], s) < 1e-4
    states, _ = eng.rollout(s, a, d, g[case + '/act_seqs'])
    check_rollout(eng, None, s, a, d, g[case + '/act_seqs'], g[case + '/state_pred'], states)


def test_fp16_range_is_reported(eng, golden):
    """Weights far beyond any trained network's (first encoder layers x 1e7): the hidden activations leave
    fp16's range, where the split relation encoder would saturate.  The engine must not return wrong-but-finite
    positions: the call either fails with DRP_ERANGE or gives the fp32 engines' answer."""
    g = golden.stress
    sd_ = {k: np.array(stress_weights(g, 'seed1')[k]) for k in stress_weights(g, 'seed1').files}
    for k in ('w/model.relation_encoder.model.0.weight', 'w/model.relation_encoder.model.0.bias'):
        sd_[k] = sd_[k] * np.float32(1e7)

    class _W(object):
        files = list(sd_.keys())

        def __getitem__(self, k):
            return sd_[k]
    blob = weights.blob_from_state_dict(_W())
    case = 'seed1'
    s, sdl, a, d = g[case + '/s_cur'], g[case + '/s_delta'], g[case + '/attr'], g[case + '/dens']
    eng.load_weights(blob, 0.08)
    eng.set_engine(_lib.ENGINE_MFMA)
    ref = eng.step(a, s, sdl, d)
    eng.set_engine(_lib.ENGINE_FUSED)
    try:
        out = eng.step(a, s, sdl, d)
    except _lib.DrpError as e:
        assert 'range' in str(e).lower()
        return
    assert disp_rel(out, ref, s) < 1e-4
