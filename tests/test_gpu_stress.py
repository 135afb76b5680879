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


def _scaled_weights(g, factor):
    w = stress_weights(g, 'seed1')
    sd_ = {k: np.array(w[k]) for k in w.files}
    for k in ('w/model.relation_encoder.model.0.weight', 'w/model.relation_encoder.model.0.bias'):
        sd_[k] = sd_[k] * np.float32(factor)

    class _W(object):
        files = list(sd_.keys())

        def __getitem__(self, k):
            return sd_[k]
    return weights.blob_from_state_dict(_W()), osp.weights_np(sd_)


@pytest.mark.parametrize('engine', ENGINES)
@pytest.mark.parametrize('factor', [1e7, 1e-6])
def test_weights_far_outside_fp16_still_give_the_fp32_answer(eng, golden, factor, engine):
    """First relation-encoder layer x 1e7 (hidden activations ~1e7: an un-shifted fp16 piece would saturate at
    65 504 and return wrong-but-finite positions) and x 1e-6 (residuals below fp16's subnormals): the range
    shift 2^k chosen from the weights keeps the split engines on the fp32 answer -- the ORACLE's (numpy fp32 on the
    same scaled weights; pinned to the reference by stress.npz at x 300 / x 3 000 / x 0.2), same lists, flat 1e-4."""
    g = golden.stress
    case = 'seed1'
    s, sdl, a, d = g[case + '/s_cur'], g[case + '/s_delta'], g[case + '/attr'], g[case + '/dens']
    blob, W = _scaled_weights(g, factor)
    eng.load_weights(blob, 0.08)
    eng.set_engine(_lib.ENGINES[engine])
    ref = osp.predict_one_step(W, a, s, sdl, d)
    assert np.isfinite(ref).all()
    idx, cnt = eng.build_graph(s, sdl)
    ridx, rcnt = osp.build_neighbours(s, sdl)
    np.testing.assert_array_equal(cnt, rcnt)
    np.testing.assert_array_equal(idx, ridx)
    out = eng.step(a, s, sdl, d)
    assert disp_rel(out, ref, s) < 1e-4


def test_inputs_beyond_the_proven_range_are_refused(eng, golden):
    """DRP_ERANGE instead of wrong-but-finite: attributes a million times the envelope the range shift was proven
    for, and weights with a matrix entry beyond fp16, are refused by the split engines (nothing is computed) and
    served by the fp32 engines."""
    g = golden.stress
    case = 'seed1'
    s, sdl, a, d = g[case + '/s_cur'], g[case + '/s_delta'], g[case + '/attr'], g[case + '/dens']
    eng.load_weights(weights.blob_from_state_dict(stress_weights(g, case)), 0.08)
    eng.set_engine(_lib.ENGINE_FUSED)
    with pytest.raises(_lib.DrpError, match='range'):
        eng.step(a + np.float32(1e6), s, sdl, d)
    with pytest.raises(_lib.DrpError, match='range'):
        eng.rollout(s, a, d * np.float32(1e9), g[case + '/act_seqs'])
    out = eng.step(a, s, sdl, d)                               # the context stays usable
    assert disp_rel(out, g[case + '/s_pred'], s) < 1e-4
    eng.set_engine(_lib.ENGINE_MFMA)
    assert np.isfinite(eng.step(a + np.float32(1e6), s, sdl, d)).all()
    # a hidden-layer weight of 1e5 cannot be written as two fp16 pieces
    w = stress_weights(g, case)
    sd_ = {k: np.array(w[k]) for k in w.files}
    sd_['w/model.relation_encoder.model.2.weight'][3, 5] = 1e5

    class _W(object):
        files = list(sd_.keys())

        def __getitem__(self, k):
            return sd_[k]
    eng.load_weights(weights.blob_from_state_dict(_W()), 0.08)
    ref = eng.step(a, s, sdl, d)                               # fp32 MFMA engine: fine
    assert np.isfinite(ref).all()
    eng.set_engine(_lib.ENGINE_FUSED)
    with pytest.raises(_lib.DrpError, match='range'):
        eng.step(a, s, sdl, d)
