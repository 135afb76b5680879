"""GPU: error behaviour of the C ABI (the reference asserts on shapes and swallows
everything else as "OOM"; here every misuse is a negative return code with a message, surfaced as
DrpError by the host layer, and the context stays usable)."""
import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
from dyn_res_pile_manip_amd._lib import DrpError
from dyn_res_pile_manip_amd.engine import Engine
from dyn_res_pile_manip_amd.planners import world2cam_affine

pytestmark = pytest.mark.gpu


def test_call_order_and_shape_errors(golden):
    eng = Engine(0)
    s0, dens, attr = syn.make_pile(40, 1, seed=0)
    acts = syn.sample_pushes(4, 2, seed=0)
    with pytest.raises(DrpError, match='weights'):
        eng.rollout(s0, attr, dens, acts)
    blob = weights.blob_from_state_dict(golden.weights_seed0)
    with pytest.raises(DrpError, match='38403|expected'):
        eng.load_weights(blob[:-1], 0.08)
    eng.load_weights(blob, 0.08)
    with pytest.raises(DrpError, match='camera'):
        eng.rollout(s0, attr, dens, acts)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    states, _ = eng.rollout(s0, attr, dens, acts)
    assert states.shape == (4, 2, 40, 3)
    with pytest.raises(DrpError, match='goal'):
        eng.rollout(s0, attr, dens, acts, want_states=False, want_reward=True)
    with pytest.raises(DrpError, match='goal|gd'):
        eng.gd_begin(s0, attr, dens, acts, 0.05, *syn.action_limits())
    with pytest.raises(DrpError):
        eng.set_engine(17)
    with pytest.raises(DrpError, match='N <= 4096|shape'):
        big = np.zeros((1, 5000, 3), np.float32)
        eng.rollout(big, np.zeros((1, 5000), np.float32), dens, acts)
    with pytest.raises(DrpError, match='multiple'):
        s2, d2, a2 = syn.make_pile(40, 3, seed=0)
        eng.rollout(s2, a2, d2, acts)                  # 4 rows are not a multiple of 3 batch columns
    with pytest.raises(DrpError, match='train_begin'):
        eng.lib.drp_train_set_lr.restype  # noqa: B018  (symbol exists)
        eng._ck(eng.lib.drp_train_set_lr(eng.h, 1e-3))
    # the context still works after every failure
    again, _ = eng.rollout(s0, attr, dens, acts)
    np.testing.assert_array_equal(again, states)
    eng.close()


def test_create_on_a_missing_device():
    with pytest.raises(DrpError, match='device'):
        Engine(63)
