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


def test_nan_weights_are_refused_by_the_split_engines(golden):
    """fmaxf drops NaN operands: a NaN in the relation encoder used to pass the range check as a finite bound
    (ADVICE round 2).  The split engines refuse such weights.  The fp32 engines have no range to check and compute;
    their ReLU is a hardware max, which drops a NaN where torch would propagate it -- a corrupt checkpoint either way."""
    eng = Engine(0)
    blob = weights.blob_from_state_dict(golden.weights_seed0).copy()
    off = 0
    for k, shape in weights.STATE_DICT_KEYS:
        if k == 'model.relation_encoder.model.2.weight':
            break
        off += int(np.prod(shape))
    blob[off + 17] = np.nan
    eng.load_weights(blob, 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    s0, dens, attr = syn.make_pile(40, 1, seed=0)
    acts = syn.sample_pushes(4, 2, seed=0)
    with pytest.raises(DrpError, match='outside the range'):
        eng.rollout(s0, attr, dens, acts)
    eng.set_engine(_lib.ENGINES['mfma'])
    states, _ = eng.rollout(s0, attr, dens, acts)
    assert states.shape == (4, 2, 40, 3)
    eng.close()


def test_default_clip_box_is_inside_the_proven_envelope(golden):
    """The longest push of the default clip box is 8.5 sqrt(2) / 24 = 0.50 camera-frame units: with the rotation's
    spectral norm (1) it sits inside the envelope the range shift was proven for (|s_r - s_s| <= 1.5); a workspace
    three times as wide does not, and is still accepted by the per-call bound with seed-0 weights."""
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(golden.weights_seed0), 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    eng.set_goal_image(syn.goal_distance_image(syn.goal_mask('I')), 200, 0, 'exact')
    s0, dens, attr = syn.make_pile(40, 1, seed=0)
    lo, hi = syn.action_limits()
    nominal = syn.nominal_pushes(2, seed=0)
    for scale in (1.0, 3.0):
        eng.mpc_begin(s0, attr, dens, nominal, n_sample=8, sigma=0.6, beta_filter=0.7, reward_weight=0.1,
                      act_lo=lo * scale, act_hi=hi * scale, seed=1)
    eng.close()


def test_a_wait_behind_a_dead_collective_returns_an_error(golden, monkeypatch):
    """Hang guard: with a communicator attached the host waits poll with a deadline.  drp_debug_stall holds the stream
    the way a collective waiting for a vanished peer would: drp_sync must come back with DRP_ECOMM after
    DRP_COMM_TIMEOUT_S, the communicator aborted, and the context usable again."""
    import time
    monkeypatch.setenv('DRP_COMM_ALWAYS', '1')
    monkeypatch.setenv('DRP_COMM_TIMEOUT_S', '1')
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(golden.weights_seed0), 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    s0, dens, attr = syn.make_pile(40, 1, seed=0)
    acts = syn.sample_pushes(4, 2, seed=0)
    ref, _ = eng.rollout(s0, attr, dens, acts)
    eng.comm_init(eng.comm_unique_id(), 0, 1)
    assert eng.comm_info()['n_ranks'] == 1
    eng.sync()                                          # nothing pending: the guarded wait returns at once
    eng.debug_stall(3000)
    t0 = time.time()
    with pytest.raises(DrpError, match='waited'):
        eng.sync()
    assert 0.9 < time.time() - t0 < 2.5
    assert eng.comm_info()['n_ranks'] == 0              # aborted: no communicator any more
    eng.sync()                                          # no communicator: a plain wait, ends with the stall
    again, _ = eng.rollout(s0, attr, dens, acts)        # what needs no other rank still works
    np.testing.assert_array_equal(again, ref)
    # the failure is sticky: every step that would have combined the ranks' shards answers DRP_ECOMM -- it must not carry
    # on with this rank's data alone -- until the caller destroys the communicator (or attaches a fresh one)
    lo, hi = syn.action_limits()
    eng.set_goal_image(syn.goal_distance_image(syn.goal_mask('I')), 200, fps_init=0, mode='cv5', want=False)
    eng.mpc_begin(s0, attr, dens, syn.nominal_pushes(2, seed=0), n_sample=8, sigma=0.6, beta_filter=0.7, reward_weight=0.1,
                  act_lo=lo, act_hi=hi, seed=1, sample_offset=0)
    eng.mpc_sample(0)
    eng.mpc_rollout(False)
    for call in (eng.mpc_update_device, lambda: eng.mpc_update_elite_device(2), lambda: eng.comm_allgather(np.arange(4.0))):
        with pytest.raises(DrpError, match='aborted'):
            call()
    eng.comm_destroy()
    eng.mpc_update_device()                             # alone, by the caller's decision
    assert eng.comm_allgather(np.arange(4.0)).shape == (1, 4)
    eng.comm_init(eng.comm_unique_id(), 0, 1)           # and a fresh communicator attaches
    assert eng.comm_info()['n_ranks'] == 1
    eng.mpc_sample(1)
    eng.mpc_rollout(False)
    eng.mpc_update_device()
    eng.sync()
    eng.close()
