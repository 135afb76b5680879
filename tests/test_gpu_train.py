"""GPU: training on the same kernels (row f4) through the C ABI, against the reference model's
own loss, autograd gradients and torch.optim.Adam trajectory (tests/golden/train.npz)."""
import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn
from dyn_res_pile_manip_amd import train_gnn_dyn as T
from dyn_res_pile_manip_amd import weights
from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel

pytestmark = pytest.mark.gpu
CASES = ['b4_r3', 'b2_r5']


def _model(golden):
    import torch
    config = syn.default_config()
    model = PropNetDiffDenModel(config, True)
    model.load_state_dict({k[2:]: torch.from_numpy(golden.weights_seed0[k]) for k in golden.weights_seed0.files
                           if k.startswith('w/')}, strict=False)
    return model


def _batch(g, case):
    return [g[case + '/' + k] for k in ('states', 'states_delta', 'attrs', 'particle_nums', 'particle_dens')]


@pytest.mark.parametrize('case', CASES)
def test_loss_and_weight_gradients(golden, case):
    g = golden.train
    model = _model(golden)
    batch = _batch(g, case)
    model.engine.train_begin(batch[0].shape[1] - 1, 1e-3, 0.9)
    loss, grad = model.engine.train_step(*batch, mode='grad', want_grad=True)
    loss_eval, _ = model.engine.train_step(*batch, mode='eval')
    model.engine.close()
    assert abs(loss - g[case + '/losses'][0]) < 1e-4 * g[case + '/losses'][0]
    assert abs(loss_eval - loss) < 1e-9
    got = weights.state_dict_from_blob(grad)
    for k, _ in weights.STATE_DICT_KEYS:
        ref = g[case + '/grad/' + k]
        gk = np.asarray(got[k]).reshape(ref.shape)
        scale = max(np.abs(ref).max(), 1e-8)
        assert np.abs(gk - ref).max() < 2e-4 * scale + 1e-9, k
        # inputs that are identically zero (the attribute columns) have exactly zero gradient in both
        np.testing.assert_array_equal(ref == 0, np.abs(gk) < 1e-12 * scale + 1e-30)


@pytest.mark.parametrize('case', CASES)
def test_adam_trajectory(golden, case):
    g = golden.train
    model = _model(golden)
    batch = _batch(g, case)
    lr, beta1 = g[case + '/lr_beta1']
    opt = T.DeviceAdam(model, lr, betas=(beta1, 0.999), n_rollout=batch[0].shape[1] - 1)
    losses = [T.run_batch(model, opt, batch, 'train') for _ in range(3)]
    sd = model.state_dict()
    model.engine.close()
    np.testing.assert_allclose(losses, g[case + '/losses'], rtol=2e-3)
    for k, _ in weights.STATE_DICT_KEYS:
        ref = g[case + '/after3/' + k]
        got = sd[k].numpy() if hasattr(sd[k], 'numpy') else np.asarray(sd[k])
        gr = g[case + '/grad/' + k]
        firm = np.abs(gr) > 1e-3 * np.abs(gr).max()           # Adam's first steps are +-lr: sign of tiny gradients is noise
        assert np.abs(got - ref)[firm].max() < 2e-5, k
        assert np.abs(got - ref).max() < 3.5 * lr, k


def test_training_reduces_the_loss_and_valid_phase_leaves_weights(golden):
    g = golden.train
    model = _model(golden)
    config = syn.default_config()
    config['train'].update({'n_rollout': 3, 'n_history': 1, 'lr': 2e-4, 'adam_beta1': 0.9, 'log_per_iter': 50,
                            'n_epoch': 6})
    batch = _batch(g, 'b4_r3') + [None]
    w0 = model.engine.get_weights().copy()
    best = []
    res = T.train(config, model, {'train': [batch] * 4, 'valid': [batch]}, on_best=lambda sd: best.append(sd))
    rmse_valid = [h[2] for h in res['history'] if h[1] == 'valid']
    assert rmse_valid[-1] < 0.99 * rmse_valid[0] and min(rmse_valid) == rmse_valid[-1]
    assert len(best) >= 2 and abs(res['best_valid_loss'] - min(rmse_valid) ** 2) < 1e-12
    assert np.abs(model.engine.get_weights() - w0).max() > 1e-4
    # the updated weights serve inference: one step with the trained model differs from the initial one
    w1 = model.engine.get_weights()
    l_a, _ = model.engine.train_step(*batch[:5], mode='eval')
    l_b, _ = model.engine.train_step(*batch[:5], mode='eval')
    assert l_a == l_b
    np.testing.assert_array_equal(model.engine.get_weights(), w1)
    model.engine.close()


def test_collate_fn_pads_like_the_reference():
    rng = np.random.default_rng(0)
    data = []
    for n in (5, 9, 3):
        data.append((rng.normal(size=(4, n, 3)), rng.normal(size=(3, n, 3)), np.zeros((4, n)), n, 100.0 + n, None))
    st, sd, at, pn, pd, _ = T.collate_fn(data)
    assert st.shape == (3, 4, 9, 3) and sd.shape == (3, 3, 9, 3) and at.shape == (3, 4, 9)
    assert st.dtype == np.float32 and pn.dtype == np.int32 and pd.dtype == np.float32
    assert (st[0, :, 5:] == 0).all() and (st[2, :, 3:] == 0).all()
    np.testing.assert_allclose(st[1], data[1][0].astype(np.float32))


def test_tiny_and_unpadded_batches(golden):
    """B = 1, a sample of 3 particles (fewer than the in-degree cap), and a batch without padding,
    against the dense torch-autograd oracle."""
    from oracle import propnet_dense as od
    model = _model(golden)
    W = {k[2:]: golden.weights_seed0[k] for k in golden.weights_seed0.files if k.startswith('w/')}
    rng = np.random.default_rng(0)
    for nums, T in (([3], 2), ([20, 20], 3), ([7, 1], 1)):
        B, N = len(nums), max(nums)
        states = np.zeros((B, T + 1, N, 3), np.float32)
        sdelta = np.zeros((B, T, N, 3), np.float32)
        attrs = np.zeros((B, T + 1, N), np.float32)
        dens = np.array([300.0 + 50 * b for b in range(B)], np.float32)
        for b, n in enumerate(nums):
            s, _, _ = syn.make_pile(n, 1, seed=5 + b, kind='blob')
            for t in range(T + 1):
                states[b, t, :n] = s[0] * 0.3 + 0.002 * t * rng.standard_normal((n, 3)).astype(np.float32) + [0, 0, 0.52]
            sdelta[b, :, :n] = 0.004 * rng.standard_normal((T, n, 3)).astype(np.float32)
        pn = np.asarray(nums, np.int32)
        model.engine.train_begin(T, 1e-3, 0.9)
        loss, grad = model.engine.train_step(states, sdelta, attrs, pn, dens, mode='grad', want_grad=True)
        ref_loss, ref_grads = od.train_loss_and_grads(W, states, sdelta, attrs, pn, dens)
        assert abs(loss - ref_loss) < 1e-4 * abs(ref_loss)
        got = weights.state_dict_from_blob(grad)
        for k, _ in weights.STATE_DICT_KEYS:
            scale = max(np.abs(ref_grads[k]).max(), 1e-8)
            assert np.abs(np.asarray(got[k]).reshape(ref_grads[k].shape) - ref_grads[k]).max() < 5e-4 * scale + 1e-9, (nums, k)
    model.engine.close()


def test_checkpoint_round_trip_is_a_torch_state_dict(golden, tmp_path):
    """train/train_gnn_dyn.py:214-215,226: `torch.save(model.state_dict(), path)` must write what torch's
    `load_state_dict` accepts -- tensors under the reference's keys and shapes -- and what this package's own
    loader reads back bit for bit."""
    import torch
    model = _model(golden)
    sd = model.state_dict()
    blob = model.engine.get_weights()
    model.engine.close()
    assert list(sd.keys()) == [k for k, _ in weights.STATE_DICT_KEYS]
    for k, shape in weights.STATE_DICT_KEYS:
        assert isinstance(sd[k], torch.Tensor) and sd[k].dtype == torch.float32 and tuple(sd[k].shape) == shape
    # a torch module with the reference's parameter names (model/gnn_dyn.py:127-145) takes it with strict=True
    class _Named(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for k, shape in weights.STATE_DICT_KEYS:
                self.register_parameter(k.replace('.', '__'), torch.nn.Parameter(torch.zeros(shape)))

        def state_dict(self, *a, **kw):
            return {k.replace('__', '.'): v for k, v in super().state_dict(*a, **kw).items()}

        def load_state_dict(self, sd_, strict=True):
            return super().load_state_dict({k.replace('.', '__'): v for k, v in sd_.items()}, strict=strict)
    path = str(tmp_path / 'net_best.pth')
    torch.save(sd, path)
    m = _Named()
    m.load_state_dict(torch.load(path, map_location='cpu'), strict=True)
    np.testing.assert_array_equal(weights.blob_from_state_dict(m.state_dict()), blob)
    np.testing.assert_array_equal(weights.load_checkpoint(path), blob)


def test_device_repack_equals_the_host_packers(golden, monkeypatch):
    """After an optimiser step the packed copies of the weights are rebuilt ON THE DEVICE (gathers through index maps
    derived from the host packers, and pack_split / pack_split6 repeated element by element).  Three Adam steps, then
    every packed buffer byte by byte against a second context that was handed the same blob through
    drp_load_weights (the host packers); and the whole trajectory against the round-2 path (DRP_NO_REPACK_DEVICE=1)."""
    from dyn_res_pile_manip_amd.engine import Engine
    g = golden.train
    case = 'b4_r3'
    batch = _batch(g, case)
    lr, beta1 = g[case + '/lr_beta1']
    runs = {}
    for host in (False, True):
        if host:
            monkeypatch.setenv('DRP_NO_REPACK_DEVICE', '1')
        else:
            monkeypatch.delenv('DRP_NO_REPACK_DEVICE', raising=False)
        model = _model(golden)
        eng = model.engine
        eng.train_begin(batch[0].shape[1] - 1, float(lr), float(beta1))
        losses = [eng.train_step(*batch, mode='update')[0] for _ in range(3)]
        blob = eng.get_weights()
        packed = {}
        for name, nbytes in (('w_raw', 38403 * 4), ('w_valu', None), ('w_mfma', None), ('w_mfma_bwd', None), ('w_split', None),
                             ('w_split6', None)):
            buf = np.zeros(1 << 20, np.uint8)
            n = eng.lib.drp_debug_fetch(eng.h, name.encode(), buf.ctypes.data, buf.nbytes)
            assert n > 0 and (nbytes is None or n == nbytes), (name, n)
            packed[name] = buf[:n].copy()
        runs[host] = (losses, blob, packed)
        if not host:
            np.testing.assert_array_equal(packed['w_raw'].view(np.float32), blob)      # the host copy follows the device's
            ref = Engine(0)
            ref.load_weights(blob, 0.08)
            for name in ('w_valu', 'w_mfma', 'w_mfma_bwd', 'w_split', 'w_split6'):
                buf = np.zeros(1 << 20, np.uint8)
                n = ref.lib.drp_debug_fetch(ref.h, name.encode(), buf.ctypes.data, buf.nbytes)
                assert n == packed[name].size, name
                assert np.array_equal(buf[:n], packed[name]), name
            ref.close()
        eng.close()
    assert runs[False][0] == runs[True][0]                       # the same losses, bit for bit
    np.testing.assert_array_equal(runs[False][1], runs[True][1])
    for name in runs[False][2]:
        assert np.array_equal(runs[False][2][name], runs[True][2][name]), name


@pytest.mark.parametrize('case', CASES)
def test_deferred_weight_gradients_equal_the_flushed_ones(golden, monkeypatch, case):
    """The weight gradients of an iteration go out at the END of the backward pass (every operand keeps a buffer per rollout
    step and propagation step; one launch per job size + one reduction that adds a matrix's jobs in queue order) instead of
    25 launch pairs in between (DRP_NO_WGRAD_DEFER=1): the same gradients and the same Adam trajectory, bit for bit --
    with the matrix-core and the VALU outer products alike."""
    g = golden.train
    batch = _batch(g, case)
    lr, beta1 = g[case + '/lr_beta1']
    # the stage kernels on both sides (the deferred jobs' default producer, kmb_step_bwd<dump>, adds the receiver term by bit
    # planes: last-bit differences from kb_edge_terms; test_the_one_launch_backward_pass_of_the_trainer compares the two)
    monkeypatch.setenv('DRP_NO_BWD_FUSED', '1')
    for valu in (False, True):
        runs = []
        for flushed in (False, True):
            for k, on in (('DRP_NO_WGRAD_DEFER', flushed), ('DRP_NO_WGRAD_MFMA', valu)):
                if on:
                    monkeypatch.setenv(k, '1')
                else:
                    monkeypatch.delenv(k, raising=False)
            model = _model(golden)
            eng = model.engine
            eng.train_begin(batch[0].shape[1] - 1, float(lr), float(beta1))
            loss, grad = eng.train_step(*batch, mode='grad', want_grad=True)
            losses = [eng.train_step(*batch, mode='update')[0] for _ in range(3)]
            runs.append((loss, grad, losses, eng.get_weights()))
            eng.close()
        assert np.isfinite(runs[0][1]).all() and np.abs(runs[0][1]).max() > 0
        assert runs[0][0] == runs[1][0] and runs[0][2] == runs[1][2]
        np.testing.assert_array_equal(runs[0][1], runs[1][1])
        np.testing.assert_array_equal(runs[0][3], runs[1][3])


def test_training_through_the_valu_stage_kernels(golden, monkeypatch):
    """DRP_BWD_VALU_STAGES=1: the trainer's node stages on the VALU row kernels (the path batches below KMB_MIN_TILES tiles took
    until round 3; no default batch reaches it now) -- the reference's loss and every parameter's gradient."""
    monkeypatch.setenv('DRP_BWD_VALU_STAGES', '1')
    from dyn_res_pile_manip_amd.engine import Engine
    g = golden.train
    case = 'b2_r5'
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(golden.weights_seed0), 0.08)
    batch = _batch(g, case)
    eng.train_begin(batch[0].shape[1] - 1, 1e-3, 0.9)
    eng.dispatch_reset()
    loss, grad = eng.train_step(*batch, mode='grad', want_grad=True)
    ran = eng.last_dispatch()
    eng.close()
    assert 'train:stages kb_*' in ran and 'train:stages kmb_*' not in ran, ran
    assert abs(loss - g[case + '/losses'][0]) < 1e-4 * g[case + '/losses'][0]
    got = weights.state_dict_from_blob(grad)
    for k, _ in weights.STATE_DICT_KEYS:
        ref = g[case + '/grad/' + k]
        scale = max(np.abs(ref).max(), 1e-8)
        assert np.abs(np.asarray(got[k]).reshape(ref.shape) - ref).max() < 2e-4 * scale + 1e-9, k


@pytest.mark.parametrize('case', CASES)
def test_the_one_launch_backward_pass_of_the_trainer(golden, monkeypatch, case):
    """Default: one launch per rollout step for everything between the loss gradient and the relation encoder's backward
    (kmb_step_bwd<dump, coop>: a group of samples shared by as many workgroups as there are CUs for, a barrier in memory
    between the phases; the edge terms of a tile gathered by the whole workgroup).  Against the stage kernels
    (DRP_NO_BWD_FUSED=1): the same loss bit for bit (the forward pass is the same), the gradients to rounding.  And the SAME
    BITS however the tiles are dealt and the edge terms gathered: one workgroup per group (DRP_TRAIN_PARTS=1: tiles on demand,
    __syncthreads between the phases) or three, a wave's own gather (DRP_TRAIN_COOP=0) or the workgroup's."""
    g = golden.train
    batch = _batch(g, case)
    lr, beta1 = g[case + '/lr_beta1']
    runs = {}
    variants = (('default', {}), ('one', {'DRP_TRAIN_PARTS': '1'}), ('three', {'DRP_TRAIN_PARTS': '3'}),
                ('wave', {'DRP_TRAIN_COOP': '0'}), ('wave-one', {'DRP_TRAIN_COOP': '0', 'DRP_TRAIN_PARTS': '1'}),
                ('coop-one', {'DRP_TRAIN_COOP': '1', 'DRP_TRAIN_PARTS': '1'}), ('stages', {'DRP_NO_BWD_FUSED': '1'}))
    for name, env in variants:
        for k in ('DRP_TRAIN_PARTS', 'DRP_TRAIN_COOP', 'DRP_NO_BWD_FUSED'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = _model(golden)
        eng = model.engine
        eng.train_begin(batch[0].shape[1] - 1, float(lr), float(beta1))
        eng.dispatch_reset()
        loss, grad = eng.train_step(*batch, mode='grad', want_grad=True)
        ran = eng.last_dispatch()
        losses = [eng.train_step(*batch, mode='update')[0] for _ in range(3)]
        runs[name] = (loss, grad, losses, eng.get_weights(), ran)
        eng.close()
    assert 'train:kmb_step_bwd<dump,coop>' in runs['default'][4] and 'train:stages kmb_*' not in runs['default'][4], runs['default'][4]
    assert "train:kmb_step_bwd<dump>" in list(runs["wave"][4]), runs["wave"][4]
    assert 'train:stages kmb_*' in runs['stages'][4] and 'train:kmb_step_bwd' not in runs['stages'][4], runs['stages'][4]
    for other in ('one', 'three', 'wave', 'wave-one', 'coop-one'):
        assert runs[other][0] == runs['default'][0] and runs[other][2] == runs['default'][2], other
        np.testing.assert_array_equal(runs[other][1], runs['default'][1], err_msg=other)
        np.testing.assert_array_equal(runs[other][3], runs['default'][3], err_msg=other)
    assert runs['stages'][0] == runs['default'][0]
    a = weights.state_dict_from_blob(runs['stages'][1])
    b = weights.state_dict_from_blob(runs['default'][1])
    for k, _ in weights.STATE_DICT_KEYS:
        scale = max(np.abs(np.asarray(a[k])).max(), 1e-8)
        assert np.abs(np.asarray(a[k]) - np.asarray(b[k])).max() < 2e-5 * scale, k
    np.testing.assert_allclose(runs['default'][2], runs['stages'][2], rtol=1e-4)


def test_a_timed_out_barrier_of_the_backward_pass_moves_nothing_and_the_step_runs_again(golden, monkeypatch):
    """kmb_step_bwd's barrier among the workgroups of a group gives up after two seconds (a shared or masked device: its
    workgroups are not all resident) and leaves a flag; the gradient of that pass is partial.  The optimiser step reads the flag
    on the device and moves neither the weights nor the moments, the iteration count stays, and drp_train_step runs the step
    again with one workgroup per group.  Forced here (DRP_DEBUG_FORCE_GIVEUP=1 sets the flag after the first pass):
    losses, gradient and weights are the bits of a context whose barrier never gave up running with DRP_TRAIN_PARTS=1 from the
    start -- three update iterations in, Adam's bias correction included."""
    g = golden.train
    case = 'b4_r3'
    batch = _batch(g, case)
    lr, beta1 = g[case + '/lr_beta1']
    runs = {}
    for name, env in (('retry', {'DRP_DEBUG_FORCE_GIVEUP': '1'}), ('one', {'DRP_TRAIN_PARTS': '1'})):
        for k in ('DRP_DEBUG_FORCE_GIVEUP', 'DRP_TRAIN_PARTS'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model = _model(golden)
        eng = model.engine
        eng.train_begin(batch[0].shape[1] - 1, float(lr), float(beta1))
        eng.dispatch_reset()
        loss, grad = eng.train_step(*batch, mode='grad', want_grad=True)
        losses = [eng.train_step(*batch, mode='update')[0] for _ in range(3)]
        runs[name] = (loss, grad, losses, eng.get_weights(), eng.last_dispatch())
        eng.close()
    assert any('barrier gave up' in v for v in runs['retry'][4]), runs['retry'][4]
    assert not any('barrier gave up' in v for v in runs['one'][4])
    assert runs['retry'][0] == runs['one'][0] and runs['retry'][2] == runs['one'][2]
    np.testing.assert_array_equal(runs['retry'][1], runs['one'][1])
    np.testing.assert_array_equal(runs['retry'][3], runs['one'][3])
