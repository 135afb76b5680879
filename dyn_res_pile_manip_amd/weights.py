"""Checkpoint -> flat fp32 blob.

The reference stores `PropNetDiffDenModel.state_dict()` with torch.save
(train/train_gnn_dyn.py:214-215) and loads it with strict=False
(visualize_mpc.py:36-41).  The C ABI takes the 38 403 floats concatenated in the
state_dict's own key order, torch Linear layout [out,in].
"""
import numpy as np

STATE_DICT_KEYS = [
    ('model.particle_encoder.model.0.weight', (64, 5)),
    ('model.particle_encoder.model.0.bias', (64,)),
    ('model.particle_encoder.model.2.weight', (64, 64)),
    ('model.particle_encoder.model.2.bias', (64,)),
    ('model.relation_encoder.model.0.weight', (64, 6)),
    ('model.relation_encoder.model.0.bias', (64,)),
    ('model.relation_encoder.model.2.weight', (64, 64)),
    ('model.relation_encoder.model.2.bias', (64,)),
    ('model.relation_encoder.model.4.weight', (64, 64)),
    ('model.relation_encoder.model.4.bias', (64,)),
    ('model.particle_propagator.linear.weight', (64, 129)),
    ('model.particle_propagator.linear.bias', (64,)),
    ('model.relation_propagator.linear.weight', (64, 193)),
    ('model.relation_propagator.linear.bias', (64,)),
    ('model.particle_predictor.linear_0.weight', (64, 64)),
    ('model.particle_predictor.linear_0.bias', (64,)),
    ('model.particle_predictor.linear_1.weight', (3, 64)),
    ('model.particle_predictor.linear_1.bias', (3,)),
]
N_WEIGHTS = sum(int(np.prod(s)) for _, s in STATE_DICT_KEYS)
assert N_WEIGHTS == 38403


def _to_numpy(v):
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.asarray(v, dtype=np.float32)


def blob_from_state_dict(sd, strict=True):
    """dict (torch tensors or numpy arrays; keys optionally prefixed 'w/') -> blob.
    With strict=False missing keys keep `default` zeros and unexpected keys are
    ignored (what `load_state_dict(..., strict=False)` does)."""
    get = {}
    for k in (sd.files if hasattr(sd, 'files') else sd.keys()):
        kk = k[2:] if k.startswith('w/') else k
        get[kk] = k
    parts = []
    for key, shape in STATE_DICT_KEYS:
        if key in get:
            a = _to_numpy(sd[get[key]])
            if a.shape != shape:
                raise ValueError('%s: shape %s, expected %s' % (key, a.shape, shape))
        elif strict:
            raise KeyError('missing key in state_dict: %s' % key)
        else:
            a = np.zeros(shape, dtype=np.float32)
        parts.append(a.ravel())
    return np.concatenate(parts).astype(np.float32)


def state_dict_from_blob(blob):
    blob = np.asarray(blob, dtype=np.float32).ravel()
    assert blob.size == N_WEIGHTS
    out, off = {}, 0
    for key, shape in STATE_DICT_KEYS:
        n = int(np.prod(shape))
        out[key] = blob[off:off + n].reshape(shape).copy()
        off += n
    return out


def random_state_dict(seed=0, predictor_scale=0.02):
    """torch.nn.Linear's default init (U(-1/sqrt(in), 1/sqrt(in)) for weight and bias)
    with numpy, last predictor layer scaled so rollouts stay a pile.
    Used by bench.py / smoke (no checkpoint is downloadable offline)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for i in range(0, len(STATE_DICT_KEYS), 2):
        (wk, ws), (bk, bs) = STATE_DICT_KEYS[i], STATE_DICT_KEYS[i + 1]
        bound = 1.0 / np.sqrt(ws[1])
        sd[wk] = rng.uniform(-bound, bound, ws).astype(np.float32)
        sd[bk] = rng.uniform(-bound, bound, bs).astype(np.float32)
    sd['model.particle_predictor.linear_1.weight'] *= np.float32(predictor_scale)
    sd['model.particle_predictor.linear_1.bias'] *= np.float32(predictor_scale)
    return sd


def load_checkpoint(path):
    """`.pth` written by the reference's trainer -> blob (PyTorch only for the pickle)."""
    import torch
    sd = torch.load(path, map_location='cpu')
    return blob_from_state_dict(sd, strict=False)
