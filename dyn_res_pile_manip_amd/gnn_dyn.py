"""Host-side mirror of the reference's `model/gnn_dyn.py` call surface, backed by the
HIP engine through the C ABI.

  PropNetDiffDenModel(config, use_gpu)                       model/gnn_dyn.py:200-207
    .load_state_dict(state_dict, strict=False)               visualize_mpc.py:36-41
    .predict_one_step(a_cur, s_cur, s_delta, particle_dens, particle_nums=None)   :209-254
    .model.forward(a_cur, s_cur, s_delta, Rr, Rs, particle_dens)                  :147-198

Arguments may be torch tensors (any device) or numpy arrays; the result comes back in
the same kind.  No computation happens on the host: a missing GPU or a missing
libdrp.so raises.
"""
import numpy as np

from . import weights as _weights
from .engine import default_engine


def _to_np(x):
    if hasattr(x, 'detach'):
        return x.detach().cpu().numpy().astype(np.float32, copy=False), x
    return np.asarray(x, dtype=np.float32), None


def _like(out, proto):
    if proto is None:
        return out
    import torch
    return torch.from_numpy(out).to(device=proto.device, dtype=proto.dtype)


def relations_to_lists(Rr, Rs):
    """Dense one-hot Rr/Rs [B,E,N] (model/gnn_dyn.py:248-251) -> receiver-major lists
    nbr_idx [B,N,10] int16, nbr_cnt [B,N] uint8 (edge order preserved per receiver)."""
    Rr, _ = _to_np(Rr)
    Rs, _ = _to_np(Rs)
    B, E, N = Rr.shape
    idx = -np.ones((B, N, 10), dtype=np.int16)
    cnt = np.zeros((B, N), dtype=np.uint8)
    bs, es = np.nonzero(Rr.sum(2) > 0.5)                 # edges in (sample, edge number) order
    recv = Rr[bs, es].argmax(1)
    send = Rs[bs, es].argmax(1)
    # slot of an edge = how many earlier edges of the sample have the same receiver: a stable sort by (sample,
    # receiver) keeps the edge order inside a group, the position inside the group is the slot
    key = bs.astype(np.int64) * N + recv
    order = np.argsort(key, kind='stable')
    ks = key[order]
    first = np.ones(ks.shape, dtype=bool)
    first[1:] = ks[1:] != ks[:-1]
    start = np.maximum.accumulate(np.where(first, np.arange(ks.size), 0))
    slot = np.arange(ks.size) - start
    if slot.size and slot.max() >= 10:
        w = order[np.argmax(slot >= 10)]
        raise ValueError('receiver %d of sample %d has more than 10 in-edges' % (recv[w], bs[w]))
    idx[bs[order], recv[order], slot] = send[order]
    np.add.at(cnt, (bs, recv), 1)
    return idx, cnt


def mask_lists(idx, particle_nums):
    """model/gnn_dyn.py:238-241 on receiver-major lists: rows and columns beyond particle_nums[b] leave the graph.
    idx [B,N,10] (-1 padded) -> (idx with the surviving senders compacted to the front in their order, cnt)."""
    B, N, K = idx.shape
    n = np.asarray(particle_nums).reshape(B, 1, 1).astype(np.int64)
    keep = (idx >= 0) & (idx < n) & (np.arange(N).reshape(1, N, 1) < n)
    order = np.argsort(~keep, axis=2, kind='stable')
    out = np.take_along_axis(idx, order, 2)
    out[~np.take_along_axis(keep, order, 2)] = -1
    return np.ascontiguousarray(out, dtype=np.int16), keep.sum(2).astype(np.uint8)


class PropModuleDiffDen(object):
    """`model.model` of the reference: forward() with explicit relations."""

    def __init__(self, owner):
        self._owner = owner
        self.nf_effect = owner.config['train']['particle']['nf_effect']

    def forward(self, a_cur, s_cur, s_delta, Rr, Rs, particle_dens, verbose=False):
        a, proto = _to_np(a_cur)
        s, _ = _to_np(s_cur)
        sd, _ = _to_np(s_delta)
        d, _ = _to_np(particle_dens)
        idx, cnt = relations_to_lists(Rr, Rs)
        self._owner._claim()
        return _like(self._owner.engine.forward(a, s, sd, d, idx, cnt), proto)

    __call__ = forward


class PropNetDiffDenModel(object):
    def __init__(self, config, use_gpu=True, device=0, engine=None):
        if config['train']['particle']['nf_effect'] != 64:
            raise NotImplementedError('the HIP kernels are built for nf_effect = 64')
        self.config = config
        self.adj_thresh = config['train']['particle']['adj_thresh']
        # the process's one context of the device (shared with utils.py's helpers and flex_rewards.config_reward_ptcl)
        # unless the caller brings its own
        self.engine = engine if engine is not None else default_engine(device)
        self.model = PropModuleDiffDen(self)
        self._blob = None
        self._device_ahead = False      # the engine's copy of the weights is newer than _blob (training steps)

    def _claim(self):
        """Models share the process's context: the one about to compute makes sure the engine holds ITS weights (another
        model's load_state_dict may have come in between), saving the outgoing model's first if a training step has moved
        them on the device."""
        import weakref
        eng = self.engine
        ref = getattr(eng, '_weights_owner', None)
        owner = ref() if ref is not None else None
        if owner is self:
            return
        if self._blob is None:
            # a model that never loaded weights must not compute with whichever other model's are on the shared context
            if owner is not None:
                raise RuntimeError('this model has no weights (load_state_dict), and the shared context holds another model\'s')
            return                                    # nobody's: the engine itself answers 'weights not loaded'
        if owner is not None and owner._device_ahead:
            owner._blob = eng.get_weights()
            owner._device_ahead = False
        eng.load_weights(self._blob, self.adj_thresh)
        eng._weights_owner = weakref.ref(self)

    # nn.Module look-alikes used by the reference's scripts
    def cuda(self, *a, **k):
        return self

    def eval(self):
        self.training = False
        return self

    def to(self, *a, **k):
        return self

    def load_state_dict(self, state_dict, strict=True):
        import weakref
        self._blob = _weights.blob_from_state_dict(state_dict, strict=strict)
        self.engine.load_weights(self._blob, self.adj_thresh)
        self.engine._weights_owner = weakref.ref(self)
        self._device_ahead = False
        return self

    def state_dict(self):
        """The current weights under the reference's keys, as torch tensors: what
        `torch.save(model.state_dict(), 'net_best.pth')` (train/train_gnn_dyn.py:214-215,226) must write
        for the reference's `load_state_dict` / visualize_mpc.py:36-41 to read it back."""
        if self._blob is None:
            raise RuntimeError('no weights loaded')
        self._claim()
        self._blob = self.engine.get_weights()       # training updates them on the device
        self._device_ahead = False
        import torch
        from collections import OrderedDict
        return OrderedDict((k, torch.from_numpy(v)) for k, v in _weights.state_dict_from_blob(self._blob).items())

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def predict_one_step(self, a_cur, s_cur, s_delta, particle_dens, particle_nums=None):
        a, proto = _to_np(a_cur)
        s, _ = _to_np(s_cur)
        sd, _ = _to_np(s_delta)
        d, _ = _to_np(particle_dens)
        assert a.shape == s.shape[:2]            # model/gnn_dyn.py:218-219
        assert s.shape == sd.shape
        self._claim()
        if particle_nums is not None:
            # model/gnn_dyn.py:238-241: rows/columns beyond particle_nums[b] leave the graph.
            # Unused by the MPC path and by training: build the lists on the
            # device, mask them on the host, run forward with explicit relations.
            idx, cnt = self.engine.build_graph(s, sd)
            idx, cnt = mask_lists(idx, particle_nums)
            return _like(self.engine.forward(a, s, sd, d, idx, cnt), proto)
        return _like(self.engine.step(a, s, sd, d), proto)
