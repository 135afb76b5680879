"""Host mirror of the reference's training script body (`train/train_gnn_dyn.py`), row f4.

  collate_fn(data)                     :20-45   variable particle counts -> zero-padded batch
  DeviceAdam(model, lr, betas)         :128-131 torch.optim.Adam(model.parameters(), ...) on the device
  run_batch(model, optimizer, data, phase)      :159-210 the loop body for one batch
  train(config, datasets, ...)         :134-246 epochs over 'train' / 'valid' phases, best-model tracking

The forward, the loss, the backward pass (state and weight gradients) and the Adam update run
in `drp_train_step` on the MI355X; nothing here computes on the host.  Data loading
(`dataset/dataset_gnn_dyn.py`: depth PNGs, pickled actions) is outside the path: `datasets`
is any pair of iterables of samples shaped like `ParticleDataset.__getitem__`'s return.
"""
import numpy as np


def _np(x, dtype=np.float32):
    if hasattr(x, 'detach'):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=dtype)


class PaddedBatch(tuple):
    """What `collate_fn` returns: the reference's 6-tuple (states, states_delta, attr, particle_num,
    particle_den, color_imgs), plus the ragged layout it was built from."""
    offsets = None          # [B+1] running particle offset of each sample in the concatenated cloud


def collate_fn(data):
    """Contract of train/train_gnn_dyn.py:20-45: samples with different particle counts become one batch
    zero-padded to the largest count -- states [B,T,n_max,3], states_delta [B,T-1,n_max,3], attr [B,T,n_max],
    particle_num [B] int32, particle_den [B] float32, color_imgs.

    Built as one ragged pack instead of a per-sample copy loop: the samples' particle axes are concatenated
    ([T, sum n, 3]), every particle gets its (sample, slot) address from the running offsets, and a single
    fancy-indexed store per field places the whole batch.  The offsets stay on the result (`.offsets`)."""
    counts = np.fromiter((int(d[3]) for d in data), dtype=np.int64, count=len(data))
    offsets = np.concatenate([[0], np.cumsum(counts)])
    B, n_max = len(data), int(counts.max())
    owner = np.repeat(np.arange(B), counts)                      # sample of every packed particle
    slot = np.arange(offsets[-1]) - offsets[owner]               # its index inside that sample
    def pack(field, time_axis_len=None):
        cloud = np.concatenate([_np(d[field]) for d in data], axis=1)           # [T, sum n(, 3)]
        out = np.zeros((B, cloud.shape[0], n_max) + cloud.shape[2:], dtype=np.float32)
        out[owner, :, slot] = np.moveaxis(cloud, 1, 0)
        return out
    imgs = None if data[0][5] is None else np.stack([np.asarray(d[5], dtype=np.float32) for d in data])
    batch = PaddedBatch((pack(0), pack(1), pack(2), counts.astype(np.int32),
                         np.asarray([d[4] for d in data], dtype=np.float32), imgs))
    batch.offsets = offsets
    return batch


class DeviceAdam(object):
    """torch.optim.Adam(model.parameters(), lr=lr, betas=(beta1, 0.999)) whose state lives in the
    model's engine (train/train_gnn_dyn.py:128-131)."""

    def __init__(self, model, lr, betas=(0.9, 0.999), n_rollout=5):
        if betas[1] != 0.999:
            raise NotImplementedError('beta2 is fixed at 0.999 as in the reference')
        self.model = model
        self.param_groups = [{'lr': float(lr)}]
        model.engine.train_begin(n_rollout, lr, betas[0])

    def set_lr(self, lr):
        self.param_groups[0]['lr'] = float(lr)
        self.model.engine.train_set_lr(lr)


def run_batch(model, optimizer, data, phase='train', n_rollout=None):
    """The loop body at train/train_gnn_dyn.py:159-210 -> loss (python float, what loss.item() is there)."""
    states, states_delta, attrs, particle_nums, particle_dens = [data[i] for i in range(5)]
    states = _np(states)
    B, length, n_obj, _ = states.shape
    if n_rollout is not None:
        assert length == n_rollout + 1                  # :166 (n_history = 1)
    mode = 'update' if phase == 'train' else 'eval'
    if hasattr(model, '_claim'):
        model._claim()                                  # models share the process's context: this one's weights in
        model._device_ahead = model._device_ahead or mode == 'update'
    loss, _ = model.engine.train_step(states, _np(states_delta), _np(attrs), _np(particle_nums, np.int32),
                                      _np(particle_dens), mode=mode)
    return loss


class AverageMeter(object):
    """utils.AverageMeter as the training loop uses it (:156, :205)."""

    def __init__(self):
        self.sum, self.count, self.avg = 0.0, 0, 0.0

    def update(self, val, n=1):
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def train(config, model, dataloaders, n_epoch=None, log=None, on_best=None):
    """train/train_gnn_dyn.py:134-246 without the file I/O: `dataloaders` = {'train': iterable of
    collated batches, 'valid': ...}.  Returns {'best_valid_loss', 'history': [(epoch, phase, rmse)]}."""
    tc = config['train']
    n_rollout = tc['n_rollout']
    assert tc['n_history'] == 1
    optimizer = DeviceAdam(model, float(tc['lr']), betas=(tc['adam_beta1'], 0.999), n_rollout=n_rollout)
    best_valid_loss = np.inf
    history = []
    for epoch in range(n_epoch if n_epoch is not None else tc['n_epoch']):
        for phase in ('train', 'valid'):
            model.train(phase == 'train')
            meter = AverageMeter()
            for i, data in enumerate(dataloaders[phase]):
                loss = run_batch(model, optimizer, data, phase, n_rollout)
                meter.update(loss, _np(data[0]).shape[0])
                if log is not None and i % tc['log_per_iter'] == 0:
                    log('%s [%d][%d] LR: %.6f, Loss: %.6f (%.6f)' % (phase, epoch, i, optimizer.param_groups[0]['lr'],
                                                                      np.sqrt(loss), np.sqrt(meter.avg)))
            history.append((epoch, phase, float(np.sqrt(meter.avg))))
            if phase == 'valid' and meter.avg < best_valid_loss:
                best_valid_loss = meter.avg
                if on_best is not None:
                    on_best(model.state_dict())           # torch.save(model.state_dict(), net_best.pth), :244
    return {'best_valid_loss': float(best_valid_loss), 'history': history}
