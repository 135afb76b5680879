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


def collate_fn(data):
    """List of (states [T,n,3], states_delta [T-1,n,3], attrs [T,n], particle_num, particle_den,
    color_imgs) -> zero-padded arrays, as train/train_gnn_dyn.py:20-45 (numpy instead of torch)."""
    states, states_delta, attrs, particle_num, particle_den, color_imgs = zip(*data)
    max_len = max(particle_num)
    batch_size = len(data)
    n_time, _, n_dim = _np(states[0]).shape
    states_tensor = np.zeros((batch_size, n_time, max_len, n_dim), dtype=np.float32)
    states_delta_tensor = np.zeros((batch_size, n_time - 1, max_len, n_dim), dtype=np.float32)
    attr = np.zeros((batch_size, n_time, max_len), dtype=np.float32)
    particle_num_tensor = np.asarray(particle_num, dtype=np.int32)
    particle_den_tensor = np.asarray(particle_den, dtype=np.float32)
    for i in range(batch_size):
        states_tensor[i, :, :particle_num[i], :] = _np(states[i])
        states_delta_tensor[i, :, :particle_num[i], :] = _np(states_delta[i])
        attr[i, :, :particle_num[i]] = _np(attrs[i])
    imgs = None if color_imgs[0] is None else np.asarray(color_imgs, dtype=np.float32)
    return states_tensor, states_delta_tensor, attr, particle_num_tensor, particle_den_tensor, imgs


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
