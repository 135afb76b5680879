"""Sample-axis sharding of the sampling planner over ranks (one process per GPU).

Samples never interact inside rollout or reward (planners.py:336-359,
env/flex_rewards.py:164-214 are batched per row), so each rank rolls out its own
contiguous block of samples with no data-path collective.  The one exchange is the
softmax-weighted update (planners.py:549-561): every rank contributes the record
    [m, Z, A[H*4], sum r, sum r^2, max r, argmax]          (6 + 4H doubles)
and the combined mean is  sum_g A_g e^(m_g - M) / sum_g Z_g e^(m_g - M),  M = max_g m_g.

Two transports for that record, both behind the `comm=` argument of
`PlannerGD.trajectory_optimization_ptcl_multi_traj`:
  * `RcclComm`: drp_mpc_update_device() all-gathers it with RCCL (xGMI) and runs the combine
    kernel -- no host hop, one collective per iteration (bench.py uses the same entry point);
  * `TorchComm`: through torch.distributed (`allgather_records`), any backend -- the world-2
    gloo tests on CPU, and two processes sharing one GPU in tests/test_gpu_sharded_planner.py.
`combine_records` is the host mirror of the combine kernel (k_mppi_update).
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous block [lo, hi) of the n_total samples owned by `rank`."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_size(H):
    return 6 + 4 * H


def make_record(reward_weight, rewards, act_seqs, sample_offset=0):
    """Host version of one rank's record (what k_mppi_partials computes on the device).
    rewards [ns] (already averaged over the batch columns), act_seqs [ns,H,4]."""
    r = np.asarray(rewards, dtype=np.float64)
    a = np.asarray(act_seqs, dtype=np.float64)
    z = float(reward_weight) * r
    m = z.max()
    w = np.exp(z - m)
    H = a.shape[1]
    rec = np.empty(record_size(H), dtype=np.float64)
    rec[0], rec[1] = m, w.sum()
    rec[2:2 + 4 * H] = (w[:, None, None] * a).sum(0).ravel()
    rec[2 + 4 * H], rec[3 + 4 * H] = r.sum(), (r * r).sum()
    rec[4 + 4 * H], rec[5 + 4 * H] = r.max(), float(np.argmax(r) + sample_offset)
    return rec


def combine_records(records, n_sample_total):
    """[n_ranks, 6+4H] -> (nominal [H,4], stats dict)."""
    rec = np.asarray(records, dtype=np.float64)
    rec = rec.reshape(-1, rec.shape[-1])
    H = (rec.shape[1] - 6) // 4
    M = rec[:, 0].max()
    scale = np.exp(rec[:, 0] - M)
    Z = (rec[:, 1] * scale).sum()
    A = (rec[:, 2:2 + 4 * H] * scale[:, None]).sum(0)
    s1, s2 = rec[:, 2 + 4 * H].sum(), rec[:, 3 + 4 * H].sum()
    n = float(n_sample_total)
    mean = s1 / n
    var = max((s2 - s1 * mean) / (n - 1.0), 0.0) if n > 1 else 0.0
    g = int(np.argmax(rec[:, 4 + 4 * H]))
    stats = {'mean': mean, 'std': float(np.sqrt(var)), 'max': rec[g, 4 + 4 * H],
             'argmax': int(rec[g, 5 + 4 * H]), 'Z': Z, 'm': M}
    return (A / Z).reshape(H, 4), stats


def allgather_records(record, group=None):
    """All-gather one rank's record over a torch.distributed process group (any backend).
    Returns [world, 6+4H] float64."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    t = torch.from_numpy(np.ascontiguousarray(record, dtype=np.float64)).to(dev)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).cpu().numpy()


# ---- elite (CEM-style) update: nominal = mean of the k best sequences over all ranks -----------------
# Not in the reference; SURVEY.md section 8e names it as the other form of the exchange.  A rank's record
# block is its k best samples as [reward, global sample index, act[4H]], best first (higher reward, ties to
# the lower index; reward -inf / index -1 pad a short rank).  Host mirrors of k_elite_local / k_elite_update.
def elite_record_size(H):
    return 2 + 4 * H


def make_elite_records(rewards, act_seqs, k, sample_offset=0):
    r = np.asarray(rewards, dtype=np.float64)
    a = np.asarray(act_seqs, dtype=np.float64)
    H = a.shape[1]
    order = np.lexsort((np.arange(r.size), -r))          # reward descending, index ascending; NaN rewards last
    order = [int(i) for i in order if not np.isnan(r[i])][:k]
    rec = np.zeros((k, elite_record_size(H)), dtype=np.float64)
    rec[:, 0], rec[:, 1] = -np.inf, -1.0
    for e, i in enumerate(order):
        rec[e, 0], rec[e, 1] = r[i], float(i + sample_offset)
        rec[e, 2:] = a[i].ravel()
    return rec


def combine_elite_records(records, k):
    """[n_ranks, k, 2+4H] -> nominal [H,4] (mean of the k best of all records), elite size, worst elite reward."""
    rec = np.asarray(records, dtype=np.float64)
    rec = rec.reshape(-1, rec.shape[-1])
    rec = rec[rec[:, 1] >= 0.0]
    order = np.lexsort((rec[:, 1], -rec[:, 0]))[:k]
    H = (rec.shape[1] - 2) // 4
    el = rec[order]
    return el[:, 2:].mean(0).reshape(H, 4), len(order), float(el[-1, 0])


# ---- what the planner is handed to shard its sample axis (comm=) ---------------------------------
class RcclComm(object):
    """One rank of an RCCL communicator attached to the engine's context: the planner's update runs as
    drp_mpc_update_device / drp_mpc_update_elite_device (partials -> ncclAllGather -> combine on the
    stream), its bookkeeping record travels through drp_comm_allgather.  `uid` = the 128-byte
    ncclUniqueId every rank received from rank 0 (Engine.comm_unique_id())."""
    device_update = True

    def __init__(self, uid, rank, n_ranks):
        self.uid, self.rank, self.n_ranks = uid, int(rank), int(n_ranks)
        self._eng = None

    def attach(self, eng):
        """Builds the communicator on first use.  A ncclUniqueId serves ONE ncclCommInitRank per rank, so the
        object stays with the engine it was first attached to: handing it a second engine raises (make a new
        RcclComm from a fresh id) instead of re-initialising with the spent id, which never returns."""
        if self._eng is None:
            eng.comm_init(self.uid, self.rank, self.n_ranks)
            self._eng = eng
        elif self._eng is not eng:
            raise RuntimeError('this RcclComm is bound to another engine: a ncclUniqueId cannot be used for a second '
                               'communicator -- create a new RcclComm from a fresh Engine.comm_unique_id()')

    def allgather(self, arr):
        return self._eng.comm_allgather(arr)


class TorchComm(object):
    """Host transport over a torch.distributed process group (gloo or nccl): the planner fetches its
    rank's record (drp_mpc_partials / drp_mpc_elite), all-gathers it here and uploads the gathered
    records to the combine kernel (drp_mpc_update / drp_mpc_update_elite)."""
    device_update = False

    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.rank, self.n_ranks = dist.get_rank(group), dist.get_world_size(group)

    def attach(self, eng):
        pass

    def allgather(self, arr):
        return allgather_records(arr, self.group).reshape((self.n_ranks,) + np.shape(arr))


def as_comm(comm):
    """None | RcclComm | TorchComm | the documented tuple (rank, n_ranks, uid) -> a comm object or None."""
    if comm is None or hasattr(comm, 'allgather'):
        return comm
    rank, n_ranks, uid = comm
    return RcclComm(uid, rank, n_ranks)


# ---- the planner's bookkeeping over a whole call: ONE exchange at its end --------------------------------------
# planners.py:721-738 keeps, per batch column, the best reward seen so far (strictly better replaces), its trajectory
# index and pushes, and per iteration the mean / std of column 0.  Every rank can keep that for its own rows; the
# global answer is the lexicographic best of the ranks' bests -- highest reward, then the EARLIEST iteration, then the
# lowest index: what the sequential "strictly better" rule over all rows would have kept -- and per-iteration sums.
# An iteration every rank ran on the same rows (the MPPI planner's iteration 0) counts once.
def make_run_record(it_sums, it_replicated, max_reward, max_idx, best_iter, best_actions):
    """it_sums [n_it, 3] (n, sum, sum of squares of column 0), it_replicated [n_it] bool; per column: best reward,
    global trajectory index, iteration it was found in, pushes [n_batch, H, 4] -> float64 record."""
    it_sums = np.asarray(it_sums, dtype=np.float64)
    nb = len(max_reward)
    acts = np.asarray(best_actions, dtype=np.float64).reshape(nb, -1)
    body = np.concatenate([np.asarray(max_reward, np.float64)[:, None], np.asarray(max_idx, np.float64)[:, None],
                           np.asarray(best_iter, np.float64)[:, None], acts], axis=1)
    return np.concatenate([it_sums.reshape(-1), np.asarray(it_replicated, np.float64), body.reshape(-1)])


def combine_run_records(records, n_it, n_batch):
    """[n_ranks, record] -> (mean [n_it], unbiased std [n_it], ran [n_it] bool; per column: best reward, index,
    pushes [n_batch, 4H])."""
    rec = np.asarray(records, dtype=np.float64)
    R = rec.shape[0]
    sums = rec[:, :3 * n_it].reshape(R, n_it, 3)
    repl = rec[0, 3 * n_it:4 * n_it] > 0.5
    w = np.ones((R, n_it, 1))
    w[1:, repl, :] = 0.0                                   # a replicated iteration counts once: rank 0's rows
    tot = (sums * w).sum(0)
    n, s1, s2 = tot[:, 0], tot[:, 1], tot[:, 2]
    ran = n > 0
    mean = np.where(ran, s1 / np.maximum(n, 1.0), 0.0)
    var = np.where(n > 1, (s2 - s1 * mean) / np.maximum(n - 1.0, 1.0), 0.0)
    std = np.sqrt(np.maximum(var, 0.0))
    body = rec[:, 4 * n_it:].reshape(R, n_batch, -1)
    best = np.zeros(n_batch, dtype=np.int64)
    for j in range(n_batch):
        keys = [(-body[r, j, 0], body[r, j, 2], body[r, j, 1], r) for r in range(R)]
        best[j] = min(keys)[3]
    cols = np.arange(n_batch)
    return mean, std, ran, body[best, cols, 0], body[best, cols, 1].astype(np.int64), body[best, cols, 3:]
