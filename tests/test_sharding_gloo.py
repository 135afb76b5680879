"""CPU, world_size 2 / 3 / 8 over gloo: the N>1 path of the sampling planner -- contiguous sample
shards, one all-gather of the per-rank records, combine == the unsharded
optimize_action (planners.py:549-561).  The rollout/reward on each rank is played by the
oracle here (no GPU in this container); on the GPU box the same records come from
k_mppi_partials and travel over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dyn_res_pile_manip_amd import sharding, synthetic as syn, weights
    from oracle import propnet_sparse as osp
    N, H, ns_total = 24, 3, 20          # 20 samples: 10 + 10 over 2 ranks, 7 + 7 + 6 over 3, 3 + 3 + 3 + 3 + 2 + 2 + 2 + 2 over 8
    sd = weights.random_state_dict(seed=0)
    W = osp.weights_np(sd)
    M34 = osp.world2cam_affine(syn.demo_cam_extrinsics(), 24)
    s0, dens, attr = syn.make_pile(N, 1, seed=3)
    acts = syn.sample_pushes(ns_total, H, seed=3)             # every rank can rebuild the global set
    obs_goal = syn.goal_distance_image(syn.goal_mask('disc'))
    G = syn.goal_field(obs_goal)
    gc = syn.goal_coor_strided(obs_goal, 5 * N)
    lo, hi = sharding.shard_range(ns_total, rank, world)
    st = osp.rollout(W, s0, dens, attr, acts[lo:hi], M34, 24.0)
    r = osp.reward(st[:, -1], G, syn.demo_cam_params(), gc)
    rec = sharding.make_record(0.1, r, acts[lo:hi], sample_offset=lo)
    allrec = sharding.allgather_records(rec)
    nominal, stats = sharding.combine_records(allrec, ns_total)
    # the elite (CEM-style) form of the exchange: k best per rank, all-gathered, re-selected
    k = 2
    erec = sharding.make_elite_records(r, acts[lo:hi], k, sample_offset=lo)
    eall = sharding.allgather_records(erec.ravel()).reshape(world, k, -1)
    e_nominal, e_n, e_worst = sharding.combine_elite_records(eall, k)
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), nominal=nominal, r=r, lo=lo, hi=hi,
             mean=stats['mean'], std=stats['std'], argmax=stats['argmax'], e_nominal=e_nominal, e_n=e_n, e_worst=e_worst)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_mppi_update_equals_single_rank(tmp_path, world):
    """2 ranks; 3 (uneven shards); 8 -- the node BASELINE configs[2] and [4] are quoted on."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from dyn_res_pile_manip_amd import synthetic as syn
    from oracle import propnet_dense as od
    got = [np.load(os.path.join(str(tmp_path), 'rank%d.npz' % r)) for r in range(world)]
    from dyn_res_pile_manip_amd.sharding import shard_range
    for q in range(1, world):
        np.testing.assert_array_equal(got[0]['nominal'], got[q]['nominal'])     # every rank agrees
    assert [(int(g['lo']), int(g['hi'])) for g in got] == [shard_range(20, q, world) for q in range(world)]
    r_all = np.concatenate([g['r'] for g in got])
    acts = syn.sample_pushes(20, 3, seed=3)
    expect = od.optimize_action(acts, r_all, 0.1)
    np.testing.assert_allclose(got[0]['nominal'], expect, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got[0]['mean'], r_all.astype(np.float64).mean(), rtol=1e-12)
    np.testing.assert_allclose(got[0]['std'], r_all.astype(np.float64).std(ddof=1), rtol=1e-9)
    assert int(got[0]['argmax']) == int(np.argmax(r_all))
    # elite update: mean of the 2 best of all 20 samples, whichever rank held them
    for q in range(1, world):
        np.testing.assert_array_equal(got[0]['e_nominal'], got[q]['e_nominal'])
    best = np.lexsort((np.arange(20), -r_all.astype(np.float64)))[:2]
    np.testing.assert_allclose(got[0]['e_nominal'], acts[best].astype(np.float64).mean(0), rtol=1e-12, atol=1e-12)
    assert int(got[0]['e_n']) == 2 and float(got[0]['e_worst']) == float(r_all[best[-1]])


def test_shard_range_covers_everything():
    from dyn_res_pile_manip_amd.sharding import shard_range
    for n, w in [(10, 3), (1024, 8), (7, 8), (8192, 8)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_run_records_fold_a_whole_call_like_the_sequential_bookkeeping():
    """The planner's bookkeeping over a whole call with ONE exchange at its end (sharding.make_run_record /
    combine_run_records): every rank keeps the running per-column best of its own rows (strictly better replaces) and
    per-iteration sums; the merge -- highest reward, then the earliest iteration, then the lowest index -- equals the
    sequential rule of planners.py:721-727 over all rows, ties across ranks and across iterations included; iteration
    0 is run by every rank on the same rows and counts once."""
    from dyn_res_pile_manip_amd import sharding
    rng = np.random.default_rng(11)
    ns, nb, H, n_it, R = 23, 3, 2, 5, 3
    r = rng.normal(-30, 5, (n_it, ns, nb)).astype(np.float32)
    a = rng.normal(0, 2, (n_it, ns, nb, H, 4)).astype(np.float32)
    top = r.max() + 1.0
    r[2, 20, 1] = r[1, 4, 1] = r[1, 17, 1] = top          # one tie inside an iteration, one across iterations
    r[3, 2, 2] = r[3, 22, 2] = top                        # a tie across ranks in the same iteration
    # the sequential rule over all rows (iteration 0: the replicated rows, global index = row)
    want_max = np.full(nb, -np.inf, np.float32); want_idx = np.zeros(nb, np.int64); want_act = np.zeros((nb, H, 4), np.float32)
    for it in range(n_it):
        cur, idx = r[it].max(0), r[it].argmax(0)
        better = cur > want_max
        want_max[better] = cur[better]; want_idx[better] = idx[better]
        want_act[better] = a[it][idx, np.arange(nb)][better]
    recs = []
    for rank in range(R):
        lo, hi = sharding.shard_range(ns, rank, R)
        mx = np.full(nb, -np.inf, np.float32); mi = np.zeros(nb, np.int64); bi = np.zeros(nb, np.int64)
        ma = np.zeros((nb, H, 4), np.float32)
        sums = np.zeros((n_it + 2, 3)); repl = np.zeros(n_it + 2, bool)      # two iterations more than were run
        for it in range(n_it):
            l, h = (0, ns) if it == 0 else (lo, hi)
            rr = r[it, l:h]
            cur, idx = rr.max(0), rr.argmax(0)
            better = cur > mx
            mx[better] = cur[better]; mi[better] = idx[better] + l; bi[better] = it
            ma[better] = a[it, l:h][idx, np.arange(nb)][better]
            c0 = rr[:, 0].astype(np.float64)
            sums[it] = (h - l, c0.sum(), (c0 * c0).sum()); repl[it] = it == 0
        recs.append(sharding.make_run_record(sums, repl, mx, mi, bi, ma))
    mean, std, ran, cmax, cidx, cact = sharding.combine_run_records(np.stack(recs), n_it + 2, nb)
    assert ran.tolist() == [True] * n_it + [False, False]
    for it in range(n_it):
        np.testing.assert_allclose(mean[it], r[it, :, 0].astype(np.float64).mean(), rtol=1e-12)
        np.testing.assert_allclose(std[it], r[it, :, 0].astype(np.float64).std(ddof=1), rtol=1e-9)
    np.testing.assert_array_equal(cmax, want_max)
    np.testing.assert_array_equal(cidx, want_idx)
    assert cidx[1] == 4 and cidx[2] == 2
    np.testing.assert_array_equal(cact.reshape(nb, H, 4), want_act)
