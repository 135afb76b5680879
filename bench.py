#!/usr/bin/env python3
"""Benchmark of the hot path: one "step" = one sampling-MPC iteration on the device
  sample pushes (MPPI) -> H-step GNN rollout (graph rebuild + impulse + PropNet each step)
  -> final-step reward -> softmax-weighted update [-> RCCL all-gather when sharded].

Metric (BASELINE.json): particle-steps/s = samples x particles x steps / wall time.
Workload at N=1: BASELINE.json configs[1] -- 300-particle pile, 1024 MPPI samples,
10-step horizon (inputs resident in HBM).  With --gpus N the sample axis is sharded,
1024 samples per GPU (weak scaling, configs[2] at N=8).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4-50|c4-150|c4-300|c4-600|c5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

--config selects a BASELINE.json workload (default c2 at one GPU, c3 = the same per-GPU share under
--gpus N); c5 is the strong-scaling one (4096 samples in total).  --force-comm (under the launcher with
one process) attaches a one-rank RCCL communicator and runs the update through ncclAllGather.
tools/scale.sh runs the 1/2/4/8-GPU series of c3 and c5.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_TFLOPS = 157.3      # dense fp32 (vector == f32-input MFMA rate)
PEAK_BF16_TFLOPS = 2500.0    # dense fp16 / bf16 MFMA (same rate)
# BASELINE.json configs -> (particles, samples, horizon, samples are per GPU (weak) or in total (strong), label)
CONFIGS = {
    'c2': (300, 1024, 10, 'weak', 'BASELINE configs[1]: 300-particle pile, 1024 MPPI samples, 10-step horizon, 1 GPU'),
    'c3': (300, 1024, 10, 'weak', 'BASELINE configs[2]: 300-particle pile, 8192 MPPI samples at 8 GPUs (1024 per GPU), 10-step horizon, RCCL exchange'),
    'c4-50': (50, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 50 particles x 1024 samples x 10 steps'),
    'c4-150': (150, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 150 particles x 1024 samples x 10 steps'),
    'c4-300': (300, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 300 particles x 1024 samples x 10 steps'),
    'c4-600': (600, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 600 particles x 1024 samples x 10 steps'),
    'c5': (1200, 4096, 20, 'strong', 'BASELINE configs[4]: 1200-particle dense pile, 4096 samples in total, 20-step horizon'),
    # the reference's LIVE planner (mpc_type 'GD', config/mpc/config.yaml:38-43): 50 trajectories x 30 particle
    # re-samplings = 1500 independent Adam problems, horizon 1; a "step" = one iteration of planners.py:682-764
    # (rollout, final-step reward, reverse mode, Adam, clip box).  samples = trajectories x 30 here.
    'gd-demo': (100, 1500, 1, 'weak', "the reference's live GD planner at its demo shape: 50 trajectories x 30 re-samplings x 100 particles, horizon 1 "
                                      '(config/mpc/config.yaml:38-43), one Adam iteration per step'),
}
GD_CLASSES = ['graph', 'node_encode', 'prop', 'reward', 'tape_copy', 'bwd_reward', 'bwd_lists', 'bwd_node', 'bwd_edge',
              'bwd_push', 'opt']
KERNEL_CLASSES = ['graph', 'node_encode', 'edge_encode', 'project', 'aggregate', 'update',
                  'predict', 'reward', 'mppi', 'prop']
# algorithmic work of one LAUNCH of each class, per particle (node) or per edge
FLOP_PER_EDGE_ENCODE = 2 * (6 * 64 + 3 * 64 * 64)
FLOP_PER_NODE = {'node_encode': 2 * (5 * 64 + 2 * 64 * 64), 'project': 2 * 2 * 64 * 64,
                 'update': 2 * 64 * 64, 'predict': 2 * (64 * 64 + 3 * 64)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=sorted(CONFIGS), default=None, help='a BASELINE.json workload (default: c2 / c3)')
    ap.add_argument('--particles', type=int, default=None)
    ap.add_argument('--samples', type=int, default=None, help='MPPI samples per GPU (weak scaling)')
    ap.add_argument('--samples-total', type=int, default=None, help='MPPI samples over all GPUs (strong scaling)')
    ap.add_argument('--horizon', type=int, default=None)
    ap.add_argument('--force-comm', action='store_true',
                    help='one process: still attach an RCCL communicator (ncclCommInitRank with one rank, the id broadcast '
                         'through torch.distributed) and run the update through ncclAllGather')
    ap.add_argument('--comm', choices=['rccl', 'gloo'], default='rccl',
                    help="transport of the one exchange: 'rccl' = ncclAllGather on the stream (the product path); 'gloo' = the "
                         "rank's record fetched, all-gathered through torch.distributed on the host and uploaded to the combine "
                         "kernel (sharding.TorchComm's path) -- with --share-gpu it lets N processes exercise the sharded "
                         "bench on ONE GPU")
    ap.add_argument('--share-gpu', action='store_true', help='every rank uses GPU 0 (with --comm gloo: a dry run of the N-rank bench on one GPU)')
    ap.add_argument('--engine', default=os.environ.get('DRP_ENGINE', 'auto'))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-samples', type=int, default=None, help='samples of the CPU baseline (default: about 10 s of host work)')
    ap.add_argument('--no-alt', action='store_true', help='skip the fp32-MFMA engine comparison run')
    ap.add_argument('--update', choices=['mppi', 'elite'], default='mppi',
                    help='the planner update that ends an iteration: softmax-weighted mean (the reference\'s optimize_action) '
                         'or the mean of the --elite best sequences; either is one small RCCL all-gather when sharded')
    ap.add_argument('--elite', type=int, default=64)
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        args.gpus = world
    cfg = args.config or ('c2' if args.gpus == 1 else 'c3')
    N, ns, H, scaling, label = CONFIGS[cfg]
    custom = any(v is not None for v in (args.particles, args.samples, args.samples_total, args.horizon))
    args.particles = args.particles or N
    args.horizon = args.horizon or H
    if args.samples_total is not None:
        scaling, ns = 'strong', args.samples_total
    elif args.samples is not None:
        scaling, ns = 'weak', args.samples
    args.scaling = scaling
    args.samples_total_job = ns if scaling == 'strong' else ns * args.gpus
    if custom:
        label = 'custom: %d particles x %d samples (%s) x %d steps' % (
            args.particles, ns, 'in total' if scaling == 'strong' else 'per GPU', args.horizon)
    args.workload = label
    if args.cpu_samples is None:
        # about 10 s of host work for the dense formulation: its cost per sample grows with N^2
        args.cpu_samples = int(min(1024, max(4, 128 * (300.0 / args.particles) ** 2 * 10.0 / args.horizon)))
    args.config_name = cfg if not custom else 'custom'
    return args


def host_cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(args, sd, s0, dens, attr, G, goal_coor, cam):
    """The oracle (dense-formulation PyTorch, the reference's algorithmic shape) on this
    box's host cores, on a bounded sample of the same workload."""
    import torch
    from oracle import propnet_dense as od
    from dyn_res_pile_manip_amd import synthetic as syn
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    W = od.load_weights(sd)
    ext = syn.demo_cam_extrinsics()
    # the dense formulation's small ops do not scale to hundreds of threads (256 threads
    # measured 20x slower than 8): pick the fastest thread count on a tiny probe
    probe = syn.sample_pushes(8, 1, seed=2)
    best, cores = None, 1
    for th in sorted(set([min(avail, t) for t in (8, 16, 32, 64)])):
        torch.set_num_threads(th)
        with torch.no_grad():
            od.rollout(W, s0, dens, attr, probe[:2], ext, 24)
            t0 = time.perf_counter()
            od.rollout(W, s0, dens, attr, probe, ext, 24)
            dtp = time.perf_counter() - t0
        if best is None or dtp < best:
            best, cores = dtp, th
    torch.set_num_threads(cores)
    ns, H, N = args.cpu_samples, args.horizon, args.particles
    acts = syn.sample_pushes(ns, H, seed=1)
    with torch.no_grad():
        od.rollout(W, s0, dens, attr, acts[:4, :1], ext, 24)          # warm-up
        t0 = time.perf_counter()
        st = od.rollout(W, s0, dens, attr, acts, ext, 24)
        r = od.config_reward_ptcl(st[:, -1], G, cam, goal_coor)
        od.optimize_action(acts, r.numpy(), 0.1)
        dt = time.perf_counter() - t0
    return {'value': ns * N * H / dt, 'unit': 'particle-steps/s', 'cores': cores, 'kind': 'port',
            'cpu_model': host_cpu_model(), 'host_threads_available': avail,
            'sample': '%d samples x %d particles x %d steps, oracle/propnet_dense.py (dense '
                      'Rr/Rs PyTorch fp32, %d of %d host threads of %s), %.1f s' % (ns, N, H, cores, avail, host_cpu_model(), dt)}


def run_gd(args):
    """--config gd-demo: the gradient-descent planner's iteration (drp_gd_step) at the demo shape."""
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    import torch.distributed as dist
    from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
    from dyn_res_pile_manip_amd.engine import Engine
    from dyn_res_pile_manip_amd.planners import world2cam_affine
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    N, H, nb = args.particles, args.horizon, 30
    rows = args.samples_total_job // world if args.scaling == 'strong' else args.samples_total_job // args.gpus
    traj = max(1, rows // nb)
    B = traj * nb
    eng = Engine(local_rank)
    sd = weights.random_state_dict(seed=0)
    eng.load_weights(weights.blob_from_state_dict(sd), 0.08)
    cam = syn.demo_cam_params()
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, cam)
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    G, goal_coor = eng.set_goal_image(obs_goal, 5 * N, fps_init=0, mode='cv5', want=True)
    s0, dens, attr = syn.make_pile(N, nb, seed=N)
    acts = np.repeat(np.stack([syn.nominal_pushes(H, seed=rank * 1000 + i) for i in range(traj)]), nb, axis=0).astype(np.float32)
    lo, hi = syn.action_limits()
    eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)

    def step():
        eng._ck(eng.lib.drp_gd_step(eng.h, None))

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    per_class = {}
    for kc in GD_CLASSES:
        eng.probe_begin(kc)
        step()
        per_class[kc] = eng.probe_read()
    dominant = max(per_class, key=lambda k: per_class[k][0])
    cnt = eng.debug_fetch('nbr_cnt', (B, N), np.uint8)
    kbar = float(cnt.mean())
    eng.probe_begin(dominant)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dom_ms, dom_n = eng.probe_read()
    eng.probe_begin(None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    per_iter = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        step()
        eng.sync()
        per_iter.append(time.perf_counter() - t1)
    med = float(np.median(per_iter))
    if rank == 0:
        avg_s = dom_ms / max(dom_n, 1) * 1e-3
        if dominant == 'prop':
            # the forward kernel of the tape-writing instantiation: the same MFMAs as bench's MPPI model
            n_cu = eng.device_info()['n_cu']
            spw = -(-B // n_cu)
            tile_max = []
            for w in range(0, B, spw):
                r = cnt[w:w + spw].ravel()
                if spw * N <= 4900 and (r == r.max()).sum() * 16 < r.size * 15:
                    r = np.sort(r)[::-1]
                r = np.pad(r, (0, (-r.size) % 32))
                tile_max.append(r.reshape(-1, 32).max(-1))
            tile_max = np.concatenate(tile_max).astype(np.float64)
            mfmas = 3 * tile_max.size * ((tile_max - 1).clip(min=0).mean() * 78 + (2 * 144 + 96) / 3.0) + tile_max.size * 204
            roof = {'bound': 'mfma', 'achieved': mfmas * 32768.0 / avg_s / 1e12, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                    'slot_iterations_per_tile': float((tile_max - 1).clip(min=0).mean()), 'tiles_per_step': int(tile_max.size)}
        elif dominant == 'bwd_node' and per_class['bwd_edge'][1] == 0:
            # kmb_step_bwd (the whole node / edge backward of a rollout step in one launch): algorithmic
            # bytes per node -- phase P: effect row + reward gradient in, g_eff / g_cnode / g_agg rows out; per
            # propagation step: own g_agg and g_eff rows, own masks, one (list entry, mask, g_agg row) per edge the
            # node feeds; steps 2 and 1 also the effect row in, g_eff / g_agg rows out and g_cnode in and out; step 0
            # g_cnode, encoder effect and impulse in, impulse gradient out
            per_node = (256 + 12 + 3 * 256) + 3 * (2 * 256 + 8 * kbar + kbar * (4 + 8 + 256)) + 2 * (256 + 2 * 256 + 2 * 256) + (2 * 256 + 24)
            work = B * N * per_node
            roof = {'bound': 'hbm', 'achieved': work / avg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'algorithmic_bytes_per_launch': work,
                    'note': 'kmb_step_bwd: fp32 MFMA for the seven 64x64 products per node, bound by the rows it moves; the gathered g_agg rows mostly hit L2'}
        elif dominant == 'bwd_edge':
            # kb_edge_terms per node and propagation step: own g_agg row + own masks read, both g_proj halves written,
            # then one (mask, g_agg row, list entry) per edge the node feeds
            work = B * N * (256 + 8 * kbar + 512 + 8 + kbar * (8 + 256 + 4))
            roof = {'bound': 'hbm', 'achieved': work / avg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
        else:
            roof = {'bound': 'hbm', 'achieved': 0.0, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
        roof.update({'frac': roof['achieved'] / roof['peak'], 'kernel': dominant, 'avg_launch_ms': avg_s * 1e3,
                     'launches': dom_n, 'traffic': None, 'traffic_source': None})
        total = world * B * N * H * args.steps
        out = {'metric': 'MPC rollout-steps/sec (samples x particles x steps/sec)', 'value': total / dt,
               'unit': 'particle-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'ms_per_step_median': med * 1e3, 'value_median': world * B * N * H / med,
               'dtype': 'f32 (forward MLP products as split fp16 / bf16 MFMA terms, backward node stages on fp32 MFMA, fp32 accumulate)',
               'data': 'synthetic',
               'config': {'workload': args.workload, 'name': args.config_name, 'n_particles': N, 'n_trajectories_per_gpu': traj,
                          'n_batch': nb, 'rows_per_gpu': B, 'n_look_ahead': H, 'engine': 'fused', 'mean_in_degree': kbar,
                          'reference_time_model_ms': None},
               'roofline': roof, 'kernel_ms_per_iteration': {k: round(v[0], 4) for k, v in per_class.items()},
               'cpu_baseline': None}
        from dyn_res_pile_manip_amd.planners import particle_num_to_iter_time
        out['config']['reference_time_model_ms'] = particle_num_to_iter_time(N)     # planners.py:25-28, batch 300 on its GPU
        if world == 1 and not args.no_cpu_baseline:
            import torch as _t
            from oracle import propnet_dense as od
            W = od.load_weights(sd)
            cpu_traj = traj            # the whole demo batch: about 5 s of host work
            a_cpu = acts[:cpu_traj * nb]
            _t.set_num_threads(min(32, os.cpu_count() or 1))
            od.gd_loss_and_grads(W, s0, dens, attr, a_cpu[:nb], G, cam, goal_coor, syn.demo_cam_extrinsics(), 24)   # warm-up
            t0 = time.perf_counter()
            od.gd_loss_and_grads(W, s0, dens, attr, a_cpu, G, cam, goal_coor, syn.demo_cam_extrinsics(), 24)
            dtc = time.perf_counter() - t0
            out['cpu_baseline'] = {'value': cpu_traj * nb * N * H / dtc, 'unit': 'particle-steps/s', 'cores': _t.get_num_threads(),
                                   'kind': 'port', 'cpu_model': host_cpu_model(),
                                   'sample': '%d trajectories x %d columns x %d particles, forward + autograd backward of '
                                             'oracle/propnet_dense.py, %.1f s' % (cpu_traj, nb, N, dtc)}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.config_name == 'gd-demo':
        return run_gd(args)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.force_comm:
        os.environ['DRP_COMM_ALWAYS'] = '1'          # read at drp_create
    import torch
    import torch.distributed as dist
    from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
    from dyn_res_pile_manip_amd.engine import Engine
    from dyn_res_pile_manip_amd.planners import world2cam_affine

    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_comm = world > 1 or args.force_comm
    gloo = args.comm == 'gloo'
    red_dev = 'cpu' if gloo else 'cuda'          # where the max-over-ranks timings are reduced
    if use_comm:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if gloo:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    from dyn_res_pile_manip_amd.sharding import shard_range
    N, H = args.particles, args.horizon
    s_lo, s_hi = shard_range(args.samples_total_job, rank, world)      # this rank's contiguous block of samples
    ns = s_hi - s_lo
    eng = Engine(local_rank)
    engine = args.engine
    if engine == 'auto':
        engine = 'fused'
    eng.set_engine(_lib.ENGINES[engine])
    sd = weights.random_state_dict(seed=0)
    eng.load_weights(weights.blob_from_state_dict(sd), 0.08)
    M34 = world2cam_affine(syn.demo_cam_extrinsics())
    cam = syn.demo_cam_params()
    eng.set_camera(M34, 24.0, cam)
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    # goal field (OpenCV's 5x5 chamfer, as the reference) and the farthest-point subsample of the goal
    # pixels, built and kept on the device (rows f3); the copies feed the CPU baseline
    G, goal_coor = eng.set_goal_image(obs_goal, 5 * N, fps_init=0, mode='cv5', want=True)
    s0, dens, attr = syn.make_pile(N, 1, seed=0)
    lo, hi = syn.action_limits()
    nominal = syn.nominal_pushes(H, seed=0)
    eng.mpc_begin(s0, attr, dens, nominal, n_sample=ns, sigma=0.3 * 24 / 12.0, beta_filter=0.7,
                  reward_weight=0.1, act_lo=lo, act_hi=hi, seed=1234, sample_offset=s_lo)
    if use_comm and not gloo:
        uid = [eng.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0], rank, world)
    if gloo:
        from dyn_res_pile_manip_amd.sharding import allgather_records

    it = [0]

    def step():
        eng.mpc_sample(it[0])
        eng.mpc_rollout(False)
        if gloo and world > 1:
            # host transport: this rank's record -> all-gather over gloo -> combine kernel
            if args.update == 'elite':
                eng.mpc_update_elite(allgather_records(eng.mpc_elite(args.elite).ravel()).reshape(world, args.elite, -1), args.elite)
            else:
                eng.mpc_update(allgather_records(eng.mpc_partials()))
        elif args.update == 'elite':
            eng.mpc_update_elite_device(args.elite)
        else:
            eng.mpc_update_device()
        it[0] += 1

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if use_comm:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # untimed calibration: one iteration per kernel class with the HIP-event probe on
    per_class = {}
    for kc in KERNEL_CLASSES:
        eng.probe_begin(kc)
        step()
        ms, n = eng.probe_read()
        per_class[kc] = (ms, n)
    dominant = max(per_class, key=lambda k: per_class[k][0])
    cnt_last = eng.debug_fetch('nbr_cnt', (ns, N), np.uint8)
    kbar = float(cnt_last.mean())
    # slot iterations the propagation kernel runs per 32-receiver tile: the largest in-degree of the tile, minus the
    # self loop when its encoder chain is replaced by the per-sample constant (attributes are zeros here).
    # km_prop3 (chip-filling batches) cuts the rows of a workgroup's samples, ordered by in-degree, into tiles;
    # km_prop cuts every sample on its own.
    self_const = engine == 'fused' and os.environ.get('DRP_NO_SELF_CONST') is None
    n_cu = eng.device_info()['n_cu']
    spw = -(-ns // n_cu)
    prop3 = (engine == 'fused' and os.environ.get('DRP_NO_PROP3') is None and ns >= n_cu and spw * ((N + 31) // 32) >= 8)
    tile_max = []
    if prop3:
        ordered = os.environ.get('DRP_NO_PROP3_ORDER') is None and spw * N <= 4900
        for w in range(0, ns, spw):
            rows = cnt_last[w:w + spw].ravel()
            if ordered and (rows == rows.max()).sum() * 16 < rows.size * 15:     # saturated piles keep the natural order
                rows = np.sort(rows)[::-1]
            rows = np.pad(rows, (0, (-rows.size) % 32))
            tile_max.append(rows.reshape(-1, 32).max(-1))
        tile_max = np.concatenate(tile_max).astype(np.float64)
    else:
        tile_max = np.pad(cnt_last, ((0, 0), (0, (-N) % 32))).reshape(ns, -1, 32).max(-1).astype(np.float64).ravel()
    n_tiles = int(tile_max.size)
    slots_per_tile = float((tile_max - (1.0 if self_const else 0.0)).clip(min=0).mean())
    eng.probe_begin(dominant)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dom_ms, dom_n = eng.probe_read()
    eng.probe_begin(None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-iteration times (each iteration synchronised; SURVEY.md 8d asks for the median): a second pass, so the
    # contract's K back-to-back iterations above stay un-synchronised
    per_iter = []
    for _ in range(args.steps):
        t1 = time.perf_counter()
        step()
        eng.sync()
        per_iter.append(time.perf_counter() - t1)
    fence()
    med = float(np.median(per_iter))
    if world > 1:
        t = torch.tensor([med], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med = float(t.item())

    # the same K steps on the pure-fp32 MFMA engine (bit-for-bit an fp32 fma chain), for reference; its un-fused
    # pipeline has the segmented sum ("scatter-add") as a kernel of its own, timed here with the HIP-event probe
    alt, scatter = None, None
    if engine != 'mfma' and not args.no_alt:
        eng.set_engine(_lib.ENGINES['mfma'])
        for _ in range(2):
            step()
        eng.probe_begin('aggregate')
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dta = time.perf_counter() - t0
        agg_ms, agg_n = eng.probe_read()
        eng.probe_begin(None)
        kbar_alt = float(eng.debug_fetch('nbr_cnt', (ns, N), np.uint8).mean())
        if agg_n > 0:
            # per receiver and propagation step, own row + K sender rows + K edge-constant rows read,
            # one row written, 256 B each
            agg_bytes = ns * N * (2 * kbar_alt + 2) * 256.0
            agg_s = agg_ms / agg_n * 1e-3
            scatter = {'kernel': 'k_aggregate (engine mfma: segmented sum over the receiver-major lists)', 'bound': 'hbm',
                       'achieved': agg_bytes / agg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                       'frac': agg_bytes / agg_s / 1e9 / PEAK_HBM_GBS, 'avg_launch_ms': agg_s * 1e3, 'launches': agg_n,
                       'algorithmic_bytes_per_launch': agg_bytes, 'mean_in_degree': kbar_alt,
                       'traffic': None, 'traffic_source': None}
        if world > 1:
            t = torch.tensor([dta], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dta = float(t.item())
        alt = {'engine': 'mfma', 'dtype': 'f32', 'value': args.samples_total_job * N * H * args.steps / dta,
               'ms_per_step': dta / args.steps * 1e3}
        eng.set_engine(_lib.ENGINES[engine])
    if rank == 0:
        total = args.samples_total_job * N * H * args.steps
        avg_s = dom_ms / max(dom_n, 1) * 1e-3
        B = ns
        tiles = n_tiles
        if dominant == 'prop':
            # km_prop: per 32-receiver tile, one 78-MFMA chain (the 3-term split
            # relation encoder) per slot iteration + the 6-term split node layers (144 MFMAs, 96 in
            # the last step); roofline on the 16-bit FLOPs actually EXECUTED, 2*32*32*16 per MFMA
            # a launch covers one propagation step (km_prop) or all three of a rollout step (km_prop3,
            # chip-filling batches): told apart by the launches the probe counted
            psteps = max(1, int(round(3.0 * H * args.steps / max(dom_n, 1))))
            mfmas = psteps * tiles * (slots_per_tile * 78 + (2 * 144 + 96) / 3.0)
            alg = psteps * B * N * (kbar * FLOP_PER_EDGE_ENCODE + 2 * 64 * 64 * (1 + 2 * 2 / 3.0 + 1 / 3.0))
            # the particle encoder runs as the first phase of km_prop3 when no launch of its own was counted:
            # 12 + 4 x 48 bf16 MFMAs per tile (first layer + four 64x64 products on the 6-term split)
            encoder_inside = psteps == 3 and per_class['node_encode'][1] == 0
            if encoder_inside:
                mfmas += tiles * 204
                alg += B * N * FLOP_PER_NODE['node_encode'] + B * N * 2 * 2 * 64 * 64
            work = mfmas * 32768.0
            roof = {'bound': 'mfma', 'achieved': work / avg_s / 1e12, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                    'mfma_dtype': 'fp16 operands for the relation encoder (fp32 values split in 2 fp16 terms), bf16 for the node layers (3 terms), fp32 accumulate',
                    'algorithmic_f32_tflops': alg / avg_s / 1e12, 'slot_iterations_per_tile': slots_per_tile,
                    'tiles_per_step': tiles, 'mean_in_degree_minus_self': kbar - (1.0 if self_const else 0.0),
                    'propagation_steps_per_launch': psteps, 'particle_encoder_in_launch': bool(encoder_inside)}
        elif dominant == 'aggregate':
            work = B * N * (2 * kbar + 2) * 256.0
            roof = {'bound': 'hbm', 'achieved': work / avg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
        elif dominant == 'edge_encode':
            work = B * N * kbar * FLOP_PER_EDGE_ENCODE
            roof = {'bound': 'mfma', 'achieved': work / avg_s / 1e12, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s'}
        elif dominant in FLOP_PER_NODE:
            work = B * N * FLOP_PER_NODE[dominant]
            roof = {'bound': 'mfma', 'achieved': work / avg_s / 1e12, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s'}
        else:
            roof = {'bound': 'hbm', 'achieved': 0.0, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
        roof['frac'] = roof['achieved'] / roof['peak']
        roof['kernel'] = dominant
        roof['avg_launch_ms'] = avg_s * 1e3
        roof['launches'] = dom_n
        roof['work_per_launch'] = work if dominant in ('prop', 'aggregate', 'edge_encode') or dominant in FLOP_PER_NODE else None
        # HBM-side bytes per launch of this kernel from rocprofv3 PMC passes (FETCH_SIZE x2 +
        # WRITE_SIZE, profiles/summarize_pmc.py); collected on this same workload
        roof['traffic'] = None
        roof['traffic_source'] = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and (N, ns) == (300, 1024):
            try:
                tj = json.load(open(tpath))
                tkey = 'prop3' if (dominant == 'prop' and roof.get('propagation_steps_per_launch') == 3) else dominant
                roof['traffic'] = tj[engine][tkey]['hbm_bytes_per_launch']
                roof['traffic_source'] = 'profiles/traffic.json (builder-side rocprofv3 --pmc passes of this command, not measured in this run)'
                if scatter is not None:
                    scatter['traffic'] = tj['mfma']['aggregate']['hbm_bytes_per_launch']
                    scatter['traffic_source'] = roof['traffic_source']
            except Exception:
                pass
        out = {
            'metric': 'MPC rollout-steps/sec (samples x particles x steps/sec)',
            'value': total / dt, 'unit': 'particle-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': args.scaling, 'vs_baseline': None,
            'ms_per_step_median': med * 1e3, 'value_median': args.samples_total_job * N * H / med,
            'dtype': 'f32' if engine in ('valu', 'mfma') else 'f32 (MLP products as split fp16 / bf16 MFMA terms, fp32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': args.workload, 'name': args.config_name,
                       'n_particles': N, 'n_sample_per_gpu': ns, 'n_sample_total': args.samples_total_job,
                       'n_look_ahead': H, 'engine': engine,
                       'communicator': ('%s, %d rank%s%s' % (args.comm, world, '' if world == 1 else 's',
                                                             ', all on GPU 0' if args.share_gpu else '')) if use_comm else None,
                       'mean_in_degree': kbar, 'parallelism': 'samples sharded x%d' % world,
                       'update': 'softmax mean (optimize_action)' if args.update == 'mppi' else 'mean of the %d best (elite)' % args.elite},
            'roofline': roof,
            'kernel_ms_per_iteration': {k: round(v[0], 4) for k, v in per_class.items()},
        }
        out['roofline_scatter'] = scatter
        out['alt_engine'] = alt
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, sd, s0, dens, attr, G, goal_coor, cam)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    eng.close()
    if use_comm:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
