#!/usr/bin/env python3
"""Benchmark of the hot path: one "step" = one sampling-MPC iteration on the device
  sample pushes (MPPI) -> H-step GNN rollout (graph rebuild + impulse + PropNet each step)
  -> final-step reward -> softmax-weighted update [-> ONE RCCL all-gather when sharded].

Metric (BASELINE.json): particle-steps/s = samples x particles x steps / wall time.
Workload at N=1: BASELINE.json configs[1] -- 300-particle pile, 1024 MPPI samples,
10-step horizon (inputs resident in HBM).  With --gpus N the sample axis is sharded,
1024 samples per GPU (weak scaling, configs[2] at N=8).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|p20|c4-50|c4-150|c4-300|c4-600|c5|c5-share|gd-demo]

`python bench.py --gpus N` with N > 1 and no launcher around it starts its own N rank processes (one per GPU) as
CHILD processes -- before anything in this process has touched the GPU -- relays rank 0's JSON line and exits with
the children's status; under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (how the
driver launches it) the ranks come from the environment.  WORLD_SIZE and --gpus must agree, and with the RCCL
transport the communicator's own ncclCommCount must equal both, or the run exits non-zero.

The ranks rendezvous over a gloo process group (the ncclUniqueId broadcast, the barrier and the max-over-ranks of the
timings); torch's NCCL process group is never created, so ONE RCCL serves the process: the one libdrp.so binds at
run time (config.rccl in the line names its version and file).

The default one-GPU run also carries a `sweep` block: the other BASELINE workloads (20 particles -- the small-pile
regime --, configs[3] 50 / 150 / 600 particles, the per-GPU share of configs[4], the reference's live GD planner
shape), 20 iterations each after 5 warm-ups.  --config selects one of them as the headline instead; c5 is the strong-scaling one (4096 samples in
total).  --force-comm attaches a one-rank RCCL communicator and runs the update through ncclAllGather.
"""
import argparse
import json
import os
import subprocess
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
    'p20': (20, 1024, 10, 'weak', "small piles (the reference re-samples the pile per MPC step at the particle count its resolution regressor asks for, env/flex_env.py:997-1020; its time model planners.py:25-28 is fitted over 10 - 100): 20 particles x 1024 samples x 10 steps"),
    'c4-50': (50, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 50 particles x 1024 samples x 10 steps'),
    'c4-150': (150, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 150 particles x 1024 samples x 10 steps'),
    'c4-300': (300, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 300 particles x 1024 samples x 10 steps'),
    'c4-600': (600, 1024, 10, 'weak', 'BASELINE configs[3]: dynamic-resolution sweep, 600 particles x 1024 samples x 10 steps'),
    'c5': (1200, 4096, 20, 'strong', 'BASELINE configs[4]: 1200-particle dense pile, 4096 samples in total, 20-step horizon'),
    'c5-share': (1200, 512, 20, 'weak', "BASELINE configs[4], one GPU's share at 8 GPUs: 1200 particles x 512 samples x 20 steps"),
    # the reference's LIVE planner (mpc_type 'GD', config/mpc/config.yaml:38-43): 50 trajectories x 30 particle
    # re-samplings = 1500 independent Adam problems, horizon 1; a "step" = one iteration of planners.py:682-764
    # (rollout, final-step reward, reverse mode, Adam, clip box).  samples = trajectories x 30 here.
    'gd-demo': (100, 1500, 1, 'weak', "the reference's live GD planner at its demo shape: 50 trajectories x 30 re-samplings x 100 particles, horizon 1 "
                                      '(config/mpc/config.yaml:38-43), one Adam iteration per step'),
}
SWEEP = ['p20', 'c4-50', 'c4-150', 'c4-600', 'c5-share', 'gd-demo']     # the default run's extra block (configs[1] is the headline)
GD_CLASSES = ['graph', 'node_encode', 'prop', 'reward', 'tape_copy', 'bwd_reward', 'bwd_lists', 'bwd_node', 'bwd_edge',
              'bwd_push', 'opt']
KERNEL_CLASSES = ['graph', 'node_encode', 'edge_encode', 'project', 'aggregate', 'update',
                  'predict', 'reward', 'mppi', 'prop']
# algorithmic work of one LAUNCH of each class, per particle (node) or per edge
FLOP_PER_EDGE_ENCODE = 2 * (6 * 64 + 3 * 64 * 64)
FLOP_PER_NODE = {'node_encode': 2 * (5 * 64 + 2 * 64 * 64), 'project': 2 * 2 * 64 * 64,
                 'update': 2 * 64 * 64, 'predict': 2 * (64 * 64 + 3 * 64)}
EXIT_WORLD_MISMATCH = 4      # WORLD_SIZE / --gpus / ncclCommCount disagree
EXIT_TIMEOUT = 124           # the self-launched ranks did not finish in --timeout seconds


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
                    help='one process: still attach an RCCL communicator (ncclCommInitRank with one rank) and run the '
                         'update through ncclAllGather')
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
    ap.add_argument('--no-sweep', action='store_true', help='skip the sweep block of the default one-GPU run')
    ap.add_argument('--sweep', action='store_true', help='add the sweep block to a run that would not carry it')
    ap.add_argument('--update', choices=['mppi', 'elite'], default='mppi',
                    help='the planner update that ends an iteration: softmax-weighted mean (the reference\'s optimize_action) '
                         'or the mean of the --elite best sequences; either is one small RCCL all-gather when sharded')
    ap.add_argument('--elite', type=int, default=64)
    ap.add_argument('--timeout', type=float, default=1500.0,
                    help='self-launched ranks (--gpus N without a launcher): seconds before the parent ends them and exits %d' % EXIT_TIMEOUT)
    ap.add_argument('--fault-rank', type=int, default=None, help=argparse.SUPPRESS)   # tests: this rank exits(7) after the warm-up
    args = ap.parse_args()
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
    args.do_sweep = args.sweep or (args.config is None and not custom and args.gpus == 1 and not args.no_sweep
                                   and args.engine in ('auto', 'fused') and not args.force_comm)
    return args


# ---- N > 1 without a launcher: this process becomes the parent of N rank processes ---------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpu_count():
    """GPUs this process could use, WITHOUT touching HIP (the parent of the ranks must not initialise the GPU): the DRM
    render nodes it can open -- a container is handed the nodes of its GPUs only -- cut down by HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES.  0 without the compute driver's device node, None when the box does not say."""
    import glob
    if not os.path.exists('/dev/kfd'):
        return 0
    nodes = glob.glob('/dev/dri/renderD*')
    if not nodes:
        return None
    n = 0
    for f in nodes:
        try:
            os.close(os.open(f, os.O_RDWR))
            n += 1
        except OSError:
            pass
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def preflight(args, out=sys.stderr):
    """--gpus N on a box with fewer GPUs must fail NOW (exit 4, one clear line), not after the ranks have waited out a
    rendezvous: checked by the parent of self-launched ranks and by every rank a launcher started.  --share-gpu (all ranks
    on GPU 0: the one-GPU dry run) lifts it."""
    if args.gpus <= 1 or args.share_gpu:
        return 0
    have = visible_gpu_count()
    if have is not None and have < args.gpus:
        out.write('bench.py: --gpus %d but this box shows %d GPU(s) (render nodes / *_VISIBLE_DEVICES): not starting; '
                  '--share-gpu --comm gloo runs the N-rank code path on one GPU\n' % (args.gpus, have))
        return EXIT_WORLD_MISMATCH
    return 0


def launch_ranks(args):
    """Start one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, as
    torch.distributed.run would set them), relay rank 0's JSON line, return the exit status.  The parent imports
    neither torch.cuda nor the engine and makes no HIP call: nothing here may initialise the GPU before the children
    exist, and no process image is ever replaced.  A rank that fails, or --timeout, ends every child (each is its own
    process group, killed by its pgid -- never by pattern)."""
    import signal
    import tempfile
    rc = preflight(args)
    if rc:
        return rc
    n = args.gpus
    port = _free_port()
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        out = tempfile.TemporaryFile() if r == 0 else subprocess.DEVNULL
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=os.getcwd(),
                                      stdout=out, start_new_session=True))

    def end_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()

    deadline = time.time() + args.timeout
    status = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                sys.stderr.write('bench.py: rank %d exited with status %d; ending the other ranks\n' % bad[0])
                status = bad[0][1] if bad[0][1] > 0 else 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                sys.stderr.write('bench.py: %d ranks did not finish in %.0f s; ending them\n' % (n, args.timeout))
                status = EXIT_TIMEOUT
                break
            time.sleep(0.05)
    finally:
        end_all()
    outs[0].seek(0)
    text = outs[0].read().decode(errors='replace')
    lines = [l for l in text.splitlines() if l.startswith('{')]
    if status == 0:
        if len(lines) != 1:
            sys.stderr.write('bench.py: expected ONE json line from rank 0, got %d\n' % len(lines))
            status = 1
        else:
            try:
                got = json.loads(lines[0]).get('n_gpus')
            except ValueError:
                got = None
            if got != n:
                sys.stderr.write('bench.py: rank 0 reports n_gpus=%r, --gpus %d\n' % (got, n))
                status = EXIT_WORLD_MISMATCH
    sys.stdout.write(text)
    sys.stdout.flush()
    return status


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
    # the dense formulation's small ops do not scale to hundreds of threads (256 threads measured 20x slower than 8),
    # and which of 32 / 64 wins differs from box to box by less than a probe of a few samples can tell: the WHOLE sample
    # leg is timed at both (at min(avail, .) threads) and the faster one is the baseline, both times in the line
    ns, H, N = args.cpu_samples, args.horizon, args.particles
    acts = syn.sample_pushes(ns, H, seed=1)
    legs, leg_samples = {}, {}
    for th in sorted(set(min(avail, t) for t in (32, 64, 128))):
        # the 128-thread leg (all physical cores of a 2 x 64-core host, what BASELINE.md asks for) runs a quarter of the
        # sample and is scaled: it is the slow one, and the default run has to end within minutes
        a = acts if th <= 64 else acts[:max(8, ns // 4)]
        torch.set_num_threads(th)
        with torch.no_grad():
            od.rollout(W, s0, dens, attr, acts[:4, :1], ext, 24)          # warm-up
            t0 = time.perf_counter()
            st = od.rollout(W, s0, dens, attr, a, ext, 24)
            r = od.config_reward_ptcl(st[:, -1], G, cam, goal_coor)
            od.optimize_action(a, r.numpy(), 0.1)
            legs[th] = (time.perf_counter() - t0) * ns / a.shape[0]
            leg_samples[th] = int(a.shape[0])
    cores = min(legs, key=legs.get)
    dt = legs[cores]
    return {'value': ns * N * H / dt, 'unit': 'particle-steps/s', 'cores': cores, 'kind': 'port',
            'seconds_by_threads': {str(k): round(v, 3) for k, v in legs.items()},
            'value_by_threads': {str(k): round(ns * N * H / v, 1) for k, v in legs.items()},
            'samples_by_threads': {str(k): v for k, v in leg_samples.items()},
            'cpu_model': host_cpu_model(), 'host_threads_available': avail,
            'cores_policy': 'the fastest of the 32-, 64- and 128-thread legs (128 = all physical cores of a 2 x 64-core host, BASELINE.md 3; '
                            'that leg runs a quarter of the sample, its seconds scaled to the whole): the dense formulation\'s small ops do not '
                            'scale with threads, every leg is in seconds_by_threads / value_by_threads',
            'sample': '%d samples x %d particles x %d steps, oracle/propnet_dense.py (dense '
                      'Rr/Rs PyTorch fp32, %d of %d host threads of %s; the same leg at %s threads: %s s), %.1f s' % (
                          ns, N, H, cores, avail, host_cpu_model(), ' / '.join(str(k) for k in legs),
                          ' / '.join('%.1f' % v for v in legs.values()), dt)}


def load_traffic():
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    except Exception:
        return {}


TRAFFIC_SOURCE = 'profiles/traffic.json (builder-side rocprofv3 --pmc passes of this command, not measured in this run)'


MFMA_CYCLES_16BIT = 32       # v_mfma_f32_32x32x16_{f16,bf16}: 8 passes of 4 cycles
NOMINAL_CLOCK_HZ = 2.4e9     # MI355X_MICROARCH.md peak engine clock (the kernels hold 2.03 - 2.09 GHz: DESIGN_NOTES 5b)


ROOF_FIRST = ('frac', 'kernel', 'avg_launch_ms', 'launches', 'traffic', 'frac_executed_16bit', 'mfma_pipe_busy_est',
              'hbm_algorithmic_frac', 'sclk_mhz_under_load', 'frac_at_measured_clock', 'bound', 'achieved', 'peak', 'unit',
              'frac_basis', 'algorithmic_frac', 'executed_over_algorithmic', 'executed_tflops', 'hbm_algorithmic_gbs',
              'mfma_pipe_busy_at_measured_clock', 'work_per_launch', 'work_per_launch_flop', 'cache_served')


def ordered(d, first=ROOF_FIRST):
    """The record a reader (or a driver that keeps the first N keys of a block) needs first: the fraction, the kernel, its
    launch time, the counter traffic, the executed-work pair and the clock -- then the rest, explanatory strings last."""
    if not isinstance(d, dict):
        return d
    head = [k for k in first if k in d]
    tail = [k for k in d if k not in head]
    tail.sort(key=lambda k: (isinstance(d[k], (str, dict)) or d[k] is None))
    return {k: d[k] for k in head + tail}


def prop_roofline(work, kbar, self_const, B, N, avg_s, H, n_cu=256):
    """km_prop / km_prop3 / km_rollout, priced as SURVEY.md 8(d) prescribes:
      achieved = ALGORITHMIC FLOPs of the formulation the kernels execute -- the factored one, F_fac(K) = 116 096 + 25 472 K
                 per particle-step at the measured mean in-degree K -- times the particle-steps one launch processes, over the
                 launch's average duration (HIP events on the context's stream);
      peak     = the dense fp32 matrix peak 8(d) names, 157.3 TFLOP/s (`peak_basis`).
    `frac` = achieved / peak rises when redundant work is removed.  Beside it, what the matrix pipe is actually given:
    `frac_executed_16bit` (the 16-bit MFMAs the kernels COUNTED themselves over the iteration before the timed ones --
    Engine.probe_work: every fp32 product runs as 3 (fp16 pair) or 6 (bf16 triple) MFMA terms, the chain re-runs per
    propagation step where it is not cached -- x 32 768 FLOP against the 2.5 PFLOP/s 16-bit peak), `mfma_pipe_busy_est`
    (those MFMAs x 32 cycles over the chip's SIMDs at the nominal clock), and `hbm_algorithmic_frac`, 8(d)'s OTHER roofline:
    the gather / segmented sum's algorithmic bytes (2K + 2 rows of 256 B per receiver and propagation step) over the same
    time against 8 TB/s -- bytes the fused kernel mostly never moves (`traffic` is what the counters saw).
    A launch covers one propagation step (km_prop), the three of a rollout step (km_prop3) or a whole rollout."""
    n = float(max(work['launches'], 1))
    flops = work['mfmas'] / n * 32768.0
    particle_steps = B * N * H / n
    useful = particle_steps * (116096.0 + 25472.0 * kbar)
    k_run = kbar - (1.0 if self_const else 0.0)                   # chains an average receiver runs: the self loop's is a constant
    slots = work['chain_slots'] + work['cached_slots']
    node_parts = work['tiles'] + work['tiles_last']
    gather_bytes = particle_steps * 3.0 * (2.0 * kbar + 2.0) * 256.0
    # the shader clock the counted launches ran at: s_memtime over s_memrealtime (100 MHz) between entry and exit, summed over
    # their workgroups (drp_probe_work [6] / [7]); the peaks above are quoted at the 2.4 GHz engine clock
    sclk = 100.0 * work['clk_cycles'] / work['clk_ticks'] if work.get('clk_ticks') else None
    at_clk = (NOMINAL_CLOCK_HZ / 1e6 / sclk) if sclk else None
    return {'sclk_mhz_under_load': sclk,
            'frac_at_measured_clock': (useful / avg_s / 1e12 / PEAK_F32_TFLOPS * at_clk) if sclk else None,
            'mfma_pipe_busy_at_measured_clock': (work['mfmas'] / n * MFMA_CYCLES_16BIT / (4.0 * n_cu) / (avg_s * sclk * 1e6)) if sclk else None,
            'sclk_basis': 'sum of s_memtime deltas / sum of s_memrealtime deltas (100 MHz) over the workgroups of the counting launches of '
                          'this kernel, in the iteration before the timed ones; frac_at_measured_clock = frac x 2400 / sclk',
            'bound': 'mfma', 'achieved': useful / avg_s / 1e12, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s',
            'peak_basis': 'SURVEY.md 8d: algorithmic FLOPs of the factored formulation, F_fac(K) = 116096 + 25472 K per particle-step '
                          '(K = %.2f, %.0f particle-steps per launch), against the dense fp32 matrix peak 8d names (157.3 TFLOP/s)' % (kbar, particle_steps),
            'work_per_launch_flop': useful,
            'frac_executed_16bit': flops / avg_s / 1e12 / PEAK_BF16_TFLOPS, 'executed_tflops': flops / avg_s / 1e12,
            'executed_over_algorithmic': flops / useful,
            'mfma_pipe_busy_est': work['mfmas'] / n * MFMA_CYCLES_16BIT / (4.0 * n_cu) / (avg_s * NOMINAL_CLOCK_HZ),
            'mfma_pipe_busy_basis': 'counted 16-bit MFMAs x 32 cycles / (4 SIMDs x %d CUs) / (launch time x 2.4 GHz nominal): a lower bound, '
                                    'the kernels hold about 2.05 GHz; the SQ_VALU_MFMA_BUSY_CYCLES pass is in profiles/' % n_cu,
            'hbm_algorithmic_frac': gather_bytes / avg_s / 1e9 / PEAK_HBM_GBS, 'hbm_algorithmic_gbs': gather_bytes / avg_s / 1e9,
            'hbm_algorithmic_basis': 'SURVEY.md 8d: (2K + 2) x 256 B per receiver and propagation step, x 3 steps x the particle-steps of a launch, against 8 TB/s',
            'mfma_dtype': 'fp16 operands for the relation encoder (fp32 values split in 2 fp16 terms), bf16 for the node layers (3 terms), fp32 accumulate',
            'numerator_executed': 'drp_probe_work: counted by the kernels over the iteration before the timed ones',
            'executed_per_launch': {k: v / n for k, v in work.items() if k not in ('launches', 'clk_cycles', 'clk_ticks')},
            'algorithmic_f32_tflops_without_self_loop': particle_steps * (116096.0 + 25472.0 * k_run) / avg_s / 1e12,
            'slot_iterations_per_tile': slots / float(max(node_parts, 1)),
            'cached_share_of_slot_iterations': work['cached_slots'] / float(max(slots, 1)),
            'tiles_per_step': node_parts / float(3 * H),
            'mean_in_degree_minus_self': k_run,
            'propagation_steps_per_launch': 3.0 * H / n,
            'particle_encoder_in_launch': bool(work['encoder_tiles'] > 0),
            'graph_build_in_launch': bool(H / n > 1.5)}, useful


class Rig(object):
    """One rank's engine with the model constants installed; workloads are begun on it one after the other."""

    def __init__(self, local_rank, engine):
        from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib
        from dyn_res_pile_manip_amd.engine import Engine
        from dyn_res_pile_manip_amd.planners import world2cam_affine
        self.syn, self._lib = syn, _lib
        self.eng = Engine(local_rank)
        self.engine = 'fused' if engine == 'auto' else engine
        self.eng.set_engine(_lib.ENGINES[self.engine])
        self.sd = weights.random_state_dict(seed=0)
        self.eng.load_weights(weights.blob_from_state_dict(self.sd), 0.08)
        self.cam = syn.demo_cam_params()
        self.eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, self.cam)
        self.obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
        self.n_cu = self.eng.device_info()['n_cu']
        self.goal_n = None

    def goal(self, N):
        # goal field (OpenCV's 5x5 chamfer, as the reference) and the farthest-point subsample of the goal pixels to
        # 5N points, built and kept on the device (rows f3); the copies feed the CPU baseline
        self.G, self.goal_coor = self.eng.set_goal_image(self.obs_goal, 5 * N, fps_init=0, mode='cv5', want=True)
        self.goal_n = N


def bench_mppi(rig, N, ns, H, s_lo, steps, warmup, step_extra, fence, classes, want_median, fault=False):
    """W warm-ups, one calibration iteration per kernel class (HIP-event probe), then EXACTLY `steps` timed iterations
    bracketed by fence().  step_extra(eng): the update that ends an iteration.  -> dict of raw measurements."""
    eng, syn = rig.eng, rig.syn
    rig.goal(N)
    s0, dens, attr = syn.make_pile(N, 1, seed=0)
    lo, hi = syn.action_limits()
    nominal = syn.nominal_pushes(H, seed=0)
    eng.mpc_begin(s0, attr, dens, nominal, n_sample=ns, sigma=0.3 * 24 / 12.0, beta_filter=0.7,
                  reward_weight=0.1, act_lo=lo, act_hi=hi, seed=1234, sample_offset=s_lo)
    it = [0]

    def step():
        eng.mpc_sample(it[0])
        eng.mpc_rollout(False)
        step_extra(eng)
        it[0] += 1

    for _ in range(warmup):
        step()
    if fault:
        eng.sync()
        os._exit(7)                     # a rank that is simply gone: no clean-up, no goodbye to the group
    fence()
    per_class = {}
    dom_work = None
    for kc in classes:
        # one iteration per kernel class; the propagation kernels also COUNT what they execute in theirs (the counters
        # cost time: never inside the timed region)
        eng.probe_begin('prop+work' if kc == 'prop' else kc)
        step()
        per_class[kc] = eng.probe_read()
        if kc == 'prop' and per_class[kc][1] > 0:
            dom_work = eng.probe_work()
            dom_work['launches'] = per_class[kc][1]
    dominant = max(per_class, key=lambda k: per_class[k][0])
    cnt_last = eng.debug_fetch('nbr_cnt', (ns, N), np.uint8)
    eng.probe_begin(dominant)
    eng.dispatch_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dom_ms, dom_n = eng.probe_read()
    eng.probe_begin(None)
    ran = eng.last_dispatch()            # the kernel variants the timed iterations launched (drp_last_dispatch)
    med = None
    if want_median:
        # per-iteration times (each iteration synchronised; SURVEY.md 8d asks for the median): a second pass, so the
        # contract's K back-to-back iterations above stay un-synchronised
        per_iter = []
        for _ in range(steps):
            t1 = time.perf_counter()
            step()
            eng.sync()
            per_iter.append(time.perf_counter() - t1)
        fence()
        med = float(np.median(per_iter))
    return {'dt': dt, 'median': med, 'per_class': per_class, 'dominant': dominant, 'dom_ms': dom_ms, 'dom_n': dom_n, 'ran': ran,
            'cnt': cnt_last, 'kbar': float(cnt_last.mean()), 'step': step, 's0': s0, 'dens': dens, 'attr': attr, 'work': dom_work}


def mppi_roofline(rig, m, N, ns, H, steps):
    """Roofline of the dominant kernel of an MPPI workload measured by bench_mppi."""
    engine, dominant = rig.engine, m['dominant']
    avg_s = m['dom_ms'] / max(m['dom_n'], 1) * 1e-3
    kbar = m['kbar']
    self_const = engine == 'fused' and os.environ.get('DRP_NO_SELF_CONST') is None
    work = None
    if dominant == 'prop':
        roof, work = prop_roofline(m['work'], kbar, self_const, ns, N, avg_s, H, rig.n_cu)
    elif dominant == 'aggregate':
        work = ns * N * (2 * kbar + 2) * 256.0
        roof = {'bound': 'hbm', 'achieved': work / avg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
    elif dominant == 'edge_encode':
        work = ns * N * kbar * FLOP_PER_EDGE_ENCODE
        roof = {'bound': 'mfma', 'achieved': work / avg_s / 1e12, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s'}
    elif dominant in FLOP_PER_NODE:
        work = ns * N * FLOP_PER_NODE[dominant]
        roof = {'bound': 'mfma', 'achieved': work / avg_s / 1e12, 'peak': PEAK_F32_TFLOPS, 'unit': 'TFLOP/s'}
    else:
        roof = {'bound': 'hbm', 'achieved': 0.0, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
    roof['frac'] = roof['achieved'] / roof['peak']
    # the class the HIP-event probe timed, and the variant(s) of it the timed iterations launched
    names = [v for v in m.get('ran', []) if v.startswith('km_')] if dominant == 'prop' else []
    roof['kernel'] = '%s: %s' % (dominant, ', '.join(names)) if names else dominant
    roof['avg_launch_ms'] = avg_s * 1e3
    roof['launches'] = m['dom_n']
    roof['work_per_launch'] = work
    return ordered(roof)


def scatter_roofline(ns, N, kbar, agg_ms, agg_n, traffic_bytes):
    """The segmented sum of the un-fused pipeline (k_aggregate_lds / k_aggregate, engine mfma).  SURVEY.md 8d's
    ALGORITHMIC bytes per receiver and propagation step -- own row + K sender rows + K edge-constant rows read, one
    row written, 256 B each -- are reported as such (`algorithmic_*`); they are not all HBM bytes: the kernel stages a
    sample's sender rows in LDS once, so (K - 1) / K of the sender-row reads never leave the CU.  `frac` is therefore
    computed from HBM-side bytes only: the PMC traffic of profiles/ when this is the profiled shape, otherwise the
    COMPULSORY bytes (every edge-constant row, every projection row and every output row once) -- never a fraction
    above 1 of the HBM peak."""
    agg_s = agg_ms / agg_n * 1e-3
    alg = ns * N * (2 * kbar + 2) * 256.0
    compulsory = ns * N * (kbar * 256.0 + 512.0 + 256.0)
    hbm, basis = (traffic_bytes, 'pmc traffic') if traffic_bytes else (compulsory, 'compulsory bytes (each row once)')
    return ordered({'kernel': 'k_aggregate (engine mfma: segmented sum over the receiver-major lists)', 'bound': 'hbm',
            'achieved': hbm / agg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': hbm / agg_s / 1e9 / PEAK_HBM_GBS,
            'frac_basis': basis, 'avg_launch_ms': agg_s * 1e3, 'launches': agg_n,
            'algorithmic_bytes_per_launch': alg, 'algorithmic_gbs': alg / agg_s / 1e9,
            'compulsory_bytes_per_launch': compulsory, 'cache_served': bool(alg > hbm), 'mean_in_degree': kbar,
            'traffic': traffic_bytes, 'traffic_source': TRAFFIC_SOURCE if traffic_bytes else None})


def bench_gd(rig, N, traj, nb, H, steps, warmup, fence, rank=0, want_median=True):
    """The gradient-descent planner's iteration (drp_gd_step) at traj x nb rows of N particles."""
    eng, syn = rig.eng, rig.syn
    rig.goal(N)
    B = traj * nb
    s0, dens, attr = syn.make_pile(N, nb, seed=N)
    acts = np.repeat(np.stack([syn.nominal_pushes(H, seed=rank * 1000 + i) for i in range(traj)]), nb, axis=0).astype(np.float32)
    lo, hi = syn.action_limits()
    eng.gd_begin(s0, attr, dens, acts, 0.05, lo, hi)

    def step():
        eng._ck(eng.lib.drp_gd_step(eng.h, None))

    for _ in range(warmup):
        step()
    fence()
    per_class = {}
    dom_work = None
    for kc in GD_CLASSES:
        eng.probe_begin('prop+work' if kc == 'prop' else kc)
        step()
        per_class[kc] = eng.probe_read()
        if kc == 'prop' and per_class[kc][1] > 0:
            dom_work = eng.probe_work()
            dom_work['launches'] = per_class[kc][1]
    dominant = max(per_class, key=lambda k: per_class[k][0])
    cnt = eng.debug_fetch('nbr_cnt', (B, N), np.uint8)
    kbar = float(cnt.mean())
    eng.probe_begin(dominant)
    eng.dispatch_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    dom_ms, dom_n = eng.probe_read()
    eng.probe_begin(None)
    ran = eng.last_dispatch()
    med = None
    if want_median:
        per_iter = []
        for _ in range(steps):
            t1 = time.perf_counter()
            step()
            eng.sync()
            per_iter.append(time.perf_counter() - t1)
        med = float(np.median(per_iter))
    avg_s = dom_ms / max(dom_n, 1) * 1e-3
    traffic = load_traffic().get('gd-demo', {}) if (N, B, H) == (100, 1500, 1) else {}
    tkey = None
    if dominant == 'prop':
        # the forward kernel of the tape-writing instantiation: the same MFMAs as the MPPI model
        roof, _ = prop_roofline(dom_work, kbar, True, B, N, avg_s, H, rig.n_cu)
        tkey = 'prop3_tape'
    elif dominant == 'bwd_node' and per_class['bwd_edge'][1] == 0:
        # kmb_rows_bwd / kmb_step_bwd (the whole node / edge backward of a rollout step in one launch, DESIGN.md 8 and DESIGN_NOTES.md 7b): algorithmic
        # bytes per node -- phase P: effect row + reward gradient in, g_eff / g_cnode / g_agg rows out; per
        # propagation step: own g_agg and g_eff rows, own masks, one (list entry, mask, g_agg row) per edge the
        # node feeds; steps 2 and 1 also the effect row in, g_eff / g_agg rows out and g_cnode in and out; step 0
        # g_cnode, encoder effect and impulse in, impulse gradient out
        per_node = (256 + 12 + 3 * 256) + 3 * (2 * 256 + 8 * kbar + kbar * (4 + 8 + 256)) + 2 * (256 + 2 * 256 + 2 * 256) + (2 * 256 + 24)
        work = B * N * per_node
        roof = {'bound': 'hbm', 'achieved': work / avg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'algorithmic_bytes_per_launch': work,
                'note': 'the GD backward in one launch (kmb_rows_bwd up to 256 particles: rows in registers, gathers from LDS; kmb_step_bwd above); '
                        'the byte model counts every row of the launch-per-phase formulation -- traffic (when present) is the HBM side'}
        tkey = 'step_bwd'
    elif dominant == 'bwd_edge':
        # kb_edge_terms per node and propagation step: own g_agg row + own masks read, both g_proj halves written,
        # then one (mask, g_agg row, list entry) per edge the node feeds
        work = B * N * (256 + 8 * kbar + 512 + 8 + kbar * (8 + 256 + 4))
        roof = {'bound': 'hbm', 'achieved': work / avg_s / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
    else:
        roof = {'bound': 'hbm', 'achieved': 0.0, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
    tb = traffic.get(tkey, {}).get('hbm_bytes_per_launch') if tkey else None
    names = [v for v in ran if v.startswith('km_')] if dominant == 'prop' else [v for v in ran if v.startswith('bwd:')]
    roof.update({'frac': roof['achieved'] / roof['peak'], 'kernel': '%s: %s' % (dominant, ', '.join(names)) if names else dominant,
                 'avg_launch_ms': avg_s * 1e3,
                 'launches': dom_n, 'traffic': tb, 'traffic_source': TRAFFIC_SOURCE if tb else None})
    if tb and roof['bound'] == 'hbm':
        # the HBM fraction proper: counter bytes over time (the algorithmic figure above counts L2-served gathers)
        roof['algorithmic_frac'] = roof['frac']
        roof['achieved'] = tb / avg_s / 1e9
        roof['frac'] = roof['achieved'] / roof['peak']
        roof['cache_served'] = bool(roof.get('algorithmic_bytes_per_launch', 0) > tb)
    # the backward launch (kmb_rows_bwd up to 256 particles, kmb_step_bwd above) beside the forward's, whichever dominates:
    # HBM side from the counters on file, matrix side from the algorithmic FLOPs of reverse mode WITHOUT weight gradients at
    # horizon 1 (no gradient into the relation encoder: its inputs are the given state): predictor 2 x (192 + 4096), per propagation
    # step W_agg^T, W_r^T, W_s^T (3 x 4096 MAC) and 128 adds per edge, node constant and particle encoder 4096 + 4096 + 320 MAC
    roof_bwd = None
    if per_class.get('bwd_node', (0, 0))[1] > 0 and per_class['bwd_edge'][1] == 0 and H == 1:
        b_s = per_class['bwd_node'][0] / per_class['bwd_node'][1] * 1e-3
        per_node_b = (256 + 12 + 3 * 256) + 3 * (2 * 256 + 8 * kbar + kbar * (4 + 8 + 256)) + 2 * (256 + 2 * 256 + 2 * 256) + (2 * 256 + 24)
        flop_b = B * N * (99328.0 + 384.0 * kbar)
        tbb = traffic.get('step_bwd', {}).get('hbm_bytes_per_launch')
        roof_bwd = ordered({'kernel': 'bwd_node (kmb_rows_bwd / kmb_step_bwd: the whole reverse pass of a rollout step in one launch)',
                            'avg_launch_ms': b_s * 1e3, 'launches': per_class['bwd_node'][1], 'traffic': tbb,
                            'bound': 'neither roof: a chain of gathers from LDS / L2 between fp32 matrix layers',
                            'frac': (tbb / b_s / 1e9 / PEAK_HBM_GBS) if tbb else None, 'frac_basis': 'counter bytes over time against 8 TB/s',
                            'achieved': (tbb / b_s / 1e9) if tbb else None, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                            'algorithmic_bytes_per_launch': B * N * per_node_b, 'algorithmic_gbs': B * N * per_node_b / b_s / 1e9,
                            'mfma_f32_frac_algorithmic': flop_b / b_s / 1e12 / PEAK_F32_TFLOPS, 'algorithmic_tflops': flop_b / b_s / 1e12,
                            'flop_model': 'F_bwd(K) = 99328 + 384 K per particle (K = %.2f): reverse mode without weight gradients, horizon 1' % kbar})
    if dom_work and dom_work.get('clk_ticks') and 'sclk_mhz_under_load' not in roof:
        # the clock of the forward kernel's counting launch of the same iteration stands for the workload's
        roof['sclk_mhz_under_load'] = 100.0 * dom_work['clk_cycles'] / dom_work['clk_ticks']
    return {'dt': dt, 'median': med, 'per_class': per_class, 'dominant': dominant, 'roofline': ordered(roof), 'roofline_backward': roof_bwd, 'kbar': kbar,
            'B': B, 's0': s0, 'dens': dens, 'attr': attr, 'acts': acts, 'step': step}


SWEEP_MIN_GPU_S = 1.0        # every sweep entry keeps the GPU busy at least this long (batches of 20 iterations)


def more_batches(step, fence, steps, dt_first, min_gpu_s):
    """Batches of `steps` iterations, each between two fences, until the entry has kept the GPU busy for min_gpu_s: a
    20-iteration batch of a sub-millisecond workload is 10 ms of GPU time -- shorter than the clock ramp of a GPU that has
    just been idle.  -> seconds per batch, the contract-style first batch included."""
    batches, acc = [dt_first], dt_first
    while acc < min_gpu_s and len(batches) < 400:
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        d = time.perf_counter() - t0
        batches.append(d)
        acc += d
    return batches


def run_sweep(rig, fence):
    """The other BASELINE workloads on this GPU: 5 warm-ups, then batches of 20 iterations until each entry has run for at
    least SWEEP_MIN_GPU_S of GPU time; `ms_per_step` / `value` are the MEDIAN batch, the fastest and the first batch beside
    them.  (Round 4 timed ONE batch: 10 - 40 ms of GPU time per entry, read 3 - 7 % low on the small piles.)"""
    out = []
    for name in SWEEP:
        N, ns, H, _, label = CONFIGS[name]
        steps, warm = 20, 5
        t_wall = time.perf_counter()
        roof_bwd = None
        if name == 'gd-demo':
            g = bench_gd(rig, N, ns // 30, 30, H, steps, warm, fence, want_median=False)
            B, roof, per_class, dt = g['B'], g['roofline'], g['per_class'], g['dt']
            kbar, step = g['kbar'], g['step']
            roof_bwd = g['roofline_backward']
        else:
            m = bench_mppi(rig, N, ns, H, 0, steps, warm, lambda e: e.mpc_update_device(), fence,
                           ['graph', 'node_encode', 'prop', 'reward', 'mppi'], want_median=False)
            B, roof, per_class, dt, kbar = ns, mppi_roofline(rig, m, N, ns, H, steps), m['per_class'], m['dt'], m['kbar']
            step = m['step']
            # HBM-side bytes per launch of the dominant kernel, where a PMC pass of this preset is on file (tools/profile_r06.sh)
            tkey = ('rollout' if roof.get('graph_build_in_launch') else
                    'prop3' if roof.get('propagation_steps_per_launch', 0) >= 3 else m['dominant'])
            tb = load_traffic().get(name, {}).get(tkey, {}).get('hbm_bytes_per_launch')
            if tb:
                roof['traffic'] = tb
        batches = more_batches(step, fence, steps, dt, SWEEP_MIN_GPU_S)
        med, best = float(np.median(batches)), float(min(batches))
        out.append({'name': name, 'value': B * N * H * steps / med, 'ms_per_step': med / steps * 1e3, 'frac': roof['frac'],
                    'dominant_kernel': roof['kernel'].split(':')[0], 'kernel': roof['kernel'], 'avg_launch_ms': roof['avg_launch_ms'], 'traffic': roof.get('traffic'),
                    'frac_executed_16bit': roof.get('frac_executed_16bit'), 'mfma_pipe_busy_est': roof.get('mfma_pipe_busy_est'),
                    'hbm_algorithmic_frac': roof.get('hbm_algorithmic_frac'),
                    'sclk_mhz_under_load': roof.get('sclk_mhz_under_load'), 'frac_at_measured_clock': roof.get('frac_at_measured_clock'),
                    'bound': roof['bound'], 'achieved': roof['achieved'], 'peak': roof['peak'], 'roofline_unit': roof['unit'],
                    'unit': 'particle-steps/s', 'n_particles': N, 'rows': B, 'n_look_ahead': H, 'steps': steps, 'warmup': warm,
                    'batches': len(batches), 'gpu_active_s': round(float(sum(batches)), 3),
                    'ms_per_step_min': best / steps * 1e3, 'value_max': B * N * H * steps / best,
                    'ms_per_step_first_batch': dt / steps * 1e3, 'mean_in_degree': kbar,
                    'wall_s': round(time.perf_counter() - t_wall, 2),
                    'kernel_ms_per_iteration': {k: round(v[0], 4) for k, v in per_class.items() if v[1] > 0},
                    'executed_per_launch': roof.get('executed_per_launch'),
                    'workload': label, 'peak_basis': roof.get('peak_basis')})
        if roof_bwd:
            out[-1]['roofline_backward'] = roof_bwd
    return out


def run_mpc_step(rig, fence):
    """One whole MPC step the way env/flex_env.py:1016-1106 runs it, on a synthetic observation: obs2ptcl_fixed_num_batch
    (720 x 720 depth image, 30 re-samplings; :1020) -> density (:1022) -> goal field + goal pixels (planners.py:620-624,
    env/flex_rewards.py:172-177; cached per goal: miss and hit timed apart) -> trajectory_optimization_ptcl_multi_traj with the
    shipped planner configuration (config/mpc/config.yaml:38-43: GD, 50 trajectories x 30 columns, horizon 1, 200 update
    iterations, time_lim 2000 ms -> the reference's iteration count) -> the best push's re-rollout and reward (planners.py:
    821-851).  Wall ms per phase, the process's CPU time beside it; the reference's budget for the planner call is its own
    time_lim.  The simulator step is out of scope."""
    import copy
    from dyn_res_pile_manip_amd import synthetic as syn, utils as dev
    from dyn_res_pile_manip_amd.gnn_dyn import PropNetDiffDenModel
    from dyn_res_pile_manip_amd.planners import PlannerGD, gd_iteration_count
    config = copy.deepcopy(syn.default_config())
    config['mpc']['mpc_type'] = 'GD'
    env = syn.SyntheticEnv(config)
    model = PropNetDiffDenModel(config, True, engine=rig.eng)
    model.load_state_dict(rig.sd, strict=False)
    dev.set_engine(rig.eng)
    planner = PlannerGD(config, env)
    subgoal = syn.goal_distance_image(syn.goal_mask('I'))
    obs = syn.render_depth(4000, seed=1, kind='uniform')
    cam = syn.demo_cam_params()
    lo, hi = syn.action_limits()
    out = []
    for N in (20, 50, 100):
        act_seq = np.stack([syn.nominal_pushes(1, seed=10 + i) for i in range(50)], axis=1)           # [1, 50, 4]

        def one(n_update_iter=200, goal_key=None):
            t = {}
            c0 = time.process_time()
            fence()
            t0 = time.perf_counter()
            np.random.seed(0)
            obs_cur, particle_r = dev.obs2ptcl_fixed_num_batch(obs, N, 30, cam, 24.0)
            particle_den = 1.0 / (particle_r * particle_r)
            t['particles'] = time.perf_counter() - t0
            attr_cur = np.zeros((obs_cur.shape[0], N), np.float32)
            t1 = time.perf_counter()
            res = planner.trajectory_optimization_ptcl_multi_traj(
                obs_cur.astype(np.float32), particle_den.astype(np.float32), attr_cur, subgoal, model, act_seq,
                np.zeros(act_seq.shape[0]), n_sample=act_seq.shape[1], n_look_ahead=1, n_update_iter=n_update_iter,
                action_lower_lim=lo, action_upper_lim=hi, use_gpu=True, time_lim=2000.0, goal_key=goal_key)
            t['planner'] = time.perf_counter() - t1
            t['total'] = time.perf_counter() - t0
            t['cpu'] = time.process_time() - c0
            return res, t

        planner._goal_key = None                                   # this pile size's goal pixels are not installed yet
        res_first, t_first = one()                                 # goal cache MISS (and the first launch of these shapes)
        runs = [one() for _ in range(3)]                           # goal cache hits, the goal identified by the digest of its content
        tm = {k: float(np.median([t[k] for _, t in runs])) for k in runs[0][1]}
        # ... and identified by the caller's name for it (goal_key: env/flex_env.py:1048 passes the same subgoal for every MPC step)
        one(goal_key=('I', N))
        named = [one(goal_key=('I', N)) for _ in range(3)]
        tn = {k: float(np.median([t[k] for _, t in named])) for k in named[0][1]}
        r = runs[-1][0]
        tt = r['times']
        out.append({'n_particles': N, 'rows': 50 * 30, 'iterations': int(r['iter_num']) + 1,
                    'reference_iteration_count': gd_iteration_count(200, 2000.0, N),
                    'ms_total': tm['total'] * 1e3, 'ms_particles': tm['particles'] * 1e3, 'ms_planner_call': tm['planner'] * 1e3,
                    'ms_goal_install_hit': tt.get('goal_time', 0.0) * 1e3, 'ms_goal_install_miss': res_first['times'].get('goal_time', 0.0) * 1e3,
                    'ms_goal_install_hit_named': float(np.median([rr['times'].get('goal_time', 0.0) for rr, _ in named])) * 1e3,
                    'ms_total_goal_named': tn['total'] * 1e3, 'ms_planner_call_goal_named': tn['planner'] * 1e3,
                    'ms_optimisation_loop': float(tt['optim_time']), 'ms_best_push_rollout_and_reward': float(tt['best_rollout_time']),
                    'ms_total_first_call_goal_miss': t_first['total'] * 1e3,
                    'host_cpu_ms': tm['cpu'] * 1e3,
                    'reference_budget_ms': 2000.0, 'budget_source': 'config/mpc/config.yaml:40 time_lim (the planner call alone)',
                    'push': [round(float(x), 3) for x in r['action_sequence'][0]], 'predicted_reward': float(r['reward'][0])})
    return out


def run_train_step(rig, fence):
    """One iteration of train/train_gnn_dyn.py:159-210 on the device (row f4): the reference's batch (config/train/gnn_dyn.yaml:
    batch_size 4, n_rollout 5) of synthetic push episodes, zero-padded as collate_fn pads; upload, five steps forward with the
    tape, loss, backward through time, every weight gradient, Adam, the packed copies of the weights rebuilt.  Timed over >= 0.5 s
    in batches of 20 iterations: the same batch every iteration (the shape every per-iteration table is cached for) and eight
    batches of different sizes in turn (what a training run does).  The engine gets its weights back afterwards."""
    from dyn_res_pile_manip_amd import synthetic as syn, weights
    eng = rig.eng
    H = 5
    fixed = syn.push_batch(0, 4, H, sizes=(300, 240, 150, 280))
    varying = [syn.push_batch(1 + i, 4, H) for i in range(8)]
    eng.train_begin(H, 1e-3, 0.9)
    out = {'batch_size': 4, 'n_rollout': H, 'source': 'config/train/gnn_dyn.yaml; train/train_gnn_dyn.py:159-210',
           'n_max_fixed': int(fixed[0].shape[2]), 'n_max_varying': [int(b[0].shape[2]) for b in varying]}

    def timed(batches, mode, min_s):
        for b in batches[:2]:
            eng.train_step(*b, mode=mode)
        fence()
        per, tot, k = [], 0.0, 0
        while tot < min_s:
            t0 = time.perf_counter()
            for _ in range(20):
                eng.train_step(*batches[k % len(batches)], mode=mode)
                k += 1
            fence()
            dt = time.perf_counter() - t0
            per.append(dt / 20 * 1e3)
            tot += dt
        return float(np.median(per)), float(np.min(per)), tot

    md, mn, t1 = timed([fixed], 'update', 0.5)
    out['ms_per_iteration'] = round(md, 4)
    out['ms_per_iteration_min'] = round(mn, 4)
    md, mn, t2 = timed([fixed], 'eval', 0.25)
    out['ms_forward_only'] = round(md, 4)
    md, mn, t3 = timed(varying, 'update', 0.5)
    out['ms_per_iteration_varying_batches'] = round(md, 4)
    out['gpu_active_s'] = round(t1 + t2 + t3, 3)
    eng.load_weights(weights.blob_from_state_dict(rig.sd), 0.08)
    return out


def run_rank(args):
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.stderr.write('bench.py: WORLD_SIZE=%d but --gpus %d: the launcher and the flag must agree\n' % (world, args.gpus))
        return EXIT_WORLD_MISMATCH
    if args.force_comm:
        os.environ['DRP_COMM_ALWAYS'] = '1'          # read at drp_create
    rc = preflight(args)                             # under a launcher every rank checks for itself, before any rendezvous
    if rc:
        return rc
    import datetime
    import torch
    import torch.distributed as dist
    from dyn_res_pile_manip_amd.sharding import shard_range

    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    gloo = args.comm == 'gloo'
    use_rccl = (world > 1 or args.force_comm) and not gloo
    if world > 1:
        # rendezvous, id broadcast, barrier and timing reductions over gloo: torch's NCCL process group is never
        # created, the only RCCL communicator of the process is the engine's
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('gloo', rank=rank, world_size=world,
                                # generous by default: on a fresh box the ranks' first `import torch` can differ by minutes
                                timeout=datetime.timedelta(seconds=float(os.environ.get('DRP_BENCH_RENDEZVOUS_TIMEOUT_S', '900'))))

    rig = Rig(local_rank, args.engine)
    eng, engine = rig.eng, rig.engine
    comm_info = None
    if use_rccl:
        uid = [eng.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0], rank, world)
        comm_info = eng.comm_info()
        if comm_info['n_ranks'] != world:
            sys.stderr.write('bench.py: ncclCommCount=%d, WORLD_SIZE=%d\n' % (comm_info['n_ranks'], world))
            return EXIT_WORLD_MISMATCH
        if world > 1:
            # every rank must have bound the same RCCL (the library follows the HIP runtime the process runs on: include/drp.h)
            seen = [None] * world
            dist.all_gather_object(seen, (comm_info['version'], comm_info['path']))
            if len(set(v for v, _ in seen)) != 1:
                sys.stderr.write('bench.py: the ranks bound different RCCL versions: %r\n' % (seen,))
                return EXIT_WORLD_MISMATCH
    # what EVERY rank's communicator says of itself (ncclCommCount, the library's version and file), for the line
    rccl_ranks = None
    if comm_info:
        mine = {'rank': rank, 'local_rank': local_rank, 'comm_count': comm_info['n_ranks'], 'version': comm_info['version_str'],
                'library': comm_info['path']}
        rccl_ranks = [mine]
        if world > 1:
            rccl_ranks = [None] * world
            dist.all_gather_object(rccl_ranks, mine)

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank_ms(dt_own):
        """Every rank's own ms per step of the timed region (the line's ms_per_step is their maximum): a straggler shows."""
        ms = dt_own / args.steps * 1e3
        if world == 1:
            return {'min': ms, 'max': ms, 'all': [round(ms, 4)]}
        t = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(t, torch.tensor([ms], dtype=torch.float64))
        v = [float(x.item()) for x in t]
        return {'min': min(v), 'max': max(v), 'all': [round(x, 4) for x in v]}

    N, H = args.particles, args.horizon
    out, cpu_args = None, None
    if args.config_name == 'gd-demo':
        nb = 30
        rows = args.samples_total_job // world
        traj = max(1, rows // nb)
        g = bench_gd(rig, N, traj, nb, H, args.steps, args.warmup, fence, rank=rank)
        dt, med = max_over_ranks(g['dt']), max_over_ranks(g['median'])
        rank_ms = per_rank_ms(g['dt'])
        if rank == 0:
            from dyn_res_pile_manip_amd.planners import particle_num_to_iter_time
            B = g['B']
            out = {'metric': 'MPC rollout-steps/sec (samples x particles x steps/sec)', 'value': world * B * N * H * args.steps / dt,
                   'unit': 'particle-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                   'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                   'ms_per_step_median': med * 1e3, 'value_median': world * B * N * H / med, 'rank_ms_per_step': rank_ms,
                   'dtype': 'f32 (forward MLP products as split fp16 / bf16 MFMA terms, backward node stages on fp32 MFMA, fp32 accumulate)',
                   'data': 'synthetic',
                   'config': {'workload': args.workload, 'name': args.config_name, 'n_particles': N, 'n_trajectories_per_gpu': traj,
                              'n_batch': nb, 'rows_per_gpu': B, 'n_look_ahead': H, 'engine': 'fused', 'mean_in_degree': g['kbar'],
                              'parallelism': 'trajectories sharded x%d, no collective' % world,
                              'reference_time_model_ms': particle_num_to_iter_time(N)},     # planners.py:25-28, batch 300 on its GPU
                   'roofline': g['roofline'], 'roofline_backward': g['roofline_backward'],
                   'kernel_ms_per_iteration': {k: round(v[0], 4) for k, v in g['per_class'].items()},
                   'cpu_baseline': None}
            if world == 1 and not args.no_cpu_baseline:
                from oracle import propnet_dense as od
                syn = rig.syn
                W = od.load_weights(rig.sd)
                torch.set_num_threads(min(32, os.cpu_count() or 1))
                a_cpu = g['acts']            # the whole demo batch: about 5 s of host work
                od.gd_loss_and_grads(W, g['s0'], g['dens'], g['attr'], a_cpu[:nb], rig.G, rig.cam, rig.goal_coor, syn.demo_cam_extrinsics(), 24)   # warm-up
                t0 = time.perf_counter()
                od.gd_loss_and_grads(W, g['s0'], g['dens'], g['attr'], a_cpu, rig.G, rig.cam, rig.goal_coor, syn.demo_cam_extrinsics(), 24)
                dtc = time.perf_counter() - t0
                out['cpu_baseline'] = {'value': traj * nb * N * H / dtc, 'unit': 'particle-steps/s', 'cores': torch.get_num_threads(),
                                       'kind': 'port', 'cpu_model': host_cpu_model(),
                                       'sample': '%d trajectories x %d columns x %d particles, forward + autograd backward of '
                                                 'oracle/propnet_dense.py, %.1f s' % (traj, nb, N, dtc)}
    else:
        s_lo, s_hi = shard_range(args.samples_total_job, rank, world)      # this rank's contiguous block of samples
        ns = s_hi - s_lo
        if gloo:
            from dyn_res_pile_manip_amd.sharding import allgather_records

        def update(e):
            if gloo and world > 1:
                # host transport: this rank's record -> all-gather over gloo -> combine kernel
                if args.update == 'elite':
                    e.mpc_update_elite(allgather_records(e.mpc_elite(args.elite).ravel()).reshape(world, args.elite, -1), args.elite)
                else:
                    e.mpc_update(allgather_records(e.mpc_partials()))
            elif args.update == 'elite':
                e.mpc_update_elite_device(args.elite)
            else:
                e.mpc_update_device()

        m = bench_mppi(rig, N, ns, H, s_lo, args.steps, args.warmup, update, fence, KERNEL_CLASSES, want_median=True,
                       fault=(args.fault_rank == rank))
        dt, med = max_over_ranks(m['dt']), max_over_ranks(m['median'])
        rank_ms = per_rank_ms(m['dt'])
        step = m['step']
        # the same K steps on the pure-fp32 MFMA engine (bit-for-bit an fp32 fma chain), for reference; its un-fused
        # pipeline has the segmented sum ("scatter-add") as a kernel of its own, timed here with the HIP-event probe
        alt, scatter = None, None
        tj = load_traffic() if (N, ns) == (300, 1024) else {}
        if engine != 'mfma' and not args.no_alt:
            eng.set_engine(rig._lib.ENGINES['mfma'])
            for _ in range(2):
                step()
            eng.probe_begin('aggregate')
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            fence()
            dta = max_over_ranks(time.perf_counter() - t0)
            agg_ms, agg_n = eng.probe_read()
            eng.probe_begin(None)
            kbar_alt = float(eng.debug_fetch('nbr_cnt', (ns, N), np.uint8).mean())
            if agg_n > 0:
                scatter = scatter_roofline(ns, N, kbar_alt, agg_ms, agg_n,
                                           tj.get('mfma', {}).get('aggregate', {}).get('hbm_bytes_per_launch'))
            alt = {'engine': 'mfma', 'dtype': 'f32', 'value': args.samples_total_job * N * H * args.steps / dta,
                   'ms_per_step': dta / args.steps * 1e3}
            eng.set_engine(rig._lib.ENGINES[engine])
        if rank == 0:
            roof = mppi_roofline(rig, m, N, ns, H, args.steps)
            # HBM-side bytes per launch of this kernel from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
            # profiles/summarize_pmc.py), collected on this same workload
            roof['traffic'], roof['traffic_source'] = None, None
            tkey = 'prop3' if (m['dominant'] == 'prop' and roof.get('propagation_steps_per_launch', 0) >= 3) else m['dominant']
            tb = tj.get(engine, {}).get(tkey, {}).get('hbm_bytes_per_launch')
            if tb:
                roof['traffic'], roof['traffic_source'] = tb, TRAFFIC_SOURCE
            roof = ordered(roof)
            comm_label = None
            if use_rccl:
                comm_label = 'rccl, %d rank%s' % (world, '' if world == 1 else 's')
            elif world > 1:
                comm_label = 'gloo, %d ranks%s' % (world, ', all on GPU 0' if args.share_gpu else '')
            out = {
                'metric': 'MPC rollout-steps/sec (samples x particles x steps/sec)',
                'value': args.samples_total_job * N * H * args.steps / dt, 'unit': 'particle-steps/s', 'n_gpus': world,
                'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
                'scaling': args.scaling, 'vs_baseline': None,
                'ms_per_step_median': med * 1e3, 'value_median': args.samples_total_job * N * H / med, 'rank_ms_per_step': rank_ms,
                'dtype': 'f32' if engine in ('valu', 'mfma') else 'f32 (MLP products as split fp16 / bf16 MFMA terms, fp32 accumulate)',
                'data': 'synthetic',
                'config': {'workload': args.workload, 'name': args.config_name,
                           'n_particles': N, 'n_sample_per_gpu': ns, 'n_sample_total': args.samples_total_job,
                           'n_look_ahead': H, 'engine': engine, 'gpus_requested': args.gpus, 'world_size': world,
                           'communicator': comm_label,
                           'rccl': ({'comm_count': comm_info['n_ranks'], 'version': comm_info['version_str'],
                                     'library': comm_info['path'], 'ranks': rccl_ranks} if comm_info else None),
                           'rendezvous': 'gloo (id broadcast, barrier, timing reductions)' if world > 1 else None,
                           'mean_in_degree': m['kbar'], 'parallelism': 'samples sharded x%d' % world,
                           'update': 'softmax mean (optimize_action)' if args.update == 'mppi' else 'mean of the %d best (elite)' % args.elite},
                'roofline': roof,
                'kernel_ms_per_iteration': {k: round(v[0], 4) for k, v in m['per_class'].items()},
                'roofline_scatter': scatter, 'alt_engine': alt,
            }
            out['cpu_baseline'] = None
            cpu_args = (args, rig.sd, m['s0'], m['dens'], m['attr'], rig.G, rig.goal_coor, rig.cam) if (world == 1 and not args.no_cpu_baseline) else None
            out['gpu_active_s'] = round(float(m['dt'] + (m['median'] or 0.0) * args.steps + (dta if alt else 0.0)), 3)
    if rank == 0 and args.do_sweep and world == 1:
        # every GPU leg before the host-side one: the GPU is busy in one stretch (the sweep alone keeps it busy for
        # len(SWEEP) x SWEEP_MIN_GPU_S seconds), the CPU baseline follows
        out['sweep'] = run_sweep(rig, fence)
        out['mpc_step'] = run_mpc_step(rig, fence)
        out['train_step'] = run_train_step(rig, fence)
        out['gpu_active_s'] = round(out.get('gpu_active_s', 0.0) + sum(e['gpu_active_s'] for e in out['sweep'])
                                    + sum(e['ms_total'] * 4e-3 for e in out['mpc_step']) + out['train_step']['gpu_active_s'], 3)
    if rank == 0 and cpu_args is not None:
        out['cpu_baseline'] = cpu_baseline(*cpu_args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)
    return run_rank(args)


if __name__ == '__main__':
    sys.exit(main())
