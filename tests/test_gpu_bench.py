"""GPU: bench.py's contract line, and its N-rank code path exercised on ONE GPU.

A test box has one MI355X, so the sharded bench runs as two processes that SHARE it (`--share-gpu`) and exchange
their records over gloo (`--comm gloo`: the host transport of sharding.TorchComm) -- everything of the N > 1 branch
except the RCCL calls themselves (those run in `--force-comm` with a one-rank communicator): launcher environment,
contiguous sample blocks and Philox offsets, barrier + max-over-ranks timing, whole-job `value`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]          # rank 0 prints ONE json line
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '1', '--samples', '256', '--cpu-samples', '4'])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'roofline_scatter'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['vs_baseline'] is None
    assert abs(d['value'] - 256 * 300 * 10 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] < 1
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'cpu_model' in c
    # the segmented sum's fraction is of HBM-side bytes (PMC traffic at the profiled shape, compulsory bytes otherwise):
    # a fraction of the HBM peak cannot pass 1; SURVEY 8d's algorithmic figure travels beside it under its own name
    sc = d['roofline_scatter']
    assert sc['bound'] == 'hbm' and 0 < sc['frac'] < 1 and sc['frac_basis'].startswith('compulsory')
    assert sc['algorithmic_bytes_per_launch'] > sc['compulsory_bytes_per_launch'] and sc['cache_served'] is True
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert 'sweep' not in d                                   # a custom shape carries no sweep block
    # `frac` is SURVEY 8d's quantity: algorithmic FLOPs of the factored formulation per launch over the launch's time against
    # the fp32 matrix peak; what the matrix pipe executes (counted by the kernels themselves, drp_probe_work) travels beside it
    # against the 16-bit peak, with the pipe's estimated occupancy and 8d's other roofline (the gather's algorithmic bytes)
    ex = r['executed_per_launch']
    assert r['numerator_executed'].startswith('drp_probe_work')
    assert ex['mfmas'] == 78 * ex['chain_slots'] + 144 * ex['tiles'] + 96 * ex['tiles_last'] + 204 * ex['encoder_tiles']
    assert r['peak'] == 157.3 and r['peak_basis'].startswith('SURVEY.md 8d')
    assert abs(r['achieved'] - r['work_per_launch_flop'] / (r['avg_launch_ms'] * 1e-3) / 1e12) < 1e-6 * r['achieved']
    k = d['config']['mean_in_degree']
    assert abs(r['work_per_launch_flop'] - 256 * 300 * (116096.0 + 25472.0 * k)) < 1e-6 * r['work_per_launch_flop']   # one rollout step per launch
    assert abs(r['frac_executed_16bit'] - ex['mfmas'] * 32768.0 / (r['avg_launch_ms'] * 1e-3) / 1e12 / 2500.0) < 1e-6
    assert 0 < r['frac_executed_16bit'] < 1 and 0 < r['mfma_pipe_busy_est'] < 1 and r['hbm_algorithmic_frac'] > 0
    assert r['executed_over_algorithmic'] > 1
    # what a reader -- or a driver that keeps the first keys of a block -- needs comes FIRST: the fraction, the
    # kernel by name, its launch time, the counters' bytes, the executed-work pair, and the shader clock the counting launches
    # saw (s_memtime over s_memrealtime inside the kernels) with the fraction at that clock
    assert list(r)[:10] == ['frac', 'kernel', 'avg_launch_ms', 'launches', 'traffic', 'frac_executed_16bit', 'mfma_pipe_busy_est',
                            'hbm_algorithmic_frac', 'sclk_mhz_under_load', 'frac_at_measured_clock'], list(r)[:10]
    assert isinstance(r[list(r)[-1]], (str, dict, type(None)))          # explanatory strings last
    assert r['kernel'].startswith('prop: km_prop')
    assert 1500.0 < r['sclk_mhz_under_load'] < 2600.0
    assert abs(r['frac_at_measured_clock'] - r['frac'] * 2400.0 / r['sclk_mhz_under_load']) < 1e-9
    assert set(c['seconds_by_threads']) == set(c['value_by_threads']) and c['cores'] == int(min(c['seconds_by_threads'], key=lambda k: c['seconds_by_threads'][k]))


def test_the_propagation_kernels_count_what_they_execute():
    """drp_probe_work against a count made here from the lists the rollout leaves behind: 8 samples of 300 particles are one
    sample per workgroup, ten tiles of 32 receivers in the natural order (a saturated pile) or ordered by in-degree, every
    tile running as many slot iterations as its largest in-degree minus the self loop -- for every propagation step of
    every rollout step; with the edge-chain cache the chain runs in a third of them."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from dyn_res_pile_manip_amd import synthetic as syn, weights
    from dyn_res_pile_manip_amd.engine import Engine
    from dyn_res_pile_manip_amd.planners import world2cam_affine
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(weights.random_state_dict(seed=0)), 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    n_cu = eng.device_info()['n_cu']
    for N, ns, H in ((300, n_cu, 1), (50, 2 * n_cu, 2), (20, 2 * n_cu, 2)):
        s0, dens, attr = syn.make_pile(N, 1, seed=1)
        acts = syn.sample_pushes(ns, H, seed=2)
        eng.probe_begin('prop+work')
        eng.rollout(s0, attr, dens, acts)
        ms, launches = eng.probe_read()
        w = eng.probe_work()
        eng.probe_begin(None)
        assert launches >= 1 and ms > 0
        assert w['mfmas'] == 78 * w['chain_slots'] + 144 * w['tiles'] + 96 * w['tiles_last'] + 204 * w['encoder_tiles']
        assert w['tiles'] == 2 * w['tiles_last'] and w['tiles_last'] % H == 0
        spw = ns // n_cu
        rows = spw * N
        if N == 300:
            # one sample per workgroup, ten tiles of 32 receivers: in the natural order when the pile is saturated, by
            # in-degree (largest first) otherwise; the self loop is a constant, not a slot
            assert w['tiles_last'] == ns * 10 and w['encoder_tiles'] == ns * 10 and w['cached_slots'] == 0
            cnt = eng.debug_fetch('nbr_cnt', (ns, N), np.uint8).astype(int)
            slots = 0
            for c in cnt:
                order = c if (c == c.max()).sum() * 16 >= c.size * 15 else np.sort(c)[::-1]
                slots += int((np.pad(order, (0, 20)).reshape(10, 32).max(-1) - 1).sum())
            assert w['chain_slots'] == 3 * slots
        else:
            tiles = (ns // spw) * (-(-rows // (16 if rows <= 128 else 32)))
            assert w['tiles_last'] == H * tiles, (w, tiles)
            assert w['cached_slots'] == 2 * w['chain_slots'] > 0            # these shapes run the cached kernels
    eng.close()


def test_the_sweep_block_carries_the_other_baseline_workloads():
    d = _run([sys.executable, 'bench.py', '--steps', '2', '--warmup', '1', '--samples', '128', '--no-alt', '--no-cpu-baseline',
              '--sweep'])
    names = [e['name'] for e in d['sweep']]
    assert names == ['p20', 'c4-50', 'c4-150', 'c4-600', 'c5-share', 'gd-demo']
    for e in d['sweep']:
        assert e['steps'] == 20 and e['warmup'] == 5 and e['value'] > 0 and 0 < e['frac'] < 1.2, e
        assert abs(e['value'] - e['rows'] * e['n_particles'] * e['n_look_ahead'] / (e['ms_per_step'] * 1e-3)) < 1e-6 * e['value']
        # batches of 20 iterations until the entry has kept the GPU busy for a second: median, fastest and first batch
        assert e['batches'] >= 1 and e['gpu_active_s'] >= 0.99 and e['ms_per_step_min'] <= e['ms_per_step'], e
        assert e['dominant_kernel'] in e['kernel_ms_per_iteration']
    assert d['sweep'][4]['n_particles'] == 1200 and d['sweep'][4]['rows'] == 512 and d['sweep'][4]['n_look_ahead'] == 20
    assert d['gpu_active_s'] >= 6.0                               # the GPU legs run in one stretch, before the host-side baseline
    # one whole MPC step (env/flex_env.py:1016-1106) per pile size of the planner's regime, phases in ms, next to the
    # reference's budget for the planner call
    assert [e['n_particles'] for e in d['mpc_step']] == [20, 50, 100]
    for e in d['mpc_step']:
        assert e['iterations'] == e['reference_iteration_count'] and e['rows'] == 1500
        assert 0 < e['ms_particles'] < e['ms_total'] and 0 < e['ms_optimisation_loop'] <= e['ms_planner_call'] < e['reference_budget_ms']
        assert e['ms_goal_install_hit'] < e['ms_goal_install_miss']
        assert all(s.syn_lo <= v <= s.syn_hi for v, s in zip(e['push'], [type('b', (), {'syn_lo': -5.01, 'syn_hi': 5.01})] * 4))
    # one training iteration (train/train_gnn_dyn.py:159-210) at the reference's batch: the same batch every iteration, and
    # eight batches of different sizes in turn
    t = d['train_step']
    assert t['batch_size'] == 4 and t['n_rollout'] == 5 and len(t['n_max_varying']) == 8
    assert 0 < t['ms_forward_only'] < t['ms_per_iteration_min'] <= t['ms_per_iteration'] < 10.0, t
    assert 0 < t['ms_per_iteration_varying_batches'] < 10.0 and t['gpu_active_s'] >= 1.2, t


@pytest.mark.parametrize('mode', ['weak', 'strong', 'elite'])
def test_two_ranks_sharing_the_gpu(mode):
    """`bench.py --gpus 2` with NO launcher around it: the parent spawns the two ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    extra = {'weak': ['--samples', '256'], 'strong': ['--samples-total', '512'],
             'elite': ['--samples', '256', '--update', 'elite', '--elite', '16']}[mode]
    d = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1',
              '--comm', 'gloo', '--share-gpu', '--no-alt'] + extra, env)
    assert d['n_gpus'] == 2 and d['cpu_baseline'] is None
    assert d['scaling'] == ('strong' if mode == 'strong' else 'weak')
    assert d['config']['n_sample_total'] == 512 and d['config']['n_sample_per_gpu'] == 256
    assert d['config']['world_size'] == 2 and d['config']['gpus_requested'] == 2
    assert abs(d['value'] - 512 * 300 * 10 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert 'gloo' in d['config']['communicator']


def test_eight_ranks_sharing_the_gpu():
    """`bench.py --gpus 8` as the scaling run calls it (no launcher, configs[2]: 1024 samples per rank would be 8192 -- here
    64 per rank): eight child ranks, gloo rendezvous, one exchange per iteration, max-over-ranks timing, ONE line."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    d = _run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '3', '--warmup', '1', '--comm', 'gloo', '--share-gpu',
              '--no-alt', '--samples', '64'], env)
    assert d['n_gpus'] == 8 and d['config']['world_size'] == 8 and d['config']['gpus_requested'] == 8
    assert d['config']['n_sample_total'] == 512 and d['config']['n_sample_per_gpu'] == 64
    assert d['scaling'] == 'weak' and d['cpu_baseline'] is None
    assert abs(d['value'] - 512 * 300 * 10 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']


def test_two_ranks_under_the_drivers_launcher():
    """The way the driver starts N > 1: torch.distributed.run around bench.py (ranks from the environment)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    d = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
              '--master-port', str(29700 + os.getpid() % 200), 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1',
              '--comm', 'gloo', '--share-gpu', '--no-alt', '--samples', '256'], env)
    assert d['n_gpus'] == 2 and d['config']['n_sample_total'] == 512


def test_a_rank_that_dies_ends_the_run_with_an_error():
    """Rank 1 exits before the first timed update (--fault-rank): rank 0 would wait in the exchange; the parent ends
    the group and exits non-zero, promptly and without a JSON line."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', DRP_COMM_TIMEOUT_S='20')
    t0 = time.time()
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--comm', 'gloo', '--share-gpu',
                        '--no-alt', '--samples', '128', '--fault-rank', '1', '--timeout', '200'], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=400)
    assert p.returncode == 7, (p.returncode, p.stderr.decode()[-1500:])
    assert 'rank 1 exited with status 7' in p.stderr.decode()
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert time.time() - t0 < 200


def test_one_rank_through_rccl():
    """--force-comm with no launcher: ncclCommInitRank and ncclAllGather with one rank; the line names the RCCL that
    served it, and the communicator's own count."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    d = _run([sys.executable, 'bench.py', '--gpus', '1', '--force-comm', '--steps', '3', '--warmup', '1',
              '--samples', '256', '--no-alt', '--no-cpu-baseline'], env)
    assert d['n_gpus'] == 1 and d['config']['communicator'] == 'rccl, 1 rank' and d['value'] > 0
    rc = d['config']['rccl']
    assert rc['comm_count'] == 1 and 'librccl' in rc['library'] and rc['version'][0].isdigit()
    # one RCCL per process: the bench imports torch, so the copy torch ships is the one the engine bound
    assert os.sep + 'torch' + os.sep in rc['library']


def test_more_ranks_than_gpus_is_refused_before_any_rank_starts():
    """One GPU here: `--gpus 2` without --share-gpu exits 4 at once with one line that says why (a driver's scaling run on a
    smaller box must not sit in a rendezvous); the count comes from the render nodes, no HIP call in the parent."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    have = bench.visible_gpu_count()
    assert have is not None and have >= 1
    t0 = time.time()
    p = subprocess.run([sys.executable, 'bench.py', '--gpus', str(have + 1), '--steps', '1', '--warmup', '0'], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 4 and b'not starting' in p.stderr and not p.stdout.strip()
    assert time.time() - t0 < 60
