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
    # algorithmic bytes over time: at this reduced size (256 samples) the edge rows fit the 256-MB infinity cache, so the
    # figure can pass the HBM peak; the full-size line (1024 samples, 0.9) is what DESIGN quotes
    assert d['roofline_scatter']['bound'] == 'hbm' and 0 < d['roofline_scatter']['frac'] < 3.0
    assert 'workload' in d['config'] and 'model' not in d['config']


@pytest.mark.parametrize('mode', ['weak', 'strong', 'elite'])
def test_two_ranks_sharing_the_gpu(mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    extra = {'weak': ['--samples', '256'], 'strong': ['--samples-total', '512'],
             'elite': ['--samples', '256', '--update', 'elite', '--elite', '16']}[mode]
    d = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
              '--master-port', str(29700 + os.getpid() % 200), 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1',
              '--comm', 'gloo', '--share-gpu', '--no-alt'] + extra, env)
    assert d['n_gpus'] == 2 and d['cpu_baseline'] is None
    assert d['scaling'] == ('strong' if mode == 'strong' else 'weak')
    assert d['config']['n_sample_total'] == 512 and d['config']['n_sample_per_gpu'] == 256
    assert abs(d['value'] - 512 * 300 * 10 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert 'gloo' in d['config']['communicator']


def test_one_rank_through_rccl():
    """--force-comm: init_process_group('nccl'), ncclCommInitRank and ncclAllGather with one rank."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    d = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
              '--master-port', str(29900 + os.getpid() % 90), 'bench.py', '--force-comm', '--steps', '3', '--warmup', '1',
              '--samples', '256', '--no-alt', '--no-cpu-baseline'], env)
    assert d['n_gpus'] == 1 and 'rccl' in d['config']['communicator'] and d['value'] > 0
