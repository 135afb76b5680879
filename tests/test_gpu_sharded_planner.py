"""GPU: the planner's comm= argument (sample axis sharded over ranks).

One MI355X is what a test box has, so: (1) a one-rank RCCL communicator through the device transport
(RcclComm, ncclAllGather forced even for one rank) must reproduce the un-sharded call bit for bit; (2) two
PROCESSES sharing GPU 0, exchanging over gloo (TorchComm: records fetched, all-gathered on the host, uploaded
to the combine kernel), must both return the un-sharded result: same pushes, same predicted observation --
the Philox stream is keyed by the GLOBAL sample index, so the shards draw exactly the un-sharded samples."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


@pytest.mark.parametrize('mpc_type', ['MPPI', 'CEM', 'GD'])
def test_one_rank_rccl_communicator_equals_no_communicator(mpc_type, monkeypatch):
    import _shard_worker as w
    from dyn_res_pile_manip_amd.engine import Engine
    from dyn_res_pile_manip_amd.sharding import RcclComm
    ref = w.run_planner(mpc_type, None)
    monkeypatch.setenv('DRP_COMM_ALWAYS', '1')         # one rank still goes through ncclAllGather
    probe = Engine(0)
    uid, uid2 = probe.comm_unique_id(), probe.comm_unique_id()     # an ncclUniqueId serves one communicator
    probe.close()
    got = w.run_planner(mpc_type, RcclComm(uid, 0, 1))
    got_t = w.run_planner(mpc_type, (0, 1, uid2))       # the documented tuple form
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
        np.testing.assert_array_equal(got_t[k], ref[k], err_msg=k)


def test_a_spent_unique_id_and_a_second_engine_are_refused(monkeypatch):
    """A ncclUniqueId serves one ncclCommInitRank: re-initialising with it would never return."""
    from dyn_res_pile_manip_amd._lib import DrpError
    from dyn_res_pile_manip_amd.engine import Engine
    from dyn_res_pile_manip_amd.sharding import RcclComm
    a, b = Engine(0), Engine(0)
    uid = a.comm_unique_id()
    comm = RcclComm(uid, 0, 1)
    comm.attach(a)
    comm.attach(a)                                      # the same engine again: nothing to do
    assert a.comm_info()['n_ranks'] == 1 and a.comm_info()['path'].find('librccl') >= 0
    with pytest.raises(RuntimeError, match='fresh'):
        comm.attach(b)
    with pytest.raises(DrpError, match='already been used'):
        b.comm_init(uid, 0, 1)
    a.comm_destroy()
    assert a.comm_info()['n_ranks'] == 0
    a.close()
    b.close()


def _two_ranks(tmp_path, mpc_type, transport):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29500 + os.getpid() % 2000), WORLD_SIZE='2',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_shard_worker.py'), str(tmp_path), mpc_type, transport],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(outs)


@pytest.mark.parametrize('mpc_type', ['MPPI', 'CEM', 'GD'])
def test_two_gpus_over_rccl_equal_one(mpc_type, tmp_path):
    """The product transport with MORE than one rank: RcclComm on two GPUs (the packed [statistics | k elite] message,
    rank_stride, the one-exchange merge of the run records).  Needs two devices; a one-GPU box skips it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    import _shard_worker as w
    ref = w.run_planner(mpc_type, None)
    _two_ranks(tmp_path, mpc_type, 'rccl')
    res = [np.load(os.path.join(str(tmp_path), '%s_rank%d.npz' % (mpc_type, r))) for r in range(2)]
    for r in res:
        assert int(r['iter_num']) == int(ref['iter_num'])
        np.testing.assert_allclose(r['action_sequence'], ref['action_sequence'], atol=0 if mpc_type == 'GD' else 1e-5)
        np.testing.assert_allclose(r['reward'], ref['reward'], rtol=1e-5)
        np.testing.assert_allclose(r['rew_mean'], ref['rew_mean'], rtol=1e-5)
    np.testing.assert_array_equal(res[0]['action_sequence'], res[1]['action_sequence'])


@pytest.mark.parametrize('mpc_type', ['MPPI', 'CEM', 'GD'])
def test_two_processes_sharing_the_gpu_equal_one(mpc_type, tmp_path):
    import _shard_worker as w
    ref = w.run_planner(mpc_type, None)
    _two_ranks(tmp_path, mpc_type, 'gloo')
    res = [np.load(os.path.join(str(tmp_path), '%s_rank%d.npz' % (mpc_type, r))) for r in range(2)]
    for r in res:
        assert int(r['iter_num']) == int(ref['iter_num'])
        if mpc_type == 'GD':
            # independent Adam problems: a shard's trajectories are bit-identical to the same rows of the full run
            np.testing.assert_array_equal(r['action_sequence'], ref['action_sequence'])
            np.testing.assert_array_equal(r['observation_sequence'], ref['observation_sequence'])
        else:
            # the softmax / elite mean is combined from two partial sums instead of one: last-ulp differences in the
            # nominal sequence, hence in later samples
            np.testing.assert_allclose(r['action_sequence'], ref['action_sequence'], atol=1e-5)
            np.testing.assert_allclose(r['observation_sequence'], ref['observation_sequence'], atol=1e-6)
        np.testing.assert_allclose(r['reward'], ref['reward'], rtol=1e-5)
        np.testing.assert_allclose(r['rew_mean'], ref['rew_mean'], rtol=1e-5)
        np.testing.assert_allclose(r['rew_std'], ref['rew_std'], rtol=1e-4)
    # both ranks return the same plan; their action_full / reward_full are the two halves of the sample axis
    np.testing.assert_array_equal(res[0]['action_sequence'], res[1]['action_sequence'])
    full = np.concatenate([res[0]['reward_full'], res[1]['reward_full']])
    assert full.shape == ref['reward_full'].shape
    np.testing.assert_allclose(full, ref['reward_full'], rtol=1e-5)


@pytest.mark.parametrize('order', ['drp_first', 'torch_first', 'no_torch'])
def test_the_communicator_comes_up_whatever_the_import_order(order):
    """PyTorch's wheel ships its own HIP runtime and its own RCCL.  Imported BEFORE this library the process runs on that
    runtime and must use that RCCL; imported AFTER it (or not at all) the library is bound to the system runtime, and the
    wheel's RCCL -- mapped by then -- would talk to a runtime nobody initialised (ncclCommInitRank: "no ROCm-capable device
    is detected").  The library binds the RCCL that sits next to the HIP runtime it itself runs on."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    # seven seconds on a healthy box.  Twice in round 4 a fresh box sat in this child for minutes (two HIP runtimes mapped
    # into one process is the very situation under test): ONE retry in a fresh child; a second hang FAILS -- a regression of
    # the run-time RCCL binding looks exactly like this -- with the child's own stack dump (faulthandler, after 45 s) and
    # whatever it printed, so that a box problem can be told from a hang inside the library
    p, hung = None, []
    for attempt in range(2):
        try:
            p = subprocess.run([sys.executable, os.path.join(root, 'tests', '_rccl_order.py'), order], cwd=root, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
            break
        except subprocess.TimeoutExpired as e:
            hung.append((e.output or b'').decode(errors='replace')[-3000:])
            p = None
    if p is None:
        pytest.fail('the child process hung twice for 60 s (import order %s); its output and stack dumps:\n--- attempt 1\n%s\n'
                    '--- attempt 2\n%s' % (order, hung[0], hung[1]))
    out = p.stdout.decode()
    assert p.returncode == 0 and 'communicator up' in out and 'ERR' not in out, out[-2000:]
    up = [l for l in out.splitlines() if l.startswith('communicator up')][0]
    assert ("torch/lib/librccl" in up) == (order == 'torch_first'), up
