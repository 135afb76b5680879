"""CPU: host-side logic that needs no GPU -- the list conversions of the `model.forward` / `particle_nums`
compatibility paths (model/gnn_dyn.py:238-251), and bench.py's launcher: `--gpus N` without a launcher around it
spawns its own N rank processes, relays one JSON line, and exits non-zero when a rank fails or the world size and
the flag disagree (nothing of that may leave a process behind)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_lists(rng, B, N):
    idx = -np.ones((B, N, 10), np.int16)
    cnt = np.zeros((B, N), np.uint8)
    for b in range(B):
        for i in range(N):
            k = int(rng.integers(0, min(10, N) + 1))
            idx[b, i, :k] = np.sort(rng.choice(N, k, replace=False))
            cnt[b, i] = k
    return idx, cnt


def test_dense_relations_become_the_lists_they_were_built_from():
    from dyn_res_pile_manip_amd.gnn_dyn import relations_to_lists
    rng = np.random.default_rng(0)
    B, N = 3, 13
    idx, cnt = _random_lists(rng, B, N)
    E = int(cnt.astype(np.int64).sum(1).max()) + 2            # zero rows pad short samples (model/gnn_dyn.py:248-251)
    Rr = np.zeros((B, E, N), np.float32)
    Rs = np.zeros((B, E, N), np.float32)
    for b in range(B):
        e = 0
        for i in range(N):                                    # nonzero() order: receiver-major, sender ascending (:247)
            for j in idx[b, i, :cnt[b, i]]:
                Rr[b, e, i] = 1
                Rs[b, e, j] = 1
                e += 1
    got_idx, got_cnt = relations_to_lists(Rr, Rs)
    np.testing.assert_array_equal(got_idx, idx)
    np.testing.assert_array_equal(got_cnt, cnt)


def test_particle_nums_masking_matches_the_row_by_row_rule():
    from dyn_res_pile_manip_amd.gnn_dyn import mask_lists
    rng = np.random.default_rng(1)
    B, N = 4, 17
    idx, cnt = _random_lists(rng, B, N)
    nums = np.array([17, 9, 1, 12])
    got_idx, got_cnt = mask_lists(idx.copy(), nums)
    for b in range(B):
        n = int(nums[b])
        for i in range(N):
            js = [j for j in idx[b, i] if 0 <= j < n] if i < n else []     # rows and columns >= n leave the graph (:238-241)
            assert got_cnt[b, i] == len(js)
            np.testing.assert_array_equal(got_idx[b, i, :len(js)], js)
            assert (got_idx[b, i, len(js):] == -1).all()


def _bench(argv, env=None, timeout=300):
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    return p.returncode, p.stdout.decode(), p.stderr.decode(), time.time() - t0


def test_world_size_and_gpus_flag_must_agree():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    rc, out, err, _ = _bench(['--gpus', '1', '--steps', '1', '--warmup', '0'], env)
    assert rc == 4 and 'WORLD_SIZE=2' in err and not out.strip()


def test_self_launch_relays_failure_and_leaves_nobody_behind():
    """No GPU here: both ranks fail in drp_create (there is no CPU fallback), the parent must notice, end the
    group and exit non-zero -- not hang in a rendezvous, not print a line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('the failure path needs a box without a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    rc, out, err, dt = _bench(['--gpus', '2', '--share-gpu', '--comm', 'gloo', '--steps', '1', '--warmup', '0',
                               '--timeout', '120'], env, timeout=200)
    assert rc not in (0, 124), (rc, err[-500:])
    assert 'exited with status' in err
    assert not [l for l in out.splitlines() if l.startswith('{')]
    assert dt < 120


def test_more_gpus_asked_for_than_the_box_has_fails_at_once():
    """`bench.py --gpus N` on a box with fewer GPUs (this container shows none) must not start ranks that then wait out a
    rendezvous: exit 4 and one line that says why, within seconds -- from the self-launching parent, and from a rank a
    launcher started (WORLD_SIZE in the environment).  --share-gpu lifts the check."""
    import bench
    have = bench.visible_gpu_count()
    if have is not None and have >= 8:
        import pytest
        pytest.skip('this box has 8 GPUs')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    rc, out, err, dt = _bench(['--gpus', '8', '--steps', '1', '--warmup', '0'], env, timeout=60)
    assert rc == 4 and '--gpus 8' in err and 'not starting' in err and not out.strip(), (rc, err[-300:])
    assert dt < 30
    env_l = dict(env, WORLD_SIZE='8', RANK='3', LOCAL_RANK='3', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    rc, out, err, dt = _bench(['--gpus', '8', '--steps', '1', '--warmup', '0'], env_l, timeout=60)
    assert rc == 4 and 'not starting' in err and not out.strip() and dt < 30


def test_visible_gpu_count_follows_the_visibility_variables(monkeypatch):
    import bench
    have = bench.visible_gpu_count()
    if have is None or have == 0:
        import pytest
        pytest.skip('no KFD topology with GPUs here')
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0')
    assert bench.visible_gpu_count() == 1


def test_the_planners_result_slots_are_the_librarys():
    """planners.GD_SLOTS mirrors include/drp.h's DRP_GD_SLOTS (drp_gd_step_async refuses slots beyond it), and the planner
    never has more iterations in flight than there are slots."""
    import os
    import re
    from dyn_res_pile_manip_amd import planners
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'drp.h')).read()
    m = re.search(r'#define\s+DRP_GD_SLOTS\s+(\d+)', header)
    assert m and int(m.group(1)) == planners.GD_SLOTS
    assert 0 < planners.GD_AHEAD < planners.GD_SLOTS


def test_channel_extrema_of_the_observation_checks():
    """utils.obs2ptcl_fixed_num_batch keeps the reference's range asserts (env/flex_env.py:903-909) but takes every channel's
    extrema in two passes over the contiguous image: same numbers as one reduction per strided channel view."""
    import numpy as np
    from dyn_res_pile_manip_amd.utils import _channel_extrema
    rng = np.random.default_rng(0)
    for shape in ((720, 720, 5), (7, 9, 5), (64, 1, 5), (65, 1, 5), (1, 1, 5)):
        x = rng.normal(size=shape).astype(np.float32)
        mx, mn = _channel_extrema(x)
        np.testing.assert_array_equal(mx, x.reshape(-1, 5).max(0))
        np.testing.assert_array_equal(mn, x.reshape(-1, 5).min(0))
    view = rng.normal(size=(40, 40, 10)).astype(np.float32)[..., ::2]           # not contiguous: copied, same answer
    mx, mn = _channel_extrema(view)
    np.testing.assert_array_equal(mx, view.reshape(-1, 5).max(0))


def test_a_named_goal_skips_the_image_on_a_hit():
    """PlannerGD._set_goal with goal_key: the second call of the same name neither copies nor hashes the image nor calls the
    engine; another name, another pile size or another transform installs again (no GPU: a recording stand-in engine)."""
    import numpy as np
    from dyn_res_pile_manip_amd import flex_rewards, synthetic as syn
    from dyn_res_pile_manip_amd.planners import PlannerGD

    class Eng(object):
        calls = 0

        def set_goal_image(self, g, max_goal_pts, fps_init, mode):
            Eng.calls += 1
    config = syn.default_config()
    planner = PlannerGD(config, syn.SyntheticEnv(config))
    eng = Eng()

    class Boom(object):
        """an 'image' that fails on any access: a hit must not touch it"""
        def __array__(self, *a, **k):
            raise AssertionError('the image was read on a cache hit')
    img = np.zeros((8, 8), np.float32)
    planner._set_goal(eng, img, None, max_goal_pts=100, goal_key='A')
    assert Eng.calls == 1
    planner._set_goal(eng, Boom(), None, max_goal_pts=100, goal_key='A')
    assert Eng.calls == 1
    planner._set_goal(eng, img, None, max_goal_pts=100, goal_key='B')
    planner._set_goal(eng, img, None, max_goal_pts=250, goal_key='B')
    assert Eng.calls == 3
    old = flex_rewards.DIST_TRANSFORM
    try:
        flex_rewards.DIST_TRANSFORM = 'exact' if old != 'exact' else 'cv5'
        planner._set_goal(eng, img, None, max_goal_pts=250, goal_key='B')
    finally:
        flex_rewards.DIST_TRANSFORM = old
    assert Eng.calls == 4
    planner._set_goal(Eng(), img, None, max_goal_pts=250, goal_key='B')          # another engine object
    assert Eng.calls == 5
    # without a name: the content digest, as before
    planner._set_goal(eng, img, None, max_goal_pts=100)
    planner._set_goal(eng, img.copy(), None, max_goal_pts=100)
    assert Eng.calls == 6
    planner._set_goal(eng, img + 1, None, max_goal_pts=100)
    assert Eng.calls == 7


def test_the_episode_generator_is_a_function_of_its_seed():
    """synthetic.push_episode / push_batch feed the reference's training loop (tests/golden/make_golden_trained.py) and the
    device trainer's test with the SAME floats: seeded, shaped like ParticleDataset.__getitem__ / collate_fn."""
    import numpy as np
    from dyn_res_pile_manip_amd import synthetic as syn
    a, b = syn.push_episode(37, 5, 11), syn.push_episode(37, 5, 11)
    for x, y in zip(a[:3], b[:3]):
        np.testing.assert_array_equal(x, y)
    assert a[0].shape == (6, 37, 3) and a[1].shape == (5, 37, 3) and a[2].shape == (6, 37) and a[3] == 37 and 15.0 <= a[4] <= 6500.0
    assert not np.array_equal(a[0], syn.push_episode(37, 5, 12)[0])
    moved = np.abs(a[1]).sum(-1) > 0
    assert moved.any() and not moved.all()                       # a push moves the particles in its band, not the pile
    np.testing.assert_allclose(a[0][1:][moved] - a[0][:-1][moved], a[1][moved], atol=0.08)   # ... to the push's end, then they spread
    st, sd, at, pn, de = syn.push_batch(3, 4, 5)
    assert st.shape[0] == 4 and st.shape[1] == 6 and st.shape[2] == pn.max() and sd.shape == (4, 5, pn.max(), 3)
    for j in range(4):
        assert (st[j, :, pn[j]:] == 0).all() and (sd[j, :, pn[j]:] == 0).all()
