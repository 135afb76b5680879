"""GPU: particle extraction from the depth image (row f2) through the C ABI, against the vectors
captured from the reference's utils.py / FlexEnv.obs2ptcl_fixed_num_batch and against the oracle.
Everything on this path is float64 / integer work with a fixed evaluation order: the bar is
bit-exact."""
import time

import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn
from dyn_res_pile_manip_amd import utils as dev
from dyn_res_pile_manip_amd._lib import DrpError
from oracle import particles as orc

pytestmark = pytest.mark.gpu
CASES = ['small', 'mid']


@pytest.fixture(scope='module')
def gold(golden):
    g = golden.particles
    return {k: g[k] for k in g.files}


@pytest.fixture(scope='module')
def eng():
    return dev.get_engine()


@pytest.mark.parametrize('name', CASES)
def test_depth2fgpcd(gold, name):
    depth = gold[name + '/depth_raw'] / np.float32(24)
    fg = dev.depth2fgpcd(depth, depth < 0.599 / 0.8, gold[name + '/cam'])
    assert fg.dtype == np.float64
    np.testing.assert_array_equal(fg, gold[name + '/fgpcd'])
    # mask=None selects the rule of env/flex_env.py:945 on the device
    np.testing.assert_array_equal(dev.depth2fgpcd(depth, None, gold[name + '/cam']), fg)
    # an arbitrary mask, and non-positive depths are dropped (utils.py:496)
    rng = np.random.default_rng(0)
    m = rng.uniform(size=depth.shape) < 0.3
    d2 = depth.copy()
    d2[::7, ::5] = 0.0
    np.testing.assert_array_equal(dev.depth2fgpcd(d2, m, gold[name + '/cam']),
                                  orc.depth2fgpcd(d2, m, gold[name + '/cam']))
    assert dev.depth2fgpcd(d2, np.zeros_like(m), gold[name + '/cam']).shape == (0, 3)


@pytest.mark.parametrize('name', CASES)
def test_downsample(gold, name):
    down = dev.downsample_pcd(gold[name + '/fgpcd'], 0.01)
    np.testing.assert_array_equal(down, gold[name + '/down'])
    for voxel in (0.004, 0.03):
        np.testing.assert_array_equal(dev.downsample_pcd(gold[name + '/fgpcd'], voxel),
                                      orc.downsample_pcd(gold[name + '/fgpcd'], voxel))


def test_downsample_shuffled_points_same_voxels(gold):
    fg = gold['small/fgpcd']
    perm = np.random.default_rng(1).permutation(fg.shape[0])
    a = dev.downsample_pcd(fg, 0.01)
    b = dev.downsample_pcd(fg[perm], 0.01)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-14)     # same voxels, sums in another order
    np.testing.assert_array_equal(b, orc.downsample_pcd(fg[perm], 0.01))


@pytest.mark.parametrize('name', CASES)
def test_fps_and_radius(gold, name):
    pts, r = dev.fps(gold[name + '/down'], int(gold[name + '/n_ptcl']), int(gold[name + '/fps_start']))
    assert pts.dtype == np.float32
    np.testing.assert_array_equal(pts, gold[name + '/fps_pts'])
    assert r == float(gold[name + '/fps_r'])


def test_fps_batch_of_starts(gold, eng):
    down = gold['mid/down']
    starts = [0, 5, down.shape[0] - 1, 77]
    pts, r = eng.fps_pcd(down, 64, starts)
    for b, s in enumerate(starts):
        want, want_r = orc.fps(down, 64, s)
        np.testing.assert_array_equal(pts[b], want)
        assert r[b] == want_r
    one, _ = eng.fps_pcd(down, 1, [9])
    np.testing.assert_array_equal(one[0, 0], down[9].astype(np.float32))
    with pytest.raises(DrpError):
        eng.fps_pcd(down, 64, [down.shape[0]])
    with pytest.raises(DrpError):
        eng.fps_pcd(down[:10], 11, [0])


@pytest.mark.parametrize('name', CASES)
def test_recenter(gold, name):
    r = float(gold[name + '/fps_r'])
    rec = dev.recenter(gold[name + '/down'], gold[name + '/fps_pts'], r=min(0.02, 0.5 * r))
    assert rec.dtype == np.float32
    np.testing.assert_array_equal(rec, gold[name + '/recenter'])


def test_recenter_empty_ball_is_nan(gold):
    far = np.array([[5.0, 5.0, 5.0]], np.float32)
    out = dev.recenter(gold['small/down'], far, r=0.01)
    assert np.isnan(out).all()


@pytest.mark.parametrize('name', CASES)
def test_obs2ptcl_batch_matches_reference(gold, name, eng):
    depth_raw = gold[name + '/depth_raw']
    ptcl, r, (nfg, nd) = eng.obs2ptcl(depth_raw, 24, gold[name + '/cam'], int(gold[name + '/n_ptcl']),
                                     gold[name + '/batch_start'].shape[0], init_idx=gold[name + '/batch_start'])
    assert nfg == gold[name + '/fgpcd'].shape[0] and nd == gold[name + '/down'].shape[0]
    np.testing.assert_array_equal(ptcl, gold[name + '/batch_ptcl'])
    np.testing.assert_array_equal(r, gold[name + '/batch_r'])


def test_obs2ptcl_full_size_against_oracle(eng):
    """The shape the environment runs: 720 x 720 depth image, 300 particles, batch 30."""
    obs = syn.render_depth(5000, seed=2, kind='uniform')
    cam = syn.demo_cam_params()
    depth_raw = obs[..., -1]
    starts = np.random.default_rng(4).integers(0, 1000, 30) % 900
    t0 = time.perf_counter()
    ptcl, r, (nfg, nd) = eng.obs2ptcl(depth_raw, 24.0, cam, 300, 30, init_idx=starts)
    t_dev = time.perf_counter() - t0
    want, want_r, fg = orc.obs2ptcl_fixed_num_batch(depth_raw, 24.0, cam, 300, starts[:4])
    assert nd == fg.shape[0] and nd >= 900
    np.testing.assert_array_equal(ptcl[:4], want)
    np.testing.assert_array_equal(r[:4], want_r)
    assert np.isfinite(ptcl).all() and (r > 0).all()
    # every particle lies inside the pile's bounding box, and the samples cover the cloud to r
    assert (ptcl.min(axis=(0, 1)) >= fg.min(0) - 1e-6).all() and (ptcl.max(axis=(0, 1)) <= fg.max(0) + 1e-6).all()
    print('obs2ptcl 720x720 -> %d fg -> %d voxels -> 30 x 300 particles: %.1f ms' % (nfg, nd, t_dev * 1e3))


def test_host_mirror_seeded_starts(eng):
    obs = syn.render_depth(1200, seed=5, kind='uniform')
    cam = syn.demo_cam_params()
    np.random.seed(3)
    a, ra = dev.obs2ptcl_fixed_num_batch(obs, 100, 6, cam, 24.0)
    np.random.seed(3)
    b, rb = dev.obs2ptcl_fixed_num_batch(obs, 100, 6, cam, 24.0)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ra, rb)
    assert a.shape == (6, 100, 3) and a.dtype == np.float64
    assert len({tuple(x) for x in a[:, 0]}) > 1            # different starts across the batch
    p1, r1 = dev.obs2ptcl_fixed_num(obs, 100, cam, 24.0, init_idx=7)
    assert p1.shape == (100, 3) and p1.dtype == np.float32 and r1 > 0
    # density used by the planner: 1 / r^2 (env/flex_env.py:1022); coarser resolution -> larger r
    _, r_coarse = dev.obs2ptcl_fixed_num(obs, 30, cam, 24.0, init_idx=7)
    assert r_coarse > r1
    with pytest.raises(AssertionError):
        dev.obs2ptcl_fixed_num_batch(obs[..., :4], 100, 6, cam, 24.0)
    blank = obs.copy()
    blank[..., -1] = 18.0
    with pytest.raises(DrpError):
        dev.obs2ptcl_fixed_num_batch(blank, 100, 2, cam, 24.0)


def test_edge_cases_small_clouds(eng):
    """Degenerate shapes the environment can hand over: one-voxel clouds, as many particles as
    voxels, one particle, a radius that swallows the whole cloud."""
    rng = np.random.default_rng(0)
    # every point in one voxel -> one output point = their mean (summed in index order)
    blob = 0.5 + 1e-3 * rng.uniform(size=(37, 3))
    down = dev.downsample_pcd(blob, 0.01)
    np.testing.assert_array_equal(down, orc.downsample_pcd(blob, 0.01))
    assert down.shape == (1, 3)
    one = dev.downsample_pcd(blob[:1], 0.01)
    np.testing.assert_array_equal(one, blob[:1])
    # as many samples as points: a permutation of the cloud, radius 0 up to float32 rounding of the samples
    cloud = rng.uniform(-0.2, 0.2, (50, 3)) + [0, 0, 0.7]
    pts, r = eng.fps_pcd(cloud, 50, [3])
    want, want_r = orc.fps(cloud, 50, 3)
    np.testing.assert_array_equal(pts[0], want)
    assert r[0] == want_r and r[0] < 1e-7
    assert len({tuple(p) for p in pts[0]}) == 50
    p1, r1 = eng.fps_pcd(cloud, 1, [7])
    w1, wr1 = orc.fps(cloud, 1, 7)
    np.testing.assert_array_equal(p1[0], w1)
    assert r1[0] == wr1
    # recentering radius larger than the cloud: every sample becomes the cloud's mean
    rec = dev.recenter(cloud, pts[0][:5], r=10.0)
    np.testing.assert_array_equal(rec, orc.recenter(cloud, pts[0][:5], r=10.0))
    assert np.abs(rec - cloud.mean(0).astype(np.float32)).max() < 1e-6
    # a 1 x W and an H x 1 depth image
    cam = [100.0, 100.0, 3.0, 0.0]
    row = np.array([[0.5, 0.9, 0.0, 0.6, 0.7, -1.0, 0.3]], np.float32)
    for img in (row, row.T.copy()):
        np.testing.assert_array_equal(dev.depth2fgpcd(img, img < 0.74875, cam), orc.depth2fgpcd(img, img < np.float32(0.74875), cam))


@pytest.mark.parametrize('name', CASES)
def test_fps_rad(gold, name, eng):
    fg = gold[name + '/fgpcd']
    radius, start = float(gold[name + '/fps_rad_radius']), int(gold[name + '/fps_rad_start'])
    pts, idx = eng.fps_rad(fg, radius, start)
    np.testing.assert_array_equal(pts, gold[name + '/fps_rad_pts'])
    assert idx[0] == start and len(set(idx.tolist())) == idx.shape[0]
    # every cloud point ends within the radius of a sample; a capped call stops early
    d = np.linalg.norm(fg[:, None, :] - pts[None, ::1, :], axis=2).min(1)
    assert d.max() <= radius
    few, _ = eng.fps_rad(fg, radius, start, cap=5)
    np.testing.assert_array_equal(few, pts[:5])
    np.random.seed(21)
    np.testing.assert_array_equal(dev.fps_rad(fg, radius), pts)
