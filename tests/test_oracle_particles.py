"""The particle-extraction oracle (oracle/particles.py) against vectors captured from the
reference's utils.py / FlexEnv.obs2ptcl_fixed_num_batch (tests/golden/make_golden_particles.py).
`down` in the fixture comes from the oracle's own restatement of open3d's voxel_down_sample
(the package is absent), so it pins nothing by itself; the other arrays are the reference's."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import particles as orc  # noqa: E402

CASES = ['small', 'mid']


@pytest.fixture(scope='module')
def gold(golden):
    g = golden.particles
    return {k: g[k] for k in g.files}


@pytest.mark.parametrize('name', CASES)
def test_depth2fgpcd(gold, name):
    depth = gold[name + '/depth_raw'] / np.float32(24)
    fg = orc.depth2fgpcd(depth, depth < np.float32(orc.FG_DEPTH), gold[name + '/cam'])
    assert fg.dtype == np.float64
    np.testing.assert_array_equal(fg, gold[name + '/fgpcd'])


@pytest.mark.parametrize('name', CASES)
def test_downsample_is_voxel_mean(gold, name):
    fg = gold[name + '/fgpcd']
    down = orc.downsample_pcd(fg, orc.VOXEL)
    np.testing.assert_array_equal(down, gold[name + '/down'])
    # properties of open3d's voxel_down_sample: one point per occupied voxel, inside it,
    # count-weighted mean of the outputs = mean of the inputs
    keys = orc.voxel_keys(fg, orc.VOXEL)
    uniq, cnt = np.unique(keys, axis=0, return_counts=True)
    assert down.shape[0] == uniq.shape[0]
    mn = fg.min(axis=0) - orc.VOXEL * 0.5
    kd = np.floor((down - mn) / orc.VOXEL).astype(np.int64)
    np.testing.assert_array_equal(kd, uniq)
    np.testing.assert_allclose((down * cnt[:, None]).sum(0) / cnt.sum(), fg.mean(0), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('name', CASES)
def test_fps_and_radius(gold, name):
    pts, r = orc.fps(gold[name + '/down'], int(gold[name + '/n_ptcl']), int(gold[name + '/fps_start']))
    assert pts.dtype == np.float32
    np.testing.assert_array_equal(pts, gold[name + '/fps_pts'])
    assert r == float(gold[name + '/fps_r'])


@pytest.mark.parametrize('name', CASES)
def test_recenter(gold, name):
    r = float(gold[name + '/fps_r'])
    rec = orc.recenter(gold[name + '/down'], gold[name + '/fps_pts'], r=min(0.02, 0.5 * r))
    assert rec.dtype == np.float32
    np.testing.assert_array_equal(rec, gold[name + '/recenter'])


@pytest.mark.parametrize('name', CASES)
def test_obs2ptcl_batch(gold, name):
    ptcl, rad, _ = orc.obs2ptcl_fixed_num_batch(gold[name + '/depth_raw'], 24, gold[name + '/cam'],
                                                int(gold[name + '/n_ptcl']), gold[name + '/batch_start'])
    np.testing.assert_array_equal(ptcl, gold[name + '/batch_ptcl'])
    np.testing.assert_array_equal(rad, gold[name + '/batch_r'])


def test_fps_np(gold):
    sel, md = orc.fps_np(gold['fpsnp/pts'], 50, 3)
    np.testing.assert_array_equal(sel, gold['fpsnp/sel'])
    assert md == gold['fpsnp/max_dist']


@pytest.mark.parametrize('name', CASES)
def test_fps_rad(gold, name):
    pts, _ = orc.fps_rad(gold[name + '/fgpcd'], float(gold[name + '/fps_rad_radius']), int(gold[name + '/fps_rad_start']))
    np.testing.assert_array_equal(pts, gold[name + '/fps_rad_pts'])
