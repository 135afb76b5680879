"""GPU: goal pre-processing on the device (row f3) through the C ABI -- the distance transform of
config_reward_ptcl (env/flex_rewards.py:172-177) and the goal pixel subsample (planners.py:620-624).
Integer / index work: bit-exact against the oracle."""
import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn
from dyn_res_pile_manip_amd._lib import DrpError
from dyn_res_pile_manip_amd.engine import Engine
from oracle import goal as og
from oracle import particles as op

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    e = Engine(0)
    yield e
    e.close()


def _sources():
    rng = np.random.default_rng(0)
    out = {'I': 1 - syn.goal_mask('I'), 'disc': 1 - syn.goal_mask('disc'),
           'inside_I': syn.goal_mask('I'),
           'sparse': (rng.uniform(size=(97, 211)) < 0.995).astype(np.uint8),
           'wide': (rng.uniform(size=(9, 2500)) < 0.999).astype(np.uint8),     # rows longer than a workgroup
           'tall': (rng.uniform(size=(1300, 7)) < 0.99).astype(np.uint8)}
    one = np.ones((64, 64), np.uint8)
    one[63, 0] = 0
    out['corner'] = one
    return out


@pytest.mark.parametrize('name', sorted(_sources()))
def test_chamfer_equals_opencv_restatement(eng, name):
    src = _sources()[name]
    np.testing.assert_array_equal(eng.distance_transform(src, 'cv5'), og.distance_transform_cv5(src))


@pytest.mark.parametrize('name', sorted(_sources()))
def test_exact_transform(eng, name):
    src = _sources()[name]
    np.testing.assert_array_equal(eng.distance_transform(src, 'exact'), og.distance_transform_edt(src))


@pytest.mark.parametrize('kind', ['I', 'disc'])
@pytest.mark.parametrize('mode', ['cv5', 'exact'])
def test_set_goal_image(eng, kind, mode):
    obs_goal = syn.goal_distance_image(syn.goal_mask(kind))
    N = 64
    field, coor = eng.set_goal_image(obs_goal, 5 * N, fps_init=0, mode=mode, want=True)
    np.testing.assert_array_equal(field, og.goal_field(obs_goal, mode))
    px = og.goal_pixels(obs_goal)
    want, _ = op.fps_np(px, min(5 * N, px.shape[0]), 0)
    np.testing.assert_array_equal(coor, want)
    # the installed constants are the ones the reward kernel uses
    s, _, _ = syn.make_pile(N, n_batch=6, seed=3)
    eng.set_camera(np.eye(4, dtype=np.float32)[:3], 24.0, syn.demo_cam_params())
    r_img = eng.reward(s)
    eng.set_goal(field, coor)
    np.testing.assert_array_equal(r_img, eng.reward(s))
    # another start index, a cap above the number of goal pixels
    _, coor7 = eng.set_goal_image(obs_goal, 10 ** 6, fps_init=7, mode=mode, want=True)
    assert coor7.shape[0] == px.shape[0]
    np.testing.assert_array_equal(coor7[0], px[7])
    assert len({tuple(c) for c in coor7}) == px.shape[0]


def test_set_goal_image_errors(eng):
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    with pytest.raises(DrpError):
        eng.set_goal_image(np.ones((64, 64), np.float32), 10)          # no goal pixel
    with pytest.raises(DrpError):
        eng.set_goal_image(np.zeros((64, 64), np.float32), 10)         # nothing but goal pixels
    with pytest.raises(DrpError):
        eng.set_goal_image(obs_goal, 10, fps_init=10 ** 7)
    with pytest.raises(KeyError):
        eng.set_goal_image(obs_goal, 10, mode='l1')


def test_transform_edge_shapes(eng):
    one_row = np.array([[1, 1, 0, 1, 1, 1, 1, 0, 1]], np.uint8)
    for src in (one_row, one_row.T.copy(), np.zeros((5, 7), np.uint8), np.array([[0]], np.uint8),
                np.pad(np.zeros((1, 1), np.uint8), 3, constant_values=1)):
        np.testing.assert_array_equal(eng.distance_transform(src, 'cv5'), og.distance_transform_cv5(src))
        np.testing.assert_array_equal(eng.distance_transform(src, 'exact'), og.distance_transform_edt(src))
    with pytest.raises(KeyError):
        eng.distance_transform(one_row, 'l1')
