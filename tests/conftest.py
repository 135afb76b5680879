import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    class G(object):
        def __getattr__(self, name):
            return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return G()


@pytest.fixture
def exact_goal_transform():
    """The reward / planner fixtures were captured from the reference with cv2.distanceTransform
    stubbed by SciPy's exact transform (tests/golden/make_golden.py); comparisons against them
    select the same transform instead of the default OpenCV chamfer."""
    from dyn_res_pile_manip_amd import flex_rewards
    old = flex_rewards.DIST_TRANSFORM
    flex_rewards.DIST_TRANSFORM = 'exact'
    yield
    flex_rewards.DIST_TRANSFORM = old
