"""Host-side mirror of `config_reward_ptcl` (env/flex_rewards.py:156-214)."""
import numpy as np

from . import synthetic as _syn


def goal_field(goal):
    """env/flex_rewards.py:172-177: G = goal - distanceTransform(goal < 0.5), shifted to
    min 0.  The reference uses OpenCV's 5x5-mask approximate transform; here SciPy's exact
    Euclidean transform (the one deviation on this path)."""
    return _syn.goal_field(np.asarray(goal, dtype=np.float32))


def config_reward_ptcl(state, goal, cam_params, goal_coor, normalize=True, offset=(0., 0.),
                       engine=None, field=None):
    """state (B,N,3), goal (H,W) distance image, goal_coor (M,2) (col,row) -> (B,) reward.
    `engine` is the Engine to run on; `field` an already-built G (skips the transform)."""
    from .gnn_dyn import _to_np, _like
    if engine is None:
        raise ValueError('config_reward_ptcl needs the Engine to run on (no CPU fallback)')
    if tuple(offset) != (0, 0) and tuple(offset) != (0., 0.):
        raise NotImplementedError('pixel offsets are only used by the real-robot path')
    st, proto = _to_np(state)
    g, _ = _to_np(goal)
    gc, _ = _to_np(goal_coor)
    engine.set_camera_intrinsics(cam_params) if hasattr(engine, 'set_camera_intrinsics') else None
    engine.set_goal(field if field is not None else goal_field(g), gc)
    return _like(engine.reward(st, normalize=normalize), proto)
