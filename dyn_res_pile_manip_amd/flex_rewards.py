"""Host-side mirror of `config_reward_ptcl` (env/flex_rewards.py:156-214)."""
import numpy as np

# Which transform stands for cv2.distanceTransform(., cv2.DIST_L2, 5) at env/flex_rewards.py:174:
# 'cv5' = OpenCV's 5x5 chamfer (what the reference runs), 'exact' = Euclidean.
DIST_TRANSFORM = 'cv5'


def goal_field(goal, engine=None, mode=None):
    """env/flex_rewards.py:172-177: G = goal - distanceTransform(goal < 0.5), shifted to
    min 0, computed on the device (it also becomes the engine's current field).  engine None: the process's default
    context (engine.default_engine) -- there is no CPU path."""
    from .engine import default_engine
    engine = engine if engine is not None else default_engine()
    field, _ = engine.set_goal_image(np.asarray(goal, dtype=np.float32), 1, 0, mode or DIST_TRANSFORM, want=True)
    return field


def config_reward_ptcl(state, goal, cam_params, goal_coor, normalize=True, offset=(0., 0.),
                       engine=None, field=None):
    """env/flex_rewards.py:156-214 with the reference's argument list (env/flex_env.py:1032-1036,1102 calls it as
    `config_reward_ptcl(state, goal, cam_params=..., goal_coor=..., normalize=True)`): state (B,N,3), goal (H,W) distance
    image, goal_coor (M,2) (col,row) -> (B,) reward.  Two arguments of this build's own, both optional: `engine`, the
    context to run on (default: the process's one context, engine.default_engine -- the one the model and the planner
    bound to it use), and `field`, an already-built G (skips the transform).  `cam_params` [fx, fy, cx, cy] are installed
    on every call."""
    from .gnn_dyn import _to_np, _like
    from .engine import default_engine
    engine = engine if engine is not None else default_engine()
    if tuple(offset) != (0, 0) and tuple(offset) != (0., 0.):
        raise NotImplementedError('pixel offsets are only used by the real-robot path')
    st, proto = _to_np(state)
    g, _ = _to_np(goal)
    gc, _ = _to_np(goal_coor)
    engine.set_camera_intrinsics(np.asarray(cam_params, dtype=np.float32))
    engine.set_goal(field if field is not None else goal_field(g, engine), gc)
    return _like(engine.reward(st, normalize=normalize), proto)
