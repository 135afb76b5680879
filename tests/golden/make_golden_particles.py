#!/usr/bin/env python3
"""Golden vectors for the particle-extraction row, captured by importing
the reference's `utils.py` and `env/flex_env.py` in the build container.

Third-party calls the reference makes on this path whose packages are absent here
(`open3d.voxel_down_sample`, `dgl.geometry.farthest_point_sampler`) are served by the
restatements in `oracle/particles.py`, with the sampler's random start recorded; every line
of the reference's own code around them (depth2fgpcd, fps's gather and radius, recenter,
obs2ptcl_fixed_num_batch's loop and dtypes) runs as written.  Usage:
    python tests/golden/make_golden_particles.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402


def load_reference():
    import torch
    from oracle import particles as orc
    mg.install_shims()
    starts = []

    def farthest_point_sampler(pos, npoints, start_idx=None):
        assert pos.shape[0] == 1
        n = pos.shape[1]
        if start_idx is None:
            start_idx = int(np.random.randint(n))
        starts.append(int(start_idx))
        idx = orc.farthest_point_sampler(pos[0].numpy(), npoints, start_idx)
        return torch.from_numpy(idx)[None]
    sys.modules['dgl.geometry'].farthest_point_sampler = farthest_point_sampler

    o3d = sys.modules['open3d']
    o3d.geometry = types.SimpleNamespace()
    o3d.utility = types.SimpleNamespace()

    class PointCloud(object):
        def __init__(self):
            self.points = None

        def voxel_down_sample(self, voxel_size):
            out = PointCloud()
            out.points = orc.downsample_pcd(np.asarray(self.points), voxel_size)
            return out
    o3d.geometry.PointCloud = PointCloud
    o3d.utility.Vector3dVector = lambda a: np.asarray(a, dtype=np.float64)

    for name in ('gym', 'pyflex', 'pybullet', 'pybullet_data', 'bs4', 'torchvision', 'torchvision.models', 'skopt'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['gym'].Env = object
    sys.modules['bs4'].BeautifulSoup = None
    sys.path.insert(0, mg.REF)
    import utils as ref_utils
    from env.flex_env import FlexEnv
    return ref_utils, FlexEnv, starts


class FakeSelf(object):
    def __init__(self, cam, gs):
        self.cam = cam
        self.global_scale = gs

    def get_cam_params(self):
        return self.cam


def main():
    from dyn_res_pile_manip_amd import synthetic as syn
    ref_utils, FlexEnv, starts = load_reference()
    out = {}
    gs = 24
    cases = [('small', 240, [869.11688245 / 3.0, 869.11688245 / 3.0, 120.0, 120.0], 260, 0.008, 40, 4, 'blob'),
             ('mid', 360, [869.11688245 / 2.0, 869.11688245 / 2.0, 180.0, 180.0], 700, 0.006, 150, 3, 'uniform')]
    for name, size, cam, n_gran, grain, n_ptcl, batch, kind in cases:
        obs = syn.render_depth(n_gran, seed=7, kind=kind, size=size, grain=grain, global_scale=gs, cam_params=cam)
        depth = obs[..., -1] / gs
        fg = ref_utils.depth2fgpcd(depth, depth < 0.599 / 0.8, cam)
        down = ref_utils.downsample_pcd(fg, 0.01)
        np.random.seed(5)
        del starts[:]
        sampled, r = ref_utils.fps(down, n_ptcl)
        rec = ref_utils.recenter(down, sampled, r=min(0.02, 0.5 * r))
        out[name + '/depth_raw'] = obs[..., -1]
        out[name + '/cam'] = np.asarray(cam, dtype=np.float64)
        out[name + '/n_ptcl'] = np.array(n_ptcl)
        out[name + '/fgpcd'] = fg
        out[name + '/down'] = down
        out[name + '/fps_start'] = np.array(starts[0])
        out[name + '/fps_pts'] = sampled
        out[name + '/fps_r'] = np.array(r)
        out[name + '/recenter'] = rec
        # the env method itself (env/flex_env.py:933-951)
        np.random.seed(11)
        del starts[:]
        b_ptcl, b_r = FlexEnv.obs2ptcl_fixed_num_batch(FakeSelf(cam, gs), obs, n_ptcl, batch)
        out[name + '/batch_start'] = np.array(starts)
        out[name + '/batch_ptcl'] = b_ptcl
        out[name + '/batch_r'] = b_r
        print(name, 'fg', fg.shape, 'down', down.shape, 'r', r, b_ptcl.dtype, rec.dtype)
    # utils.fps_rad (utils.py:438-449), the dataset's radius-terminated sampler, on the raw foreground cloud
    for name, radius in (('small', 0.03), ('mid', 0.045)):
        fg = out[name + '/fgpcd']
        np.random.seed(21)
        start = int(np.random.randint(fg.shape[0]))          # what fps_rad draws first
        np.random.seed(21)
        sel = ref_utils.fps_rad(fg, radius)
        assert np.array_equal(sel[0], fg[start])
        out[name + '/fps_rad_radius'] = np.array(radius)
        out[name + '/fps_rad_start'] = np.array(start)
        out[name + '/fps_rad_pts'] = sel
        print(name, 'fps_rad', radius, '->', sel.shape[0], 'points')
    # utils.fps_np (utils.py:451-466), the numpy sampler used for the goal pixels
    rng = np.random.default_rng(3)
    pts2 = rng.integers(0, 720, (2000, 2)).astype(np.float32)
    sel, md = ref_utils.fps_np(pts2, 50, 3)
    out['fpsnp/pts'] = pts2
    out['fpsnp/sel'] = sel
    out['fpsnp/max_dist'] = np.array(md)
    np.savez_compressed(os.path.join(HERE, 'particles.npz'), **out)
    print('particles.npz %.1f KB' % (os.path.getsize(os.path.join(HERE, 'particles.npz')) / 1024.0))


if __name__ == '__main__':
    main()
