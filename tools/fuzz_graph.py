"""Fuzz the three neighbour-list kernels against each other: plain sweep (k_graph), x strips (k_graph_strips),
two-dimensional cells (k_graph_cells, forced for every size, random band height and first halo) -- random particle
counts, batch sizes (so that the rounded-up grids of spread_item() have idle workgroups), pile shapes (uniform, blob,
a line along x, a line along y, clusters of coincident particles, lattice = exact distance ties), scales and radii.
Lists must agree bit for bit.   python tools/fuzz_graph.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.planners import world2cam_affine


def engine(env, radius):
    from dyn_res_pile_manip_amd.engine import Engine
    for k in ('DRP_NO_GRAPH_STRIPS', 'DRP_NO_GRAPH_CELLS', 'DRP_GRAPH_CELLS_MIN_N', 'DRP_GRAPH_CELLS_HB', 'DRP_GRAPH_CELLS_HALO'):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(0)
    e.load_weights(BLOB, radius)
    e.set_camera(M34, 24.0, syn.demo_cam_params())
    return e


def pile(rng, N, kind):
    s = np.empty((N, 3), np.float64)
    if kind == 'uniform':
        s[:, :2] = rng.uniform(-0.2, 0.2, (N, 2))
    elif kind == 'blob':
        r = 0.12 * np.sqrt(rng.uniform(0, 1, N)); th = rng.uniform(0, 2 * np.pi, N)
        s[:, 0] = r * np.cos(th); s[:, 1] = r * np.sin(th)
    elif kind == 'xline':
        s[:, 0] = rng.uniform(-0.3, 0.3, N); s[:, 1] = rng.normal(0, 0.003, N)
    elif kind == 'yline':
        s[:, 1] = rng.uniform(-0.3, 0.3, N); s[:, 0] = rng.normal(0, 0.003, N)
    elif kind == 'dupes':
        c = rng.uniform(-0.2, 0.2, (max(N // 12, 1), 2))
        s[:, :2] = c[rng.integers(0, len(c), N)]
    elif kind == 'lattice':
        m = int(np.ceil(np.sqrt(N)))
        g = np.stack(np.meshgrid(np.arange(m), np.arange(m)), -1).reshape(-1, 2)[:N]
        s[:, :2] = (g - m / 2) * (0.4 / m)
    s[:, 2] = 0.75 - (rng.uniform(0, 0.01, N) if kind not in ('dupes', 'lattice') else 0.0)
    return s


cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
BLOB = weights.blob_from_state_dict(weights.random_state_dict(seed=0))
M34 = world2cam_affine(syn.demo_cam_extrinsics())
bad = 0
for c in range(cases):
    N = int(rng.choice([rng.integers(129, 400), rng.integers(400, 1300), rng.integers(1300, 3000)], p=[0.4, 0.4, 0.2]))
    B = int(rng.integers(1, 12))
    kind = str(rng.choice(['uniform', 'blob', 'xline', 'yline', 'dupes', 'lattice']))
    scale = float(rng.choice([0.3, 1.0, 1.0, 2.5]))
    radius = float(rng.choice([0.02, 0.08, 0.08, 0.3]))
    hb = float(rng.choice([0.02, 0.035, 0.05, 0.09, 0.2, 0.7]))
    halo = float(rng.choice([0.005, 0.02, 0.04, 0.1]))
    s = np.stack([pile(rng, N, kind) for _ in range(B)]).astype(np.float32)
    s[..., :2] *= scale
    sd = np.zeros_like(s) if kind in ('dupes', 'lattice') else (0.004 * rng.standard_normal(s.shape)).astype(np.float32)
    ref = None
    for name, env in (('plain', {'DRP_NO_GRAPH_STRIPS': '1'}), ('strips', {'DRP_NO_GRAPH_CELLS': '1'}),
                      ('cells', {'DRP_GRAPH_CELLS_MIN_N': '1', 'DRP_GRAPH_CELLS_HB': repr(hb), 'DRP_GRAPH_CELLS_HALO': repr(halo)})):
        e = engine(env, radius)
        idx, cnt = e.build_graph(s, sd)
        e.close()
        if ref is None:
            ref = (idx, cnt)
        elif not (np.array_equal(idx, ref[0]) and np.array_equal(cnt, ref[1])):
            bad += 1
            print('MISMATCH %s: case %d N=%d B=%d %s scale=%g radius=%g hb=%g halo=%g: %d rows differ' %
                  (name, c, N, B, kind, scale, radius, hb, halo, int((idx != ref[0]).any(-1).sum())))
print('%d cases, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)
