"""GPU: the x-strip neighbour build (k_graph_sort + k_graph_strips, samples of more than 128 particles) and the
two-dimensional cell build (k_graph_sort2 + k_graph_cells, forced here for every size and with band heights from a
quarter of the radius to more than the workspace) against
the plain two-sweep kernel (DRP_NO_GRAPH_STRIPS=1) -- the same lists bit for bit, in both emission orders
(ascending index through drp_build_graph; self loop first inside the fused engine's rollouts) -- and against
the reference's own lists through the golden one-step cases elsewhere (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from dyn_res_pile_manip_amd import synthetic as syn, weights
from dyn_res_pile_manip_amd.planners import world2cam_affine

pytestmark = pytest.mark.gpu


def _engines(monkeypatch):
    from dyn_res_pile_manip_amd.engine import Engine
    blob = weights.blob_from_state_dict(weights.random_state_dict(seed=0))
    M34 = world2cam_affine(syn.demo_cam_extrinsics())
    out = {}
    # False: x strips; True: the plain sweep; 'cells', 'cells_lo', 'cells_hi': two-dimensional cells for every size
    # with the default, a 2-cm and a 70-cm band height
    for key, env in ((False, {'DRP_NO_GRAPH_CELLS': '1'}), (True, {'DRP_NO_GRAPH_STRIPS': '1'}),
                     ('cells', {'DRP_GRAPH_CELLS_MIN_N': '1'}),
                     ('cells_lo', {'DRP_GRAPH_CELLS_MIN_N': '1', 'DRP_GRAPH_CELLS_HB': '0.02'}),
                     ('cells_hi', {'DRP_GRAPH_CELLS_MIN_N': '1', 'DRP_GRAPH_CELLS_HB': '0.7'})):
        for k in ('DRP_NO_GRAPH_STRIPS', 'DRP_NO_GRAPH_CELLS', 'DRP_GRAPH_CELLS_MIN_N', 'DRP_GRAPH_CELLS_HB'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(0)
        eng.load_weights(blob, 0.08)
        eng.set_camera(M34, 24.0, syn.demo_cam_params())
        out[key] = eng
    return out


CELLS = ('cells', 'cells_lo', 'cells_hi')


@pytest.mark.parametrize('N,B,kind,scale', [(129, 5, 'uniform', 1.0), (300, 6, 'uniform', 1.0), (300, 4, 'blob', 1.0),
                                            (515, 3, 'uniform', 3.0), (700, 2, 'blob', 0.3), (1200, 2, 'uniform', 1.0),
                                            (2500, 2, 'blob', 1.0), (4096, 1, 'uniform', 1.0)])
def test_lists_equal_plain_sweep(monkeypatch, N, B, kind, scale):
    """Jittered piles: uniform over the workspace, clumped (everything within a radius or two: the strips prune
    nothing), spread out (most strips empty, particles beyond the clamped end strips)."""
    engs = _engines(monkeypatch)
    rng = np.random.default_rng(N)
    s0, _, _ = syn.make_pile(N, 1, seed=N, kind=kind)
    s = np.tile(s0, (B, 1, 1)).astype(np.float32)
    s[..., :2] *= scale
    s += 0.002 * rng.standard_normal(s.shape).astype(np.float32)
    sd = (0.004 * rng.standard_normal(s.shape)).astype(np.float32)
    i0, c0 = engs[False].build_graph(s, sd)
    i1, c1 = engs[True].build_graph(s, sd)
    cells = {k: engs[k].build_graph(s, sd) for k in CELLS}
    for e in engs.values():
        e.close()
    assert np.array_equal(c0, c1)
    assert np.array_equal(i0, i1)
    for k in CELLS:
        assert np.array_equal(cells[k][1], c1), k
        assert np.array_equal(cells[k][0], i1), k
    assert c0.max() == 10 or scale > 1.0


def test_coincident_particles_tie_at_the_cut(monkeypatch):
    """Sixty particles on one spot and a few exact duplicates elsewhere: more senders at exactly the 10th
    distance than slots left -- the lowest indices win in both kernels."""
    engs = _engines(monkeypatch)
    N, B = 260, 3
    s0, _, _ = syn.make_pile(N, 1, seed=5)
    s = np.tile(s0, (B, 1, 1)).astype(np.float32)
    s[:, 40:100] = s[:, 40:41]
    s[1, 200:204] = s[1, 10:11]
    s[2, 150] = s[2, 7]
    sd = np.zeros_like(s)
    i0, c0 = engs[False].build_graph(s, sd)
    i1, c1 = engs[True].build_graph(s, sd)
    cells = {k: engs[k].build_graph(s, sd) for k in CELLS}
    for e in engs.values():
        e.close()
    assert np.array_equal(c0, c1)
    assert np.array_equal(i0, i1)
    for k in CELLS:
        assert np.array_equal(cells[k][1], c1), k
        assert np.array_equal(cells[k][0], i1), k
    assert (i0[0, 40:100, :10] < 100).all() and (i0[0, 40:100, 0] == 40).all()


def test_rollouts_equal_with_self_loop_first(monkeypatch):
    """Inside the fused engine the lists start with the self loop; whole rollouts (pushes included) agree bitwise."""
    engs = _engines(monkeypatch)
    N, ns, H = 300, 40, 3
    s0, dens, attr = syn.make_pile(N, 1, seed=2)
    acts = syn.sample_pushes(ns, H, seed=2)
    r0, _ = engs[False].rollout(s0, attr, dens, acts)
    r1, _ = engs[True].rollout(s0, attr, dens, acts)
    rc, _ = engs['cells'].rollout(s0, attr, dens, acts)
    for e in engs.values():
        e.close()
    assert np.isfinite(r0).all()
    assert np.array_equal(r0, r1)
    assert np.array_equal(rc, r1)


def test_cells_on_the_reference_cases(monkeypatch, golden):
    """The two-dimensional build against the REFERENCE's own lists (tests/golden/one_step.npz), forced for every size."""
    engs = _engines(monkeypatch)
    g = golden.one_step
    try:
        for case in ('n64', 'n50', 'n150', 'n300', 'n600', 'n8', 'blob150'):
            for k in CELLS:
                idx, cnt = engs[k].build_graph(g[case + '/s_cur'], g[case + '/s_delta'])
                np.testing.assert_array_equal(cnt, g[case + '/nbr_cnt'], err_msg='%s %s' % (case, k))
                np.testing.assert_array_equal(idx, g[case + '/nbr_idx'], err_msg='%s %s' % (case, k))
    finally:
        for e in engs.values():
            e.close()


@pytest.mark.parametrize('N,B,kind', [(64, 3, 'uniform'), (129, 2, 'uniform'), (300, 4, 'uniform'), (300, 4, 'blob'), (515, 2, 'lattice'),
                                      (1000, 4, 'padded'), (257, 3, 'dupes'), (4096, 1, 'uniform'), (130, 5, 'padded')])
def test_four_threads_per_receiver_give_the_plain_sweeps_lists(monkeypatch, N, B, kind):
    """k_graph_q4 (a handful of samples: four threads per receiver, each over a quarter of the senders, the partial
    lists merged in LDS) against k_graph, bit for bit, in both emission orders -- jittered piles, a lattice (exact
    distance ties at the cut), clusters of coincident particles, and zero-padded training batches (hundreds of
    coincident padding rows: every one of them ties with every other at distance 0)."""
    from dyn_res_pile_manip_amd.engine import Engine
    rng = np.random.default_rng(N + B)
    if kind == 'lattice':
        g = int(np.ceil(np.sqrt(N)))
        xy = np.stack(np.meshgrid(np.arange(g), np.arange(g)), -1).reshape(-1, 2)[:N].astype(np.float32) * 0.03 - 0.2
        s0 = np.concatenate([xy, np.full((N, 1), 0.75, np.float32)], 1)[None]
    elif kind == 'dupes':
        base, _, _ = syn.make_pile(max(N // 8, 1), 1, seed=N)
        s0 = np.repeat(base, 8, axis=1)[:, :N]
        if s0.shape[1] < N:
            s0 = np.concatenate([s0, s0[:, :N - s0.shape[1]]], 1)
    else:
        s0, _, _ = syn.make_pile(N, 1, seed=N, kind='blob' if kind == 'blob' else 'uniform')
    s = np.tile(s0, (B, 1, 1)).astype(np.float32)
    sd = np.zeros_like(s) if kind in ('lattice', 'dupes') else (0.004 * rng.standard_normal(s.shape)).astype(np.float32)
    if kind == 'padded':
        for b in range(B):
            n_b = int(rng.integers(N // 3, N))
            s[b, n_b:] = 0.0                       # collate_fn's zero rows: coincident particles at the camera origin
            sd[b, n_b:] = 0.0
    blob = weights.blob_from_state_dict(weights.random_state_dict(seed=0))
    M34 = world2cam_affine(syn.demo_cam_extrinsics())
    acts = syn.sample_pushes(B, 1, seed=N)
    res = {}
    for mode in ('0', '2'):
        monkeypatch.setenv('DRP_NO_GRAPH_STRIPS', '1')
        monkeypatch.setenv('DRP_NO_ROLLOUT_FUSED', '1')
        monkeypatch.setenv('DRP_GRAPH_Q4', mode)
        eng = Engine(0)
        eng.load_weights(blob, 0.08)
        eng.set_camera(M34, 24.0, syn.demo_cam_params())
        idx, cnt = eng.build_graph(s, sd)                                  # ascending sender order
        st, _ = eng.rollout(s0.astype(np.float32), np.zeros((1, N), np.float32), np.array([N / 0.16], np.float32), acts)
        res[mode] = (idx, cnt, eng.debug_fetch('nbr_idx', (B, N, 10), np.int16), eng.debug_fetch('nbr_cnt', (B, N), np.uint8),
                     eng.debug_fetch('s_delta', (B, N, 3)), st)           # self loop first, impulses from the push
        eng.close()
    for a, b in zip(res['0'], res['2']):
        assert np.array_equal(a, b)
    assert res['2'][1].max() <= 10 and res['2'][1].min() >= 1
