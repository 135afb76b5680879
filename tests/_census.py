"""The parity census (tests/golden/make_golden_census.py -> census.npz) on the device: helpers shared by
tests/test_gpu_census.py and tests/test_oracle_census.py.

Every quantity is per ROW (one unfiltered 10-step push sequence) and STEP:
  margin      the reference trajectory's distance from a graph decision changing (fixture)
  flips_dev   receivers whose sender list on the device differs from the reference's OWN list of that step (hashes of the
              reference's Rr / Rs in the fixture; the device's lists from its own previous state through drp_gen_s_delta +
              drp_build_graph, the kernels the rollout's lists are bit-equal to)
  dev         max |device - reference| over the row's particles
  twin_*      the same three for the reference's own second and third runs, started one ulp up / one ulp down ([2, B, H])

TAU: a step is NEAR A TIE when `margin` is below it.  5e-8 -- the margin make_golden_trained.py demands of every step of the
rows it KEEPS at 300 particles -- is a hundred ulps of adj_thresh^2 = 0.0064 in fp32 (4.66e-10 each): the decisions of
model/gnn_dyn.py:231-237 (`topk` on fp32 squared distances, `dis - thr < 0`) that rounding-level differences of the positions
decide.  Observed: the reference's own twins change lists at margins up to 2.7e-9, the device up to 6.5e-9 on the census rows
and 2.8e-8 on one of the 4 096 MPPI rows (late in the rollout, where ten steps of a chaotic map have grown the deviation to 1e-5).
"""
import numpy as np

SIZES = ['n20', 'n50', 'n100', 'n300', 'n600']
TAU = 5e-8
TAU_MASK = 1e-6      # camera-frame units: a particle this close to an end of the push band (planners.py:248, 0 < u < L) -- the other
                     # discontinuity of a step; the smallest such distance in the fixture is 5.6e-8 (600 particles), no run flipped one


def near_tie(g, p):
    """[B,H] bool: the step's graph (margin) or push mask (mask_margin) is within rounding-level reach of changing."""
    return (g[p + 'margin'] < TAU) | (g[p + 'mask_margin'] < TAU_MASK)


def fmix32(x):
    """murmur3's 32-bit finaliser (make_golden_census.fmix32)."""
    x = np.atleast_1d(np.asarray(x)).astype(np.uint32)
    with np.errstate(over='ignore'):
        x ^= x >> np.uint32(16)
        x = (x * np.uint32(0x85ebca6b)).astype(np.uint32)
        x ^= x >> np.uint32(13)
        x = (x * np.uint32(0xc2b2ae35)).astype(np.uint32)
        x ^= x >> np.uint32(16)
    return x


def list_hash(idx, cnt):
    """Receiver-major lists idx [B,N,10] (int16, -1 padded), cnt [B,N] -> [B,N] uint32: wrapped sum of fmix32(sender + 1)."""
    valid = np.arange(idx.shape[2])[None, None, :] < cnt[:, :, None].astype(np.int64)
    term = np.where(valid, fmix32(idx.astype(np.int64) + 1), np.uint32(0)).astype(np.uint32)
    return term.sum(2, dtype=np.uint32)


def row_hash(recv_hash):
    """[..., N] receiver hashes -> [...] uint32: one word for a whole row's lists (make_golden_census.row_hash)."""
    n = recv_hash.shape[-1]
    salt = fmix32(np.arange(1, n + 1))
    return fmix32(recv_hash ^ salt).sum(-1, dtype=np.uint32)


def device_rows(eng, g, p, acts=None, row_hashes=None):
    """Roll the census rows of group p out on the device -> states [B,H,N,3], rewards [B,H], flips_dev [B,H] (receivers
    whose list differs from the reference's; with `row_hashes` [B,H] -- the MPPI population, other pushes on the group's pile
    -- 0 / 1 per row and step), dev [B,H] (None for the MPPI population: no reference states in the fixture)."""
    s0, attr, dens = g[p + 's_cur'], g[p + 'attr'], g[p + 'dens']
    own = acts is None                     # other push sequences on the group's pile (the MPPI population): no lists, no states to compare
    acts = g[p + 'act_seqs'] if own else acts
    states, rew = eng.rollout(s0, attr, dens, acts, want_reward=True)
    B, H = acts.shape[:2]
    flips = None
    if own or row_hashes is not None:
        ref_h = g[p + 'recv_hash'] if own else None
        flips = np.zeros((B, H), np.int64)
        prev = np.tile(s0, (B, 1, 1))
        for t in range(H):
            sd = eng.gen_s_delta(prev, acts[:, t])
            idx, cnt = eng.build_graph(prev, sd)
            h = list_hash(idx, cnt)
            flips[:, t] = (h != ref_h[:, t]).sum(1) if own else (row_hash(h) != row_hashes[:, t])
            prev = states[:, t]
    dev = None
    if own and (p + 'state_pred') in g.files:
        dev = np.abs(states - g[p + 'state_pred']).max((2, 3)).astype(np.float64)
    return states, rew, flips, dev


def first_true(mask):
    """[B,H] bool -> [B] index of the first True per row, H where none."""
    H = mask.shape[1]
    return np.where(mask.any(1), mask.argmax(1), H)


def displacement(g, p):
    """max |ref[t] - ref[t-1]| over the particles of a row: [B,H]."""
    ref = g[p + 'state_pred']
    B = ref.shape[0]
    prev = np.concatenate([np.tile(g[p + 's_cur'], (B, 1, 1))[:, None], ref[:, :-1]], 1)
    return np.abs(ref - prev).max((2, 3))
