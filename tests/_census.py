"""The parity census (tests/golden/make_golden_census.py -> census.npz) on the device: helpers shared by
tests/test_gpu_census.py and tools/census_report.py.

Every quantity is per ROW (one unfiltered 10-step push sequence) and STEP:
  margin      the reference trajectory's distance from a graph decision changing (fixture)
  flips_dev   receivers whose sender list on the device differs from the reference's OWN list of that step (hashes of the
              reference's Rr / Rs in the fixture; the device's lists from its own previous state through drp_gen_s_delta +
              drp_build_graph, the kernels the rollout's lists are bit-equal to)
  dev         max |device - reference| over the row's particles
  twin_*      the same three for the reference's second run, started one ulp away
"""
import numpy as np

SIZES = ['n20', 'n50', 'n100', 'n300']


def fmix32(x):
    """murmur3's 32-bit finaliser (make_golden_census.fmix32)."""
    x = np.asarray(x).astype(np.uint32)
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


def device_rows(eng, g, p, acts=None):
    """Roll the census rows of group p out on the device -> states [B,H,N,3], rewards [B,H], flips_dev [B,H] (None when the
    group has no recv_hash), dev [B,H]."""
    s0, attr, dens = g[p + 's_cur'], g[p + 'attr'], g[p + 'dens']
    own = acts is None                     # other push sequences on the group's pile (the MPPI population): no lists, no states to compare
    acts = g[p + 'act_seqs'] if own else acts
    states, rew = eng.rollout(s0, attr, dens, acts, want_reward=True)
    B, H = acts.shape[:2]
    flips = None
    if own and (p + 'recv_hash') in g.files:
        ref_h = g[p + 'recv_hash']
        flips = np.zeros((B, H), np.int64)
        prev = np.tile(s0, (B, 1, 1))
        for t in range(H):
            sd = eng.gen_s_delta(prev, acts[:, t])
            idx, cnt = eng.build_graph(prev, sd)
            flips[:, t] = (list_hash(idx, cnt) != ref_h[:, t]).sum(1)
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
