"""ORACLE (test infrastructure, not product code): goal pre-processing.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.

    G = goal - cv2.distanceTransform((goal < 0.5).astype(uint8), cv2.DIST_L2, 5)
    G = G - G.min()                                     (env/flex_rewards.py:172-177)
    goal_coor = flip((goal < 0.5).nonzero())  -> fps_np(., min(5N, count))   (planners.py:620-624)

PARITY UNPINNED for `cv2.distanceTransform`: opencv-python is absent here and unpinned in the
reference's env.yaml, and the reference holds no vector for it.  `distance_transform_cv5`
restates OpenCV's published `distanceTransform_5x5` (modules/imgproc/src/distransform.cpp):
a 16.16 fixed-point chamfer with the DIST_L2 5x5 weights (1, 1.4, 2.1969), INIT_DIST0 =
INT_MAX >> 2 on a 2-pixel border, a forward and a backward raster pass, output tmp * 2^-16.
(OpenCV builds with IPP may route this call to ippiDistanceTransform_5x5 instead.)
`distance_transform_edt` is the exact Euclidean transform, the one the golden fixtures of the
reward/planner rows were generated with (tests/golden/make_golden.py's cv2 stub).
"""
import numpy as np

INIT0 = 0x7fffffff >> 2
HV = int(np.rint(np.float32(1.0) * np.float32(65536)))
DIAG = int(np.rint(np.float32(1.4) * np.float32(65536)))
LONG = int(np.rint(np.float32(2.1969) * np.float32(65536)))


def distance_transform_cv5_loop(src):
    """The two raster passes exactly as the C loop runs them (pure Python: small images only)."""
    src = np.asarray(src)
    h, w = src.shape
    B = 2
    tmp = np.full((h + 2 * B, w + 2 * B), INIT0, dtype=np.int64)
    for i in range(h):
        r = i + B
        for j in range(w):
            c = j + B
            if not src[i, j]:
                tmp[r, c] = 0
                continue
            t0 = tmp[r - 2, c - 1] + LONG
            t0 = min(t0, tmp[r - 2, c + 1] + LONG)
            t0 = min(t0, tmp[r - 1, c - 2] + LONG)
            t0 = min(t0, tmp[r - 1, c - 1] + DIAG)
            t0 = min(t0, tmp[r - 1, c] + HV)
            t0 = min(t0, tmp[r - 1, c + 1] + DIAG)
            t0 = min(t0, tmp[r - 1, c + 2] + LONG)
            t0 = min(t0, tmp[r, c - 1] + HV)
            tmp[r, c] = t0
    out = np.zeros((h, w), dtype=np.float32)
    for i in range(h - 1, -1, -1):
        r = i + B
        for j in range(w - 1, -1, -1):
            c = j + B
            t0 = tmp[r, c]
            if t0 > HV:
                t0 = min(t0, tmp[r + 2, c + 1] + LONG)
                t0 = min(t0, tmp[r + 2, c - 1] + LONG)
                t0 = min(t0, tmp[r + 1, c + 2] + LONG)
                t0 = min(t0, tmp[r + 1, c + 1] + DIAG)
                t0 = min(t0, tmp[r + 1, c] + HV)
                t0 = min(t0, tmp[r + 1, c - 1] + DIAG)
                t0 = min(t0, tmp[r + 1, c - 2] + LONG)
                t0 = min(t0, tmp[r, c + 1] + HV)
                tmp[r, c] = t0
            out[i, j] = np.float32(np.float32(t0) * np.float32(1.0 / 65536.0))
    return out


def distance_transform_cv5(src):
    """Same integers, one numpy pass per row: the in-row recurrence
    tmp[j] = min(c[j], tmp[j-1] + HV) equals HV*j + cummin(c[k] - HV*k)."""
    src = np.asarray(src) != 0
    h, w = src.shape
    B = 2
    tmp = np.full((h + 2 * B, w + 2 * B), INIT0, dtype=np.int64)
    jj = np.arange(w, dtype=np.int64)
    sl = slice(B, B + w)

    def sh(row, d):
        return tmp[row, B + d:B + d + w]
    for i in range(h):
        r = i + B
        c = np.minimum.reduce([sh(r - 2, -1) + LONG, sh(r - 2, 1) + LONG, sh(r - 1, -2) + LONG,
                               sh(r - 1, -1) + DIAG, sh(r - 1, 0) + HV, sh(r - 1, 1) + DIAG,
                               sh(r - 1, 2) + LONG])
        c = np.where(src[i], c, 0)
        t = np.minimum.accumulate(c - HV * jj) + HV * jj
        tmp[r, sl] = np.minimum(t, INIT0 + HV * (jj + 1))
    for i in range(h - 1, -1, -1):
        r = i + B
        c = np.minimum.reduce([tmp[r, sl], sh(r + 2, 1) + LONG, sh(r + 2, -1) + LONG, sh(r + 1, 2) + LONG,
                               sh(r + 1, 1) + DIAG, sh(r + 1, 0) + HV, sh(r + 1, -1) + DIAG,
                               sh(r + 1, -2) + LONG])
        cr = c[::-1]
        t = np.minimum.accumulate(cr - HV * jj) + HV * jj
        tmp[r, sl] = np.minimum(t, INIT0 + HV * (jj + 1))[::-1]
    return (tmp[B:B + h, sl].astype(np.float32) * np.float32(1.0 / 65536.0)).astype(np.float32)


def distance_transform_edt(src):
    from scipy import ndimage
    return ndimage.distance_transform_edt(np.asarray(src) != 0).astype(np.float32)


def goal_field(obs_goal, mode='cv5'):
    """[env/flex_rewards.py:172-177] the shifted signed field the reward samples."""
    goal = np.asarray(obs_goal, dtype=np.float32)
    seg = (goal < 0.5).astype(np.uint8)
    neg = distance_transform_cv5(seg) if mode == 'cv5' else distance_transform_edt(seg)
    g = goal - neg
    return g - g.min()


def goal_pixels(obs_goal):
    """[planners.py:620-621] (col, row) float32 of the pixels with goal < 0.5, row-major."""
    rc = np.argwhere(np.asarray(obs_goal) < 0.5)
    return np.ascontiguousarray(rc[:, ::-1]).astype(np.float32)
