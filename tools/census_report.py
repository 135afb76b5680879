#!/usr/bin/env python3
"""The parity census on the device, printed: every unfiltered 10-step push sequence of tests/golden/census.npz through
drp_rollout on the trained weights, against the reference's own trajectory, lists and one-ulp twin (tests/_census.py).
Writes the raw per-row arrays to gpurun_out/census_dev.npz.   python tools/census_report.py [fused|mfma|valu|split]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import _census as C  # noqa: E402
from dyn_res_pile_manip_amd import synthetic as syn, weights, _lib  # noqa: E402
from dyn_res_pile_manip_amd.engine import Engine  # noqa: E402
from dyn_res_pile_manip_amd.planners import world2cam_affine  # noqa: E402


def main():
    engine = sys.argv[1] if len(sys.argv) > 1 else 'fused'
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'census.npz'))
    w = np.load(os.path.join(ROOT, 'tests', 'golden', 'weights_trained.npz'))
    eng = Engine(0)
    eng.load_weights(weights.blob_from_state_dict(w), 0.08)
    eng.set_camera(world2cam_affine(syn.demo_cam_extrinsics()), 24.0, syn.demo_cam_params())
    eng.set_engine(_lib.ENGINES[engine])
    obs_goal = syn.goal_distance_image(syn.goal_mask('I'))
    out = {}
    np.set_printoptions(linewidth=200, precision=3)
    for name in C.SIZES:
        p = 'census/' + name + '/'
        eng.set_goal(syn.goal_field(obs_goal), g[p + 'goal_coor'])
        states, rew, flips, dev = C.device_rows(eng, g, p)
        margin, disp = g[p + 'margin'], C.displacement(g, p)
        tw_dev, tw_flips, tw_r, ref_r = g[p + 'twin_dev'], g[p + 'twin_flips'], g[p + 'twin_next_r'], g[p + 'next_r']
        B, H = margin.shape
        for k, v in (('flips', flips), ('dev', dev), ('rew', rew)):
            out[p + k] = v
        print('==== %s (%s engine): %d rows x %d steps' % (name, engine, B, H))
        for tau in (1e-8, 5e-8, 2e-7, 1e-6, 5e-6):
            fc = C.first_true(margin < tau)
            print('  margin < %.0e: %2d rows cross; first crossing step histogram %s' %
                  (tau, (fc < H).sum(), np.bincount(fc, minlength=H + 1)))
        for who, fl, dv, rr in (('device', flips, dev, rew), ('twin  ', tw_flips, tw_dev, tw_r)):
            ff = C.first_true(fl > 0)
            rows_f = ff < H
            # the margin of the reference trajectory at each row's first flipped step, and the deviation just before it
            m_at = np.array([margin[b, ff[b]] for b in range(B) if rows_f[b]])
            before = np.array([dv[b, :ff[b]].max() if ff[b] > 0 else 0.0 for b in range(B)])
            after = np.array([dv[b, ff[b]:].max() if ff[b] < H else 0.0 for b in range(B)])
            print('  %s: rows with a flipped list %2d; first-flip step histogram %s' % (who, rows_f.sum(), np.bincount(ff, minlength=H + 1)))
            if rows_f.any():
                print('          margin at the first flip: median %.1e, max %.1e; receivers flipped at that step: max %d' %
                      (np.median(m_at), m_at.max(), max(fl[b, ff[b]] for b in range(B) if rows_f[b])))
            print('          max dev before a row\'s first flip %.2e (rel. to the step displacement: %.2e); after: median %.2e max %.2e' %
                  (before.max(), max((dv[b, :ff[b]] / np.maximum(disp[b, :ff[b]], 1e-12)).max() if ff[b] > 0 else 0.0 for b in range(B)),
                   np.median(after[rows_f]) if rows_f.any() else 0.0, after.max()))
            print('          |d final reward|: max %.2e, rows without a flip: %.2e; relative max %.2e' %
                  (np.abs(rr[:, -1] - ref_r[:, -1]).max(), np.abs(rr[:, -1] - ref_r[:, -1])[~rows_f].max() if (~rows_f).any() else 0.0,
                   (np.abs(rr[:, -1] - ref_r[:, -1]) / np.abs(ref_r[:, -1])).max()))
            print('          dev per step, max over rows: %s' % np.array2string(dv.max(0), formatter={'float_kind': lambda x: '%.1e' % x}))
        # the device's flips against the margin: the largest margin at which a list changed
        fl_steps = flips > 0
        new_flip = fl_steps & ~np.concatenate([np.zeros((B, 1), bool), np.cumsum(fl_steps, 1)[:, :-1] > 0], 1)
        if new_flip.any():
            print('  device: margins at first flips sorted: %s' % np.sort(margin[new_flip]))
        m = 'mppi/' + name + '/'
        macts = g[m + 'act_seqs']
        _, mrew, _, _ = C.device_rows(eng, g, p, macts)
        out[m + 'rew'] = mrew[:, -1]
        r_ref, r_tw = g[m + 'reward'], g[m + 'twin_reward']
        from scipy.special import softmax
        upd = (softmax(0.1 * mrew[:, -1].astype(np.float64))[:, None, None] * macts.astype(np.float64)).sum(0)
        print('  mppi 1024 rows: max |d reward| device %.2e (twin %.2e); median device %.2e (twin %.2e); |d update| device %.2e (twin %.2e); '
              'arg-max device %d ref %d twin %d; reward spread of the population %.3f' %
              (np.abs(mrew[:, -1] - r_ref).max(), np.abs(r_tw - r_ref).max(), np.median(np.abs(mrew[:, -1] - r_ref)),
               np.median(np.abs(r_tw - r_ref)), np.abs(upd - g[m + 'update']).max(), np.abs(g[m + 'twin_update'] - g[m + 'update']).max(),
               mrew[:, -1].argmax(), r_ref.argmax(), r_tw.argmax(), r_ref.std()))
        big = np.abs(mrew[:, -1] - r_ref) > 10 * np.median(np.abs(mrew[:, -1] - r_ref)) + 1e-4 * np.abs(r_ref)
        print('          rows whose reward differs by more than rounding: device %d, twin %d (twin rows with a flipped list: %d)' %
              (big.sum(), (np.abs(r_tw - r_ref) > 10 * np.median(np.abs(r_tw - r_ref)) + 1e-4 * np.abs(r_ref)).sum(),
               g[m + 'twin_flip_rows'].sum()))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'census_dev_%s.npz' % engine), **out)
    eng.close()


if __name__ == '__main__':
    main()
