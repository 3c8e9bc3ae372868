"""TEST INFRASTRUCTURE - how far the REFERENCE's own float32 fit of the demo frame (BASELINE configs[0]) moves when the start is
perturbed in the last bits: its staged fit (create_fitting_closure + LBFGSLs + run_fitting in the stage loop of
non_linear_solver.py:156-211, cfg_files/fit_smpl.yaml:40-68) from 48 starts = the reference's initial guess and 47 copies
perturbed by 1e-6 (relative), the generator of tests/test_gpu_demo.py::test_demo_fit_spread_against_the_reference_spread (seed 0,
drawn in the C ABI's 118-parameter layout so that the device fits the SAME 48 starts).

    python -m oracle.make_golden_demo_spread [n] [first]      (build container; ~6 s per fit)

Writes tests/golden/demo_spread<n>.npz: x0 [n,118] (float64 starts), final32 [n], ncl32 [n,4].  The starts are drawn from ONE
stream, row after row, so the first 48 starts of a longer run are the starts of demo_spread48.npz (round 6: 192 starts - 48 give
the distributional test too little power; `first` > 0 re-uses the fits [0, first) of the file with that many starts, after
checking that it holds the same starts).  The last verdict asked for
exactly this: the device ends ~4 % of such starts at ~44.4 k against a reference band of 34-39 k recorded from SIX starts;
whether the reference does the same once in 25 starts is what this file answers."""
from __future__ import annotations

import os
import sys

import numpy as np

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle import ref_import as ri
from oracle.make_golden import GOLD, run_reference_fit


def main(n=48, first=0):
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    vpw = {k: v for k, v in np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')).items() if k != 'source'}
    d = np.load(os.path.join(GOLD, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, kp_regressor=(d['rows'], d['cols'], d['vals']))
    assert np.array_equal(np.array(syn.model_checksum(model)), g['model_checksum'])
    cams = tuple(g[k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
    stages = [dict(data_weight=float(w[0]), body_pose_weight=float(w[1]), shape_weight=float(w[2]),
                   bending_prior_weight=float(w[3]), rho=float(w[4])) for w in g['stage_w']]
    lay, D = cn.param_layout(True)
    sl118 = dict(betas=(0, 10), global_orient=(10, 13), transl=(82, 85), scale=(85, 86), pose_embedding=(86, 118))
    x118 = np.zeros(118)
    x118[85] = 1.0
    for name, (a, b) in lay.items():
        x118[sl118[name][0]:sl118[name][1]] = g['x0'][a:b]
    xs = np.repeat(x118[None], n, 0)
    rng = np.random.default_rng(0)
    xs[1:] *= 1.0 + 1e-6 * rng.standard_normal((n - 1, 118))
    finals, ncls = [], []
    if first:
        old = np.load(os.path.join(GOLD, 'demo_spread%d.npz' % first))
        assert np.array_equal(old['x0'], xs[:first]), 'the shorter file holds other starts'
        finals, ncls = list(old['final32']), [list(r) for r in old['ncl32']]
    for i in range(first, n):
        x49 = np.zeros(D)
        for name, (a, b) in lay.items():
            x49[a:b] = xs[i, sl118[name][0]:sl118[name][1]]
        rp = ri.RefProblem(model, cams, g['gt_xy'], g['conf'], 'float32', use_vposer=True, vposer_weights=vpw)
        final, xf, ncl, trace = run_reference_fit(rp, x49.astype(np.float32), stages)
        finals.append(final); ncls.append(ncl)
        print(i, 'final', final, 'closures/stage', ncl, flush=True)
    np.savez_compressed(os.path.join(GOLD, 'demo_spread%d.npz' % n), x0=xs, final32=np.asarray(finals, np.float64), ncl32=np.asarray(ncls))
    f = np.sort(np.asarray(finals))
    print('reference float32, %d starts: min %.0f median %.0f max %.0f; > 1.1 x median: %d' % (n, f[0], np.median(f), f[-1], (f > 1.1 * np.median(f)).sum()))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 48, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
