"""TEST INFRASTRUCTURE - the REFERENCE's own float32 fits of BASELINE configs[1] problems (1 person x 8 views, GMoF + L2 pose
prior + shape + angle priors, yaml stage weights; SURVEY 8(d) config 2: frames seeded 1000 + f) from 24 starts per frame = the
bench's start (zeros, scale 1) and 23 copies perturbed by 1e-6 (absolute, N(0, 1e-6) on all 86 parameters), for 4 frames:
create_fitting_closure + LBFGSLs + run_fitting in the stage loop of non_linear_solver.py:156-211, cfg_files/fit_smpl.yaml:40-68.

    python -m oracle.make_golden_configs1_spread [frames] [starts]      (build container; ~4 s per fit)

Writes tests/golden/configs1_spread.npz: gt_xy [F,8,17,2], conf [F,8,17] (the problems, made with the float64 oracle's
keypoints), x0 [F,S,118] (starts in the C ABI's layout), final32 [F,S], ncl32 [F,S,4].  The device fits the SAME starts in one
batch (tests/test_gpu_demo.py::test_configs1_fit_spread_against_the_reference_spread): fit-level parity on the headline
workload as a comparison of distributions instead of "<= 1.05 x the worse of two reference runs"."""
from __future__ import annotations

import os
import sys

import numpy as np

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle import ref_import as ri
from oracle.make_golden import GOLD, run_reference_fit, stage_weights


def main(nf=4, ns=24):
    d = np.load(os.path.join(GOLD, 'lsp_regressor.npz'))
    model = syn.make_body_model(0, skin_topk=4, kp_regressor=(d['rows'], d['cols'], d['vals']))
    cams = syn.make_camera_ring(8)
    orc = cn.ClosureOracle(model, np.float64)
    fr = syn.make_frames(nf, seed0=1000)
    kp = np.stack([orc.body(dict({k: fr[k][b] for k in fr}, use_vposer=False), want_cache=False)['joints'] for b in range(nf)])
    gt, conf = syn.make_observations(kp, cams, seed=1007)
    stages = [stage_weights(s) for s in range(4)]
    lay, D = cn.param_layout(False)
    assert D == 86 and all(lay[k] == v for k, v in dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85),
                                                         scale=(85, 86)).items()), lay
    rng = np.random.default_rng(61)
    x0 = np.zeros((nf, ns, 118))
    x0[:, :, 85] = 1.0
    x0[:, 1:, :86] += 1e-6 * rng.standard_normal((nf, ns - 1, 86))
    finals = np.zeros((nf, ns))
    ncls = np.zeros((nf, ns, 4), np.int64)
    for f in range(nf):
        for s in range(ns):
            rp = ri.RefProblem(model, cams, gt[f], conf[f], 'float32', use_vposer=False)
            final, xf, ncl, trace = run_reference_fit(rp, x0[f, s, :86].astype(np.float32), stages)
            finals[f, s] = final; ncls[f, s] = ncl
            print(f, s, 'final', final, 'closures/stage', ncl, flush=True)
    np.savez_compressed(os.path.join(GOLD, 'configs1_spread.npz'), gt_xy=gt, conf=conf, x0=x0, final32=finals, ncl32=ncls,
                        model_checksum=np.array(syn.model_checksum(model)))
    for f in range(nf):
        q = np.sort(finals[f])
        print('frame %d: reference float32, %d starts: min %.4f median %.4f max %.4f; closures %d ... %d'
              % (f, ns, q[0], np.median(q), q[-1], ncls[f].sum(1).min(), ncls[f].sum(1).max()))


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 24)
