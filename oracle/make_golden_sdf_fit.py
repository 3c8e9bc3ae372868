"""TEST INFRASTRUCTURE - writes tests/golden/fit_sdf.npz: staged fits WITH the interpenetration term run by the REFERENCE'S
OWN code (create_loss(interpenetration=True), create_fitting_closure, LBFGSLs, run_fitting - the stage loop of
non_linear_solver.py:156-211 as restated by oracle/make_golden.run_reference_fit; the `sdf` package bound to the
reference's kernel source compiled for the host, see oracle/make_golden_sdf_term.py).  Run in the build container:

    make -C oracle && python -m oracle.make_golden_sdf_fit

Two stages, BOTH carrying the term (the yaml's stage-3 / stage-4 weights incl. coll_loss_weights 1000 / 4500,
cfg_files/fit_smpl.yaml:40-68), from the start points of tests/golden/sdf_term_ref.npz (bodies with a vertex in the
first triangle's shadow: the term and its gradient are non-zero from the first closure on).  So every closure of these
fits is a round of the device's CHAINED structure (vertex pass -> term kernels -> step kernel): the (x, loss) of every
closure call (float32: the only precision the reference's term runs in) is what tests/test_gpu_trajectory.py follows the device fit against - a wrong branch
that only shows in the chained rounds of a full fit shows there."""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from mvsmplfitting_amd import synthetic as syn          # noqa: E402
from oracle import ref_import as ri                      # noqa: E402
from oracle import sdf_ref                               # noqa: E402
from oracle.make_golden import CASES, run_reference_fit, stage_weights      # noqa: E402
from oracle.make_golden_sdf_term import bind_reference_sdf_module           # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
TRACE_LEN = 120
FIT_CASES = ['l2_s3_v6', 'l2_top4_v8']
# round 6: the VPoser branch (cfg_files/fit_smpl.yaml:35-37) with the term - 'vp_s0_v8' over the two stages that carry it (like
# the cases above), and 'vp_s0_v8/yaml4' over ALL FOUR yaml stages (coll_loss_weights 0, 0, 1000, 4500): the device's lead
# stages, the hand-over at the stage boundary and the service rounds in one fit
VP_CASES = [('vp_s0_v8', (2, 3)), ('vp_s0_v8/yaml4', (0, 1, 2, 3))]
COLL_W = {0: 0.0, 1: 0.0, 2: 1000.0, 3: 4500.0}


def main():
    assert ri.available() and sdf_ref.available(), 'needs /root/reference and oracle/_ref (make -C oracle)'
    bind_reference_sdf_module()
    lsp = ri.real_lsp_regressor()
    t = dict(np.load(os.path.join(GOLD, 'sdf_term_ref.npz')))
    out = {}
    stages = [dict(stage_weights(st), coll_loss_weight=COLL_W[st]) for st in (2, 3)]
    for name in FIT_CASES:
        cfg = CASES[name]
        model = syn.make_body_model(0, skin_topk=cfg.get('skin_topk'), kp_regressor=lsp)
        cams = tuple(t[name + '/' + k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
        x0 = np.asarray(t[name + '/x'], np.float64)
        # float32 only: the reference's term cannot run in float64 (the op returns a float32 grid whatever the vertices'
        # dtype, sdf/sdf/sdf.py:21-24, and grid_sample refuses the mix)
        for dtn, sfx in (('float32', '32'),):
            rp = ri.RefProblem(model, cams, t[name + '/gt_xy'], t[name + '/conf'], dtype=dtn, use_vposer=False,
                               interpenetration=True)
            final, xf, ncl, trace = run_reference_fit(rp, x0, stages)
            print(name, dtn, 'closures/stage', ncl, 'final', final, 'first losses', trace[:3, -1])
            out[name + '/trace' + sfx] = trace[:TRACE_LEN]
            out[name + '/ncl' + sfx] = np.asarray(ncl)
            out[name + '/final' + sfx] = np.float64(final)
            out[name + '/xf' + sfx] = xf
        out[name + '/x0'] = x0
        out[name + '/gt_xy'] = t[name + '/gt_xy']
        out[name + '/conf'] = t[name + '/conf']
        for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'):
            out[name + '/' + k] = t[name + '/' + k]
        out[name + '/model_checksum'] = np.float64(syn.model_checksum(model))
    for key, stage_ids in VP_CASES:
        name = key.split('/')[0]
        cfg = CASES[name]
        model = syn.make_body_model(0, skin_topk=cfg.get('skin_topk'), kp_regressor=lsp)
        vpw = syn.make_vposer_decoder(**cfg['vp'])
        cams = tuple(t[name + '/' + k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
        x0 = np.asarray(t[name + '/x'], np.float64)
        st_v = [dict(stage_weights(st), coll_loss_weight=COLL_W[st]) for st in stage_ids]
        rp = ri.RefProblem(model, cams, t[name + '/gt_xy'], t[name + '/conf'], dtype='float32', use_vposer=True, vposer_weights=vpw,
                           interpenetration=True)
        final, xf, ncl, trace = run_reference_fit(rp, x0, st_v)
        print(key, 'float32 closures/stage', ncl, 'final', final, 'first losses', trace[:3, -1], flush=True)
        out[key + '/trace32'] = trace[:TRACE_LEN]
        out[key + '/ncl32'] = np.asarray(ncl)
        out[key + '/final32'] = np.float64(final)
        out[key + '/xf32'] = xf
        out[key + '/x0'] = x0
        out[key + '/stage_index'] = np.asarray(stage_ids)
        out[key + '/gt_xy'] = t[name + '/gt_xy']
        out[key + '/conf'] = t[name + '/conf']
        for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'):
            out[key + '/' + k] = t[name + '/' + k]
        out[key + '/model_checksum'] = np.float64(syn.model_checksum(model))
    out['stage_index'] = np.asarray([2, 3])
    out['coll_w'] = np.asarray([COLL_W[2], COLL_W[3]])
    np.savez_compressed(os.path.join(GOLD, 'fit_sdf.npz'), **out)
    print('wrote', os.path.join(GOLD, 'fit_sdf.npz'))


def spread(n=24):
    """python -m oracle.make_golden_sdf_fit spread [n]: the reference's own float32 fits of the two VPoser cases from n starts = the
    recorded start and n - 1 copies perturbed by 1e-6 (relative) -> tests/golden/fit_sdf_vp_spread.npz (x0 [n, 49] and final32 [n]
    per case).  These fits have several optima (3.6 k / 4.6 k / 12.2 k / 13.8 k were all seen on the device from the one recorded
    start): which one a float32 program ends on is decided by its last bits, so the device's end point is held against the
    reference's own spread (tests/test_gpu_trajectory.py), like the demo and the real-caller VPoser case."""
    assert ri.available() and sdf_ref.available(), 'needs /root/reference and oracle/_ref (make -C oracle)'
    bind_reference_sdf_module()
    lsp = ri.real_lsp_regressor()
    t = dict(np.load(os.path.join(GOLD, 'sdf_term_ref.npz')))
    out = {}
    for key, stage_ids in VP_CASES:
        name = key.split('/')[0]
        cfg = CASES[name]
        model = syn.make_body_model(0, skin_topk=cfg.get('skin_topk'), kp_regressor=lsp)
        vpw = syn.make_vposer_decoder(**cfg['vp'])
        cams = tuple(t[name + '/' + k] for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'))
        x0 = np.repeat(np.asarray(t[name + '/x'], np.float64)[None], n, 0)
        x0[1:] *= 1.0 + 1e-6 * np.random.default_rng(17).standard_normal(x0[1:].shape)
        st_v = [dict(stage_weights(st), coll_loss_weight=COLL_W[st]) for st in stage_ids]
        finals, ncls = [], []
        for i in range(n):
            rp = ri.RefProblem(model, cams, t[name + '/gt_xy'], t[name + '/conf'], dtype='float32', use_vposer=True, vposer_weights=vpw,
                               interpenetration=True)
            final, xf, ncl, trace = run_reference_fit(rp, x0[i], st_v)
            finals.append(final); ncls.append(sum(ncl))
            print(key, i, 'final', final, 'closures', ncl, flush=True)
        out[key + '/x0'] = x0
        out[key + '/final32'] = np.asarray(finals, np.float64)
        out[key + '/ncl32'] = np.asarray(ncls)
    np.savez_compressed(os.path.join(GOLD, 'fit_sdf_vp_spread.npz'), **out)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'spread':
        spread(int(sys.argv[2]) if len(sys.argv) > 2 else 24)
    else:
        main()
