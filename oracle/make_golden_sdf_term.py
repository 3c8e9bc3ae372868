"""TEST INFRASTRUCTURE - writes tests/golden/sdf_term_ref.npz: the interpenetration term as the REFERENCE'S OWN
SMPLifyLoss.forward computes it (code/utils/fitting.py:251-253 `from sdf import SDF`, :282-288 boxes, :352-393 the
term), run here in float32 on the CPU.  Run in the build container:

    make -C oracle && python -m oracle.make_golden_sdf_term

The reference's `sdf` package is a CUDA/pybind extension whose wrapper does not build against this PyTorch; its
kernel source does (oracle/Makefile -> oracle/_ref/libsdf_ref.so, unmodified source, launch geometry replayed).
`sys.modules['sdf']` is bound to a module whose `SDF.forward(faces, vertices, grid_size)` is the reference's
sdf/sdf/sdf.py:19-24 with `_C.sdf` replaced by that library: phi = zeros; num_faces = faces.size(0) (the launcher,
sdf_cuda_kernel.cu:314) - the reference's call site passes faces.reshape(1, -1, 3), so that is ONE triangle.
Everything else - create_loss(interpenetration=True), reset_loss_weights, create_fitting_closure, the bounding
boxes, grid_sample, the square, autograd through SMPL / VPoser - is the reference's code, unmodified.

Per case: flat parameters x, the observations, and for each evaluated weight set the reference's total loss and
gradient WITH the term and with coll_loss_weight = 0 (same x), i.e. pen = difference and S = sqrt(pen) / w.
Cases are chosen with S > 0 (a vertex under the first triangle's shadow): L2 prior (6 views), the top-4 skinning body
(8 views), VPoser; each at a single hand-picked weight and at the yaml's stage-3 / stage-4 weights
(cfg_files/fit_smpl.yaml:55-59: 0, 0, 1000, 4500).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from mvsmplfitting_amd import synthetic as syn          # noqa: E402
from oracle import closure_np as cn                      # noqa: E402
from oracle import ref_import as ri                      # noqa: E402
from oracle import sdf_ref                               # noqa: E402
from oracle.make_golden import CASES, stage_weights      # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
YAML_COLL_W = [0.0, 0.0, 1000.0, 4500.0]

# (closure golden the inputs come from, problem index, weight sets: (stage, coll_loss_weight))
TERM_CASES = {
    'l2_s3_v6': dict(src='l2_s3_v6', weights=[(3, 40.0), (2, 1000.0), (3, 4500.0)]),
    'l2_top4_v8': dict(src='l2_top4_v8', weights=[(2, 40.0), (2, 1000.0), (3, 4500.0)]),
    'vp_s0_v8': dict(src='vp_s0_v8', weights=[(0, 40.0), (2, 1000.0), (3, 4500.0)]),
}


def bind_reference_sdf_module():
    """sys.modules['sdf'].SDF = sdf/sdf/sdf.py:19-24 on the reference's kernel compiled for the host."""
    import torch

    class SDF(torch.nn.Module):
        def forward(self, faces, vertices, grid_size=32):
            f = faces.detach().cpu().numpy().astype(np.int32)
            num_faces = f.shape[0]                              # faces.size(0): sdf_cuda_kernel.cu:314
            v = vertices.detach().cpu().numpy().astype(np.float32)
            phi = sdf_ref.sdf(f.reshape(-1, 3)[:num_faces], v, int(grid_size))
            return torch.from_numpy(phi)

    m = types.ModuleType('sdf')
    m.SDF = SDF
    sys.modules['sdf'] = m


def main():
    assert ri.available() and sdf_ref.available(), 'needs /root/reference and oracle/_ref (make -C oracle)'
    bind_reference_sdf_module()
    lsp = ri.real_lsp_regressor()
    out = {}
    for name, tc in TERM_CASES.items():
        cfg = CASES[tc['src']]
        g = dict(np.load(os.path.join(GOLD, 'closure_%s.npz' % tc['src'])))
        model = syn.make_body_model(0, skin_topk=cfg.get('skin_topk'), kp_regressor=lsp)
        vpw = syn.make_vposer_decoder(**cfg['vp']) if cfg['use_vposer'] else None
        cams = (g['cam_R'], g['cam_t'], g['cam_f'], g['cam_c'])
        # pick the problems of this golden whose first-triangle shadow holds a vertex (S > 0), by the reference itself
        picked = []
        for b in range(g['x'].shape[0]):
            rp = ri.RefProblem(model, cams, g['gt_xy'][b], g['conf'][b], dtype='float32', use_vposer=cfg['use_vposer'],
                               vposer_weights=vpw, interpenetration=True)
            x = g['x'][b]
            rows = []
            for stage, cw in tc['weights']:
                w = dict(stage_weights(stage), coll_loss_weight=cw)
                L1, g1, verts, _ = rp.eval_closure(x, w)
                L0, g0, _, _ = rp.eval_closure(x, dict(w, coll_loss_weight=0.0))
                rows.append((stage, cw, L1, L0, g1, g0))
            pen = rows[0][2] - rows[0][3]
            print(name, 'b', b, 'pen(w=%g) = %.6g  S = %.6g' % (rows[0][1], pen, np.sqrt(max(pen, 0.0)) / rows[0][1]))
            if pen > 0:
                picked.append((b, x, rows, verts))
        for attempt in range(200):
            if picked:
                break
            # no problem of the golden has a vertex under the first triangle: seeded draws around its first problem
            # (pose part only) until the reference reports S > 0
            rng = np.random.default_rng(7000 + attempt)
            x = np.array(g['x'][0], np.float64)
            n0 = 17 if cfg['use_vposer'] else 13
            x[n0:] += rng.normal(0, 0.6 if cfg['use_vposer'] else 0.3, x.shape[0] - n0)
            x[10:13] += rng.normal(0, 0.5, 3)
            rp = ri.RefProblem(model, cams, g['gt_xy'][0], g['conf'][0], dtype='float32', use_vposer=cfg['use_vposer'],
                               vposer_weights=vpw, interpenetration=True)
            stage, cw = tc['weights'][0]
            w = dict(stage_weights(stage), coll_loss_weight=cw)
            if rp.eval_closure(x, w)[0] - rp.eval_closure(x, dict(w, coll_loss_weight=0.0))[0] <= 0:
                continue
            rows = []
            for stage, cw in tc['weights']:
                w = dict(stage_weights(stage), coll_loss_weight=cw)
                L1, g1, verts, _ = rp.eval_closure(x, w)
                L0, g0, _, _ = rp.eval_closure(x, dict(w, coll_loss_weight=0.0))
                rows.append((stage, cw, L1, L0, g1, g0))
            print(name, 'seeded draw', attempt, 'pen(w=%g) = %.6g' % (rows[0][1], rows[0][2] - rows[0][3]))
            picked.append((0, x, rows, verts))
        assert picked, name
        b, x, rows, verts = picked[0]
        out[name + '/b'] = np.int32(b)
        out[name + '/x'] = np.asarray(x, np.float64)
        out[name + '/gt_xy'] = g['gt_xy'][b]
        out[name + '/conf'] = g['conf'][b]
        for k in ('cam_R', 'cam_t', 'cam_f', 'cam_c'):
            out[name + '/' + k] = g[k]
        out[name + '/stage'] = np.asarray([r[0] for r in rows], np.int32)
        out[name + '/coll_w'] = np.asarray([r[1] for r in rows], np.float64)
        out[name + '/loss_with'] = np.asarray([r[2] for r in rows], np.float64)
        out[name + '/loss_without'] = np.asarray([r[3] for r in rows], np.float64)
        out[name + '/grad_with'] = np.stack([r[4] for r in rows]).astype(np.float64)
        out[name + '/grad_without'] = np.stack([r[5] for r in rows]).astype(np.float64)
        out[name + '/verts32'] = verts.astype(np.float32)[::10]          # every 10th vertex: a vertex-level cross-check
        out[name + '/model_checksum'] = np.float64(syn.model_checksum(model))
    np.savez_compressed(os.path.join(GOLD, 'sdf_term_ref.npz'), **out)
    print('wrote', os.path.join(GOLD, 'sdf_term_ref.npz'))


if __name__ == '__main__':
    main()
