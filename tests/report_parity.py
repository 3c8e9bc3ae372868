"""Prints the measured parity errors of the HIP closure against the reference goldens
(run on the GPU box; output is copied into DESIGN.md / profiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvsmplfitting_amd import _lib
from tests.gpu_helpers import flags_for, from118, make_engine, to118
from tests.helpers import CASES, load_case

print('%-18s %-6s %10s %10s %10s %10s | ref fp32: %10s %10s' % ('case', 'mode', 'loss_rel', 'grad_rel', 'verts_abs', 'joints_abs', 'loss_rel', 'grad_rel'))
for name in sorted(CASES):
    cfg, g, model, vpw, gmm, wts, cams = load_case(name)
    eng = make_engine(model, vpw, gmm)
    B = g['x'].shape[0]
    eng.set_problems(cams, g['gt_xy'], g['conf'])
    if 'joints3d' in g:
        eng.set_joints3d(g['joints3d'][:, :, :3], g['joints3d'][:, :, 3])
    x = np.stack([to118(g['x'][b], cfg['use_vposer']) for b in range(B)]).astype(np.float32)
    for sparse in (False, True):
        w = dict(wts); w['flags'] = flags_for(cfg) | (_lib.F_SPARSE_VERTS if sparse else 0)
        out = eng.closure(x, w, want_grad=True, want_verts=True, want_joints=True)
        loss = out['loss'].cpu().numpy().astype(np.float64)
        grad = out['grad'].cpu().numpy().astype(np.float64)
        verts = out['verts'].cpu().numpy().astype(np.float64)
        joints = out['joints'].cpu().numpy().astype(np.float64)
        el = (np.abs(loss - g['loss64']) / np.abs(g['loss64'])).max()
        eg = 0; eg32 = 0
        for b in range(B):
            gm = from118(grad[b], cfg['use_vposer']); gr = g['grad64'][b]
            if cfg.get('fix_shape'): gm = gm[10:]
            eg = max(eg, np.abs(gm - gr).max() / np.abs(gr).max())
            eg32 = max(eg32, np.abs(g['grad32'][b] - gr).max() / np.abs(gr).max())
        ev = np.abs(verts[:2] - g['verts64_as32']).max()
        ej = np.abs(joints - g['joints64']).max()
        el32 = (np.abs(g['loss32'] - g['loss64']) / np.abs(g['loss64'])).max()
        print('%-18s %-6s %10.2e %10.2e %10.2e %10.2e | %20.2e %10.2e' % (name, 'sparse' if sparse else 'full', el, eg, ev, ej, el32, eg32))
    eng.close()
