import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import stage_weights as eng_stage_weights
from tests.gpu_helpers import from118, make_engine, to118
from tests.helpers import GOLD, body_model
g = dict(np.load(os.path.join(GOLD, 'fit_l2.npz')))
eng = make_engine(body_model(), None)
cams = (g['cam_R'], g['cam_t'], g['cam_f'], g['cam_c'])
B = g['x0'].shape[0]
eng.set_problems(cams, g['gt_xy'], g['conf'])
x0 = np.stack([to118(g['x0'][b], False) for b in range(B)]).astype(np.float32)
for sparse in (0, 1):
    stages = eng_stage_weights(1536.0, flags=_lib.F_SPARSE_VERTS if sparse else 0)
    tr = eng.fit_trace(120)
    xf, st = eng.fit(x0, stages)
    tr = tr.cpu().numpy().astype(np.float64)
    eng.fit_trace(0)
    for b in range(B):
        ex = [np.abs(from118(tr[b, k, :118], False) - g['trace32'][b][k, :-1]).max() for k in range(40)]
        er = [np.abs(g['trace64'][b][k, :-1] - g['trace32'][b][k, :-1]).max() for k in range(40)]
        print('sparse', sparse, 'b', b, 'dev-vs-ref32:', ' '.join('%.1e' % v for v in ex))
        print('                ref64-vs-ref32:', ' '.join('%.1e' % v for v in er))
