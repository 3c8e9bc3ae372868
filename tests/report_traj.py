"""Report (not a test): device fit traces next to the reference float32 traces.  PYTHONPATH=. python tests/report_traj.py"""
import os, numpy as np
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import stage_weights as esw
from tests.gpu_helpers import from118, make_engine, to118
from tests.helpers import GOLD, body_model
for name,use_vp in (('l2',False),('vposer',True)):
    g=dict(np.load(os.path.join(GOLD,'fit_%s.npz'%name)))
    eng=make_engine(body_model(), syn.make_vposer_decoder() if use_vp else None)
    eng.set_problems((g['cam_R'],g['cam_t'],g['cam_f'],g['cam_c']),g['gt_xy'],g['conf'])
    x0=np.stack([to118(g['x0'][b],use_vp) for b in range(2)]).astype(np.float32)
    for sparse in (0,1):
        st=esw(1536.0,flags=(_lib.F_VPOSER if use_vp else 0)|(_lib.F_SPARSE_VERTS if sparse else 0))
        tr=eng.fit_trace(120); xf,s=eng.fit(x0,st); T=tr.cpu().numpy().astype(np.float64); eng.fit_trace(0)
        for b in range(2):
            n=min(60,int(s['n_closure'][b]))
            ex=[np.abs(from118(T[b,k,:118],use_vp)-g['trace32'][b][k,:-1]).max() for k in range(n)]
            el=[abs(T[b,k,118]-g['trace32'][b][k,-1])/abs(g['trace32'][b][k,-1]) for k in range(n)]
            print(name,'sparse',sparse,b,'ncl',int(s['n_closure'][b]),'ref32',g['ncl32'][b].sum(),'final',float(s['final_loss'][b]),g['final32'][b],g['final'][b])
            print('  ex',' '.join('%.0e'%v for v in ex[::2])); print('  el',' '.join('%.0e'%v for v in el[::2]))
    eng.close()
