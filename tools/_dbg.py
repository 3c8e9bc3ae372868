import sys, numpy as np
sys.path.insert(0, '.')
from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights
import bench
for B, opts in ((5, dict(resident_pass=3)), (5, dict(resident_pass=0)), (33, {}), (5, {})):
    eng = MvFit(syn.make_body_model(0, skin_topk=4), options=opts)
    cams, gt, conf, x0 = bench.build_inputs(eng, syn, 0, B, 1, 8)
    print('inputs ok', B, opts, flush=True)
    xf, st = eng.fit(x0, stage_weights(1536.0))
    print('fit ok', st['passes'], eng.pass_profile()['form'], flush=True)
    eng.close()
