"""Developer tool: what the work queue of a single-launch fit can reach at best (list scheduling of the problems' own closure counts on 128
rows, every round at the plain kernels' time per round) against what it measures.  python tools/queue_bound.py [frames]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mvsmplfitting_amd import synthetic as syn  # noqa: E402
from mvsmplfitting_amd.engine import MvFit, stage_weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
stages = stage_weights(1536.0)
res = {}
for wq in (1, 0):
    eng = MvFit(syn.make_body_model(0, skin_topk=4), options=dict(work_queue=wq))
    cams, gt, conf, x0 = bench.build_inputs(eng, syn, 0, B, 1, 8)
    x0_d = torch.tensor(x0, device=eng.device)
    eng.fit(x0_d, stages)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        xf, st = eng.fit(x0_d, stages)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    res[wq] = (1e3 * float(np.median(ts)), st['n_closure'].cpu().numpy())
    eng.close()
ncl = res[1][1]
rows = np.zeros(128)
for b in range(B):                       # the queue hands problems out in index order to the row that is free first
    r = int(np.argmin(rows)) if b >= 128 else b
    rows[r] += ncl[b]
per = -(-B // 128)
sub = sum(int(ncl[lo:lo + (B + per - 1) // per].max()) for lo in range(0, B, (B + per - 1) // per))
print('%d frames: closures per problem %d ... %d (mean %.0f); sub-batches: %d rounds (measured %.2f ms = %.1f us per round); '
      'list scheduling on 128 rows: %d rounds of the busiest row (perfect packing: %.0f) -> at the sub-batches\' time per round %.2f ms; '
      'measured with the queue %.2f ms' % (B, ncl.min(), ncl.max(), ncl.mean(), sub, res[0][0], 1e3 * res[0][0] / sub, int(rows.max()),
                                           ncl.sum() / 128.0, rows.max() * res[0][0] / sub, res[1][0]))
