"""Report (not a test): LBS vertex pass duration and roofline fraction vs the number of problems.
PYTHONPATH=. python tests/report_vertex_pass.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import MvFit

CONST4 = 82680 + 826800 + 17114760 + 6890 * 4 * 8       # v_template + shapedirs + posedirs + 4 (weight, joint) pairs per vertex
PER_PROBLEM = 2032 + 82680


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [32, 64, 128, 256]
    model = syn.make_body_model(0, skin_topk=4)
    eng = MvFit(model)
    cams = syn.make_camera_ring(8)
    for B in Bs:
        rng = np.random.default_rng(B)
        x = np.zeros((B, 118), np.float32)
        x[:, :86] = rng.normal(0, 0.2, (B, 86))
        x[:, 85] = 1.0
        eng.set_problems(cams, np.zeros((B, 8, 17, 2), np.float32), np.ones((B, 8, 17), np.float32))
        eng.vertices(x)
        torch.cuda.synchronize()
        ms = min(eng.profile_vertex_pass_ms(64) for _ in range(3))
        byts = CONST4 + PER_PROBLEM * B
        print('B %4d  %.2f us  algorithmic %.2f MB  %.0f GB/s  frac %.3f' % (B, ms * 1e3, byts / 1e6, byts / ms / 1e6, byts / ms / 1e6 / 8000.0))
    eng.close()


if __name__ == '__main__':
    main()
