"""Developer tool: in-kernel timeline of the chunk-loop vertex pass (needs a -DMVFIT_TIMING build; builds one).
PYTHONPATH=. python tests/vp_timeline.py [B]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
subprocess.run('make -C %s/mvsmplfitting_amd/csrc -B CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../include -DMVFIT_TIMING"' % ROOT,
               shell=True, check=True, stdout=subprocess.DEVNULL)
import numpy as np
import torch
from mvsmplfitting_amd import synthetic as syn
from mvsmplfitting_amd.engine import MvFit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = MvFit(syn.make_body_model(0, skin_topk=4))
cams = syn.make_camera_ring(8)
x = np.zeros((B, 118), np.float32)
x[:, :86] = np.random.default_rng(0).normal(0, 0.2, (B, 86))
x[:, 85] = 1
eng.set_problems(cams, np.zeros((B, 8, 17, 2), np.float32), np.ones((B, 8, 17), np.float32))
eng.vertices(x)
torch.cuda.synchronize()
vb = (C.c_longlong * 16)()
eng._lib.mvfit_debug_vp(vb)
base = [vb[i] for i in range(16)]
N = 50
for _ in range(N):
    eng.vertices(x)
torch.cuda.synchronize()
eng._lib.mvfit_debug_vp(vb)
d = [(vb[i] - base[i]) / N for i in range(16)]
if os.environ.get('MVFIT_VP_LOCKSTEP'):
    names = ['operand wait', 'barrier', 'blend', 'mfma+partials', 'apply', 'barrier2', 'stores issued']
    print('B = %d, lock-step chunk loop, iteration 1 of workgroup 5, cumulative shader-clock cycles since the iteration start' % B)
    print('  wave 0 (contraction):', ' '.join('%s=%.0f' % (names[i], d[i]) for i in range(7)))
    print('  wave 4 (blend only) :', ' '.join('%s=%.0f' % (names[i], d[8 + i]) for i in range(7)))
else:
    ln = ['counted operand wait', 'barrier 1', 'coefficient reads + 42 MFMAs', 'partials written', 'barrier 2', 'request of chunk + 2 issued']
    wn = ['-', 'barrier 1', 'two blends', '-', 'barrier 2', 'apply + stores issued']
    print('B = %d, two-role pipeline, third chunk of workgroup 5, cumulative shader-clock cycles since the iteration start' % B)
    print('  wave 0 (loader / contraction):', ' | '.join('%s=%.0f' % (ln[i], d[i]) for i in range(6)))
    print('  wave 4 (worker)              :', ' | '.join('%s=%.0f' % (wn[i], d[8 + i]) for i in (1, 2, 4, 5)))
print('  avg launch us (64 back-to-back):', eng.profile_vertex_pass_ms(64) * 1e3)
