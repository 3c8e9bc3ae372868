"""Developer tool: in-kernel timeline of a steady-state chunk of the RESIDENT vertex pass (needs the -DMVFIT_TIMING build
mvsmplfitting_amd/libmvfit_timing.so: make -C mvsmplfitting_amd/csrc OBJDIR=build_timing OUT=../libmvfit_timing.so EXTRA=-DMVFIT_TIMING).
MVFIT_LIBRARY=$PWD/mvsmplfitting_amd/libmvfit_timing.so python tests/vp_resident_timeline.py [B]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mvsmplfitting_amd import synthetic as syn  # noqa: E402
from mvsmplfitting_amd.engine import MvFit, stage_weights  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
eng = MvFit(syn.make_body_model(0, skin_topk=4))
cams, gt, conf, x0 = bench.build_inputs(eng, syn, 0, B, 1, 8)
xf, st = eng.fit(x0, stage_weights(1536.0))
if len(sys.argv) > 2:
    eng.set_options(resident_pass=int(sys.argv[2]))
    xf, st = eng.fit(x0, stage_weights(1536.0))
print('B = %d, fit passes %s, %s' % (B, st['passes'], eng.pass_profile()))
vb = (C.c_longlong * 16)()
eng._lib.mvfit_debug_vp(vb)
base = [vb[i] for i in range(16)]
alone = eng.profile_resident_pass_ms(100)
torch.cuda.synchronize()
eng._lib.mvfit_debug_vp(vb)
form = eng.pass_profile()['form']
if form == 3:
    cn = ['transform request issued (next chunk)', 'contraction done', 'partials written', 'barrier P', 'coefficient request issued (next chunk)',
          'requests landed', 'barrier Y']
    wn = ['-', 'four items blended', 'barrier P', 'applied + stores issued', '-', '-', 'barrier Y']
    for half, who, nm in ((0, 'wave 0 (contraction wave)', cn), (1, 'wave 4 (worker)', wn)):
        n = max(vb[7 + 8 * half] - base[7 + 8 * half], 1)
        print('  %s, %d middle chunks of workgroup 5, cumulative shader-clock cycles since the chunk start:' % (who, n))
        print('     ' + ' | '.join('%s=%.0f' % (nm[i], (vb[i + 8 * half] - base[i + 8 * half]) / n) for i in range(7) if nm[i] != '-'))
    nr = max(vb[8] - base[8], 1)
    print('  whole rounds with four live chunks (%d, worker wave): verdict -> first operands landed (barrier Q) %.0f cycles, -> round end %.0f '
          'cycles; alone %.2f us per round -> the shader clock runs at >= %.2f GHz (the poll is not in the cycles)'
          % (nr, (vb[12] - base[12]) / nr, (vb[13] - base[13]) / nr, alone * 1e3, (vb[13] - base[13]) / nr / (alone * 1e3) * 1e-3))
else:
    names = ['contraction + partials written', 'barrier 1', 'next chunk requested', 'tile 0 blended / applied / stored', 'last tile done',
             'operand + store wait', 'closing barrier']
    for half, who in ((0, 'wave 0 (two chains)'), (1, 'wave 4 (one chain)')):
        n = max(vb[7 + 8 * half] - base[7 + 8 * half], 1)
        print('  %s, %d chunks in buffer 1 of workgroup 5, cumulative shader-clock cycles since the chunk start:' % (who, n))
        print('     ' + ' | '.join('%s=%.0f' % (names[i], (vb[i + 8 * half] - base[i + 8 * half]) / n) for i in range(7)))
print('  alone: %.2f us per round' % (alone * 1e3))
