"""Developer tool: per-phase shader-clock breakdown of the step kernel (needs a -DMVFIT_TIMING build)."""
import sys, os, ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the timing build: prebuilt in the build container (make -C mvsmplfitting_amd/csrc OUT=../libmvfit_timing.so OBJDIR=build_timing
# EXTRA=-DMVFIT_TIMING; it travels with the snapshot), else compiled here
TLIB = os.path.join(ROOT, 'mvsmplfitting_amd', 'libmvfit_timing.so')
if not os.path.isfile(TLIB):
    subprocess.run(['make', '-C', os.path.join(ROOT, 'mvsmplfitting_amd', 'csrc'), '-j8', 'OUT=../libmvfit_timing.so', 'OBJDIR=build_timing',
                    'EXTRA=-DMVFIT_TIMING'], check=True)
os.environ['MVFIT_LIBRARY'] = TLIB
import numpy as np, torch
from mvsmplfitting_amd import _lib, synthetic as syn
from mvsmplfitting_amd.engine import MvFit, stage_weights
B, V = 32, 8
model = syn.make_body_model(0, skin_topk=4); cams = syn.make_camera_ring(V)
eng = MvFit(model, vposer=syn.make_vposer_decoder())
lib = eng._lib
fr = syn.make_frames(B); xgt = np.zeros((B,118), np.float32)
for k,(a,b) in dict(betas=(0,10), global_orient=(10,13), body_pose=(13,82), transl=(82,85), scale=(85,86)).items(): xgt[:,a:b] = fr[k]
eng.set_problems(cams, np.zeros((B,V,17,2),np.float32), np.ones((B,V,17),np.float32))
_, joints = eng.vertices(xgt)
gt, conf = syn.make_observations(joints.cpu().numpy(), cams); eng.set_problems(cams, gt, conf)
x0 = np.zeros((B,118), np.float32); x0[:,85]=1
buf = (C.c_longlong*32)()
hb = (C.c_longlong*16)()
ab = (C.c_longlong*16)()
cb = (C.c_longlong*16)()
for name, flags in (('full',0), ('sparse',_lib.F_SPARSE_VERTS), ('vposer_sparse', _lib.F_VPOSER|_lib.F_SPARSE_VERTS), ('vposer_sparse_helpers_off', _lib.F_VPOSER|_lib.F_SPARSE_VERTS)):
    eng.set_options(vposer_helpers=0 if name.endswith('off') else 1)
    lib.mvfit_debug_timing_helpers(hb, 1)
    lib.mvfit_debug_timing_adv(ab, 1)
    lib.mvfit_debug_timing_calls(cb, 1)
    lib.mvfit_debug_timing(buf, 1)
    xf, st = eng.fit(x0, stage_weights(1536.0, flags=flags))
    lib.mvfit_debug_timing(buf, 1)
    n = max(buf[13],1)
    names = ['pose_prep','chain||stream','T/xs','loss','E5 gx','E6 gA','E7 chainT||streamT','E8 gR/gbeta','E9 rodT+asm(+vpbwd)','lb load','lb advance A','lb direction entry/exit','lb advance B+store']
    print(name, 'rounds(block0)=%d avg hist=%.1f direction calls=%d' % (n, buf[14]/n, buf[15]))
    print('   ', ' | '.join('%s=%.0f' % (names[i], buf[i]/n) for i in range(13)), '| total=%.0f cycles/round' % (sum(buf[i] for i in range(13))/n))
    print('    step-kernel prologue=%.0f epilogue (state store + pose/chain of next x + publish)=%.0f' % (buf[24]/n, buf[25]/n))
    print('    wave-0 chain fwd=%.0f (rest of slot 1 = waiting for the basis stream) ; chain bwd=%.0f (rest of slot 6 = waiting for the transposed stream)' % (buf[22]/n, buf[23]/n))
    print('    vposer (helpers: request sent / answers in / summed ; adjoint: joints / answers in / summed): L1=%.0f L2=%.0f out=%.0f | (GS+quat in pose_prep rest) | bwd: joints=%.0f W3T=%.0f W2T=%.0f (W1T in E9 rest)' % tuple(buf[i]/n for i in (26,27,28,29,30,31)))
    lib.mvfit_debug_timing_adv(ab, 1)
    print('    inside advance (cycles per round): ' + ' | '.join('%s=%.0f' % (nm, ab[i]/n) for i, nm in enumerate(['entry->ls_first', 'gtd dot', 'wolfe checks->ls_return', 'ls_return->iter', 'iter->insert done', 'dir end->resume', 'resume->emit'])))
    lib.mvfit_debug_timing_helpers(hb, 1)
    print('    E9 probes (cycles per round): g_beta lanes done=%.0f | Rodrigues adjoint lanes done=%.0f | barrier behind both=%.0f' % (ab[8]/n, ab[9]/n, ab[10]/n))
    print('    basis streams (cycles per round, from the phase start): forward wave 1 / wave 7 done=%.0f / %.0f | transposed wave 1 / wave 7 done=%.0f / %.0f' % (ab[12]/n, ab[14]/n, ab[11]/n, ab[13]/n))
    lib.mvfit_debug_timing_calls(cb, 1)
    nm = ['fast accept', 'fast resume', 'general: step start', 'general: first trial', 'general: bracket', 'general: zoom', 'general: behind a direction']
    print('    optimiser calls (wave 0): ' + ' | '.join('%s %d x %.0f' % (nm[i], cb[2 * i + 1], cb[2 * i] / max(1, cb[2 * i + 1])) for i in range(7)) + ' cycles')
    if hb[2] and (flags & _lib.F_VPOSER): print('    decoder helper (set 0, slice 0): forward %.0f cycles per request (%d), adjoint %.0f (%d); poll iterations %d, with a request %d' % (hb[0]/max(1,hb[2]), hb[2], hb[1]/max(1,hb[3]), hb[3], hb[4], hb[5]))
    if buf[15]:
        sub = [buf[16+i]/buf[15] for i in range(6)]
        print('    compact direction (lb_direction_compact): %.0f cycles per call = ' % (sum(sub) + buf[11]/buf[15]) +
              ' | '.join('%s=%.0f' % (nm, v) for nm, v in zip(['p,u = S^T(q|y_new)', 'w = R^-1 p (+ new column)', 't = Y w - q', 'z = D w + gamma Y^T t', 'a = R^-T z', 'd = S a - gamma t'], sub)) +
              ' | entry/exit=%.0f ; per round (x direction calls / rounds): %.0f' % (buf[11]/buf[15], (sum(sub)*buf[15] + buf[11])/n))
    print('    (cycles = clock64() of workgroup 0, thread 0 - shader clocks, ~2.4 GHz; the build with the marks is slower than the shipped one: relative weights, not absolute times)')

vb = (C.c_longlong*16)()
lib.mvfit_debug_vp(vb)
nl = max(vb[7], 1)
print('vertex pass (workgroup 5, %d launches), cumulative cycles at each mark: [issue+stage | sync1 | mfma-or-blend | sync2 | apply | sync3 | end]' % nl)
print('   contraction wave 0:', ' '.join('%.0f' % (vb[i]/nl) for i in range(7)))
print('   blend wave 4      :', ' '.join('%.0f' % (vb[8+i]/nl) for i in range(7)))
