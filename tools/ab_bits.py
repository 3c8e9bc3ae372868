"""Developer tool: A/B bit comparison of two builds of libmvfit on the same inputs - closure gradients at random points and
whole fits (rounds, final parameters).  python tools/ab_bits.py <lib A> <lib B> [vposer|gmm]
The other build: `git stash; make -C mvsmplfitting_amd/csrc OUT=../libmvfit_old.so OBJDIR=build_old; git stash pop` (or a checkout of
the commit to compare with); tools/ab_check.sh runs the three modes + the phase timing + three bench lines in one gpurun call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mvsmplfitting_amd import _lib, synthetic as syn  # noqa: E402
from mvsmplfitting_amd.engine import MvFit, stage_weights  # noqa: E402
import bench  # noqa: E402

la, lb = sys.argv[1], sys.argv[2]
use_vp = len(sys.argv) > 3 and sys.argv[3] == 'vposer'
use_gmm = len(sys.argv) > 3 and sys.argv[3] == 'gmm'
B, V = 32, 8
model = syn.make_body_model(0, skin_topk=4)
res = []
for lib in (la, lb):
    eng = MvFit(model, vposer=syn.make_vposer_decoder() if use_vp else None, gmm=syn.gmm_constants(syn.make_gmm()) if use_gmm else None,
                library=os.path.abspath(lib))
    cams, gt, conf, x0 = bench.build_inputs(eng, syn, 0, B, 1, V)
    rng = np.random.RandomState(5)
    flags = _lib.F_VPOSER if use_vp else (_lib.F_PRIOR_GMM if use_gmm else 0)
    stages = stage_weights(1536.0, flags=flags)
    out = []
    for k in range(3):
        x = x0.copy()
        x[:, :85] += 0.2 * rng.randn(B, 85).astype(np.float32)
        if use_vp:
            x[:, 86:118] = 0.5 * rng.randn(B, 32).astype(np.float32)
        c = eng.closure(x, stages[min(k, len(stages) - 1)])
        out.append((c['loss'].cpu().numpy(), c['grad'].cpu().numpy()))
    xf, st = eng.fit(x0, stages)
    res.append((out, xf.cpu().numpy(), st['n_closure'].cpu().numpy(), st['final_loss'].cpu().numpy()))
    eng.close()
(oa, xa, na, fa), (ob, xb, nb, fb) = res
names = ['betas 0:10', 'global_orient 10:13', 'body_pose 13:82', 'transl 82:85', 'scale 85', 'embedding 86:118']
sl = [slice(0, 10), slice(10, 13), slice(13, 82), slice(82, 85), slice(85, 86), slice(86, 118)]
for k, ((l1, g1), (l2, g2)) in enumerate(zip(oa, ob)):
    print('closure %d: loss bits equal %s; gradient words differing %d of %d' % (k, (l1.view(np.uint32) == l2.view(np.uint32)).all(),
          int((g1.view(np.uint32) != g2.view(np.uint32)).sum()), g1.size))
    for nm, s in zip(names, sl):
        d = (g1[:, s].view(np.uint32) != g2[:, s].view(np.uint32)).sum()
        if d:
            print('    %-22s %d words differ, max abs %.3e (max |g| %.3e)' % (nm, d, np.abs(g1[:, s] - g2[:, s]).max(), np.abs(g1[:, s]).max()))
print('fit: closures equal %s (A max %d, B max %d); final parameters bit-equal %s; final losses bit-equal %s' % (
    (na == nb).all(), na.max(), nb.max(), (xa.view(np.uint32) == xb.view(np.uint32)).all(), (fa.view(np.uint32) == fb.view(np.uint32)).all()))
