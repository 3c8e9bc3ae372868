"""CPU: the interpenetration term of the oracle (oracle/sdf_term_np.py inside oracle/closure_np.py) against the term
as the REFERENCE'S OWN SMPLifyLoss.forward computes it (code/utils/fitting.py:251-253, 282-288, 352-393), recorded in
tests/golden/sdf_term_ref.npz by oracle/make_golden_sdf_term.py: create_loss(interpenetration=True) + the reference's
closure in float32, its `sdf` package bound to the reference's kernel source compiled for the host (oracle/_ref).

The reference runs in float32, the oracle in float64.  Measured agreement: penalty 2e-6 ... 3e-5 relative (a sum of
trilinear samples whose cells are chosen by float32 coordinates), its gradient 7e-6 ... 6e-4 of the largest entry, the
loss without the term 1e-7.  Asserted: total loss within 1e-5 of the term-free loss + 1e-4 of the penalty; the penalty
within 1e-4 relative + the float32 resolution of the reference's total (it is recovered as a difference of two float32
losses); penalty gradient within 2e-3 of its largest entry.  tests/test_gpu_sdf_term.py checks the device against the
same file."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import synthetic as syn
from oracle.make_golden import CASES, stage_weights
from oracle.make_golden_sdf_term import TERM_CASES
from tests.helpers import GOLD, body_model, oracle_for


def load_term_case(name):
    g = np.load(os.path.join(GOLD, 'sdf_term_ref.npz'))
    c = {k.split('/', 1)[1]: g[k] for k in g.files if k.startswith(name + '/')}
    cfg = CASES[TERM_CASES[name]['src']]
    model = body_model(0, cfg.get('skin_topk'))
    assert abs(syn.model_checksum(model) - float(c['model_checksum'])) < 1e-6 * float(c['model_checksum'])
    vpw = syn.make_vposer_decoder(**cfg['vp']) if cfg['use_vposer'] else None
    cams = (c['cam_R'], c['cam_t'], c['cam_f'], c['cam_c'])
    return cfg, c, model, vpw, cams


@pytest.mark.parametrize('name', sorted(TERM_CASES))
def test_oracle_term_matches_the_references_own_forward(name):
    cfg, c, model, vpw, cams = load_term_case(name)
    orc = oracle_for(model, vpw, None)
    sdf = dict(faces=model['faces'], num_faces=1, grid_size=128)        # as wired: fitting.py:367-368
    for i in range(len(c['stage'])):
        w = dict(stage_weights(int(c['stage'][i])), coll_loss_weight=float(c['coll_w'][i]))
        L1, g1, o = orc.closure(c['x'], cams, c['gt_xy'], c['conf'], w, use_vposer=cfg['use_vposer'], sdf=sdf)
        L0, g0, _ = orc.closure(c['x'], cams, c['gt_xy'], c['conf'], dict(w, coll_loss_weight=0.0),
                                use_vposer=cfg['use_vposer'], sdf=sdf)
        pen_ref = c['loss_with'][i] - c['loss_without'][i]
        assert pen_ref > 0 and o['sdf']['S'] > 0, 'case does not exercise the term'
        assert np.abs(o['vertices'][::10] - c['verts32']).max() < 1e-5
        assert abs(L0 - c['loss_without'][i]) <= 1e-5 * abs(c['loss_without'][i])
        assert abs(L1 - c['loss_with'][i]) <= 1e-5 * abs(c['loss_without'][i]) + 1e-4 * pen_ref, (L1, c['loss_with'][i])
        assert abs((L1 - L0) - pen_ref) <= 1e-4 * pen_ref + 2.4e-7 * abs(c['loss_with'][i]), (L1 - L0, pen_ref)
        gp, gp_ref = g1 - g0, c['grad_with'][i] - c['grad_without'][i]
        assert np.abs(gp - gp_ref).max() <= 2e-3 * np.abs(gp_ref).max(), (np.abs(gp - gp_ref).max(), np.abs(gp_ref).max())
        assert np.abs(g0 - c['grad_without'][i]).max() <= 2e-4 * np.abs(c['grad_without'][i]).max()
        assert np.abs(g1 - c['grad_with'][i]).max() <= 2e-4 * np.abs(c['grad_without'][i]).max() + 2e-3 * np.abs(gp_ref).max()
