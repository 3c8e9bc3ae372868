"""CPU: the NumPy oracle against golden vectors produced by the reference itself."""
import numpy as np
import pytest

from oracle import closure_np as cn
from tests.helpers import CASES, load_case, oracle_for


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_matches_reference_golden(name):
    cfg, g, model, vpw, gmm, wts, cams = load_case(name)
    orc = oracle_for(model, vpw, gmm)
    prior = cn.PRIOR_GMM if cfg['prior'] == 'gmm' else cn.PRIOR_L2
    fix_shape = cfg.get('fix_shape', False)
    for b in range(g['x'].shape[0]):
        j3 = (g['joints3d'][b][:, :3], g['joints3d'][b][:, 3]) if 'joints3d' in g else None
        L, grad, out = orc.closure(g['x'][b], cams, g['gt_xy'][b], g['conf'][b], wts,
                                   use_vposer=cfg['use_vposer'], prior=prior, fix_shape=fix_shape, joints3d=j3)
        assert abs(L - g['loss64'][b]) <= 1e-12 * abs(g['loss64'][b])
        gref = g['grad64'][b]
        gmine = grad[10:] if fix_shape else grad
        assert np.abs(gmine - gref).max() <= 1e-10 * np.abs(gref).max()
        assert np.abs(out['joints'] - g['joints64'][b]).max() < 1e-12
        if b < g['verts64_as32'].shape[0]:
            assert np.abs(out['vertices'] - g['verts64_as32'][b]).max() < 2e-7   # stored as f32


def test_fp32_oracle_within_reference_fp32_floor():
    """float32 instantiation of the oracle is as close to fp64 as the reference's own fp32 run."""
    cfg, g, model, vpw, gmm, wts, cams = load_case('l2_s3_v6')
    orc = oracle_for(model, vpw, gmm, np.float32)
    for b in range(2):
        L, grad, out = orc.closure(g['x'][b].astype(np.float32), cams, g['gt_xy'][b], g['conf'][b],
                                   wts)
        assert abs(L - g['loss64'][b]) <= 2e-6 * abs(g['loss64'][b])
        assert np.abs(grad - g['grad64'][b]).max() <= 2e-5 * np.abs(g['grad64'][b]).max()


def test_param_layout_matches_reference_order():
    lay, D = cn.param_layout(False)
    assert D == 86 and lay['betas'] == (0, 10) and lay['body_pose'] == (13, 82) and lay['scale'] == (85, 86)
    lay, D = cn.param_layout(True)
    assert D == 49 and lay['pose_embedding'] == (17, 49)


def test_triangulation_golden_file_matches_oracle():
    import os
    from oracle import triangulate_np as tn
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'triangulate.npz'))
    for name in ('v8', 'v2', 'v16'):
        for b in range(g[name + '_kps'].shape[0]):
            d = np.abs(tn.recompute3d(g[name + '_extris'], g[name + '_intris'], g[name + '_kps'][b]) - g[name + '_joints3d'][b]).max()
            assert d <= 1e-12
