"""CPU: the SDF-term restatement (oracle/sdf_term_np.py) against torch grid_sample + autograd, and the
closure gradient with the term against central differences of the loss with phi held fixed."""
import numpy as np
import pytest
import torch

from oracle import closure_np as cn
from oracle import sdf_term_np as st
from tests.helpers import load_case, oracle_for


def _torch_term(verts, phi, coll_w):
    """fitting.py:352-393 on one person, phi given (it is a no_grad constant there)."""
    v = torch.tensor(verts, dtype=torch.float64, requires_grad=True)
    vertices = v[None]
    boxes = torch.zeros(1, 2, 3, dtype=torch.float64)
    boxes[0, 0, :] = vertices[0].min(dim=0)[0]
    boxes[0, 1, :] = vertices[0].max(dim=0)[0]
    center = boxes.mean(dim=1).unsqueeze(dim=1)
    scale = (1 + 0.2) * 0.5 * (boxes[:, 1] - boxes[:, 0]).max(dim=-1)[0][:, None, None]
    local = (vertices - center[0].unsqueeze(0)) / scale[0].unsqueeze(0)
    grid = local.view(1, -1, 1, 1, 3)
    phi_val = torch.nn.functional.grid_sample(torch.tensor(phi, dtype=torch.float64)[None, None], grid,
                                              align_corners=False).view(1, -1)
    pen = (coll_w * phi_val.sum() / 1) ** 2
    pen.backward()
    return float(pen), v.grad.numpy()


@pytest.mark.parametrize('G', [16, 32])
def test_sampling_and_box_adjoint_match_torch(G):
    rng = np.random.default_rng(3)
    verts = rng.normal(size=(500, 3)) * np.array([0.3, 0.8, 0.2]) + np.array([0.1, -0.2, 2.5])
    phi = np.abs(rng.normal(size=(G, G, G))).astype(np.float32)
    phi[rng.random((G, G, G)) < 0.5] = 0
    pen, g, aux = st.sdf_term(verts, None, 0.7, grid_size=G, phi=phi)
    pen_t, g_t = _torch_term(verts, phi, 0.7)
    assert abs(pen - pen_t) <= 1e-12 * abs(pen_t)
    assert np.abs(g - g_t).max() <= 1e-10 * np.abs(g_t).max()


def test_term_with_real_op_is_consistent():
    """phi from the restated op on the first triangle (as wired): nonzero somewhere, >= 0, pen finite; the
    closure gradient equals central differences with phi frozen."""
    cfg, g, model, vpw, gmm, wts, cams = load_case('l2_s3_v6')
    orc = oracle_for(model, vpw, gmm)
    wts = dict(wts, coll_loss_weight=0.5)
    sdf = dict(faces=model['faces'], num_faces=1, grid_size=32)
    x = g['x'][0]
    L, grad, out = orc.closure(x, cams, g['gt_xy'][0], g['conf'][0], wts, sdf=sdf)
    assert out['sdf']['phi'].min() >= 0
    L0, grad0, _ = orc.closure(x, cams, g['gt_xy'][0], g['conf'][0], dict(wts, coll_loss_weight=0.0), sdf=sdf)
    S = out['sdf']['S']
    assert abs((L - L0) - (0.5 * S) ** 2) <= 1e-12 * max(1.0, abs(L))
    if S == 0:
        pytest.skip('no vertex inside the first triangle\'s shadow for this pose')
    phi = out['sdf']['phi']
    def pen_of(xx):
        o = orc.body(dict(cn.unpack(xx, False), use_vposer=False))
        return st.sdf_term(o['vertices'], None, 0.5, grid_size=32, phi=phi)[0]
    gd = grad - grad0
    for i in [0, 11, 14, 40, 83, 85]:
        h = 1e-6
        e = np.zeros_like(x); e[i] = h
        fd = (pen_of(x + e) - pen_of(x - e)) / (2 * h)
        assert abs(fd - gd[i]) <= 1e-5 * max(1.0, abs(gd).max()), (i, fd, gd[i])
