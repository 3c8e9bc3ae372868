"""Similarity alignment of the per-frame initial guess (reference code/utils/umeyama.py:16-109,
init_guess.py:95-106).  CPU: the restatement against the reference function itself and the rotation-vector conversion
against scipy (cv2 is absent).  The reference's full-rank formula U diag(d) Vh^T is not invariant under the SVD's sign
freedom - "what LAPACK returned" is the only thing it can be equal to - so the device's 3 x 3 SVD walks LAPACK's own
dgesdd path (csrc/lapack_svd3.h): the shipped header is compiled for the host here and checked against np.linalg.svd
(signs exact), together with its Python twin oracle/lapack_svd3_np.py.  GPU: mvfit_umeyama must equal the restatement
with numpy's own singular-vector signs - rotation, translation (from the second candidate, the reference's quirk) and
scale."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import ref_import as ri
from oracle import umeyama_np as un


def _cases(n=12, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        npts = 4 if i % 2 == 0 else 17
        src = rng.normal(0, 0.3, (npts, 3))
        a = rng.normal(size=3)
        a *= rng.uniform(0.2, 3.0) / np.linalg.norm(a)
        from scipy.spatial.transform import Rotation
        R = Rotation.from_rotvec(a).as_matrix()
        s = rng.uniform(0.5, 3.0)
        dst = s * src @ R.T + rng.normal(0, 2.0, 3) + rng.normal(0, 0.01, (npts, 3))
        out.append((src, dst))
    return out


@pytest.mark.skipif(not ri.available(), reason='reference tree not mounted')
def test_restatement_equals_reference_umeyama():
    ri.load()
    from utils.umeyama import umeyama as ref_umeyama
    for est in (True, False):
        for src, dst in _cases():
            r0, t0, s0 = ref_umeyama(src.copy(), dst.copy(), est)
            r1, t1, s1, _ = un.umeyama(src, dst, est)
            assert np.abs(r0 - r1).max() < 1e-12 and np.abs(t0 - t1).max() < 1e-12 and abs(s0 - s1) < 1e-12


def test_rotvec_equals_scipy():
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    for i in range(200):
        a = rng.normal(size=3)
        a *= (np.pi - 1e-7 if i < 5 else rng.uniform(0, np.pi - 1e-3)) / np.linalg.norm(a)
        R = Rotation.from_rotvec(a).as_matrix()
        want = Rotation.from_matrix(R).as_rotvec()
        got = un.rotvec(R)
        assert np.abs(got - want).max() < 1e-6 or np.abs(got + want).max() < 1e-6       # theta = pi: +-axis is the same rotation
    assert np.array_equal(un.rotvec(np.eye(3)), np.zeros(3))


def _svd_test_matrices(n, seed):
    rng = np.random.default_rng(seed)
    mats = []
    for k in range(n):
        kind = k % 6
        A = rng.normal(size=(3, 3))
        if kind == 1:
            A = A * np.array([1, 1e-3, 1e-6])                      # graded columns
        elif kind == 2:                                            # what umeyama factorises: dst^T src / num of 4 points
            s_ = rng.normal(size=(4, 3))
            R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
            d_ = s_ @ R.T * rng.uniform(0.5, 2) + rng.normal(0, 0.01, (4, 3))
            A = (d_ - d_.mean(0)).T @ (s_ - s_.mean(0)) / 4
        elif kind == 3:
            A = np.diag(rng.normal(size=3)) @ np.linalg.qr(rng.normal(size=(3, 3)))[0]
        elif kind == 4:
            A = np.triu(A)
        elif kind == 5:
            A = rng.normal(size=(3, 2)) @ rng.normal(size=(2, 3))  # rank 2
        mats.append(A)
    return np.ascontiguousarray(np.stack(mats))


def test_device_svd_header_returns_numpys_singular_vector_pairs():
    """csrc/lapack_svd3.h (the code mvfit_umeyama runs per frame) built for the host with g++, no FMA contraction."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(tempfile.mkdtemp(), 'libsvd3_host.so')
    subprocess.run(['g++', '-O2', '-ffp-contract=off', '-shared', '-fPIC', os.path.join(root, 'tests', 'svd3_host_shim.cpp'), '-o', so],
                   check=True)
    lib = C.CDLL(so)
    A = _svd_test_matrices(12000, seed=1)
    n = A.shape[0]
    U, S, Vh = np.zeros((n, 3, 3)), np.zeros((n, 3)), np.zeros((n, 3, 3))
    lib.svd3_batch(*(a.ctypes.data_as(C.c_void_p) for a in (A, U, S, Vh)), n)
    Un, Sn, Vn = np.linalg.svd(A)
    assert np.abs(S - Sn).max() <= 1e-13 * max(1.0, Sn.max())
    full = Sn[:, 2] > 1e-9 * Sn[:, 0]
    assert np.abs(U[full] - Un[full]).max() < 1e-10 and np.abs(Vh[full] - Vn[full]).max() < 1e-10
    # rank 2: the pair of the vanishing singular value is rounding noise (its sign too); the other two are numpy's,
    # and umeyama's rank-2 branch (:60-68) does not depend on that sign
    assert np.abs(U[~full][:, :, :2] - Un[~full][:, :, :2]).max() < 1e-7
    assert np.abs(Vh[~full][:, :2] - Vn[~full][:, :2]).max() < 1e-7
    assert np.abs(np.einsum('nij,nj,njk->nik', U, S, Vh) - A).max() < 1e-12
    # the Python twin (oracle/lapack_svd3_np.py) is the same arithmetic
    from oracle import lapack_svd3_np as L
    for i in range(0, 600):
        u, s_, vh = L.svd3(A[i])
        assert np.abs(np.asarray(u) - U[i]).max() < 1e-14 and np.abs(np.asarray(vh) - Vh[i]).max() < 1e-14


@pytest.mark.gpu
def test_gpu_umeyama_equals_the_restatement_with_numpys_signs():
    import torch
    from tests.gpu_helpers import make_engine
    from tests.helpers import body_model
    eng = make_engine(body_model())
    for est in (True, False):
        cases = _cases(16, seed=3)
        for npts in (4, 17):
            sub = [c for c in cases if c[0].shape[0] == npts]
            src = sub[0][0]
            dst = np.stack([c[1] if i else sub[0][1] for i, c in enumerate(sub)])      # one src (the rest pose), many dst
            out = eng.umeyama(src, dst, estimate_scale=est)
            rot, rvec, trans, scale = (out[k].cpu().numpy() for k in ('rot', 'rvec', 'trans', 'scale'))
            for b in range(dst.shape[0]):
                r1, t1, s1, losses = un.umeyama(src, dst[b], est)                       # numpy's own signs = the reference's
                assert np.abs(rot[b] - r1).max() < 1e-9 and np.abs(trans[b] - t1).max() < 1e-8, (est, npts, b)
                assert abs(scale[b] - s1) < 1e-10 * max(1.0, abs(s1))
                assert abs(abs(np.linalg.det(rot[b])) - 1) < 1e-9
                assert np.abs(rvec[b] - un.rotvec(rot[b])).max() < 1e-9
    eng.close()


@pytest.mark.gpu
def test_single_view_depth_guess_equals_its_restatement():
    """mvfit_depth_guess (through init_guess.single_view_joints3d, batched) against the line-by-line NumPy restatement of
    init_guess.py:54-74 on the demo's first camera and keypoints."""
    import os
    import torch
    from mvsmplfitting_amd.init_guess import single_view_joints3d
    from oracle import init_guess_np as ig
    from tests.gpu_helpers import make_engine
    from tests.helpers import GOLD, body_model
    g = dict(np.load(os.path.join(GOLD, 'demo_fit_smpl.npz')))
    rest = g['init_joints_rest']
    kp = g['keypoints'].reshape(6, 17, 3)
    frames = np.stack([kp[0], kp[0] * np.array([0.9, 1.1, 1.0], np.float32), kp[2]]).astype(np.float32)
    eng = make_engine(body_model())
    out = single_view_joints3d(eng, torch.as_tensor(rest, dtype=torch.float64), g['extris'][0], g['intris'][0], frames).cpu().numpy()
    eng.close()
    for b in range(3):
        ref = ig.single_view_joints3d(rest, g['extris'][0], g['intris'][0], frames[b])
        assert np.abs(out[b] - ref).max() < 1e-9 * np.abs(ref).max()
