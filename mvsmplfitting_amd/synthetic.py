"""Seeded synthetic inputs for the fitting hot path (numpy only, no compute path).

No SMPL model file, GMM prior file or multi-frame keypoint set ships with the
reference (reference models/smpl/readme.txt:1-3, code/prior.py:119-125), so every
benchmark / parity input is generated here from a seed:

* ``make_body_model``  - an SMPL-*shaped* body: 6890 vertices / 13776 faces (closed
  UV-sphere topology stretched to a 1.7 m ellipsoid), 24-joint kinematic tree with
  SMPL's parent table, dense or top-k skinning weights, shapedirs/posedirs of the
  SMPL sizes.  Field names follow the reference's ``data_struct``
  (code/smplx/body_models_scale.py:169-305) so the same dict can also be wrapped in
  the reference's ``Struct`` by the oracle harness.
* ``make_lsp_regressor`` - a 14x6890 sparse keypoint regressor shaped like
  data/J_regressor_lsp.npz (4-9 non-zeros per row, rows sum to 1).
* ``make_camera_ring`` - V pinhole cameras on a circle looking at the origin
  (reference camera model: code/camera.py:93-117, fx == fy, code/init.py:113-119).
* ``make_vposer_decoder`` - decoder weights with the VPoser layer sizes
  (code/model/VPoser.py:188-195) whose zero latent decodes to a near-rest pose.
* ``make_gmm`` - a max-mixture pose prior dict with the pickle's keys
  (code/prior.py:128-131).
* ``make_frames`` - per-frame ground-truth parameter draws.

Everything is ``numpy.random.default_rng(seed)`` driven and float32 at the boundary,
like the reference's ``to_np`` (code/smplx/utils.py:36-39).
"""
from __future__ import annotations

import numpy as np

NUM_VERTS = 6890
NUM_FACES = 13776
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_BASIS = 207
NUM_KP = 17

# SMPL kinematic tree (kintree_table[0] with the root set to -1,
# reference code/smplx/body_models_scale.py:300-302).
SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int32)

# nose, leye, reye, lear, rear (reference code/smplx/vertex_ids.py:25-29 in the order of
# code/smplx/vertex_joint_selector.py:38-43).
FACE_VERTEX_IDS = np.array([332, 2800, 6260, 583, 4071], dtype=np.int32)

# 19 = 14 LSP + 5 face keypoints -> 17 dataset keypoints
# (reference code/utils/utils.py:453-457, 'lsp14' + 'smpllsp').
LSP_JOINT_MAP = np.array([14, 15, 16, 17, 18, 9, 8, 10, 7, 11, 6, 3, 2, 4, 1, 5, 0],
                         dtype=np.int32)

# Approximate SMPL rest-pose joint locations (metres, y up), only used to lay out
# the synthetic skeleton.
_REST_JOINTS = np.array([
    [0.00, 0.00, 0.00], [0.07, -0.09, 0.00], [-0.07, -0.09, 0.00], [0.00, 0.11, -0.02],
    [0.10, -0.47, 0.01], [-0.10, -0.47, 0.01], [0.00, 0.25, 0.00], [0.09, -0.87, -0.03],
    [-0.09, -0.87, -0.03], [0.00, 0.30, 0.02], [0.11, -0.93, 0.09], [-0.11, -0.93, 0.09],
    [0.00, 0.51, -0.03], [0.08, 0.42, -0.02], [-0.08, 0.42, -0.02], [0.00, 0.58, 0.02],
    [0.17, 0.45, -0.02], [-0.17, 0.45, -0.02], [0.43, 0.44, -0.03], [-0.43, 0.44, -0.03],
    [0.68, 0.44, -0.03], [-0.68, 0.44, -0.03], [0.76, 0.43, -0.02], [-0.76, 0.43, -0.02],
], dtype=np.float64)


def _uv_sphere(rings: int = 84, segs: int = 82):
    """Closed genus-0 triangle mesh: rings*segs + 2 vertices, 2*rings*segs faces."""
    th = np.pi * (np.arange(1, rings + 1) / (rings + 1))          # polar angle per ring
    ph = 2.0 * np.pi * np.arange(segs) / segs
    st, ct = np.sin(th)[:, None], np.cos(th)[:, None]
    ring_xyz = np.stack([st * np.cos(ph)[None, :],
                         np.broadcast_to(ct, (rings, segs)),
                         st * np.sin(ph)[None, :]], axis=-1).reshape(-1, 3)
    verts = np.concatenate([[[0.0, 1.0, 0.0]], ring_xyz, [[0.0, -1.0, 0.0]]], axis=0)
    top, bot = 0, rings * segs + 1

    def vid(r, s):
        return 1 + r * segs + (s % segs)

    faces = []
    for s in range(segs):
        faces.append([top, vid(0, s + 1), vid(0, s)])
        faces.append([bot, vid(rings - 1, s), vid(rings - 1, s + 1)])
    for r in range(rings - 1):
        for s in range(segs):
            a, b, c, d = vid(r, s), vid(r, s + 1), vid(r + 1, s), vid(r + 1, s + 1)
            faces.append([a, b, c])
            faces.append([b, d, c])
    return verts, np.asarray(faces, dtype=np.int32)


def make_lsp_regressor(seed: int = 7):
    """14 x 6890 keypoint regressor, CSR-like triplets + dense view."""
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for r in range(14):
        nnz = int(rng.integers(4, 10))
        c = np.sort(rng.choice(NUM_VERTS, size=nnz, replace=False))
        w = rng.random(nnz) + 0.05
        w = (w / w.sum()).astype(np.float32)
        rows += [r] * nnz
        cols += c.tolist()
        vals += w.tolist()
    return (np.asarray(rows, np.int32), np.asarray(cols, np.int32),
            np.asarray(vals, np.float32))


def dense_from_triplets(rows, cols, vals, shape=(14, NUM_VERTS)):
    m = np.zeros(shape, dtype=np.float32)
    m[rows, cols] = vals
    return m


def make_body_model(seed: int = 0, skin_topk: int | None = None, kp_regressor=None):
    """Synthetic SMPL-shaped model.

    Returns a dict of float32 / int32 arrays:
      v_template[6890,3] shapedirs[6890,3,10] posedirs[207,20670] J_regressor[24,6890]
      parents[24] lbs_weights[6890,24] kp_regressor[14,6890] face_vertex_ids[5]
      joint_map[17] faces[13776,3]
    ``posedirs`` is stored the way the reference keeps its buffer: [207, 6890*3],
    column = 3*vertex + coord (code/smplx/body_models_scale.py:292-297).
    ``skin_topk``: keep only the k largest skinning weights per vertex (real SMPL has
    <= 4 non-zeros per row); None keeps the dense softmax rows.
    ``kp_regressor``: (rows, cols, vals) triplets; default = make_lsp_regressor().
    """
    rng = np.random.default_rng(seed)
    sph, faces = _uv_sphere()
    assert sph.shape[0] == NUM_VERTS and faces.shape[0] == NUM_FACES
    # ellipsoid body: 0.56 x 1.74 x 0.36 m, centred a little below the pelvis
    v_template = sph * np.array([0.28, 0.87, 0.18]) + np.array([0.0, -0.12, 0.0])
    v_template += rng.normal(0.0, 0.002, size=v_template.shape)

    # skeleton squeezed into the ellipsoid so every joint has nearby vertices
    J_rest = _REST_JOINTS * np.array([0.33, 0.88, 1.0]) + np.array([0.0, -0.05, 0.0])

    d2 = ((v_template[:, None, :] - J_rest[None, :, :]) ** 2).sum(-1)       # [Nv,24]
    logits = -d2 / (2.0 * 0.09 ** 2)
    logits -= logits.max(axis=1, keepdims=True)
    W = np.exp(logits)
    W /= W.sum(axis=1, keepdims=True)
    if skin_topk is not None:
        kth = np.sort(W, axis=1)[:, -skin_topk][:, None]
        W = np.where(W >= kth, W, 0.0)
        W /= W.sum(axis=1, keepdims=True)

    J_regressor = np.zeros((NUM_JOINTS, NUM_VERTS))
    for j in range(NUM_JOINTS):
        near = np.argsort(d2[:, j])[:30]
        w = rng.random(30) + 0.1
        J_regressor[j, near] = w / w.sum()

    # smooth-ish shape basis: low-frequency functions of the template position
    freq = rng.normal(0.0, 2.5, size=(NUM_BETAS, 3, 3))
    phase = rng.uniform(0, 2 * np.pi, size=(NUM_BETAS, 3))
    shapedirs = np.empty((NUM_VERTS, 3, NUM_BETAS))
    for l in range(NUM_BETAS):
        arg = v_template @ freq[l].T + phase[l][None, :]
        shapedirs[:, :, l] = 0.012 * np.sin(arg) / (1.0 + 0.25 * l)
    posedirs_v = rng.normal(0.0, 0.0005, size=(NUM_VERTS, 3, NUM_POSE_BASIS))
    posedirs = posedirs_v.reshape(NUM_VERTS * 3, NUM_POSE_BASIS).T          # [207, 20670]

    if kp_regressor is None:
        kp_regressor = make_lsp_regressor()
    kp_dense = dense_from_triplets(*kp_regressor)

    f32 = np.float32
    return dict(
        v_template=np.ascontiguousarray(v_template, f32),
        shapedirs=np.ascontiguousarray(shapedirs, f32),
        posedirs=np.ascontiguousarray(posedirs, f32),
        J_regressor=np.ascontiguousarray(J_regressor, f32),
        parents=SMPL_PARENTS.copy(),
        lbs_weights=np.ascontiguousarray(W, f32),
        kp_regressor=kp_dense,
        face_vertex_ids=FACE_VERTEX_IDS.copy(),
        joint_map=LSP_JOINT_MAP.copy(),
        faces=faces,
    )


def look_at_rotation(eye, target=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0)):
    """World->camera rotation with +z looking from eye to target, image y down."""
    eye = np.asarray(eye, np.float64)
    z = np.asarray(target, np.float64) - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64))       # right-handed: x = z x up -> image right
    x /= np.linalg.norm(x)
    y = np.cross(z, x)                                  # image down
    return np.stack([x, y, z], axis=0)


def make_camera_ring(num_views: int = 8, radius: float = 4.0, height: float = 0.0,
                     focal: float = 2400.0, center=(1024.0, 768.0)):
    """cam_R[V,3,3], cam_t[V,3], cam_f[V], cam_c[V,2]; x_cam = R x_world + t."""
    Rs, ts = [], []
    for v in range(num_views):
        a = 2.0 * np.pi * v / num_views
        eye = np.array([radius * np.sin(a), height, radius * np.cos(a)])
        R = look_at_rotation(eye)
        Rs.append(R)
        ts.append(-R @ eye)
    f32 = np.float32
    return (np.asarray(Rs, f32), np.asarray(ts, f32),
            np.full((num_views,), focal, f32),
            np.tile(np.asarray(center, f32)[None, :], (num_views, 1)))


def make_vposer_decoder(seed: int = 11, hidden: int = 512, latent: int = 32,
                        num_joints: int = 23, gain: float = 0.35, identity_bias: bool = True):
    """fc1[512,32]+b, fc2[512,512]+b, out[138,512]+b (reference VPoser.py:188-195).

    With ``identity_bias`` the output bias is laid out so that z = 0 decodes close to
    identity rotations: out.reshape(23,3,2) has a1 = [:, :, 0], a2 = [:, :, 1]
    (VPoser.py:165-174).  Without it the decoded rotations are arbitrary, which exercises
    all four branches of the matrix->quaternion conversion (VPoser.py:64-96).
    """
    rng = np.random.default_rng(seed)

    def lin(n_out, n_in, g):
        bound = g / np.sqrt(n_in)
        return (rng.uniform(-bound, bound, size=(n_out, n_in)).astype(np.float32),
                rng.uniform(-bound, bound, size=(n_out,)).astype(np.float32))

    w1, b1 = lin(hidden, latent, 1.0)
    w2, b2 = lin(hidden, hidden, 1.0)
    w3, b3 = lin(num_joints * 6, hidden, gain)
    if identity_bias:
        b3 = np.tile(np.array([1, 0, 0, 1, 0, 0], np.float32), num_joints)
        b3 = b3 + rng.normal(0, 0.05, size=b3.shape).astype(np.float32)
    return dict(fc1_w=w1, fc1_b=b1, fc2_w=w2, fc2_b=b2, out_w=w3, out_b=b3)


def make_gmm(seed: int = 5, num_gaussians: int = 8, dim: int = 69):
    """{'means','covars','weights'} like priors/gmm_XX.pkl (reference prior.py:128-131)."""
    rng = np.random.default_rng(seed)
    means = rng.normal(0.0, 0.2, size=(num_gaussians, dim))
    covars = np.empty((num_gaussians, dim, dim))
    for m in range(num_gaussians):
        q, _ = np.linalg.qr(rng.normal(size=(dim, dim)))
        covars[m] = (q * rng.uniform(0.01, 0.2, size=dim)[None, :]) @ q.T
    w = rng.random(num_gaussians) + 0.2
    return dict(means=means, covars=covars, weights=w / w.sum())


def gmm_constants(gmm, dtype=np.float32):
    """means[M,69], precisions[M,69,69], nll_weights[M] exactly as the reference's
    MaxMixturePrior.__init__ derives them (code/prior.py:135-160)."""
    means = gmm['means'].astype(dtype)
    covs = gmm['covars'].astype(dtype)
    precisions = np.stack([np.linalg.inv(c) for c in covs]).astype(dtype)
    sqrdets = np.array([np.sqrt(np.linalg.det(c)) for c in gmm['covars']])
    const = (2 * np.pi) ** (69 / 2.)
    nll_weights = np.asarray(gmm['weights'] / (const * (sqrdets / sqrdets.min())))
    return means, precisions, nll_weights.astype(dtype)


def make_frames(num_frames: int, seed0: int = 1000, betas=None):
    """Ground-truth draws per frame: dict of [B, .] float32."""
    out = dict(betas=[], global_orient=[], body_pose=[], transl=[], scale=[])
    for f in range(num_frames):
        rng = np.random.default_rng(seed0 + f)
        out['betas'].append(rng.normal(0, 0.5, NUM_BETAS) if betas is None else betas)
        out['global_orient'].append(rng.normal(0, 0.3, 3))
        out['body_pose'].append(rng.normal(0, 0.2, 69))
        out['transl'].append(rng.normal(0, 0.1, 3))
        out['scale'].append(np.ones(1))
    return {k: np.asarray(v, np.float32) for k, v in out.items()}


def project_points(points, cam_R, cam_t, cam_f, cam_c):
    """Pinhole projection of points[B,K,3] into V views -> [B,V,K,2] (float64)."""
    p = np.einsum('vij,bkj->bvki', cam_R.astype(np.float64), points.astype(np.float64))
    p = p + cam_t.astype(np.float64)[None, :, None, :]
    uv = p[..., :2] / p[..., 2:3]
    return uv * cam_f.astype(np.float64)[None, :, None, None] + \
        cam_c.astype(np.float64)[None, :, None, :]


def make_observations(joints, cams, seed: int = 4242, noise_px: float = 2.0):
    """Noisy 2-D keypoints + confidences from 3-D keypoints[B,17,3].

    Returns gt_xy[B,V,17,2], conf[B,V,17] float32 (reference layout per view:
    keypoints [P,17,3] = x, y, conf; code/utils/data_parser.py:387-393).
    """
    rng = np.random.default_rng(seed)
    uv = project_points(joints, *cams)
    uv = uv + rng.normal(0.0, noise_px, size=uv.shape)
    conf = rng.uniform(0.5, 1.0, size=uv.shape[:-1])
    return uv.astype(np.float32), conf.astype(np.float32)


def model_checksum(model) -> float:
    """Cheap drift detector for seeded models (sum of abs of every float field)."""
    s = 0.0
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights',
              'kp_regressor'):
        s += float(np.abs(model[k].astype(np.float64)).sum())
    return s
