"""TEST INFRASTRUCTURE - imports the *reference's own* Python modules as the live oracle.

Only usable where the read-only reference tree is mounted (the build container);
``available()`` is False on the GPU box and every caller must skip.  Nothing here is
copied from the reference: we put ``<ref>/code`` on sys.path, stub the GUI / GL
imports the hot path never touches (cv2, pyrender, trimesh, OpenGL, torchgeometry:
reference utils/utils.py:25-30, fitting.py:33, optimizers/lbfgs_ls.py:9,
model/VPoser.py:5) and instantiate the reference classes with our synthetic arrays
(SMPL accepts ``data_struct=``: body_models_scale.py:98,169-180).
"""
from __future__ import annotations

import contextlib
import os
import sys
import types

import numpy as np

_STAGE = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'reference_stage.tgz')


def _resolve_root() -> str:
    """The reference tree: $MVFIT_REFERENCE, else /root/reference (the build container), else the archive `make -C oracle
    stage` made of the reference's own Python path (oracle/_ref/reference_stage.tgz: git-ignored, travels to the GPU box
    with the snapshot like libsdf_ref.so), unpacked once per archive version into a temporary directory."""
    env = os.environ.get('MVFIT_REFERENCE')
    if env:
        return env
    if os.path.isfile('/root/reference/code/utils/fitting.py') or not os.path.isfile(_STAGE):
        return '/root/reference'
    import hashlib
    import shutil
    import tarfile
    import tempfile
    st = os.stat(_STAGE)
    key = hashlib.sha1(('%s:%d:%d' % (_STAGE, st.st_size, int(st.st_mtime))).encode()).hexdigest()[:12]
    # a per-user directory (mode 0700), never a predictable world-writable path: the archive holds code that is imported and a
    # checkpoint that is unpickled - on a shared box nobody else may be able to put their own tree where this one is expected
    base = os.path.join(os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache'), 'mvfit_reference_stage')
    try:
        os.makedirs(base, mode=0o700, exist_ok=True)
        if os.stat(base).st_uid != os.getuid() or (os.stat(base).st_mode & 0o077):
            raise OSError('not ours')
    except OSError:
        base = tempfile.mkdtemp(prefix='mvfit_reference_stage_')          # (mode 0700, ours by construction; unpacked per process)
    root = os.path.join(base, key)
    if not os.path.isfile(os.path.join(root, '.complete')):
        tmp = tempfile.mkdtemp(prefix='unpack_', dir=base)
        with tarfile.open(_STAGE) as tf:
            try:
                tf.extractall(tmp, filter='data')       # no absolute paths, no links out of the tree, no device files
            except TypeError:                            # (a Python without extraction filters: validate the member paths ourselves)
                for m in tf.getmembers():
                    dest = os.path.realpath(os.path.join(tmp, m.name))
                    if not (m.isfile() or m.isdir()) or not dest.startswith(os.path.realpath(tmp) + os.sep):
                        raise RuntimeError('unexpected member in %s: %r' % (_STAGE, m.name))
                tf.extractall(tmp)
        open(os.path.join(tmp, '.complete'), 'w').close()
        try:
            os.rename(tmp, root)
        except OSError:                              # another process of this user was faster: use its copy
            shutil.rmtree(tmp, ignore_errors=True)
    return root


REF_ROOT = _resolve_root()
STAGED = REF_ROOT != '/root/reference' and not os.environ.get('MVFIT_REFERENCE')
_mods = None


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, 'code', 'utils', 'fitting.py'))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


def load():
    """Import the reference modules once; returns a namespace."""
    global _mods
    if _mods is not None:
        return _mods
    if not available():
        raise RuntimeError('reference tree not mounted at %s' % REF_ROOT)
    _stub('cv2')
    pr = _stub('pyrender')
    prc = _stub('pyrender.constants', RenderFlags=type('RenderFlags', (), {}))
    pr.constants = prc
    _stub('trimesh')
    gl = _stub('OpenGL')
    gl.GLUT = _stub('OpenGL.GLUT')
    _stub('torchgeometry')
    code = os.path.join(REF_ROOT, 'code')
    if code not in sys.path:
        sys.path.insert(0, code)
    import warnings
    warnings.filterwarnings('ignore')
    with _cwd(REF_ROOT):
        import smplx                                  # noqa  (reference package, shadows nothing here)
        from smplx import body_models_scale
        from smplx.utils import Struct
        import camera as ref_camera
        import prior as ref_prior
        from utils import fitting as ref_fitting
        from utils import utils as ref_utils
        from optimizers import optim_factory, lbfgs_ls
        from model import VPoser as ref_vposer
    ns = types.SimpleNamespace(body_models_scale=body_models_scale, Struct=Struct,
                               camera=ref_camera, prior=ref_prior, fitting=ref_fitting,
                               utils=ref_utils, optim_factory=optim_factory,
                               lbfgs_ls=lbfgs_ls, vposer=ref_vposer)
    _mods = ns
    return ns


def real_lsp_regressor():
    """The shipped 14x6890 LSP regressor as (rows, cols, vals) triplets."""
    R = np.load(os.path.join(REF_ROOT, 'data', 'J_regressor_lsp.npz'))['joint_regressor']
    r, c = np.nonzero(R)
    return r.astype(np.int32), c.astype(np.int32), R[r, c].astype(np.float32)


class RefProblem:
    """The reference's model + cameras + loss + closure for ONE problem (B = 1), built the
    way code/init.py:85-159 and code/utils/non_linear_solver.py:127-192 build them."""

    def __init__(self, model: dict, cams, gt_xy, conf, dtype='float64', use_vposer=False,
                 vposer_weights=None, prior='l2', gmm=None, fix_shape=False, rho=100.0,
                 joint_weights=None, joints3d=None, interpenetration=False):
        import torch
        ref = load()
        self.ref = ref
        self.torch = torch
        dt = torch.float64 if dtype == 'float64' else torch.float32
        self.dt = dt
        self.use_vposer = use_vposer
        nv = model['v_template'].shape[0]
        posedirs_v = model['posedirs'].T.reshape(nv, 3, 207)
        kin = np.stack([np.where(model['parents'] < 0, 2 ** 32 - 1, model['parents']),
                        np.arange(24)]).astype(np.int64)
        struct = ref.Struct(f=model['faces'].astype(np.int64), v_template=model['v_template'],
                            shapedirs=model['shapedirs'], J_regressor=model['J_regressor'],
                            posedirs=posedirs_v, kintree_table=kin, weights=model['lbs_weights'])
        mapper = ref.utils.JointMapper(ref.utils.smpl_to_annotation(
            model_type='smpllsp', pose_format='lsp14'))
        with _cwd(REF_ROOT):       # relative np.load('data/J_regressor_lsp.npz'), body_models_scale.py:284
            smpl = ref.body_models_scale.create_scale(
                'unused', model_type='smpllsp', data_struct=struct, joint_mapper=mapper,
                create_global_orient=True, create_body_pose=not use_vposer, create_betas=True,
                create_transl=True, create_scale=True, dtype=dt)
        # the keypoint regressor under test may differ from the shipped file
        smpl.joint_regressor = torch.tensor(model['kp_regressor'], dtype=dt)
        self.smpl = smpl
        cam_R, cam_t, cam_f, cam_c = cams
        self.cameras = []
        for v in range(cam_R.shape[0]):
            cam = ref.camera.create_camera(
                focal_length_x=float(cam_f[v]), focal_length_y=float(cam_f[v]),
                translation=torch.tensor(cam_t[v], dtype=dt).unsqueeze(0),
                rotation=torch.tensor(cam_R[v], dtype=dt).unsqueeze(0),
                center=torch.tensor(cam_c[v], dtype=dt).unsqueeze(0), dtype=dt)
            cam.rotation.requires_grad = False
            cam.translation.requires_grad = False
            self.cameras.append(cam)
        self.vposer = None
        self.pose_embedding = None
        if use_vposer:
            vp = ref.vposer.VPoser(num_neurons=512, latentD=32, data_shape=[1, 23, 3])
            sd = vp.state_dict()
            names = dict(fc1_w='bodyprior_dec_fc1.weight', fc1_b='bodyprior_dec_fc1.bias',
                         fc2_w='bodyprior_dec_fc2.weight', fc2_b='bodyprior_dec_fc2.bias',
                         out_w='bodyprior_dec_out.weight', out_b='bodyprior_dec_out.bias')
            for k, n in names.items():
                sd[n] = torch.tensor(vposer_weights[k])
            vp.load_state_dict(sd)
            vp = vp.to(dtype=dt)
            vp.eval()
            self.vposer = vp
            self.pose_embedding = torch.zeros([1, 32], dtype=dt, requires_grad=True)
        if prior == 'gmm':
            import pickle
            import tempfile
            d = tempfile.mkdtemp()
            M = gmm['means'].shape[0]
            with open(os.path.join(d, 'gmm_%02d.pkl' % M), 'wb') as fh:
                pickle.dump(gmm, fh)
            body_prior = ref.prior.create_prior('gmm', prior_folder=d, num_gaussians=M, dtype=dt)
        else:
            body_prior = ref.prior.create_prior('l2', dtype=dt)
        shape_prior = ref.prior.create_prior('l2', dtype=dt)
        angle_prior = ref.prior.create_prior('angle', dtype=dt)
        self.loss = ref.fitting.create_loss(
            loss_type='smplify', rho=rho, use_joints_conf=True, body_pose_prior=body_prior,
            shape_prior=shape_prior, angle_prior=angle_prior, interpenetration=interpenetration,
            dtype=dt, use_3d=joints3d is not None, fix_shape=fix_shape)
        self.use_3d = joints3d is not None
        self.gt_joints3d = self.joints3d_conf = None
        if joints3d is not None:      # non_linear_solver.py:86-99: gt_joints3d [17,3], joints3d_conf [1,17]
            self.gt_joints3d = torch.tensor(np.asarray(joints3d[0]), dtype=dt)
            self.joints3d_conf = torch.tensor(np.asarray(joints3d[1]), dtype=dt).reshape(1, -1)
        self.gt_joints = torch.tensor(np.asarray(gt_xy)[:, None, :, :], dtype=dt)   # [V,1,17,2]
        self.joints_conf = [torch.tensor(np.asarray(conf)[v][None, :], dtype=dt)
                            for v in range(cam_R.shape[0])]
        jw = np.ones(17, np.float32) if joint_weights is None else joint_weights
        self.joint_weights = torch.tensor(jw, dtype=dt).unsqueeze(0)
        self.monitor = ref.fitting.FittingMonitor(maxiters=30, ftol=1e-9, gtol=1e-9)

    def final_params(self):
        ps = [p for p in self.smpl.parameters() if p.requires_grad]
        if self.vposer is not None:
            ps.append(self.pose_embedding)
        return ps

    def set_flat(self, x):
        torch = self.torch
        off = 0
        with torch.no_grad():
            for p in self.final_params():
                n = p.numel()
                p.copy_(torch.tensor(np.asarray(x[off:off + n]), dtype=self.dt).view_as(p))
                off += n
        assert off == len(x)

    def get_flat(self):
        return np.concatenate([p.detach().cpu().numpy().reshape(-1) for p in self.final_params()])

    def set_weights(self, wts: dict):
        self.loss.reset_loss_weights({k: float(v) for k, v in wts.items() if k != 'rho'})

    def make_optimizer(self, maxiters=30):
        opt, _ = self.ref.optim_factory.create_optimizer(
            self.final_params(), optim_type='lbfgsls', lr=1.0, maxiters=maxiters)
        return opt

    def make_closure(self, optimizer):
        return self.monitor.create_fitting_closure(
            optimizer, self.smpl, camera=self.cameras, gt_joints=self.gt_joints,
            joints_conf=self.joints_conf, joint_weights=self.joint_weights, loss=self.loss,
            gt_joints3d=self.gt_joints3d, joints3d_conf=self.joints3d_conf,
            create_graph=False, use_vposer=self.use_vposer, vposer=self.vposer,
            pose_embedding=self.pose_embedding, return_verts=True, return_full_pose=True,
            use_3d=self.use_3d)

    def eval_closure(self, x, wts):
        """loss, grad, vertices, joints at flat params x (the reference's own code path)."""
        self.set_flat(x)
        self.set_weights(wts)
        opt = self.make_optimizer()
        closure = self.make_closure(opt)
        loss = closure(backward=True)
        grad = np.concatenate([
            (p.grad if p.grad is not None else self.torch.zeros_like(p)).detach().numpy().reshape(-1)
            for p in self.final_params()])
        with self.torch.no_grad():
            bp = self.vposer.decode(self.pose_embedding, output_type='aa').view(1, -1) \
                if self.use_vposer else None
            out = self.smpl(return_verts=True, body_pose=bp, return_full_pose=True)
        return (float(loss), grad, out.vertices[0].detach().numpy(),
                out.joints[0].detach().numpy())


def time_reference_fits(model, cams, gt, conf, stages, use_vposer=False, vposer_weights=None, threads=1, budget_s=10.0,
                        dtype='float32', max_frames=None):
    """bench.py's cpu_baseline leg: the reference ITSELF - its SMPL / cameras / create_loss / create_optimizer('lbfgsls') /
    FittingMonitor.create_fitting_closure / run_fitting (code/utils/fitting.py:71-205, code/optimizers/lbfgs_ls.py:256-445)
    in the stage loop of non_linear_solver.py:156-211 - timed on this box's host cores: whole staged fits of the batch's
    frames, one after the other as main.py runs them, from the rest pose, until `budget_s` is used up.
    Returns dict(closures, frames, seconds, final_losses)."""
    import contextlib as _cl
    import io
    import time
    import torch
    torch.set_num_threads(int(threads))
    lay_d = 49 if use_vposer else 86
    ncl = nfr = 0
    finals = []
    t0 = time.time()
    for b in range(gt.shape[0] if max_frames is None else min(max_frames, gt.shape[0])):
        rp = RefProblem(model, cams, gt[b], conf[b], dtype, use_vposer=use_vposer, vposer_weights=vposer_weights)
        x0 = rp.get_flat() * 0.0
        assert x0.shape[0] == lay_d, x0.shape
        # scale = 1 (the only non-zero start value; order: betas, global_orient, [body_pose], transl, scale, [embedding])
        x0[(10 + 3 + (0 if use_vposer else 69) + 3)] = 1.0
        rp.set_flat(x0)
        final = None
        for wts in stages:
            rp.set_weights({k: v for k, v in wts.items() if k in ('data_weight', 'body_pose_weight', 'shape_weight',
                                                               'bending_prior_weight', 'coll_loss_weight')})
            opt = rp.make_optimizer()
            inner = rp.make_closure(opt)

            def closure(backward=True, inner=inner):
                nonlocal ncl
                ncl += 1
                return inner(backward)
            with _cl.redirect_stdout(io.StringIO()):
                final = rp.monitor.run_fitting(opt, closure, rp.final_params(), rp.smpl, use_vposer=rp.use_vposer,
                                               pose_embedding=rp.pose_embedding, vposer=rp.vposer)
        finals.append(final)
        nfr += 1
        if time.time() - t0 > budget_s:
            break
    return dict(closures=ncl, frames=nfr, seconds=time.time() - t0, final_losses=finals)


def time_reference_closure(model, cams, gt, conf, weights, use_vposer=False, vposer_weights=None, threads=1, calls=200, warm=5,
                           dtype='float32'):
    """Raw fitting_func(backward=True) rate of the reference (fitting.py:162-203) at the rest pose: mean over `calls`."""
    import time
    import torch
    torch.set_num_threads(int(threads))
    rp = RefProblem(model, cams, gt, conf, dtype, use_vposer=use_vposer, vposer_weights=vposer_weights)
    rp.set_weights({k: v for k, v in weights.items() if k in ('data_weight', 'body_pose_weight', 'shape_weight',
                                                             'bending_prior_weight', 'coll_loss_weight')})
    closure = rp.make_closure(rp.make_optimizer())
    for _ in range(warm):
        closure(backward=True)
    t0 = time.time()
    for _ in range(calls):
        closure(backward=True)
    return calls / (time.time() - t0)
