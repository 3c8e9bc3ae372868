"""Host-side handle on libmvfit: one ``MvFit`` = one mvfit_ctx = one GPU + one stream.

PyTorch is used only as the device-memory container (tensors' ``data_ptr()`` cross the C ABI);
every number is produced by the HIP kernels in mvsmplfitting_amd/csrc.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib

D = _lib.D
D_MODEL = _lib.D_MODEL

# slices of the flat parameter vector (include/mvfit.h)
SL = dict(betas=(0, 10), global_orient=(10, 13), body_pose=(13, 82), transl=(82, 85),
          scale=(85, 86), pose_embedding=(86, 118))

# reference cfg_files/fit_smpl.yaml:40-68
YAML_POSE_W = (404.0, 404.0, 57.4, 4.78)
YAML_SHAPE_W = (100.0, 50.0, 10.0, 5.0)


class MvFitError(RuntimeError):
    pass


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def stage_weights(image_height: float, flags: int = 0, rho: float = 100.0,
                  pose_w=YAML_POSE_W, shape_w=YAML_SHAPE_W, coll_w=None):
    """Per-stage weights exactly as non_linear_solver builds them (reference
    code/utils/non_linear_solver.py:109-124,148-150,177-180): data_weight = 500/H,
    bending = 3.17 * body_pose_weight - the product rounded to float32 like the reference's float32 weight tensor."""
    out = []
    for s in range(len(pose_w)):
        out.append(dict(data_weight=500.0 / image_height, body_pose_weight=pose_w[s],
                        shape_weight=shape_w[s], bending_prior_weight=float(np.float32(3.17) * np.float32(pose_w[s])),
                        coll_loss_weight=0.0 if coll_w is None else coll_w[s], rho=rho, flags=flags))
    return out


_SDF_WALK_WARNING = ('%s walked over every face (about ten times slower, same bits): the workspace of the face lists did not '
                     'fit in half of the free device memory (is the GPU shared?)')

# Options every new engine starts from (field names of include/mvfit.h:mvfit_options; empty = the library's defaults).  The
# reference-seam mirror (fitting.py) builds its engine internally, so callers that need another default for a whole process
# (tests, bench.py's named configs) set it here; per-engine values go to MvFit(..., options=...) / set_options().
DEFAULT_OPTIONS: dict = {}
CONTRACTIONS = dict(split_fp16=_lib.CONTRACTION_SPLIT_FP16, exact_fp32=_lib.CONTRACTION_EXACT_FP32,
                    half_basis=_lib.CONTRACTION_HALF_BASIS)


class MvFit:
    def __init__(self, model: dict, vposer: dict | None = None, gmm=None, device: int = 0, options: dict | None = None,
                 library: str | None = None):
        """model: dict from mvsmplfitting_amd.synthetic.make_body_model (or real SMPL arrays with the
        same keys); vposer: decoder weight dict; gmm: (means, precisions, nll_weights); options: mvfit_options fields
        (``contraction`` also by name: 'split_fp16' | 'exact_fp32' | 'half_basis'); library: another build of libmvfit."""
        if not torch.cuda.is_available():
            raise MvFitError('MvFit needs a HIP device (torch.cuda.is_available() is False); '
                             'there is no CPU fallback')
        self._lib = _lib.load(library)
        self.device = torch.device('cuda', device)
        self.dtype = torch.float32
        self.nv = int(model['v_template'].shape[0])
        keep = []

        def fp(a):
            a = _f32(a)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_float))

        def ip(a):
            a = _i32(a)
            keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_int32))
        m = _lib.Model()
        m.num_verts = self.nv
        m.num_faces = int(model['faces'].shape[0]) if model.get('faces') is not None else 0
        m.v_template = fp(model['v_template'])
        m.shapedirs = fp(model['shapedirs'])
        m.posedirs = fp(model['posedirs'])
        m.J_regressor = fp(model['J_regressor'])
        m.parents = ip(model['parents'])
        m.lbs_weights = fp(model['lbs_weights'])
        m.kp_regressor = fp(model['kp_regressor'])
        m.face_vertex_ids = ip(model['face_vertex_ids'])
        m.joint_map = ip(model['joint_map'])
        m.faces = ip(model['faces']) if model.get('faces') is not None else None
        if vposer is not None:
            m.vp_fc1_w = fp(vposer['fc1_w']); m.vp_fc1_b = fp(vposer['fc1_b'])
            m.vp_fc2_w = fp(vposer['fc2_w']); m.vp_fc2_b = fp(vposer['fc2_b'])
            m.vp_out_w = fp(vposer['out_w']); m.vp_out_b = fp(vposer['out_b'])
        if gmm is not None:
            means, prec, nllw = gmm
            m.gmm_M = int(means.shape[0])
            m.gmm_means = fp(means); m.gmm_precisions = fp(prec); m.gmm_nll_weights = fp(nllw)
        self.has_vposer = vposer is not None
        self.has_gmm = gmm is not None
        ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            o = self._options_struct(dict(DEFAULT_OPTIONS, **(options or {})))
            rc = self._lib.mvfit_create_ex(C.byref(ctx), device, C.c_void_p(stream), C.byref(m), C.byref(o))
        self._ctx = ctx
        self._check(rc)
        self.B = 0
        self.V = 0

    # ------------------------------------------------------------------ options (include/mvfit.h:mvfit_options)
    def _options_struct(self, values: dict, base=None):
        o = _lib.Options()
        if base is None:
            self._lib.mvfit_options_default(C.byref(o))
        else:
            C.memmove(C.byref(o), C.byref(base), C.sizeof(o))
        names = {f[0] for f in _lib.Options._fields_} - {'struct_size'}
        for k, v in values.items():
            if k not in names:
                raise MvFitError('unknown option %r (mvfit_options has %s)' % (k, sorted(names)))
            if k == 'contraction' and isinstance(v, str):
                v = CONTRACTIONS[v]
            setattr(o, k, int(v))
        return o

    def options(self) -> dict:
        o = _lib.Options()
        self._check(self._lib.mvfit_get_options(self._ctx, C.byref(o)))
        return {f[0]: int(getattr(o, f[0])) for f in _lib.Options._fields_ if f[0] != 'struct_size'}

    def set_options(self, **values):
        """Change run-time selectors between calls (round_mode, resident_pass, sdf_two_phase, sdf_face_lists, vposer_helpers,
        vposer_sets, closure_vposer_helpers, pass_kernel, sdf_service); returns the previous values of the ones changed."""
        cur = _lib.Options()
        self._check(self._lib.mvfit_get_options(self._ctx, C.byref(cur)))
        old = {k: int(getattr(cur, k)) for k in values}
        self._check(self._lib.mvfit_set_options(self._ctx, C.byref(self._options_struct(values, base=cur))))
        return old

    def sdf_info(self) -> dict:
        """Which path served the last sdf() call / the SDF term of the last fit: 'walk', 'face_lists', or
        'walk_workspace_did_not_fit' (include/mvfit.h:mvfit_sdf_info)."""
        a, b = C.c_int(), C.c_int()
        self._check(self._lib.mvfit_sdf_info(self._ctx, C.byref(a), C.byref(b)))
        names = ('walk', 'face_lists', 'walk_workspace_did_not_fit')
        return dict(op=names[a.value], term=names[b.value])

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.mvfit_last_error(self._ctx)
            raise MvFitError('libmvfit error %d: %s' % (rc, msg.decode() if msg else '?'))

    def close(self):
        if getattr(self, '_ctx', None):
            self._lib.mvfit_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _dev(self, a, shape=None):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(_f32(a))
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None:
            assert tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)
        return t

    @staticmethod
    def _weights(w: dict) -> _lib.Weights:
        return _lib.Weights(w['data_weight'], w['body_pose_weight'], w['shape_weight'],
                            w['bending_prior_weight'], w.get('coll_loss_weight', 0.0),
                            w.get('rho', 100.0), int(w.get('flags', 0)))

    def sync(self):
        self._check(self._lib.mvfit_sync(self._ctx))

    # ------------------------------------------------------------------ inputs
    def set_problems(self, cams, gt_xy, w_conf):
        """cams = (R[V,3,3] | [B,V,3,3], t, f, c); gt_xy[B,V,17,2]; w_conf[B,V,17]."""
        cam_R, cam_t, cam_f, cam_c = cams
        gt = self._dev(gt_xy)
        B, V = int(gt.shape[0]), int(gt.shape[1])
        wc = self._dev(w_conf, (B, V, 17))
        if B != self.B or V != self.V:           # hook buffers were sized for the old batch (the C side drops them too)
            if getattr(self, '_trace', None) is not None:
                self.fit_trace(0)
            if getattr(self, '_capture', None) is not None:
                self.capture_pass(None)
        batched = 1 if np.ndim(cam_R) == 4 else 0
        R, t, f, c = self._dev(cam_R), self._dev(cam_t), self._dev(cam_f), self._dev(cam_c)
        self._check(self._lib.mvfit_set_problems(
            self._ctx, B, V, batched, R.data_ptr(), t.data_ptr(), f.data_ptr(), c.data_ptr(),
            gt.data_ptr(), wc.data_ptr()))
        self.B, self.V = B, V

    def set_joints3d(self, gt3d, conf3d):
        """use_3d targets: gt3d[B,17,3], conf3d[B,17] (reference non_linear_solver.py:86-99)."""
        g = self._dev(gt3d, (self.B, 17, 3))
        c = self._dev(conf3d, (self.B, 17))
        self._check(self._lib.mvfit_set_joints3d(self._ctx, g.data_ptr(), c.data_ptr()))

    # ------------------------------------------------------------------ compute
    def closure(self, params, weights: dict, want_grad=True, want_verts=False, want_joints=False):
        """One closure evaluation of all B problems.  params [B,118] (tensor or array).
        Returns dict(loss[B], grad[B,118]?, verts[B,Nv,3]?, joints[B,17,3]?) of CUDA tensors."""
        x = self._dev(params, (self.B, D))
        out = dict(loss=torch.empty(self.B, device=self.device, dtype=torch.float32))
        grad = torch.empty(self.B, D, device=self.device) if want_grad else None
        verts = torch.empty(self.B, self.nv, 3, device=self.device) if want_verts else None
        joints = torch.empty(self.B, 17, 3, device=self.device) if want_joints else None
        w = self._weights(weights)
        self._check(self._lib.mvfit_closure(
            self._ctx, C.byref(w), x.data_ptr(), out['loss'].data_ptr(),
            grad.data_ptr() if want_grad else None, verts.data_ptr() if want_verts else None,
            joints.data_ptr() if want_joints else None))
        if want_grad:
            out['grad'] = grad
        if want_verts:
            out['verts'] = verts
        if want_joints:
            out['joints'] = joints
        return out

    def vertices(self, params, flags=0):
        x = self._dev(params, (self.B, D))
        verts = torch.empty(self.B, self.nv, 3, device=self.device)
        joints = torch.empty(self.B, 17, 3, device=self.device)
        self._check(self._lib.mvfit_vertices(self._ctx, x.data_ptr(), flags, verts.data_ptr(),
                                             joints.data_ptr()))
        return verts, joints

    def full_pose(self, params, flags=0):
        """ModelOutput.full_pose [B,72] = global_orient | body_pose (decoded from the embedding with F_VPOSER):
        include/mvfit.h:mvfit_full_pose."""
        x = self._dev(params, (self.B, D))
        out = torch.empty(self.B, 72, device=self.device)
        self._check(self._lib.mvfit_full_pose(self._ctx, x.data_ptr(), int(flags), out.data_ptr()))
        return out

    def fit(self, params, stages, lr=1.0, max_iter=30, history=100, tolerance_grad=1e-5,
            tolerance_change=1e-9, maxiters=30, ftol=1e-9, gtol=1e-9, max_rounds=0):
        """Device-resident staged fit.  params [B,118] -> (params_out tensor, stats dict)."""
        x = self._dev(params, (self.B, D)).clone()
        arr = (_lib.Weights * len(stages))(*[self._weights(s) for s in stages])
        o = _lib.LbfgsOpts(lr, max_iter, history, tolerance_grad, tolerance_change, maxiters, ftol,
                           gtol, len(stages), max_rounds)
        final = torch.empty(self.B, device=self.device)
        ncl = torch.empty(self.B, device=self.device, dtype=torch.int32)
        nit = torch.empty(self.B, device=self.device, dtype=torch.int32)
        rc = self._lib.mvfit_fit(self._ctx, arr, C.byref(o), x.data_ptr(), final.data_ptr(),
                                 ncl.data_ptr(), nit.data_ptr())
        self._check(rc)
        st4 = (C.c_uint32 * 4)()
        self._check(self._lib.mvfit_fit_stats(self._ctx, st4))
        stats = dict(final_loss=final, n_closure=ncl, n_iter=nit,
                     passes=dict(run=int(st4[0]), skipped=int(st4[1]), missed=int(st4[2]), timed_out=int(st4[3])))
        if stats['passes']['missed'] or stats['passes']['timed_out']:
            # never silent: "every closure round got its full vertex pass" does not hold for this fit (the fitted parameters
            # do not depend on the passes - they consume what the optimiser publishes - but the per-round vertices do)
            warnings.warn('vertex passes degraded in this fit: %d lost their operands before reading them, %d gave up waiting '
                          'for them (is the GPU shared, or were the pass workgroups not all resident?)'
                          % (stats['passes']['missed'], stats['passes']['timed_out']), RuntimeWarning)
        if any(float(dict(s).get('coll_loss_weight', 0.0)) > 0.0 for s in stages) and \
                self.sdf_info()['term'] == 'walk_workspace_did_not_fit':
            warnings.warn(_SDF_WALK_WARNING % 'the SDF term of this fit', RuntimeWarning)
        if any(int(dict(s).get('flags', 0)) & _lib.F_VPOSER for s in stages):
            # decoder helpers (vposer_service.h): an answer that timed out makes that problem decode in its own workgroup
            # from then on - another summation order, i.e. last-bit differences from run to run.  Never silent.
            dec = self.decoder_stats()
            stats['decoder'] = dec
            if dec['answers_timed_out'] or dec['helpers_gave_up']:
                warnings.warn('VPoser decoder helpers degraded in this fit (%d answers timed out, %d helpers gave up): the '
                              'affected problems decoded locally (results differ in the last bits; is the GPU shared?)'
                              % (dec['answers_timed_out'], dec['helpers_gave_up']), RuntimeWarning)
        return x, stats

    def decoder_stats(self):
        """Counters of the VPoser decoder helpers of the last fit (include/mvfit.h:mvfit_decoder_stats): launches that
        carried helpers, answers that timed out, helpers that gave up (both expected 0).  Waits for the fit."""
        st3 = (C.c_uint32 * 3)()
        self._check(self._lib.mvfit_decoder_stats(self._ctx, st3))
        return dict(launches=int(st3[0]), answers_timed_out=int(st3[1]), helpers_gave_up=int(st3[2]))

    def fit_trace(self, max_closures=0):
        """Record (x_trial[118], loss) of the first ``max_closures`` closure calls of every problem during the next
        fits (include/mvfit.h:mvfit_fit_trace); returns the [B, max_closures, 119] tensor (NaN where nothing was
        written).  ``max_closures=0`` switches tracing off."""
        if max_closures <= 0:
            self._check(self._lib.mvfit_fit_trace(self._ctx, None, 0))
            self._trace = None
            return None
        self._trace = torch.full((self.B, int(max_closures), D + 1), float('nan'), device=self.device)
        self._check(self._lib.mvfit_fit_trace(self._ctx, self._trace.data_ptr(), int(max_closures)))
        return self._trace

    def capture_pass(self, round_index=None):
        """Test hook (include/mvfit.h:mvfit_debug_capture_pass): the vertex pass of closure round ``round_index`` of
        the next asynchronous fits writes into the returned [B, Nv, 3] tensor.  None switches it off."""
        if round_index is None:
            self._check(self._lib.mvfit_debug_capture_pass(self._ctx, -1, None))
            self._capture = None
            return None
        self._capture = torch.full((self.B, self.nv, 3), float('nan'), device=self.device)
        self._check(self._lib.mvfit_debug_capture_pass(self._ctx, int(round_index), self._capture.data_ptr()))
        return self._capture

    def sdf(self, faces, vertices, grid_size=32):
        """phi[B,G,G,G] of the SDF voxelisation op (include/mvfit.h:mvfit_sdf).  faces: int tensor whose
        size(0) is taken as the number of triangles, like the reference binding does."""
        f = faces if isinstance(faces, torch.Tensor) else torch.as_tensor(np.asarray(faces))
        f = f.to(device=self.device, dtype=torch.int32).contiguous()
        v = self._dev(vertices)
        if v.dim() != 3 or v.shape[2] != 3:
            raise MvFitError('vertices must be [B, Nv, 3]')
        phi = torch.zeros(v.shape[0], grid_size, grid_size, grid_size, device=self.device)
        self._check(self._lib.mvfit_sdf(self._ctx, f.data_ptr(), int(f.shape[0]), v.data_ptr(), int(v.shape[0]),
                                        int(v.shape[1]), int(grid_size), phi.data_ptr()))
        if self.sdf_info()['op'] == 'walk_workspace_did_not_fit':
            warnings.warn(_SDF_WALK_WARNING % 'this mvfit_sdf call', RuntimeWarning)
        return phi

    def set_sdf(self, faces, num_faces=1, grid_size=128):
        """Configure the interpenetration term (include/mvfit.h:mvfit_set_sdf).  ``faces`` [F,3]; ``num_faces``
        = how many leading triangles the op sees: 1 reproduces the reference's call site
        (faces.reshape(1,-1,3), fitting.py:367-368), None = all of them.  faces=None removes the term."""
        if faces is None:
            self._check(self._lib.mvfit_set_sdf(self._ctx, None, 0, 0))
            return
        f = faces.detach().cpu().numpy() if isinstance(faces, torch.Tensor) else np.asarray(faces)
        f = np.ascontiguousarray(f.reshape(-1, 3), dtype=np.int32)
        n = f.shape[0] if num_faces is None else int(num_faces)
        if n < 1 or n > f.shape[0]:
            raise MvFitError(f'num_faces={n} outside [1, {f.shape[0]}]')
        self._check(self._lib.mvfit_set_sdf(self._ctx, f.ctypes.data, n, int(grid_size)))

    def triangulate(self, keypoints, intris, extris):
        """joints3d [B,17,3] float64 from keypoints [B,V,17,3] (u, v, confidence) and the float64 camera matrices
        intris [V,3,3], extris [V,4,4] (include/mvfit.h:mvfit_triangulate; reference recompute3D)."""
        kp = self._dev(keypoints)
        if kp.dim() != 4 or kp.shape[2] != 17 or kp.shape[3] != 3:
            raise MvFitError('keypoints must be [B, V, 17, 3]')
        B, V = int(kp.shape[0]), int(kp.shape[1])
        K = torch.as_tensor(np.asarray(intris, np.float64) if not isinstance(intris, torch.Tensor) else intris,
                            dtype=torch.float64, device=self.device).contiguous()
        E = torch.as_tensor(np.asarray(extris, np.float64) if not isinstance(extris, torch.Tensor) else extris,
                            dtype=torch.float64, device=self.device).contiguous()
        if tuple(K.shape) != (V, 3, 3) or tuple(E.shape) != (V, 4, 4):
            raise MvFitError('intris must be [V,3,3] and extris [V,4,4] with V = %d' % V)
        out = torch.empty(B, 17, 3, dtype=torch.float64, device=self.device)
        self._check(self._lib.mvfit_triangulate(self._ctx, B, V, kp.data_ptr(), K.data_ptr(), E.data_ptr(), out.data_ptr()))
        return out

    def depth_guess(self, rest_joints, extri, intri, keypoints):
        """joints3d [B,17,3] float64 for frames seen by ONE camera (include/mvfit.h:mvfit_depth_guess; reference
        init_guess.py:54-74): rest_joints [17,3] float64, extri [4,4], intri [3,3] float64, keypoints [B,17,3]
        (u, v, confidence)."""
        kp = self._dev(keypoints)
        if kp.dim() != 3 or kp.shape[1] != 17 or kp.shape[2] != 3:
            raise MvFitError('keypoints must be [B, 17, 3]')
        f64 = lambda a, shape: self._f64(a, shape)
        R_, E_, K_ = f64(rest_joints, (17, 3)), f64(extri, (4, 4)), f64(intri, (3, 3))
        B = int(kp.shape[0])
        out = torch.empty(B, 17, 3, dtype=torch.float64, device=self.device)
        self._check(self._lib.mvfit_depth_guess(self._ctx, B, R_.data_ptr(), E_.data_ptr(), K_.data_ptr(), kp.data_ptr(),
                                                out.data_ptr()))
        return out

    def _f64(self, a, shape):
        t = torch.as_tensor(np.asarray(a, np.float64) if not isinstance(a, torch.Tensor) else a, dtype=torch.float64,
                            device=self.device).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise MvFitError('expected shape %s, got %s' % (tuple(shape), tuple(t.shape)))
        return t

    def umeyama(self, src, dst, estimate_scale=True):
        """The reference's similarity alignment + cv2.Rodrigues (include/mvfit.h:mvfit_umeyama): src [npts,3],
        dst [B,npts,3] float64 -> dict(rot [B,3,3], rvec [B,3], trans [B,3], scale [B]) float64 tensors."""
        s_ = torch.as_tensor(np.asarray(src, np.float64) if not isinstance(src, torch.Tensor) else src,
                             dtype=torch.float64, device=self.device).contiguous()
        d_ = torch.as_tensor(np.asarray(dst, np.float64) if not isinstance(dst, torch.Tensor) else dst,
                             dtype=torch.float64, device=self.device).contiguous()
        if s_.dim() != 2 or d_.dim() != 3 or d_.shape[1:] != s_.shape or s_.shape[1] != 3:
            raise MvFitError('src must be [npts,3] and dst [B,npts,3]')
        B, npts = int(d_.shape[0]), int(s_.shape[0])
        out = dict(rot=torch.empty(B, 3, 3, dtype=torch.float64, device=self.device),
                   rvec=torch.empty(B, 3, dtype=torch.float64, device=self.device),
                   trans=torch.empty(B, 3, dtype=torch.float64, device=self.device),
                   scale=torch.empty(B, dtype=torch.float64, device=self.device))
        self._check(self._lib.mvfit_umeyama(self._ctx, B, npts, s_.data_ptr(), d_.data_ptr(), 1 if estimate_scale else 0,
                                            out['rot'].data_ptr(), out['rvec'].data_ptr(), out['trans'].data_ptr(),
                                            out['scale'].data_ptr()))
        return out

    def project(self, points):
        """uv [B, V, N, 2] = every view's pinhole projection of points [B, N, 3] with the cameras of set_problems
        (include/mvfit.h:mvfit_project_points; reference cam(verts), utils/utils.py:603-607)."""
        p = self._dev(points)
        if p.dim() != 3 or p.shape[0] != self.B or p.shape[2] != 3:
            raise MvFitError('points must be [B, N, 3] with B = %d' % self.B)
        uv = torch.empty(self.B, self.V, int(p.shape[1]), 2, device=self.device)
        self._check(self._lib.mvfit_project_points(self._ctx, p.data_ptr(), int(p.shape[1]), uv.data_ptr()))
        return uv

    def gather(self, rccl_comm, send, nranks):
        """All-gather of a contiguous device tensor over a raw RCCL communicator (include/mvfit.h:mvfit_gather) - the
        entry point of hosts that own an ncclComm_t; the Python adapters use torch.distributed (sharding.py), which does
        not hand out its communicator.  rccl_comm: the ncclComm_t as an integer / c_void_p.  -> [nranks, *send.shape]."""
        s_ = send.contiguous()
        if not s_.is_cuda:
            raise MvFitError('send must be a device tensor')
        recv = torch.empty((int(nranks),) + tuple(s_.shape), dtype=s_.dtype, device=s_.device)
        self._check(self._lib.mvfit_gather(self._ctx, rccl_comm, s_.data_ptr(), recv.data_ptr(), s_.numel() * s_.element_size()))
        return recv

    def sdf_term_read(self):
        """(samples [B,Nv,4] = phi_v and its local-coordinate gradient, S [B]) of the last evaluated term."""
        smp = torch.empty(self.B, self.nv, 4, device=self.device)
        S = torch.empty(self.B, device=self.device)
        self._check(self._lib.mvfit_sdf_term_read(self._ctx, smp.data_ptr(), S.data_ptr()))
        return smp, S

    # ------------------------------------------------------------------ profiling
    def profile(self, enable=True):
        self._check(self._lib.mvfit_profile(self._ctx, 1 if enable else 0))

    def profile_vertex_pass_ms(self, launches=64, as_in_async_fit=False):
        """Average duration (ms) of `launches` back-to-back vertex-pass launches inside one hipEvent pair;
        ``as_in_async_fit``: the launch flavour of the asynchronous fit (include/mvfit.h)."""
        a = C.c_double()
        self._check(self._lib.mvfit_profile_vertex_pass_ex(self._ctx, int(launches), 1 if as_in_async_fit else 0, C.byref(a)))
        return a.value

    def profile_resident_pass_ms(self, rounds=100):
        """Per-round time (ms) of the RESIDENT vertex pass alone: one launch serving ``rounds`` (<= 128) closure rounds whose
        operands the last asynchronous fit left in the ring, inside one hipEvent pair (include/mvfit.h, flavour 2)."""
        a = C.c_double()
        self._check(self._lib.mvfit_profile_vertex_pass_ex(self._ctx, int(rounds), 2, C.byref(a)))
        return a.value

    def pass_profile(self):
        """How the vertex passes of the last asynchronous fit ran (include/mvfit.h:mvfit_pass_profile)."""
        tpw, wgs, n = C.c_int(), C.c_int(), C.c_int()
        span, busy, slowest = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.mvfit_pass_profile(self._ctx, C.byref(tpw), C.byref(wgs), C.byref(n), C.byref(span), C.byref(busy),
                                                 C.byref(slowest)))
        form = tpw.value               # 0 per-round launches; 1 one tile per workgroup; 3 two tiles, role-split (form 2 was dropped)
        kernel = {0: None, 1: 'lbs_vertex_pass_resident_kernel<1>', 3: 'lbs_vertex_pass_resident_roles_kernel'}[form]
        return dict(form=form, tiles_per_workgroup=2 if form == 3 else form, kernel=kernel, workgroups=wgs.value, rounds_stamped=n.value,
                    round_span_ms=span.value, workgroup_busy_ms=busy.value, slowest_workgroup_ms=slowest.value)

    def profile_read(self):
        a, b = C.c_double(), C.c_double()
        n1, n2 = C.c_int(), C.c_int()
        self._check(self._lib.mvfit_profile_read(self._ctx, C.byref(a), C.byref(n1), C.byref(b),
                                                 C.byref(n2)))
        return dict(vertex_pass_ms=a.value, vertex_pass_launches=n1.value,
                    step_ms=b.value, step_launches=n2.value)


def pack_params(betas=None, global_orient=None, body_pose=None, transl=None, scale=None,
                pose_embedding=None, B=1):
    """Assemble [B,118] float32 from per-tensor arrays (missing ones: zeros, scale ones)."""
    x = np.zeros((B, D), np.float32)
    x[:, 85] = 1.0
    for name, val in (('betas', betas), ('global_orient', global_orient), ('body_pose', body_pose),
                      ('transl', transl), ('scale', scale), ('pose_embedding', pose_embedding)):
        if val is not None:
            a, b = SL[name]
            x[:, a:b] = np.asarray(val, np.float32).reshape(B, b - a)
    return x


def lbfgs_kat(kind: int, D_: int, segs, x0, max_trace=80, device=0, **opts):
    """Run the float64 device L-BFGS on an analytic objective (include/mvfit.h:mvfit_lbfgs_kat)."""
    lib = _lib.load()
    o = _lib.LbfgsOpts(opts.get('lr', 1.0), opts.get('max_iter', 30), opts.get('history', 100),
                       opts.get('tolerance_grad', 1e-5), opts.get('tolerance_change', 1e-9),
                       opts.get('maxiters', 30), opts.get('ftol', 1e-9), opts.get('gtol', 1e-9), 1, 0)
    x = np.ascontiguousarray(x0, np.float64).copy()
    trace = np.zeros((max_trace, D_ + 1), np.float64)
    sg = _i32(segs)
    n = C.c_int()
    fl = C.c_double()
    rc = lib.mvfit_lbfgs_kat(device, kind, D_, sg.ctypes.data_as(C.POINTER(C.c_int32)), len(sg) - 1,
                             C.byref(o), x.ctypes.data_as(C.POINTER(C.c_double)),
                             trace.ctypes.data_as(C.POINTER(C.c_double)), max_trace, C.byref(n),
                             C.byref(fl))
    if rc != 0:
        raise MvFitError('mvfit_lbfgs_kat failed: %d' % rc)
    return x, trace[:min(n.value, max_trace)], n.value, fl.value
