"""ctypes binding of libmvfit.so (include/mvfit.h).  No fallback: if the HIP library is
missing or does not load, importing this module raises - the product path never runs on
anything but the hand-written gfx950 kernels."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVFIT_LIBRARY (the only environment variable of the product, read by this loader, never by the library): developer builds of
# the same library - the -DMVFIT_TIMING / -DMVFIT_DEBUG_HOOKS variants the timing tools and the fault-injection test load
LIB_PATH = os.environ.get('MVFIT_LIBRARY') or os.path.join(_HERE, 'libmvfit.so')

D = 118
D_MODEL = 86
NUM_KP = 17
MAX_VIEWS = 16
MAX_STAGES = 8

F_VPOSER = 1
F_PRIOR_GMM = 2
F_FIX_SHAPE = 4
F_FIX_SCALE = 8
F_SPARSE_VERTS = 16
F_USE_3D = 32
F_REUSE_OUTER_VALUE = 64

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class Model(C.Structure):
    _fields_ = [
        ('num_verts', C.c_int32), ('num_faces', C.c_int32),
        ('v_template', _fp), ('shapedirs', _fp), ('posedirs', _fp), ('J_regressor', _fp),
        ('parents', _ip), ('lbs_weights', _fp), ('kp_regressor', _fp),
        ('face_vertex_ids', _ip), ('joint_map', _ip), ('faces', _ip),
        ('vp_fc1_w', _fp), ('vp_fc1_b', _fp), ('vp_fc2_w', _fp), ('vp_fc2_b', _fp),
        ('vp_out_w', _fp), ('vp_out_b', _fp),
        ('gmm_M', C.c_int32), ('gmm_means', _fp), ('gmm_precisions', _fp), ('gmm_nll_weights', _fp),
    ]


class Weights(C.Structure):
    _fields_ = [('data_weight', C.c_float), ('body_pose_weight', C.c_float),
                ('shape_weight', C.c_float), ('bending_prior_weight', C.c_float),
                ('coll_loss_weight', C.c_float), ('rho', C.c_float), ('flags', C.c_uint32)]


class LbfgsOpts(C.Structure):
    _fields_ = [('lr', C.c_float), ('max_iter', C.c_int32), ('history', C.c_int32),
                ('tolerance_grad', C.c_float), ('tolerance_change', C.c_float),
                ('maxiters', C.c_int32), ('ftol', C.c_float), ('gtol', C.c_float),
                ('num_stages', C.c_int32), ('max_rounds', C.c_int32)]


CONTRACTION_SPLIT_FP16, CONTRACTION_EXACT_FP32, CONTRACTION_HALF_BASIS = 0, 1, 2


class Options(C.Structure):
    """mvfit_options (include/mvfit.h): precision and path selectors - the library reads no environment variable."""
    _fields_ = [('struct_size', C.c_uint32), ('contraction', C.c_int32), ('dense_skinning', C.c_int32),
                ('round_mode', C.c_int32), ('resident_pass', C.c_int32), ('sdf_two_phase', C.c_int32),
                ('sdf_face_lists', C.c_int32), ('vposer_helpers', C.c_int32), ('vposer_sets', C.c_int32),
                ('closure_vposer_helpers', C.c_int32), ('pass_kernel', C.c_int32), ('sdf_service', C.c_int32), ('work_queue', C.c_int32)]


EXPORTS = ['mvfit_create', 'mvfit_destroy', 'mvfit_last_error', 'mvfit_sync', 'mvfit_set_problems', 'mvfit_set_joints3d',
           'mvfit_closure', 'mvfit_vertices', 'mvfit_full_pose', 'mvfit_fit', 'mvfit_fit_trace', 'mvfit_fit_stats', 'mvfit_decoder_stats', 'mvfit_debug_capture_pass', 'mvfit_sdf', 'mvfit_set_sdf', 'mvfit_sdf_term_read', 'mvfit_triangulate', 'mvfit_depth_guess', 'mvfit_umeyama', 'mvfit_project_points', 'mvfit_gather', 'mvfit_profile', 'mvfit_profile_read', 'mvfit_profile_vertex_pass', 'mvfit_profile_vertex_pass_ex', 'mvfit_pass_profile',
           'mvfit_options_default', 'mvfit_create_ex', 'mvfit_set_options', 'mvfit_get_options', 'mvfit_sdf_info',
           'mvfit_lbfgs_kat']


def load(path=None):
    # libmvfit.so is linked against libamdhip64.so.7; PyTorch ships its own copy under the same
    # SONAME.  Exactly one HIP runtime may live in the process (tensors and kernels must share a
    # context), so torch's is loaded first and the dynamic loader binds libmvfit to it.
    import torch  # noqa: F401
    path = path or LIB_PATH
    if not os.path.isfile(path):
        raise ImportError('libmvfit.so not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                          'or `make -C mvsmplfitting_amd/csrc` (expected %s)' % path)
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.mvfit_create.argtypes = [C.POINTER(vp), C.c_int, vp, C.POINTER(Model)]
    lib.mvfit_create.restype = C.c_int
    lib.mvfit_options_default.argtypes = [C.POINTER(Options)]
    lib.mvfit_options_default.restype = None
    lib.mvfit_create_ex.argtypes = [C.POINTER(vp), C.c_int, vp, C.POINTER(Model), C.POINTER(Options)]
    lib.mvfit_create_ex.restype = C.c_int
    lib.mvfit_set_options.argtypes = [vp, C.POINTER(Options)]
    lib.mvfit_set_options.restype = C.c_int
    lib.mvfit_get_options.argtypes = [vp, C.POINTER(Options)]
    lib.mvfit_get_options.restype = C.c_int
    lib.mvfit_sdf_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mvfit_sdf_info.restype = C.c_int
    lib.mvfit_destroy.argtypes = [vp]
    lib.mvfit_destroy.restype = None
    lib.mvfit_last_error.argtypes = [vp]
    lib.mvfit_last_error.restype = C.c_char_p
    lib.mvfit_sync.argtypes = [vp]
    lib.mvfit_sync.restype = C.c_int
    lib.mvfit_set_problems.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.mvfit_set_problems.restype = C.c_int
    lib.mvfit_set_joints3d.argtypes = [vp, vp, vp]
    lib.mvfit_set_joints3d.restype = C.c_int
    lib.mvfit_closure.argtypes = [vp, C.POINTER(Weights), vp, vp, vp, vp, vp]
    lib.mvfit_closure.restype = C.c_int
    lib.mvfit_vertices.argtypes = [vp, vp, C.c_uint32, vp, vp]
    lib.mvfit_vertices.restype = C.c_int
    lib.mvfit_fit.argtypes = [vp, C.POINTER(Weights), C.POINTER(LbfgsOpts), vp, vp, vp, vp]
    lib.mvfit_fit.restype = C.c_int
    lib.mvfit_debug_capture_pass.argtypes = [vp, C.c_int, vp]
    lib.mvfit_debug_capture_pass.restype = C.c_int
    lib.mvfit_fit_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.mvfit_fit_stats.restype = C.c_int
    lib.mvfit_decoder_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.mvfit_decoder_stats.restype = C.c_int
    lib.mvfit_fit_trace.argtypes = [vp, vp, C.c_int]
    lib.mvfit_fit_trace.restype = C.c_int
    lib.mvfit_sdf.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.mvfit_sdf.restype = C.c_int
    lib.mvfit_set_sdf.argtypes = [vp, vp, C.c_int, C.c_int]
    lib.mvfit_set_sdf.restype = C.c_int
    lib.mvfit_sdf_term_read.argtypes = [vp, vp, vp]
    lib.mvfit_sdf_term_read.restype = C.c_int
    lib.mvfit_triangulate.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.mvfit_triangulate.restype = C.c_int
    lib.mvfit_depth_guess.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.mvfit_depth_guess.restype = C.c_int
    lib.mvfit_umeyama.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.mvfit_umeyama.restype = C.c_int
    lib.mvfit_project_points.argtypes = [vp, vp, C.c_int, vp]
    lib.mvfit_project_points.restype = C.c_int
    lib.mvfit_full_pose.argtypes = [vp, vp, C.c_uint32, vp]
    lib.mvfit_full_pose.restype = C.c_int
    lib.mvfit_gather.argtypes = [vp, vp, vp, vp, C.c_size_t]
    lib.mvfit_gather.restype = C.c_int
    lib.mvfit_profile.argtypes = [vp, C.c_int]
    lib.mvfit_profile.restype = C.c_int
    lib.mvfit_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.mvfit_profile_read.restype = C.c_int
    lib.mvfit_profile_vertex_pass.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
    lib.mvfit_profile_vertex_pass.restype = C.c_int
    lib.mvfit_profile_vertex_pass_ex.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.mvfit_profile_vertex_pass_ex.restype = C.c_int
    lib.mvfit_pass_profile.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.mvfit_pass_profile.restype = C.c_int
    lib.mvfit_lbfgs_kat.argtypes = [C.c_int, C.c_int, C.c_int, _ip, C.c_int, C.POINTER(LbfgsOpts),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.mvfit_lbfgs_kat.restype = C.c_int
    return lib
