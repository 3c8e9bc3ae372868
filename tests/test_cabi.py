"""CPU: the C-ABI library loads and exports every symbol include/mvfit.h declares."""
import os
import re
import shutil
import subprocess

from mvsmplfitting_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'mvfit.h')).read()
    declared = set(re.findall(r'\b(mvfit_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'mvfit_ctx'}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_library_exports_nothing_but_the_declared_symbols():
    """The drop-in boundary is the C ABI and nothing else (csrc/libmvfit.map): no kernel host stubs, no weak STL
    instantiations that could interpose with the host's, no runtime-internal __hip_* symbols."""
    hdr = open(os.path.join(ROOT, 'include', 'mvfit.h')).read()
    declared = set(re.findall(r'\b(mvfit_[a-z0-9_]+)\s*\(', hdr)) - {'mvfit_ctx'}
    so = os.path.join(ROOT, 'mvsmplfitting_amd', 'libmvfit.so')
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    out = subprocess.run([nm, '-D', '--defined-only', so], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_header_constants_match_binding():
    hdr = open(os.path.join(ROOT, 'include', 'mvfit.h')).read()
    c = dict(re.findall(r'#define (MVFIT_[A-Z0-9_]+) (\(?-?\d+u?\)?)', hdr))
    assert int(c['MVFIT_D']) == _lib.D == 118
    assert int(c['MVFIT_F_VPOSER'].rstrip('u')) == _lib.F_VPOSER
    assert int(c['MVFIT_F_PRIOR_GMM'].rstrip('u')) == _lib.F_PRIOR_GMM
    assert int(c['MVFIT_F_FIX_SHAPE'].rstrip('u')) == _lib.F_FIX_SHAPE
    assert int(c['MVFIT_F_FIX_SCALE'].rstrip('u')) == _lib.F_FIX_SCALE
    assert int(c['MVFIT_F_SPARSE_VERTS'].rstrip('u')) == _lib.F_SPARSE_VERTS
    assert int(c['MVFIT_F_USE_3D'].rstrip('u')) == _lib.F_USE_3D


def test_options_struct_mirrors_the_header_and_the_defaults_are_the_documented_ones():
    """mvfit_options (include/mvfit.h) field by field against the ctypes mirror, and mvfit_options_default - a host function, no
    GPU needed - against the defaults the header documents (round 6: sdf_service and work_queue are on)."""
    import ctypes as C
    hdr = open(os.path.join(ROOT, 'include', 'mvfit.h')).read()
    body = hdr[hdr.index('typedef struct mvfit_options {'):hdr.index('} mvfit_options;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = re.findall(r'\b(?:u?int32_t)\s+([a-z_0-9]+)\s*;', body)
    assert fields == [f[0] for f in _lib.Options._fields_], (fields, [f[0] for f in _lib.Options._fields_])
    assert all(C.sizeof(f[1]) == 4 for f in _lib.Options._fields_)
    lib = _lib.load()
    o = _lib.Options()
    lib.mvfit_options_default(C.byref(o))
    assert o.struct_size == C.sizeof(_lib.Options)
    got = {f[0]: int(getattr(o, f[0])) for f in _lib.Options._fields_ if f[0] != 'struct_size'}
    assert got == dict(contraction=0, dense_skinning=0, round_mode=0, resident_pass=-1, sdf_two_phase=1, sdf_face_lists=1,
                       vposer_helpers=1, vposer_sets=0, closure_vposer_helpers=0, pass_kernel=0, sdf_service=1, work_queue=1), got
