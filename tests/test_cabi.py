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
