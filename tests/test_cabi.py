"""CPU: the C-ABI library loads and exports every symbol include/mvfit.h declares."""
import os
import re

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
