"""Helpers shared by the -m gpu tests: golden layouts (D=86 / D=49) <-> the C ABI's D=118."""
import numpy as np

from mvsmplfitting_amd import _lib
from mvsmplfitting_amd import synthetic as syn


def to118(x, use_vposer, fix_shape_betas=None):
    x = np.asarray(x, np.float64)
    out = np.zeros(118)
    out[85] = 1.0
    if use_vposer:
        out[0:10] = x[0:10]; out[10:13] = x[10:13]; out[82:85] = x[13:16]; out[85] = x[16]
        out[86:118] = x[17:49]
    else:
        out[0:86] = x[0:86]
    return out


def from118(g, use_vposer):
    g = np.asarray(g, np.float64)
    if use_vposer:
        return np.concatenate([g[0:10], g[10:13], g[82:85], g[85:86], g[86:118]])
    return g[0:86].copy()


def flags_for(cfg):
    f = 0
    if cfg['use_vposer']:
        f |= _lib.F_VPOSER
    if cfg['prior'] == 'gmm':
        f |= _lib.F_PRIOR_GMM
    if cfg.get('fix_shape'):
        f |= _lib.F_FIX_SHAPE
    if cfg.get('use_3d'):
        f |= _lib.F_USE_3D
    return f


def make_engine(model, vpw=None, gmm=None, **options):
    """An engine; options = fields of include/mvfit.h:mvfit_options (the library reads no environment variable)."""
    from mvsmplfitting_amd.engine import MvFit
    return MvFit(model, vposer=vpw, gmm=None if gmm is None else syn.gmm_constants(gmm, np.float32), options=options)
