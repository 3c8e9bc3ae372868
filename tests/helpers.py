"""Shared test plumbing: rebuild golden-case inputs (seeded) and load golden outputs."""
import functools
import os

import numpy as np

from mvsmplfitting_amd import synthetic as syn
from oracle import closure_np as cn
from oracle.make_golden import CASES, stage_weights  # noqa: F401  (case table is shared)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@functools.lru_cache(maxsize=None)
def lsp_triplets():
    d = np.load(os.path.join(GOLD, 'lsp_regressor.npz'))
    return d['rows'], d['cols'], d['vals']


@functools.lru_cache(maxsize=None)
def body_model(seed=0, skin_topk=None):
    return syn.make_body_model(seed, skin_topk=skin_topk, kp_regressor=lsp_triplets())


def load_case(name):
    cfg = CASES[name]
    g = dict(np.load(os.path.join(GOLD, 'closure_%s.npz' % name)))
    model = body_model(0, cfg.get('skin_topk'))
    assert abs(syn.model_checksum(model) - float(g['model_checksum'])) < 1e-6 * float(g['model_checksum']), \
        'seeded synthetic model drifted from the one the goldens were made with'
    vpw = syn.make_vposer_decoder(**cfg['vp']) if cfg['use_vposer'] else None
    gmm = syn.make_gmm() if cfg['prior'] == 'gmm' else None
    w = g['wts']
    wts = dict(data_weight=float(w[0]), body_pose_weight=float(w[1]), shape_weight=float(w[2]),
               bending_prior_weight=float(w[3]), rho=float(w[4]))
    cams = (g['cam_R'], g['cam_t'], g['cam_f'], g['cam_c'])
    return cfg, g, model, vpw, gmm, wts, cams


def oracle_for(model, vpw, gmm, dtype=np.float64):
    return cn.ClosureOracle(model, dtype, vposer=vpw,
                            gmm=None if gmm is None else syn.gmm_constants(gmm, dtype))
