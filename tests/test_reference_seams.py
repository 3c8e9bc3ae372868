"""CPU (build container only): the reference's own SMPL module exposes exactly the buffers that
mvsmplfitting_amd.fitting.model_arrays() hands to libmvfit, and the mirror's signatures accept the
reference call sites' keyword arguments."""
import inspect

import numpy as np
import pytest

from oracle import ref_import
from mvsmplfitting_amd import synthetic as syn

pytestmark = pytest.mark.skipif(not ref_import.available(), reason='reference tree not mounted')


def test_model_arrays_from_reference_module():
    from mvsmplfitting_amd import fitting as mf
    from tests.helpers import body_model
    model = body_model()
    cams = syn.make_camera_ring(2)
    prob = ref_import.RefProblem(model, cams, np.zeros((2, 17, 2), np.float32), np.ones((2, 17), np.float32),
                                 dtype='float32')
    arr = mf.model_arrays(prob.smpl)
    for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'kp_regressor'):
        assert arr[k].shape == model[k].shape, k
        assert np.array_equal(arr[k], model[k].astype(np.float32)), k
    assert np.array_equal(arr['parents'], model['parents'])
    assert np.array_equal(arr['face_vertex_ids'], model['face_vertex_ids'])
    assert np.array_equal(arr['joint_map'], model['joint_map'])


def test_signatures_accept_reference_call_sites():
    from mvsmplfitting_amd import fitting as mf
    ref = ref_import.load()
    ref_cls = type(ref.fitting.FittingMonitor(maxiters=1))     # the class is wrapped by @torch.no_grad() (fitting.py:36)
    for name in ('create_fitting_closure', 'run_fitting'):
        ours = inspect.signature(getattr(mf.FittingMonitor, name)).parameters
        theirs = inspect.signature(getattr(ref_cls, name)).parameters
        assert list(ours)[:len(theirs)] == list(theirs) or set(theirs) <= set(ours), name
    ours = inspect.signature(mf.create_optimizer).parameters
    for k in ('parameters', 'optim_type', 'lr', 'maxiters', 'gtol', 'ftol'):
        assert k in ours
