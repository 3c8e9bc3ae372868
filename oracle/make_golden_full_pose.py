"""Golden vectors for mvfit_full_pose = ModelOutput.full_pose of the reference's SMPL.forward
(code/models/body_models_scale.py:392-412), the body pose decoded by the reference's VPoser.decode(z, 'aa')
(code/models/VPoser.py:218-232) from the shipped checkpoint's decoder (tests/golden/vposer_poser_epoch091_decoder.npz,
exported by oracle/make_golden_demo.py) and from the synthetic "wild" decoder that reaches all four branches of
rotation_matrix_to_quaternion.  TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    python -m oracle.make_golden_full_pose      ->  tests/golden/full_pose.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import as ri                      # noqa: E402
from mvsmplfitting_amd import synthetic as syn           # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def decode_cases(model, vpw, xs):
    """Reference float64 full_pose for flat parameter rows xs [n, 49] (betas, global_orient, transl, scale, embedding)."""
    import torch
    cams = syn.make_camera_ring(2)
    rp = ri.RefProblem(model, cams, np.zeros((2, 17, 2)), np.ones((2, 17)), dtype='float64', use_vposer=True,
                       vposer_weights=vpw)
    out = []
    for x in xs:
        rp.set_flat(x)
        with torch.no_grad():
            bp = rp.vposer.decode(rp.pose_embedding, output_type='aa').view(1, -1)            # fitting.py:170-173
            o = rp.smpl(return_verts=False, body_pose=bp, return_full_pose=True)
        out.append(o.full_pose.detach().numpy().reshape(72))
    return np.stack(out)


def main():
    assert ri.available()
    model = syn.make_body_model(0, skin_topk=4)
    rng = np.random.default_rng(77)
    res = {}
    real = dict(np.load(os.path.join(GOLD, 'vposer_poser_epoch091_decoder.npz')))
    real = {k: real[k] for k in ('fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'out_w', 'out_b')}
    for name, vpw, zs in (('real', real, 1.0), ('wild', syn.make_vposer_decoder(seed=2, gain=1.0, identity_bias=False), 1.5)):
        xs = np.zeros((6, 49))
        xs[:, 10:13] = rng.normal(0, 0.8, (6, 3))
        xs[:, 16] = 1.0
        xs[:, 17:49] = rng.normal(0, zs, (6, 32))
        xs[0, 17:49] = 0.0                                   # the fit's starting embedding
        res[name + '_x'] = xs
        res[name + '_full_pose'] = decode_cases(model, vpw, xs)
    np.savez_compressed(os.path.join(GOLD, 'full_pose.npz'), **res)
    print({k: v.shape for k, v in res.items()})


if __name__ == '__main__':
    main()
