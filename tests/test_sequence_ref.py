"""CPU: the sequence-mode carry-over rules of mvsmplfitting_amd.sequence against the reference's own `is_seq` chain
(tests/golden/sequence_ref.npz, recorded by oracle/make_golden_sequence.py from the reference's load_init + fix_params +
non_linear_solver, code/main.py:76-88, code/utils/init_guess.py:137-215, code/utils/non_linear_solver.py:156-180):
given the reference's result of frame t - 1, frame t must start from EXACTLY the reference's start vector, with
exactly its cold / warm decision (the 5000-loss rule) and exactly its per-stage weights."""
import os

import numpy as np
import pytest

from mvsmplfitting_amd import sequence as sq
from mvsmplfitting_amd.engine import stage_weights
from tests.gpu_helpers import to118
from tests.helpers import GOLD

CHAINS = {'l2_a': False, 'l2_b': False, 'vp_a': True}


def load_chain(name, dtype='float32'):
    g = np.load(os.path.join(GOLD, 'sequence_ref.npz'))
    c = {k.split('/', 1)[1]: g[k] for k in g.files if k.startswith(name + '/') and k.count('/') == 1}
    c.update({k.split('/', 2)[2]: g[k] for k in g.files if k.startswith('%s/%s/' % (name, dtype))})
    c['model_checksum'] = float(g['model_checksum'])
    return c


def x_init_118(c, use_vp):
    """The frame's own full initial guess in the C ABI's layout, body pose = fix_params' start (init_guess.py:199-203)."""
    xi = np.stack([to118(x, use_vp) for x in c['x_init']]).astype(np.float32)
    xi[:, 13:19] = 1.0
    return xi


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
@pytest.mark.parametrize('name', sorted(CHAINS))
def test_carry_over_equals_the_references_load_init_and_fix_params(name, dtype):
    use_vp = CHAINS[name]
    c = load_chain(name, dtype)
    xi = x_init_118(c, use_vp)
    T = xi.shape[0]
    full = stage_weights(1536.0, flags=1 if use_vp else 0)
    warm = sq.sequence_stages(full)
    assert c['seq_start'][0] and c['nstages'][0] == 4
    restarts = 0
    for t in range(1, T):
        prev = to118(c['xf'][t - 1], use_vp)[None]
        x0, cold = sq.carry_over(prev, np.asarray([c['loss'][t - 1]]), xi[t:t + 1].astype(np.float64), use_vp)
        assert bool(cold[0]) == bool(c['seq_start'][t]), (t, c['loss'][t - 1])
        ref0 = to118(c['x0'][t], use_vp)
        if use_vp:
            x0[0, 13:82] = 0.0; ref0[13:82] = 0.0           # no body_pose parameter with VPoser (decoded from the embedding)
        assert np.array_equal(x0[0], ref0), (t, np.abs(x0[0] - ref0).max())
        stg = full if cold[0] else warm
        assert len(stg) == c['nstages'][t]
        restarts += bool(cold[0])
        if dtype == 'float32':                               # the weights the reference's float32 loss module was given
            mine = np.asarray([[s['data_weight'], s['body_pose_weight'], s['shape_weight'], s['bending_prior_weight']] for s in stg],
                              np.float32)
            assert np.array_equal(mine, c['stages'][t][:len(stg)].astype(np.float32)), (mine, c['stages'][t])
    if name == 'l2_b':
        assert restarts == 1 and c['loss'][1] > sq.RESTART_LOSS
    else:
        assert restarts == 0
    if dtype == 'float32':                                   # frame 0: the four yaml stages
        mine = np.asarray([[s['data_weight'], s['body_pose_weight'], s['shape_weight'], s['bending_prior_weight']] for s in full], np.float32)
        assert np.array_equal(mine, c['stages'][0].astype(np.float32))
    x0_first = to118(c['x0'][0], use_vp)
    if use_vp:
        x0_first[13:82] = 0.0
        assert np.array_equal(x0_first[:13], xi[0, :13]) and np.array_equal(x0_first[82:], xi[0, 82:])
    else:
        assert np.array_equal(x0_first, xi[0].astype(np.float64))
