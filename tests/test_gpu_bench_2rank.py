"""bench.py as the driver launches it for N > 1 - here two ranks on the one GPU of the test box over gloo
(--single-device), so that the frame sharding, the barrier / max-over-ranks timing, the final gather and the JSON
contract of the multi-GPU line are exercised on the real kernels and not only by the CPU stub tests."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _line(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    e['MASTER_ADDR'] = '127.0.0.1'
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_device_keep_the_contract():
    common = ['--steps', '2', '--warmup', '1', '--no-pmc', '--no-cpu-baseline', '--no-variants']
    one = _line([sys.executable, 'bench.py', '--gpus', '1'] + common)
    two = _line([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                 '--master-port', '29731', 'bench.py', '--gpus', '2', '--dist-backend', 'gloo', '--single-device'] + common)
    for out, n in ((one, 1), (two, 2)):
        assert out['n_gpus'] == n and out['steps'] == 2 and out['warmup'] == 1
        assert out['scaling'] == 'weak' and 'linear by construction' in out['scaling_note']
        assert out['config']['problems_total'] == 32 * n and out['config']['problems_per_gpu'] == 32
        assert out['vertex_passes_lost_in_timed_fits'] == {'missed': 0, 'timed_out': 0}
        assert 'invalid_reason' not in out
        assert out['value'] > 0 and out['higher_is_better'] is True and out['unit'] == one['unit']
    # the same seeded frames 0..31 are rank 0's shard in both runs: same closures per frame for that shard.  Two ranks sharing
    # one GPU (a dry run: per-round pass launches, bench.py --single-device) take about twice one rank alone - which runs the
    # resident pass and has the device to itself; no performance claim hangs on this, the bound only catches a stall
    assert len(two['per_rank_busy_ms_per_step']) == 2
    assert two['ms_per_step'] < 3.5 * one['ms_per_step']


def test_default_line_is_reproducible_within_5_percent():
    common = ['--steps', '5', '--warmup', '1', '--no-pmc', '--no-cpu-baseline', '--no-variants']
    a = _line([sys.executable, 'bench.py'] + common)
    b = _line([sys.executable, 'bench.py', '--gpus', '1'] + common)
    assert abs(a['value'] - b['value']) <= 0.05 * max(a['value'], b['value']), (a['value'], b['value'])
    assert a['closure_rounds_per_fit'] == b['closure_rounds_per_fit']
