"""bench.py end to end on one GPU: the JSON contract, and the multi-GPU code path (process group,
DDP wrapper with gradients as bucket views, RCCL all-reduce hooks) forced on for a single rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *flags):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline', *flags], env=env, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize('force_ddp', ['0', '1'])
def test_bench_json_contract(force_ddp):
    d = _run({'UBV_FORCE_DDP': force_ddp})
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['value'] > 0 and d['unit'] == 'samples/s' and 'workload' in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and 0 < r['frac'] < 1
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
