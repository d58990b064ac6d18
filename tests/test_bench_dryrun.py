"""bench.py's own N-rank path on CPU (gloo, world_size 2): the launcher (`python bench.py --gpus 2` re-executes itself
under torch.distributed.run, as the driver's multi-GPU run does through torchrun), the rank environment, the process
group, the flat-gradient exchange in both modes against its closed form, the max-over-ranks timing and rank 0's single
JSON line — everything of the data-parallel protocol that is not RCCL- or HIP-graph-specific (`--dry-run`).  The
reference launches its job the same way (tools/train_UniBEV.py:242-249: one process per GPU, NCCL process group)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT',
                                                            'TORCHELASTIC_RUN_ID', 'UBV_BENCH_CHILD', 'UBV_FORCE_DDP')}
    env.update(OMP_NUM_THREADS='1', **kw)
    return env


def _json_lines(stdout):
    return [json.loads(ln) for ln in stdout.splitlines() if ln.startswith('{')]


@pytest.mark.parametrize('exchange', ['auto', 'single', 'split'])
def test_self_launched_two_ranks_print_one_line(exchange):
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run', '--steps', '3', '--warmup', '1',
                        '--exchange', exchange], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # rank 0 only
    d = lines[0]
    assert r.stdout.strip().splitlines()[-1].startswith('{')          # ... and it is the last line of stdout
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['dry_run'] is True
    assert d['exchange_ok'] is True
    cfg = d['config']
    assert cfg['rccl_ranks'] == 2 and cfg['parallelism'] == 'dp2' and cfg['global_batch'] == 4
    assert cfg['gradient_exchange'] == ('split' if exchange == 'split' else 'single')      # auto = single
    assert cfg['launcher'].startswith('self')
    assert cfg['collective']['backend'] in ('gloo', 'nccl') and cfg['collective']['ranks'] == 2


def test_under_torchrun_as_the_driver_launches_it():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), BENCH, '--gpus', '2', '--steps', '2', '--warmup', '1', '--dry-run']
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['config']['rccl_ranks'] == 2 and lines[0]['config']['launcher'] == 'torchrun'


def test_world_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '1', '--dry-run'], env=_env(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0'),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in (r.stderr + r.stdout)


def test_allreduce_algorithm_is_a_recorded_choice(monkeypatch):
    """`auto`: the ring only for the overlapped exchange; the pin is an argument of dp.init_distributed, not a process-wide
    default."""
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    for exchange, algo, want in (('auto', 'auto', 'default'), ('single', 'auto', 'default'), ('split', 'auto', 'ring'),
                                 ('single', 'ring', 'ring'), ('split', 'default', 'default')):
        assert bench.allreduce_algo(argparse.Namespace(exchange=exchange, allreduce_algo=algo)) == want
    import inspect
    from unibev_amd import dp
    src = inspect.getsource(dp.init_distributed)
    assert "setdefault('NCCL_ALGO'" not in src and "algo == 'ring'" in src
