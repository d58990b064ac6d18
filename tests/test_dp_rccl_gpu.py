"""Two ranks over RCCL with the product ops on both GPUs (skipped on a one-GPU box: the round-end
8-GPU run exercises it): GraphedStep-style flat gradient exchange after a HIP-kernel forward/backward
of a small encoder, and the bench's own launch path under torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs2 = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs 2 GPUs')


@needs2
def test_bench_two_ranks_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', '29547', os.path.join(ROOT, 'bench.py'),
                          '--gpus', '2', '--steps', '3', '--warmup', '1', '--dtype', 'bf16', '--no-extras'],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['rccl_ranks'] == 2 and d['config']['global_batch'] == 4
    assert d['value'] > 0 and d['scaling'] == 'weak'


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import torch.distributed as dist
    from _util import encoder_case, t
    from unibev_amd import build_transformer, dp
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dp.init_distributed('nccl', dev)
    cfg, sd, inp, g = encoder_case('cnw')
    model = build_transformer(json.loads(json.dumps(cfg))).to(dev).eval()
    model.load_state_dict({k: t(v) for k, v in sd.items()}, strict=False)
    params = [p for n, p in model.named_parameters() if not n.startswith('reference_points')]
    scale = 1.0 + rank                                      # ranks see different data
    fused = model.encode([t(x, device=dev) * scale for x in inp['img']], [t(x, device=dev) * scale for x in inp['pts']],
                         t(inp['bev_q'], device=dev), inp['bev_h'], inp['bev_w'],
                         bev_pos=t(inp['bev_pos'], device=dev), img_metas=inp['metas'])
    fused.square().mean().backward()
    local = torch.cat([p.grad.flatten() for p in params]).clone()
    fg = dp.FlatGradients(params)
    fg.collect()
    fg.attach()
    fg.all_reduce_mean()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ref = sum(gathered) / world
    ret[rank] = bool(torch.allclose(fg.flat, ref, rtol=1e-5, atol=1e-7))
    dist.destroy_process_group()


@needs2
def test_flat_gradient_allreduce_two_gpus():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29549, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert ret[0] and ret[1]
