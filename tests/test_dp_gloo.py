"""The data-parallel harness on CPU: 2 processes, gloo.  Checks the sample sharding contract, that
DDP-averaged gradients (and the flat-buffer single all-reduce the bench uses) of encoder-side torch modules (FFN + LayerNorm built through the registry)
equal the average of the per-rank gradients, and the max-over-ranks timing reduction."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from unibev_amd import dp
    from unibev_amd.registry import build_feedforward_network
    torch.set_num_threads(1)
    r, w = dp.init_distributed('gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(
        build_feedforward_network(dict(type='FFN', embed_dims=32, feedforward_channels=64,
                                       ffn_drop=0.0)),
        torch.nn.LayerNorm(32))
    ddp = dp.wrap_ddp(net)
    data = torch.randn(7, 5, 32, generator=torch.Generator().manual_seed(1))      # 7 samples
    mine = dp.shard_samples(7, rank, world)
    loss = ddp(data[mine]).square().sum() / len(mine)
    loss.backward()
    grads = torch.cat([p.grad.flatten() for p in net.parameters()])
    # reference: average over ranks of the per-rank gradients, computed without DDP
    ref = torch.zeros_like(grads)
    for rr in range(world):
        torch.manual_seed(0)
        net2 = torch.nn.Sequential(
            build_feedforward_network(dict(type='FFN', embed_dims=32, feedforward_channels=64,
                                           ffn_drop=0.0)),
            torch.nn.LayerNorm(32))
        idx = dp.shard_samples(7, rr, world)
        (net2(data[idx]).square().sum() / len(idx)).backward()
        ref += torch.cat([p.grad.flatten() for p in net2.parameters()]) / world
    ok = torch.allclose(grads, ref, rtol=1e-5, atol=1e-6)
    # the flat-gradient exchange bench.py / graph_step.GraphedStep use: ONE all-reduce of one buffer
    torch.manual_seed(0)
    net3 = torch.nn.Sequential(
        build_feedforward_network(dict(type='FFN', embed_dims=32, feedforward_channels=64,
                                       ffn_drop=0.0)),
        torch.nn.LayerNorm(32))
    (net3(data[mine]).square().sum() / len(mine)).backward()
    fg = dp.FlatGradients(list(net3.parameters()))
    fg.collect()
    fg.attach()
    fg.all_reduce_mean()
    flat = torch.cat([p.grad.flatten() for p in net3.parameters()])
    ok = ok and torch.allclose(flat, ref, rtol=1e-5, atol=1e-6) and torch.equal(flat, fg.flat) \
        and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(net3.parameters(), fg.views))
    # the segmented exchange of a split backward (graph_step.GraphedStep(split_after=...)): the first parameters form
    # segment 0, collected and sent on its way first, the rest follows; same averages as the single message
    torch.manual_seed(0)
    net4 = torch.nn.Sequential(
        build_feedforward_network(dict(type='FFN', embed_dims=32, feedforward_channels=64,
                                       ffn_drop=0.0)),
        torch.nn.LayerNorm(32))
    (net4(data[mine]).square().sum() / len(mine)).backward()
    ps = list(net4.parameters())
    fs = dp.FlatGradients(ps, first_segment=2)
    assert len(fs.segments) == 2 and fs.segments[0].numel() == ps[0].numel() + ps[1].numel()
    fs.collect(0, 2, [ps[0].grad, ps[1].grad])
    fs.start_segment(0)                                   # in flight while "the rest of the backward" is collected
    fs.collect(2, len(ps))
    fs.start_segment(1)
    fs.finish_segments()
    ok = ok and torch.allclose(fs.flat, ref, rtol=1e-5, atol=1e-6) and not fs.missing and not fs._pending
    tmax = dp.max_over_ranks(1.0 + rank)
    dp.barrier()
    ret[rank] = (ok, mine, tmax)
    dist.destroy_process_group()


def test_sample_sharding_contract():
    from unibev_amd.dp import shard_samples
    for n in (0, 1, 7, 8, 16, 17):
        for w in (1, 2, 3, 8):
            parts = [shard_samples(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            sizes = [len(p) for p in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gradient_allreduce_gloo():
    ctx = mp.get_context('spawn')
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == [0, 1, 2, 3] and ret[1][1] == [4, 5, 6]
    assert ret[0][2] == ret[1][2] == 2.0
