#!/usr/bin/env python3
"""Headline benchmark: samples/s of the UniBEV BEV-encoder hot path, forward + backward, on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one training pass of the hot path over one batch of synthetic nuScenes-shaped inputs
already resident in HBM (BASELINE.json north_star: 6 cameras x 256x704 -> 8x22 stride-32 feature
tokens each, a 180x180x256 LiDAR BEV feature map as the voxelize/backbone front end leaves it,
200x200x256 BEV queries, L+C CNW fusion with modality dropout, bs = 2 per GPU = configs[3]):
``UniBEV_Head.forward_bev`` (BEV queries + learned positional encoding -> both 3-layer encoders ->
CNW -> fusion), backward from a fixed random cotangent on ``fused_bev_embed`` to every encoder
parameter and the input features, RCCL gradient all-reduce (DDP, one process per GPU) and an AdamW
step.  Data parallel only: the per-GPU batch is fixed, so scaling is weak.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant deformable-sampling kernel: algorithmic bytes per launch (DESIGN.md,
                SURVEY.md section 8(d)) / its average launch duration, measured live with HIP
                events on the launch stream inside the timed region;
  cpu_baseline  the oracle (CPU port of the reference path) timed on this host (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s measured copy peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--bs', type=int, default=2, help='samples per GPU (cfg4: 2)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp16', 'fp32'])
    ap.add_argument('--workload', default='LC_cnw', choices=['LC_cnw', 'C', 'L', 'LC_cat128'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--eval-mode', action='store_true', help='dropout / modality dropout off')
    ap.add_argument('--fp32-stream', action='store_true',
                    help='keep the encoder residual stream in f32 under autocast (default: the '
                         'autocast dtype, as the reference\'s fp16 mode runs it)')
    return ap.parse_args()


WORKLOADS = {
    # name: (cfg kwargs, image (H, W), lidar feature hw, description)
    'LC_cnw': (dict(embed_dims=256, fusion_method='linear', feature_norm='ChannelNormWeights',
                    drop_modality=0.5), (256, 704), (180, 180),
               'unibev_nus_LC_cnw_256_modality_dropout: L+C CNW, 6x(8x22) img tokens [256x704/32], '
               '180x180 LiDAR BEV feats, 200x200x256 BEV'),
    'C': (dict(embed_dims=256, feature_norm=None, drop_modality=None, modalities='C'), (256, 704),
          None, 'unibev_nus_C: camera-only, 6x(8x22) img tokens, 200x200x256 BEV'),
    'L': (dict(embed_dims=256, feature_norm=None, drop_modality=None, modalities='L'), (256, 704),
          (180, 180), 'unibev_nus_L: LiDAR-only, 180x180 LiDAR BEV feats, 200x200x256 BEV'),
    'LC_cat128': (dict(embed_dims=128, fusion_method='cat', feature_norm=None, drop_modality=0.5),
                  (800, 1440), (180, 180),
                  'unibev_nus_LC_cat_128_modality_dropout: L+C cat, 6x(25x45) img tokens '
                  '[800x1440/32], 200x200x128 BEV'),
}


def build_head(workload, device):
    from unibev_amd import configs as cfgs
    from unibev_amd.registry import HEADS
    kw, img_hw, pts_hw, _ = WORKLOADS[workload]
    C = kw['embed_dims']
    tcfg = cfgs.transformer_cfg(decoder=None, **kw)
    tcfg.pop('decoder')
    head = HEADS.build(dict(type='UniBEV_Head', bev_h=200, bev_w=200, num_query=900, num_classes=10,
                            in_channels=C, transformer=tcfg,
                            bbox_coder=dict(pc_range=cfgs.PC_RANGE),
                            positional_encoding=dict(type='LearnedPositionalEncoding',
                                                     num_feats=C // 2, row_num_embed=200,
                                                     col_num_embed=200)))
    head.init_weights()
    # decoder-side parameters are not on the path
    for n, p in head.named_parameters():
        if n.startswith('query_embedding') or n.startswith('transformer.reference_points'):
            p.requires_grad_(False)
    return head.to(device), tcfg


def synth_inputs(workload, bs, dtype, device, rank):
    from unibev_amd import synthetic as syn
    kw, img_hw, pts_hw, _ = WORKLOADS[workload]
    C = kw['embed_dims']
    g = torch.Generator(device='cpu').manual_seed(1000 + rank)
    img = pts = None
    mods = kw.get('modalities', 'LC')
    if 'C' in mods:
        fh, fw = img_hw[0] // 32, img_hw[1] // 32
        img = [torch.randn(bs, 6, C, fh, fw, generator=g).to(device=device, dtype=dtype)
               .requires_grad_()]
    if 'L' in mods:
        pts = [torch.randn(bs, C, *pts_hw, generator=g).to(device=device, dtype=dtype)
               .requires_grad_()]
    l2i = torch.from_numpy(np.stack([syn.camera_rig(6, img_hw) for _ in range(bs)])).float().to(device)
    metas = [dict(lidar2img=l2i[b], img_shape=[(img_hw[0], img_hw[1], 3)] * 6) for b in range(bs)]
    return img, pts, metas


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         f'python -m torch.distributed.run --nproc-per-node {args.gpus} ...')
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    from unibev_amd import dp
    dp.init_distributed('nccl', device)                       # RCCL over xGMI (no-op for N = 1)

    from unibev_amd import functional as UF
    torch.manual_seed(0)          # identical replicas
    np.random.seed(rank)          # modality dropout is per process, as in the reference
    head, tcfg = build_head(args.workload, device)
    head.train(not args.eval_mode)
    head.transformer.lowp_stream = not args.fp32_stream
    dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[args.dtype]
    img, pts, metas = synth_inputs(args.workload, args.bs, dtype, device, rank)

    head.forward = head.forward_bev           # DDP calls module.forward
    model = dp.wrap_ddp(head, device_ids=[local])
    fwd = lambda: model(img, pts, metas)       # noqa: E731
    params = [p for p in head.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=0.01, fused=True)
    s = 2 if WORKLOADS[args.workload][0].get('fusion_method') == 'cat' else 1
    C = WORKLOADS[args.workload][0]['embed_dims']
    cot = torch.randn(200 * 200, args.bs, C * s, device=device) / 200.0

    def step():
        opt.zero_grad(set_to_none=True)
        for x in (img or []) + (pts or []):
            x.grad = None
        with torch.autocast('cuda', dtype=dtype, enabled=dtype != torch.float32):
            fused = fwd()
        (fused.float() * cot).sum().backward()
        torch.nn.utils.clip_grad_norm_(params, 35.0)
        opt.step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_kernel_timing:
        UF.kernel_profile(True)          # HIP events around every sampling kernel, on its stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_dt = time.perf_counter() - t0       # host-side enqueue time (before the final sync)
    barrier()
    dt = time.perf_counter() - t0
    prof = {} if args.no_kernel_timing else UF.kernel_profile()
    UF.kernel_profile(False)
    dt = dp.max_over_ranks(dt, device)

    if rank == 0:
        out = {
            'metric': 'nuScenes samples/sec BEV-encoder fwd+bwd', 'value': world * args.bs * args.steps / dt,
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'host_enqueue_ms_per_step': 1e3 * host_dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': WORKLOADS[args.workload][3], 'per_gpu_batch': args.bs,
                       'global_batch': world * args.bs, 'encoder_layers': 3,
                       'mode': 'eval' if args.eval_mode else 'train (dropout 0.1, modality dropout)',
                       'residual_stream': 'f32' if (args.fp32_stream or args.dtype == 'fp32') else args.dtype,
                       'step': 'fwd + bwd + grad all-reduce + clip + AdamW',
                       'parallelism': f'dp{world}'},
        }
        # ---- roofline of the dominant sampling kernel (HIP events inside the library) -------
        detail = []
        for name, r in prof.items():
            detail.append({'kernel': name, 'launches': r['launches'], 'avg_us': r['avg_us'],
                           'algorithmic_bytes_per_launch': r['bytes_per_launch'],
                           'achieved_GBps': r['bytes_per_launch'] / (r['avg_us'] * 1e-6) / 1e9})
        detail.sort(key=lambda d: -d['avg_us'] * d['launches'])
        if detail:
            dom = detail[0]
            traffic = None
            tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tfile):
                rec = json.load(open(tfile)).get(dom['kernel'].split(',')[0])   # name<P=..
                if rec and 'write' in rec and 'fetch_corrected' in rec:          # HBM-side bytes per launch from the PMC passes (see the file's note)
                    traffic = rec['fetch_corrected'] + rec['write']
            out['roofline'] = {'bound': 'hbm', 'achieved': dom['achieved_GBps'], 'peak': HBM_PEAK_GBS,
                               'unit': 'GB/s', 'frac': dom['achieved_GBps'] / HBM_PEAK_GBS,
                               'traffic': traffic, 'kernel': dom['kernel'],
                               'avg_launch_us': dom['avg_us'],
                               'algorithmic_bytes_per_launch': dom['algorithmic_bytes_per_launch']}
            out['roofline_detail'] = detail
        else:
            out['roofline'] = None
        # ---- CPU baseline: the oracle's forward on this host ------------------------------
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args, tcfg, head)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def cpu_baseline(args, tcfg, head):
    """The oracle (CPU restatement of the reference path, oracle/unibev_ref.py) timed on the host:
    forward only, fp32, bs = 1, eval mode, 1 warm-up + 2 timed passes (~10-20 s)."""
    from oracle import unibev_ref as R
    # a 256-thread host oversubscribes torch's small CPU kernels (78.9 s/pass measured with all
    # threads vs ~5 s with 16): the baseline uses at most 16 threads and says so
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sd = {k[len('transformer.'):]: v.detach().float().cpu() for k, v in head.state_dict().items()
          if k.startswith('transformer.')}
    img, pts, metas = synth_inputs(args.workload, 1, torch.float32, 'cpu', 0)
    metas = [dict(lidar2img=[m.numpy() for m in metas[0]['lidar2img']], img_shape=metas[0]['img_shape'])]
    bev_q = head.state_dict()['bev_embedding.weight'].float().cpu()
    C = bev_q.shape[1]
    pos = R.learned_positional_encoding(head.state_dict()['positional_encoding.row_embed.weight'].float().cpu(),
                                        head.state_dict()['positional_encoding.col_embed.weight'].float().cpu(),
                                        1, 200, 200)
    cfg = json.loads(json.dumps(tcfg))

    def run():
        with torch.no_grad():
            return R.transformer_encode_fuse(sd, cfg, None if img is None else [x.detach() for x in img],
                                             None if pts is None else [x.detach() for x in pts],
                                             bev_q, 200, 200, pos, metas)
    run()
    n = 2
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    dt = (time.perf_counter() - t0) / n
    return {'value': 1.0 / dt, 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'forward only, fp32, bs=1, eval, {n} timed passes of the same workload '
                      f'({dt:.2f} s each) through oracle/unibev_ref.py (torch CPU)'}


if __name__ == '__main__':
    main()
