#!/usr/bin/env python
"""bench.py -- images/sec of the PackNet01(+PoseNet) self-supervised TRAINING step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = zero_grad -> PackNet01 + PoseNet forward -> multi-view photometric loss (4 scales, SSIM+L1, automask,
smoothness) -> backward -> (N>1: RCCL gradient all-reduce, overlapped with backward) -> Adam, on a synthetic
KITTI-shaped batch of 192x640 triplets, batch 4 per GPU (BASELINE.json configs[1]); fp32 end to end.
Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how `roofline` and `cpu_baseline` are defined.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn.functional as F  # noqa: E402

# algorithmic FLOPs of one trained image at 192x640 (fwd + dgrad + wgrad of every conv; SURVEY.md 8d / BASELINE.md 2)
GFLOP_PER_IMAGE_192x640 = 1232.0
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense

LOSS_DEFAULTS = dict(  # configs/default_config.py:88-103 of the reference
    num_scales=4, progressive_scaling=0.0, flip_lr_prob=0.5, rotation_mode='euler', upsample_depth_maps=True,
    ssim_loss_weight=0.85, occ_reg_weight=0.1, smooth_loss_weight=0.001, C1=1e-4, C2=9e-4,
    photometric_reduce_op='min', disp_norm=True, clip_loss=0.0, padding_mode='zeros', automask_loss=True)


def synthetic_batch(B, H, W, seed, device):
    """KITTI-shaped triplets: smooth random scene, context frames = small horizontal camera shifts."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(B, 3, H // 16 + 2, W // 16 + 4, generator=g)
    big = F.interpolate(base, size=(H + 16, W + 32), mode='bicubic', align_corners=True).clamp(0, 1)
    big = (big + 0.03 * torch.rand(big.shape, generator=g)).clamp(0, 1)
    frames = [big[:, :, 8:8 + H, 16 + dx:16 + dx + W].contiguous() for dx in (-3, 0, 3)]
    K = torch.tensor([[0.58 * W, 0., 0.5 * W], [0., 1.92 * H, 0.5 * H], [0., 0., 1.]], dtype=torch.float64).repeat(B, 1, 1)
    rgb, ctx = frames[1].to(device), [frames[0].to(device), frames[2].to(device)]
    return {'rgb': rgb, 'rgb_context': ctx, 'rgb_original': rgb, 'rgb_context_original': ctx, 'intrinsics': K.to(device)}


def build_model(device, depth_net='PackNet01'):
    from packnet_sfm.models.SelfSupModel import SelfSupModel
    from packnet_sfm.networks.pose.PoseNet import PoseNet
    import importlib
    # resolved by name like the reference's load_class (utils/load.py:79-111); the headline metric is PackNet01
    PackNet01 = getattr(importlib.import_module('packnet_sfm.networks.depth.' + depth_net), depth_net)
    torch.manual_seed(42)          # same seed on every rank: replicas start identical (reference: model_wrapper.py:44)
    random.seed(42)
    model = SelfSupModel(**LOSS_DEFAULTS)
    model.add_depth_net(PackNet01(dropout=0.0, version='1A'))
    model.add_pose_net(PoseNet(nb_ref_imgs=2, rotation_mode='euler'))
    return model.to(device).train()


def cpu_baseline(H, W, seconds_budget=30.0):
    """The oracle (oracle/packnet_oracle.py: the reference's algorithm restated on stock torch CPU ops) timed on
    this box's host cores: full training steps (fwd + loss + bwd + Adam) at batch 1, 1 warm-up + up to 2 timed."""
    from oracle import packnet_oracle as O
    threads = torch.get_num_threads()
    sd = {k: v.requires_grad_(True) for k, v in O.init_params(O.packnet01_param_shapes('1A'), seed=42).items()}
    psd = {k: v.requires_grad_(True) for k, v in O.init_params(O.posenet_param_shapes(2), seed=43).items()}
    opt = torch.optim.Adam([{'params': list(sd.values()), 'lr': 2e-4}, {'params': list(psd.values()), 'lr': 2e-4}])
    batch = synthetic_batch(1, H, W, 1234, 'cpu')
    kw = {k: LOSS_DEFAULTS[k] for k in ('num_scales', 'ssim_loss_weight', 'smooth_loss_weight', 'C1', 'C2',
                                         'photometric_reduce_op', 'automask_loss')}
    times = []
    t_start = time.time()
    for i in range(3):
        t0 = time.time()
        opt.zero_grad()
        out = O.selfsup_forward(sd, psd, batch, flip=False, **kw)
        out['loss'].sum().backward()
        opt.step()
        dt = time.time() - t0
        if i > 0:
            times.append(dt)
        if time.time() - t_start > seconds_budget and times:
            break
    best = min(times)
    return {'value': round(1.0 / best, 4), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': 'oracle (torch CPU fp32 restatement of the reference path), full train step, batch 1, %dx%d, '
                      '1 warm-up + %d timed steps, best' % (H, W, len(times))}


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json;
    PMC counters cannot be collected from inside this process).  None if no profile has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))['pnsfm::conv2d_mfma_kernel']
        return {'hbm_bytes_per_launch': round(d['hbm_bytes_per_launch']), 'algorithmic_bytes_per_launch':
                round(d.get('algorithmic_bytes_per_launch', 0)), 'source': os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--height', type=int, default=192)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--batch', type=int, default=4, help='images per GPU')
    ap.add_argument('--depth-net', default='PackNet01', choices=['PackNet01', 'PackNetSlim01'],
                    help='PackNet01 = the BASELINE.json metric; PackNetSlim01 = the d=4 / 32-channel-stem variant (not the metric)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prof', action='store_true', help='do not bracket the conv kernels with events')
    ap.add_argument('--optimizer', default='torch', choices=['flat', 'torch'],
                    help="'flat': FlatAdam (one gfx950 adam_kernel launch per group); 'torch': torch.optim.Adam(fused=True)")
    ap.add_argument('--layer-table', default='', help='write the per-launch conv table (CSV) of the timed region here')
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP kernels have no CPU fallback')
    from packnet_sfm.hip import ops
    from packnet_sfm.rccl import hvd

    hvd.init()
    rank, world = hvd.rank(), hvd.size()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N>1)' % (args.gpus, world))
    device = torch.device('cuda', hvd.local_rank() % torch.cuda.device_count())
    torch.cuda.set_device(device)

    H, W, B = args.height, args.width, args.batch
    model = build_model(device, args.depth_net)
    batch = synthetic_batch(B, H, W, 1234 + rank, device)
    groups = [{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0},
              {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0}]
    if args.optimizer == 'flat':
        from packnet_sfm.rccl.flat_adam import FlatAdam
        optimizer = FlatAdam(groups)
    else:
        optimizer = torch.optim.Adam(groups, fused=True)
    force_ddp = os.environ.get('PNSFM_FORCE_DDP') == '1'   # single-GPU rehearsal of the N>1 path (1-rank RCCL group)
    if world > 1 or force_ddp:
        optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters(),
                                             compression=hvd.Compression.none, force_collectives=force_ddp)

    def step():
        optimizer.zero_grad()
        out = model(batch, progress=0.0)
        out['loss'].backward()
        optimizer.step()
        return out['loss']

    def fence():
        if world > 1 or force_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # the library autotunes every layer shape on first use and the caching allocator settles over the first steps: at
    # least two untimed steps always run, whatever --warmup says
    for _ in range(max(0, 2 - args.warmup)):
        step()
    for _ in range(args.warmup):
        loss = step()
    fence()
    if not args.no_prof:
        ops.prof_reset()
        ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if not args.no_prof:
        ops.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.detach().float().item())

    if rank == 0 and args.layer_table and not args.no_prof:
        ops.prof_dump(args.layer_table)
    timed = None
    if not args.no_prof:
        timed = (ops.prof_collect(0), ops.prof_collect(1))
        # The timed region overlaps every weight-gradient kernel with the data-gradient chain on a second HIP stream
        # (DESIGN.md 3e), so its per-launch durations are those of kernels SHARING the GPU.  Two extra, untimed steps
        # with the side stream off give the same kernels' durations in isolation (kernel quality, not step throughput).
        from packnet_sfm.hip import functional as HF
        was = HF._WgradStream.enabled
        HF.set_wgrad_stream(False)
        step()
        fence()
        ops.prof_reset()
        ops.prof_enable(True)
        for _ in range(2):
            step()
        fence()
        ops.prof_enable(False)
        HF.set_wgrad_stream(was)
    if rank == 0:
        images = B * world * args.steps
        value = images / elapsed
        scale = (H * W) / (192.0 * 640.0)
        roofline = None
        traffic = measured_traffic()
        if not args.no_prof:
            (ms0, fl0, n0), (ms1, fl1, n1) = timed     # conv2d_mfma_kernel (forward + backward-data), conv2d_wgrad_kernel
            ims0, ifl0, in0 = ops.prof_collect(0)
            ims1, ifl1, in1 = ops.prof_collect(1)
            if n0 > 0 and ms0 > 0:
                ach = fl0 / (ms0 * 1e-3) / 1e12
                roofline = {
                    'bound': 'mfma', 'kernel': 'conv2d_mfma_kernel (fwd + dgrad implicit GEMM)',
                    'achieved': round(ach, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                    # HBM bytes per launch (FETCH_SIZE + WRITE_SIZE PMC passes, profiles/rNN_traffic.json) or null
                    'traffic': (traffic or {}).get('hbm_bytes_per_launch'), 'traffic_detail': traffic,
                    'launches': int(n0), 'avg_launch_ms': round(ms0 / n0, 4),
                    'flop_per_launch_avg': round(fl0 / n0, 1),
                    'wgrad_kernel': {'achieved': round(fl1 / (ms1 * 1e-3) / 1e12, 2) if ms1 > 0 else None,
                                     'launches': int(n1), 'avg_launch_ms': round(ms1 / max(n1, 1), 4)},
                    'conv_kernel_time_over_step_time': round((ms0 + ms1) * 1e-3 / elapsed, 4),   # > 1: the two streams overlap
                    'isolated': {       # same kernels, 2 extra steps with the weight-gradient side stream off
                        'achieved': round(ifl0 / (ims0 * 1e-3) / 1e12, 2) if ims0 > 0 else None,
                        'frac': round(ifl0 / (ims0 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if ims0 > 0 else None,
                        'avg_launch_ms': round(ims0 / max(in0, 1), 4),
                        'wgrad_achieved': round(ifl1 / (ims1 * 1e-3) / 1e12, 2) if ims1 > 0 else None},
                    'whole_step_vs_mfma_peak': round(value * GFLOP_PER_IMAGE_192x640 * scale / 1e3 / world / FP32_MFMA_PEAK_TFLOPS, 4),
                }
        result = {
            'metric': 'images/sec %s self-sup train %dx%d' % (args.depth_net, H, W), 'value': round(value, 3), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': args.depth_net + '(1A)+PoseNet self-supervised train step (fwd+photometric loss+bwd+allreduce+Adam), '
                                   'KITTI-shaped %dx%d triplets, batch %d/GPU (%s)' % (
                                       H, W, B, 'BASELINE.json configs[1]' if (H, W, B) == (192, 640, 4) else
                                       ('BASELINE.json configs[2] shape' if (H, W, B) == (384, 1280, 2) else 'custom shape')),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'final_loss': round(loss_val, 6)},
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(H, W)
        print(json.dumps(result), flush=True)
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
