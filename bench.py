#!/usr/bin/env python
"""bench.py -- images/sec of the PackNet01(+PoseNet) self-supervised TRAINING step on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W          # spawns its own N ranks (torch.distributed.run) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --height 384 --width 1280 --batch 2    # BASELINE.json configs[2] shape

One "step" = zero_grad -> PackNet01 + PoseNet forward -> multi-view photometric loss (4 scales, SSIM+L1, automask,
smoothness) -> backward -> (N>1: RCCL gradient all-reduce, overlapped with backward) -> Adam, on a synthetic
KITTI-shaped batch of 192x640 triplets, batch 4 per GPU (BASELINE.json configs[1]); fp32 tensors end to end, the conv
GEMMs computed as fp32-from-exact-bf16-splits on the bf16 matrix pipe (PNSFM_CONV_MATH=f32 for the f32 MFMA).
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
BF16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: v_mfma_f32_32x32x16_bf16, dense (no sparsity)

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


def _cpu_baseline_one(H, W, threads, batch_size, steps):
    """The oracle (oracle/packnet_oracle.py: the reference's algorithm restated on stock torch CPU ops) timed on
    this box's host cores with `threads` intra-op threads: full training steps (fwd + loss + bwd + Adam), 1 warm-up +
    `steps` timed; returns the best step time in seconds."""
    from oracle import packnet_oracle as O
    torch.set_num_threads(threads)
    sd = {k: v.requires_grad_(True) for k, v in O.init_params(O.packnet01_param_shapes('1A'), seed=42).items()}
    psd = {k: v.requires_grad_(True) for k, v in O.init_params(O.posenet_param_shapes(2), seed=43).items()}
    opt = torch.optim.Adam([{'params': list(sd.values()), 'lr': 2e-4}, {'params': list(psd.values()), 'lr': 2e-4}])
    batch = synthetic_batch(batch_size, H, W, 1234, 'cpu')
    kw = {k: LOSS_DEFAULTS[k] for k in ('num_scales', 'ssim_loss_weight', 'smooth_loss_weight', 'C1', 'C2',
                                         'photometric_reduce_op', 'automask_loss')}
    times = []
    for i in range(1 + steps):
        t0 = time.time()
        opt.zero_grad()
        out = O.selfsup_forward(sd, psd, batch, flip=False, **kw)
        out['loss'].sum().backward()
        opt.step()
        if i > 0:
            times.append(time.time() - t0)
    return min(times)


def cpu_baseline(H, W, seconds_budget=45.0):
    """CPU baseline = the oracle's training step on the host cores of this box (rank 0, N=1 only).  The intra-op thread
    count is SWEPT (8/16/32/64, capped at the cores present: oversubscribing all 128+ hardware threads measured 3x slower
    than 8 threads in round 1) at batch 1, the best count is re-timed at batch 4 (SURVEY.md 8d asks for both), and the
    best images/sec is reported with `cores` = the threads that produced it.  Bounded sample: every leg is 1 warm-up + 1
    timed step and the sweep stops when the time budget is spent."""
    ncpu = os.cpu_count() or 8
    saved = torch.get_num_threads()
    t_start = time.time()
    sweep = {}
    for th in [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]:
        sweep[th] = 1.0 / _cpu_baseline_one(H, W, th, 1, 1)
        if time.time() - t_start > seconds_budget:
            break
    best_th = max(sweep, key=sweep.get)
    best, best_b = sweep[best_th], 1
    b4 = None
    if time.time() - t_start < seconds_budget:
        b4 = 4.0 / _cpu_baseline_one(H, W, best_th, 4, 1)
        if b4 > best:
            best, best_b = b4, 4
    torch.set_num_threads(saved)
    return {'value': round(best, 4), 'unit': 'images/sec', 'cores': best_th, 'kind': 'port',
            'thread_sweep_batch1': {str(k): round(v, 4) for k, v in sweep.items()},
            'batch4_at_best_threads': round(b4, 4) if b4 else None, 'host_cpus': ncpu,
            'sample': 'oracle (torch CPU fp32 restatement of the reference path), full train step (fwd+loss+bwd+Adam), %dx%d, '
                      '1 warm-up + 1 timed step per leg; best of the thread sweep at batch 1 and of batch 4 at that thread '
                      'count (batch %d won)' % (H, W, best_b)}


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json;
    PMC counters cannot be collected from inside this process).  None if no profile has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        d = d.get('pnsfm::conv2d_bx3_kernel') or d['pnsfm::conv2d_mfma_kernel']
        return {'hbm_bytes_per_launch': round(d['hbm_bytes_per_launch']), 'algorithmic_bytes_per_launch':
                round(d.get('algorithmic_bytes_per_launch', 0)), 'source': os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def _self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-run this script as N ranks under torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) -- the analogue of the reference's `mpirun -np NGPUS` (Makefile:42-53)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: the only mode the host driver supports
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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
    ap.add_argument('--no-prof', action='store_true', help='skip the per-launch event timing of the conv kernels (roofline = null)')
    ap.add_argument('--optimizer', default='flat', choices=['flat', 'torch'],
                    help="'flat': FlatAdam (one gfx950 adam_kernel launch per group); 'torch': torch.optim.Adam(fused=True)")
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='replay the whole step as a hipGraph (packnet_sfm/hip/graph.py).  auto = off: measured on MI355X the replay is '
                         '3-5 %% SLOWER than eager launches (84-85 vs 88-90 img/s; the step is GPU-bound, eager already hides '
                         'launch latency behind kernels, and the graph serialises the weight-gradient side stream) -- kept as '
                         'an option and parity-tested; never available for N>1 (RCCL all-reduces start from autograd hooks)')
    ap.add_argument('--layer-table', default='', help='write the per-launch conv table (CSV) of the profiled steps here')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(_self_launch(args))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP kernels have no CPU fallback')
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.hip import ops
    from packnet_sfm.rccl import hvd

    hvd.init()
    rank, world = hvd.rank(), hvd.size()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    ndev = torch.cuda.device_count()
    device = torch.device('cuda', hvd.local_rank() % ndev)
    torch.cuda.set_device(device)

    H, W, B = args.height, args.width, args.batch
    model = build_model(device, args.depth_net)
    batch = synthetic_batch(B, H, W, 1234 + rank, device)
    force_ddp = os.environ.get('PNSFM_FORCE_DDP') == '1'   # single-GPU rehearsal of the N>1 path (1-rank RCCL group)
    ddp = world > 1 or force_ddp
    use_graph = args.graph == 'on'
    if use_graph and ddp:
        raise SystemExit('--graph on is for 1 GPU (collectives are launched from autograd hooks, outside any capture)')
    groups = [{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0},
              {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0}]
    if args.optimizer == 'flat':
        from packnet_sfm.rccl.flat_adam import FlatAdam
        optimizer = FlatAdam(groups)
    else:
        optimizer = torch.optim.Adam(groups, fused=True, capturable=use_graph)
    if ddp:
        optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters(),
                                             compression=hvd.Compression.none, force_collectives=force_ddp)

    def eager_step():
        optimizer.zero_grad()
        out = model(batch, progress=0.0)
        out['loss'].backward()
        optimizer.step()
        return out['loss']

    def fence():
        if ddp:
            dist.barrier()
        torch.cuda.synchronize()

    # The library autotunes every layer shape on first use (timing synchronises, so it cannot happen inside a capture) and
    # the caching allocator settles over the first steps: two untimed EAGER steps always run first, whatever --warmup says.
    for _ in range(2):
        eager_step()
    fence()
    step = eager_step
    if use_graph:
        from packnet_sfm.hip.graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, optimizer, batch, progress=0.0)
        step = lambda: graphed(batch)      # noqa: E731  (copies the batch into the static inputs, draws the flip, replays)
    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.detach().float().item())

    # ---- roofline of the dominant kernel: per-launch hipEvent timing inside the library (pnsfm_prof_*), on the stream
    # each kernel is launched on.  Events cannot bracket the nodes of a replayed graph, so the SAME step is run eagerly
    # for a few extra steps right after the timed region: (a) as trained -- weight gradients overlapped with the
    # data-gradient chain on the side stream, i.e. durations of kernels SHARING the GPU -- and (b) with the side stream
    # off, i.e. each kernel alone (kernel quality).  rocprofv3 of this command shows the same kernels in both regions.
    timed = iso = None
    if not args.no_prof:
        def profiled(nsteps):
            eager_step()
            fence()
            ops.prof_reset()
            ops.prof_enable(True)
            for _ in range(nsteps):
                eager_step()
            fence()
            ops.prof_enable(False)
            return ops.prof_collect(0), ops.prof_collect(1)
        timed = profiled(3)
        if rank == 0 and args.layer_table:
            ops.prof_dump(args.layer_table)
        was = HF._WgradStream.enabled
        if was:                 # weight gradients on a side stream: measure the kernels alone as well
            HF.set_wgrad_stream(False)
            iso = profiled(2)
            HF.set_wgrad_stream(True)
        else:                   # (default) every kernel already runs alone on the compute stream
            iso = timed
    if rank == 0:
        images = B * world * args.steps
        value = images / elapsed
        scale = (H * W) / (192.0 * 640.0)
        roofline = None
        traffic = measured_traffic()
        if timed is not None:
            (ms0, fl0, n0), (ms1, fl1, n1) = timed     # conv2d_mfma_kernel (forward + backward-data), conv2d_wgrad_kernel
            (ims0, ifl0, in0), (ims1, ifl1, in1) = iso
            if n0 > 0 and ms0 > 0:
                ach = fl0 / (ms0 * 1e-3) / 1e12
                # flops the conv kernels actually EXECUTE per step (the Conv3d*Conv2d collapse removes ~35 % of the
                # reference's 1 232 GFLOP/image) -> utilisation of the matrix pipe over the whole step
                exec_gflop_step = (fl0 + fl1) / 3.0 / 1e9
                # Arithmetic of the forward / backward-data kernels.  'bx3': fp32 rebuilt on the bf16 matrix pipe (exact 3-way
                # bf16 split of every operand, 6 of the 9 piece products, fp32 accumulate: csrc/conv2d_bx3.h) -- each
                # algorithmic MAC costs 6 bf16 MACs, so the pipe's ceiling in ALGORITHMIC fp32 flops is 2500 / 6 TFLOP/s.
                # 'f32': v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s.
                bx3 = HF.get_conv_math() == 'bx3'
                peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if bx3 else FP32_MFMA_PEAK_TFLOPS
                roofline = {
                    'bound': 'mfma',
                    'kernel': ('conv2d_bx3_kernel (fwd + dgrad implicit GEMM, fp32 from 6 bf16 MFMA products)' if bx3
                               else 'conv2d_mfma_kernel (fwd + dgrad implicit GEMM)'),
                    'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4),
                    'peak_detail': ('algorithmic fp32 flops against the bf16 dense MFMA peak (2500 TFLOP/s) / 6 products per MAC; '
                                    'executed bf16 rate = 6 x achieved = %.0f TFLOP/s; the 6-product instruction stream alone '
                                    'sustains 1838 TFLOP/s bf16 = 306 fp32-equivalent on this part (tools/micro/bf16x3_check.hip)'
                                    % (6 * ach)) if bx3 else 'v_mfma_f32_32x32x2_f32 dense peak',
                    'vs_f32_mfma_peak': round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                    'measured_in': '3 eager steps right after the timed region (same kernels and shapes; events cannot '
                                   'bracket nodes of a replayed hipGraph)' if use_graph else '3 eager steps after the timed region',
                    # HBM bytes per launch (FETCH_SIZE + WRITE_SIZE PMC passes, profiles/rNN_traffic.json) or null
                    'traffic': (traffic or {}).get('hbm_bytes_per_launch'), 'traffic_detail': traffic,
                    'launches': int(n0), 'avg_launch_ms': round(ms0 / n0, 4),
                    'flop_per_launch_avg': round(fl0 / n0, 1),
                    'wgrad_kernel': {'achieved': round(fl1 / (ms1 * 1e-3) / 1e12, 2) if ms1 > 0 else None,
                                     'launches': int(n1), 'avg_launch_ms': round(ms1 / max(n1, 1), 4)},
                    'isolated': {       # same kernels with the weight-gradient side stream off (= the timed kernels by default)
                        'achieved': round(ifl0 / (ims0 * 1e-3) / 1e12, 2) if ims0 > 0 else None,
                        'frac': round(ifl0 / (ims0 * 1e-3) / 1e12 / peak, 4) if ims0 > 0 else None,
                        'vs_f32_mfma_peak': round(ifl0 / (ims0 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if ims0 > 0 else None,
                        'avg_launch_ms': round(ims0 / max(in0, 1), 4),
                        'wgrad_achieved': round(ifl1 / (ims1 * 1e-3) / 1e12, 2) if ims1 > 0 else None},
                    'whole_step_vs_mfma_peak': {
                        'reference_flops': round(value * GFLOP_PER_IMAGE_192x640 * scale / 1e3 / world / FP32_MFMA_PEAK_TFLOPS, 4),
                        'executed_flops': round(exec_gflop_step / 1e3 / (elapsed / args.steps) / FP32_MFMA_PEAK_TFLOPS, 4),
                        'executed_gflop_per_step': round(exec_gflop_step, 1)},
                }
        backend = dist.get_backend() if dist.is_initialized() else None
        shape_tag = ('BASELINE.json configs[1]' if (H, W, B) == (192, 640, 4) else
                     ('BASELINE.json configs[2] shape' if (H, W, B) == (384, 1280, 2) else 'custom shape'))
        result = {
            'metric': 'images/sec %s self-sup train %dx%d' % (args.depth_net, H, W), 'value': round(value, 3), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': args.depth_net + '(1A)+PoseNet self-supervised train step (fwd+photometric loss+bwd+allreduce+Adam), '
                                   'KITTI-shaped %dx%d triplets, batch %d/GPU (%s)' % (H, W, B, shape_tag),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'final_loss': round(loss_val, 6),
                       'wgrad_side_stream': bool(HF._WgradStream.enabled),
                       'tuning': ('user database %s' % os.environ['PNSFM_TUNE_DB']) if os.environ.get('PNSFM_TUNE_DB') else
                                 ('shipped database (%d decisions) + autotune for unlisted shapes' % ops.tune_shipped_entries()
                                  if ops.tune_shipped_entries() else 'autotune during warm-up'),
                       'step_launch': 'hipGraph replay (one graph per flip state)' if use_graph else 'eager',
                       'collective_backend': backend, 'devices_visible': ndev},
            'roofline': roofline,
        }
        if world > ndev:
            result['config']['note'] = ('%d ranks share %d device(s): functional rehearsal of the N>1 path over gloo, not a '
                                        'scaling measurement' % (world, ndev))
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(H, W)
        try:        # RCCL prints a version banner through C stdio (block-buffered when stdout is a file): push it out FIRST
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(result), flush=True)
    if ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
