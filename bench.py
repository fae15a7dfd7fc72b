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


def _cpu_step_fn(H, W, batch_size):
    """One CPU training step (fwd + loss + bwd + Adam) of the path, as a closure, and what ran it:
    'reference' = the reference's own modules imported from /root/reference (only where that checkout exists: this container;
    oracle/_refstubs.py stubs the third-party imports it lacks), 'port' = oracle/packnet_oracle.py, the restatement of the same
    algorithm on stock torch CPU ops (the GPU box has no /root/reference)."""
    batch = synthetic_batch(batch_size, H, W, 1234, 'cpu')
    if os.path.isdir('/root/reference') and os.environ.get('PNSFM_CPU_BASELINE', '') != 'port':
        try:
            return _reference_step_fn(batch), 'reference'
        except Exception as e:              # the oracle still gives a baseline
            print('cpu_baseline: reference import failed (%s); timing the port' % e, file=sys.stderr)
    from oracle import packnet_oracle as O
    sd = {k: v.requires_grad_(True) for k, v in O.init_params(O.packnet01_param_shapes('1A'), seed=42).items()}
    psd = {k: v.requires_grad_(True) for k, v in O.init_params(O.posenet_param_shapes(2), seed=43).items()}
    opt = torch.optim.Adam([{'params': list(sd.values()), 'lr': 2e-4}, {'params': list(psd.values()), 'lr': 2e-4}])
    kw = {k: LOSS_DEFAULTS[k] for k in ('num_scales', 'ssim_loss_weight', 'smooth_loss_weight', 'C1', 'C2',
                                         'photometric_reduce_op', 'automask_loss')}

    def step():
        opt.zero_grad()
        out = O.selfsup_forward(sd, psd, batch, flip=False, **kw)
        out['loss'].sum().backward()
        opt.step()
    return step, 'port'


def _reference_step_fn(batch):
    """The reference's SelfSupModel + PackNet01 + PoseNet on CPU, run from a SUBPROCESS-free import: the reference package has the
    same name as ours (packnet_sfm), so it is loaded under a private alias by temporarily swapping sys.modules."""
    import importlib
    saved = {k: v for k, v in sys.modules.items() if k == 'packnet_sfm' or k.startswith('packnet_sfm.')}
    saved_path = list(sys.path)
    for k in saved:
        del sys.modules[k]
    try:
        sys.path.insert(0, '/root/reference')
        sys.dont_write_bytecode = True
        from oracle import _refstubs
        _refstubs.install()
        RefSelfSup = importlib.import_module('packnet_sfm.models.SelfSupModel').SelfSupModel
        RefPackNet01 = importlib.import_module('packnet_sfm.networks.depth.PackNet01').PackNet01
        RefPoseNet = importlib.import_module('packnet_sfm.networks.pose.PoseNet').PoseNet
        torch.manual_seed(42)
        model = RefSelfSup(**LOSS_DEFAULTS)
        model.add_depth_net(RefPackNet01(dropout=0.0, version='1A'))
        model.add_pose_net(RefPoseNet(nb_ref_imgs=2, rotation_mode='euler'))
        model.train()
        ref_mods = {k: v for k, v in sys.modules.items() if k == 'packnet_sfm' or k.startswith('packnet_sfm.')}
    finally:
        for k in [k for k in sys.modules if k == 'packnet_sfm' or k.startswith('packnet_sfm.')]:
            del sys.modules[k]
        sys.modules.update(saved)
        sys.path[:] = saved_path
    opt = torch.optim.Adam([{'params': model.depth_net.parameters(), 'lr': 2e-4}, {'params': model.pose_net.parameters(), 'lr': 2e-4}])

    def step():
        # the reference's lazy imports (inside forward) must resolve to ITS modules
        mine = {k: v for k, v in sys.modules.items() if k == 'packnet_sfm' or k.startswith('packnet_sfm.')}
        for k in mine:
            del sys.modules[k]
        sys.modules.update(ref_mods)
        try:
            random.seed(0)
            opt.zero_grad()
            out = model(batch, progress=0.0)
            out['loss'].sum().backward()
            opt.step()
            ref_mods.update({k: v for k, v in sys.modules.items() if k == 'packnet_sfm' or k.startswith('packnet_sfm.')})
        finally:
            for k in [k for k in sys.modules if k == 'packnet_sfm' or k.startswith('packnet_sfm.')]:
                del sys.modules[k]
            sys.modules.update(mine)
    return step


def _time_cpu_steps(step, warmup, timed):
    for _ in range(warmup):
        step()
    out = []
    for _ in range(timed):
        t0 = time.time()
        step()
        out.append(time.time() - t0)
    return out


def cpu_baseline(H, W, seconds_budget=75.0):
    """CPU baseline on the host cores of this box (rank 0, N=1 only): full training steps (fwd + loss + bwd + Adam) of the
    reference path -- the reference's own modules where /root/reference exists (`kind: "reference"`), else the oracle port.
    Bounded sample (BASELINE.md 3 / SURVEY 8d: median of >= 3 steps after a warm-up):
      1. the intra-op thread count is SWEPT at batch 1 (8/16/32/64, capped at the cores present; 1 warm-up + 1 timed step
         each -- oversubscribing all 128+ hardware threads measured 3x slower than 8 threads in round 1);
      2. at the best count the batch-4 step (the bench workload's batch) is timed 1 warm-up + 3 steps: `value` = 4 / MEDIAN.
    If the budget runs out before step 2 completes, the batch-1 leg at the best count is extended to 3 timed steps instead."""
    ncpu = os.cpu_count() or 8
    saved = torch.get_num_threads()
    t_start = time.time()
    sweep, kind = {}, None
    step1, kind = _cpu_step_fn(H, W, 1)
    first = True
    for th in [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        sweep[th] = 1.0 / _time_cpu_steps(step1, 1 if first else 0, 1)[0]
        first = False
        if time.time() - t_start > 0.35 * seconds_budget:
            break
    best_th = max(sweep, key=sweep.get)
    torch.set_num_threads(best_th)
    est_b4 = 4.0 / sweep[best_th] * 0.75 * 4        # ~4 batch-4 steps at (slightly better than) 4x the batch-1 time
    times, nb = None, 4
    if time.time() - t_start + est_b4 < seconds_budget:
        step4, _ = _cpu_step_fn(H, W, 4)
        times = _time_cpu_steps(step4, 1, 3)
    else:
        nb = 1
        times = _time_cpu_steps(step1, 0, 3)
    med = sorted(times)[len(times) // 2]
    torch.set_num_threads(saved)
    return {'value': round(nb / med, 4), 'unit': 'images/sec', 'cores': best_th, 'kind': kind,
            # the REFERENCE's own modules (kind "reference") cannot run on the GPU box (/root/reference is absent there); what they
            # measured in the build container (8 vCPU Xeon @2.1 GHz, torch 2.10 CPU, 8 threads, batch 1, best of 3 steps)
            'reference_kind_in_build_container': {'value': 0.30, 'unit': 'images/sec', 'cores': 8, 'batch': 1,
                                                  'source': 'BASELINE.md section 3 (survey probe: 3.32 s per 192x640 step)'},
            'thread_sweep_batch1': {str(k): round(v, 4) for k, v in sweep.items()},
            'timed_steps_s': [round(t, 3) for t in times], 'batch': nb, 'host_cpus': ncpu,
            'sample': '%s, full train step (fwd+loss+bwd+Adam), %dx%d: thread sweep at batch 1 (1 timed step per count), then '
                      'batch %d at %d threads: 1 warm-up + %d timed steps, value = batch / median'
                      % ('the reference itself (/root/reference modules on torch CPU fp32)' if kind == 'reference' else
                         'oracle port (torch CPU fp32 restatement of the reference path; no /root/reference on this box)',
                         H, W, nb, best_th, len(times))}


def gpu_eager_baseline(H, W, B, device, seconds_budget=90.0, wait_for_go=False):
    """Like-for-like GPU baseline: the reference path's math through STOCK PyTorch-ROCm eager
    ops (MIOpen / ATen, fp32) on this same MI355X -- the full training step incl. torch.optim.Adam.  The GPU box has no
    /root/reference, so the ops are driven by oracle/packnet_oracle.py, the torch restatement of the reference's modules that
    the parity tests pin against the reference (the same role it has in `cpu_baseline`; a reported baseline, never the product).
    Bounded: 1 warm-up step (MIOpen picks its kernels there) + up to 3 timed steps inside `seconds_budget`."""
    from oracle import packnet_oracle as O
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    batch = synthetic_batch(B, H, W, 1234, device)
    sd = {k: v.to(device).requires_grad_(True) for k, v in O.init_params(O.packnet01_param_shapes('1A'), seed=42).items()}
    psd = {k: v.to(device).requires_grad_(True) for k, v in O.init_params(O.posenet_param_shapes(2), seed=43).items()}
    opt = torch.optim.Adam([{'params': list(sd.values()), 'lr': 2e-4}, {'params': list(psd.values()), 'lr': 2e-4}])
    kw = {k: LOSS_DEFAULTS[k] for k in ('num_scales', 'ssim_loss_weight', 'smooth_loss_weight', 'C1', 'C2',
                                         'photometric_reduce_op', 'automask_loss')}

    def step():
        opt.zero_grad()
        out = O.selfsup_forward(sd, psd, batch, flip=False, **kw)
        out['loss'].sum().backward()
        opt.step()

    t_start = time.perf_counter()
    step()
    torch.cuda.synchronize()
    warm = time.perf_counter() - t_start
    if wait_for_go:
        # the parent times its CPU baseline on the host cores meanwhile: the timed steps below are host-launch-bound eager steps
        # and must not share the cores with it -- block until the parent says go (or closes the pipe)
        sys.stdin.readline()
        t_start = time.perf_counter()
        step()                       # one more untimed step: clocks and caches back to the loaded state after the wait
        torch.cuda.synchronize()
    times = []
    while len(times) < 3 and (not times or time.perf_counter() - t_start + times[-1] < seconds_budget):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    del opt, sd, psd
    torch.cuda.empty_cache()
    return {'value': round(B / med, 3), 'unit': 'images/sec', 'kind': 'port',
            'ms_per_step': round(1e3 * med, 2), 'timed_steps_s': [round(t, 4) for t in times], 'warmup_step_s': round(warm, 2),
            'sample': 'stock PyTorch-ROCm eager ops (MIOpen / ATen, fp32, no TF32) driven by the oracle port of the reference path, '
                      'full train step (fwd+loss+bwd+Adam), %dx%d batch %d on this GPU: 1 warm-up + %d timed steps, value = batch / median'
                      % (H, W, B, len(times))}


def gpu_eager_baseline_start(H, W, B):
    """Start `gpu_eager_baseline` in a child process (its own session, so that an overrun can be killed by process group -- that
    group and nothing else).  The child runs while the parent times the CPU baseline (host cores only), which hides most of MIOpen's
    first-use kernel compilation (~260 s on a fresh box) behind work the default run does anyway."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--gpu-baseline-worker', '%d,%d,%d' % (H, W, B)]
    # MIOpen's first-use compilation is host work: the child is confined to the LAST cores of the box (at most 8, at most a quarter
    # of them) so that the CPU baseline leg, which runs meanwhile on up to 64 threads, keeps cores of its own
    ncpu = os.cpu_count() or 8
    keep = max(1, min(8, ncpu // 4))
    cpus = set(range(ncpu - keep, ncpu))

    def confine():
        try:
            os.sched_setaffinity(0, cpus & os.sched_getaffinity(0) or os.sched_getaffinity(0))
        except (AttributeError, OSError):
            pass
    try:
        return (subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, start_new_session=True,
                                 text=True, preexec_fn=confine), time.time())
    except OSError as e:
        return (None, 'could not start the baseline process: %s' % e)


def gpu_eager_baseline_finish(handle, H, W, B, value, budget_s):
    """Collect the child of `gpu_eager_baseline_start` (budget_s of wall clock counted from its start; None: wait).  A child that
    overruns is killed and the line quotes the figure an earlier unbounded run of this same leg recorded
    (profiles/r*_gpu_eager_baseline.json) -- as `recorded`, with the ratio named `speedup_vs_recorded` (that figure may
    come from another box or build; `speedup_of_value` is only ever computed against a value measured in THIS run)."""
    import glob
    import signal
    import subprocess
    p, t_start = handle
    res, note = None, None
    if p is None:
        note = t_start
    else:
        try:
            left = None if budget_s is None else max(1.0, budget_s - (time.time() - t_start))
            out, _ = p.communicate(input='go\n', timeout=left)      # the CPU leg is over: the child may time its steps now
            line = [ln for ln in out.splitlines() if ln.startswith('{')]
            res = json.loads(line[-1]) if line else None
            if res is None:
                note = 'the baseline process printed no result (exit code %s)' % p.returncode
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            p.communicate()
            note = 'not finished within %.0f s (MIOpen compiles its kernels on first use on a fresh box)' % budget_s
    if res is None:
        res = {'value': None, 'unit': 'images/sec', 'note': note}
        for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_gpu_eager_baseline*.json')), reverse=True):
            try:
                rec = json.load(open(f))
                if tuple(rec.get('workload_shape', (192, 640, 4))) == (H, W, B):
                    res['recorded'] = {k: rec[k] for k in ('value', 'ms_per_step', 'sample') if k in rec}
                    res['recorded']['source'] = os.path.relpath(f, ROOT)
                    break
            except Exception:
                continue
    if res.get('value'):
        res['speedup_of_value'] = round(value / res['value'], 2)
        res['measured'] = ('in this run, on this GPU: a child process whose MIOpen warm-up step ran beside the cpu_baseline leg (confined '
                           'to the last <= 8 host cores) and whose timed steps ran after that leg had finished')
    elif (res.get('recorded') or {}).get('value'):
        res['speedup_vs_recorded'] = round(value / res['recorded']['value'], 2)
    return res


def csrc_sha():
    """sha256 (first 16 hex digits) over the kernel sources the running library was built from (csrc/*.hip, *.h, sorted)."""
    import hashlib
    d = os.path.join(ROOT, 'packnet-sfm_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith('.hip') or f.endswith('.h'):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def measured_traffic(H, W, B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json; PMC
    counters cannot be collected from inside this process).  Only returned for the workload the passes were collected on
    (192x640 batch 4 unless the file says otherwise) AND for the kernel sources they were collected on: a profile whose `csrc_sha`
    differs from the running build's is refused (None) -- the number must describe the kernel that ran.  FETCH_SIZE is corrected
    with the factor calibrated on this part (tools/micro/fetch_calib.hip, profiles/r04_pmc_calibration.json: 2.0 for 4-byte and
    16-byte per-lane streams and LDS-DMA alike; WRITE_SIZE 1.0)."""
    import glob
    sha = csrc_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic*.json')), reverse=True):      # latest round first
        try:
            d = json.load(open(f))
            if tuple(d.get('workload_shape', (192, 640, 4))) != (H, W, B):
                continue
            if d.get('csrc_sha') != sha:
                continue
            # the forward / backward-data implicit GEMM is two kernel names since round 5 (conv2d_bx3_kernel and the ping-pong
            # workgroup conv2d_bx3pp_kernel): launch-weighted mean over both; the algorithmic bytes are the layer table's average
            # over ALL forward / backward-data launches (tools/pmc_traffic.py files them under conv2d_bx3_kernel)
            names = [n for n in ('pnsfm::conv2d_bx3_kernel', 'pnsfm::conv2d_bx3pp_kernel') if n in d] or ['pnsfm::conv2d_mfma_kernel']
            ks = [d[n] for n in names]
            nl = sum(k.get('launches_in_pass', 1) for k in ks)
            mean = lambda key: sum(k.get(key, 0) * k.get('launches_in_pass', 1) for k in ks) / nl
            alg = ks[0].get('algorithmic_bytes_per_launch', 0)
            hbm = mean('hbm_bytes_per_launch')
            return {'hbm_bytes_per_launch': round(hbm), 'algorithmic_bytes_per_launch': round(alg),
                    'ratio_vs_algorithmic': round(hbm / alg, 3) if alg else None,
                    'fetch_bytes_corrected': round(mean('fetch_bytes_per_launch')), 'write_bytes': round(mean('write_bytes_per_launch')),
                    'kernels': {n: {'launches': k.get('launches_in_pass'), 'hbm_bytes_per_launch': round(k['hbm_bytes_per_launch'])} for n, k in zip(names, ks)},
                    'fetch_correction_factor': d.get('fetch_factor'), 'csrc_sha': sha, 'source': os.path.relpath(f, ROOT)}
        except Exception:
            continue
    return None


def _gpu_sysfs_dir(device):
    """hwmon directory of `device` in sysfs (freq1_input = sclk in Hz, power1_input = socket power in uW), found through the PCI
    address torch reports; None when the box does not expose it."""
    import glob
    try:
        pr = torch.cuda.get_device_properties(device)
        addr = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        hw = glob.glob('/sys/bus/pci/devices/%s/hwmon/hwmon*' % addr)
        return hw[0] if hw and os.path.exists(os.path.join(hw[0], 'freq1_input')) else None
    except Exception:
        return None


def pin_to_gpu_numa(device):
    """Keep this process (and the autograd threads it creates later) on the host cores next to `device` (sysfs local_cpulist of the
    GPU's PCI function): the host enqueue time of one step scattered 9-18 ms from process to process on one box depending on where the
    scheduler had put the process.  Returns (cpus pinned to | None, all cpus before)."""
    try:
        before = os.sched_getaffinity(0)
        pr = torch.cuda.get_device_properties(device)
        addr = '%04x:%02x:%02x.0' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        txt = open('/sys/bus/pci/devices/%s/local_cpulist' % addr).read().strip()
        cpus = set()
        for part in txt.split(','):
            if '-' in part:
                a, b = part.split('-')
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= before
        # (one hardware thread per core: on these hosts the second half of the CPU numbers are the SMT siblings of the first half)
        half = (os.cpu_count() or 0) // 2
        phys = {c for c in cpus if c < half}
        if len(phys) >= 4 and os.environ.get('PNSFM_PIN_SMT', '0') != '1':
            cpus = phys
        if len(cpus) >= 4:
            os.sched_setaffinity(0, cpus)
            return sorted(cpus), before
        return None, before
    except Exception:
        return None, None


class ClockSampler:
    """Shader clock and socket power of the GPU, read from sysfs by a background thread every 20 ms while a region runs (one pread
    of two tiny files per sample: ~30 us of host time, 0.15 % of a core)."""

    def __init__(self, device):
        self.dir = _gpu_sysfs_dir(device)
        self.rows = []
        self._stop = None

    def _read(self):
        try:
            f = int(open(os.path.join(self.dir, 'freq1_input')).read()) / 1e6
            pw = int(open(os.path.join(self.dir, 'power1_input')).read()) / 1e6
            return f, pw
        except Exception:
            return None

    def __enter__(self):
        import threading
        if self.dir is None:
            return self
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                r = self._read()
                if r:
                    self.rows.append(r)
                self._stop.wait(0.02)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()
        return self

    def __exit__(self, *exc):
        if self._stop is not None:
            self._stop.set()
            self._th.join()
        return False

    def summary(self):
        if not self.rows:
            return None
        fr, pw = [r[0] for r in self.rows], [r[1] for r in self.rows]
        return {'sclk_mhz_avg': round(sum(fr) / len(fr), 1), 'sclk_mhz_min': round(min(fr), 1), 'sclk_mhz_max': round(max(fr), 1),
                'power_w_avg': round(sum(pw) / len(pw), 1), 'power_w_max': round(max(pw), 1), 'samples': len(fr),
                'source': 'sysfs hwmon freq1_input / power1_input of this GPU, one sample per 20 ms'}


def box_calibration(device):
    """What THIS box sustains, measured right before the timed region (the driver's lease and the builder's leases
    of the same SKU differ by 5 %, more than a round of kernel work moves): (a) the bare six-product bf16 MFMA stream of the split
    arithmetic, operands in registers (csrc/calib.hip; ~0.5 s of it, the sustained second half quoted) in fp32-equivalent TFLOP/s -- the ceiling the conv kernels
    would reach with nothing but their MFMAs; (b) a 1 GiB -> 1 GiB float4 streaming copy in GB/s; (c) shader clock and socket power
    while (a) ran.  `roofline.frac_vs_box` divides by (a) instead of the guide's 2500 / 6."""
    from packnet_sfm.hip import ops
    sink = torch.empty(1024 * 256, dtype=torch.float32, device=device)
    ops.calib_mfma(sink, 1024, 200)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # ~0.5 s of the loop in 8 segments of 8 launches (~7 ms each): a burst of a few tens of ms runs at clocks the power limit has not
    # caught up with yet (372 fp32-equivalent TFLOP/s measured that way against 306 sustained), so the figure quoted is the rate over
    # the LAST half of the run and the first segment is kept next to it as `mfma_tflops_burst`
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    per_seg = 0.0
    with ClockSampler(device) as cs:
        evs[0].record()
        for seg in range(8):
            per_seg = 0.0
            for _ in range(8):
                per_seg += ops.calib_mfma(sink, 1024, 4000)
            evs[seg + 1].record()
        torch.cuda.synchronize(device)
    ms = evs[4].elapsed_time(evs[8])
    bf16_tf = 4 * per_seg / (ms * 1e-3) / 1e12
    burst_tf = per_seg / (evs[0].elapsed_time(evs[1]) * 1e-3) / 1e12
    ms_total = evs[0].elapsed_time(evs[8])
    n = (1 << 30) // 4
    src = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
    dst = torch.empty_like(src)
    best = 0.0
    for i in range(4):
        e0.record()
        nbytes = ops.calib_copy(src, dst)
        e1.record()
        torch.cuda.synchronize(device)
        if i:
            best = max(best, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst, sink
    torch.cuda.empty_cache()
    return {'mfma_tflops': round(bf16_tf / 6.0, 1), 'mfma_bf16_tflops': round(bf16_tf, 1), 'mfma_tflops_burst': round(burst_tf / 6.0, 1),
            'mfma_loop_ms': round(ms_total, 1),
            'hbm_gbps': round(best, 1),
            'clocks_during_mfma_loop': cs.summary(),
            'what': 'six-product v_mfma_f32_32x32x16_bf16 stream with register operands (fp32-equivalent = bf16 rate / 6) and a 1 GiB '
                    'float4 streaming copy (read + write bytes), both on this GPU right before the timed region (csrc/calib.hip)'}


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


def roofline_of(timed, iso, value, H, W, B, world, ms_per_step, nprof):
    """`roofline` object of one workload from the library's per-launch event timing (pnsfm_prof_*: hipEvents recorded on the
    launch stream around every conv launch of `nprof` eager steps)."""
    from packnet_sfm.hip import functional as HF
    (ms0, fl0, n0), (ms1, fl1, n1) = timed         # forward + backward-data launches, weight-gradient launches
    (ims0, ifl0, in0), (ims1, ifl1, in1) = iso
    if not (n0 > 0 and ms0 > 0):
        return None
    scale = (H * W) / (192.0 * 640.0)
    # `achieved` / `frac` describe the KERNEL: with the side streams on (weight gradients, pose branch: the default since round 5)
    # an event pair around a launch also counts the time the launch waited for compute units another stream's workgroups held, so
    # the headline figures come from the pass with the side streams off (each kernel alone on the GPU -- what rocprofv3's
    # serialised PMC passes see too) and the as-run figures of the training configuration are kept next to them (`as_run`).
    streams_on = iso is not timed
    as_run = fl0 / (ms0 * 1e-3) / 1e12
    as_run_w = fl1 / (ms1 * 1e-3) / 1e12 if ms1 > 0 else None
    if streams_on and ims0 > 0:
        (ms0, fl0, n0), (ms1, fl1, n1) = (ims0, ifl0, in0), (ims1, ifl1, in1)
    ach = fl0 / (ms0 * 1e-3) / 1e12
    # flops the conv kernels actually EXECUTE per step (the Conv3d*Conv2d collapse removes ~35 % of the reference's
    # 1 232 GFLOP/image) -> utilisation of the matrix pipe over the whole step
    exec_gflop_step = (fl0 + fl1) / float(nprof) / 1e9
    # Arithmetic of the conv kernels.  'bx3': fp32 rebuilt on the bf16 matrix pipe (exact 3-way bf16 split of every operand, 6
    # of the 9 piece products, fp32 accumulate: csrc/conv2d_bx3.h) -- each algorithmic MAC costs 6 bf16 MACs, so the pipe's
    # ceiling in ALGORITHMIC fp32 flops is 2500 / 6 TFLOP/s.  'f32': v_mfma_f32_32x32x2_f32, 157.3 TFLOP/s.
    bx3 = HF.get_conv_math() == 'bx3'
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if bx3 else FP32_MFMA_PEAK_TFLOPS
    traffic = measured_traffic(H, W, B)
    return {
        'bound': 'mfma',
        'kernel': ('conv2d_bx3_kernel (fwd + dgrad implicit GEMM, fp32 from 6 bf16 MFMA products)' if bx3
                   else 'conv2d_mfma_kernel (fwd + dgrad implicit GEMM)'),
        'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
        'peak_detail': ('algorithmic fp32 flops against the bf16 dense MFMA peak (2500 TFLOP/s) / 6 products per MAC; '
                        'executed bf16 rate = 6 x achieved = %.0f TFLOP/s; the 6-product instruction stream alone '
                        'sustains 1838 TFLOP/s bf16 = 306 fp32-equivalent on this part (tools/micro/bf16x3_check.hip)'
                        % (6 * ach)) if bx3 else 'v_mfma_f32_32x32x2_f32 dense peak',
        'vs_f32_mfma_peak': round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
        'measured_in': ('isolated pass: %d steps right after the timed region with the weight-gradient and pose-branch side streams '
                        'switched OFF (same launches, each kernel alone; hipEvents on the launch stream around each one); the '
                        'as-run pass with the side streams on is `as_run`' % nprof) if streams_on else
                       '%d steps right after the timed region (same launches; events on the launch stream around each one)' % nprof,
        'as_run': {'achieved': round(as_run, 2), 'frac': round(as_run / peak, 4),
                   'wgrad_achieved': round(as_run_w, 2) if as_run_w else None,
                   'note': 'event pairs around launches that SHARE the GPU with the other streams\' kernels'} if streams_on else None,
        # HBM bytes per launch (FETCH_SIZE + WRITE_SIZE PMC passes, profiles/rNN_traffic.json) or null
        'traffic': (traffic or {}).get('hbm_bytes_per_launch'), 'traffic_detail': traffic,
        'launches': int(n0), 'avg_launch_ms': round(ms0 / n0, 4), 'flop_per_launch_avg': round(fl0 / n0, 1),
        'wgrad_kernel': {'achieved': round(fl1 / (ms1 * 1e-3) / 1e12, 2) if ms1 > 0 else None,
                         'frac': round(fl1 / (ms1 * 1e-3) / 1e12 / peak, 4) if ms1 > 0 else None,
                         'launches': int(n1), 'avg_launch_ms': round(ms1 / max(n1, 1), 4)},
        'whole_step_vs_mfma_peak': {
            'reference_flops': round(value * GFLOP_PER_IMAGE_192x640 * scale / 1e3 / world / FP32_MFMA_PEAK_TFLOPS, 4),
            'executed_flops': round(exec_gflop_step / 1e3 / (ms_per_step * 1e-3) / FP32_MFMA_PEAK_TFLOPS, 4),
            'executed_gflop_per_step': round(exec_gflop_step, 1)},
    }


def run_workload(model, optimizer, H, W, B, steps, warmup, ctx, want_prof=True, layer_table=''):
    """W untimed warm-up steps, then exactly K timed steps between barrier + synchronize fences; max over ranks.  Returns the
    measurement (elapsed seconds, final loss, conv-launch timing of a few extra eager steps)."""
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.hip import ops
    rank, world, device, ddp = ctx['rank'], ctx['world'], ctx['device'], ctx['ddp']
    batch = synthetic_batch(B, H, W, 1234 + rank, device)

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

    # The library autotunes every layer shape on first use (timing synchronises) and the caching allocator settles over the
    # first steps: two untimed steps always run first, whatever --warmup says.
    for _ in range(2):
        eager_step()
    fence()
    step = eager_step
    for _ in range(warmup):
        loss = step()
    fence()
    reducer = getattr(optimizer, '_reducer', None)
    if reducer is not None:
        reducer.exposed_reset(True)        # events around the end-of-backward join: all-reduce time NOT hidden behind backward
    sampler = ClockSampler(device) if rank == 0 else None
    if sampler is not None:
        sampler.__enter__()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if sampler is not None:
        sampler.__exit__()
    exposed = reducer.exposed_ms() if reducer is not None else None
    host_us = reducer.host_us_per_collective() if reducer is not None else None
    if reducer is not None:
        reducer.exposed_reset(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res = {'elapsed': elapsed, 'loss': float(loss.detach().float().item()), 'timed': None, 'iso': None, 'nprof': 3,
           'clocks': sampler.summary() if sampler is not None else None,
           'exposed_allreduce_ms_per_step': (round(exposed / steps, 4) if exposed is not None else None),
           'host_us_per_collective': (round(host_us, 1) if host_us is not None else None)}
    # Host-issue headroom: wall time of the Python + ctypes + hipLaunchKernel work that ENQUEUES one step, measured
    # with an empty GPU queue in front of it (fence, then one step, clock stopped when the last launch call returns -- nothing in the
    # step synchronises, and one step's ~800 launches never fill the queue, so the GPU cannot push back).  If this approaches
    # ms_per_step the step is host-bound; the GPU-side figure to compare with is ms_per_step itself.
    issue = []
    for _ in range(7):          # (median of 7: single steps scatter by +-20 % -- allocator state, the autograd thread's wake-up)
        fence()
        ti = time.perf_counter()
        step()
        issue.append(time.perf_counter() - ti)
    fence()
    res['host_issue_ms'] = 1e3 * sorted(issue)[3]
    res['host_issue_ms_min'] = 1e3 * min(issue)

    # ---- roofline of the dominant kernel: per-launch hipEvent timing inside the library (pnsfm_prof_*), on the stream
    # each kernel is launched on, over a few extra steps right after the timed region: (a) as trained and (b), when weight
    # gradients run on a side stream, with that stream off, i.e. each kernel alone (kernel quality).  rocprofv3 of this command
    # shows the same kernels.
    if want_prof:
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
        res['timed'] = profiled(res['nprof'])
        ws_on, br_on = bool(HF._WgradStream.enabled), bool(HF._BRANCH_ON)
        if ws_on or br_on:     # side streams (the default): measure the kernels alone as well
            HF.set_wgrad_stream(False)
            HF.set_branch_stream(False)
            res['iso'] = profiled(res['nprof'])
            HF.set_wgrad_stream(ws_on)
            HF.set_branch_stream(br_on)
        else:                           # every kernel already runs alone on the compute stream
            res['iso'] = res['timed']
        # the per-launch table is the ISOLATED pass's (the library's buffer holds the last profiled pass): with the side streams on, a
        # backward-data launch shares the GPU with a weight-gradient launch and the event pair around either reads ~1.4x long
        if rank == 0 and layer_table:
            ops.prof_dump(layer_table)
    return res


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
    ap.add_argument('--gpu-baseline', default='auto', choices=['auto', 'on', 'off'],
                    help='stock PyTorch-ROCm eager baseline of the same step on this GPU (`gpu_eager_baseline`, N=1 only).  On a '
                         'fresh box MIOpen compiles its kernels during the first step (~4 minutes for the ~100 conv shapes), so '
                         '`auto` runs it in a child process beside the CPU baseline leg, waits up to 420 s and otherwise quotes the '
                         'figure recorded in profiles/; `on` waits without a limit')
    ap.add_argument('--gpu-baseline-worker', default='', help=argparse.SUPPRESS)
    ap.add_argument('--no-prof', action='store_true', help='skip the per-launch event timing of the conv kernels (roofline = null)')
    ap.add_argument('--no-calibration', action='store_true', help='skip the box calibration (bare MFMA stream + streaming copy, ~0.2 s)')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the short 384x1280 batch-2 measurement (BASELINE.json configs[2] shape) that the default 192x640 '
                         'single-GPU run appends to its JSON line as `extra`')
    ap.add_argument('--optimizer', default='flat', choices=['flat', 'torch'],
                    help="'flat': FlatAdam (one gfx950 adam_kernel launch per group); 'torch': torch.optim.Adam(fused=True)")
    ap.add_argument('--layer-table', default='', help='write the per-launch conv table (CSV) of the profiled steps (the isolated pass: side streams off) here')
    args = ap.parse_args()

    if args.gpu_baseline_worker:          # child of gpu_eager_baseline_bounded: one JSON line, nothing else
        h, w, b = (int(v) for v in args.gpu_baseline_worker.split(','))
        rec = gpu_eager_baseline(h, w, b, torch.device('cuda', 0), seconds_budget=1e9, wait_for_go=True)
        rec['workload_shape'] = [h, w, b]
        print(json.dumps(rec), flush=True)
        return
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
    pinned, all_cpus = pin_to_gpu_numa(device) if os.environ.get('PNSFM_PIN_NUMA', '1') != '0' else (None, None)
    model = build_model(device, args.depth_net)
    force_ddp = os.environ.get('PNSFM_FORCE_DDP') == '1'   # single-GPU rehearsal of the N>1 path (1-rank RCCL group)
    ddp = world > 1 or force_ddp
    groups = [{'name': 'Depth', 'params': list(model.depth_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0},
              {'name': 'Pose', 'params': list(model.pose_net.parameters()), 'lr': 2e-4, 'weight_decay': 0.0}]
    if args.optimizer == 'flat':
        from packnet_sfm.rccl.flat_adam import FlatAdam
        optimizer = FlatAdam(groups)
    else:
        optimizer = torch.optim.Adam(groups, fused=True)
    if ddp:
        optimizer = hvd.DistributedOptimizer(optimizer, named_parameters=model.named_parameters(),
                                             compression=hvd.Compression.none, force_collectives=force_ddp)
    ctx = {'rank': rank, 'world': world, 'device': device, 'ddp': ddp}
    calibration = box_calibration(device) if (rank == 0 and not args.no_calibration) else None

    m = run_workload(model, optimizer, H, W, B, args.steps, args.warmup, ctx, want_prof=not args.no_prof,
                     layer_table=args.layer_table)
    # configs[2] shape on the same model and optimizer, right behind the headline measurement (single GPU, default shape only)
    extra = None
    if world == 1 and not ddp and (H, W, B) == (192, 640, 4) and not args.no_extra and args.depth_net == 'PackNet01':
        extra = run_workload(model, optimizer, 384, 1280, 2, 6, 1, ctx, want_prof=not args.no_prof)

    if rank == 0:
        def line_of(meas, H, W, B, steps, warmup):
            value = B * world * steps / meas['elapsed']
            ms_step = 1e3 * meas['elapsed'] / steps
            roof = None
            if meas['timed'] is not None:
                roof = roofline_of(meas['timed'], meas['iso'], value, H, W, B, world, ms_step, meas['nprof'])
            return value, ms_step, roof

        value, ms_step, roofline = line_of(m, H, W, B, args.steps, args.warmup)
        backend = dist.get_backend() if dist.is_initialized() else None
        shape_tag = ('BASELINE.json configs[1]' if (H, W, B) == (192, 640, 4) else
                     ('BASELINE.json configs[2] shape' if (H, W, B) == (384, 1280, 2) else 'custom shape'))
        bx3 = HF.get_conv_math() == 'bx3'
        result = {
            'metric': 'images/sec %s self-sup train %dx%d' % (args.depth_net, H, W), 'value': round(value, 3), 'unit': 'images/sec',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            # host time to ENQUEUE one step (empty queue in front of it, clock stopped before any fence) and its share of the step
            'host_issue_ms_per_step': round(m['host_issue_ms'], 3), 'host_issue_frac': round(m['host_issue_ms'] / ms_step, 3),
            'host_issue_ms_min': round(m['host_issue_ms_min'], 3),
            'config': {'workload': args.depth_net + '(1A)+PoseNet self-supervised train step (fwd+photometric loss+bwd+allreduce+Adam), '
                                   'KITTI-shaped %dx%d triplets, batch %d/GPU (%s)' % (H, W, B, shape_tag),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world, 'final_loss': round(m['loss'], 6),
                       # fp32 tensors and accumulators; the conv GEMMs as 6 bf16 MFMA products of exact 3-way operand splits
                       'arithmetic': ('fp32 via 6xbf16 MFMA (exact 3-way bf16 operand split, fp32 accumulate; csrc/conv2d_bx3.h, '
                                      'conv2d_wgrad3.hip, conv2d_wgrad4.hip; Cin<16 / stride-2 layers on v_mfma_f32_32x32x2_f32)') if bx3 else
                                     'fp32 on v_mfma_f32_32x32x2_f32',
                       'optimizer': ('FlatAdam (flat arenas = all-reduce buckets, conv weight gradients written in place)'
                                     if args.optimizer == 'flat' else 'torch.optim.Adam(fused=True)'),
                       'wgrad_side_stream': bool(HF._WgradStream.enabled), 'pose_branch_stream': bool(HF._BRANCH_ON),
                       'tuning': ('user database %s' % os.environ['PNSFM_TUNE_DB']) if os.environ.get('PNSFM_TUNE_DB') else
                                 ('shipped database (%d decisions) + autotune for unlisted shapes' % ops.tune_shipped_entries()
                                  if ops.tune_shipped_entries() else 'autotune during warm-up'),
                       'step_launch': 'eager',
                       # host cores this process was kept on (the GPU's local NUMA node, sysfs local_cpulist); null: not pinned
                       'host_cpus_pinned': ('%d cpus: %d-%d' % (len(pinned), pinned[0], pinned[-1])) if pinned else None,
                       'collective_backend': backend, 'devices_visible': ndev,
                       # ranks of the RCCL communicator the gradient all-reduce ran on (null: single process, no collective)
                       'rccl_ranks': dist.get_world_size() if (dist.is_initialized() and backend == 'nccl') else None},
            'roofline': roofline,
        }
        if calibration is not None:
            calibration['clocks_during_timed_region'] = m.get('clocks')
            result['calibration'] = calibration
            if roofline is not None and calibration.get('mfma_tflops'):
                # the same achieved figure against what THIS box's matrix pipe sustains with nothing but the MFMAs (not a peak from a
                # data sheet): comparable across leases; `frac` (against the guide's 2500 / 6) stays the headline
                roofline['frac_vs_box'] = round(roofline['achieved'] / calibration['mfma_tflops'], 4)
                if roofline['wgrad_kernel'].get('achieved'):
                    roofline['wgrad_kernel']['frac_vs_box'] = round(roofline['wgrad_kernel']['achieved'] / calibration['mfma_tflops'], 4)
        if ddp:
            red = optimizer._reducer
            sizes = [b.flat.numel() * b.flat.element_size() for b in red.buckets]
            result['config']['allreduce'] = {
                'buckets': len(red.buckets), 'bytes_per_step': red.total_bytes,
                'largest_bucket_bytes': max(sizes), 'bucket_bytes': sizes,
                # a bucket larger than chunk_bytes (one huge parameter: pack5's 302 MB weight) is reduced in slices of <= chunk_bytes
                'chunk_bytes': red.chunk_bytes,
                'collectives_per_step': sum(max(1, -(-sz // red.chunk_bytes)) for sz in sizes),
                'in_place_on_optimizer_arena': args.optimizer == 'flat',
                'overlap_with_backward': bool(red.overlap),
                # time the compute stream spent waiting for the communication stream at the end-of-backward join
                'exposed_ms_per_step': m['exposed_allreduce_ms_per_step'],
                # host time of one collective call (ProcessGroup enqueue + the communication stream's stream-ordered wait for it)
                'host_us_per_collective': m['host_us_per_collective']}
        if world > ndev:
            result['config']['note'] = ('%d ranks share %d device(s): functional rehearsal of the N>1 path over gloo, not a '
                                        'scaling measurement' % (world, ndev))
        if extra is not None:
            ev, ems, eroof = line_of(extra, 384, 1280, 2, 6, 1)
            result['extra'] = {'metric': 'images/sec PackNet01 self-sup train 384x1280', 'value': round(ev, 3), 'unit': 'images/sec',
                               'ms_per_step': round(ems, 3), 'steps': 6, 'warmup': 1, 'n_gpus': 1,
                               'config': {'workload': 'same model and optimizer, KITTI-shaped 384x1280 triplets, batch 2/GPU '
                                                      '(BASELINE.json configs[2] shape)', 'global_batch': 2,
                                          'final_loss': round(extra['loss'], 6)},
                               'roofline': eroof}
        # the GPU eager baseline runs in a child process, started BEFORE the CPU baseline so that MIOpen's first-use compilation
        # (~260 s on a fresh box, host-side) overlaps the ~75 s CPU leg; `auto` waits up to 420 s from its start (VERDICT r04 item 8:
        # the driver gives this script 1 800 s and the round-4 run used 96 s of them), `on` waits without a limit
        child = None
        if world == 1 and not ddp and args.gpu_baseline != 'off':
            torch.cuda.empty_cache()
            child = gpu_eager_baseline_start(H, W, B)
        if world == 1 and not args.no_cpu_baseline:
            if pinned is not None and all_cpus:
                os.sched_setaffinity(0, all_cpus)          # the CPU leg sweeps thread counts over the whole box
            result['cpu_baseline'] = cpu_baseline(H, W)
        if child is not None:
            result['gpu_eager_baseline'] = gpu_eager_baseline_finish(child, H, W, B, value, None if args.gpu_baseline == 'on' else 420.0)
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
