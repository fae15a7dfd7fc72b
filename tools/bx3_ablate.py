"""What-if timing of the split-bf16 conv kernel (debug build, -DPNSFM_BX3_ABLATE; results are wrong by construction): per layer and
configuration, the launch time with individual parts of the pipeline switched off --
  1 no weight DMA in the loop, 2 no patch loads / split / ds_write at chunk ends, 4 no stage barriers, 8 fragments read once per
  stage, 16 no MFMAs, 32 no epilogue stores (combinations by OR).
usage: python tools/bx3_ablate.py  [B Cin Cout H W ks NT variant narrowM tilemode split ...]   (default: the 3x3 body layers under
the shipped tuning database's configurations)"""
import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
CS = os.path.join(ROOT, 'packnet-sfm_amd', 'csrc')
LIB = os.path.join(ROOT, 'tools', 'micro', 'libpnsfm_bx3ablate.so')
if not os.path.exists(LIB) or '--build' in sys.argv:
    srcs = [os.path.join(CS, f) for f in ('api.hip', 'conv2d.hip', 'conv2d_wgrad2.hip', 'conv2d_wgrad3.hip', 'conv2d_wgrad4.hip')]
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-munsafe-fp-atomics',
                           '-DPNSFM_BX3_ABLATE', '-w', '-o', LIB] + srcs)
    if '--build' in sys.argv:
        sys.exit(0)
lib = ctypes.CDLL(LIB)
vp = ctypes.c_void_p
lib.pnsfm_conv2d_packed_elems_fwd.restype = ctypes.c_size_t
args = [int(v) for v in sys.argv[1:] if not v.startswith('--')]
if not args:
    # (NT, variant, narrow-M, tile mode, split) of the shipped database for these shapes are looked up below (-1 = library decision)
    args = []
    for sh in ((4, 64, 64, 96, 320, 3), (4, 128, 128, 48, 160, 3), (4, 256, 256, 24, 80, 3), (4, 512, 512, 12, 40, 3), (4, 64, 64, 192, 640, 7),
               (4, 64, 129, 192, 640, 3)):
        args += list(sh) + [-1, 0, 0, 0, 0]
MODES = [0, 1, 2, 3, 4, 8, 16, 32, 1 | 2 | 4, 1 | 2 | 4 | 8, 1 | 2 | 4 | 8 | 32, 16 | 32]
PAD = 0
if '--one-wg' in sys.argv:
    # one workgroup per CU (LDS padded to > half a CU's): what ONE wave per SIMD sustains with and without the producer work --
    # the ceiling of a design that gives the loads / split / DMA to separate producer waves
    MODES = [0, 1 | 2 | 4, 1 | 2 | 4 | 32, 1 | 2 | 4 | 8 | 32]
    PAD = 40 * 1024
for i in range(0, len(args), 11):
    B, Cin, Cout, H, W, ks, NT, variant, narrow, tm, split = args[i:i + 11]
    n = lib.pnsfm_conv2d_packed_elems_fwd(Cin, Cout, ks)
    x = torch.randn(B, Cin, H, W, device='cuda'); w = torch.randn(Cout, Cin, ks, ks, device='cuda') * 0.05
    wp = torch.zeros(n, device='cuda'); y = torch.empty(B, Cout, H, W, device='cuda')
    lib.pnsfm_conv2d_pack_weights(vp(w.data_ptr()), vp(wp.data_ptr()), vp(0), Cin, Cout, ks, vp(0))
    if NT >= 0:
        key = (ctypes.c_int * 7)(110, B, Cin, Cout, H, W, ks)
        lib.pnsfm_tune_set(key, NT | (variant << 4) | (narrow << 8) | (tm << 9), split)
    fwd = lambda: lib.pnsfm_conv2d_forward(vp(x.data_ptr()), vp(wp.data_ptr()), vp(0), vp(y.data_ptr()), B, Cin, Cout, H, W, ks, vp(0))
    fl = 2.0 * B * Cin * Cout * ks * ks * H * W
    out = []
    lib.pnsfm_debug_set_smem_pad(PAD)
    for mode in MODES:
        lib.pnsfm_debug_set_ablate(mode)
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fwd()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        out.append((mode, best))
    lib.pnsfm_debug_set_ablate(0)
    base = out[0][1]
    print('%s cfg %s: full %.1f us (%.0f TF) | ' % ((B, Cin, Cout, H, W, ks), (NT, variant, narrow, tm, split), base * 1e3, fl / base / 1e9) +
          '  '.join('[%d] %.1f' % (m, t * 1e3) for m, t in out[1:]), flush=True)
