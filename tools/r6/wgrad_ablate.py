"""What-if timing of the split-bf16 weight-gradient kernels: one build of conv2d_wgrad3.hip / conv2d_wgrad4.hip per compile-time mask
(-DPNSFM_WG_ABLATE=<mask>; results are wrong by construction) -- per layer, under the shipped tuning database's configuration, the
launch time with parts of the kernel switched off:
  1 no dY split (raw bits as pieces), 2 no neighbour LDS reads, 4 no shifted operands (v_alignbit), 8 patch staged for the first tile
  only, 16 no MFMAs, 32 dY loaded once (combinations by OR).  Mask 0 is checked against the production library.
usage: python tools/r6/wgrad_ablate.py [--build]      (--build: compile the libraries, e.g. in the build container, and exit)"""
import ctypes, os, shutil, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
CS = os.path.join(ROOT, 'packnet-sfm_amd', 'csrc')
OUT = os.path.join(ROOT, 'tools', 'micro', 'wgabl')
MASKS = [0, 1, 2 | 4, 8, 32, 1 | 32, 1 | 2 | 4, 1 | 2 | 4 | 8 | 32, 16, 16 | 1 | 2 | 4 | 8 | 32]
HIPCC = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics', '-w']


def lib_of(m):
    return os.path.join(OUT, 'm%d' % m, 'libpnsfm_wgabl.so')


if '--build' in sys.argv or not all(os.path.exists(lib_of(m)) for m in MASKS):
    os.makedirs(OUT, exist_ok=True)
    common = []
    procs = []
    for f in ('api', 'conv2d', 'conv2d_wgrad2'):
        o = os.path.join(OUT, f + '.o')
        common.append(o)
        procs.append(subprocess.Popen(HIPCC + ['-c', os.path.join(CS, f + '.hip'), '-o', o]))
    for m in MASKS:
        os.makedirs(os.path.dirname(lib_of(m)), exist_ok=True)
        for f in ('conv2d_wgrad3', 'conv2d_wgrad4'):
            procs.append(subprocess.Popen(HIPCC + ['-DPNSFM_WG_ABLATE=%d' % m, '-c', os.path.join(CS, f + '.hip'), '-o',
                                                   os.path.join(OUT, 'm%d' % m, f + '.o')]))
        if len(procs) >= 7:
            for p in procs:
                assert p.wait() == 0
            procs = []
    for p in procs:
        assert p.wait() == 0
    for m in MASKS:
        d = os.path.dirname(lib_of(m))
        subprocess.check_call(HIPCC + ['-shared', '-o', lib_of(m)] + common + [os.path.join(d, 'conv2d_wgrad3.o'), os.path.join(d, 'conv2d_wgrad4.o')])
        shutil.copy(os.path.join(CS, 'tuned_gfx950.db'), os.path.join(d, 'tuned_gfx950.db'))
    if '--build' in sys.argv:
        sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
vp = ctypes.c_void_p
libs = [('prod', ctypes.CDLL(os.path.join(CS, 'libpnsfm_hip.so')))] + [('[%d]' % m, ctypes.CDLL(lib_of(m))) for m in MASKS]
SHAPES = [(4, 64, 64, 192, 640, 7), (4, 256, 64, 96, 320, 7), (4, 129, 64, 192, 640, 3), (4, 64, 64, 96, 320, 3), (4, 128, 128, 48, 160, 3),
          (4, 256, 256, 24, 80, 3), (4, 512, 512, 12, 40, 3), (4, 16384, 512, 6, 20, 3), (4, 8192, 256, 12, 40, 3), (8, 2048, 64, 4, 320, 5),
          (4, 256, 64, 48, 160, 5), (4, 512, 128, 24, 80, 5)]
for (B, Cin, Cout, H, W, ks) in SHAPES:
    x = torch.randn(B, Cin, H, W, device='cuda'); dy = torch.randn(B, Cout, H, W, device='cuda')
    dw = torch.empty(Cout, Cin, ks, ks, device='cuda'); db = torch.empty(Cout, device='cuda')
    fl = 2.0 * B * Cin * Cout * ks * ks * H * W
    out = []
    for name, lib in libs:
        run = lambda: lib.pnsfm_conv2d_backward_weight(vp(x.data_ptr()), vp(dy.data_ptr()), vp(dw.data_ptr()), vp(db.data_ptr()), B, Cin, Cout, H, W, ks, vp(0))
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        out.append((name, best))
    print('%-28s prod %.1f us (%.0f TF) | ' % ((B, Cin, Cout, H, W, ks), out[0][1] * 1e3, fl / out[0][1] / 1e9) +
          '  '.join('%s %.0f' % (n, t * 1e3) for n, t in out[1:]), flush=True)
