"""Per-wave cycle breakdown of the ping-pong conv kernel (debug build, -DPNSFM_PIPE_TRACE; csrc/conv2d_bx3pp.h): compute half-steps
(and the part of them in front of the first MFMA batch), staging half-steps, time parked at the barrier behind each kind.
usage: python tools/pp_trace.py B Cin Cout H W ks NT narrowM tilemode split  [more configurations ...]"""
import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
CS = os.path.join(ROOT, 'packnet-sfm_amd', 'csrc')
LIB = os.path.join(ROOT, 'gpurun_out', 'libpnsfm_pptrace.so')
srcs = [os.path.join(CS, f) for f in ('api.hip', 'conv2d.hip', 'conv2d_wgrad2.hip', 'conv2d_wgrad3.hip', 'conv2d_wgrad4.hip')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-munsafe-fp-atomics',
                       '-DPNSFM_PIPE_TRACE', '-w', '-o', LIB] + srcs)
lib = ctypes.CDLL(LIB)
vp = ctypes.c_void_p
lib.pnsfm_conv2d_packed_elems_fwd.restype = ctypes.c_size_t
args = [int(v) for v in sys.argv[1:]]
for i in range(0, len(args), 10):
    B, Cin, Cout, H, W, ks, NT, narrow, tm, split = args[i:i + 10]
    n = lib.pnsfm_conv2d_packed_elems_fwd(Cin, Cout, ks)
    x = torch.randn(B, Cin, H, W, device='cuda'); w = torch.randn(Cout, Cin, ks, ks, device='cuda') * 0.05
    wp = torch.zeros(n, device='cuda'); y = torch.empty(B, Cout, H, W, device='cuda')
    lib.pnsfm_conv2d_pack_weights(vp(w.data_ptr()), vp(wp.data_ptr()), vp(0), Cin, Cout, ks, vp(0))
    key = (ctypes.c_int * 7)(110, B, Cin, Cout, H, W, ks)
    lib.pnsfm_tune_set(key, NT | (7 << 4) | (narrow << 8) | (tm << 9), split)
    trace = torch.zeros(8 * (1 << 18), dtype=torch.int64, device='cuda')
    lib.pnsfm_debug_set_trace(vp(trace.data_ptr()))
    fwd = lambda: lib.pnsfm_conv2d_forward(vp(x.data_ptr()), vp(wp.data_ptr()), vp(0), vp(y.data_ptr()), B, Cin, Cout, H, W, ks, vp(0))
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fwd()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    t = trace.cpu().view(-1, 8)
    t = t[t[:, 4] > 0]
    fl = 2.0 * B * Cin * Cout * ks * ks * H * W
    print('%s NT%d narrow%d tm%d split%d: %.1f us %.0f TF (traced build)' % ((B, Cin, Cout, H, W, ks), NT, narrow, tm, split, ms * 1e3, fl / ms / 1e9), flush=True)
    for g in (0, 1):
        tg = t[(t[:, 7] % 2) == g]
        if not len(tg):
            continue
        bws = (tg[:, 7] // 4096).double().mean()
        nh = ((tg[:, 7] % 4096) // 2).double().mean()          # stages = compute half-steps of the wave
        m = tg.double().mean(0)
        print('   group %d: %d waves, %d stages; per wave cycles: total %.0f = prologue %.0f + compute %.0f (head %.0f) + barrier-after-compute %.0f '
              '+ staging %.0f + barrier-after-staging %.0f + epilogue %.0f | per stage: compute %.0f (head %.0f) wait %.0f staging %.0f wait %.0f'
              % (g, len(tg), int(nh), m[4], m[5], m[0], m[1], m[3], m[2], bws, m[6], m[0] / nh, m[1] / nh, m[3] / nh, m[2] / nh, bws / nh), flush=True)
