"""Per-wave cycle breakdown of the pipelined conv kernel (debug build with -DPNSFM_PIPE_TRACE): where does a wave spend its
time -- waiting at the stage barrier (DMA drain + meeting the other waves), in the stage body (MFMAs + DMA issue), prologue,
epilogue?  usage: python tools/pipe_trace.py B Cin Cout H W ks [NT fMT split]"""
import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'packnet-sfm_amd'))
import torch
CS = os.path.join(ROOT, 'packnet-sfm_amd', 'csrc')
LIB = os.path.join(ROOT, 'gpurun_out', 'libpnsfm_trace.so')
if not os.path.exists(LIB):
    srcs = [os.path.join(CS, f) for f in ('api.hip', 'conv2d.hip', 'conv2d_wgrad2.hip')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-munsafe-fp-atomics',
                           '-DPNSFM_PIPE_TRACE', '-w', '-o', LIB] + srcs)
lib = ctypes.CDLL(LIB)
B, Cin, Cout, H, W, ks = (int(v) for v in sys.argv[1:7])
NT, fMT, split = (int(v) for v in sys.argv[7:10]) if len(sys.argv) >= 10 else (1, 0, 1)
lib.pnsfm_conv2d_packed_elems_fwd.restype = ctypes.c_size_t
n = lib.pnsfm_conv2d_packed_elems_fwd(Cin, Cout, ks)
x = torch.randn(B, Cin, H, W, device='cuda'); w = torch.randn(Cout, Cin, ks, ks, device='cuda') * 0.05
wp = torch.zeros(n, device='cuda'); y = torch.empty(B, Cout, H, W, device='cuda')
vp = ctypes.c_void_p
lib.pnsfm_conv2d_pack_weights(vp(w.data_ptr()), vp(wp.data_ptr()), vp(0), Cin, Cout, ks, vp(0))
key = (ctypes.c_int * 7)(10, B, Cin, Cout, H, W, ks)
lib.pnsfm_tune_set(key, NT | (2 << 4) | (fMT << 8), split)
trace = torch.zeros(6 * (1 << 20), dtype=torch.int64, device='cuda')
lib.pnsfm_debug_set_trace(vp(trace.data_ptr()))
lib.pnsfm_debug_set_trace_flags(int(os.environ.get('TRACE_FLAGS', '0')))
for _ in range(3):
    lib.pnsfm_conv2d_forward(vp(x.data_ptr()), vp(wp.data_ptr()), vp(0), vp(y.data_ptr()), B, Cin, Cout, H, W, ks, vp(0))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.pnsfm_conv2d_forward(vp(x.data_ptr()), vp(wp.data_ptr()), vp(0), vp(y.data_ptr()), B, Cin, Cout, H, W, ks, vp(0))
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
t = trace.cpu().view(-1, 6)
t = t[t[:, 4] > 0].double()
fl = 2.0 * B * Cin * Cout * ks * ks * H * W
print('shape', (B, Cin, Cout, H, W, ks), 'NT', NT, 'fMT', fMT, 'split', split, '%.3f ms %.1f TFLOP/s' % (ms, fl / ms / 1e9), 'waves traced', len(t))
m = t.mean(0)
print('per wave (cycles, mean): total %.0f = prologue %.0f + barrier-wait %.0f + stage bodies %.0f + epilogue %.0f ; stages %d' % (m[4], m[2], m[0], m[1], m[3], int(m[5])))
MT = 1 if (fMT or Cout % 64 > 32 or Cout <= 32) else 2
print('per stage: barrier-wait %.0f, body %.0f cycles' % (m[0] / m[5], m[1] / m[5]))
