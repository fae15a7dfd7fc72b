"""Tensor-level wrappers over the C ABI of libpnsfm_hip.so (no autograd here; see functional.py).

Every wrapper validates dtype / contiguity / device, allocates outputs with torch (PyTorch is the device
allocator and stream owner -- plumbing, not compute) and enqueues the HIP kernels on torch's current stream.
"""
import ctypes
import threading

import torch

from . import _lib


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream       # (device index) -> hipStream_t as int: no Stream object per launch
except AttributeError:                                      # pragma: no cover
    _raw_stream = None


# a raw hipStream_t (int) and the device index it belongs to: every launch of THIS THREAD on THAT device goes there instead of torch's
# current stream (stream_override).  Per thread (a process-global override also redirected launches of another thread --
# a second device's autograd worker, a prefetch thread -- onto the wrong stream, even a stream of another device).
_TLS = threading.local()


class stream_override:
    """`with stream_override(raw_handle):` -- the ops inside enqueue on that HIP stream WITHOUT switching torch's current stream
    (torch.cuda.stream costs ~20 us per enter / exit).  Only for code that calls nothing but these ops: a torch operation inside would
    still go to torch's current stream.  Allocations stay in the current stream's pool: the caller orders their reuse (functional.py)."""

    def __init__(self, raw, device_index=None):
        self.ov = (raw, device_index)

    def __enter__(self):
        self.prev = getattr(_TLS, 'ov', None)
        _TLS.ov = self.ov
        return self

    def __exit__(self, *exc):
        _TLS.ov = self.prev
        return False


def current_raw_stream(device):
    """torch's current stream of `device` as an int handle."""
    if _raw_stream is not None:
        return int(_raw_stream(device.index))
    return int(torch.cuda.current_stream(device).cuda_stream)


def launch_stream_raw(device):
    """The raw stream (int) a launch for `device` goes to right now: this thread's override, else torch's current stream; 0 for the
    host-emulated build's CPU tensors."""
    if device.type != 'cuda':
        return 0
    ov = getattr(_TLS, 'ov', None)
    if ov is not None and (ov[1] is None or ov[1] == device.index):
        return ov[0]
    if _raw_stream is not None:
        return int(_raw_stream(device.index))
    return int(torch.cuda.current_stream(device).cuda_stream)


def stream_wait_stream(waiter_raw, signaler_raw):
    _lib.check(_lib.get().pnsfm_stream_wait_stream(ctypes.c_void_p(waiter_raw), ctypes.c_void_p(signaler_raw)), "stream_wait_stream")


def _stream(t):
    """torch's CURRENT stream of t's device as a hipStream_t.  ~800 launches per training step go through here: the raw-handle
    query costs ~0.3 us against ~5 us for torch.cuda.current_stream(...).cuda_stream (round 5 host profile)."""
    if not t.is_cuda:
        return ctypes.c_void_p(0)
    ov = getattr(_TLS, 'ov', None)
    if ov is not None and (ov[1] is None or ov[1] == t.device.index):
        return ctypes.c_void_p(ov[0])
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(t.device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _chk(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if _lib.REQUIRE_CUDA and not t.is_cuda:
            raise RuntimeError("packnet_sfm HIP op got a %s tensor: the HIP kernels run on MI355X only, "
                               "there is no CPU fallback" % t.device)
        if not t.is_contiguous():
            raise RuntimeError("packnet_sfm HIP op needs contiguous tensors")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("packnet_sfm HIP op got tensors on different devices")


def _f32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError("packnet_sfm HIP op needs float32 tensors, got %s" % t.dtype)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# ---------------------------------------------------------------------------------------------- conv2d
def tune_shipped_entries():
    return int(_lib.get().pnsfm_tune_shipped_entries())


def conv2d_packed_sizes(Cin, Cout, ks):
    lib = _lib.get()
    return (int(lib.pnsfm_conv2d_packed_elems_fwd(Cin, Cout, ks)), int(lib.pnsfm_conv2d_packed_elems_bwd(Cin, Cout, ks)))


def conv2d_pack(w, wp_fwd=None, wp_bwd=None, want_fwd=True, want_bwd=True):
    """w: [Cout, Cin, k, k] (reference layout) -> packed forward / backward-data weights."""
    _chk(w, wp_fwd, wp_bwd); _f32(w)
    Cout, Cin, ks, ks2 = w.shape
    assert ks == ks2
    nf, nb = conv2d_packed_sizes(Cin, Cout, ks)
    if want_fwd and wp_fwd is None:
        wp_fwd = torch.empty(nf, dtype=torch.float32, device=w.device)
    if want_bwd and wp_bwd is None:
        wp_bwd = torch.empty(nb, dtype=torch.float32, device=w.device)
    rc = _lib.get().pnsfm_conv2d_pack_weights(_ptr(w), _ptr(wp_fwd if want_fwd else None),
                                              _ptr(wp_bwd if want_bwd else None), Cin, Cout, ks, _stream(w))
    _lib.check(rc, "conv2d_pack_weights")
    return wp_fwd, wp_bwd


def conv2d_pack_table_build(entries, device):
    """entries: [(w, wp_fwd, wp_bwd)] with w [Cout, Cin, k, k] and packed buffers of conv2d_packed_sizes().  Returns
    (table tensor on `device`, number of items, total blocks, indices of the entries the table covers) -- entries whose shape
    does not take the split-bf16 layout in both directions are left out (pack them with conv2d_pack)."""
    import ctypes
    lib = _lib.get()
    isz = int(lib.pnsfm_conv2d_pack_item_bytes())
    host = (ctypes.c_ubyte * (isz * max(1, len(entries))))()
    n, blocks, covered = 0, 0, []
    for i, (w, pf, pb) in enumerate(entries):
        _chk(w, pf, pb); _f32(w, pf, pb)
        Cout, Cin, ks, _ = w.shape
        rc = lib.pnsfm_conv2d_pack_item_fill(ctypes.addressof(host) + n * isz, _ptr(w), _ptr(pf), _ptr(pb), Cin, Cout, ks, blocks)
        if rc < 0:
            _lib.check(rc, "conv2d_pack_item_fill")
        if rc > 0:
            n += 1
            blocks += rc
            covered.append(i)
    table = torch.frombuffer(bytearray(bytes(host)[:max(1, n) * isz]), dtype=torch.uint8).to(device)
    return table, n, blocks, covered


def conv2d_pack_table_run(table, n, blocks):
    if n:
        _lib.check(_lib.get().pnsfm_conv2d_pack_table(_ptr(table), n, blocks, _stream(table)), "conv2d_pack_table")


def conv2d_forward(x, wp_fwd, bias, Cout, ks):
    _chk(x, wp_fwd, bias); _f32(x, wp_fwd, bias)
    B, Cin, H, W = x.shape
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
    rc = _lib.get().pnsfm_conv2d_forward(_ptr(x), _ptr(wp_fwd), _ptr(bias), _ptr(y), B, Cin, Cout, H, W, ks, _stream(x))
    _lib.check(rc, "conv2d_forward")
    return y


def conv2d_backward_data(dy, wp_bwd, Cin, ks, addend=None):
    """dx [B,Cin,H,W]; addend (optional): the tensor's other gradient, added in the launch's epilogue -- [B,Cin,H,W], dense or a
    channel slice of a wider dense tensor (any sample stride)."""
    _chk(dy, wp_bwd); _f32(dy, wp_bwd)
    B, Cout, H, W = dy.shape
    dx = torch.empty((B, Cin, H, W), dtype=torch.float32, device=dy.device)
    if addend is None:
        rc = _lib.get().pnsfm_conv2d_backward_data(_ptr(dy), _ptr(wp_bwd), _ptr(dx), B, Cin, Cout, H, W, ks, _stream(dy))
        _lib.check(rc, "conv2d_backward_data")
        return dx
    _f32(addend)
    if addend.device != dy.device:
        raise RuntimeError("conv2d_backward_data: addend lives on %s, dy on %s" % (addend.device, dy.device))
    if tuple(addend.shape) != (B, Cin, H, W):
        raise RuntimeError("conv2d_backward_data: addend %s does not match dx %s" % (tuple(addend.shape), (B, Cin, H, W)))
    st = addend.stride()
    if not (st[3] == 1 and st[2] == W and st[1] == H * W and (B == 1 or st[0] >= Cin * H * W)):
        addend = addend.contiguous()
        st = addend.stride()
    rc = _lib.get().pnsfm_conv2d_backward_data_add(_ptr(dy), _ptr(wp_bwd), _ptr(dx), _ptr(addend), int(st[0]) if B > 1 else Cin * H * W, B,
                                                   Cin, Cout, H, W, ks, _stream(dy))
    _lib.check(rc, "conv2d_backward_data_add")
    return dx


def _alloc_dw_db(Cout, Cin, ks, want_bias, device):
    """dw and dbias in one allocation (dbias right behind dw): a split-K weight gradient zero-fills both with one memset."""
    n = Cout * Cin * ks * ks
    buf = torch.empty((n + (Cout if want_bias else 0),), dtype=torch.float32, device=device)
    return buf[:n].view(Cout, Cin, ks, ks), (buf[n:] if want_bias else None)


def conv2d_backward_weight(x, dy, ks, want_bias=True, dw_out=None, db_out=None):
    """dw_out / db_out: write the gradients into these (contiguous, right-shaped) tensors instead of fresh ones -- the flat
    gradient arena of rccl/flat_adam.py hands out views of its all-reduce buckets here, so the weight gradient lands
    where the collective and the optimizer read it, with no gather copy."""
    _chk(x, dy); _f32(x, dy)
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    if dw_out is not None:
        _chk(dw_out, db_out); _f32(dw_out, db_out)
        if tuple(dw_out.shape) != (Cout, Cin, ks, ks) or (want_bias and (db_out is None or db_out.numel() != Cout)):
            raise RuntimeError("conv2d_backward_weight: gradient slot has the wrong shape")
        # fresh view objects: AccumulateGrad adopts a gradient only if nobody else holds the tensor object
        dw, db = dw_out.view(dw_out.shape), (db_out.view(db_out.shape) if want_bias else None)
    else:
        dw, db = _alloc_dw_db(Cout, Cin, ks, want_bias, x.device)
    rc = _lib.get().pnsfm_conv2d_backward_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), B, Cin, Cout, H, W, ks, _stream(x))
    _lib.check(rc, "conv2d_backward_weight")
    return dw, db


def conv2d_forward_cat(xs, wp_fwd, bias, Cout, ks):
    """conv(cat(xs, 1)) without the concatenated tensor; xs: 2 or 3 contiguous NCHW tensors of equal B, H, W.  Raises HipError when
    the shape is outside the split kernels' envelope (the caller concatenates instead)."""
    _chk(*xs, wp_fwd, bias); _f32(*xs, wp_fwd, bias)
    B, _, H, W = xs[0].shape
    C = [t.shape[1] for t in xs] + [0] * (3 - len(xs))
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=xs[0].device)
    rc = _lib.get().pnsfm_conv2d_forward_cat(_ptr(xs[0]), C[0], _ptr(xs[1]), C[1], _ptr(xs[2] if len(xs) > 2 else None), C[2],
                                             _ptr(wp_fwd), _ptr(bias), _ptr(y), B, Cout, H, W, ks, _stream(xs[0]))
    _lib.check(rc, "conv2d_forward_cat")
    return y


def conv2d_cat_wgrad_supported(channels, Cout, H, W, ks, B=1):
    """Will conv2d_backward_weight_cat take sources with these channel counts (list of 2 or 3)?  Pure query."""
    C = list(channels) + [0] * (3 - len(channels))
    return bool(_lib.get().pnsfm_conv2d_cat_wgrad_supported(C[0], C[1], C[2], Cout, B, H, W, ks))


def conv2d_backward_weight_cat(xs, dy, ks, want_bias=True, dw_out=None, db_out=None):
    _chk(*xs, dy); _f32(*xs, dy)
    B, _, H, W = xs[0].shape
    C = [t.shape[1] for t in xs] + [0] * (3 - len(xs))
    Cin, Cout = sum(C), dy.shape[1]
    if dw_out is not None:
        _chk(dw_out, db_out); _f32(dw_out, db_out)
        if tuple(dw_out.shape) != (Cout, Cin, ks, ks) or (want_bias and (db_out is None or db_out.numel() != Cout)):
            raise RuntimeError("conv2d_backward_weight_cat: gradient slot has the wrong shape")
        dw, db = dw_out.view(dw_out.shape), (db_out.view(db_out.shape) if want_bias else None)
    else:
        dw, db = _alloc_dw_db(Cout, Cin, ks, want_bias, dy.device)
    rc = _lib.get().pnsfm_conv2d_backward_weight_cat(_ptr(xs[0]), C[0], _ptr(xs[1]), C[1], _ptr(xs[2] if len(xs) > 2 else None), C[2],
                                                     _ptr(dy), _ptr(dw), _ptr(db), B, Cout, H, W, ks, _stream(dy))
    _lib.check(rc, "conv2d_backward_weight_cat")
    return dw, db


def conv2d_forward_strided(x, wp_fwd, bias, Cout, ks, stride):
    _chk(x, wp_fwd, bias); _f32(x, wp_fwd, bias)
    B, Cin, H, W = x.shape
    P = ks // 2
    Ho, Wo = (H + 2 * P - ks) // stride + 1, (W + 2 * P - ks) // stride + 1
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    rc = _lib.get().pnsfm_conv2d_forward_strided(_ptr(x), _ptr(wp_fwd), _ptr(bias), _ptr(y), B, Cin, Cout, H, W, ks, stride,
                                                 _stream(x))
    _lib.check(rc, "conv2d_forward_strided")
    return y


def conv2d_backward_weight_strided(x, dy, ks, stride, want_bias=True):
    _chk(x, dy); _f32(x, dy)
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    dw, db = _alloc_dw_db(Cout, Cin, ks, want_bias, x.device)
    rc = _lib.get().pnsfm_conv2d_backward_weight_strided(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), B, Cin, Cout, H, W, ks, stride,
                                                         _stream(x))
    _lib.check(rc, "conv2d_backward_weight_strided")
    return dw, db


# ------------------------------------------------------------------------------------------- groupnorm
ACT_NONE, ACT_ELU, ACT_RELU = 0, 1, 2
GN_MAX_SPLIT = 64     # PNSFM_GN_MAX_SPLIT in include/pnsfm.h


_GN_WS = {}


def _gn_ws_doubles(B, C, G):
    k = (B, C, G)
    v = _GN_WS.get(k)
    if v is None:
        v = _GN_WS[k] = int(_lib.get().pnsfm_groupnorm_ws_doubles(B, C, G))
    return v


def groupnorm_act_forward(x, res, gamma, beta, G, eps, act):
    _chk(x, res, gamma, beta); _f32(x, res, gamma, beta)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    ms = torch.empty((2, B * G), dtype=torch.float32, device=x.device)       # mean | rstd in one allocation
    mean, rstd = ms[0], ms[1]
    # (no workspace tensor: the two-launch form keeps its partial sums in the stream's scratch buffer, the one-launch form has none)
    rc = _lib.get().pnsfm_groupnorm_act_forward(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd),
                                                None, B, C, HW, G, float(eps), act, _stream(x))
    _lib.check(rc, "groupnorm_act_forward")
    return y, mean, rstd


def groupnorm_act_backward(dy, x, res, gamma, beta, mean, rstd, G, act):
    _chk(dy, x, res, gamma, beta, mean, rstd); _f32(dy, x, res, gamma, beta, mean, rstd)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    dx = torch.empty_like(x)
    dgb = torch.empty((2, C), dtype=torch.float32, device=x.device)          # dgamma | dbeta in one allocation
    dgamma, dbeta = dgb[0].view_as(gamma), dgb[1].view_as(beta)
    rc = _lib.get().pnsfm_groupnorm_act_backward(_ptr(dy), _ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd),
                                                 _ptr(dx), _ptr(dgamma), _ptr(dbeta), None, B, C, HW, G, act, _stream(x))
    _lib.check(rc, "groupnorm_act_backward")
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------------ packing / conv3d
def _is_channel_slice(x):
    """[B,C,H,W] whose images are dense but B-strided (a channel slice of a wider contiguous tensor)."""
    B, C, H, W = x.shape
    st = x.stride()
    return st[3] == 1 and st[2] == W and st[1] == H * W and st[0] >= C * H * W


def space_to_depth(x):
    """x may be contiguous or a channel slice of a wider contiguous NCHW tensor (no copy in either case)."""
    B, C, H, W = x.shape
    if x.is_contiguous() or not _is_channel_slice(x):
        x = x.contiguous()
    _chk(y := torch.empty((B, 4 * C, H // 2, W // 2), dtype=torch.float32, device=x.device)); _f32(x)
    if _lib.REQUIRE_CUDA and not x.is_cuda:
        raise RuntimeError("packnet_sfm HIP op got a %s tensor: the HIP kernels run on MI355X only" % x.device)
    _lib.check(_lib.get().pnsfm_space_to_depth_strided(_ptr(x), _ptr(y), B, C, H, W, x.stride(0) if B > 1 else C * H * W,
                                                       _stream(x)), "space_to_depth")
    return y


def depth_to_space(x):
    _chk(x); _f32(x)
    B, C4, H, W = x.shape
    assert C4 % 4 == 0
    C = C4 // 4
    y = torch.empty((B, C, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.get().pnsfm_depth_to_space(_ptr(x), _ptr(y), B, C, H, W, _stream(x)), "depth_to_space")
    return y


def _nf_of(w3):
    """Number of 3-D feature maps of a Conv3d(1, NF, 3) weight [NF,1,3,3,3] (or [NF,27])."""
    nf = w3.shape[0]
    if w3.numel() != nf * 27:
        raise RuntimeError("conv3d: weight must be [NF,1,3,3,3]")
    return nf


def conv3d_forward(p, w3, b3):
    """p [B,D,H,W] -> [B,NF*D,H,W] (channel f*D+d), NF = w3.shape[0] in {4, 8}.  b3 None = no bias."""
    _chk(p, w3, b3); _f32(p, w3, b3)
    B, D, H, W = p.shape
    nf = _nf_of(w3)
    out = torch.empty((B, nf * D, H, W), dtype=torch.float32, device=p.device)
    _lib.check(_lib.get().pnsfm_conv3d_forward(_ptr(p), _ptr(w3), _ptr(b3), _ptr(out), B, D, H, W, nf, _stream(p)),
               "conv3d_forward")
    return out


def conv3d_backward_data(dout, w3):
    _chk(dout, w3); _f32(dout, w3)
    B, DF, H, W = dout.shape
    nf = _nf_of(w3)
    D = DF // nf
    dp = torch.empty((B, D, H, W), dtype=torch.float32, device=dout.device)
    _lib.check(_lib.get().pnsfm_conv3d_backward_data(_ptr(dout), _ptr(w3), _ptr(dp), B, D, H, W, nf, _stream(dout)),
               "conv3d_backward_data")
    return dp


def conv3d_backward_weight(p, dout):
    _chk(p, dout); _f32(p, dout)
    B, D, H, W = p.shape
    nf = dout.shape[1] // D
    dw3 = torch.empty((nf, 1, 3, 3, 3), dtype=torch.float32, device=p.device)
    db3 = torch.empty((nf,), dtype=torch.float32, device=p.device)
    ws = torch.empty((8 * 28,), dtype=torch.float64, device=p.device)
    _lib.check(_lib.get().pnsfm_conv3d_backward_weight(_ptr(p), _ptr(dout), _ptr(dw3), _ptr(db3), _ptr(ws), B, D, H, W, nf,
                                                       _stream(p)), "conv3d_backward_weight")
    return dw3, db3


# ------------------------------------------------------------------- composed packing convolution: bias
def pack_bias_eff_forward(W2, b2, b3):
    """(bias_eff [C], Ssum [C,d]) of the composed packing convolution: include/pnsfm.h, pnsfm_pack_bias_eff_forward."""
    _chk(W2, b2, b3); _f32(W2, b2, b3)
    C, d = W2.shape[0], b3.numel()
    blk = W2[0].numel() // d
    Ssum = torch.empty((C, d), dtype=torch.float32, device=W2.device)
    bias = torch.empty((C,), dtype=torch.float32, device=W2.device)
    _lib.check(_lib.get().pnsfm_pack_bias_eff_forward(_ptr(W2), _ptr(b2), _ptr(b3), _ptr(Ssum), _ptr(bias), C, d, blk, _stream(W2)),
               "pack_bias_eff_forward")
    return bias, Ssum


def pack_bias_eff_backward(g, Ssum, b3, dWeff_full, k, want_db3=True, want_dW2=True):
    """g [C] = d(bias_eff); dWeff_full [C, d*D, k+2, k+2] = the composition's gradient on the padded taps.
    Returns (db3 [d] or None, dW2 [C, d*D, k, k] or None)."""
    _chk(g, Ssum, b3, dWeff_full); _f32(g, Ssum, b3, dWeff_full)
    C, d = Ssum.shape
    D = dWeff_full.shape[1] // d
    assert tuple(dWeff_full.shape) == (C, d * D, k + 2, k + 2) and g.numel() == C
    db3 = torch.empty((d,), dtype=torch.float32, device=g.device) if want_db3 else None
    dW2 = torch.empty((C, d * D, k, k), dtype=torch.float32, device=g.device) if want_dW2 else None
    _lib.check(_lib.get().pnsfm_pack_bias_eff_backward(_ptr(g), _ptr(Ssum), _ptr(b3), _ptr(dWeff_full), _ptr(db3), _ptr(dW2), C, d, D,
                                                       k, _stream(g)), "pack_bias_eff_backward")
    return db3, dW2


# --------------------------------------------------------------------------------- nearest up-sampling
def upsample_nearest_ok(x, s):
    return x.dim() == 4 and x.dtype == torch.float32 and s >= 1 and (x.shape[3] * s) % 4 == 0 and x.numel() * s * s < 2 ** 32


def upsample_nearest_forward(x, s):
    _chk(x); _f32(x)
    B, C, h, w = x.shape
    y = torch.empty((B, C, h * s, w * s), dtype=torch.float32, device=x.device)
    _lib.check(_lib.get().pnsfm_upsample_nearest_forward(_ptr(x), _ptr(y), B * C, h, w, s, _stream(x)), "upsample_nearest_forward")
    return y


def upsample_nearest_backward(dy, s):
    _chk(dy); _f32(dy)
    B, C, Ho, Wo = dy.shape
    h, w = Ho // s, Wo // s
    dx = torch.empty((B, C, h, w), dtype=torch.float32, device=dy.device)
    _lib.check(_lib.get().pnsfm_upsample_nearest_backward(_ptr(dy), _ptr(dx), B * C, h, w, s, _stream(dy)), "upsample_nearest_backward")
    return dx


# ------------------------------------------------------------------------------- scalar tail of the loss
def loss_combine_forward(photometric, smoothness, weight):
    """photometric / smoothness: lists of 0-dim device tensors -> out3 = (loss, weighted smoothness, photometric)."""
    n, ns = len(photometric), len(smoothness)
    ts = list(photometric) + list(smoothness)
    _chk(*ts); _f32(*ts)
    P = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in photometric])
    S = (ctypes.c_void_p * 8)(*[t.data_ptr() for t in smoothness])
    out = torch.empty((3,), dtype=torch.float32, device=ts[0].device)
    _lib.check(_lib.get().pnsfm_loss_combine_forward(ctypes.byref(P), n, ctypes.byref(S), ns, float(weight), _ptr(out), _stream(out)), "loss_combine_forward")
    return out


def loss_combine_backward(g, n, ns, weight):
    """g: 0-dim device tensor -> [16]: entries i < n = d/dP[i], 8 + i (i < ns) = d/dS[i]."""
    _chk(g); _f32(g)
    dout = torch.empty((16,), dtype=torch.float32, device=g.device)
    _lib.check(_lib.get().pnsfm_loss_combine_backward(_ptr(g), n, ns, float(weight), _ptr(dout), _stream(g)), "loss_combine_backward")
    return dout


# ------------------------------------------------------------------------------------------ invdepth
def invdepth_act_forward(x, min_depth):
    _chk(x); _f32(x)
    y = torch.empty_like(x)
    _lib.check(_lib.get().pnsfm_invdepth_act_forward(_ptr(x), _ptr(y), x.numel(), float(min_depth), _stream(x)),
               "invdepth_act_forward")
    return y


def invdepth_act_backward(dy, y, min_depth):
    _chk(dy, y); _f32(dy, y)
    dx = torch.empty_like(y)
    _lib.check(_lib.get().pnsfm_invdepth_act_backward(_ptr(dy), _ptr(y), _ptr(dx), y.numel(), float(min_depth), _stream(y)),
               "invdepth_act_backward")
    return dx


def invdepth_conv_forward(x, w, bias, min_depth):
    """x [B,C,H,W], w [1,C,3,3], bias [1] -> sigmoid(conv3x3(x) + b) / min_depth, [B,1,H,W]."""
    _chk(x, w, bias); _f32(x, w, bias)
    B, C, H, W = x.shape
    if tuple(w.shape) != (1, C, 3, 3) or bias is None or bias.numel() != 1:
        raise RuntimeError("invdepth_conv: weight must be [1,%d,3,3] with a bias of one element" % C)
    y = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.get().pnsfm_invdepth_conv_forward(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), B, C, H, W, float(min_depth),
                                                      _stream(x)), "invdepth_conv_forward")
    return y


def invdepth_conv_backward(x, w, dz):
    """dz [B,1,H,W] = gradient at the conv output -> (dx [B,C,H,W], dw [1,C,3,3], db [1])."""
    _chk(x, w, dz); _f32(x, w, dz)
    B, C, H, W = x.shape
    dx = torch.empty_like(x)
    buf = torch.empty((C * 9 + 1,), dtype=torch.float32, device=x.device)      # db right behind dw: one fill
    dw, db = buf[:C * 9].view(1, C, 3, 3), buf[C * 9:]
    _lib.check(_lib.get().pnsfm_invdepth_conv_backward(_ptr(x), _ptr(w), _ptr(dz), _ptr(dx), _ptr(dw), _ptr(db), B, C, H, W,
                                                       _stream(x)), "invdepth_conv_backward")
    return dx, dw, db


def pose_vec2mat_forward(vec):
    _chk(vec); _f32(vec)
    N = vec.shape[0]
    mat = torch.empty((N, 4, 4), dtype=torch.float32, device=vec.device)
    _lib.check(_lib.get().pnsfm_pose_vec2mat_forward(_ptr(vec), _ptr(mat), N, _stream(vec)), "pose_vec2mat_forward")
    return mat


def pose_vec2mat_backward(vec, dmat):
    _chk(vec, dmat); _f32(vec, dmat)
    N = vec.shape[0]
    dvec = torch.empty((N, 6), dtype=torch.float32, device=vec.device)
    _lib.check(_lib.get().pnsfm_pose_vec2mat_backward(_ptr(vec), _ptr(dmat), _ptr(dvec), N, _stream(vec)),
               "pose_vec2mat_backward")
    return dvec


# ---------------------------------------------------------------------------------------------- loss
PADDING_MODES = {'zeros': 0, 'border': 1, 'reflection': 2}     # F.grid_sample's padding_mode (align_corners=True)


def view_synthesis_forward(inv_depth, ref, K, refK, T, padding_mode=0):
    """inv_depth [B,1,H,W]; ref [J,B,3,H,W]; K, refK [B,3,3]; T [J,B,4,4] -> warped [J,B,3,H,W]."""
    _chk(inv_depth, ref, K, refK, T); _f32(inv_depth, ref, K, refK, T)
    J, B, _, H, W = ref.shape
    warped = torch.empty_like(ref)
    _lib.check(_lib.get().pnsfm_view_synthesis_forward_pad(_ptr(inv_depth), _ptr(ref), _ptr(K), _ptr(refK), _ptr(T), _ptr(warped),
                                                           J, B, H, W, int(padding_mode), _stream(ref)), "view_synthesis_forward")
    return warped


def view_synthesis_backward(d_warped, inv_depth, ref, K, refK, T, padding_mode=0):
    _chk(d_warped, inv_depth, ref, K, refK, T); _f32(d_warped, inv_depth, ref, K, refK, T)
    J, B, _, H, W = ref.shape
    d_inv = torch.empty_like(inv_depth)
    dT = torch.empty_like(T)
    ws = torch.empty((J * B * 12,), dtype=torch.float64, device=ref.device)
    _lib.check(_lib.get().pnsfm_view_synthesis_backward_pad(_ptr(d_warped), _ptr(inv_depth), _ptr(ref), _ptr(K), _ptr(refK),
                                                            _ptr(T), _ptr(d_inv), _ptr(dT), _ptr(ws), J, B, H, W,
                                                            int(padding_mode), _stream(ref)), "view_synthesis_backward")
    return d_inv, dT


def photometric_forward(warped, ref, target, ssim_weight, C1, C2, automask, reduce_op, clip_loss=0.0):
    """-> (loss_sum float64[1], argmin uint8[B,H,W]).  clip_loss > 0: candidates clamped at mean + clip_loss*std."""
    _chk(warped, ref, target); _f32(warped, ref, target)
    J, B, _, H, W = warped.shape
    loss_sum = torch.empty((1,), dtype=torch.float64, device=warped.device)
    argmin = torch.empty((B, H, W), dtype=torch.uint8, device=warped.device)
    if clip_loss > 0.0:
        stats = torch.empty((12,), dtype=torch.float64, device=warped.device)
        thr = torch.empty((6,), dtype=torch.float32, device=warped.device)
        _lib.check(_lib.get().pnsfm_photometric_forward_clip(
            _ptr(warped), _ptr(ref), _ptr(target), _ptr(loss_sum), _ptr(argmin), J, B, H, W, float(ssim_weight), float(C1),
            float(C2), int(automask), int(reduce_op), float(clip_loss), _ptr(stats), _ptr(thr), _stream(warped)),
            "photometric_forward_clip")
        return loss_sum, argmin
    _lib.check(_lib.get().pnsfm_photometric_forward(_ptr(warped), _ptr(ref), _ptr(target), _ptr(loss_sum), _ptr(argmin), J, B, H,
                                                    W, float(ssim_weight), float(C1), float(C2), int(automask), int(reduce_op),
                                                    _stream(warped)), "photometric_forward")
    return loss_sum, argmin


def photometric_backward(warped, target, argmin, grad_scale, ssim_weight, C1, C2, automask, reduce_op, clip=False):
    _chk(warped, target, argmin); _f32(warped, target)
    J, B, _, H, W = warped.shape
    d_warped = torch.empty_like(warped)
    fn = _lib.get().pnsfm_photometric_backward_clip if clip else _lib.get().pnsfm_photometric_backward
    _lib.check(fn(_ptr(warped), _ptr(target), _ptr(argmin), _ptr(d_warped), float(grad_scale),
                                                     J, B, H, W, float(ssim_weight), float(C1), float(C2), int(automask),
                                                     int(reduce_op), _stream(warped)), "photometric_backward")
    return d_warped


def photometric_forward_mean(warped, ref, target, ssim_weight, C1, C2, automask, reduce_op):
    """-> (loss float32[1] = pixel mean of the reduced photometric map, argmin uint8[B,H,W]); no clipping."""
    _chk(warped, ref, target); _f32(warped, ref, target)
    J, B, _, H, W = warped.shape
    loss = torch.empty((1,), dtype=torch.float32, device=warped.device)
    argmin = torch.empty((B, H, W), dtype=torch.uint8, device=warped.device)
    _lib.check(_lib.get().pnsfm_photometric_forward_mean(_ptr(warped), _ptr(ref), _ptr(target), _ptr(loss), _ptr(argmin), J, B, H, W,
                                                         float(ssim_weight), float(C1), float(C2), int(automask), int(reduce_op),
                                                         _stream(warped)), "photometric_forward_mean")
    return loss, argmin


def photometric_backward_dev(warped, target, argmin, grad_scale, upstream, ssim_weight, C1, C2, automask, reduce_op, clip=False):
    """d_warped = grad_scale * upstream[0] * d(loss_sum)/d(warped); `upstream`: float32 device scalar (or None)."""
    _chk(warped, target, argmin); _f32(warped, target)
    if upstream is not None:
        _chk(upstream); _f32(upstream)
    J, B, _, H, W = warped.shape
    d_warped = torch.empty_like(warped)
    _lib.check(_lib.get().pnsfm_photometric_backward_dev(_ptr(warped), _ptr(target), _ptr(argmin), _ptr(d_warped), float(grad_scale),
                                                         _ptr(upstream), J, B, H, W, float(ssim_weight), float(C1), float(C2),
                                                         int(automask), int(reduce_op), int(bool(clip)), _stream(warped)),
               "photometric_backward_dev")
    return d_warped


def photometric_l1_forward(warped, ref, target, automask, reduce_op, clip_loss):
    """L1-only photometric loss on per-channel candidate maps -> (loss float32[1], rec int32[B,H,W])."""
    _chk(warped, ref, target); _f32(warped, ref, target)
    J, B, _, H, W = warped.shape
    loss = torch.empty((1,), dtype=torch.float32, device=warped.device)
    rec = torch.empty((B, H, W), dtype=torch.int32, device=warped.device)
    _lib.check(_lib.get().pnsfm_photometric_l1_forward(_ptr(warped), _ptr(ref), _ptr(target), _ptr(loss), _ptr(rec), J, B, H, W,
                                                       int(automask), int(reduce_op), float(clip_loss), _stream(warped)),
               "photometric_l1_forward")
    return loss, rec


def photometric_l1_backward(warped, target, rec, grad_scale, upstream, automask, reduce_op):
    _chk(warped, target, rec); _f32(warped, target)
    if upstream is not None:
        _chk(upstream); _f32(upstream)
    J, B, _, H, W = warped.shape
    d_warped = torch.empty_like(warped)
    _lib.check(_lib.get().pnsfm_photometric_l1_backward(_ptr(warped), _ptr(target), _ptr(rec), _ptr(d_warped), float(grad_scale),
                                                        _ptr(upstream), J, B, H, W, int(automask), int(reduce_op), _stream(warped)),
               "photometric_l1_backward")
    return d_warped


def smoothness_norm_forward(inv_depth, image):
    """-> (loss float32[1], mean float32[B]): smoothness of inv_depth / clamp(mean_hw(inv_depth), 1e-6)."""
    _chk(inv_depth, image); _f32(inv_depth, image)
    B, _, H, W = image.shape
    loss = torch.empty((1,), dtype=torch.float32, device=image.device)
    mean = torch.empty((B,), dtype=torch.float32, device=image.device)
    _lib.check(_lib.get().pnsfm_smoothness_norm_forward(_ptr(inv_depth), _ptr(image), _ptr(loss), _ptr(mean), B, H, W, _stream(image)),
               "smoothness_norm_forward")
    return loss, mean


def smoothness_norm_backward(inv_depth, image, mean, upstream):
    _chk(inv_depth, image, mean); _f32(inv_depth, image, mean)
    if upstream is not None:
        _chk(upstream); _f32(upstream)
    B, _, H, W = image.shape
    d = torch.empty_like(inv_depth)
    _lib.check(_lib.get().pnsfm_smoothness_norm_backward(_ptr(inv_depth), _ptr(image), _ptr(mean), _ptr(upstream), _ptr(d), B, H, W,
                                                         _stream(image)), "smoothness_norm_backward")
    return d


def smoothness_forward(inv_norm, image):
    _chk(inv_norm, image); _f32(inv_norm, image)
    B, _, H, W = image.shape
    sums = torch.empty((2,), dtype=torch.float64, device=image.device)
    _lib.check(_lib.get().pnsfm_smoothness_forward(_ptr(inv_norm), _ptr(image), _ptr(sums), B, H, W, _stream(image)),
               "smoothness_forward")
    return sums


def smoothness_backward(inv_norm, image, gx, gy):
    _chk(inv_norm, image); _f32(inv_norm, image)
    B, _, H, W = image.shape
    d = torch.empty_like(inv_norm)
    _lib.check(_lib.get().pnsfm_smoothness_backward(_ptr(inv_norm), _ptr(image), _ptr(d), float(gx), float(gy), B, H, W,
                                                    _stream(image)), "smoothness_backward")
    return d


# ---------------------------------------------------------------------------------- supervised loss
SUP_METHODS = {'l1': 0, 'mse': 1, 'abs_rel': 2, 'berhu': 3, 'silog': 4}


def supervised_loss_forward(pred, gt, method, sparse):
    """pred, gt: same-shape fp32 tensors -> (loss [1] fp32, ws for the backward call)."""
    _chk(pred, gt); _f32(pred, gt)
    if pred.shape != gt.shape:
        raise RuntimeError("supervised_loss: prediction %s and ground truth %s differ in shape" % (tuple(pred.shape), tuple(gt.shape)))
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    ws = torch.empty((8 + 1024,), dtype=torch.float64, device=pred.device)
    _lib.check(_lib.get().pnsfm_supervised_loss_forward(_ptr(pred), _ptr(gt), _ptr(loss), _ptr(ws), pred.numel(), int(method),
                                                        int(bool(sparse)), _stream(pred)), "supervised_loss_forward")
    return loss, ws


def supervised_loss_backward(pred, gt, ws, grad_out, method, sparse):
    _chk(pred, gt, ws, grad_out); _f32(pred, gt, grad_out)
    dpred = torch.empty_like(pred)
    _lib.check(_lib.get().pnsfm_supervised_loss_backward(_ptr(pred), _ptr(gt), _ptr(ws), _ptr(grad_out), _ptr(dpred),
                                                         pred.numel(), int(method), int(bool(sparse)), _stream(pred)),
               "supervised_loss_backward")
    return dpred


# ------------------------------------------------------------------------------------------------ NRS
def nrs_project_forward(direction, ray, temperature):
    """direction, ray: [3,h,w] -> (coords [h,w,2] = expected (row, col), stat [h,w,2] for backward)."""
    _chk(direction, ray); _f32(direction, ray)
    _, h, w = direction.shape
    coords = torch.empty((h, w, 2), dtype=torch.float32, device=ray.device)
    stat = torch.empty((h, w, 2), dtype=torch.float32, device=ray.device)
    _lib.check(_lib.get().pnsfm_nrs_project_forward(_ptr(direction), _ptr(ray), _ptr(coords), _ptr(stat), h, w, float(temperature),
                                                    _stream(ray)), "nrs_project_forward")
    return coords, stat


def nrs_project_backward(direction, ray, coords, stat, gcoords, temperature, want_dir=True, want_ray=True):
    _chk(direction, ray, coords, stat, gcoords); _f32(direction, ray, coords, stat, gcoords)
    _, h, w = direction.shape
    gdir = torch.empty_like(direction) if want_dir else None
    gray = torch.empty_like(ray) if want_ray else None
    _lib.check(_lib.get().pnsfm_nrs_project_backward(_ptr(direction), _ptr(ray), _ptr(coords), _ptr(stat), _ptr(gcoords), _ptr(gdir),
                                                     _ptr(gray), h, w, float(temperature), _stream(ray)), "nrs_project_backward")
    return gdir, gray


# ------------------------------------------------------------------------------------- input pipeline
def resample8(img, kk, bounds, out_size, axis):
    """One axis of PIL's 8-bit separable resample.  img: uint8 [N,H,W,C] (NHWC); kk int32 [out, ksize]; bounds int32 [out, 2]."""
    _chk(img, kk, bounds)
    if img.dtype != torch.uint8 or kk.dtype != torch.int32 or bounds.dtype != torch.int32:
        raise RuntimeError("resample8: uint8 image, int32 coefficient tables expected")
    N, H, W, C = img.shape
    oH, oW = (H, out_size) if axis == 1 else (out_size, W)
    out = torch.empty((N, oH, oW, C), dtype=torch.uint8, device=img.device)
    _lib.check(_lib.get().pnsfm_resample8(_ptr(img), _ptr(out), _ptr(kk), _ptr(bounds), kk.shape[1], N, H, W, oH, oW, C, axis,
                                          _stream(img)), "resample8")
    return out


# struct JitterOps of csrc/augment.hip: 4 op codes, 4 blend factors, hue_add, enabled, 3 'color' matrix diagonal entries, has_color
JITTER_RECORD = '4i4f2i3fi'
JITTER_RECORD_BYTES = 56


def jitter_record(order=(-1, -1, -1, -1), factors=(1.0, 1.0, 1.0, 1.0), hue_add=0, enabled=0, color=None):
    import struct
    c = tuple(float(v) for v in color) if color is not None else (1.0, 1.0, 1.0)
    return struct.pack(JITTER_RECORD, *order, *[float(f) for f in factors], int(hue_add), int(enabled), *c, 1 if color is not None else 0)


def jitter_totensor(img, ops_records, want_original=True):
    """img: uint8 [N,H,W,3]; ops_records: uint8 tensor holding N packed 56-byte JitterOps records (JITTER_RECORD) -> (jittered, original) float32
    [N,3,H,W] (original is None unless requested)."""
    _chk(img, ops_records)
    N, H, W, C = img.shape
    if img.dtype != torch.uint8 or C != 3 or ops_records.dtype != torch.uint8 or ops_records.numel() != JITTER_RECORD_BYTES * N:
        raise RuntimeError("jitter_totensor: uint8 [N,H,W,3] image and N %d-byte operation records expected" % JITTER_RECORD_BYTES)
    out = torch.empty((N, 3, H, W), dtype=torch.float32, device=img.device)
    orig = torch.empty((N, 3, H, W), dtype=torch.float32, device=img.device) if want_original else None
    ws = torch.empty((N,), dtype=torch.int64, device=img.device)
    _lib.check(_lib.get().pnsfm_jitter_totensor(_ptr(img), _ptr(ops_records), _ptr(ws), _ptr(out), _ptr(orig), N, H, W, _stream(img)),
               "jitter_totensor")
    return out, orig


# -------------------------------------------------------------------------------------- sparse tensors (PackNet-SAN)
def _i32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.int32:
            raise RuntimeError("packnet_sfm HIP op needs int32 index tensors, got %s" % t.dtype)


def sparse_compact(src, cap=None):
    """src: fp32 tensor over the cells of a [B, h, w] grid (a depth map or a 0/1 mask); active = src > 0.
    -> (imap int32 [ncell], sites int32 [cap or ncell], count int32 [1] on the device).  cap=None sizes `sites` for every cell (the
    caller trims it once it knows the count); otherwise rows past `cap` are dropped."""
    _chk(src); _f32(src)
    ncell = src.numel()
    dev = src.device
    imap = torch.empty((ncell,), dtype=torch.int32, device=dev)
    sites = torch.empty((ncell if cap is None else max(cap, 1),), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = torch.empty((int(_lib.get().pnsfm_sparse_compact_ws_ints(ncell)),), dtype=torch.int32, device=dev)
    _lib.check(_lib.get().pnsfm_sparse_compact(_ptr(src), ncell, _ptr(imap), _ptr(sites), sites.numel() if cap is None else cap,
                                               _ptr(count), _ptr(ws), _stream(src)), "sparse_compact")
    return imap, sites, count


def sparse_pool_cells(imap, B, h, w):
    _chk(imap); _i32(imap)
    mask = torch.empty((B * ((h + 1) // 2) * ((w + 1) // 2),), dtype=torch.float32, device=imap.device)     # odd grids: ceil
    _lib.check(_lib.get().pnsfm_sparse_pool_cells(_ptr(imap), B, h, w, _ptr(mask), _stream(imap)), "sparse_pool_cells")
    return mask


def sparse_neighbors(imap, sites, count, cap, h, w, ks):
    _chk(imap, sites, count); _i32(imap, sites, count)
    nbr = torch.empty((max(cap, 1), ks * ks), dtype=torch.int32, device=imap.device)
    _lib.check(_lib.get().pnsfm_sparse_neighbors(_ptr(imap), _ptr(sites), _ptr(count), cap, h, w, ks, _ptr(nbr), _stream(imap)),
               "sparse_neighbors")
    return nbr


def sparse_conv(feats, kern, nbr, count, ks, flip=False):
    """feats [cap, Cin], kern [ks*ks, Cin, Cout] -> [cap, Cout] (rows past count are zero)."""
    _chk(feats, kern, nbr, count); _f32(feats, kern); _i32(nbr, count)
    cap, Cin = feats.shape
    KK, Cin2, Cout = kern.shape
    if KK != ks * ks or Cin2 != Cin:
        raise RuntimeError("sparse_conv: kernel %s does not match %d input channels / k=%d" % (tuple(kern.shape), Cin, ks))
    out = torch.empty((cap, Cout), dtype=torch.float32, device=feats.device)
    _lib.check(_lib.get().pnsfm_sparse_conv(_ptr(feats), _ptr(kern), _ptr(nbr), _ptr(count), _ptr(out), cap, Cin, Cout, ks,
                                            1 if flip else 0, _stream(feats)), "sparse_conv")
    return out


def sparse_conv_backward_weight(feats, dout, nbr, count, ks):
    _chk(feats, dout, nbr, count); _f32(feats, dout); _i32(nbr, count)
    cap, Cin = feats.shape
    Cout = dout.shape[1]
    dk = torch.empty((ks * ks, Cin, Cout), dtype=torch.float32, device=feats.device)
    _lib.check(_lib.get().pnsfm_sparse_conv_backward_weight(_ptr(feats), _ptr(dout), _ptr(nbr), _ptr(count), _ptr(dk), cap, Cin, Cout,
                                                            ks, _stream(feats)), "sparse_conv_backward_weight")
    return dk


def sparse_maxpool_forward(fin, imap_in, sites_out, count_out, cap_out, h, w):
    _chk(fin, imap_in, sites_out, count_out); _f32(fin); _i32(imap_in, sites_out, count_out)
    C = fin.shape[1]
    fout = torch.empty((cap_out, C), dtype=torch.float32, device=fin.device)
    arg = torch.empty((cap_out, C), dtype=torch.int32, device=fin.device)
    _lib.check(_lib.get().pnsfm_sparse_maxpool_forward(_ptr(fin), _ptr(imap_in), _ptr(sites_out), _ptr(count_out), _ptr(fout), _ptr(arg),
                                                       cap_out, C, h, w, _stream(fin)), "sparse_maxpool_forward")
    return fout, arg


def sparse_maxpool_backward(dout, arg, cap_in):
    _chk(dout, arg); _f32(dout); _i32(arg)
    cap_out, C = dout.shape
    din = torch.empty((cap_in, C), dtype=torch.float32, device=dout.device)
    _lib.check(_lib.get().pnsfm_sparse_maxpool_backward(_ptr(dout), _ptr(arg), _ptr(din), cap_out, cap_in, C, _stream(dout)),
               "sparse_maxpool_backward")
    return din


def sparse_densify(feats, imap, B, hw):
    _chk(feats, imap); _f32(feats); _i32(imap)
    C = feats.shape[1]
    dense = torch.empty((B, C, hw), dtype=torch.float32, device=feats.device)
    _lib.check(_lib.get().pnsfm_sparse_densify(_ptr(feats), _ptr(imap), _ptr(dense), B, C, hw, _stream(feats)), "sparse_densify")
    return dense


def sparse_gather(dense, sites, count, cap):
    """dense [B, C, hw] -> rows [cap, C] at the active sites (zeros past count)."""
    _chk(dense, sites, count); _f32(dense); _i32(sites, count)
    B, C, hw = dense.shape
    rows = torch.empty((cap, C), dtype=torch.float32, device=dense.device)
    _lib.check(_lib.get().pnsfm_sparse_gather(_ptr(dense), _ptr(sites), _ptr(count), _ptr(rows), cap, C, hw, _stream(dense)),
               "sparse_gather")
    return rows


# ---------------------------------------------------------------------------------------------- adam
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, step):
    _chk(param, grad, exp_avg, exp_avg_sq); _f32(param, grad, exp_avg, exp_avg_sq)
    _lib.check(_lib.get().pnsfm_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), float(lr),
                                          float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale),
                                          int(step), _stream(param)), "adam_step")


def adam_flat_step(param, grad, exp_avg, exp_avg_sq, hp):
    """Adam on flat fp32 buffers with device-resident state hp = float[12]: [step, lr, beta1, beta2, eps, weight_decay, grad_scale, 1-beta1, 1-beta2, ...]."""
    _chk(param, grad, exp_avg, exp_avg_sq, hp); _f32(param, grad, exp_avg, exp_avg_sq, hp)
    _lib.check(_lib.get().pnsfm_adam_flat_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), _ptr(hp),
                                               _stream(param)), "adam_flat_step")


def adam_fused_plan(conv_items, segments, device):
    """Device tables of the two-launch optimizer tail (include/pnsfm.h: pnsfm_adam_pack_table / pnsfm_adam_segments).
    conv_items: [(w, g, m, v, hp, wp_fwd, wp_bwd)] with w a [Cout, Cin, k, k] VIEW into the parameter arena and g, m, v the same slices
    of the other arenas; segments: [(p, g, m, v, hp)] flat slices.  Returns (item table, n items, blocks, indices of the conv items the
    table took, segment table, n segments, blocks); conv items it did not take must be added to `segments` by the caller."""
    lib = _lib.get()
    isz = int(lib.pnsfm_adam_pack_item_bytes())
    host = (ctypes.c_ubyte * (isz * max(1, len(conv_items))))()
    n, blocks, covered = 0, 0, []
    for i, (w, g, m, v, hp, pf, pb) in enumerate(conv_items):
        _chk(w, g, m, v, hp, pf, pb); _f32(w, g, m, v, hp, pf, pb)
        Cout, Cin, ks, _ = w.shape
        rc = lib.pnsfm_adam_pack_item_fill(ctypes.addressof(host) + n * isz, _ptr(w), _ptr(g), _ptr(m), _ptr(v), _ptr(hp), _ptr(pf), _ptr(pb),
                                           Cin, Cout, ks, blocks)
        if rc < 0:
            _lib.check(rc, "adam_pack_item_fill")
        if rc > 0:
            n += 1
            blocks += rc
            covered.append(i)
    table = torch.frombuffer(bytearray(bytes(host)[:max(1, n) * isz]), dtype=torch.uint8).to(device)
    return table, n, blocks, covered


def adam_segment_table(segments, device):
    lib = _lib.get()
    ssz = int(lib.pnsfm_adam_seg_bytes())
    host = (ctypes.c_ubyte * (ssz * max(1, len(segments))))()
    blocks = 0
    for i, (p, g, m, v, hp) in enumerate(segments):
        _chk(p, g, m, v, hp); _f32(p, g, m, v, hp)
        rc = lib.pnsfm_adam_seg_fill(ctypes.addressof(host) + i * ssz, _ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(hp), p.numel(), blocks)
        if rc < 0:
            _lib.check(rc, "adam_seg_fill")
        blocks += rc
    table = torch.frombuffer(bytearray(bytes(host)[:max(1, len(segments)) * ssz]), dtype=torch.uint8).to(device)
    return table, len(segments), blocks


def adam_pack_table_run(table, n, blocks):
    if n:
        _lib.check(_lib.get().pnsfm_adam_pack_table(_ptr(table), n, blocks, _stream(table)), "adam_pack_table")


def adam_segments_run(table, n, blocks):
    if n:
        _lib.check(_lib.get().pnsfm_adam_segments(_ptr(table), n, blocks, _stream(table)), "adam_segments")


def adam_flat_update(param, grad, exp_avg, exp_avg_sq, hp, tick):
    """adam_flat_step on a slice of the arenas; tick: advance the group's step counter first (one slice per group and step)."""
    _chk(param, grad, exp_avg, exp_avg_sq, hp); _f32(param, grad, exp_avg, exp_avg_sq, hp)
    _lib.check(_lib.get().pnsfm_adam_flat_update(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), _ptr(hp),
                                                 1 if tick else 0, _stream(param)), "adam_flat_update")


# ------------------------------------------------------------------------------- batched region ops
class _RegionOp(ctypes.Structure):          # mirrors pnsfm_region_op (include/pnsfm.h)
    _fields_ = [('src', ctypes.c_void_p), ('dst', ctypes.c_void_p), ('n', ctypes.c_int * 4), ('src_stride', ctypes.c_longlong * 4),
                ('dst_stride', ctypes.c_longlong * 4), ('op', ctypes.c_int)]


REGION_COPY, REGION_ADD, REGION_ZERO = 0, 1, 2
MAX_REGION_OPS = 12


def region_ops(items):
    """items: list of (op, dst_view, src_view | None).  Views are <= 4-D windows (any strides) of fp32 tensors on one device; shapes of
    dst and src must match.  ONE kernel launch per MAX_REGION_OPS items (csrc/elementwise.hip: region_ops_kernel)."""
    if not items:
        return
    ref = items[0][1]
    from . import _seq
    sq = _seq.get()
    if sq is not None:
        sq.region_ops([(int(op), dst, src) for op, dst, src in items], launch_stream_raw(ref.device))
        return
    for i0 in range(0, len(items), MAX_REGION_OPS):
        chunk = items[i0:i0 + MAX_REGION_OPS]
        arr = (_RegionOp * len(chunk))()
        for k, (op, dst, src) in enumerate(chunk):
            if dst.dtype != torch.float32 or (src is not None and src.dtype != torch.float32):
                raise RuntimeError("region_ops needs float32 tensors")
            if _lib.REQUIRE_CUDA and not dst.is_cuda:
                raise RuntimeError("packnet_sfm HIP op got a %s tensor: the HIP kernels run on MI355X only, there is no CPU fallback" % dst.device)
            if dst.dim() > 4 or (src is not None and tuple(src.shape) != tuple(dst.shape)) or dst.device != ref.device:
                raise RuntimeError("region_ops: windows must be <= 4-D, of equal shape, on one device")
            pad = 4 - dst.dim()
            shape = (1,) * pad + tuple(dst.shape)
            if 0 in shape:
                raise RuntimeError("region_ops: empty window")
            arr[k].dst = dst.data_ptr()
            arr[k].src = src.data_ptr() if src is not None else None
            arr[k].op = int(op)
            for d in range(4):
                arr[k].n[d] = shape[d]
                arr[k].dst_stride[d] = 0 if d < pad else dst.stride(d - pad)
                arr[k].src_stride[d] = 0 if (src is None or d < pad) else src.stride(d - pad)
        _lib.check(_lib.get().pnsfm_region_ops(ctypes.byref(arr), len(chunk), _stream(ref)), "region_ops")


# ---------------------------------------------------------------------------------------------- prof
def calib_mfma(sink, blocks, iters):
    """The bare six-product bf16 MFMA stream (csrc/calib.hip); returns the bf16 flops of the launch."""
    _chk(sink); _f32(sink)
    assert sink.numel() >= blocks * 256
    _lib.check(_lib.get().pnsfm_calib_mfma(_ptr(sink), blocks, iters, _stream(sink)), "calib_mfma")
    return blocks * 4.0 * iters * 24.0 * (2.0 * 32 * 32 * 16)


def calib_copy(src, dst):
    _chk(src, dst); _f32(src, dst)
    assert src.numel() == dst.numel()
    _lib.check(_lib.get().pnsfm_calib_copy(_ptr(src), _ptr(dst), src.numel(), _stream(src)), "calib_copy")
    return 2.0 * 4.0 * src.numel()


def prof_enable(on):
    _lib.get().pnsfm_prof_enable(1 if on else 0)


def prof_reset():
    _lib.get().pnsfm_prof_reset()


def prof_collect(kind):
    ms, fl, n = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_longlong(0)
    _lib.get().pnsfm_prof_collect(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    return ms.value, fl.value, n.value


def prof_dump(path):
    _lib.check(_lib.get().pnsfm_prof_dump(path.encode()), "prof_dump")
