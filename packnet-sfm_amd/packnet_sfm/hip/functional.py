"""torch.autograd bindings of the gfx950 kernels (forward AND hand-written backward kernels).

These Functions are what the drop-in modules (networks/layers/packnet/layers01.py,
losses/multiview_photometric_loss.py) are made of.  Autograd only sequences the launches; every derivative
is computed by a HIP kernel of libpnsfm_hip.so.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops
from . import _seq
from ._lib import HipError as _HipError

# bumped by optimizers that update parameters through raw pointers (bypassing tensor._version)
_WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    _WEIGHT_EPOCH[0] += 1


def get_conv_math():
    from . import _lib
    return 'bx3' if _lib.get().pnsfm_get_conv_math() else 'f32'


def set_conv_math(mode):
    """Arithmetic of the conv forward / backward-data kernels: 'bx3' (default; fp32 rebuilt from exact bf16 splits on the
    bf16 matrix pipe, 6 products, fp32 accumulate) or 'f32' (v_mfma_f32_32x32x2_f32).  See include/pnsfm.h
    (pnsfm_set_conv_math).  The packed-weight layout depends on the mode: every cached packed weight is invalidated.
    Returns the previous mode."""
    from . import _lib
    m = {'bx3': 1, 'f32': 0, 1: 1, 0: 0}[mode]
    prev = _lib.get().pnsfm_set_conv_math(m)
    bump_weight_epoch()
    _PACK_TABLES.clear()      # the batched packer's device table records which weights the mode covers (and their layout)
    return 'bx3' if prev else 'f32'


# Fused / multi-tensor optimizers (torch.optim.Adam(fused=True), the one bench.py uses) update parameters WITHOUT bumping
# tensor._version, so the version alone cannot tell a packed weight copy is stale.  Every optimizer step of any
# torch.optim optimizer therefore advances the epoch (global post-step hook), which re-packs each conv weight on its next
# use: one read + two writes of the weight per step (~1.5 GB for PackNet01), inside the timed region of the bench.
# (Re-packing all weights right away on a side stream, underneath the next forward pass, measured slower: 82.7 vs 84.7
# img/s -- ~110 tiny launches contending with the first layers -- so packing stays lazy, in front of each conv.)
try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(lambda optimizer, args, kwargs: bump_weight_epoch())
except ImportError:      # older torch: callers must invalidate themselves (FlatAdam does)
    pass


class PackedConvWeight:
    """Per-layer cache of the MFMA-friendly weight layouts ([k*k][K][M], see csrc/conv2d.hip).

    Re-packed only when the parameter changed (tensor version / storage / optimizer epoch), so eval loops and the
    backward pass of the same step reuse the buffers.  Cost when stale: one read + two writes of the weight.
    """

    def __init__(self, volatile=False):
        self.wp_fwd = None
        self.wp_bwd = None
        self.key_fwd = None
        self.key_bwd = None
        self.volatile = volatile      # weight is a freshly computed tensor every call (e.g. a composed kernel): always repack
        self._serial = 0
        self._weight_ref = None       # the parameter this cache last packed (repack_all re-packs it in the batched launch)

    def get(self, weight, need_bwd):
        if self.volatile:
            self._serial += 1
            key = ('volatile', self._serial)
        else:
            # (torch.Size and torch.device compare and hash natively: no tuple() / str() per call -- ~300 calls per step)
            key = (weight._version, weight.data_ptr(), _WEIGHT_EPOCH[0], weight.shape, weight.device)
        do_f = self.key_fwd != key
        do_b = need_bwd and self.key_bwd != key
        if do_f or do_b:
            w = weight.detach()
            if do_f and self.wp_fwd is not None and (self.wp_fwd.device != w.device):
                self.wp_fwd = None
            if do_b and self.wp_bwd is not None and (self.wp_bwd.device != w.device):
                self.wp_bwd = None
            f, b = ops.conv2d_pack(w.contiguous(), self.wp_fwd if do_f else None, self.wp_bwd if do_b else None,
                                   want_fwd=do_f, want_bwd=do_b)
            if do_f:
                self.wp_fwd, self.key_fwd = f, key
            if do_b:
                self.wp_bwd, self.key_bwd = b, key
            if not self.volatile and weight.is_leaf and weight.requires_grad:
                _register_packed(self, weight)
        return self.wp_fwd, (self.wp_bwd if need_bwd else None)

    @staticmethod
    def key_of(weight):
        return (weight._version, weight.data_ptr(), _WEIGHT_EPOCH[0], weight.shape, weight.device)


# ---- batched re-pack: every conv weight of the model in ONE launch, right after the optimizer step ----------------------------
# The lazy re-pack in front of each conv is ~100 launches of ~10 us per step (profiles/r03_*: 0.55 ms of the 0.83 ms the packers
# take are per-launch floor, not bytes).  Caches register the LEAF parameter they pack; repack_all() -- called by FlatAdam.step
# -- rebuilds the device table when the set of (parameter, buffers) changed, runs pnsfm_conv2d_pack_table and stamps the caches
# with the new key, so the next forward finds them fresh.  PNSFM_PACK_BATCH=0 keeps the lazy path only.
_PACK_REG = {}          # id(cache) -> (weakref(cache), weakref(parameter))
_PACK_TABLES = {}       # device -> [signature, table tensor, n items, blocks, indices of the covered (cache, parameter) pairs]


def _register_packed(cache, weight):
    import weakref
    ent = _PACK_REG.get(id(cache))
    if ent is not None and ent[0]() is cache and ent[1]() is weight:
        return
    cid = id(cache)
    _PACK_REG[cid] = (weakref.ref(cache, lambda r, cid=cid: _PACK_REG.pop(cid, None)), weakref.ref(weight))
    cache._weight_ref = _PACK_REG[cid][1]


def _pack_pairs(only=None, exclude=None):
    """{device: [(cache, parameter)]} of the registered conv weights whose forward AND backward-data buffers exist; `only` /
    `exclude`: sets of id(parameter)."""
    by_dev = {}
    for cref, wref in list(_PACK_REG.values()):
        c, w = cref(), wref()
        if c is None or w is None or c.wp_fwd is None or c.wp_bwd is None or c.wp_fwd.device != w.device:
            continue
        if (only is not None and id(w) not in only) or (exclude is not None and id(w) in exclude):
            continue
        by_dev.setdefault(w.device, []).append((c, w))
    return by_dev


def _run_pack_table(slot, dev, pairs):
    """One pnsfm_conv2d_pack_table launch over `pairs` on the CURRENT stream; the device table is cached under `slot` and rebuilt
    when the set of (parameter, buffers) or the arithmetic mode changed.  Returns the pairs the table covers."""
    # the table's coverage and layout follow from the arithmetic mode too (a table built under 'bx3' replayed after
    # set_conv_math('f32') wrote split-bf16 bytes into f32-layout buffers and stamped them fresh)
    sig = (get_conv_math(),) + tuple((w.data_ptr(), tuple(w.shape), c.wp_fwd.data_ptr(), c.wp_bwd.data_ptr()) for c, w in pairs)
    tab = _PACK_TABLES.get(slot)
    if tab is None or tab[0] != sig:
        table, n, blocks, covered = ops.conv2d_pack_table_build([(w.detach(), c.wp_fwd, c.wp_bwd) for c, w in pairs], dev)
        tab = _PACK_TABLES[slot] = [sig, table, n, blocks, covered]      # (indices into `pairs`: no strong references kept)
    if not tab[2]:
        return []
    with torch.cuda.device(dev) if dev.type == 'cuda' else _nullctx():
        ops.conv2d_pack_table_run(tab[1], tab[2], tab[3])
    return [pairs[i] for i in tab[4]]


def _pack_batch_on():
    import os
    if os.environ.get('PNSFM_PACK_BATCH', '1') == '0' or not _PACK_REG:
        return False
    return not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


def repack_all(exclude=None):
    """Re-pack every registered conv weight whose forward AND backward-data buffers exist, one launch per device, and stamp the
    caches fresh.  `exclude`: ids of parameters already re-packed (the optimizer's fused tail).  Returns the number of weights packed.  No-op
    under PNSFM_PACK_BATCH=0 and while a stream is being captured."""
    if not _pack_batch_on():
        return 0
    done = 0
    for dev, pairs in _pack_pairs(exclude=exclude).items():
        covered = _run_pack_table((dev, 'all' if not exclude else ('rest', len(exclude))), dev, pairs)
        stamp_packed(covered)
        done += len(covered)
    return done


def stamp_packed(pairs):
    for c, w in pairs:
        c.key_fwd = c.key_bwd = PackedConvWeight.key_of(w)


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _WgradStream:
    """Weight gradients on a side HIP stream.  ON by default since round 5 (PNSFM_WGRAD_STREAM=0 switches it off): most launches of
    the backward pass are resident in ONE round of workgroups (tools/bx3_ablate.py: prologue -> matrix work -> store burst, all
    workgroups in step), so a second stream's kernel fills the ramp-up and the store tail of the first: +1.8 .. +2.2 % images/s in
    three same-box A/Bs on the atomics-free kernels (profiles/r04_ab_wgrad_side_stream.txt, profiles/r05_ab_streams.txt).  History:
    +4 % with the f32-MFMA kernels of round 1, a LOSS in round 2 (116.6 vs 123.8 img/s: the zero-fills and atomics of the then
    split-K / pixel-split kernels doubled the memset and event traffic), off from round 2 to round 4 -- in round 4 only because the
    bench's per-launch roofline is event-timed as-run; bench.py now prices the kernels in a separate pass with the side streams off.

    Within a layer's backward the data gradient is on the critical path (the next layer waits for it) while the weight
    gradient is only needed by the optimizer / the gradient all-reduce.  With a second stream the GPU can co-schedule
    the two kernels -- the low-resolution layers cannot fill 256 CUs on their own -- and the tail of one hides under
    the other.  Ordering: the side stream waits for the compute stream before each weight gradient (dy is ready), the
    inputs are `record_stream`-ed so the caching allocator does not recycle them early, and the compute stream joins the
    side stream once, from an autograd-engine callback at the end of the backward pass (before any optimizer / reducer
    kernel can touch the gradients).

    Only gradients of LEAF parameters that are used ONCE in the step go to the side stream: their sole consumer is the
    engine's AccumulateGrad (a pointer hand-over, no kernel).  A non-leaf weight (the composed kernel of the collapsed
    packing block) or a parameter used by several nodes (that block's Conv2d / Conv3d weights) has its gradient read or
    accumulated by compute-stream kernels during the same backward pass: for those the weight gradient still runs on the
    side stream next to the node's own data gradient, but the compute stream waits for it before the node returns.
    """
    import os as _os
    _env = _os.environ.get('PNSFM_WGRAD_STREAM', '1')
    enabled = _env not in ('0', '')
    # PNSFM_WGRAD_STREAM=<n> with n > 1: only layers whose gradient map has at most n pixels (batch x H x W) go to the side stream --
    # the low-resolution layers, whose launches are resident in one round and latency-bound (tools/bx3_ablate.py), overlap well;
    # the full-resolution ones are MFMA-bound and only contend
    max_pixels = int(_env) if _env.isdigit() and int(_env) > 1 else None

    @classmethod
    def use_for(cls, dy):
        if not (cls.enabled and dy.is_cuda):
            return False
        return cls.max_pixels is None or dy.shape[0] * dy.shape[-2] * dy.shape[-1] <= cls.max_pixels
    _streams = {}
    _pending = set()
    _uses = {}          # id(parameter) -> [forward uses whose backward has not run yet, shared-in-this-step flag]
    _cb_task = None     # autograd graph-task id whose end-of-backward callback is queued (None: no backward in flight)

    @classmethod
    def note_use(cls, recording, *params):
        """Called in forward for every parameter a node will produce a gradient for.  `recording`: autograd was recording
        when the op was CALLED (torch.is_grad_enabled() outside the Function -- inside forward grad mode is always off and
        needs_input_grad ignores no_grad), so validation under torch.no_grad() is not counted.  Entries of graphs that are
        never back-propagated make the next step fall back to the in-node wait and are dropped at the end of that step's
        backward pass."""
        if not recording:
            return
        for p in params:
            if p is not None and p.requires_grad:
                u = cls._uses.setdefault(id(p), [0, False])
                u[0] += 1
                if u[0] > 1:
                    u[1] = True

    @classmethod
    def side_ok(cls, *params):
        """Called once in backward: may this node's parameter gradients be left in flight on the side stream until the
        end-of-backward join (True), or must the node wait for them itself (False)?"""
        ok = True
        task = torch._C._current_graph_task_id()
        if cls._cb_task != task:        # first parameter gradient of this backward pass: arrange the end-of-pass clean-up
            if cls._cb_task is not None:
                # the previous backward pass never reached its callback (it raised): join what it left in flight now
                cls._join_pending()
            cls._cb_task = task
            from torch.autograd import Variable
            Variable._execution_engine.queue_callback(cls._end_of_backward)
        for p in params:
            if p is None:
                continue
            u = cls._uses.get(id(p))
            if not p.is_leaf or u is None or u[1]:
                ok = False
            # AccumulateGrad only hands the pointer over when .grad is None and nothing hooks the gradient; otherwise it
            # runs `p.grad += dw` (or the hook) on the COMPUTE stream, which knows nothing about the side stream: gradient
            # accumulation over several backward() calls, zero_grad(set_to_none=False), Tensor.register_hook
            if p.is_leaf and (p.grad is not None or p._backward_hooks):
                ok = False
            if u is not None:
                u[0] -= 1
                if u[0] <= 0:
                    del cls._uses[id(p)]
        return ok

    @classmethod
    def get(cls, device):
        st = cls._streams.get(device)
        if st is None:
            st = cls._streams[device] = torch.cuda.Stream(device=device)
        return st

    @classmethod
    def run(cls, fn, *tensors, detached=True, pure=True):
        """fn() -> outputs, executed on the side stream of tensors[0].device.  detached: nothing on the compute stream
        will touch the outputs before the end-of-backward join; otherwise returns (outputs, wait) and the caller must
        call wait() on the compute stream after it has enqueued its own independent work.
        pure (round 5): fn calls nothing but ops.* launches -- the fork is one C call (ops.stream_wait_stream) and the launches are
        redirected with ops.stream_override instead of torch's Stream.wait_stream + stream context (~50 -> ~10 us of host time per
        weight gradient, ~75 of them per step).  Outputs are then allocated in the COMPUTE stream's pool: they are gradients that live
        until the optimizer has run, long after the end-of-backward join, so the allocator cannot hand their memory out early."""
        dev = tensors[0].device
        side = cls.get(dev)
        if pure and _FAST_FORK:
            main_raw, side_raw = ops.current_raw_stream(dev), side.cuda_stream
            ops.stream_wait_stream(side_raw, main_raw)
            with ops.stream_override(side_raw, dev.index):
                out = fn()
            for t in tensors:
                t.record_stream(side)       # inputs live in the compute stream's pool but are read here
            if not detached:
                return out, (lambda: ops.stream_wait_stream(main_raw, side_raw))
            cls._pending.add(dev)
            return out
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = fn()
            ev = None if detached else side.record_event()
        for t in tensors:
            t.record_stream(side)           # inputs live in the compute stream's pool but are read here
        for o in out:
            if o is not None:
                o.record_stream(main)       # outputs live in the side stream's pool but are consumed on the compute stream
        if not detached:
            return out, (lambda: main.wait_event(ev))
        cls._pending.add(dev)            # joined by _end_of_backward (queued by side_ok, which every caller ran first)
        return out

    @classmethod
    def _end_of_backward(cls):
        """Autograd-engine callback at the end of a backward pass: the compute streams wait for the weight gradients
        still in flight, and the per-pass use counts are dropped."""
        cls._cb_task = None
        cls._join_pending()
        cls._uses.clear()

    @classmethod
    def _join_pending(cls):
        for dev in list(cls._pending):
            torch.cuda.current_stream(dev).wait_stream(cls.get(dev))
        cls._pending.clear()


_FAST_FORK = os.environ.get('PNSFM_FAST_FORK', '1') != '0'


def _seq_streams(dy, want_w):
    """(main raw stream, side raw stream or 0, side torch Stream or None) for a sequencer backward body: the weight gradient goes to the
    side stream under exactly the conditions _WgradStream.run(pure=True) takes the one-call fork."""
    if not dy.is_cuda:
        return 0, 0, None
    main_raw = ops.launch_stream_raw(dy.device)
    if want_w and _FAST_FORK and _WgradStream.use_for(dy):
        side = _WgradStream.get(dy.device)
        return main_raw, side.cuda_stream, side
    return main_raw, 0, None


def set_wgrad_stream(on):
    _WgradStream.enabled = bool(on)


# Gradient slots: id(parameter) -> (weak reference to the parameter, tensor its gradient should be WRITTEN into: a view of the
# flat gradient arena of rccl/flat_adam.py, which is also the all-reduce bucket).  Only used when the node's gradient is the
# parameter's only one this step (leaf, used once, .grad undefined, no hooks -- exactly `side_ok`): then AccumulateGrad just
# adopts the view.  The entry dies with the parameter (weakref callback) and is only honoured while the weak reference still
# points at the very tensor object asking -- CPython reuses ids, so a later model must never find a dead optimizer's arena.
_GRAD_SLOTS = {}


def _drop_slot(key, ref):
    ent = _GRAD_SLOTS.get(key)
    if ent is not None and ent[0] is ref:
        del _GRAD_SLOTS[key]


def register_grad_slots(pairs):
    """pairs: iterable of (parameter, view).  Returns a handle whose .remove() unregisters them (also done automatically when a
    parameter is garbage-collected, and when a newer optimizer registers the same parameter: the newer registration wins)."""
    import weakref
    mine = []
    for p, v in pairs:
        key = id(p)
        ref = weakref.ref(p, lambda r, key=key: _drop_slot(key, r))
        _GRAD_SLOTS[key] = (ref, v)
        mine.append((key, ref))

    class _Handle:
        def remove(self):
            for key, ref in mine:
                _drop_slot(key, ref)
            del mine[:]
    return _Handle()


def _slot_of(p):
    """The registered gradient slot of parameter `p`, or None (stale entries of a collected parameter never match)."""
    if p is None:
        return None
    ent = _GRAD_SLOTS.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    return ent[1]


def _slots_for(weight, bias, usable):
    if not usable or not _GRAD_SLOTS:
        return None, None
    sw = _slot_of(weight)
    sb = _slot_of(bias)
    if sw is None or (bias is not None and sb is None):
        return None, None
    if tuple(sw.shape) != tuple(weight.shape) or (bias is not None and sb.numel() != bias.numel()):
        return None, None           # the parameter was re-shaped behind the optimizer's back: plain gradient path
    return sw, sb


def side_streams(device):
    """The side streams gradients may still be in flight on (weight-gradient stream, independent-branch stream) -- for a consumer
    that wants ITS stream to wait for them instead of joining them into the compute stream (rccl/reducer.py)."""
    if device.type != 'cuda':
        return []
    return [st for st in (_WgradStream._streams.get(device), _BRANCH_STREAMS.get(device)) if st is not None]


def join_wgrad_stream(device):
    """Make the current stream of `device` wait for every weight gradient launched so far (no-op when unused) and for the
    independent-branch stream (branch_stream) -- the gradient all-reduce calls this before it gathers a bucket in the middle
    of the backward pass, and a bucket may hold parameters of both networks."""
    if device.type != 'cuda':
        return
    cur = torch.cuda.current_stream(device)
    for st in (_WgradStream._streams.get(device), _BRANCH_STREAMS.get(device)):
        if st is not None and st != cur:
            cur.wait_stream(st)


# Independent-branch stream: the pose network shares nothing with the depth network until the loss, and its kernels are tiny
# (a few workgroups each, latency-bound), so models/SfmModel.py enqueues it on a second HIP stream where it fills the launch gaps
# and tail rounds of the depth network's kernels.  Autograd replays every node on the stream its forward ran on and orders the
# two streams with events, so the pose network's backward pass overlaps the depth decoder's backward pass the same way.
# ON by default since round 5 (PNSFM_BRANCH_STREAM=0 / set_branch_stream(False) switch it off): +0.85 % images/s
# (profiles/r04_ab_branch_stream.txt), bit-identical results (tests/test_gpu_round4.py::test_branch_stream_is_bit_identical).
# Kernels that share the GPU stretch each other, so bench.py measures the per-kernel roofline in a pass with both side streams off.
_BRANCH_STREAMS = {}
_BRANCH_ON = os.environ.get('PNSFM_BRANCH_STREAM', '1') == '1'


def set_branch_stream(on):
    global _BRANCH_ON
    _BRANCH_ON = bool(on)


def branch_stream(t):
    """The second compute stream of t's device, or None (CPU tensors, switched off)."""
    if not (_BRANCH_ON and torch.is_tensor(t) and t.is_cuda):
        return None
    st = _BRANCH_STREAMS.get(t.device)
    if st is None:
        st = _BRANCH_STREAMS[t.device] = torch.cuda.Stream(device=t.device)
    return st


class Conv2dFn(Function):
    """y = conv2d(zero_pad_{k//2}(x), weight) + bias, stride 1 (fp32 implicit GEMM on the matrix pipe; arithmetic: set_conv_math)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, recording=True, tap=False):
        x = x.contiguous()
        need_dx = ctx.needs_input_grad[0]
        wp_fwd, wp_bwd = cache.get(weight, need_dx)
        Cout, Cin, ks, _ = weight.shape
        if x.shape[1] != Cin:
            raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (x.shape[1], Cin))
        sq = _seq.get()
        if sq is not None:
            y = sq.conv2d_forward([x], wp_fwd, bias, Cin, Cout, ks, ops.launch_stream_raw(x.device))
        else:
            y = ops.conv2d_forward(x, wp_fwd, bias.detach() if bias is not None else None, Cout, ks)
        ctx.save_for_backward(x, wp_bwd if wp_bwd is not None else x.new_empty(0))
        ctx.meta = (Cin, Cout, ks, bias is not None)
        ctx.params = (weight, bias)
        _WgradStream.note_use(recording, weight, bias)
        # tap (see conv2d_tap): the input comes back as a second output; what its other consumers send back arrives here as g_tap
        # (None when nobody read the tap: no zero tensor is materialised) and is added inside the backward-data launch
        if tap:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, g_tap=None):
        if dy is None:         # only the tap was used: its gradient passes straight through
            return g_tap, None, None, None, None, None
        x, wp_bwd = ctx.saved_tensors
        Cin, Cout, ks, has_bias = ctx.meta
        dy = dy.contiguous()
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        detached = _WgradStream.side_ok(*ctx.params)
        sw, sb = _slots_for(ctx.params[0], ctx.params[1], detached and ctx.needs_input_grad[1])
        sq = _seq.get()
        if sq is not None and (g_tap is None or get_conv_math() == 'bx3'):
            main_raw, side_raw, side = _seq_streams(dy, want_w)
            dx, dw, db = sq.conv2d_backward(dy, [x], wp_bwd, Cin, Cout, ks, has_bias, bool(ctx.needs_input_grad[0]), bool(want_w), sw, sb, g_tap,
                                            main_raw, side_raw, bool(detached), side)
            if side is not None and detached:
                _WgradStream._pending.add(dy.device)
            return dx, dw, db, None, None, None
        wait = None
        if want_w and _WgradStream.use_for(dy):
            r = _WgradStream.run(lambda: ops.conv2d_backward_weight(x, dy, ks, want_bias=has_bias, dw_out=sw, db_out=sb), x, dy,
                                 detached=detached)
            (dw, db), wait = (r, None) if detached else r
            want_w = False
        if ctx.needs_input_grad[0]:
            dx = _dgrad_tap(dy, wp_bwd, Cin, ks, g_tap)
        if want_w:
            dw, db = ops.conv2d_backward_weight(x, dy, ks, want_bias=has_bias, dw_out=sw, db_out=sb)
        if wait is not None:
            wait()
        return dx, dw, db, None, None, None


def conv2d(x, weight, bias, cache):
    return Conv2dFn.apply(x, weight, bias, cache, torch.is_grad_enabled())


# Round 5: gradient taps.  A tensor read by a convolution AND by something else (the encoder feature that is also a decoder skip input;
# the ResidualConv input read by conv1 and by the 1x1 shortcut; the decoder feature read by the next unpack block and by the InvDepth
# head) receives two gradients, which autograd adds with an elementwise kernel: 3 passes over the tensor and a launch, 18 times per
# PackNet01 step (1.2 GB).  conv2d_tap / conv2d_gn_act_tap return (y, x_tap): x_tap is x again, but as an OUTPUT of the convolution's
# node -- the other consumers read x_tap, their gradient comes back to the node as g_tap, and the convolution's backward-data launch
# adds it in its epilogue (ops.conv2d_backward_data(addend=...)): one read of the addend, no extra launch.  For a tensor with two
# consumers the bits are the untapped graph's (fl(acc + addend) either way); with three (ResidualConv input that is also a skip) the
# association differs from autograd's arrival order -- fixed, deterministic, inside every golden's tolerance.
# PNSFM_GRAD_TAPS=0 / set_grad_taps(False): plain graph.
_GRAD_TAPS = os.environ.get('PNSFM_GRAD_TAPS', '1') != '0'


def set_grad_taps(on):
    global _GRAD_TAPS
    _GRAD_TAPS = bool(on)


def grad_taps():
    return _GRAD_TAPS


def _dgrad_tap(dy, wp_bwd, Cin, ks, g_tap):
    """backward-data (+ the tap's gradient).  The tap was granted at forward time under the arithmetic mode in force then; should the
    mode have left 'bx3' since, the sum is formed the ordinary way instead of raising."""
    if g_tap is not None and get_conv_math() != 'bx3':
        return ops.conv2d_backward_data(dy, wp_bwd, Cin, ks) + g_tap
    return ops.conv2d_backward_data(dy, wp_bwd, Cin, ks, addend=g_tap)


def _tap_ok(weight):
    # the addend lives in the split-bf16 kernels' epilogue: backward-data's K is the layer's Cout
    return get_conv_math() == 'bx3' and weight.shape[0] >= 16


def conv2d_tap(x, weight, bias, cache):
    """(conv2d(x), x_tap)."""
    if not (_GRAD_TAPS and torch.is_grad_enabled() and x.requires_grad and _tap_ok(weight)):
        return conv2d(x, weight, bias, cache), x
    return Conv2dFn.apply(x, weight, bias, cache, True, True)


class Conv2dCatFn(Function):
    """y = conv2d(zero_pad(cat((x0, x1[, x2]), 1)), weight) + bias with the concatenation folded into the K loop of the split-bf16
    kernels (the decoder's skip connections, reference PackNet01.py:138-174): the concatenated tensor -- 254 MB for iconv1 at
    192x640 batch 4 -- is neither written nor read.  Backward-data produces the gradient of the whole concatenation in one tensor;
    the inputs receive channel slices of it (views)."""

    @staticmethod
    def forward(ctx, weight, bias, cache, recording, cat_wgrad, *xs):
        xs = tuple(t.contiguous() for t in xs)
        ctx.cat_wgrad = cat_wgrad
        need_dx = any(ctx.needs_input_grad[5:])
        wp_fwd, wp_bwd = cache.get(weight, need_dx)
        Cout, Cin, ks, _ = weight.shape
        if sum(t.shape[1] for t in xs) != Cin:
            raise RuntimeError("conv2d_cat: inputs have %d channels, weight expects %d" % (sum(t.shape[1] for t in xs), Cin))
        sq = _seq.get()
        if sq is not None:
            y = sq.conv2d_forward(list(xs), wp_fwd, bias, Cin, Cout, ks, ops.launch_stream_raw(xs[0].device))
        else:
            y = ops.conv2d_forward_cat(xs, wp_fwd, bias.detach() if bias is not None else None, Cout, ks)
        ctx.save_for_backward(wp_bwd if wp_bwd is not None else xs[0].new_empty(0), *xs)
        ctx.meta = (Cin, Cout, ks, bias is not None)
        ctx.params = (weight, bias)
        _WgradStream.note_use(recording, weight, bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        wp_bwd, *xs = ctx.saved_tensors
        Cin, Cout, ks, has_bias = ctx.meta
        dy = dy.contiguous()
        dxs = [None] * len(xs)
        dw = db = None
        detached = _WgradStream.side_ok(*ctx.params)
        sw, sb = _slots_for(ctx.params[0], ctx.params[1], detached and ctx.needs_input_grad[0])
        want_w = ctx.needs_input_grad[0] or (has_bias and ctx.needs_input_grad[1])
        sq = _seq.get()
        if sq is not None and (not want_w or (ctx.cat_wgrad and get_conv_math() == 'bx3')):
            need_dx = any(ctx.needs_input_grad[5:])
            main_raw, side_raw, side = _seq_streams(dy, want_w)
            dx, dw, db = sq.conv2d_backward(dy, list(xs), wp_bwd, Cin, Cout, ks, has_bias, need_dx, bool(want_w), sw, sb, None, main_raw, side_raw,
                                            bool(detached), side)
            if side is not None and detached:
                _WgradStream._pending.add(dy.device)
            if need_dx:
                c0 = 0
                for i, t in enumerate(xs):
                    if ctx.needs_input_grad[5 + i]:
                        dxs[i] = dx[:, c0:c0 + t.shape[1]]
                    c0 += t.shape[1]
            return (dw, db, None, None, None) + tuple(dxs)

        def wgrad():
            # (the envelope was decided at forward time under the arithmetic mode in force THEN; a set_conv_math('f32') between
            # forward and backward takes the multi-source kernel away -- ADVICE r04 -- so the mode is re-checked here)
            if ctx.cat_wgrad and get_conv_math() == 'bx3':
                return ops.conv2d_backward_weight_cat(xs, dy, ks, want_bias=has_bias, dw_out=sw, db_out=sb)
            # outside the multi-source weight-gradient kernel's envelope (decided once, in conv2d_cat): concatenate for this kernel
            return ops.conv2d_backward_weight(torch.cat(xs, 1), dy, ks, want_bias=has_bias, dw_out=sw, db_out=sb)

        wait = None
        if want_w and _WgradStream.use_for(dy):
            # (pure: the closure launches kernels only -- the concatenating fallback calls torch.cat and takes the stream-context path)
            r = _WgradStream.run(wgrad, dy, *xs, detached=detached, pure=(len(xs) == 1 or (ctx.cat_wgrad and get_conv_math() == 'bx3')))
            (dw, db), wait = (r, None) if detached else r
            want_w = False
        if any(ctx.needs_input_grad[5:]):
            dx = ops.conv2d_backward_data(dy, wp_bwd, Cin, ks)
            c0 = 0
            for i, t in enumerate(xs):
                if ctx.needs_input_grad[5 + i]:
                    dxs[i] = dx[:, c0:c0 + t.shape[1]]
                c0 += t.shape[1]
        if want_w:
            dw, db = wgrad()
        if wait is not None:
            wait()
        return (dw, db, None, None, None) + tuple(dxs)


def _cat_wgrad_ok(xs, weight):
    """May the multi-source weight-gradient kernels (wgrad3 / wgrad4, pnsfm_conv2d_backward_weight_cat) take these sources?  Their
    channel tiles are 32 wide (64 with two ci tiles per wave): every source but the last must end on such a boundary.  Decided ONCE
    per call here (the backward used to find out by catching every HipError, real launch failures included)."""
    return ops.conv2d_cat_wgrad_supported([t.shape[1] for t in xs], weight.shape[0], xs[0].shape[2], xs[0].shape[3], weight.shape[2],
                                          B=xs[0].shape[0])


def conv2d_cat(xs, weight, bias, cache):
    """xs: tuple of 2 or 3 NCHW tensors.  Falls back to torch.cat + conv2d when the shape is outside the multi-source FORWARD
    kernel's envelope (first / second tensor not ending on a 16-channel boundary, < 16 channels, f32 arithmetic mode); a shape
    the forward takes but the multi-source weight-gradient kernels do not concatenates for that one kernel only."""
    import os
    C0 = xs[0].shape[1]
    ok = os.environ.get('PNSFM_CAT_FOLD', '1') != '0' and len(xs) in (2, 3) and C0 % 16 == 0 and (len(xs) == 2 or (C0 + xs[1].shape[1]) % 16 == 0) and get_conv_math() == 'bx3' \
        and sum(t.shape[1] for t in xs) >= 16
    if not ok:
        return conv2d(torch.cat(xs, 1), weight, bias, cache)
    return Conv2dCatFn.apply(weight, bias, cache, torch.is_grad_enabled(), _cat_wgrad_ok(xs, weight), *xs)


class ConvGnActFn(Function):
    """act(GroupNorm_G(conv2d(zero_pad(cat(xs, 1)), weight) + bias)) as ONE autograd node (round 5): the reference's Conv2D block
    (layers01.py:28-37).  xs: 1..3 tensors (the decoder's concatenations stay folded into the K loop).  One node instead of two halves
    the autograd / Python overhead of the block (profiles/r05_host_profile.txt: ~35 us per Function.apply, the host needs 16-18 ms
    to enqueue a 24 ms step).
    Round 6: both bodies are ONE call into the block sequencer (csrc/seq/pnsfm_seq.cpp) when it is loaded.  (The conv epilogue that
    also left the GroupNorm statistics behind -- round 5, measured neutral twice, profiles/r05_ab_conv_gn_stats.txt -- is gone.)"""

    @staticmethod
    def forward(ctx, weight, bias, gamma, beta, cache, recording, cat_wgrad, G, eps, act, tap, *xs):
        xs = tuple(t.contiguous() for t in xs)
        ctx.cat_wgrad = cat_wgrad
        need_dx = any(ctx.needs_input_grad[11:])
        wp_fwd, wp_bwd = cache.get(weight, need_dx)
        Cout, Cin, ks, _ = weight.shape
        if sum(t.shape[1] for t in xs) != Cin:
            raise RuntimeError("conv_gn_act: inputs have %d channels, weight expects %d" % (sum(t.shape[1] for t in xs), Cin))
        bias_d = bias.detach() if bias is not None else None
        sq = _seq.get()
        if sq is not None:
            # ONE call: conv -> GroupNorm statistics -> normalise + activation (csrc/seq/pnsfm_seq.cpp)
            out, y, ms = sq.conv_gn_act_forward(list(xs), wp_fwd, bias_d, gamma, beta, Cin, Cout, ks, G, float(eps), act,
                                                ops.launch_stream_raw(xs[0].device))
        else:
            y = ops.conv2d_forward(xs[0], wp_fwd, bias_d, Cout, ks) if len(xs) == 1 else ops.conv2d_forward_cat(xs, wp_fwd, bias_d, Cout, ks)
            out, mean, rstd = ops.groupnorm_act_forward(y, None, gamma.detach(), beta.detach(), G, eps, act)
            ms = mean._base        # [mean | rstd]: one allocation (ops.groupnorm_act_forward)
        ctx.save_for_backward(wp_bwd if wp_bwd is not None else y.new_empty(0), y, gamma, beta, ms, *xs)
        ctx.meta = (Cin, Cout, ks, bias is not None, G, act)
        ctx.params = (weight, bias)
        _WgradStream.note_use(recording, weight, bias)
        # tap (single input only, see conv2d_tap): x comes back as a second output, its other consumers' gradient is added in backward-data
        if tap:
            ctx.set_materialize_grads(False)
            return out, xs[0].view_as(xs[0])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, g_tap=None):
        if dout is None:       # only the tap was used: its gradient passes straight through
            return (None,) * 11 + (g_tap,)
        wp_bwd, y, gamma, beta, ms, *xs = ctx.saved_tensors
        Cin, Cout, ks, has_bias, G, act = ctx.meta
        dxs = [None] * len(xs)
        dw = db = None
        detached = _WgradStream.side_ok(*ctx.params)
        sw, sb = _slots_for(ctx.params[0], ctx.params[1], detached and ctx.needs_input_grad[0])
        want_w = ctx.needs_input_grad[0] or (has_bias and ctx.needs_input_grad[1])
        sq = _seq.get()
        if sq is not None and (not want_w or len(xs) == 1 or (ctx.cat_wgrad and get_conv_math() == 'bx3')) and \
                (g_tap is None or get_conv_math() == 'bx3'):
            # ONE call: GroupNorm backward -> fork -> weight gradient (side stream) -> backward-data (+ tap) [-> join]
            need_dx = any(ctx.needs_input_grad[11:])
            main_raw, side_raw, side = _seq_streams(dout, want_w)
            dx, dw, db, dgamma, dbeta = sq.conv_gn_act_backward(dout, y, gamma, beta, ms, list(xs), wp_bwd, Cin, Cout, ks, G, act, has_bias, need_dx,
                                                                bool(want_w), sw, sb, g_tap if len(xs) == 1 else None, main_raw, side_raw,
                                                                bool(detached), side)
            if side is not None and detached:
                _WgradStream._pending.add(dout.device)
            if need_dx:
                if len(xs) == 1:
                    dxs[0] = dx
                else:
                    c0 = 0
                    for i, t in enumerate(xs):
                        if ctx.needs_input_grad[11 + i]:
                            dxs[i] = dx[:, c0:c0 + t.shape[1]]
                        c0 += t.shape[1]
            return (dw, db, dgamma, dbeta, None, None, None, None, None, None, None) + tuple(dxs)
        dy, dgamma, dbeta = ops.groupnorm_act_backward(dout.contiguous(), y, None, gamma.detach(), beta.detach(), ms[0], ms[1], G, act)

        def wgrad():
            if len(xs) == 1:
                return ops.conv2d_backward_weight(xs[0], dy, ks, want_bias=has_bias, dw_out=sw, db_out=sb)
            if ctx.cat_wgrad and get_conv_math() == 'bx3':
                return ops.conv2d_backward_weight_cat(xs, dy, ks, want_bias=has_bias, dw_out=sw, db_out=sb)
            return ops.conv2d_backward_weight(torch.cat(xs, 1), dy, ks, want_bias=has_bias, dw_out=sw, db_out=sb)

        wait = None
        if want_w and _WgradStream.use_for(dy):
            # (pure: the closure launches kernels only -- the concatenating fallback calls torch.cat and takes the stream-context path)
            r = _WgradStream.run(wgrad, dy, *xs, detached=detached, pure=(len(xs) == 1 or (ctx.cat_wgrad and get_conv_math() == 'bx3')))
            (dw, db), wait = (r, None) if detached else r
            want_w = False
        if any(ctx.needs_input_grad[11:]):
            dx = _dgrad_tap(dy, wp_bwd, Cin, ks, g_tap if len(xs) == 1 else None)
            if len(xs) == 1:
                dxs[0] = dx
            else:
                c0 = 0
                for i, t in enumerate(xs):
                    if ctx.needs_input_grad[11 + i]:
                        dxs[i] = dx[:, c0:c0 + t.shape[1]]
                    c0 += t.shape[1]
        if want_w:
            dw, db = wgrad()
        if wait is not None:
            wait()
        return (dw, db, dgamma, dbeta, None, None, None, None, None, None, None) + tuple(dxs)


_CONV_GN_FUSE = os.environ.get('PNSFM_CONV_GN_FUSE', '1') != '0'

def set_conv_gn_fuse(on):
    global _CONV_GN_FUSE
    _CONV_GN_FUSE = bool(on)


def conv2d_gn_act(x, weight, bias, gamma, beta, cache, G=16, eps=1e-5, act=ops.ACT_ELU):
    """The Conv2D block: x a tensor or a tuple of 2-3 tensors standing for their channel concatenation."""
    xs = tuple(x) if isinstance(x, (tuple, list)) else (x,)
    if len(xs) > 1:
        C0 = xs[0].shape[1]
        fold = os.environ.get('PNSFM_CAT_FOLD', '1') != '0' and len(xs) in (2, 3) and C0 % 16 == 0 and \
            (len(xs) == 2 or (C0 + xs[1].shape[1]) % 16 == 0) and get_conv_math() == 'bx3' and sum(t.shape[1] for t in xs) >= 16
        if not fold:
            xs = (torch.cat(xs, 1),)
    if not _CONV_GN_FUSE:
        y = conv2d(xs[0], weight, bias, cache) if len(xs) == 1 else conv2d_cat(xs, weight, bias, cache)
        return groupnorm_act(y, gamma, beta, G, eps, act)
    cat_wgrad = _cat_wgrad_ok(xs, weight) if len(xs) > 1 else False
    return ConvGnActFn.apply(weight, bias, gamma, beta, cache, torch.is_grad_enabled(), cat_wgrad, G, eps, act, False, *xs)


def conv2d_gn_act_tap(x, weight, bias, gamma, beta, cache, G=16, eps=1e-5, act=ops.ACT_ELU):
    """The Conv2D block with a gradient tap on its (single-tensor) input: (out, x_tap) -- see conv2d_tap."""
    if not (_GRAD_TAPS and _CONV_GN_FUSE and torch.is_grad_enabled() and torch.is_tensor(x) and x.requires_grad and _tap_ok(weight)):
        return conv2d_gn_act(x, weight, bias, gamma, beta, cache, G, eps, act), x
    return ConvGnActFn.apply(weight, bias, gamma, beta, cache, True, False, G, eps, act, True, x)


class Conv2dStride2Fn(Function):
    """Stride-2 conv with zero padding k//2 (PoseNet).  Forward and weight-gradient run the strided MFMA kernels; the
    data-gradient is the stride-1 backward-data kernel on dy zero-upsampled onto the input grid."""

    @staticmethod
    def forward(ctx, x, weight, bias, cache, recording=True):
        x = x.contiguous()
        need_dx = ctx.needs_input_grad[0]
        wp_fwd, wp_bwd = cache.get(weight, need_dx)
        Cout, Cin, ks, _ = weight.shape
        y = ops.conv2d_forward_strided(x, wp_fwd, bias.detach() if bias is not None else None, Cout, ks, 2)
        ctx.save_for_backward(x, wp_bwd if wp_bwd is not None else x.new_empty(0))
        ctx.meta = (Cin, Cout, ks, bias is not None)
        ctx.params = (weight, bias)
        _WgradStream.note_use(recording, weight, bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wp_bwd = ctx.saved_tensors
        Cin, Cout, ks, has_bias = ctx.meta
        dy = dy.contiguous()
        dx = dw = db = None
        want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        detached = _WgradStream.side_ok(*ctx.params)
        wait = None
        if want_w and _WgradStream.use_for(dy):
            r = _WgradStream.run(lambda: ops.conv2d_backward_weight_strided(x, dy, ks, 2, want_bias=has_bias), x, dy,
                                 detached=detached)
            (dw, db), wait = (r, None) if detached else r
            want_w = False
        if ctx.needs_input_grad[0]:
            B, _, H, W = x.shape
            up = dy.new_zeros((B, Cout, H, W))
            up[:, :, ::2, ::2] = dy                      # dy[y][x] sits at (2y, 2x) of the input grid
            dx = ops.conv2d_backward_data(up, wp_bwd, Cin, ks)
        if want_w:
            dw, db = ops.conv2d_backward_weight_strided(x, dy, ks, 2, want_bias=has_bias)
        if wait is not None:
            wait()
        return dx, dw, db, None, None


def conv2d_stride2(x, weight, bias, cache):
    return Conv2dStride2Fn.apply(x, weight, bias, cache, torch.is_grad_enabled())


class GroupNormActFn(Function):
    """y = act(GroupNorm_G(x [+ res]))."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, G, eps, act):
        sq = _seq.get()
        if sq is not None:
            y, ms, x, res = sq.gn_act_forward(x, res, gamma, beta, G, float(eps), act, ops.launch_stream_raw(x.device))
        else:
            x = x.contiguous()
            res = res.contiguous() if res is not None else None
            y, mean, rstd = ops.groupnorm_act_forward(x, res, gamma.detach(), beta.detach(), G, eps, act)
            ms = mean._base
        ctx.save_for_backward(x, res if res is not None else x.new_empty(0), gamma, beta, ms)
        ctx.meta = (G, act, res is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, res, gamma, beta, ms = ctx.saved_tensors
        G, act, has_res = ctx.meta
        sq = _seq.get()
        if sq is not None:
            dx, dgamma, dbeta = sq.gn_act_backward(dy, x, res if has_res else None, gamma, beta, ms, G, act, ops.launch_stream_raw(x.device))
        else:
            dx, dgamma, dbeta = ops.groupnorm_act_backward(dy.contiguous(), x, res if has_res else None, gamma.detach(), beta.detach(),
                                                           ms[0], ms[1], G, act)
        return dx, (dx if has_res else None), dgamma, dbeta, None, None, None


def groupnorm_act(x, gamma, beta, G=16, eps=1e-5, act=ops.ACT_ELU, res=None):
    return GroupNormActFn.apply(x, res, gamma, beta, G, eps, act)


class SpaceToDepthFn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.space_to_depth(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.depth_to_space(dy.contiguous())


class DepthToSpaceFn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.depth_to_space(x.contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.space_to_depth(dy)       # dy is often a channel slice (cat backward): handled without a copy


def space_to_depth(x):
    return SpaceToDepthFn.apply(x)


def depth_to_space(x):
    return DepthToSpaceFn.apply(x)


class Conv3d1to8Fn(Function):
    """[B,D,H,W] -> [B,NF*D,H,W]: Conv3d(1,NF,3,pad 1) over (channel,y,x), channel index f*D+d; NF = 8 (PackNet01)
    or 4 (PackNetSlim01 / PackNetSAN01)."""

    @staticmethod
    def forward(ctx, p, w3, b3, recording=True):
        p = p.contiguous()
        out = ops.conv3d_forward(p, w3.detach().contiguous(), b3.detach().contiguous())
        ctx.save_for_backward(p, w3)
        ctx.params = (w3, b3)
        _WgradStream.note_use(recording, w3, b3)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        p, w3 = ctx.saved_tensors
        dout = dout.contiguous()
        dp = dw3 = db3 = None
        want_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        detached = _WgradStream.side_ok(*ctx.params)
        wait = None
        if want_w and _WgradStream.use_for(dout):
            r = _WgradStream.run(lambda: ops.conv3d_backward_weight(p, dout), p, dout, detached=detached)
            (dw3, db3), wait = (r, None) if detached else r
            want_w = False
        if ctx.needs_input_grad[0]:
            dp = ops.conv3d_backward_data(dout, w3.detach().contiguous())
        if want_w:
            dw3, db3 = ops.conv3d_backward_weight(p, dout)
        if wait is not None:
            wait()
        return dp, dw3, db3, None


def conv3d_1to8(p, w3, b3):
    return Conv3d1to8Fn.apply(p, w3, b3, torch.is_grad_enabled())


class ComposePackWeightFn(Function):
    """W_eff = Conv3d(1->8) composed with the Conv2d that follows it inside PackLayerConv3d (no non-linearity between
    them, reference layers01.py:243-246):

        W_eff[co, ci, U, V] = sum_{f,dz,dy,dx} W3[f,dz,dy,dx] * W2[co, f*D + (ci-dz+1), U-dy, V-dx]       (k -> k+2 taps)

    which is exactly the backward-data stencil of the Conv3d applied to the zero-ring-padded W2, so the gfx950 conv3d
    kernels do the composition and its two gradients (dW2 = forward stencil of dW_eff, dW3 = weight-gradient stencil)."""

    @staticmethod
    def forward(ctx, W2, W3, recording=True):
        W2pad = torch.nn.functional.pad(W2.detach(), (1, 1, 1, 1)).contiguous()      # [C, 8D, k+2, k+2]
        w3 = W3.detach().contiguous()
        Weff = ops.conv3d_backward_data(W2pad, w3)                                    # [C, D, k+2, k+2]
        ctx.save_for_backward(W2pad, w3)
        ctx.params = (W2, W3)
        _WgradStream.note_use(recording, W2, W3)
        return Weff

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        W2pad, w3 = ctx.saved_tensors
        g = g.contiguous()
        _WgradStream.side_ok(*ctx.params)            # bookkeeping only: this node always runs on the compute stream
        dW2 = dW3 = None
        if ctx.needs_input_grad[0]:
            dW2 = ops.conv3d_forward(g, w3, torch.zeros(w3.shape[0], device=g.device, dtype=g.dtype))[:, :, 1:-1, 1:-1].contiguous()
        if ctx.needs_input_grad[1]:
            dW3, _ = ops.conv3d_backward_weight(g, W2pad)
        return dW2, dW3, None


def compose_pack_weight(W2, W3):
    return ComposePackWeightFn.apply(W2, W3, torch.is_grad_enabled())


class ComposePackParamsFn(Function):
    """(W_eff, bias_eff) of the composed packing convolution in one node: ComposePackWeightFn's composition plus

        bias_eff[co] = b2[co] + sum_f b3[f] * sum_{ci,taps} W2[co, f*D + ci, taps]

    (a Conv3d bias is a constant plane, which the Conv2d turns into a per-channel constant away from the border).  One node instead
    of two plus a chain of reshape / sum / mul / add: forward = pad, composition stencil, one bias kernel; backward = the two
    stencils and two small kernels that also fold the bias path's rank-one term into dW2 (no second gradient to accumulate)."""

    @staticmethod
    def forward(ctx, W2, b2, W3, b3, recording=True):
        W2d, w3, b3d = W2.detach().contiguous(), W3.detach().contiguous(), b3.detach().contiguous()
        W2pad = torch.nn.functional.pad(W2d, (1, 1, 1, 1))                            # [C, d*D, k+2, k+2]
        Weff = ops.conv3d_backward_data(W2pad, w3)                                    # [C, D, k+2, k+2]
        bias_eff, Ssum = ops.pack_bias_eff_forward(W2d, None if b2 is None else b2.detach().contiguous(), b3d)
        ctx.save_for_backward(W2pad, w3, b3d, Ssum)
        ctx.params = (W2, W3)
        ctx.k = W2.shape[-1]
        _WgradStream.note_use(recording, W2, W3)
        return Weff, bias_eff

    @staticmethod
    @once_differentiable
    def backward(ctx, gW, gb):
        W2pad, w3, b3, Ssum = ctx.saved_tensors
        _WgradStream.side_ok(*ctx.params)            # bookkeeping only: this node always runs on the compute stream
        need_W2, need_b2, need_W3, need_b3 = ctx.needs_input_grad[:4]
        C = W2pad.shape[0]
        if gW is None:
            gW = W2pad.new_zeros((C, W2pad.shape[1] // w3.shape[0], ctx.k + 2, ctx.k + 2))
        if gb is None:
            gb = W2pad.new_zeros((C,))
        gW, gb = gW.contiguous(), gb.contiguous()
        dW2 = dW3 = db3 = None
        if need_W2 or need_b3:
            # the composition's gradient on the padded taps; cropped and completed by the bias path's term in the same kernel
            full = ops.conv3d_forward(gW, w3, None) if need_W2 else W2pad       # (W2pad: right shape, not read when dW2 is not wanted)
            db3, dW2 = ops.pack_bias_eff_backward(gb, Ssum, b3, full, ctx.k, want_db3=need_b3, want_dW2=need_W2)
        if need_W3:
            dW3, _ = ops.conv3d_backward_weight(gW, W2pad)
        return dW2, (gb if need_b2 else None), dW3, db3, None


def compose_pack_params(W2, b2, W3, b3):
    return ComposePackParamsFn.apply(W2, b2, W3, b3, torch.is_grad_enabled())


def _R(op, dst, src=None):
    return (op, dst, src)


class PackBorderSplitFn(Function):
    """Collapsed packing block, input side: P -> (P itself for the interior convolution, the top+bottom row strips,
    the left+right column strips; strips batched along dim 0).  Plain slicing would make autograd zero-fill and add a
    full-size dP four times (slice_backward); here the strip gradients are added straight into the interior path's dP.
    Round 4: the four strip gathers are ONE launch (ops.region_ops), and so are the four gradient accumulations."""

    @staticmethod
    def forward(ctx, P, S, lr_t=False):
        B, C, h, w = P.shape
        ctx.S, ctx.lr_t = S, lr_t
        tb = P.new_empty((2 * B, C, S, w))
        # lr_t (round 6): the column strips are stored TRANSPOSED, [2B, C, S, h] -- rows of h pixels instead of rows of S = 3 / 5 --
        # so that every kernel behind them (Conv3d stencil, convolution, weight gradient) sees a map as wide as the row strips'
        lr = P.new_empty((2 * B, C, S, h)) if lr_t else P.new_empty((2 * B, C, h, S))
        lrv = lr.transpose(2, 3) if lr_t else lr
        ops.region_ops([_R(ops.REGION_COPY, tb[:B], P[:, :, :S]), _R(ops.REGION_COPY, tb[B:], P[:, :, h - S:]),
                        _R(ops.REGION_COPY, lrv[:B], P[:, :, :, :S]), _R(ops.REGION_COPY, lrv[B:], P[:, :, :, w - S:])])
        return P.view_as(P), tb, lr

    @staticmethod
    @once_differentiable
    def backward(ctx, dP, d_tb, d_lr):
        S = ctx.S
        if dP is None:
            raise RuntimeError('pack_border_split: the interior path must be used')
        dP = dP.contiguous()        # the interior conv's freshly written backward-data buffer: accumulate in place
        B, h, w = dP.shape[0], dP.shape[2], dP.shape[3]
        # (the row strips and the column strips overlap in the corners: two launches keep every destination single-writer per launch)
        # (... and when the map is smaller than two strips -- S <= h < 2S -- the leading and the trailing strip overlap too: the
        # operations of ONE region_ops launch must not share destinations (include/pnsfm.h), so those run as separate launches)
        def add_pair(a, b, overlap):
            if overlap:
                ops.region_ops([a])
                ops.region_ops([b])
            else:
                ops.region_ops([a, b])
        if d_tb is not None:
            d_tb = d_tb.contiguous()
            add_pair(_R(ops.REGION_ADD, dP[:, :, :S], d_tb[:B]), _R(ops.REGION_ADD, dP[:, :, h - S:], d_tb[B:]), h < 2 * S)
        if d_lr is not None:
            d_lr = d_lr.contiguous()
            if ctx.lr_t:
                d_lr = d_lr.transpose(2, 3)
            add_pair(_R(ops.REGION_ADD, dP[:, :, :, :S], d_lr[:B]), _R(ops.REGION_ADD, dP[:, :, :, w - S:], d_lr[B:]), w < 2 * S)
        return dP, None, None


def pack_border_split(P, S, lr_t=False):
    return PackBorderSplitFn.apply(P, S, lr_t)


class StripSelectFn(Function):
    """z: [2B, C, S, w] (dim=2) or [2B, C, h, S] (dim=3), the Conv3d of the batched strips.  Keeps the 2r rows/cols whose
    Conv3d neighbourhood lies inside the strip: the first 2r of the leading half, the last 2r of the trailing half.
    Backward writes every element of dz exactly once (no zero-fill + copy).  One launch each way (ops.region_ops)."""

    @staticmethod
    def forward(ctx, z, B, r, dim):
        ctx.meta = (B, r, dim, tuple(z.shape))
        n2 = 2 * r
        if dim == 2:
            out = z.new_empty((z.shape[0], z.shape[1], n2, z.shape[3]))
            ops.region_ops([_R(ops.REGION_COPY, out[:B], z[:B, :, :n2]), _R(ops.REGION_COPY, out[B:], z[B:, :, 1:])])
        else:
            out = z.new_empty((z.shape[0], z.shape[1], z.shape[2], n2))
            ops.region_ops([_R(ops.REGION_COPY, out[:B], z[:B, :, :, :n2]), _R(ops.REGION_COPY, out[B:], z[B:, :, :, 1:])])
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, r, dim, shape = ctx.meta
        g = g.contiguous()
        dz = g.new_empty(shape)
        n2 = 2 * r
        if dim == 2:
            ops.region_ops([_R(ops.REGION_COPY, dz[:B, :, :n2], g[:B]), _R(ops.REGION_ZERO, dz[:B, :, n2:]),
                            _R(ops.REGION_COPY, dz[B:, :, 1:], g[B:]), _R(ops.REGION_ZERO, dz[B:, :, :1])])
        else:
            ops.region_ops([_R(ops.REGION_COPY, dz[:B, :, :, :n2], g[:B]), _R(ops.REGION_ZERO, dz[:B, :, :, n2:]),
                            _R(ops.REGION_COPY, dz[B:, :, :, 1:], g[B:]), _R(ops.REGION_ZERO, dz[B:, :, :, :1])])
        return dz, None, None, None


def strip_select(z, B, r, dim):
    return StripSelectFn.apply(z, B, r, dim)


class PackBorderPasteFn(Function):
    """Collapsed packing block, output side: overwrite the r-pixel frame of the interior result y (in place: y is the
    interior conv's own output buffer) with the strip results o_tb [2B,C,2r,w] / o_lr [2B,C,h,2r] (only their outer r
    rows / columns are valid).  Replaces two torch.cat copies forward and a zero-fill + copy backward; one launch forward, one
    backward (every element of d_tb, d_lr and dy has exactly one writer)."""

    @staticmethod
    def forward(ctx, y, o_tb, o_lr, r, lr_t=False):
        B, h, w = y.shape[0], y.shape[2], y.shape[3]
        ctx.r, ctx.lr_t = r, lr_t
        o_tb, o_lr = o_tb.contiguous(), o_lr.contiguous()
        s_lr = tuple(o_lr.shape)
        if lr_t:                     # [2B, C, 2r, h] holds the column strips transposed
            o_lr = o_lr.transpose(2, 3)
        ops.region_ops([_R(ops.REGION_COPY, y[:, :, r:h - r, :r], o_lr[:B, :, r:h - r, :r]),
                        _R(ops.REGION_COPY, y[:, :, r:h - r, w - r:], o_lr[B:, :, r:h - r, r:]),
                        _R(ops.REGION_COPY, y[:, :, :r], o_tb[:B, :, :r]),
                        _R(ops.REGION_COPY, y[:, :, h - r:], o_tb[B:, :, r:])])
        ctx.mark_dirty(y)
        ctx.shapes = (tuple(o_tb.shape), s_lr)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        r = ctx.r
        g = g.contiguous()
        B, h, w = g.shape[0], g.shape[2], g.shape[3]
        s_tb, s_lr = ctx.shapes
        d_tb, d_lr_st, dy = g.new_empty(s_tb), g.new_empty(s_lr), torch.empty_like(g)
        d_lr = d_lr_st.transpose(2, 3) if ctx.lr_t else d_lr_st
        ops.region_ops([
            # d_tb [2B,C,2r,w]: leading half rows 0..r-1 <- top frame, rows r.. zero; trailing half rows r.. <- bottom frame, rows ..r-1 zero
            _R(ops.REGION_COPY, d_tb[:B, :, :r], g[:, :, :r]), _R(ops.REGION_ZERO, d_tb[:B, :, r:]),
            _R(ops.REGION_COPY, d_tb[B:, :, r:], g[:, :, h - r:]), _R(ops.REGION_ZERO, d_tb[B:, :, :r]),
            # dy: the interior of g, frame zero (the frame's gradient belongs to the strips)
            _R(ops.REGION_COPY, dy[:, :, r:h - r, r:w - r], g[:, :, r:h - r, r:w - r]),
            _R(ops.REGION_ZERO, dy[:, :, :r]), _R(ops.REGION_ZERO, dy[:, :, h - r:]),
            _R(ops.REGION_ZERO, dy[:, :, r:h - r, :r]), _R(ops.REGION_ZERO, dy[:, :, r:h - r, w - r:])])
        ops.region_ops([
            # d_lr [2B,C,h,2r]: only rows r..h-r-1 of the outer r columns carry gradient (the corners went to the row strips)
            _R(ops.REGION_COPY, d_lr[:B, :, r:h - r, :r], g[:, :, r:h - r, :r]), _R(ops.REGION_ZERO, d_lr[:B, :, r:h - r, r:]),
            _R(ops.REGION_COPY, d_lr[B:, :, r:h - r, r:], g[:, :, r:h - r, w - r:]), _R(ops.REGION_ZERO, d_lr[B:, :, r:h - r, :r]),
            _R(ops.REGION_ZERO, d_lr[:, :, :r]), _R(ops.REGION_ZERO, d_lr[:, :, h - r:])])
        return dy, d_tb, d_lr_st, None, None


def pack_border_paste(y, o_tb, o_lr, r, lr_t=False):
    return PackBorderPasteFn.apply(y, o_tb, o_lr, r, lr_t)


class InvDepthActFn(Function):
    @staticmethod
    def forward(ctx, x, min_depth):
        y = ops.invdepth_act_forward(x.contiguous(), min_depth)
        ctx.save_for_backward(y)
        ctx.min_depth = min_depth
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.invdepth_act_backward(dy.contiguous(), y, ctx.min_depth), None


def invdepth_act(x, min_depth):
    return InvDepthActFn.apply(x, min_depth)


class InvDepthConvFn(Function):
    """y = sigmoid(conv3x3(zero_pad1(x), w) + b) / min_depth with ONE output channel (the InvDepth head), fused."""

    @staticmethod
    def forward(ctx, x, weight, bias, min_depth):
        x = x.contiguous()
        w = weight.detach().contiguous()
        y = ops.invdepth_conv_forward(x, w, bias.detach(), min_depth)
        ctx.save_for_backward(x, w, y)
        ctx.min_depth = min_depth
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dz = ops.invdepth_act_backward(dy.contiguous(), y, ctx.min_depth)
        dx, dw, db = ops.invdepth_conv_backward(x, w, dz)
        return dx, dw, db, None


def invdepth_conv(x, weight, bias, min_depth):
    return InvDepthConvFn.apply(x, weight, bias, min_depth)


class PoseVec2MatFn(Function):
    """[N,6] (tx,ty,tz,rx,ry,rz) -> [N,4,4] rigid transforms, R = Rx*Ry*Rz (euler)."""

    @staticmethod
    def forward(ctx, vec):
        vec = vec.contiguous()
        ctx.save_for_backward(vec)
        return ops.pose_vec2mat_forward(vec)

    @staticmethod
    @once_differentiable
    def backward(ctx, dmat):
        (vec,) = ctx.saved_tensors
        return ops.pose_vec2mat_backward(vec, dmat.contiguous())


def pose_vec2mat44(vec):
    return PoseVec2MatFn.apply(vec)


class UpsampleNearestFn(Function):
    """Nearest-neighbour up-sampling by an integer factor (F.interpolate(mode='nearest') / nn.Upsample of the reference:
    models/model_utils.py:163-180, networks/depth/PackNet01.py:87-89); backward = s x s block sums."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return ops.upsample_nearest_forward(x.contiguous(), s)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return ops.upsample_nearest_backward(dy.contiguous(), ctx.s), None


def upsample_nearest(x, size=None, scale_factor=None):
    """x [B,C,h,w] up-sampled (nearest) to `size` = (H, W) or by `scale_factor`.  Integer factors of fp32 device maps run the gfx950
    kernel (factor 1: the tensor itself); anything else is torch's interpolate (same index rule for integer factors)."""
    h, w = x.shape[-2:]
    if size is not None:
        H, W = int(size[-2]), int(size[-1])
    else:
        H, W = int(h * scale_factor), int(w * scale_factor)
    s = H // h if h else 0
    from . import _lib
    if s >= 1 and (H, W) == (h * s, w * s) and (x.is_cuda or not _lib.REQUIRE_CUDA) and ops.upsample_nearest_ok(x, s):
        return x if s == 1 else UpsampleNearestFn.apply(x, s)
    if size is not None:
        return torch.nn.functional.interpolate(x, size=(H, W), mode='nearest')
    return torch.nn.functional.interpolate(x, scale_factor=scale_factor, mode='nearest')


class LossCombineFn(Function):
    """loss = mean_i P[i] + weight * mean_i (S[i] / 2^i) from the per-scale device scalars, one launch each way (the reference does
    this with Python sums of 0-dim tensors: losses/multiview_photometric_loss.py:248-252, 275-280, 337-338; same operation order).
    Returns the 3-vector (loss, weighted smoothness, photometric); only [0] carries a gradient."""

    @staticmethod
    def forward(ctx, weight, n, *terms):
        ctx.meta = (float(weight), n, len(terms) - n)
        ctx.set_materialize_grads(False)
        out = ops.loss_combine_forward([t.detach().reshape(()) for t in terms[:n]], [t.detach().reshape(()) for t in terms[n:]],
                                       weight)
        smooth, photo = out[1], out[2]
        ctx.mark_non_differentiable(smooth, photo)
        return out[0], smooth, photo

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _gs, _gp):
        weight, n, ns = ctx.meta
        if g is None:
            return (None,) * (2 + n + ns)
        d = ops.loss_combine_backward(g.contiguous(), n, ns, weight)
        return (None, None) + tuple(d[i] for i in range(n)) + tuple(d[8 + i] for i in range(ns))


def loss_combine(photometric, smoothness, weight):
    """(loss, weighted smoothness term, photometric term): 0-dim tensors, the last two without gradient (metrics)."""
    return LossCombineFn.apply(float(weight), len(photometric), *photometric, *smoothness)


class SupervisedLossFn(Function):
    """One scale of the supervised inverse-depth loss (l1 / mse / abs_rel / berhu / silog, optionally only where gt > 0)."""

    @staticmethod
    def forward(ctx, pred, gt, method, sparse):
        pred, gt = pred.contiguous(), gt.contiguous()
        loss, ws = ops.supervised_loss_forward(pred, gt, method, sparse)
        ctx.save_for_backward(pred, gt, ws)
        ctx.meta = (method, sparse)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        pred, gt, ws = ctx.saved_tensors
        method, sparse = ctx.meta
        g = g.reshape(1).to(torch.float32).contiguous()
        return ops.supervised_loss_backward(pred, gt, ws, g, method, sparse), None, None, None


def supervised_loss(pred, gt, method, sparse):
    return SupervisedLossFn.apply(pred, gt, method, sparse)


class ViewSynthesisFn(Function):
    """warped[j] = grid_sample(ref[j], project(reconstruct(1/inv_depth))) for the J context views of one scale.
    Differentiable w.r.t. inv_depth and the [J,B,4,4] pose matrices (the context image is data)."""

    @staticmethod
    def forward(ctx, inv_depth, ref, K, refK, T, padding_mode=0):
        inv_depth, ref, K, refK, T = (t.contiguous() for t in (inv_depth, ref, K, refK, T))
        warped = ops.view_synthesis_forward(inv_depth, ref, K, refK, T.detach(), padding_mode)
        ctx.save_for_backward(inv_depth, ref, K, refK, T)
        ctx.padding_mode = padding_mode
        return warped

    @staticmethod
    @once_differentiable
    def backward(ctx, d_warped):
        inv_depth, ref, K, refK, T = ctx.saved_tensors
        d_inv, dT = ops.view_synthesis_backward(d_warped.contiguous(), inv_depth, ref, K, refK, T.detach(), ctx.padding_mode)
        return d_inv, None, None, None, dT, None


def view_synthesis(inv_depth, ref, K, refK, T, padding_mode='zeros'):
    """padding_mode: 'zeros' | 'border' | 'reflection' (F.grid_sample semantics, align_corners=True)."""
    if padding_mode not in ops.PADDING_MODES:
        raise ValueError('Unknown padding_mode {}'.format(padding_mode))
    return ViewSynthesisFn.apply(inv_depth, ref, K, refK, T, ops.PADDING_MODES[padding_mode])


class NrsProjectFn(Function):
    """(unit directions [3,h,w], ray surface [3,h,w]) -> expected candidate coordinates [h,w,2] (row, col): the softmax over the
    41x41 patch of ray-surface dot products of GenericCamera.project, fused; differentiable w.r.t. both inputs."""

    @staticmethod
    def forward(ctx, direction, ray, temperature):
        direction, ray = direction.contiguous(), ray.contiguous()
        coords, stat = ops.nrs_project_forward(direction, ray, temperature)
        ctx.save_for_backward(direction, ray, coords, stat)
        ctx.temperature = temperature
        return coords

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        direction, ray, coords, stat = ctx.saved_tensors
        gdir, gray = ops.nrs_project_backward(direction, ray, coords, stat, g.contiguous(), ctx.temperature,
                                              want_dir=ctx.needs_input_grad[0], want_ray=ctx.needs_input_grad[1])
        return gdir, gray, None


def nrs_project(direction, ray, temperature):
    return NrsProjectFn.apply(direction, ray, temperature)


class SparseConvFn(Function):
    """out[n] = sum_i kern[i]^T . feats[nbr[n][i]] over the active sites (ME.MinkowskiConvolution, stride 1, no bias)."""

    @staticmethod
    def forward(ctx, feats, kern, nbr, count, ks):
        feats, kern = feats.contiguous(), kern.contiguous()
        ctx.save_for_backward(feats, kern, nbr, count)
        ctx.ks = ks
        return ops.sparse_conv(feats, kern.detach(), nbr, count, ks, flip=False)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        feats, kern, nbr, count = ctx.saved_tensors
        dout = dout.contiguous()
        dfeats = dkern = None
        if ctx.needs_input_grad[0]:
            dfeats = ops.sparse_conv(dout, kern.detach().transpose(1, 2).contiguous(), nbr, count, ctx.ks, flip=True)
        if ctx.needs_input_grad[1]:
            dkern = ops.sparse_conv_backward_weight(feats, dout, nbr, count, ctx.ks)
        return dfeats, dkern, None, None, None


def sparse_conv(feats, kern, nbr, count, ks):
    return SparseConvFn.apply(feats, kern, nbr, count, ks)


class SparseMaxPoolFn(Function):
    """ME.MinkowskiMaxPooling(3, 2) on feature rows: fine rows (imap_in of the [B, h, w] grid) -> coarse rows (sites_out / count_out)."""

    @staticmethod
    def forward(ctx, fin, imap_in, sites_out, count_out, cap_out, h, w):
        fin = fin.contiguous()
        fout, arg = ops.sparse_maxpool_forward(fin, imap_in, sites_out, count_out, cap_out, h, w)
        ctx.save_for_backward(arg)
        ctx.cap_in = fin.shape[0]
        return fout

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        return ops.sparse_maxpool_backward(dout.contiguous(), arg, ctx.cap_in), None, None, None, None, None, None


def sparse_maxpool(fin, imap_in, sites_out, count_out, cap_out, h, w):
    return SparseMaxPoolFn.apply(fin, imap_in, sites_out, count_out, cap_out, h, w)


class SparseDensifyFn(Function):
    """feature rows -> dense [B, C, h, w], zeros at inactive cells (densify_features, reference minkowski.py:60-83)."""

    @staticmethod
    def forward(ctx, feats, imap, sites, count, B, h, w):
        ctx.save_for_backward(sites, count)
        ctx.cap = feats.shape[0]
        return ops.sparse_densify(feats.contiguous(), imap, B, h * w).view(B, feats.shape[1], h, w)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        sites, count = ctx.saved_tensors
        B, C, h, w = g.shape
        return ops.sparse_gather(g.contiguous().view(B, C, h * w), sites, count, ctx.cap), None, None, None, None, None, None


def sparse_densify(feats, imap, sites, count, B, h, w):
    return SparseDensifyFn.apply(feats, imap, sites, count, B, h, w)


class SparseGatherFn(Function):
    """dense [B, C, h, w] -> feature rows at the active sites (the dense half of map_add_features, reference minkowski.py:116-136)."""

    @staticmethod
    def forward(ctx, dense, imap, sites, count, cap):
        B, C, h, w = dense.shape
        ctx.save_for_backward(imap)
        ctx.shape = (B, C, h, w)
        return ops.sparse_gather(dense.contiguous().view(B, C, h * w), sites, count, cap)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (imap,) = ctx.saved_tensors
        B, C, h, w = ctx.shape
        return ops.sparse_densify(g.contiguous(), imap, B, h * w).view(B, C, h, w), None, None, None, None


def sparse_gather(dense, imap, sites, count, cap):
    return SparseGatherFn.apply(dense, imap, sites, count, cap)


REDUCE_MIN, REDUCE_MEAN = 0, 1


class PhotometricFn(Function):
    """mean over pixels of min/mean over candidates of (w*SSIM-loss + (1-w)*L1); scalar float32.  Without clipping the scalar is
    finished on the device (pixel mean as float32) and backward hands the upstream gradient to the kernel as a device scalar: no
    ATen launch around the kernels (round 4; the `d * g` pass over the [J,B,3,H,W] gradient alone was 25 us per scale)."""

    @staticmethod
    def forward(ctx, warped, ref, target, ssim_w, C1, C2, automask, reduce_op, clip_loss=0.0):
        warped, ref, target = warped.contiguous(), ref.contiguous(), target.contiguous()
        J, B, _, H, W = warped.shape
        ctx.meta = (ssim_w, C1, C2, automask, reduce_op, B * H * W, clip_loss > 0.0)
        if clip_loss > 0.0:
            loss_sum, argmin = ops.photometric_forward(warped, ref, target, ssim_w, C1, C2, automask, reduce_op, clip_loss)
            ctx.save_for_backward(warped, target, argmin)
            return (loss_sum / float(B * H * W)).to(torch.float32).reshape(())
        loss, argmin = ops.photometric_forward_mean(warped, ref, target, ssim_w, C1, C2, automask, reduce_op)
        ctx.save_for_backward(warped, target, argmin)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        warped, target, argmin = ctx.saved_tensors
        ssim_w, C1, C2, automask, reduce_op, n, clip = ctx.meta
        up = g.reshape(1).to(torch.float32).contiguous()
        d = ops.photometric_backward_dev(warped, target, argmin, 1.0 / n, up, ssim_w, C1, C2, automask, reduce_op, clip)
        return d, None, None, None, None, None, None, None, None


class PhotometricL1Fn(Function):
    """L1-only photometric loss (ssim_loss_weight == 0) with the 'min' reduce op and / or clipping: the reference reduces and clips
    per-CHANNEL candidate maps then (multiview_photometric_loss.py:205-219, 238-246); csrc/loss.hip: l1cand_*_kernel."""

    @staticmethod
    def forward(ctx, warped, ref, target, automask, reduce_op, clip_loss):
        warped, ref, target = warped.contiguous(), ref.contiguous(), target.contiguous()
        J, B, _, H, W = warped.shape
        loss, rec = ops.photometric_l1_forward(warped, ref, target, automask, reduce_op, clip_loss)
        ctx.save_for_backward(warped, target, rec)
        ctx.meta = (automask, reduce_op, B * H * W)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        warped, target, rec = ctx.saved_tensors
        automask, reduce_op, n = ctx.meta
        up = g.reshape(1).to(torch.float32).contiguous()
        return ops.photometric_l1_backward(warped, target, rec, 1.0 / n, up, automask, reduce_op), None, None, None, None, None


def warp_photometric(inv_depth, ref, target, K, refK, T, ssim_w, C1, C2, automask, reduce_op, clip_loss=0.0, padding_mode='zeros'):
    """view_synthesis + photometric terms of one scale.  (Round 5's single-kernel form -- the warp inside the photometric tile loaders --
    measured -0.3 % twice, profiles/r05_ab_loss_fuse.txt, and was removed in round 6.)"""
    if padding_mode not in ops.PADDING_MODES:
        raise ValueError('Unknown padding_mode {}'.format(padding_mode))
    warped = view_synthesis(inv_depth, ref, K, refK, T, padding_mode)
    return photometric(warped, ref, target, ssim_w, C1, C2, automask, reduce_op, clip_loss)


def photometric(warped, ref, target, ssim_w, C1, C2, automask, reduce_op, clip_loss=0.0):
    if ssim_w == 0.0 and (reduce_op == REDUCE_MIN or clip_loss > 0.0):
        return PhotometricL1Fn.apply(warped, ref, target, automask, reduce_op, clip_loss)
    return PhotometricFn.apply(warped, ref, target, ssim_w, C1, C2, automask, reduce_op, clip_loss)


class SmoothnessFn(Function):
    """mean|Sx| + mean|Sy| of the edge-aware first differences of a (mean-normalised) inverse depth map."""

    @staticmethod
    def forward(ctx, inv_norm, image):
        inv_norm, image = inv_norm.contiguous(), image.contiguous()
        B, _, H, W = image.shape
        sums = ops.smoothness_forward(inv_norm, image)
        ctx.save_for_backward(inv_norm, image)
        nx, ny = float(B * H * (W - 1)), float(B * (H - 1) * W)
        ctx.meta = (nx, ny)
        return (sums[0] / nx + sums[1] / ny).to(torch.float32).reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        inv_norm, image = ctx.saved_tensors
        nx, ny = ctx.meta
        return ops.smoothness_backward(inv_norm, image, 1.0 / nx, 1.0 / ny) * g, None


def smoothness(inv_norm, image):
    return SmoothnessFn.apply(inv_norm, image)


class SmoothnessNormFn(Function):
    """mean|Sx| + mean|Sy| of the edge-aware first differences of inv_depth / clamp(mean_hw(inv_depth), 1e-6): the reference's
    calc_smoothness_loss (multiview_photometric_loss.py:255-285) for ONE scale, the mean normalisation fused into the kernels
    (3 launches forward, 2 backward; the ATen form around SmoothnessFn is ~10 + ~12)."""

    @staticmethod
    def forward(ctx, inv_depth, image):
        inv_depth, image = inv_depth.contiguous(), image.contiguous()
        loss, mean = ops.smoothness_norm_forward(inv_depth, image)
        ctx.save_for_backward(inv_depth, image, mean)
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        inv_depth, image, mean = ctx.saved_tensors
        up = g.reshape(1).to(torch.float32).contiguous()
        return ops.smoothness_norm_backward(inv_depth, image, mean, up), None


def smoothness_norm(inv_depth, image):
    return SmoothnessNormFn.apply(inv_depth, image)
