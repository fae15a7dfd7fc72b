"""Whole training steps as hipGraphs.

One PackNet01 self-supervised step is ~1 200 kernel launches, about half of them shorter than the ~20-40 us the host
needs to issue one (Python autograd node + ctypes call + allocator).  Measured on MI355X the eager step leaves the GPU idle
for 3.7 ms of 50 (rocprofv3 kernel trace, low-resolution encoder/decoder layers and the loss), and the gap grows as the
kernels get faster.  `GraphedTrainStep` therefore captures zero_grad -> forward -> loss -> backward -> optimizer step
ONCE into a hipGraph (torch.cuda.CUDAGraph is hipGraph on ROCm: every launch of libpnsfm_hip.so goes to torch's current
stream, which is the capturing stream; the weight-gradient side stream forks/joins through captured events) and replays
it with a single host call per step.

What stays outside the graph because it is Python control flow in the reference:
  * the random left-right flip of the depth-network input (SfmModel.py:84 of the reference draws `random.random()` per
    step): one graph per flip state is captured and the draw picks the graph, so the RNG sequence is the reference's;
  * the learning-rate schedule: capturable torch optimizers keep `lr` / `step` in device tensors, FlatAdam does the same.

Requirements: static shapes (the batch is copied into static input tensors before each replay), an optimizer that is
capture-safe (`torch.optim.Adam(..., fused=True, capturable=True)` or `packnet_sfm.rccl.flat_adam.FlatAdam`), at least
one eager step before capture (autotuning of the conv kernels synchronises and cannot run inside a capture), no host
read-back inside the model's forward -- and NO live reference to the outputs of earlier eager steps (drop the loss
tensor): an autograd graph that is still alive keeps its AccumulateGrad nodes, which are bound to the stream they were
created on (the default stream), and the captured backward would hand its gradients over to work on that stream outside
the capture.  As in PyTorch's whole-network-capture recipe the constructor therefore runs one full warm-up step on a
side stream first, so that every AccumulateGrad node the capture meets was created off the default stream.
The optimizer's state tensors are captured by address: restore a checkpoint with in-place copies (load_state_dict of
torch optimizers REPLACES the state tensors; re-create the GraphedTrainStep after it).
"""
import gc
import random

import torch

from packnet_sfm.hip import functional as HF


def _clone_static(obj):
    if torch.is_tensor(obj):
        return obj.clone()
    if isinstance(obj, dict):
        return {k: _clone_static(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_clone_static(v) for v in obj]
    return obj


def _copy_into(dst, src):
    if torch.is_tensor(dst):
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_into(d, s)


class GraphedTrainStep:
    """
    Parameters
    ----------
    model : SfmModel-like module; `model(batch, progress=...)` returns {'loss': Tensor[1], ...}
    optimizer : capture-safe optimizer (see module docstring); may be a rccl.hvd.DistributedOptimizer ONLY for world size 1
    example_batch : dict of device tensors (shapes/dtypes are frozen)
    progress : float, handed to the model (ProgressiveScaling input; constant inside a graph)
    flip_prob : probability of the mirrored depth-network pass (None: read model.flip_lr_prob)
    """

    def __init__(self, model, optimizer, example_batch, progress=0.0, flip_prob=None):
        if not torch.cuda.is_available():
            raise RuntimeError('GraphedTrainStep needs an MI355X (hipGraph capture)')
        self.model, self.optimizer, self.progress = model, optimizer, progress
        self.flip_prob = float(getattr(model, 'flip_lr_prob', 0.0) if flip_prob is None else flip_prob)
        self.batch = _clone_static(example_batch)
        self.graphs, self.loss = {}, {}
        flips = [False] if self.flip_prob <= 0.0 else ([True] if self.flip_prob >= 1.0 else [False, True])
        self._warm_up(flips[0])
        for flip in flips:
            self._capture(flip)

    def _warm_up(self, flip):
        """One eager step on a side stream (PyTorch's capture recipe): AccumulateGrad nodes of dead graphs are gone after
        the collection, the new ones are created on a non-default stream."""
        gc.collect()
        model, opt = self.model, self.optimizer
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        model._flip_override = flip
        try:
            with torch.cuda.stream(s):
                opt.zero_grad(set_to_none=True)
                out = model(self.batch, progress=self.progress)
                out['loss'].backward()
                # NOT opt.step(): the warm-up must not change the training state
                del out
                opt.zero_grad(set_to_none=True)
        finally:
            model._flip_override = None
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gc.collect()

    def _capture(self, flip):
        model, opt = self.model, self.optimizer
        # every graph must contain its own weight re-pack launches: invalidate the packed copies first
        HF.bump_weight_epoch()
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        # Each graph gets its OWN memory pool: sharing one (torch's `pool=` idiom) is only safe when the graphs are replayed in
        # capture order, and here the flip draw picks the graph; 288 GB of HBM makes the second copy of the activations free.
        model._flip_override = flip
        try:
            with torch.cuda.graph(g):
                out = model(self.batch, progress=self.progress)
                out['loss'].backward()
                opt.step()
        finally:
            model._flip_override = None
        self.graphs[flip] = g
        self.loss[flip] = out['loss'].detach()
        HF.bump_weight_epoch()          # eager code after a replay must re-pack too

    def __call__(self, batch=None, flip=None):
        """One training step; returns the (device, static) loss tensor of the replayed graph.  `flip`: force the flip state
        (tests); default: drawn from Python's global RNG exactly like the reference's SfmModel."""
        if batch is not None:
            _copy_into(self.batch, batch)
        sync = getattr(self.optimizer, 'sync_hyperparams', None)
        if sync is not None:
            sync()                 # FlatAdam keeps lr & co. on the device: push what an LR scheduler changed since the last step
        draw = random.random() < self.flip_prob        # drawn every step, like the reference, so the RNG sequence matches
        if flip is not None:
            draw = bool(flip)
        flip = draw if len(self.graphs) > 1 else next(iter(self.graphs))
        self.graphs[flip].replay()
        HF.bump_weight_epoch()
        return self.loss[flip]
