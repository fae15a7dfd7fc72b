"""Trainer base: device transfer of a batch and the progress/checkpoint hooks (API of the reference's
packnet_sfm/trainers/base_trainer.py)."""
import torch

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    tqdm = None


def sample_to_cuda(data, dtype=None):
    """Recursively move a batch (dict / list / tensor; strings pass through) to the current HIP device.
    Only floating-point tensors are cast to `dtype`."""
    if isinstance(data, str):
        return data
    if isinstance(data, dict):
        return {key: sample_to_cuda(val, dtype) for key, val in data.items()}
    if isinstance(data, (list, tuple)):
        return [sample_to_cuda(val, dtype) for val in data]
    return data.to('cuda', dtype=dtype if torch.is_floating_point(data) else None, non_blocking=True)


class BaseTrainer:
    """Epoch bounds, checkpoint hook and rank-aware progress bars; subclasses provide `proc_rank` / `world_size`."""

    def __init__(self, min_epochs=0, max_epochs=50, validate_first=False, checkpoint=None, **kwargs):
        self.min_epochs, self.max_epochs = min_epochs, max_epochs
        self.validate_first = validate_first
        self.checkpoint = checkpoint
        self.module = None

    def _abstract(self):
        raise NotImplementedError('Not implemented for BaseTrainer')

    proc_rank = property(_abstract)
    world_size = property(_abstract)
    is_rank_0 = property(lambda self: self.proc_rank == 0)

    def check_and_save(self, module, output):
        if self.checkpoint:
            self.checkpoint.check_and_save(module, output)

    def _bar(self, dataloader, config, desc=None, ncols=120):
        """enumerate(dataloader), wrapped in a tqdm bar (images/s over all ranks) on rank 0 when tqdm is installed."""
        steps = enumerate(dataloader, 0)
        if tqdm is None:
            return steps
        per_rank = getattr(config, 'batch_size', 1)
        if isinstance(per_rank, (list, tuple)):
            per_rank = per_rank[0]
        return tqdm(steps, total=len(dataloader), unit=' images', unit_scale=self.world_size * per_rank, smoothing=0,
                    disable=not self.is_rank_0, ncols=ncols, desc=desc)

    def train_progress_bar(self, dataloader, config, ncols=120):
        return self._bar(dataloader, config, ncols=ncols)

    def val_progress_bar(self, dataloader, config, n=0, ncols=120):
        return self._bar(dataloader, config, desc='val[%d]' % n, ncols=ncols)

    def test_progress_bar(self, dataloader, config, n=0, ncols=120):
        return self._bar(dataloader, config, desc='test[%d]' % n, ncols=ncols)


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
