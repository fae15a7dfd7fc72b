"""HorovodTrainer: the reference's data-parallel training loop (packnet_sfm/trainers/horovod_trainer.py), on RCCL.

Same class name, constructor (`HorovodTrainer(**config.arch)`) and fit/train/validate/test protocol, so code that builds
the trainer from a config keeps working; `hvd` is the RCCL facade (packnet_sfm/rccl/hvd.py) -- one process per MI355X
launched with torch.distributed.run, gradients averaged by bucketed all-reduce overlapped with backward.

The `module` handed to fit() must expose what the reference's ModelWrapper does: configure_optimizers() -> sets
.optimizer/.scheduler, train_dataloader()/val_dataloader()/test_dataloader(), training_step(batch, i) -> {'loss': ...},
validation_step/test_step(batch, i, n), *_epoch_end(outputs), .current_epoch, .config.
"""
import os

import torch

from packnet_sfm.rccl import hvd
from packnet_sfm.trainers.base_trainer import BaseTrainer, sample_to_cuda


class _AvgMeter:
    """Running mean over the last `n` values (progress-bar loss)."""

    def __init__(self, n=50):
        self.n, self.values = n, []

    def __call__(self, value):
        self.values = (self.values + [value])[-self.n:]
        return sum(self.values) / len(self.values)


class HorovodTrainer(BaseTrainer):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        hvd.init()
        torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', 1)))
        if torch.cuda.is_available():
            torch.cuda.set_device(hvd.local_rank() % torch.cuda.device_count())
        self.avg_loss = _AvgMeter(50)
        self.dtype = kwargs.get('dtype', None)
        self.log_every = int(kwargs.get('log_every', 1))   # host sync for the loss readout every N steps (reference: 1)

    proc_rank = property(lambda self: hvd.rank())
    world_size = property(lambda self: hvd.size())

    # ---- training ----------------------------------------------------------------------------------------------------
    def fit(self, module):
        """Train `module` from its current epoch to `max_epochs`, validating (and checkpointing) after every epoch."""
        module.trainer = self
        module = module.to('cuda')
        module.configure_optimizers()
        # gradient averaging over RCCL: bucketed all-reduce on a side stream, overlapped with backward
        optimizer = hvd.DistributedOptimizer(module.optimizer, named_parameters=module.named_parameters(),
                                             compression=hvd.Compression.none)
        loaders = {'train': module.train_dataloader(), 'val': module.val_dataloader()}

        def validate_and_save():
            self.check_and_save(module, self.validate(loaders['val'], module))

        if self.validate_first:
            validate_and_save()
        while module.current_epoch < self.max_epochs:
            self.train(loaders['train'], module, optimizer)
            validate_and_save()
            module.current_epoch += 1
            module.scheduler.step()

    def _train_step(self, module, optimizer, batch, index):
        optimizer.zero_grad()
        output = module.training_step(sample_to_cuda(batch), index)
        output['loss'].backward()           # a bucket's all-reduce starts as soon as its last gradient exists
        optimizer.step()                    # joins the side streams, then the optimizer update
        output['loss'] = output['loss'].detach()
        return output

    def train(self, dataloader, module, optimizer):
        module.train()
        sampler = getattr(dataloader, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(module.current_epoch)
        bar = self.train_progress_bar(dataloader, module.config.datasets.train)
        show = self.is_rank_0 and hasattr(bar, 'set_description')
        outputs = []
        for i, batch in bar:
            outputs.append(self._train_step(module, optimizer, batch, i))
            if show and i % self.log_every == 0:
                mean = self.avg_loss(outputs[-1]['loss'].item())
                bar.set_description('Epoch {} | Avg.Loss {:.4f}'.format(module.current_epoch, mean))
        return module.training_epoch_end(outputs)

    # ---- evaluation --------------------------------------------------------------------------------------------------
    def _sweep(self, dataloaders, module, step, bar_config, dtype=None):
        """outputs[n][i] = module.<step>(batch i of dataloader n, i, n), in eval mode."""
        module.eval()
        return [[getattr(module, step)(sample_to_cuda(batch, dtype), i, n)
                 for i, batch in self.val_progress_bar(loader, bar_config, n)]
                for n, loader in enumerate(dataloaders)]

    def validate(self, dataloaders, module):
        return module.validation_epoch_end(
            self._sweep(dataloaders, module, 'validation_step', module.config.datasets.validation))

    def test(self, module):
        module = module.to('cuda', dtype=self.dtype)
        self.evaluate(module.test_dataloader(), module)

    @torch.no_grad()
    def evaluate(self, dataloaders, module):
        return module.test_epoch_end(self._sweep(dataloaders, module, 'test_step', module.config.datasets.test, self.dtype))


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
