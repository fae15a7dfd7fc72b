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

    @property
    def proc_rank(self):
        return hvd.rank()

    @property
    def world_size(self):
        return hvd.size()

    def fit(self, module):
        module.trainer = self
        module = module.to('cuda')
        module.configure_optimizers()
        optimizer = hvd.DistributedOptimizer(module.optimizer, named_parameters=module.named_parameters(),
                                             compression=hvd.Compression.none)
        scheduler = module.scheduler
        train_dataloader = module.train_dataloader()
        val_dataloaders = module.val_dataloader()
        if self.validate_first:
            self.check_and_save(module, self.validate(val_dataloaders, module))
        for _ in range(module.current_epoch, self.max_epochs):
            self.train(train_dataloader, module, optimizer)
            self.check_and_save(module, self.validate(val_dataloaders, module))
            module.current_epoch += 1
            scheduler.step()

    def train(self, dataloader, module, optimizer):
        module.train()
        if hasattr(dataloader.sampler, 'set_epoch'):
            dataloader.sampler.set_epoch(module.current_epoch)
        progress_bar = self.train_progress_bar(dataloader, module.config.datasets.train)
        outputs = []
        for i, batch in progress_bar:
            optimizer.zero_grad()
            batch = sample_to_cuda(batch)
            output = module.training_step(batch, i)
            output['loss'].backward()       # bucketed RCCL all-reduces start as soon as a bucket's gradients exist
            optimizer.step()                # joins the side stream, averages, then the optimizer update
            output['loss'] = output['loss'].detach()
            outputs.append(output)
            if self.is_rank_0 and hasattr(progress_bar, 'set_description') and i % self.log_every == 0:
                progress_bar.set_description('Epoch {} | Avg.Loss {:.4f}'.format(
                    module.current_epoch, self.avg_loss(output['loss'].item())))
        return module.training_epoch_end(outputs)

    def _run_eval(self, dataloaders, module, step_name, config, dtype=None):
        module.eval()
        all_outputs = []
        for n, dataloader in enumerate(dataloaders):
            bar = self.val_progress_bar(dataloader, config, n)
            outputs = []
            for i, batch in bar:
                batch = sample_to_cuda(batch, dtype)
                outputs.append(getattr(module, step_name)(batch, i, n))
            all_outputs.append(outputs)
        return all_outputs

    def validate(self, dataloaders, module):
        outputs = self._run_eval(dataloaders, module, 'validation_step', module.config.datasets.validation)
        return module.validation_epoch_end(outputs)

    def test(self, module):
        module = module.to('cuda', dtype=self.dtype)
        self.evaluate(module.test_dataloader(), module)

    @torch.no_grad()
    def evaluate(self, dataloaders, module):
        outputs = self._run_eval(dataloaders, module, 'test_step', module.config.datasets.test, self.dtype)
        return module.test_epoch_end(outputs)
