"""BaseModel: the model-plugin contract the wrapper/trainer relies on (API of the reference's
packnet_sfm/models/base_model.py)."""
import torch.nn as nn


class BaseModel(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        self._logs = {}
        self._losses = {}
        self._network_requirements = []     # networks the model needs ('depth_net', 'pose_net', ...)
        self._train_requirements = []       # ground truth needed at training time
        self._input_keys = ['rgb']          # batch keys handed to the depth network

    def _forward_unimplemented(self, *args):
        pass

    @property
    def logs(self):
        return self._logs

    @property
    def losses(self):
        return self._losses

    def add_loss(self, key, val):
        self._losses[key] = val.detach()

    @property
    def network_requirements(self):
        return self._network_requirements

    @property
    def train_requirements(self):
        return self._train_requirements

    def add_net(self, network_module, network_name):
        assert network_name in self._network_requirements, "Network module not required!"
        setattr(self, network_name, network_module)

    def forward(self, batch, return_logs=False, **kwargs):
        raise NotImplementedError("Please implement forward function in your own subclass model.")
