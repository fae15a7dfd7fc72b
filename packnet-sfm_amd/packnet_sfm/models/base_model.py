"""BaseModel: the plugin contract `setup_model` / the trainers rely on (reference: packnet_sfm/models/base_model.py) --
`network_requirements`, `train_requirements`, `add_net`, `add_loss`, `logs`, `losses`, `forward(batch, ...)`."""
import torch.nn as nn

from packnet_sfm.utils.reporting import Reporting


class BaseModel(Reporting, nn.Module):
    def __init__(self, **kwargs):
        nn.Module.__init__(self)
        # subclasses edit these lists in place (SfmModel adds its two networks, SemiSupModel drops the pose network)
        self._network_requirements, self._train_requirements = [], []
        self._input_keys = ['rgb']                      # batch entries forwarded to the depth network

    logs = property(lambda self: self._report('logs'))
    losses = property(lambda self: self._report('losses'))
    network_requirements = property(lambda self: self._network_requirements, doc="networks the model needs, by attribute name")
    train_requirements = property(lambda self: self._train_requirements, doc="ground truth needed at training time")

    def add_loss(self, key, val):
        self._record('losses', key, val)

    def add_net(self, network_module, network_name):
        if network_name not in self._network_requirements:
            raise AssertionError("Network module not required!")
        setattr(self, network_name, network_module)

    def forward(self, batch, return_logs=False, **kwargs):
        raise NotImplementedError("Please implement forward function in your own subclass model.")


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
