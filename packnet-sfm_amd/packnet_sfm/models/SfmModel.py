"""SfmModel: a depth network and a pose network behind one `forward(batch)` (contract of the reference's
packnet_sfm/models/SfmModel.py: `add_depth_net`, `add_pose_net`, `compute_depth_net`, `compute_pose_net`,
`depth_net_flipping`, output dict {'inv_depths', 'poses'})."""
import random

import torch

from packnet_sfm.geometry.pose import Pose
from packnet_sfm.models.base_model import BaseModel
from packnet_sfm.models.model_utils import flip_batch_input, flip_output, upsample_output
from packnet_sfm.utils.misc import filter_dict


def _branch_stream(t):
    from packnet_sfm.hip.functional import branch_stream
    return branch_stream(t)


class SfmModel(BaseModel):
    """
    depth_net / pose_net : nn.Module   (usually attached later through add_depth_net / add_pose_net)
    rotation_mode : str                pose-vector rotation parametrisation ('euler')
    flip_lr_prob : float               probability of running the depth network on a mirrored batch (training only)
    upsample_depth_maps : bool         nearest-upsample all predicted scales to full resolution (training only)
    """

    def __init__(self, depth_net=None, pose_net=None, rotation_mode='euler', flip_lr_prob=0.0,
                 upsample_depth_maps=False, **kwargs):
        super().__init__()
        self._network_requirements.extend(('depth_net', 'pose_net'))
        self.depth_net, self.pose_net = depth_net, pose_net
        self.rotation_mode, self.flip_lr_prob = rotation_mode, flip_lr_prob
        self.upsample_depth_maps = upsample_depth_maps

    def add_depth_net(self, depth_net):
        self.depth_net = depth_net

    def add_pose_net(self, pose_net):
        self.pose_net = pose_net

    # ---- depth -------------------------------------------------------------------------------------------------------
    def depth_net_flipping(self, batch, flip):
        """The depth network on the batch -- or, when `flip`, on its mirror image with the prediction mirrored back."""
        net_input = {key: batch[key] for key in filter_dict(batch, self._input_keys)}
        if flip:
            return flip_output(self.depth_net(**flip_batch_input(net_input)))
        return self.depth_net(**net_input)

    # set by tests that need an explicit flip state (Python's global RNG is shared with the rest of the process);
    # None = draw it here as the reference does (SfmModel.py:84)
    _flip_override = None

    def _draw_flip(self, force_flip):
        # ONE python-RNG draw per training batch (replicas agree because their seeds agree); evaluation never draws
        if not self.training:
            return force_flip
        if self._flip_override is not None:
            return bool(self._flip_override)
        return random.random() < self.flip_lr_prob

    def compute_depth_net(self, batch, force_flip=False):
        output = self.depth_net_flipping(batch, self._draw_flip(force_flip))
        if self.training and self.upsample_depth_maps:
            output = upsample_output(output, mode='nearest', align_corners=None)
        return output

    # ---- pose --------------------------------------------------------------------------------------------------------
    def compute_pose_net(self, image, contexts):
        """[B,J,6] pose vectors -> one Pose (target -> context j) per context image."""
        vectors = self.pose_net(image, contexts)
        return [Pose.from_vec(v, self.rotation_mode) for v in vectors.unbind(1)]

    def forward(self, batch, return_logs=False, force_flip=False):
        has_contexts = 'rgb_context' in batch and self.pose_net is not None
        side = _branch_stream(batch['rgb']) if has_contexts else None
        if side is not None:
            main = torch.cuda.current_stream(batch['rgb'].device)
            inputs_ready = main.record_event()           # the pose network only needs the batch, not the depth network
        output = dict(self.compute_depth_net(batch, force_flip=force_flip))
        if not has_contexts:
            output['poses'] = None
        elif side is None:
            output['poses'] = self.compute_pose_net(batch['rgb'], batch['rgb_context'])
        else:
            # The pose network on the second compute stream (hip/functional.py: branch_stream).  It is enqueued AFTER the depth
            # network on purpose: autograd runs the nodes created last first, so its backward pass starts together with the
            # depth decoder's, and in the forward pass the host runs far enough ahead of the device for it to overlap the
            # depth network's tail.  Same kernels, same order within each stream: results are bit-identical either way.
            side.wait_event(inputs_ready)
            with torch.cuda.stream(side):
                poses = self.compute_pose_net(batch['rgb'], batch['rgb_context'])
            for t in [batch['rgb']] + list(batch['rgb_context']):
                t.record_stream(side)                    # allocated on the compute stream, read on the side stream
            main.wait_stream(side)
            for pose in poses:
                pose.mat.record_stream(main)             # and the other way round
            output['poses'] = poses
        return output


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
