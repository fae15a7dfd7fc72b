"""SfmModel: depth network + pose network (API of the reference's packnet_sfm/models/SfmModel.py)."""
import random

from packnet_sfm.geometry.pose import Pose
from packnet_sfm.models.base_model import BaseModel
from packnet_sfm.models.model_utils import flip_batch_input, flip_output, upsample_output
from packnet_sfm.utils.misc import filter_dict


class SfmModel(BaseModel):
    """
    depth_net / pose_net : nn.Module
    rotation_mode : str            pose-vector rotation parametrisation ('euler')
    flip_lr_prob : float           probability of running the depth network on a mirrored batch (training)
    upsample_depth_maps : bool     nearest-upsample all predicted scales to full resolution (training)
    """

    def __init__(self, depth_net=None, pose_net=None, rotation_mode='euler', flip_lr_prob=0.0,
                 upsample_depth_maps=False, **kwargs):
        super().__init__()
        self.depth_net = depth_net
        self.pose_net = pose_net
        self.rotation_mode = rotation_mode
        self.flip_lr_prob = flip_lr_prob
        self.upsample_depth_maps = upsample_depth_maps
        self._network_requirements = ['depth_net', 'pose_net']

    def add_depth_net(self, depth_net):
        self.depth_net = depth_net

    def add_pose_net(self, pose_net):
        self.pose_net = pose_net

    def depth_net_flipping(self, batch, flip):
        """Depth network on the batch, or on its mirror image with the prediction mirrored back."""
        net_input = {key: batch[key] for key in filter_dict(batch, self._input_keys)}
        if not flip:
            return self.depth_net(**net_input)
        return flip_output(self.depth_net(**flip_batch_input(net_input)))

    def compute_depth_net(self, batch, force_flip=False):
        # one python-RNG draw per batch, exactly like the reference (replicas agree because seeds agree)
        flip = random.random() < self.flip_lr_prob if self.training else force_flip
        output = self.depth_net_flipping(batch, flip)
        if self.training and self.upsample_depth_maps:
            output = upsample_output(output, mode='nearest', align_corners=None)
        return output

    def compute_pose_net(self, image, contexts):
        pose_vec = self.pose_net(image, contexts)
        return [Pose.from_vec(pose_vec[:, i], self.rotation_mode) for i in range(pose_vec.shape[1])]

    def forward(self, batch, return_logs=False, force_flip=False):
        depth_output = self.compute_depth_net(batch, force_flip=force_flip)
        pose_output = None
        if 'rgb_context' in batch and self.pose_net is not None:
            pose_output = self.compute_pose_net(batch['rgb'], batch['rgb_context'])
        return {**depth_output, 'poses': pose_output}
