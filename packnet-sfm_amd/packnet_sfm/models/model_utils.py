"""Batch / output plumbing around the networks (API of the reference's packnet_sfm/models/model_utils.py)."""
from packnet_sfm.utils.image import flip_lr, interpolate_scales
from packnet_sfm.utils.misc import filter_dict
from packnet_sfm.utils.types import is_list, is_numpy, is_tensor

_FLIP_INPUT_KEYS = ['rgb', 'rgb_context', 'input_depth', 'input_depth_context']
_FLIP_OUTPUT_KEYS = [
    'uncertainty', 'logits_semantic', 'ord_probability',
    'inv_depths', 'inv_depths_context', 'inv_depths1', 'inv_depths2',
    'pred_depth', 'pred_depth_context', 'pred_depth1', 'pred_depth2',
    'pred_inv_depth', 'pred_inv_depth_context', 'pred_inv_depth1', 'pred_inv_depth2',
]


def flip(tensor, flip_fn):
    """Apply flip_fn to a tensor, a list of tensors or a list of lists of tensors."""
    if not is_list(tensor):
        return flip_fn(tensor)
    if not is_list(tensor[0]):
        return [flip_fn(t) for t in tensor]
    return [[flip_fn(t) for t in ts] for ts in tensor]


def merge_outputs(*outputs):
    """Merge output dicts: 'metrics' sub-dicts are combined, 'loss' is dropped, other keys must be unique."""
    merged = {'metrics': {}}
    for output in outputs:
        for key, val in output.items():
            if key == 'metrics':
                for sub_key, sub_val in val.items():
                    assert sub_key not in merged['metrics'], 'Combining duplicated key {} to {}'.format(sub_key, key)
                    merged['metrics'][sub_key] = sub_val
            elif key != 'loss':
                assert key not in merged, 'Adding duplicated key {}'.format(key)
                merged[key] = val
    return merged


def stack_batch(batch):
    """[1,N,C,H,W] multi-camera batches become [N,C,H,W]."""
    if len(batch['rgb'].shape) == 5:
        assert batch['rgb'].shape[0] == 1, 'Only batch size 1 is supported for multi-cameras'
        for key in batch.keys():
            if is_list(batch[key]):
                if is_tensor(batch[key][0]) or is_numpy(batch[key][0]):
                    batch[key] = [sample[0] for sample in batch[key]]
            else:
                batch[key] = batch[key][0]
    return batch


def flip_batch_input(batch):
    """Left-right flip of the image-like inputs (and of the principal point if intrinsics are present)."""
    for key in filter_dict(batch, _FLIP_INPUT_KEYS):
        batch[key] = flip(batch[key], flip_lr)
    for key in filter_dict(batch, ['intrinsics']):
        batch[key] = batch[key].clone()
        batch[key][:, 0, 2] = batch['rgb'].shape[3] - batch[key][:, 0, 2]
    return batch


def flip_output(output):
    for key in filter_dict(output, _FLIP_OUTPUT_KEYS):
        output[key] = flip(output[key], flip_lr)
    return output


def upsample_output(output, mode='nearest', align_corners=None):
    """Bring every scale of the multi-scale outputs to the resolution of the first one."""
    for key in filter_dict(output, ['inv_depths', 'uncertainty']):
        output[key] = interpolate_scales(output[key], mode=mode, align_corners=align_corners)
    for key in filter_dict(output, ['inv_depths_context']):
        output[key] = [interpolate_scales(val, mode=mode, align_corners=align_corners) for val in output[key]]
    return output


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
