"""Depth-map helpers on the training path + the evaluation metrics (API subset of the reference's
packnet_sfm/utils/depth.py).  inv2depth on the loss path is fused into the view-synthesis kernel."""
import torch

from packnet_sfm.utils.types import is_seq


def inv2depth(inv_depth):
    if is_seq(inv_depth):
        return [inv2depth(item) for item in inv_depth]
    return 1. / inv_depth.clamp(min=1e-6)


def depth2inv(depth):
    if is_seq(depth):
        return [depth2inv(item) for item in depth]
    inv_depth = 1. / depth.clamp(min=1e-6)
    inv_depth[depth <= 0.] = 0.
    return inv_depth


def inv_depths_normalize(inv_depths):
    """Divide each map by its per-sample spatial mean (clamped at 1e-6)."""
    means = [d.mean(2, True).mean(3, True) for d in inv_depths]
    return [d / m.clamp(min=1e-6) for d, m in zip(inv_depths, means)]


# Evaluation crop of Garg et al. (fractions of the image height / width), used by the KITTI protocol when config.crop == 'garg'
_GARG_ROWS, _GARG_COLS = (0.40810811, 0.99189189), (0.03594771, 0.96405229)

# name -> f(gt, pred) on the 1-D vectors of valid pixels; order = the reference's metric vector
_DEPTH_METRICS = (
    ('abs_rel', lambda g, p: ((g - p).abs() / g).mean()),
    ('sqr_rel', lambda g, p: ((g - p) ** 2 / g).mean()),
    ('rmse', lambda g, p: ((g - p) ** 2).mean().sqrt()),
    ('rmse_log', lambda g, p: ((g.log() - p.log()) ** 2).mean().sqrt()),
    ('a1', lambda g, p: (torch.maximum(g / p, p / g) < 1.25).float().mean()),
    ('a2', lambda g, p: (torch.maximum(g / p, p / g) < 1.25 ** 2).float().mean()),
    ('a3', lambda g, p: (torch.maximum(g / p, p / g) < 1.25 ** 3).float().mean()),
)


def _to_gt_resolution(pred, gt, how):
    """'resize': bilinear (align_corners) to the ground-truth size; 'top-center': paste into a zero map, flush with the
    bottom edge and centred horizontally (predictions made on a top-cropped image)."""
    import torch.nn.functional as funct
    if how == 'resize':
        if tuple(pred.shape[-2:]) == tuple(gt.shape[-2:]):
            return pred
        return funct.interpolate(pred, size=gt.shape[-2:], mode='bilinear', align_corners=True)
    if how != 'top-center':
        raise NotImplementedError('Depth scale function {} not implemented.'.format(how))
    canvas = pred.new_zeros(gt.shape)
    dh, dw = gt.shape[2] - pred.shape[2], (gt.shape[3] - pred.shape[3]) // 2
    canvas[:, :, dh:dh + pred.shape[2], dw:dw + pred.shape[3]] = pred
    return canvas


def compute_depth_metrics(config, gt, pred, use_gt_scale=True):
    """[abs_rel, sqr_rel, rmse, rmse_log, a1, a2, a3], summed over the images that have valid pixels and divided by the
    batch size (the reference's convention, utils/depth.py:258-324).  config: .min_depth, .max_depth, .crop ('' | 'garg')
    and optionally .scale_output ('resize' | 'top-center'); gt, pred: [B,1,H,W] depth maps."""
    B, _, H, W = gt.shape
    pred = _to_gt_resolution(pred, gt, getattr(config, 'scale_output', 'resize'))
    inside = torch.ones((H, W), dtype=torch.bool, device=gt.device)
    if config.crop == 'garg':
        inside.zero_()
        inside[int(_GARG_ROWS[0] * H):int(_GARG_ROWS[1] * H), int(_GARG_COLS[0] * W):int(_GARG_COLS[1] * W)] = True
    totals = torch.zeros(len(_DEPTH_METRICS), dtype=torch.float64)
    for g_img, p_img in zip(gt[:, 0], pred[:, 0]):
        keep = inside & (g_img > config.min_depth) & (g_img < config.max_depth)
        if not bool(keep.any()):
            continue
        g, p = g_img[keep], p_img[keep]
        if use_gt_scale:                                  # median scaling of the (scale-ambiguous) prediction
            p = p * (g.median() / p.median())
        p = p.clamp(config.min_depth, config.max_depth)
        totals += torch.stack([fn(g, p) for _, fn in _DEPTH_METRICS]).double().cpu()
    return (totals / B).type_as(gt)


# names of the reference's module of the same path that the hot path does not re-implement (packnet_sfm/_merge.py)
from packnet_sfm._merge import reference_fallback as _reference_fallback  # noqa: E402
__getattr__ = _reference_fallback(__name__, __file__)
