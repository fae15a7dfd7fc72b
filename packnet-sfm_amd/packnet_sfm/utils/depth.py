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


def compute_depth_metrics(config, gt, pred, use_gt_scale=True):
    """abs_rel, sqr_rel, rmse, rmse_log, a1, a2, a3 averaged over the batch (reference: utils/depth.py:258-324).
    config needs .crop ('' | 'garg'), .min_depth, .max_depth; gt, pred: [B,1,H,W] (pred is resized to gt)."""
    import torch.nn.functional as funct
    crop = config.crop == 'garg'
    batch_size, _, gt_height, gt_width = gt.shape
    abs_diff = abs_rel = sq_rel = rmse = rmse_log = a1 = a2 = a3 = 0.0
    pred = funct.interpolate(pred, gt.shape[-2:], mode='bilinear', align_corners=True)
    if crop:
        crop_mask = torch.zeros(gt.shape[-2:], dtype=torch.bool, device=gt.device)
        y1, y2 = int(0.40810811 * gt_height), int(0.99189189 * gt_height)
        x1, x2 = int(0.03594771 * gt_width), int(0.96405229 * gt_width)
        crop_mask[y1:y2, x1:x2] = True
    for pred_i, gt_i in zip(pred, gt):
        gt_i, pred_i = torch.squeeze(gt_i), torch.squeeze(pred_i)
        valid = (gt_i > config.min_depth) & (gt_i < config.max_depth)
        valid = valid & crop_mask if crop else valid
        gt_i, pred_i = gt_i[valid], pred_i[valid]
        if use_gt_scale:
            pred_i = pred_i * torch.median(gt_i) / torch.median(pred_i)
        pred_i = pred_i.clamp(config.min_depth, config.max_depth)
        thresh = torch.max((gt_i / pred_i), (pred_i / gt_i))
        a1 += (thresh < 1.25).float().mean()
        a2 += (thresh < 1.25 ** 2).float().mean()
        a3 += (thresh < 1.25 ** 3).float().mean()
        diff_i = gt_i - pred_i
        abs_diff += torch.mean(torch.abs(diff_i))
        abs_rel += torch.mean(torch.abs(diff_i) / gt_i)
        sq_rel += torch.mean(diff_i ** 2 / gt_i)
        rmse += torch.sqrt(torch.mean(diff_i ** 2))
        rmse_log += torch.sqrt(torch.mean((torch.log(gt_i) - torch.log(pred_i)) ** 2))
    return torch.tensor([a / batch_size for a in [abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3]]).type_as(gt)
