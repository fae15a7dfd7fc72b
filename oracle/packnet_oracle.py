"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path under packnet-sfm_amd/).

A plain PyTorch fp32, functional restatement of the PackNet-SfM hot path (PackNet01 depth network, PoseNet,
pose algebra, multi-view photometric loss), written from the reference's math with every function citing the
reference lines it restates (paths relative to /root/reference).  It runs on CPU (or on any torch device) and is
the checker for the HIP kernels: forward values directly, gradients through torch.autograd.

Parity pinning: the reference ships no tests / golden vectors.  This oracle is pinned
instead against the reference's own modules imported from /root/reference in this container
(oracle/pin_against_reference.py), and the resulting input/output vectors are committed under tests/golden/.

All network functions take a *state dict with the reference's key names* (e.g. 'pack1.conv3d.weight',
'conv2.0.conv1.conv_base.weight'), which is the checkpoint contract of packnet_sfm/utils/load.py:114-163.
"""
import math

import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------------------------
# PackNet01 building blocks  (packnet_sfm/networks/layers/packnet/layers01.py)
# ------------------------------------------------------------------------------------------------------------------


def conv2d_gn_elu(x, sd, p, k):
    """Conv2D: zero-pad k//2, conv (stride 1), GroupNorm(16), ELU.  layers01.py:10-37"""
    y = F.conv2d(F.pad(x, [k // 2] * 4), sd[p + '.conv_base.weight'], sd[p + '.conv_base.bias'])
    y = F.group_norm(y, 16, sd[p + '.normalize.weight'], sd[p + '.normalize.bias'], eps=1e-5)
    return F.elu(y)


def residual_conv(x, sd, p):
    """ResidualConv: ELU(GN(Conv2D(Conv2D(x)) + conv1x1(x))), dropout off.  layers01.py:40-72"""
    out = conv2d_gn_elu(x, sd, p + '.conv1', 3)
    out = conv2d_gn_elu(out, sd, p + '.conv2', 3)
    shortcut = F.conv2d(x, sd[p + '.conv3.weight'], sd[p + '.conv3.bias'])
    y = F.group_norm(out + shortcut, 16, sd[p + '.normalize.weight'], sd[p + '.normalize.bias'], eps=1e-5)
    return F.elu(y)


def residual_block(x, sd, p, num_blocks):
    """ResidualBlock: sequence of ResidualConv.  layers01.py:75-95"""
    for i in range(num_blocks):
        x = residual_conv(x, sd, '%s.%d' % (p, i))
    return x


def inv_depth_head(x, sd, p, min_depth=0.5):
    """InvDepth: zero-pad 1, 3x3 conv C->1, sigmoid / min_depth.  layers01.py:98-122"""
    return torch.sigmoid(F.conv2d(F.pad(x, [1] * 4), sd[p + '.conv1.weight'], sd[p + '.conv1.bias'])) / min_depth


def packing(x, r=2):
    """Space-to-depth: out[b, c*r*r + i*r + j, h, w] = x[b, c, h*r+i, w*r+j].  layers01.py:126-148"""
    b, c, h, w = x.shape
    x = x.reshape(b, c, h // r, r, w // r, r)
    return x.permute(0, 1, 3, 5, 2, 4).reshape(b, c * r * r, h // r, w // r)


def conv3d_1to8(x, w3, b3):
    """Conv3d(1, 8, 3, padding 1) over (channel, y, x), viewed back to [B, 8*D, H, W].  layers01.py:241-245"""
    y = F.conv3d(x.unsqueeze(1), w3, b3, padding=1)
    b, f, d, h, w = y.shape
    return y.reshape(b, f * d, h, w)


def pack_layer_conv3d(x, sd, p, k):
    """PackLayerConv3d: packing -> Conv3d(1->8) -> view -> Conv2D(k).  layers01.py:213-247"""
    y = conv3d_1to8(packing(x), sd[p + '.conv3d.weight'], sd[p + '.conv3d.bias'])
    return conv2d_gn_elu(y, sd, p + '.conv', k)


def compose_pack_weight(W2, W3):
    """Algebra behind the MI355X collapsed packing block (not a reference function):
    conv2d(conv3d_1to8(x)) has no non-linearity in between (layers01.py:243-246), so in the image interior it equals
    one (k+2)x(k+2) conv over the D packed channels with
        W_eff[co, ci, U, V] = sum_{f,dz,dy,dx} W3[f,0,dz,dy,dx] * W2[co, f*D + (ci-dz+1), U-dy, V-dx]."""
    C, DF, k, _ = W2.shape
    NF = W3.shape[0]                      # 3-D feature maps: 8 (PackNet01) or 4 (PackNetSlim01)
    D = DF // NF
    W2v = W2.reshape(C, NF, D, k, k)
    Weff = W2.new_zeros(C, D, k + 2, k + 2)
    for dz in range(3):
        lo, hi = max(0, dz - 1), min(D, D + dz - 1)          # ci range with d = ci-dz+1 in [0, D)
        for dy in range(3):
            for dx in range(3):
                w = W3[:, 0, dz, dy, dx].view(1, NF, 1, 1, 1)
                contrib = (W2v[:, :, lo - dz + 1:hi - dz + 1] * w).sum(1)          # [C, hi-lo, k, k]
                Weff[:, lo:hi, dy:dy + k, dx:dx + k] += contrib
    return Weff


def unpack_layer_conv3d(x, sd, p, k):
    """UnpackLayerConv3d: Conv2D(k) -> Conv3d(1->8) -> view -> PixelShuffle(2).  layers01.py:250-286"""
    y = conv2d_gn_elu(x, sd, p + '.conv', k)
    y = conv3d_1to8(y, sd[p + '.conv3d.weight'], sd[p + '.conv3d.bias'])
    return F.pixel_shuffle(y, 2)


def packnet01_forward(sd, rgb, version='1A', training=True):
    """PackNet01.forward.  packnet_sfm/networks/depth/PackNet01.py:106-185
    Returns [disp1, disp2, disp3, disp4] when training else disp1 (a tensor, :182-185)."""
    cat = version[1:] == 'A'
    x = conv2d_gn_elu(rgb, sd, 'pre_calc', 5)
    x1 = conv2d_gn_elu(x, sd, 'conv1', 7)
    x1p = pack_layer_conv3d(x1, sd, 'pack1', 5)
    x2 = residual_block(x1p, sd, 'conv2', 2)
    x2p = pack_layer_conv3d(x2, sd, 'pack2', 3)
    x3 = residual_block(x2p, sd, 'conv3', 2)
    x3p = pack_layer_conv3d(x3, sd, 'pack3', 3)
    x4 = residual_block(x3p, sd, 'conv4', 3)
    x4p = pack_layer_conv3d(x4, sd, 'pack4', 3)
    x5 = residual_block(x4p, sd, 'conv5', 3)
    x5p = pack_layer_conv3d(x5, sd, 'pack5', 3)
    skip1, skip2, skip3, skip4, skip5 = x, x1p, x2p, x3p, x4p

    def up2(t):  # nn.Upsample(scale_factor=2, mode='nearest')  PackNet01.py:87-89
        return F.interpolate(t, scale_factor=2, mode='nearest')

    unpack5 = unpack_layer_conv3d(x5p, sd, 'unpack5', 3)
    iconv5 = conv2d_gn_elu(torch.cat((unpack5, skip5), 1) if cat else unpack5 + skip5, sd, 'iconv5', 3)
    unpack4 = unpack_layer_conv3d(iconv5, sd, 'unpack4', 3)
    iconv4 = conv2d_gn_elu(torch.cat((unpack4, skip4), 1) if cat else unpack4 + skip4, sd, 'iconv4', 3)
    disp4 = inv_depth_head(iconv4, sd, 'disp4_layer')
    unpack3 = unpack_layer_conv3d(iconv4, sd, 'unpack3', 3)
    c3 = torch.cat((unpack3, skip3, up2(disp4)), 1) if cat else torch.cat((unpack3 + skip3, up2(disp4)), 1)
    iconv3 = conv2d_gn_elu(c3, sd, 'iconv3', 3)
    disp3 = inv_depth_head(iconv3, sd, 'disp3_layer')
    unpack2 = unpack_layer_conv3d(iconv3, sd, 'unpack2', 3)
    c2 = torch.cat((unpack2, skip2, up2(disp3)), 1) if cat else torch.cat((unpack2 + skip2, up2(disp3)), 1)
    iconv2 = conv2d_gn_elu(c2, sd, 'iconv2', 3)
    disp2 = inv_depth_head(iconv2, sd, 'disp2_layer')
    unpack1 = unpack_layer_conv3d(iconv2, sd, 'unpack1', 3)
    c1 = torch.cat((unpack1, skip1, up2(disp2)), 1) if cat else torch.cat((unpack1 + skip1, up2(disp2)), 1)
    iconv1 = conv2d_gn_elu(c1, sd, 'iconv1', 3)
    disp1 = inv_depth_head(iconv1, sd, 'disp1_layer')
    return [disp1, disp2, disp3, disp4] if training else disp1


# ------------------------------------------------------------------------------------------------------------------
# PoseNet + pose algebra
# ------------------------------------------------------------------------------------------------------------------


def posenet_forward(sd, image, contexts):
    """PoseNet.forward: cat -> 7x (stride-2 conv, GN16, ReLU) -> 1x1 conv -> spatial mean -> *0.01 -> [B,2,6].
    packnet_sfm/networks/pose/PoseNet.py:11-34,67-84"""
    x = torch.cat([image] + list(contexts), 1)
    ks = [7, 5, 3, 3, 3, 3, 3]
    for i, k in enumerate(ks):
        p = 'conv%d' % (i + 1)
        x = F.conv2d(x, sd[p + '.0.weight'], sd[p + '.0.bias'], stride=2, padding=(k - 1) // 2)
        x = F.relu(F.group_norm(x, 16, sd[p + '.1.weight'], sd[p + '.1.bias'], eps=1e-5))
    pose = F.conv2d(x, sd['pose_pred.weight'], sd['pose_pred.bias'])
    pose = pose.mean(3).mean(2)
    return 0.01 * pose.view(pose.size(0), len(contexts), 6)


def euler2mat(angle):
    """R = Rx(x) @ Ry(y) @ Rz(z).  packnet_sfm/geometry/pose_utils.py:8-37"""
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    zeros, ones = torch.zeros_like(z), torch.ones_like(z)
    cz, sz, cy, sy, cx, sx = torch.cos(z), torch.sin(z), torch.cos(y), torch.sin(y), torch.cos(x), torch.sin(x)
    zmat = torch.stack([cz, -sz, zeros, sz, cz, zeros, zeros, zeros, ones], 1).view(-1, 3, 3)
    ymat = torch.stack([cy, zeros, sy, zeros, ones, zeros, -sy, zeros, cy], 1).view(-1, 3, 3)
    xmat = torch.stack([ones, zeros, zeros, zeros, cx, -sx, zeros, sx, cx], 1).view(-1, 3, 3)
    return xmat.bmm(ymat).bmm(zmat)


def pose_vec2mat44(vec):
    """[B,6] (tx,ty,tz,rx,ry,rz) -> [B,4,4].  pose_utils.py:41-52 + geometry/pose.py:40-46"""
    B = vec.shape[0]
    top = torch.cat([euler2mat(vec[:, 3:]), vec[:, :3].unsqueeze(-1)], 2)
    bottom = torch.tensor([0., 0., 0., 1.], dtype=vec.dtype, device=vec.device).view(1, 1, 4).repeat(B, 1, 1)
    return torch.cat([top, bottom], 1)


# ------------------------------------------------------------------------------------------------------------------
# Camera geometry + view synthesis
# ------------------------------------------------------------------------------------------------------------------


def scale_intrinsics(K, x_scale, y_scale):
    """geometry/camera_utils.py:16-22"""
    K = K.clone()
    K[..., 0, 0] *= x_scale
    K[..., 1, 1] *= y_scale
    K[..., 0, 2] = (K[..., 0, 2] + 0.5) * x_scale - 0.5
    K[..., 1, 2] = (K[..., 1, 2] + 0.5) * y_scale - 0.5
    return K


def k_inverse(K):
    """Camera.Kinv: K with fx,fy,cx,cy entries replaced by the analytic inverse.  geometry/camera.py:72-80"""
    Kinv = K.clone()
    Kinv[:, 0, 0] = 1. / K[:, 0, 0]
    Kinv[:, 1, 1] = 1. / K[:, 1, 1]
    Kinv[:, 0, 2] = -1. * K[:, 0, 2] / K[:, 0, 0]
    Kinv[:, 1, 2] = -1. * K[:, 1, 2] / K[:, 1, 1]
    return Kinv


def reconstruct(depth, K):
    """Camera.reconstruct with identity pose: X = (Kinv [u,v,1]^T) * depth.  camera.py:112-148, utils/image.py:218-282"""
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.linspace(0, H - 1, H, dtype=depth.dtype, device=depth.device),
                            torch.linspace(0, W - 1, W, dtype=depth.dtype, device=depth.device), indexing='ij')
    grid = torch.stack([xs, ys, torch.ones_like(xs)], 0).view(1, 3, -1).repeat(B, 1, 1)
    return (k_inverse(K).bmm(grid)).view(B, 3, H, W) * depth


def project(X, K, T):
    """Camera.project in the 'w' frame with Tcw = T: normalised sampling grid [B,H,W,2].  camera.py:150-191"""
    B, _, H, W = X.shape
    Xc = T[:, :3, :3].bmm(X.view(B, 3, -1)) + T[:, :3, -1].unsqueeze(-1)   # geometry/pose.py:80-86
    Xc = K.bmm(Xc)
    Z = Xc[:, 2].clamp(min=1e-5)
    Xn = 2 * (Xc[:, 0] / Z) / (W - 1) - 1.
    Yn = 2 * (Xc[:, 1] / Z) / (H - 1) - 1.
    return torch.stack([Xn, Yn], dim=-1).view(B, H, W, 2)


def inv2depth(inv_depth):
    """utils/depth.py:103-120"""
    return 1. / inv_depth.clamp(min=1e-6)


def view_synthesis(ref_image, inv_depth, K, ref_K, T, padding_mode='zeros'):
    """inv2depth -> reconstruct -> project -> grid_sample(bilinear, align_corners=True).
    geometry/camera_utils.py:27-59, losses/multiview_photometric_loss.py:159-163"""
    X = reconstruct(inv2depth(inv_depth), K)
    grid = project(X, ref_K, T)
    return F.grid_sample(ref_image, grid, mode='bilinear', padding_mode=padding_mode, align_corners=True)


# ------------------------------------------------------------------------------------------------------------------
# Photometric loss
# ------------------------------------------------------------------------------------------------------------------


def ssim(x, y, C1=1e-4, C2=9e-4):
    """losses/multiview_photometric_loss.py:14-53"""
    x, y = F.pad(x, [1] * 4, mode='reflect'), F.pad(y, [1] * 4, mode='reflect')
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x * x, 3, 1) - mu_x * mu_x
    sigma_y = F.avg_pool2d(y * y, 3, 1) - mu_y * mu_y
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x * mu_x + mu_y * mu_y + C1) * (sigma_x + sigma_y + C2)
    return n / d


def photometric_map(est, image, ssim_w=0.85, C1=1e-4, C2=9e-4):
    """calc_photometric_loss for one (estimate, image) pair, clip_loss == 0.  :188-223 and :169-186"""
    l1 = (est - image).abs()
    if not ssim_w > 0.0:
        return l1                       # L1 only: the reference keeps the 3-channel map (:205-213)
    s = torch.clamp((1. - ssim(est, image, C1, C2)) / 2., 0., 1.)
    return ssim_w * s.mean(1, True) + (1 - ssim_w) * l1.mean(1, True)


def reduce_candidates(maps, op='min'):
    """reduce_function of reduce_photometric_loss.  :238-246"""
    if op == 'mean':
        return sum(m.mean() for m in maps) / len(maps)
    return torch.cat(maps, 1).min(1, True)[0].mean()


def smoothness_terms(inv_depth, image):
    """calc_smoothness for one scale: returns (mean|Sx|, mean|Sy|).  utils/depth.py:146-198, utils/image.py:85-113,
    losses/multiview_photometric_loss.py:276-278"""
    mean = inv_depth.mean(2, True).mean(3, True)
    d = inv_depth / mean.clamp(min=1e-6)
    gx = d[:, :, :, :-1] - d[:, :, :, 1:]
    gy = d[:, :, :-1, :] - d[:, :, 1:, :]
    wx = torch.exp(-(image[:, :, :, :-1] - image[:, :, :, 1:]).abs().mean(1, True))
    wy = torch.exp(-(image[:, :, :-1, :] - image[:, :, 1:, :]).abs().mean(1, True))
    return (gx * wx).abs().mean(), (gy * wy).abs().mean()


def match_scales(image, targets):
    """utils/image.py:178-214 (bilinear, align_corners=True; identity when shapes agree)"""
    out = []
    for t in targets:
        if tuple(image.shape[-2:]) == tuple(t.shape[-2:]):
            out.append(image)
        else:
            out.append(F.interpolate(image, size=t.shape[-2:], mode='bilinear', align_corners=True))
    return out


def multiview_photometric_loss(image, context, inv_depths, K, ref_K, pose_mats, num_scales=4, ssim_loss_weight=0.85,
                               smooth_loss_weight=0.001, C1=1e-4, C2=9e-4, photometric_reduce_op='min',
                               automask_loss=True, padding_mode='zeros', clip_loss=0.0):
    """MultiViewPhotometricLoss.forward (progressive_scaling = 0).  :287-344; clip_loss: :214-219 (each candidate map is
    clamped at the float mean + clip_loss * std of itself).
    pose_mats: list of [B,4,4] (Pose.mat of the target->context transforms).  Returns (loss[1], photo, smooth)."""
    n = num_scales
    H, W = image.shape[-2:]
    images = match_scales(image, inv_depths[:n])
    cands = [[] for _ in range(n)]

    def clip(m):
        if clip_loss > 0.0:
            return torch.clamp(m, max=float((m.mean() + clip_loss * m.std()).detach()))
        return m

    for ref_image, T in zip(context, pose_mats):
        ref_images = match_scales(ref_image, inv_depths[:n])
        for i in range(n):
            DW = inv_depths[i].shape[-1]
            s = DW / float(W)
            Ki = K.float() if s == 1. else scale_intrinsics(K.float(), s, s)          # :153-157, camera.py:84-108
            rKi = ref_K.float() if s == 1. else scale_intrinsics(ref_K.float(), s, s)
            warped = view_synthesis(ref_images[i], inv_depths[i], Ki, rKi, T, padding_mode)
            cands[i].append(clip(photometric_map(warped, images[i], ssim_loss_weight, C1, C2)))
            if automask_loss:
                cands[i].append(clip(photometric_map(ref_images[i], images[i], ssim_loss_weight, C1, C2)))
    photo = sum(reduce_candidates(cands[i], photometric_reduce_op) for i in range(n)) / n
    loss = photo
    smooth = torch.zeros((), dtype=image.dtype, device=image.device)
    if smooth_loss_weight > 0.0:
        terms = [smoothness_terms(inv_depths[i], images[i]) for i in range(n)]
        smooth = smooth_loss_weight * sum((sx + sy) / 2 ** i for i, (sx, sy) in enumerate(terms)) / n
        loss = loss + smooth
    return loss.unsqueeze(0), photo, smooth


# ------------------------------------------------------------------------------------------------------------------
# SelfSupModel step
# ------------------------------------------------------------------------------------------------------------------


def selfsup_forward(sd_depth, sd_pose, batch, flip=False, upsample_depth_maps=True, **loss_kwargs):
    """SelfSupModel.forward in training mode (models/SelfSupModel.py:63-97, models/SfmModel.py:53-127,
    models/model_utils.py:97-180).  `flip` replaces the python-RNG draw of SfmModel.py:84."""
    rgb = batch['rgb']
    if flip:
        inv_depths = [torch.flip(d, [3]) for d in packnet01_forward(sd_depth, torch.flip(rgb, [3]), training=True)]
    else:
        inv_depths = packnet01_forward(sd_depth, rgb, training=True)
    if upsample_depth_maps:
        shape = inv_depths[0].shape[-2:]
        inv_depths = [F.interpolate(d, shape, mode='nearest') for d in inv_depths]
    pose_vec = posenet_forward(sd_pose, rgb, batch['rgb_context'])
    poses = [pose_vec2mat44(pose_vec[:, i]) for i in range(pose_vec.shape[1])]
    loss, photo, smooth = multiview_photometric_loss(batch['rgb_original'], batch['rgb_context_original'], inv_depths,
                                                     batch['intrinsics'], batch['intrinsics'], poses, **loss_kwargs)
    return {'loss': loss, 'photometric_loss': photo, 'smoothness_loss': smooth, 'inv_depths': inv_depths,
            'pose_vec': pose_vec}


# ------------------------------------------------------------------------------------------------------------------
# parameter construction (shapes + init of PackNet01.__init__/init_weights and PoseNet.__init__/init_weights)
# ------------------------------------------------------------------------------------------------------------------


def _xavier(shape, gen):
    fan_in = shape[1] * int(math.prod(shape[2:]))
    fan_out = shape[0] * int(math.prod(shape[2:]))
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2 - 1) * a


def supervised_loss(inv_depths, gt_inv_depth, supervised_method='sparse-l1', num_scales=4):
    """SupervisedLoss.forward / calculate_loss with the loss functions of get_loss_func.
    losses/supervised_loss.py:11-88 (BerHuLoss threshold 0.2, SilogLoss ratio 10 / ratio2 0.85) and :138-181."""
    def match(gt, target):
        if tuple(gt.shape[-2:]) == tuple(target.shape[-2:]):
            return gt
        return F.interpolate(gt, size=target.shape[-2:], mode='nearest')

    def one(pred, gt):
        if supervised_method.startswith('sparse'):
            mask = gt > 0.
            pred, gt = pred[mask], gt[mask]
        if supervised_method.endswith('abs_rel'):
            return torch.mean(torch.abs(pred - gt) / pred)
        if supervised_method.endswith('l1'):
            return torch.mean(torch.abs(pred - gt))
        if supervised_method.endswith('mse'):
            return torch.mean((pred - gt) ** 2)
        if supervised_method.endswith('berhu'):
            c = 0.2 * torch.max(pred - gt)
            diff = (pred - gt).abs()
            diff2 = diff[(diff > c).detach()] ** 2
            return torch.cat((diff.flatten(), diff2.flatten())).mean()
        if supervised_method.endswith('silog'):
            ld = torch.log(pred * 10) - torch.log(gt * 10)
            return torch.sqrt(torch.mean(ld ** 2) - 0.85 * ld.mean() ** 2) * 10
        raise ValueError(supervised_method)

    return sum(one(inv_depths[i], match(gt_inv_depth, inv_depths[i])) for i in range(num_scales)) / num_scales


def packnet01_param_shapes(version='1A', ni=64, n1=64, d=8):
    """Every parameter of PackNet01 with the reference's key and shape.  PackNet01.py:25-96, layers01.py
    (ni = n1 = 32, d = 4 gives PackNetSlim01: PackNetSlim01.py:33-39)."""
    assert version[1:] in ('A', 'B')
    concat = version[1:] == 'A'                 # 'B': skip connections are added (PackNet01.py:46-52)
    no = 1
    n2, n3, n4, n5 = 64, 128, 256, 512
    shapes = {}

    def conv2D(p, cin, cout, k):
        shapes[p + '.conv_base.weight'] = (cout, cin, k, k)
        shapes[p + '.conv_base.bias'] = (cout,)
        shapes[p + '.normalize.weight'] = (cout,)
        shapes[p + '.normalize.bias'] = (cout,)

    def resconv(p, cin, cout):
        conv2D(p + '.conv1', cin, cout, 3)
        conv2D(p + '.conv2', cout, cout, 3)
        shapes[p + '.conv3.weight'] = (cout, cin, 1, 1)
        shapes[p + '.conv3.bias'] = (cout,)
        shapes[p + '.normalize.weight'] = (cout,)
        shapes[p + '.normalize.bias'] = (cout,)

    def resblock(p, cin, cout, n):
        resconv(p + '.0', cin, cout)
        for i in range(1, n):
            resconv('%s.%d' % (p, i), cout, cout)

    def conv3d(p):
        shapes[p + '.conv3d.weight'] = (d, 1, 3, 3, 3)
        shapes[p + '.conv3d.bias'] = (d,)

    conv2D('pre_calc', 3, ni, 5)
    for name, c, k in (('pack1', n1, 5), ('pack2', n2, 3), ('pack3', n3, 3), ('pack4', n4, 3), ('pack5', n5, 3)):
        conv2D(name + '.conv', c * 4 * d, c, k)
        conv3d(name)
    conv2D('conv1', ni, n1, 7)
    resblock('conv2', n1, n2, 2)
    resblock('conv3', n2, n3, 2)
    resblock('conv4', n3, n4, 3)
    resblock('conv5', n4, n5, 3)
    if concat:
        n1o, n2o, n3o, n4o, n5o = n1, n2, n3, n4, n5
        n1i, n2i, n3i, n4i, n5i = n1 + ni + no, n2 + n1 + no, n3 + n2 + no, n4 + n3, n5 + n4
    else:
        n1o, n2o, n3o, n4o, n5o = n1, n2, n3 // 2, n4 // 2, n5 // 2
        n1i, n2i, n3i, n4i, n5i = n1 + no, n2 + no, n3 // 2 + no, n4 // 2, n5 // 2
    for name, cin, cout in (('unpack5', n5, n5o), ('unpack4', n5, n4o), ('unpack3', n4, n3o), ('unpack2', n3, n2o),
                            ('unpack1', n2, n1o)):
        conv2D(name + '.conv', cin, cout * 4 // d, 3)
        conv3d(name)
    conv2D('iconv5', n5i, n5, 3)
    conv2D('iconv4', n4i, n4, 3)
    conv2D('iconv3', n3i, n3, 3)
    conv2D('iconv2', n2i, n2, 3)
    conv2D('iconv1', n1i, n1, 3)
    for name, c in (('disp4_layer', n4), ('disp3_layer', n3), ('disp2_layer', n2), ('disp1_layer', n1)):
        shapes[name + '.conv1.weight'] = (no, c, 3, 3)
        shapes[name + '.conv1.bias'] = (no,)
    return shapes


def posenet_param_shapes(nb_ref_imgs=2):
    """PoseNet.py:38-56"""
    ch = [16, 32, 64, 128, 256, 256, 256]
    ks = [7, 5, 3, 3, 3, 3, 3]
    shapes = {}
    cin = 3 * (1 + nb_ref_imgs)
    for i, (c, k) in enumerate(zip(ch, ks)):
        p = 'conv%d' % (i + 1)
        shapes[p + '.0.weight'] = (c, cin, k, k)
        shapes[p + '.0.bias'] = (c,)
        shapes[p + '.1.weight'] = (c,)
        shapes[p + '.1.bias'] = (c,)
        cin = c
    shapes['pose_pred.weight'] = (6 * nb_ref_imgs, cin, 1, 1)
    shapes['pose_pred.bias'] = (6 * nb_ref_imgs,)
    return shapes


def init_params(shapes, seed=0, randomize_affine=False):
    """xavier_uniform conv weights, zero conv biases, GroupNorm affine (1, 0) as in init_weights
    (PackNet01.py:98-104, PoseNet.py:58-63).  randomize_affine perturbs biases / GN affine for stronger tests."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in shapes.items():
        if len(shp) >= 4:
            sd[k] = _xavier(shp, gen)
        elif k.endswith('normalize.weight') or k.endswith('.1.weight'):
            sd[k] = torch.ones(shp) + (0.2 * torch.randn(shp, generator=gen) if randomize_affine else 0)
        else:
            sd[k] = 0.1 * torch.randn(shp, generator=gen) if randomize_affine else torch.zeros(shp)
    return sd
