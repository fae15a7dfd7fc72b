"""TEST INFRASTRUCTURE -- gather-based restatement of the MinkowskiEngine operations PackNet-SAN's depth branch uses
(/root/reference/packnet_sfm/networks/layers/minkowski_encoder.py:10-131, minkowski.py:33-83), on explicit coordinate lists.

PARITY UNPINNED: MinkowskiEngine is a third-party dependency that the reference neither vendors nor versions
(docker/Dockerfile installs it from git master) and it is not installed here; the reference has no test or golden vector
for this branch.  The rules below follow the MinkowskiEngine 0.5 documentation (generalised sparse convolution with a
centred hyper-cube kernel on the input coordinates for stride 1; strided pooling creates the coordinates
floor(c / s) * s and reduces over the kernel region around them; MinkowskiBatchNorm = BatchNorm1d over the feature rows).
This oracle is deliberately written the SPARSE way (dict of coordinates, per-offset gathers) so that it is an independent
check of the product's dense-plus-mask formulation.  Only tests / smoke may import it."""
import torch


def sparsify(depth):
    """[B,1,H,W] -> (coords int64 [N,3] = (b, y, x) of pixels with depth > 0, feats [N,1])."""
    idx = (depth[:, 0] > 0).nonzero()
    return idx, depth[idx[:, 0], 0, idx[:, 1], idx[:, 2]].unsqueeze(1)


def _index(coords):
    return {tuple(c.tolist()): i for i, c in enumerate(coords)}


def conv(coords, feats, kernel, ts):
    """kernel [k*k, in, out], offsets (dy, dx) * ts with kernel index i = (dy + r) + k * (dx + r); outputs on the input coords."""
    kk = kernel.shape[0]
    k = int(round(kk ** 0.5))
    r = k // 2
    table = _index(coords)
    out = feats.new_zeros(feats.shape[0], kernel.shape[2])
    for i in range(kk):
        dy, dx = (i % k) - r, (i // k) - r
        src, dst = [], []
        for n, c in enumerate(coords.tolist()):
            j = table.get((c[0], c[1] + dy * ts, c[2] + dx * ts))
            if j is not None:
                src.append(j)
                dst.append(n)
        if src:
            out = out.index_add(0, torch.tensor(dst), feats[torch.tensor(src)] @ kernel[i])
    return out


def maxpool3s2(coords, feats, ts):
    """MaxPooling(3, stride 2): new coordinates floor(c / 2ts) * 2ts; max over the active inputs at origin + {-ts, 0, ts}^2."""
    ns = 2 * ts
    oc = coords.clone()
    oc[:, 1:] = torch.div(coords[:, 1:], ns, rounding_mode='floor') * ns
    oc = torch.unique(oc, dim=0)
    table = _index(coords)
    rows = []
    for c in oc.tolist():
        cand = [table.get((c[0], c[1] + dy * ts, c[2] + dx * ts)) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
        cand = [j for j in cand if j is not None]
        rows.append(feats[torch.tensor(cand)].max(0).values)
    return oc, torch.stack(rows), ns


def batchnorm(feats, weight, bias, eps=1e-5):
    mean = feats.mean(0)
    var = feats.var(0, unbiased=False)
    return (feats - mean) * torch.rsqrt(var + eps) * weight + bias


def densify(coords, feats, shape, ts):
    B, _, H, W = shape
    dense = feats.new_zeros(B, -(-H // ts), -(-W // ts), feats.shape[1])        # ceil: floor(c / ts) of the last coordinate + 1
    dense[coords[:, 0], coords[:, 1] // ts, coords[:, 2] // ts] = feats
    return dense.permute(0, 3, 1, 2).contiguous()


def mink_conv2d(block_params, coords, feats, ts, prefix):
    """One MinkConv2D block (pool -> three conv stacks -> sum -> BN -> ReLU) from a state dict of the product module."""
    P = lambda n: block_params[prefix + n]                                   # noqa: E731
    coords, feats, ts = maxpool3s2(coords, feats, ts)

    def stack(name, n_convs):
        f = feats
        for c in range(n_convs):
            f = conv(coords, f, P('%s.%d.kernel' % (name, 3 * c)), ts)
            if c + 1 < n_convs:
                f = torch.relu(batchnorm(f, P('%s.%d.bn.weight' % (name, 3 * c + 1)), P('%s.%d.bn.bias' % (name, 3 * c + 1))))
        return f
    s = stack('layer1', 1) + stack('layer2', 2) + stack('layer3', 3)
    out = torch.relu(batchnorm(s, P('layer_final.0.bn.weight'), P('layer_final.0.bn.bias')))
    return coords, out, ts
