"""Sparse tensors of PackNet-SAN's depth branch (csrc/sparse.hip, networks/layers/minkowski*.py).

MinkowskiEngine -- what the reference runs this branch on (networks/layers/minkowski_encoder.py:10-131) -- is absent here, so its
DOCUMENTED rules are pinned by hand-computed vectors first (a 4x4 grid with four LiDAR returns; every number below can be checked
with pencil and paper) on BOTH the oracle (oracle/minkowski_oracle.py, a coordinate-dictionary restatement) and the kernels;
random cases then compare the kernels (values and gradients) with that oracle.  CPU: host-emulated kernels; GPU: gfx950."""
import pytest
import torch

import parity_cases as P

# ---- the hand-computable case ---------------------------------------------------------------------------------------------
# depth[y][x] on a 4x4 grid: returns at (0,0)=1, (0,1)=2, (1,2)=5, (2,2)=3 (row-major site order)
HAND_SITES = [(0, 0), (0, 1), (1, 2), (2, 2)]
HAND_VALUES = [1.0, 2.0, 5.0, 3.0]
# 3x3 kernel, one channel in / out, kernel[i] = i + 1 with i = (dy + 1) + 3 * (dx + 1)   (MinkowskiEngine's offset order)
#   (0,0): self 5*1 + right neighbour (0,1) via offset (0,+1) = i 7: 8*2                           = 21
#   (0,1): self 5*2 + (0,0) via (0,-1) = i 1: 2*1 + (1,2) via (+1,+1) = i 8: 9*5                   = 57
#   (1,2): self 5*5 + (0,1) via (-1,-1) = i 0: 1*2 + (2,2) via (+1,0) = i 5: 6*3                   = 45
#   (2,2): self 5*3 + (1,2) via (-1,0) = i 3: 4*5                                                  = 35
HAND_CONV = [21.0, 57.0, 45.0, 35.0]
# MaxPooling(3, stride 2): coarse cells floor(c/2) -> (0,0) <- {(0,0),(0,1)}, (0,1) <- {(1,2)}, (1,1) <- {(2,2)};
# value = max over the ACTIVE fine cells of the 3x3 window centred on the coarse cell's origin (2Y, 2X):
#   (0,0): rows -1..1, cols -1..1 -> {1, 2}      = 2        (0,1): rows -1..1, cols 1..3 -> {2, 5} = 5
#   (1,1): rows  1..3, cols  1..3 -> {5, 3}      = 5
HAND_POOL_SITES = [(0, 0), (0, 1), (1, 1)]
HAND_POOL = [2.0, 5.0, 5.0]


def _hand_depth(device='cpu'):
    d = torch.zeros(1, 1, 4, 4)
    for (y, x), v in zip(HAND_SITES, HAND_VALUES):
        d[0, 0, y, x] = v
    return d.to(device)


def _hand_kernel():
    return torch.arange(1.0, 10.0).view(9, 1, 1)


def test_oracle_matches_hand_computed_vectors():
    from oracle import minkowski_oracle as MO
    coords, feats = MO.sparsify(_hand_depth())
    assert coords.tolist() == [[0, y, x] for y, x in HAND_SITES] and feats[:, 0].tolist() == HAND_VALUES
    assert MO.conv(coords, feats, _hand_kernel(), 1)[:, 0].tolist() == HAND_CONV
    pc, pf, ts = MO.maxpool3s2(coords, feats, 1)
    assert ts == 2 and pc.tolist() == [[0, 2 * y, 2 * x] for y, x in HAND_POOL_SITES] and pf[:, 0].tolist() == HAND_POOL
    bn = MO.batchnorm(torch.tensor(HAND_CONV).view(4, 1), torch.tensor([2.0]), torch.tensor([0.5]), eps=0.0)
    ref = [(v - 39.5) / 174.75 ** 0.5 * 2.0 + 0.5 for v in HAND_CONV]         # mean 39.5, population variance 174.75
    assert torch.allclose(bn[:, 0], torch.tensor(ref), atol=1e-6)
    dense = MO.densify(pc, pf, (1, 1, 4, 4), ts)
    assert dense.shape == (1, 1, 2, 2) and dense[0, 0].tolist() == [[2.0, 5.0], [0.0, 5.0]]


def _check_hand_vectors(device):
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.layers import minkowski as M
    from packnet_sfm.networks.layers.minkowski_encoder import MinkowskiBatchNorm, MinkowskiMaxPooling
    x = M.sparsify_depth(_hand_depth(device))
    n = len(HAND_SITES)
    assert int(x.count) == n and x.cap == 32
    assert x.sites[:n].tolist() == [y * 4 + xx for y, xx in HAND_SITES]
    assert x.imap.view(4, 4).tolist() == [[0, 1, -1, -1], [-1, -1, 2, -1], [-1, -1, 3, -1], [-1, -1, -1, -1]]
    assert x.F[:n, 0].tolist() == HAND_VALUES and float(x.F[n:].abs().sum()) == 0.0
    nbr = x.neighbors(3)
    # site (0,1): offset (0,-1) = i 1 -> site 0, itself = i 4, offset (+1,+1) = i 8 -> site 2, nothing else
    assert nbr[1].tolist() == [-1, 0, -1, -1, 1, -1, -1, -1, 2]
    y = HF.sparse_conv(x.F, _hand_kernel().to(device), nbr, x.count, 3)
    assert y[:n, 0].tolist() == HAND_CONV and float(y[n:].abs().sum()) == 0.0
    p = MinkowskiMaxPooling(3, 2)(x)
    m = len(HAND_POOL_SITES)
    assert int(p.count) == m and (p.h, p.w, p.tensor_stride) == (2, 2, 2)
    assert p.sites[:m].tolist() == [yy * 2 + xx for yy, xx in HAND_POOL_SITES] and p.F[:m, 0].tolist() == HAND_POOL
    assert M.densify_features(p, (1, 1, 4, 4))[0, 0].tolist() == [[2.0, 5.0], [0.0, 5.0]]
    bn = MinkowskiBatchNorm(1, eps=0.0).to(device).train()
    with torch.no_grad():
        bn.bn.weight.fill_(2.0); bn.bn.bias.fill_(0.5)
    z = bn(x.with_features(y)).F
    ref = torch.tensor([(v - 39.5) / 174.75 ** 0.5 * 2.0 + 0.5 for v in HAND_CONV])
    assert torch.allclose(z[:n, 0].cpu(), ref, atol=1e-5) and float(z[n:].abs().sum()) == 0.0


def test_kernels_match_hand_computed_vectors_emulated(emulated_kernels):
    _check_hand_vectors('cpu')


@pytest.mark.gpu
def test_kernels_match_hand_computed_vectors_gpu():
    _check_hand_vectors('cuda')


# ---- random cases against the oracle ---------------------------------------------------------------------------------------
def _sparse_depth(B, H, W, density, seed):
    g = torch.Generator().manual_seed(seed)
    d = torch.rand(B, 1, H, W, generator=g) * 60 + 2
    return d * (torch.rand(B, 1, H, W, generator=g) < density)


def _rows_of(grid, t):
    """Feature rows of the active sites, in site order, on the CPU."""
    return t[:int(grid.count)].detach().cpu()


def _check_conv_and_pool(device, B, H, W, density, Cin, Cout, ks, seed):
    from oracle import minkowski_oracle as MO
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.layers import minkowski as M
    from packnet_sfm.networks.layers.minkowski_encoder import MinkowskiMaxPooling
    depth = _sparse_depth(B, H, W, density, seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = M.sparsify_depth(depth.to(device))
    coords, _ = MO.sparsify(depth)
    n = len(coords)
    assert int(x.count) == n
    cells = coords[:, 0] * H * W + coords[:, 1] * W + coords[:, 2]
    assert torch.equal(x.sites[:n].cpu().long(), cells)
    if n == 0:          # no return at all: every kernel must cope with an empty site list (zeros out, zero gradients)
        fh = torch.zeros(x.cap, Cin, device=device, requires_grad=True)
        kh = torch.randn(ks * ks, Cin, Cout, generator=g).to(device).requires_grad_(True)
        y = HF.sparse_conv(fh, kh, x.neighbors(ks), x.count, ks)
        y.sum().backward()
        assert float(y.abs().sum()) == 0.0 and float(kh.grad.abs().sum()) == 0.0 and float(fh.grad.abs().sum()) == 0.0
        p = MinkowskiMaxPooling(3, 2)(x.with_features(fh))
        assert int(p.count) == 0 and float(M.densify_features(p, depth.shape).abs().sum()) == 0.0
        return
    # features of Cin channels on those coordinates
    f0 = torch.randn(n, Cin, generator=g)
    kern = torch.randn(ks * ks, Cin, Cout, generator=g) * 0.2
    dout = torch.randn(n, Cout, generator=g)
    fr, kr = f0.clone().requires_grad_(True), kern.clone().requires_grad_(True)
    yr = MO.conv(coords, fr, kr, 1)
    yr.backward(dout)
    pad = lambda t: torch.cat([t, t.new_zeros(x.cap - n, t.shape[1])], 0).to(device)     # noqa: E731
    fh, kh = pad(f0).requires_grad_(True), kern.clone().to(device).requires_grad_(True)
    y = HF.sparse_conv(fh, kh, x.neighbors(ks), x.count, ks)
    assert float(y[n:].abs().sum()) == 0.0
    y.backward(pad(dout))
    P.check(_rows_of(x, y), yr, 2e-5, 'sparse conv forward')
    P.check(_rows_of(x, fh.grad), fr.grad, 2e-5, 'sparse conv backward-data')
    P.check(kh.grad, kr.grad, 2e-5, 'sparse conv weight gradient')
    # pooling of those features: coordinates, values, gradient routing
    xin = x.with_features(pad(f0).requires_grad_(True))
    p = MinkowskiMaxPooling(3, 2)(xin)
    pc, pf, ts = MO.maxpool3s2(coords, f0.clone().requires_grad_(True), 1)
    m = len(pc)
    assert int(p.count) == m and ts == 2
    hc, wc = (H + 1) // 2, (W + 1) // 2                          # odd grids: the last odd row / column is a coarse cell of its own
    assert torch.equal(p.sites[:m].cpu().long(), pc[:, 0] * hc * wc + (pc[:, 1] // 2) * wc + pc[:, 2] // 2)
    P.check(_rows_of(p, p.F), pf, 0.0, 'max pooling values')
    fr2 = f0.clone().requires_grad_(True)
    _, pf2, _ = MO.maxpool3s2(coords, fr2, 1)
    gp = torch.randn(m, Cin, generator=g)
    pf2.backward(gp)
    p.F.backward(torch.cat([gp, gp.new_zeros(p.cap - m, Cin)], 0).to(device))
    P.check(_rows_of(x, xin.F.grad), fr2.grad, 1e-6, 'max pooling gradient')
    # densify and the dense pick-up are each other's adjoint and agree with the oracle
    dense = M.densify_features(p, depth.shape)
    P.check(dense, MO.densify(pc, pf.detach(), depth.shape, 2), 0.0, 'densify')
    back = HF.sparse_gather(dense, p.imap, p.sites, p.count, p.cap)
    P.check(_rows_of(p, back), pf.detach(), 0.0, 'gather(densify(rows)) == rows')


CASES = [  # B, H, W, density, Cin, Cout, k, seed
    (2, 8, 12, 0.3, 1, 8, 5, 0),         # level-0 shape: one input channel, 5x5
    (1, 6, 10, 0.5, 40, 33, 3, 1),       # ragged channel counts (tile tails in every dimension)
    (2, 8, 8, 1.0, 8, 96, 3, 2),         # fully occupied grid: must equal a dense zero-padded convolution
    (1, 12, 16, 0.05, 16, 16, 3, 3),     # LiDAR-like occupancy: most offsets have no neighbour (skipped)
    (1, 4, 6, 0.0, 4, 4, 3, 4),          # no return at all
    (2, 5, 7, 0.6, 8, 16, 3, 8),         # odd grid: coarse size ceil(h / 2) x ceil(w / 2), children outside the grid do not exist
]


@pytest.mark.parametrize('case', CASES)
def test_sparse_kernels_vs_oracle_emulated(emulated_kernels, case):
    _check_conv_and_pool('cpu', *case)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES + [(4, 48, 160, 0.15, 64, 128, 5, 5), (2, 24, 80, 0.8, 256, 256, 3, 6), (4, 6, 20, 1.0, 1024, 1024, 3, 7)])
def test_sparse_kernels_vs_oracle_gpu(case):
    _check_conv_and_pool('cuda', *case)


def test_full_grid_equals_dense_convolution(emulated_kernels):
    """On a fully occupied grid MinkowskiConvolution is a plain zero-padded cross-correlation with W[co][ci][dy][dx] = kernel[(dy + r) +
    k (dx + r)][ci][co] -- the statement that fixes the offset ORDER (first coordinate fastest), checked against F.conv2d."""
    import torch.nn.functional as F
    from packnet_sfm.hip import functional as HF
    from packnet_sfm.networks.layers import minkowski as M
    g = torch.Generator().manual_seed(3)
    B, C, H, W, Co, k = 2, 5, 6, 7, 4, 3
    x = torch.randn(B, C, H, W, generator=g)
    kern = torch.randn(k * k, C, Co, generator=g)
    s = M.sparsify_features(x)
    y = M.densify_features(s.with_features(HF.sparse_conv(s.F, kern, s.neighbors(k), s.count, k)), (B, Co, H, W))
    w = kern.view(k, k, C, Co).permute(3, 2, 1, 0).contiguous()        # i = ky + k * kx  ->  [co][ci][ky][kx]
    P.check(y, F.conv2d(x, w, padding=k // 2), 1e-5, 'full grid == dense conv2d')
