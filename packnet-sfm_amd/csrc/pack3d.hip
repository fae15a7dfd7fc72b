// pack3d.hip -- the data-movement and 3-D stencil half of the packing / unpacking blocks.
//
//  space_to_depth  == packing(x, r=2)        /root/reference/packnet_sfm/networks/layers/packnet/layers01.py:126-148
//  depth_to_space  == nn.PixelShuffle(2)     layers01.py:275,285
//  conv3d_1to8     == nn.Conv3d(1, 8, 3x3x3, padding 1) applied to x.unsqueeze(1) and viewed back as
//                     [B, 8*D, H, W] with channel f*D + d      layers01.py:236-237,241-245 and :276-277,280-284
//
// All three are HBM-bound (27-tap VALU stencil, 1 read : 8 writes forward); no matrix cores here.
// Zero padding is along ALL three stencil axes, including the channel ("depth") axis d.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

// out-of-volume neighbours are read from here (pointer select keeps the loads unconditional, see conv2d.hip)
__device__ __attribute__((aligned(16))) float pnsfm_zero_page3[64];

// y[b][4c+2i+j][h][w] = x[b][c][2h+i][2w+j]; one thread per input 2x2 quad column pair
__global__ void __launch_bounds__(256) s2d_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   int C, int H, int W, size_t total, size_t x_batch_stride) {
  // thread per OUTPUT element, w fastest (coalesced writes; reads are stride-2 but both parities are
  // consumed by neighbouring output channels of the same block row, i.e. served from L1/L2)
  const int h2 = H >> 1, w2 = W >> 1;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int w = (int)(idx % w2);
    size_t r = idx / w2;
    const int h = (int)(r % h2);
    r /= h2;
    const int oc = (int)(r % (4 * C));
    const int b = (int)(r / (4 * C));
    const int c = oc >> 2, i = (oc >> 1) & 1, j = oc & 1;
    y[idx] = x[(size_t)b * x_batch_stride + ((size_t)c * H + 2 * h + i) * W + 2 * w + j];
  }
}

// y[b][c][2h+i][2w+j] = x[b][4c+2i+j][h][w]; thread per OUTPUT element (x-fastest)
__global__ void __launch_bounds__(256) d2s_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                   int C, int H, int W, size_t total) {
  const int H2 = 2 * H, W2 = 2 * W;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int ww = (int)(idx % W2);
    size_t r = idx / W2;
    const int hh = (int)(r % H2);
    r /= H2;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const int ic = 4 * c + 2 * (hh & 1) + (ww & 1);
    y[idx] = x[(((size_t)b * 4 * C + ic) * H + (hh >> 1)) * W + (ww >> 1)];
  }
}

// 16-byte forms of the two shuffles (W % 8 == 0 / 4, 16-byte aligned tensors): 126 MB each way at full resolution, so they are
// written as streaming copies -- one thread moves a 2 x 8 input block (s2d) / four 4-wide channel rows (d2s) with float4
// accesses and 32-bit index arithmetic (the per-element forms above spend their time in 64-bit divisions: 2.9 TB/s).
__global__ void __launch_bounds__(256) s2d_v4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W,
                                                      unsigned total_q, size_t x_batch_stride) {
  const unsigned h2 = H >> 1, wqn = W >> 3;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= total_q) return;
  const unsigned wq = idx % wqn, r = idx / wqn;
  const unsigned h = r % h2, bc = r / h2;
  const unsigned c = bc % (unsigned)C, b = bc / (unsigned)C;
  const float* src = x + (size_t)b * x_batch_stride + ((size_t)c * H + 2 * h) * W + 8 * wq;
  const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(src + W), b1 = *reinterpret_cast<const float4*>(src + W + 4);
  const size_t plane = (size_t)h2 * (W >> 1);
  float* dst = y + (((size_t)b * 4 * C + 4 * c) * h2 + h) * (W >> 1) + 4 * wq;
  *reinterpret_cast<float4*>(dst) = make_float4(a0.x, a0.z, a1.x, a1.z);                 // i = 0, j = 0
  *reinterpret_cast<float4*>(dst + plane) = make_float4(a0.y, a0.w, a1.y, a1.w);         // i = 0, j = 1
  *reinterpret_cast<float4*>(dst + 2 * plane) = make_float4(b0.x, b0.z, b1.x, b1.z);     // i = 1, j = 0
  *reinterpret_cast<float4*>(dst + 3 * plane) = make_float4(b0.y, b0.w, b1.y, b1.w);     // i = 1, j = 1
}

__global__ void __launch_bounds__(256) d2s_v4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W,
                                                      unsigned total_q) {
  const unsigned wqn = W >> 2;
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= total_q) return;
  const unsigned wq = idx % wqn, r = idx / wqn;
  const unsigned h = r % (unsigned)H, bc = r / (unsigned)H;       // bc = b * C + c
  const size_t plane = (size_t)H * W;
  const float* src = x + ((size_t)bc * 4 * H + h) * W + 4 * wq;
  const float4 p00 = *reinterpret_cast<const float4*>(src), p01 = *reinterpret_cast<const float4*>(src + plane);
  const float4 p10 = *reinterpret_cast<const float4*>(src + 2 * plane), p11 = *reinterpret_cast<const float4*>(src + 3 * plane);
  float* dst = y + ((size_t)bc * 2 * H + 2 * h) * (2 * W) + 8 * wq;
  *reinterpret_cast<float4*>(dst) = make_float4(p00.x, p01.x, p00.y, p01.y);
  *reinterpret_cast<float4*>(dst + 4) = make_float4(p00.z, p01.z, p00.w, p01.w);
  *reinterpret_cast<float4*>(dst + 2 * W) = make_float4(p10.x, p11.x, p10.y, p11.y);
  *reinterpret_cast<float4*>(dst + 2 * W + 4) = make_float4(p10.z, p11.z, p10.w, p11.w);
}

// out[b][f*D+d][y][x] = b3[f] + sum_{dz,dy,dx} w3[f][dz][dy][dx] * p[b][d+dz-1][y+dy-1][x+dx-1]
// grid: (ceil(D*HW/256), 1, B): one thread per voxel (d, y, x) -- flattened so that small planes (the 7x7 / 5x5 weight
// volumes of the kernel composition, or 6x20 feature maps) still fill every lane -- produces all 8 features;
// 27 branch-free neighbour loads (L1/L2 serve the overlap between neighbouring threads), 216 FMAs, 8 stores.
template <int NF>   // number of 3-D feature maps: 8 (PackNet01) or 4 (PackNetSlim01 / PackNetSAN01, `d=num_3d_feat`)
__global__ void __launch_bounds__(256) conv3d_fwd_kernel(const float* __restrict__ p, const float* __restrict__ w3,
                                                          const float* __restrict__ b3, float* __restrict__ out,
                                                          int D, int H, int W) {
  __shared__ float ws[NF * 27 + NF];
  for (int i = threadIdx.x; i < NF * 27 + NF; i += 256) ws[i] = i < NF * 27 ? w3[i] : (b3 ? b3[i - NF * 27] : 0.f);
  __syncthreads();
  const int HW = H * W, DHW = D * HW;
  const int vox = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.z;
  const bool active = vox < DHW;
  const int d = active ? vox / HW : 0;
  const int pix = active ? vox - d * HW : 0;
  const int y = pix / W, x = pix - y * W;
  const float* pb = p + (size_t)b * DHW;
  float v[27];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    const int dd = d + dz - 1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = x + dx - 1;
        const bool ok = active && dd >= 0 && dd < D && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float* src = ok ? pb + (dd * HW + yy * W + xx) : pnsfm_zero_page3;
        v[dz * 9 + dy * 3 + dx] = *src;
      }
    }
  }
  float acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = ws[NF * 27 + f];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = fmaf(ws[f * 27 + tap], v[tap], acc[f]);
  if (active) {
    float* ob = out + (size_t)b * NF * DHW + vox;
#pragma unroll
    for (int f = 0; f < NF; ++f) ob[(size_t)f * DHW] = acc[f];
  }
}

// Four voxels per thread along x (W % 4 == 0, 16-byte aligned tensors): the 9 rows of the stencil are read as one 16-byte load
// plus the two edge values each (27 load instructions per 4 voxels instead of 108), the weights as [tap][feature] float4
// broadcasts from LDS, the NF outputs leave as float4 stores.  The kernel writes NF x its input: HBM-write-bound.
template <int NF>
__global__ void __launch_bounds__(256) conv3d_fwd_x4_kernel(const float* __restrict__ p, const float* __restrict__ w3,
                                                             const float* __restrict__ b3, float* __restrict__ out,
                                                             int D, int H, int W) {
  __shared__ __attribute__((aligned(16))) float wt[28][NF];      // [tap][feature]; row 27 = bias
  for (int i = threadIdx.x; i < 28 * NF; i += 256) {
    const int tap = i / NF, f = i - tap * NF;
    wt[tap][f] = tap < 27 ? w3[f * 27 + tap] : (b3 ? b3[f] : 0.f);
  }
  __syncthreads();
  const int HW = H * W, DHW = D * HW, Wq = W >> 2, HWq = H * Wq;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= D * HWq) return;
  const int d = idx / HWq;
  const int pq = idx - d * HWq;
  const int y = pq / Wq, x0 = 4 * (pq - y * Wq);
  const float* pb = p + (size_t)blockIdx.z * DHW;
  const float* zero = pnsfm_zero_page3;
  float acc[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[f][i] = wt[27][f];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    const int dd = d + dz - 1;
    float v[3][6];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
      const bool ok = dd >= 0 && dd < D && yy >= 0 && yy < H;
      const float* row = pb + (size_t)(ok ? dd : 0) * HW + (ok ? yy : 0) * W + x0;
      const float4 c = *reinterpret_cast<const float4*>(ok ? row : zero);
      v[dy][0] = *((ok && x0 > 0) ? row - 1 : zero);
      v[dy][1] = c.x; v[dy][2] = c.y; v[dy][3] = c.z; v[dy][4] = c.w;
      v[dy][5] = *((ok && x0 + 4 < W) ? row + 4 : zero);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        float wv[NF];
#pragma unroll
        for (int q = 0; q < NF / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(&wt[dz * 9 + dy * 3 + dx][4 * q]);
          wv[4 * q] = t.x; wv[4 * q + 1] = t.y; wv[4 * q + 2] = t.z; wv[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[f][i] = fmaf(wv[f], v[dy][i + dx], acc[f][i]);      // p[.][y + dy - 1][x0 + i + dx - 1]
      }
  }
  float* ob = out + (size_t)blockIdx.z * NF * DHW + (size_t)d * HW + y * W + x0;
#pragma unroll
  for (int f = 0; f < NF; ++f)
    *reinterpret_cast<float4*>(ob + (size_t)f * DHW) = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
}

// dp[b][d][y][x] = sum_{f,dz,dy,dx} w3[f][dz][dy][dx] * dout[b][f*D + d-dz+1][y-dy+1][x-dx+1]
// The per-voxel form of this stencil issues 216 loads per output and is bound by the L1/TA path (measured: 166 us for
// unpack1, 5x its HBM time).  Here one thread owns a column of `len` consecutive d at a fixed (y, x) and slides along d:
// each dout plane (8 features x 9 in-plane neighbours = 72 raw buffer loads; zero padding by the hardware range check)
// is loaded once and feeds the three outputs d = dd-1, dd, dd+1 (three rolling accumulators) -> 72*(len+2)/len loads
// per output, ~80 for the run lengths the launcher picks.  (Round 3 tried four outputs per thread along x -- one 16-byte
// load + two edge values per row, 18 loads per output -- with 2 / 4 / 8 features per unrolled step: 123 / 140 / 542 us on
// unpack1 against this kernel's 94, and weights re-read from LDS instead of the ~240 spilled scalar registers: the compiler
// then keeps all 216 in VGPRs.  Dropped.)
// Lanes still run along x (coalesced); the flattened (chunk, pixel) index keeps tiny planes (6x20 maps, 5x5 / 7x7 weight
// volumes of the kernel composition) on full waves.  Weights are uniform global reads (scalar loads -> SGPR operands).
// Round 4: the kernel is bound by the number of load instructions (72 per plane and thread: ~90 per output, TA-limited at 1.24 TB/s
// effective).  Lanes run along x, so the x - 1 / x + 1 neighbours of a lane's centre value ARE its neighbour lanes' centre values:
// a thread now loads the three centre values of a plane (one per dy) per feature -- 24 loads instead of 72 -- and takes the other
// 48 from the lanes next to it (wave shifts).  A wave covers 62 outputs: lanes 0 and 63 are halo lanes that only supply centres
// (97 % lane efficiency); row / image ends are masked (x == 0 has no left neighbour whatever lane holds pixel - 1).  The d loop
// has the SAME trip count (len + 2) on every lane -- lanes of a wave may sit in different d chunks -- so the shifts are never
// executed under divergence.
template <int NF>
__global__ void __launch_bounds__(256) conv3d_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w3,
                                                            float* __restrict__ dp, int D, int H, int W, int len, int xmap) {
  const int HW = H * W, DHW = D * HW;
  const int nchunk = (D + len - 1) / len;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows y - 1 / y + 1 of a block's pixels are the centre rows of the blocks a few places before / after it: an XCD gets a
  // contiguous range of the block order (pnsfm_common.h) so that those rows are found in ITS L2 instead of being fetched by three
  const unsigned bx = xmap ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const long lin = ((long)bx * 4 + wave) * 62 + lane - 1;      // flattened (chunk, pixel) index of this lane
  const bool inr = lin >= 0 && lin < (long)nchunk * HW;
  const bool active = inr && lane >= 1 && lane <= 62;
  const int chunk = inr ? (int)(lin / HW) : 0;
  const int pix = inr ? (int)(lin - (long)chunk * HW) : 0;
  const int y = pix / W, x = pix - y * W;
  const int d0 = chunk * len;
  const int dend = (d0 + len < D) ? d0 + len : D;
  const bool has_l = x > 0, has_r = x + 1 < W;
  // one descriptor per feature slab [D][H][W] (the range-checked window); all wave-uniform, they live in SGPRs
  pnsfm_buf gbuf[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) gbuf[f] = pnsfm_make_buf(dout + ((size_t)blockIdx.z * NF + f) * DHW, (unsigned)DHW * 4u);
  float* ob = dp + (size_t)blockIdx.z * DHW + pix;
  const unsigned kOut = 0x7fffffffu;       // out-of-range byte offset -> the load returns 0
  unsigned off[3];                         // in-plane byte offsets of the centre column in rows y + 1, y, y - 1 (kOut outside the image)
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = y - dy + 1;
    off[dy] = (inr && yy >= 0 && yy < H) ? (unsigned)(yy * W + x) * 4u : kOut;
  }
  // a_lo -> output dd-1 (complete after this plane), a_mid -> output dd, a_hi -> output dd+1
  float a_lo = 0.f, a_mid = 0.f, a_hi = 0.f;
#pragma unroll 1
  for (int t = 0; t < len + 2; ++t) {
    const int dd = d0 - 1 + t;
    const bool dok = dd >= 0 && dd < D && dd <= dend;
    const unsigned plane = dok ? (unsigned)dd * (unsigned)HW * 4u : 0u;
    float g[NF][9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const unsigned vo = (dok && off[dy] != kOut) ? plane + off[dy] : kOut;
#pragma unroll
      for (int f = 0; f < NF; ++f) g[f][dy * 3 + 1] = pnsfm_buf_load(gbuf[f], vo, 0u);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const float c = g[f][dy * 3 + 1];
        const float r = __shfl_down(c, 1), l = __shfl_up(c, 1);
        g[f][dy * 3 + 0] = has_r ? r : 0.f;       // dx = 0: xx = x + 1
        g[f][dy * 3 + 2] = has_l ? l : 0.f;       // dx = 2: xx = x - 1
      }
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        a_lo = fmaf(w3[f * 27 + tp], g[f][tp], a_lo);             // dz = 0: d = dd - 1
        a_mid = fmaf(w3[f * 27 + 9 + tp], g[f][tp], a_mid);       // dz = 1: d = dd
        a_hi = fmaf(w3[f * 27 + 18 + tp], g[f][tp], a_hi);        // dz = 2: d = dd + 1
      }
    if (active && dd - 1 >= d0 && dd - 1 < dend) ob[(size_t)(dd - 1) * HW] = a_lo;
    a_lo = a_mid;
    a_mid = a_hi;
    a_hi = 0.f;
  }
}

// The same stencil for runs of LEN = 8 / 4 / 2 planes (every volume on the training step): the kernel above needs all NF * 27 = 216
// weights as scalar operands in every plane step, the SGPR file holds ~100, and hipcc parks the rest in VGPR lanes -- 353
// v_readlane (+ wait states) per plane step against 144 FMA instructions.  Here a thread keeps the LEN outputs of its column in
// registers and walks the run once per FEATURE: that feature's 27 weights stay in SGPRs across the fully unrolled plane loop (the
// accumulator a plane feeds is a compile-time index, every FMA is half of a v_pk_fma_f32), nothing is parked or re-loaded, one
// buffer descriptor instead of eight, 94 VGPRs.  Loads and wave shifts per output are unchanged.  unpack1's volume (4 x 32 x 96 x
// 320): 107 -> 64 us; the step's nine volumes 417 -> 265 us (profiles/r04_ab_conv3d_dgrad_column.txt).
template <int NF, int LEN>
__global__ void __launch_bounds__(256) conv3d_dgrad_col_kernel(const float* __restrict__ dout, const float* __restrict__ w3,
                                                                float* __restrict__ dp, int D, int H, int W, int xmap) {
  constexpr int FG = 1;                    // features per pass (2: 54 weights + the plane masks overflow the SGPR file again, 76 vs 64 us)
  const int HW = H * W, DHW = D * HW;
  const int nchunk = (D + LEN - 1) / LEN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned bx = xmap ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const long lin = ((long)bx * 4 + wave) * 62 + lane - 1;      // flattened (chunk, pixel) index of this lane
  const bool inr = lin >= 0 && lin < (long)nchunk * HW;
  const bool active = inr && lane >= 1 && lane <= 62;
  const int chunk = inr ? (int)(lin / HW) : 0;
  const int pix = inr ? (int)(lin - (long)chunk * HW) : 0;
  const int y = pix / W, x = pix - y * W;
  const int d0 = chunk * LEN;
  const int dend = (d0 + LEN < D) ? d0 + LEN : D;
  const bool has_l = x > 0, has_r = x + 1 < W;
  const unsigned kOut = 0x7fffffffu;       // out-of-range byte offset -> the load returns 0
  unsigned off[3];                         // in-plane byte offsets of the centre column in rows y + 1, y, y - 1 (kOut outside the image)
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = y - dy + 1;
    off[dy] = (inr && yy >= 0 && yy < H) ? (unsigned)(yy * W + x) * 4u : kOut;
  }
  float acc[LEN];
#pragma unroll
  for (int i = 0; i < LEN; ++i) acc[i] = 0.f;
#pragma unroll 1
  for (int f0 = 0; f0 < NF; f0 += FG) {
    pnsfm_buf gbuf[FG];
#pragma unroll
    for (int u = 0; u < FG; ++u) gbuf[u] = pnsfm_make_buf(dout + ((size_t)blockIdx.z * NF + f0 + u) * DHW, (unsigned)DHW * 4u);
    const float* wf = w3 + f0 * 27;
#pragma unroll
    for (int t = 0; t < LEN + 2; ++t) {
      const int dd = d0 - 1 + t;           // this plane feeds the outputs d0 + t - 2 (dz = 0), d0 + t - 1 (dz = 1), d0 + t (dz = 2)
      const bool dok = dd >= 0 && dd < D && dd <= dend;
      const unsigned plane = dok ? (unsigned)dd * (unsigned)HW * 4u : 0u;
      float g[FG][9];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const unsigned vo = (dok && off[dy] != kOut) ? plane + off[dy] : kOut;
#pragma unroll
        for (int u = 0; u < FG; ++u) g[u][dy * 3 + 1] = pnsfm_buf_load(gbuf[u], vo, 0u);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int u = 0; u < FG; ++u) {
          const float c = g[u][dy * 3 + 1];
          const float r = __shfl_down(c, 1), l = __shfl_up(c, 1);
          g[u][dy * 3 + 0] = has_r ? r : 0.f;       // dx = 0: xx = x + 1
          g[u][dy * 3 + 2] = has_l ? l : 0.f;       // dx = 2: xx = x - 1
        }
#pragma unroll
      for (int u = 0; u < FG; ++u)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          if (t >= 2) acc[t >= 2 ? t - 2 : 0] = fmaf(wf[u * 27 + tp], g[u][tp], acc[t >= 2 ? t - 2 : 0]);
          if (t >= 1 && t <= LEN) acc[(t >= 1 && t <= LEN) ? t - 1 : 0] = fmaf(wf[u * 27 + 9 + tp], g[u][tp], acc[(t >= 1 && t <= LEN) ? t - 1 : 0]);
          if (t < LEN) acc[t < LEN ? t : 0] = fmaf(wf[u * 27 + 18 + tp], g[u][tp], acc[t < LEN ? t : 0]);
        }
    }
  }
  float* ob = dp + (size_t)blockIdx.z * DHW + pix;
#pragma unroll
  for (int i = 0; i < LEN; ++i)
    if (active && d0 + i < dend) ob[(size_t)(d0 + i) * HW] = acc[i];
}

// dw3[f][tap] = sum_{b,d,y,x} dout[b][f*D+d][y][x] * p[b][d+dz-1][y+dy-1][x+dx-1];  db3[f] = sum dout[b][f*D+d][y][x]
// A reduction of 8 x 28 numbers over every voxel; 216 FMAs per voxel, so the floor is the fp32 VALU rate (43 us for
// unpack1), close to the HBM time of reading dout once (35 us).  (An MFMA formulation -- A = dout[f], B = the 27 shifted
// copies of p -- needs one gathered operand load per lane per 2 voxels and measured 217 us: the taps of one B fragment
// touch 9 different rows.)  Same column-sliding scheme as the data gradient: a thread owns (b, 4 of the 8 features, a
// run of `len` consecutive d, one (y, x)); it keeps the three p planes d-1, d, d+1 (27 registers, rotated by a 3x
// unrolled loop instead of moves), loads 9 new p values + 4 dout values per step, and accumulates 4 x 27 products (+ 4
// bias sums) in registers.  Lanes run along x, so every load is coalesced.  The 112 per-thread partials are reduced
// through LDS 16 at a time (conflict-free padded rows, then a 16-lane shuffle), one partial slot per value per block.
// block reduction of a thread's 4 x 28 partials (27 taps + bias per feature), kW3Pass values per pass; one slot per (block, value):
// plain stores, no zero-fill; conv3d_wgrad_finish_kernel adds the blocks in a fixed order (round 3: fp64 atomics)
constexpr int kW3Pass = 16, kW3Row = 256 + 16;
template <int FPB>
__device__ __forceinline__ void conv3d_wgrad_block_reduce(const float (&acc)[FPB][27], const float (&bs)[FPB], float* red,
                                                          float* __restrict__ part, int fg) {
  const int tid = threadIdx.x;
  const int rv = tid >> 4, rj = tid & 15;
  constexpr int NV = FPB * 28;
#pragma unroll
  for (int pass = 0; pass < (NV + kW3Pass - 1) / kW3Pass; ++pass) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kW3Pass; ++u) {
      const int v = pass * kW3Pass + u;      // v = f*28 + t  (t = 27: bias)
      const int f = v < NV ? v / 28 : 0, t = v < NV ? v - f * 28 : 0;
      red[u * kW3Row + tid] = (t < 27) ? acc[f][t < 27 ? t : 0] : bs[f];
    }
    __syncthreads();
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc += red[rv * kW3Row + rj + 16 * i];
    sacc += __shfl_xor(sacc, 8);
    sacc += __shfl_xor(sacc, 4);
    sacc += __shfl_xor(sacc, 2);
    sacc += __shfl_xor(sacc, 1);
    if (rj == 0 && pass * kW3Pass + rv < NV)
      part[(((size_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + fg) * NV + pass * kW3Pass + rv] = sacc;
  }
}

__global__ void __launch_bounds__(256) conv3d_wgrad_kernel(const float* __restrict__ p, const float* __restrict__ dout,
                                                            float* __restrict__ part, int D, int H, int W, int len, int NF) {
  __shared__ float red[kW3Pass * kW3Row];
  const int tid = threadIdx.x;
  const int HW = H * W, DHW = D * HW;
  const int nchunk = (D + len - 1) / len;
  const int idx = blockIdx.x * 256 + tid;
  const bool active = idx < nchunk * HW;
  const int chunk = active ? idx / HW : 0;
  const int pix = active ? idx - chunk * HW : 0;
  const int y = pix / W, x = pix - y * W;
  const int fg = blockIdx.y, b = blockIdx.z;
  const int d0 = chunk * len;
  const int dend = (d0 + len < D) ? d0 + len : D;
  const float* pb = p + (size_t)b * DHW;
  const float* gb = dout + ((size_t)b * NF + fg * 4) * DHW + pix;
  int off[9];
  bool okp[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int yy = y + dy - 1, xx = x + dx - 1;
      okp[dy * 3 + dx] = active && yy >= 0 && yy < H && xx >= 0 && xx < W;
      off[dy * 3 + dx] = yy * W + xx;
    }
  auto load_plane = [&](float (&q)[9], int dd) {
    const bool dok = dd >= 0 && dd < D;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* src = (dok && okp[t]) ? pb + ((size_t)dd * HW + off[t]) : pnsfm_zero_page3;
      q[t] = *src;
    }
  };
  float acc[4][27], bs[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    bs[f] = 0.f;
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[f][t] = 0.f;
  }
  auto step = [&](int d, const float (&q0)[9], const float (&q1)[9], const float (&q2)[9]) {
    const bool ok = active && d < dend;
    const float* src = ok ? gb + (size_t)d * HW : pnsfm_zero_page3;
    const unsigned fstride = ok ? (unsigned)DHW : 0u;
    float g[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) g[f] = src[f * fstride];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      bs[f] += g[f];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        acc[f][t] = fmaf(g[f], q0[t], acc[f][t]);
        acc[f][9 + t] = fmaf(g[f], q1[t], acc[f][9 + t]);
        acc[f][18 + t] = fmaf(g[f], q2[t], acc[f][18 + t]);
      }
    }
  };
  float pa[9], pc[9], pd[9];
  load_plane(pa, d0 - 1);
  load_plane(pc, d0);
  for (int d = d0; d < dend; d += 3) {
    load_plane(pd, d + 1);
    step(d, pa, pc, pd);
    load_plane(pa, d + 2);
    step(d + 1, pc, pd, pa);
    load_plane(pc, d + 3);
    step(d + 2, pd, pa, pc);
  }
  conv3d_wgrad_block_reduce<4>(acc, bs, red, part, fg);
}

// Round 4: the kernel above uses every load in the step that issues it, computes thirteen 64-bit addresses (and their zero-page
// selects) per step, spends two v_mov per v_pk_fma_f32 on building operand pairs (27 taps per feature: the pairs straddle planes
// and features) and sits at 246 VGPRs: two waves per SIMD, each waiting out a full memory latency per plane of the dout stream
// (unpack1: 83 us against ~22 us of FMAs).  Same arithmetic here with the data gradient's addressing: lanes run along the flattened
// (chunk, pixel) index, 62 outputs per wave, buffer loads through wave-uniform descriptors (zero padding = an out-of-range
// offset), a thread LOADS only the centre column of the three rows of a p plane and takes x - 1 / x + 1 from its neighbour lanes
// (halo lanes 0 and 63 read dout as zero and therefore add nothing).  A plane's nine values are held as FIVE register pairs -- the
// tenth element is 1 in the dz = 0 plane, 0 elsewhere -- and so are a feature's accumulators (3 x 5 pairs): every FMA is one half
// of a v_pk_fma_f32 on naturally aligned pairs, and the tenth accumulator of the dz = 0 row is the bias gradient.  The centre
// triples of the next PF planes and the dout values of the next PF steps are in flight in rings whose slots are compile-time (PF
// steps per trip, PF a multiple of 3: the three expanded planes d - 1, d, d + 1 rotate with the same period).
typedef float c3d_f32x2 __attribute__((ext_vector_type(2)));
template <int PF, int FPB>      // planes / steps in flight; features per workgroup (the grid's y extent is NF / FPB)
__global__ void __launch_bounds__(256, 2) conv3d_wgrad_ring_kernel(const float* __restrict__ p, const float* __restrict__ dout,
                                                                    float* __restrict__ part, int D, int H, int W, int len, int NF,
                                                                    int xmap) {
  __shared__ float red[kW3Pass * kW3Row];
  const int HW = H * W, DHW = D * HW;
  const int nchunk = (D + len - 1) / len;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned bx = xmap ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const long lin = ((long)bx * 4 + wave) * 62 + lane - 1;      // flattened (chunk, pixel) index of this lane
  const bool inr = lin >= 0 && lin < (long)nchunk * HW;
  const bool active = inr && lane >= 1 && lane <= 62;
  const int chunk = inr ? (int)(lin / HW) : 0;
  const int pix = inr ? (int)(lin - (long)chunk * HW) : 0;
  const int y = pix / W, x = pix - y * W;
  const int fg = blockIdx.y, b = blockIdx.z;
  const int d0 = chunk * len;
  const int dend = (d0 + len < D) ? d0 + len : D;
  const bool has_l = x > 0, has_r = x + 1 < W;
  const unsigned kOut = 0x7fffffffu;       // out-of-range byte offset -> the load returns 0
  const pnsfm_buf pbuf = pnsfm_make_buf(p + (size_t)b * DHW, (unsigned)DHW * 4u);
  pnsfm_buf gbuf[FPB];
#pragma unroll
  for (int f = 0; f < FPB; ++f) gbuf[f] = pnsfm_make_buf(dout + ((size_t)b * NF + fg * FPB + f) * DHW, (unsigned)DHW * 4u);
  unsigned off[3];                         // in-plane byte offsets of the centre column in rows y - 1, y, y + 1 (kOut outside the image)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int yy = y + r - 1;
    off[r] = (inr && yy >= 0 && yy < H) ? (unsigned)(yy * W + x) * 4u : kOut;
  }
  const unsigned goff = active ? (unsigned)pix * 4u : kOut;
  // plane k of this thread's run is dd = d0 - 1 + k; step s is d = d0 + s and uses the planes s, s + 1, s + 2
  auto load_centres = [&](float (&c)[3], int k) {
    const int dd = d0 - 1 + k;
    const bool dok = dd >= 0 && dd < D;
    const unsigned plane = dok ? (unsigned)dd * (unsigned)HW * 4u : 0u;
#pragma unroll
    for (int r = 0; r < 3; ++r) c[r] = pnsfm_buf_load(pbuf, (dok && off[r] != kOut) ? plane + off[r] : kOut, 0u);
  };
  // nine taps t = 3 * row + dx (dx = 0: x - 1) as pairs (t0,t1) (t2,t3) (t4,t5) (t6,t7) (t8, pad)
  auto expand = [&](c3d_f32x2 (&q)[5], const float (&c)[3]) {
    float v[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float lft = __shfl_up(c[r], 1), rgt = __shfl_down(c[r], 1);
      v[r * 3 + 0] = has_l ? lft : 0.f;
      v[r * 3 + 1] = c[r];
      v[r * 3 + 2] = has_r ? rgt : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { q[i].x = v[2 * i]; q[i].y = v[2 * i + 1]; }
    q[4].x = v[8];
    q[4].y = 0.f;
  };
  auto load_g = [&](float (&g)[FPB], int sidx) {
    const int d = d0 + sidx;
    const unsigned vo = (goff != kOut && d < dend) ? (unsigned)d * (unsigned)HW * 4u + goff : kOut;
#pragma unroll
    for (int f = 0; f < FPB; ++f) g[f] = pnsfm_buf_load(gbuf[f], vo, 0u);
  };
  c3d_f32x2 acc[FPB][3][5];                  // [feature][dz][pair]; acc[f][0][4].y collects the bias gradient
#pragma unroll
  for (int f = 0; f < FPB; ++f)
#pragma unroll
    for (int z = 0; z < 3; ++z)
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[f][z][i] = (c3d_f32x2)(0.f);
  c3d_f32x2 Q[3][5];
  float C[PF][3], G[PF][FPB];
  {
    float c0[3], c1[3], c2[3];
    load_centres(c0, 0);
    load_centres(c1, 1);
    load_centres(c2, 2);
#pragma unroll
    for (int u = 0; u < PF; ++u) load_centres(C[u], 3 + u);
#pragma unroll
    for (int u = 0; u < PF; ++u) load_g(G[u], u);
    expand(Q[0], c0);
    expand(Q[1], c1);
    expand(Q[2], c2);
  }
#pragma unroll 1
  for (int s0 = 0; s0 < len; s0 += PF) {        // len is a multiple of PF (host); steps past the run's end read dout as zero
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      c3d_f32x2 q0[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) q0[i] = Q[u % 3][i];
      q0[4].y = 1.f;                            // the dz = 0 row's tenth column sums dout: the bias gradient
      const c3d_f32x2 (&q1)[5] = Q[(u + 1) % 3];
      const c3d_f32x2 (&q2)[5] = Q[(u + 2) % 3];
#pragma unroll
      for (int f = 0; f < FPB; ++f) {
        const c3d_f32x2 g = (c3d_f32x2)(G[u][f]);
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          acc[f][0][i] = g * q0[i] + acc[f][0][i];
          acc[f][1][i] = g * q1[i] + acc[f][1][i];
          acc[f][2][i] = g * q2[i] + acc[f][2][i];
        }
      }
      expand(Q[u % 3], C[u]);                   // plane s + 3 replaces plane s
      load_centres(C[u], s0 + u + 3 + PF);
      load_g(G[u], s0 + u + PF);
    }
  }
  float accf[FPB][27], bs[FPB];
#pragma unroll
  for (int f = 0; f < FPB; ++f) {
    bs[f] = acc[f][0][4].y;
#pragma unroll
    for (int z = 0; z < 3; ++z)
#pragma unroll
      for (int t = 0; t < 9; ++t) accf[f][z * 9 + t] = (t & 1) ? acc[f][z][t >> 1].y : acc[f][z][t >> 1].x;
  }
  conv3d_wgrad_block_reduce<FPB>(accf, bs, red, part, fg);
}

// dw3 / db3 entry (f, t) = sum over the blocks' partials (fp64), one wave per entry: lane l adds blocks l, l + 64, ... in order, the
// lane sums meet in a fixed shuffle tree -- bit-reproducible
__global__ void __launch_bounds__(64) conv3d_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw3,
                                                                 float* __restrict__ db3, int nblk, int ngroups, int fpb) {
  const int i = blockIdx.x, lane = threadIdx.x;             // i = f * 28 + t over all NF features; fpb features per group
  const int f = i / 28, t = i - f * 28, fg = f / fpb, v = (f - fg * fpb) * 28 + t;
  double s = 0.0;
  for (int p = lane; p < nblk; p += 64) s += (double)part[((size_t)p * ngroups + fg) * (fpb * 28) + v];
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
  if (lane == 0) { if (t < 27) dw3[f * 27 + t] = (float)s; else db3[f] = (float)s; }
}

static int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_space_to_depth_strided(const float* x, float* y, int B, int C, int H, int W, size_t x_batch_stride, void* stream) {
  if ((H & 1) || (W & 1)) { set_error("space_to_depth: H, W must be even (got %d x %d)", H, W); return -1; }
  if (x_batch_stride < (size_t)C * H * W) { set_error("space_to_depth: batch stride smaller than one image"); return -1; }
  const size_t total = (size_t)B * C * H * W;
  const bool al16 = (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && x_batch_stride % 4 == 0;
  if (W % 8 == 0 && al16 && total / 16 < (1ull << 31)) {
    const unsigned total_q = (unsigned)(total / 16);      // one thread per 2 x 8 input block
    PNSFM_LAUNCH(s2d_v4_kernel, dim3((total_q + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, total_q, x_batch_stride);
  } else {
    PNSFM_LAUNCH(s2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, total, x_batch_stride);
  }
  return check_launch("space_to_depth");
}

int pnsfm_space_to_depth(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  return pnsfm_space_to_depth_strided(x, y, B, C, H, W, (size_t)C * H * W, stream);
}

int pnsfm_depth_to_space(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  const size_t total = (size_t)B * C * 4 * H * W;
  if (W % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && total / 16 < (1ull << 31)) {
    const unsigned total_q = (unsigned)(total / 16);      // one thread per four 4-wide channel rows
    PNSFM_LAUNCH(d2s_v4_kernel, dim3((total_q + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, total_q);
  } else {
    PNSFM_LAUNCH(d2s_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, total);
  }
  return check_launch("depth_to_space");
}

static bool nf_ok(int NF, const char* what) {
  if (NF == 4 || NF == 8) return true;
  set_error("%s: the gfx950 stencil kernels are built for 4 or 8 3-D feature maps (got %d)", what, NF);
  return false;
}

int pnsfm_conv3d_forward(const float* p, const float* w3, const float* b3, float* out, int B, int D, int H, int W, int NF,
                         void* stream) {
  if (!nf_ok(NF, "conv3d_forward")) return -1;
  if (W % 4 == 0 && (((uintptr_t)p | (uintptr_t)out) & 15) == 0) {      // four voxels per thread
    const dim3 gq(ceil_div(D * H * (W / 4), 256), 1, B);
    if (NF == 8) PNSFM_LAUNCH((conv3d_fwd_x4_kernel<8>), gq, dim3(256), 0, (hipStream_t)stream, p, w3, b3, out, D, H, W);
    else PNSFM_LAUNCH((conv3d_fwd_x4_kernel<4>), gq, dim3(256), 0, (hipStream_t)stream, p, w3, b3, out, D, H, W);
    return check_launch("conv3d_forward");
  }
  const dim3 grid(ceil_div(D * H * W, 256), 1, B);
  if (NF == 8) PNSFM_LAUNCH((conv3d_fwd_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, p, w3, b3, out, D, H, W);
  else PNSFM_LAUNCH((conv3d_fwd_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, p, w3, b3, out, D, H, W);
  return check_launch("conv3d_forward");
}

int pnsfm_conv3d_backward_data(const float* dout, const float* w3, float* dp, int B, int D, int H, int W, int NF, void* stream) {
  if (!nf_ok(NF, "conv3d_backward_data")) return -1;
  if ((size_t)D * H * W * 4 >= 0x7fffffffull) { set_error("conv3d_backward_data: feature slab exceeds the 2 GiB buffer window"); return -1; }
  // run length along d: 8 (25 % halo planes) when that still gives every CU a few blocks, shorter for small volumes
  int len = D < 8 ? D : 8;
  while (len > 2 && (long)B * ceil_div(D, len) * H * W < 2L * 256 * 256) len = ceil_div(len, 2);
  if (const char* e = getenv("PNSFM_CONV3D_LEN")) {          // tests: pin the run length (small volumes on the 8- / 4-plane kernels)
    const int v = atoi(e);
    if (v >= 1 && v <= 8) len = v;
  }
  const dim3 grid(ceil_div(ceil_div(ceil_div(D, len) * H * W, 62), 4), 1, B);      // 62 outputs per wave (two halo lanes)
  static const int xm = [] { const char* e = getenv("PNSFM_STENCIL_XCD_MAP"); return (e && e[0] == '0') ? 0 : 1; }();      // A/B switch
  const int xmap = (block_map_mode() >= 2 && xm) ? 1 : 0;
  static const int col = [] { const char* e = getenv("PNSFM_CONV3D_DGRAD_COL"); return (e && e[0] == '0') ? 0 : 1; }();      // A/B switch
  if (col && len == 8) {
    if (NF == 8) PNSFM_LAUNCH((conv3d_dgrad_col_kernel<8, 8>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, xmap);
    else PNSFM_LAUNCH((conv3d_dgrad_col_kernel<4, 8>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, xmap);
  } else if (col && len == 4) {
    if (NF == 8) PNSFM_LAUNCH((conv3d_dgrad_col_kernel<8, 4>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, xmap);
    else PNSFM_LAUNCH((conv3d_dgrad_col_kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, xmap);
  } else if (col && len == 2) {
    if (NF == 8) PNSFM_LAUNCH((conv3d_dgrad_col_kernel<8, 2>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, xmap);
    else PNSFM_LAUNCH((conv3d_dgrad_col_kernel<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, xmap);
  }
  else if (NF == 8) PNSFM_LAUNCH((conv3d_dgrad_kernel<8>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, len, xmap);
  else PNSFM_LAUNCH((conv3d_dgrad_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, len, xmap);
  return check_launch("conv3d_backward_data");
}

int pnsfm_conv3d_backward_weight(const float* p, const float* dout, float* dw3, float* db3, double* ws, int B, int D, int H,
                                 int W, int NF, void* stream) {
  if (!nf_ok(NF, "conv3d_backward_weight")) return -1;
  hipStream_t s = (hipStream_t)stream;
  (void)ws;     // (round 3's zero-filled fp64 atomics target; the per-block partials now live in the stream's scratch buffer)
  // run length along d per thread: as long as possible (amortises the block reduction) while the grid still gives every
  // CU a few blocks
  int len = D;
  while (len > 12 && (long)B * (NF / 4) * ceil_div(D, len) * H * W < 4L * 256 * 256) len = ceil_div(len, 2);
  // A/B switch: PNSFM_CONV3D_WGRAD_RING=0 = round 3's kernel.  (Measured and dropped, profiles/r04_ab_conv3d_wgrad_ring.txt: two
  // features per workgroup with rings of 3 / 6 -- 154 / 209 VGPRs -- are slower, 340 / 375 us over the step's volumes against 323:
  // the per-plane work is shared by half as many FMAs.)
  const int ring = [] { const char* e = getenv("PNSFM_CONV3D_WGRAD_RING"); return (e && e[0] == '0') ? 0 : 34; }();
  const int pf = ring / 10, fpb = ring ? ring % 10 : 4;
  if (ring && (size_t)D * H * W * 4 >= 0x7fffffffull) { set_error("conv3d_backward_weight: feature slab exceeds the 2 GiB buffer window"); return -1; }
  len = ring ? ceil_div(len, pf) * pf : ceil_div(len, 3) * 3;       // whole trips of the unrolled loop
  const dim3 grid(ring ? ceil_div(ceil_div(ceil_div(D, len) * H * W, 62), 4) : ceil_div(ceil_div(D, len) * H * W, 256), NF / fpb, B);
  ScratchLease lease(s, (size_t)grid.x * grid.y * grid.z * fpb * 28 * sizeof(float));
  float* const part = lease.as<float>();
  if (!part) return -1;
  const int xmap = block_map_mode() >= 2 ? 1 : 0;
  if (ring == 34) PNSFM_LAUNCH((conv3d_wgrad_ring_kernel<3, 4>), grid, dim3(256), 0, s, p, dout, part, D, H, W, len, NF, xmap);
  else PNSFM_LAUNCH(conv3d_wgrad_kernel, grid, dim3(256), 0, s, p, dout, part, D, H, W, len, NF);
  int e = check_launch("conv3d_backward_weight");
  if (e) return e;
  PNSFM_LAUNCH(conv3d_wgrad_finish_kernel, dim3(NF * 28), dim3(64), 0, s, (const float*)part, dw3, db3, (int)(grid.x * grid.z),
               (int)grid.y, fpb);
  return check_launch("conv3d_backward_weight_finish");
}

// PackNet01's fixed d = 8 (the original entry points)
int pnsfm_conv3d_1to8_forward(const float* p, const float* w3, const float* b3, float* out, int B, int D, int H, int W,
                              void* stream) {
  return pnsfm_conv3d_forward(p, w3, b3, out, B, D, H, W, 8, stream);
}
int pnsfm_conv3d_1to8_backward_data(const float* dout, const float* w3, float* dp, int B, int D, int H, int W, void* stream) {
  return pnsfm_conv3d_backward_data(dout, w3, dp, B, D, H, W, 8, stream);
}
int pnsfm_conv3d_1to8_backward_weight(const float* p, const float* dout, float* dw3, float* db3, double* ws, int B, int D,
                                      int H, int W, void* stream) {
  return pnsfm_conv3d_backward_weight(p, dout, dw3, db3, ws, B, D, H, W, 8, stream);
}

}  // extern "C"
