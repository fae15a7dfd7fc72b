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
                                                   int C, int H, int W, size_t total) {
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
    y[idx] = x[(((size_t)b * C + c) * H + 2 * h + i) * W + 2 * w + j];
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

// out[b][f*D+d][y][x] = b3[f] + sum_{dz,dy,dx} w3[f][dz][dy][dx] * p[b][d+dz-1][y+dy-1][x+dx-1]
// grid: (ceil(D*HW/256), 1, B): one thread per voxel (d, y, x) -- flattened so that small planes (the 7x7 / 5x5 weight
// volumes of the kernel composition, or 6x20 feature maps) still fill every lane -- produces all 8 features;
// 27 branch-free neighbour loads (L1/L2 serve the overlap between neighbouring threads), 216 FMAs, 8 stores.
__global__ void __launch_bounds__(256) conv3d_fwd_kernel(const float* __restrict__ p, const float* __restrict__ w3,
                                                          const float* __restrict__ b3, float* __restrict__ out,
                                                          int D, int H, int W) {
  __shared__ float ws[8 * 27 + 8];
  for (int i = threadIdx.x; i < 8 * 27 + 8; i += 256) ws[i] = i < 216 ? w3[i] : b3[i - 216];
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
  float acc[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) acc[f] = ws[216 + f];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap)
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = fmaf(ws[f * 27 + tap], v[tap], acc[f]);
  if (active) {
    float* ob = out + (size_t)b * 8 * DHW + vox;
#pragma unroll
    for (int f = 0; f < 8; ++f) ob[(size_t)f * DHW] = acc[f];
  }
}

// dp[b][d][y][x] = sum_{f,dz,dy,dx} w3[f][dz][dy][dx] * dout[b][f*D + d-dz+1][y-dy+1][x-dx+1]
// same thread mapping as the forward stencil; the 8x9 loads of one dz slab are issued together (unconditional loads,
// pointer-selected against the zero page) before they are consumed.  (A variant that register-blocks 4 consecutive x per
// thread to halve the loads measured slower -- 1.66 vs 1.14 ms per step -- and was dropped.)
__global__ void __launch_bounds__(256) conv3d_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w3,
                                                            float* __restrict__ dp, int D, int H, int W) {
  __shared__ float ws[8 * 27];
  for (int i = threadIdx.x; i < 8 * 27; i += 256) ws[i] = w3[i];
  __syncthreads();
  const int HW = H * W, DHW = D * HW;
  const int vox = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.z;
  const bool active = vox < DHW;
  const int d = active ? vox / HW : 0;
  const int pix = active ? vox - d * HW : 0;
  const int y = pix / W, x = pix - y * W;
  const float* gb = dout + (size_t)b * 8 * DHW;
  float acc = 0.f;
#pragma unroll 1
  for (int dz = 0; dz < 3; ++dz) {
    const int dd = d - dz + 1;
    const bool dok = active && dd >= 0 && dd < D;
    float g[8][9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y - dy + 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = x - dx + 1;
        const bool ok = dok && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float* src = ok ? gb + (dd * HW + yy * W + xx) : pnsfm_zero_page3;
        const size_t fstride = ok ? (size_t)DHW : 0;
#pragma unroll
        for (int f = 0; f < 8; ++f) g[f][dy * 3 + dx] = src[f * fstride];
      }
    }
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = fmaf(ws[f * 27 + dz * 9 + t], g[f][t], acc);
  }
  if (active) dp[(size_t)b * DHW + vox] = acc;
}

// dw3[f][tap] = sum_{b,d,y,x} dout[b][f*D+d][y][x] * p[b][d+dz-1][y+dy-1][x+dx-1];  db3[f] = sum dout[b][f*D+d][y][x]
// A reduction of 8 x 28 numbers over every voxel -- done on the matrix cores: per v_mfma_f32_32x32x2_f32,
//   A[m = feature f (8 of 32 rows used)][k = 2 consecutive voxels]  = dout[f][voxel]
//   B[k][n = tap (27 of 32 columns) | n = 27: a column of ones]     = p[voxel + offset(tap)]  (zero outside the volume)
// so D[f][tap] accumulates dw3 and D[f][27] accumulates db3.  Operands come straight from global/L1 (neighbouring voxels
// share cache lines); each wave walks whole (b, d, y) rows so no index division sits in the x loop.  No LDS staging, no
// barriers; one LDS reduction over the block's 4 waves at the end, then 224 fp64 atomics per block.
__global__ void __launch_bounds__(256) conv3d_wgrad_kernel(const float* __restrict__ p, const float* __restrict__ dout,
                                                            double* __restrict__ ws, int B, int D, int H, int W) {
  __shared__ float red[4][8 * 28];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
  const int HW = H * W;
  const bool fa = l32 < 8;                  // A row = feature
  const bool tb = l32 < 27, ones = l32 == 27;   // B column = tap / ones column
  const int dz = l32 / 9, dy = (l32 - dz * 9) / 3, dx = l32 - dz * 9 - dy * 3;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const long rows = (long)B * D * H;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const int y = (int)(row % H);
    const long bd = row / H;
    const int d = (int)(bd % D), b = (int)(bd / D);
    const float* grow = dout + (((size_t)b * 8 + (fa ? l32 : 0)) * D + d) * HW + (size_t)y * W;
    const int dd = d + dz - 1, yy = y + dy - 1;
    const bool rowok = tb && dd >= 0 && dd < D && yy >= 0 && yy < H;
    const float* prow = p + ((size_t)b * D + (rowok ? dd : 0)) * HW + (size_t)(rowok ? yy : 0) * W;
    for (int x0 = 0; x0 < W; x0 += 8) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = x0 + 2 * u + half;
        const bool okx = x < W;
        const float* sa = (fa && okx) ? grow + x : pnsfm_zero_page3;
        av[u] = *sa;
        const int xx = x + dx - 1;
        const bool okb = rowok && okx && xx >= 0 && xx < W;
        const float* sb = okb ? prow + xx : pnsfm_zero_page3;
        const float t = *sb;
        bv[u] = (ones && okx) ? 1.f : t;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = pnsfm_mfma_32x32x2(av[u], bv[u], acc);
    }
  }
  // D row = (r&3) + 8*(r>>2) + 4*half: rows 0..7 live in r = 0..3 -> feature f = r + 4*half; column = l32 (tap, 27 = bias)
  if (l32 < 28) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(r + 4 * half) * 28 + l32] = acc[r];
  }
  __syncthreads();
  if (tid < 8 * 28) {
    const double s = (double)red[0][tid] + (double)red[1][tid] + (double)red[2][tid] + (double)red[3][tid];
    atomicAdd(&ws[tid], s);
  }
}

__global__ void conv3d_wgrad_finish_kernel(const double* __restrict__ ws, float* __restrict__ dw3, float* __restrict__ db3) {
  const int i = threadIdx.x;
  if (i < 8 * 28) {
    const int f = i / 28, t = i - f * 28;
    if (t < 27) dw3[f * 27 + t] = (float)ws[i]; else db3[f] = (float)ws[i];
  }
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

int pnsfm_space_to_depth(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  if ((H & 1) || (W & 1)) { set_error("space_to_depth: H, W must be even (got %d x %d)", H, W); return -1; }
  const size_t total = (size_t)B * C * H * W;
  PNSFM_LAUNCH(s2d_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, total);
  return check_launch("space_to_depth");
}

int pnsfm_depth_to_space(const float* x, float* y, int B, int C, int H, int W, void* stream) {
  const size_t total = (size_t)B * C * 4 * H * W;
  PNSFM_LAUNCH(d2s_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W, total);
  return check_launch("depth_to_space");
}

int pnsfm_conv3d_1to8_forward(const float* p, const float* w3, const float* b3, float* out, int B, int D, int H, int W,
                              void* stream) {
  PNSFM_LAUNCH(conv3d_fwd_kernel, dim3(ceil_div(D * H * W, 256), 1, B), dim3(256), 0, (hipStream_t)stream, p, w3, b3, out, D, H, W);
  return check_launch("conv3d_forward");
}

int pnsfm_conv3d_1to8_backward_data(const float* dout, const float* w3, float* dp, int B, int D, int H, int W, void* stream) {
  PNSFM_LAUNCH(conv3d_dgrad_kernel, dim3(ceil_div(D * H * W, 256), 1, B), dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W);
  return check_launch("conv3d_backward_data");
}

int pnsfm_conv3d_1to8_backward_weight(const float* p, const float* dout, float* dw3, float* db3, double* ws, int B, int D,
                                      int H, int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int e = (int)hipMemsetAsync(ws, 0, 8 * 28 * sizeof(double), s);
  if (e) { set_error("conv3d_backward_weight: memset failed"); return e; }
  long nrows = (long)B * D * H;
  int nblk = (int)((nrows + 3) / 4);
  if (nblk > 2048) nblk = 2048;
  if (nblk < 1) nblk = 1;
  PNSFM_LAUNCH(conv3d_wgrad_kernel, dim3(nblk), dim3(256), 0, s, p, dout, ws, B, D, H, W);
  e = check_launch("conv3d_backward_weight");
  if (e) return e;
  PNSFM_LAUNCH(conv3d_wgrad_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, dw3, db3);
  return check_launch("conv3d_backward_weight_finish");
}

}  // extern "C"
