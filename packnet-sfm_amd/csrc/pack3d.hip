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
__global__ void __launch_bounds__(256) conv3d_fwd_kernel(const float* __restrict__ p, const float* __restrict__ w3,
                                                          const float* __restrict__ b3, float* __restrict__ out,
                                                          int D, int H, int W, size_t total) {
  __shared__ float ws[8 * 27 + 8];
  for (int i = threadIdx.x; i < 8 * 27 + 8; i += 256) ws[i] = i < 216 ? w3[i] : b3[i - 216];
  __syncthreads();
  const size_t HW = (size_t)H * W;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int x = (int)(idx % W);
    size_t r = idx / W;
    const int y = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    const int b = (int)(r / D);
    const float* pb = p + (size_t)b * D * HW;
    float acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = ws[216 + f];
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int dd = d + dz - 1;
      if (dd < 0 || dd >= D) continue;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = x + dx - 1;
          if (xx < 0 || xx >= W) continue;
          const float v = pb[(size_t)dd * HW + (size_t)yy * W + xx];
          const int tap = dz * 9 + dy * 3 + dx;
#pragma unroll
          for (int f = 0; f < 8; ++f) acc[f] = fmaf(ws[f * 27 + tap], v, acc[f]);
        }
      }
    }
    float* ob = out + (size_t)b * 8 * D * HW + (size_t)d * HW + (size_t)y * W + x;
#pragma unroll
    for (int f = 0; f < 8; ++f) ob[(size_t)f * D * HW] = acc[f];
  }
}

// dp[b][d][y][x] = sum_{f,dz,dy,dx} w3[f][dz][dy][dx] * dout[b][f*D + d-dz+1][y-dy+1][x-dx+1]
__global__ void __launch_bounds__(256) conv3d_dgrad_kernel(const float* __restrict__ dout, const float* __restrict__ w3,
                                                            float* __restrict__ dp, int D, int H, int W, size_t total) {
  __shared__ float ws[8 * 27];
  for (int i = threadIdx.x; i < 8 * 27; i += 256) ws[i] = w3[i];
  __syncthreads();
  const size_t HW = (size_t)H * W;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int x = (int)(idx % W);
    size_t r = idx / W;
    const int y = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    const int b = (int)(r / D);
    const float* gb = dout + (size_t)b * 8 * D * HW;
    float acc = 0.f;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int dd = d - dz + 1;
      if (dd < 0 || dd >= D) continue;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = y - dy + 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = x - dx + 1;
          if (xx < 0 || xx >= W) continue;
          const int tap = dz * 9 + dy * 3 + dx;
          const float* g = gb + (size_t)dd * HW + (size_t)yy * W + xx;
#pragma unroll
          for (int f = 0; f < 8; ++f) acc = fmaf(ws[f * 27 + tap], g[(size_t)f * D * HW], acc);
        }
      }
    }
    dp[idx] = acc;
  }
}

// dw3[f][tap] = sum_{b,d,y,x} dout[b][f*D+d][y][x] * p[b][d+dz-1][y+dy-1][x+dx-1];  db3[f] = sum dout[b][f*D+d][y][x]
// grid: (nblk, 8 features). ws: double[8*28] zeroed by the caller, accumulated with atomics.
__global__ void __launch_bounds__(256) conv3d_wgrad_kernel(const float* __restrict__ p, const float* __restrict__ dout,
                                                            double* __restrict__ ws, int D, int H, int W, size_t total) {
  __shared__ double red[4];
  const int f = blockIdx.y;
  const size_t HW = (size_t)H * W;
  float acc[28];
#pragma unroll
  for (int t = 0; t < 28; ++t) acc[t] = 0.f;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int x = (int)(idx % W);
    size_t r = idx / W;
    const int y = (int)(r % H);
    r /= H;
    const int d = (int)(r % D);
    const int b = (int)(r / D);
    const float g = dout[((size_t)b * 8 * D + (size_t)f * D + d) * HW + (size_t)y * W + x];
    const float* pb = p + (size_t)b * D * HW;
    acc[27] += g;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      const int dd = d + dz - 1;
      if (dd < 0 || dd >= D) continue;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = x + dx - 1;
          if (xx < 0 || xx >= W) continue;
          acc[dz * 9 + dy * 3 + dx] = fmaf(g, pb[(size_t)dd * HW + (size_t)yy * W + xx], acc[dz * 9 + dy * 3 + dx]);
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 28; ++t) {
    double v = (double)acc[t];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&ws[f * 28 + t], red[0] + red[1] + red[2] + red[3]);
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
  const size_t total = (size_t)B * D * H * W;
  PNSFM_LAUNCH(conv3d_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p, w3, b3, out, D, H, W, total);
  return check_launch("conv3d_forward");
}

int pnsfm_conv3d_1to8_backward_data(const float* dout, const float* w3, float* dp, int B, int D, int H, int W, void* stream) {
  const size_t total = (size_t)B * D * H * W;
  PNSFM_LAUNCH(conv3d_dgrad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dout, w3, dp, D, H, W, total);
  return check_launch("conv3d_backward_data");
}

int pnsfm_conv3d_1to8_backward_weight(const float* p, const float* dout, float* dw3, float* db3, double* ws, int B, int D,
                                      int H, int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  int e = (int)hipMemsetAsync(ws, 0, 8 * 28 * sizeof(double), s);
  if (e) { set_error("conv3d_backward_weight: memset failed"); return e; }
  const size_t total = (size_t)B * D * H * W;
  int nblk = grid_for(total);
  if (nblk > 512) nblk = 512;
  PNSFM_LAUNCH(conv3d_wgrad_kernel, dim3(nblk, 8), dim3(256), 0, s, p, dout, ws, D, H, W, total);
  e = check_launch("conv3d_backward_weight");
  if (e) return e;
  PNSFM_LAUNCH(conv3d_wgrad_finish_kernel, dim3(1), dim3(256), 0, s, (const double*)ws, dw3, db3);
  return check_launch("conv3d_backward_weight_finish");
}

}  // extern "C"
