// loss.hip -- the self-supervised photometric objective of PackNet-SfM as fused gfx950 kernels.
//
//  view_synthesis_{forward,backward}: inv2depth -> Camera.reconstruct -> Camera.project -> grid_sample
//      /root/reference/packnet_sfm/losses/multiview_photometric_loss.py:127-165
//      /root/reference/packnet_sfm/utils/depth.py:103-120, geometry/camera.py:72-80,112-191,
//      geometry/camera_utils.py:27-59 (bilinear, padding 'zeros', align_corners=True)
//  photometric_{forward,backward}: SSIM (:14-53), clamp((1-ssim)/2) (:169-186), 0.85/0.15 SSIM/L1 mix (:188-223,
//      clip_loss == 0), automask + per-pixel min / mean over candidates (:225-253, :321-334)
//  smoothness_{forward,backward}: utils/depth.py:165-198 + utils/image.py:85-113 + loss :276-278
//
// The reference runs this as ~1500 ATen launches moving 2.7 GB per image; here one scale is three launches
// forward and three backward, each HBM-bound: every pixel's geometry lives in registers, SSIM windows are
// staged through LDS tiles (reflect halo), and all scalar reductions are wave-shuffle -> LDS -> one fp64 atomic.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

__device__ __forceinline__ double block_sum_256d(double v, double* red) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// stage 2 of the scalar reductions: out[v] = sum_i part[i * nvals + v] (nvals <= 4), ONE workgroup, fixed order: thread t adds
// partials t, t + 256, ... and the 256 thread sums meet in a fixed tree (block_sum_256d).
__global__ void __launch_bounds__(256) sum_partials_kernel(const double* __restrict__ part, int n, int nvals, double* __restrict__ out) {
  __shared__ double red[4];
  for (int v = 0; v < nvals; ++v) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[(size_t)i * nvals + v];
    const double tot = block_sum_256d(acc, red);
    if (threadIdx.x == 0) out[v] = tot;
    __syncthreads();
  }
}

// the same second stage, finishing the scalar the caller wants: fout[0] = (float)(sum_v out[v] * scale[v]) -- the pixel mean of a loss
// map, or sum|Sx|/nx + sum|Sy|/ny -- so that no ATen launch is needed between the kernels and autograd (round 4)
__global__ void __launch_bounds__(256) sum_partials_scaled_kernel(const double* __restrict__ part, int n, int nvals, double s0, double s1,
                                                                  double* __restrict__ out, float* __restrict__ fout) {
  __shared__ double red[4];
  double acc_out = 0.0;
  for (int v = 0; v < nvals; ++v) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[(size_t)i * nvals + v];
    const double tot = block_sum_256d(acc, red);
    if (threadIdx.x == 0 && out) out[v] = tot;
    acc_out += tot * (v == 0 ? s0 : s1);
    __syncthreads();
  }
  if (threadIdx.x == 0) fout[0] = (float)acc_out;
}

struct Cam {
  float k[9];     // target intrinsics K (row major)
  float ki[9];    // Kinv as the reference builds it (camera.py:72-80): K with 4 entries replaced
  float rk[9];    // context ("ref") intrinsics
};

__device__ __forceinline__ void load_cam(const float* K, const float* refK, int b, Cam& c) {
#pragma unroll
  for (int i = 0; i < 9; ++i) { c.k[i] = K[b * 9 + i]; c.rk[i] = refK[b * 9 + i]; c.ki[i] = c.k[i]; }
  const float fx = c.k[0], fy = c.k[4], cx = c.k[2], cy = c.k[5];
  c.ki[0] = 1.f / fx;
  c.ki[4] = 1.f / fy;
  c.ki[2] = -1.f * cx / fx;
  c.ki[5] = -1.f * cy / fy;
}

struct Proj {
  float X[3];      // 3-D point in the target camera frame
  float ray[3];    // Kinv * [u, v, 1]
  float d;         // depth
  float Xr[3];     // point in the context camera frame
  float p[3];      // refK * Xr
  float z;         // clamped p.z
  float ix, iy;    // un-normalised sampling coordinates
};

__device__ __forceinline__ void project_pixel(const Cam& c, const float* T, float rho, int u, int v, int H, int W, Proj& q) {
  q.d = 1.f / fmaxf(rho, 1e-6f);
  const float fu = (float)u, fv = (float)v;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    q.ray[i] = c.ki[3 * i + 0] * fu + c.ki[3 * i + 1] * fv + c.ki[3 * i + 2];
    q.X[i] = q.ray[i] * q.d;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) q.Xr[i] = T[4 * i + 0] * q.X[0] + T[4 * i + 1] * q.X[1] + T[4 * i + 2] * q.X[2] + T[4 * i + 3];
#pragma unroll
  for (int i = 0; i < 3; ++i) q.p[i] = c.rk[3 * i + 0] * q.Xr[0] + c.rk[3 * i + 1] * q.Xr[1] + c.rk[3 * i + 2] * q.Xr[2];
  q.z = fmaxf(q.p[2], 1e-5f);
  const float xn = 2.f * (q.p[0] / q.z) / (float)(W - 1) - 1.f;
  const float yn = 2.f * (q.p[1] / q.z) / (float)(H - 1) - 1.f;
  q.ix = ((xn + 1.f) * 0.5f) * (float)(W - 1);
  q.iy = ((yn + 1.f) * 0.5f) * (float)(H - 1);
}

// grid_sample padding modes (align_corners=True), ATen's grid_sampler_compute_source_index_set_grad:
//   0 zeros      : coordinates are used as they are, out-of-image corners contribute 0
//   1 border     : clip to [0, size-1]; d(clipped)/d(coordinate) = 0 outside
//   2 reflection : reflect about 0 and size-1 (period 2(size-1)), then clip; the multiplier carries the reflection's sign
__device__ __forceinline__ float pad_coordinate(float x, int size, int mode, float& mult) {
  mult = 1.f;
  if (mode == 0) return x;
  const float hi = (float)(size - 1);
  if (mode == 2) {
    if (hi <= 0.f) { mult = 0.f; return 0.f; }
    float m = 1.f;
    if (x < 0.f) { x = -x; m = -1.f; }
    const float extra = fmodf(x, hi);
    const int flips = (int)floorf(x / hi);
    if (flips & 1) { x = hi - extra; m = -m; } else { x = extra; }
    mult = m;
  }
  if (x <= 0.f) { mult = 0.f; return 0.f; }      // clip (ATen: gradient 0 at and beyond the border)
  if (x >= hi) { mult = 0.f; return hi; }
  return x;
}

// The bilinear sample of the three channels of context image (j, b) at the projection of target pixel (u, v): the body of
// view_synthesis_fwd_kernel.  (Round 5 also called it from photometric kernels that computed the warped images in their tile loaders;
// measured -0.3 % twice, profiles/r05_ab_loss_fuse.txt, removed in round 6.)
__device__ __forceinline__ void warp_sample3(const Cam& cam, const float* Tm, float rho, int u, int v, int H, int W, int pad_mode,
                                             const float* __restrict__ rb, float (&out)[3]) {
  const int HW = H * W;
  Proj q;
  project_pixel(cam, Tm, rho, u, v, H, W, q);
  out[0] = out[1] = out[2] = 0.f;
  float mx, my;
  q.ix = pad_coordinate(q.ix, W, pad_mode, mx);
  q.iy = pad_coordinate(q.iy, H, pad_mode, my);
  if (q.ix > -1.f && q.ix < (float)W && q.iy > -1.f && q.iy < (float)H) {
    const float fx0 = floorf(q.ix), fy0 = floorf(q.iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float ax = q.ix - fx0, ay = q.iy - fy0;
    const float wnw = (1.f - ax) * (1.f - ay), wne = ax * (1.f - ay), wsw = (1.f - ax) * ay, wse = ax * ay;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* rc = rb + (size_t)c * HW;
      float acc = 0.f;
      if (vx0 && vy0) acc += rc[y0 * W + x0] * wnw;
      if (vx1 && vy0) acc += rc[y0 * W + x1] * wne;
      if (vx0 && vy1) acc += rc[y1 * W + x0] * wsw;
      if (vx1 && vy1) acc += rc[y1 * W + x1] * wse;
      out[c] = acc;
    }
  }
}

// Backward of warp_sample3 for one target pixel and one context image: g[c] = d loss / d warped[c] -> the pixel's contribution to
// d loss / d inv_depth (returned) and to the 12 entries of d loss / d T[:3, :4] (gT): the body of view_synthesis_bwd_kernel.
__device__ __forceinline__ float warp_backward3(const Cam& cam, const float* Tm, float rho, int u, int v, int H, int W, int pad_mode,
                                                const float* __restrict__ rb, const float (&g)[3], float (&gT)[12]) {
  const int HW = H * W;
  Proj q;
  project_pixel(cam, Tm, rho, u, v, H, W, q);
  float gix = 0.f, giy = 0.f;
  float mx, my;
  q.ix = pad_coordinate(q.ix, W, pad_mode, mx);
  q.iy = pad_coordinate(q.iy, H, pad_mode, my);
  if (q.ix > -1.f && q.ix < (float)W && q.iy > -1.f && q.iy < (float)H) {
    const float fx0 = floorf(q.ix), fy0 = floorf(q.iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float ax = q.ix - fx0, ay = q.iy - fy0;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* rc = rb + (size_t)c * HW;
      const float nw = (vx0 && vy0) ? rc[y0 * W + x0] : 0.f;
      const float ne = (vx1 && vy0) ? rc[y0 * W + x1] : 0.f;
      const float sw = (vx0 && vy1) ? rc[y1 * W + x0] : 0.f;
      const float se = (vx1 && vy1) ? rc[y1 * W + x1] : 0.f;
      gix += g[c] * ((ne - nw) * (1.f - ay) + (se - sw) * ay);
      giy += g[c] * ((sw - nw) * (1.f - ax) + (se - ne) * ax);
    }
  }
  gix *= mx;
  giy *= my;
  const float iz = 1.f / q.z;
  float gp[3];
  gp[0] = gix * iz;
  gp[1] = giy * iz;
  gp[2] = (q.p[2] >= 1e-5f) ? -(gix * q.p[0] + giy * q.p[1]) * iz * iz : 0.f;
  float gXr[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) gXr[i] = cam.rk[0 + i] * gp[0] + cam.rk[3 + i] * gp[1] + cam.rk[6 + i] * gp[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    gT[4 * i + 0] = gXr[i] * q.X[0];
    gT[4 * i + 1] = gXr[i] * q.X[1];
    gT[4 * i + 2] = gXr[i] * q.X[2];
    gT[4 * i + 3] = gXr[i];
  }
  float gd = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float gX = Tm[0 + k] * gXr[0] + Tm[4 + k] * gXr[1] + Tm[8 + k] * gXr[2];
    gd += q.ray[k] * gX;
  }
  return rho >= 1e-6f ? -gd * q.d * q.d : 0.f;
}

// grid: (ceil(HW/256), B, J)
__global__ void __launch_bounds__(256) view_synthesis_fwd_kernel(const float* __restrict__ inv_depth, const float* __restrict__ ref,
                                                                  const float* __restrict__ K, const float* __restrict__ refK,
                                                                  const float* __restrict__ T, float* __restrict__ warped,
                                                                  int B, int H, int W, int pad_mode) {
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y, j = blockIdx.z;
  if (pix >= HW) return;
  Cam cam;
  load_cam(K, refK, b, cam);
  float Tm[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) Tm[i] = T[((size_t)j * B + b) * 16 + i];
  const int v = pix / W, u = pix - v * W;
  Proj q;
  project_pixel(cam, Tm, inv_depth[(size_t)b * HW + pix], u, v, H, W, q);
  float out[3] = {0.f, 0.f, 0.f};
  float mx, my;
  q.ix = pad_coordinate(q.ix, W, pad_mode, mx);
  q.iy = pad_coordinate(q.iy, H, pad_mode, my);
  if (q.ix > -1.f && q.ix < (float)W && q.iy > -1.f && q.iy < (float)H) {
    const float fx0 = floorf(q.ix), fy0 = floorf(q.iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float ax = q.ix - fx0, ay = q.iy - fy0;
    const float wnw = (1.f - ax) * (1.f - ay), wne = ax * (1.f - ay), wsw = (1.f - ax) * ay, wse = ax * ay;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    const float* rb = ref + ((size_t)j * B + b) * 3 * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* rc = rb + (size_t)c * HW;
      float acc = 0.f;
      if (vx0 && vy0) acc += rc[y0 * W + x0] * wnw;
      if (vx1 && vy0) acc += rc[y0 * W + x1] * wne;
      if (vx0 && vy1) acc += rc[y1 * W + x0] * wsw;
      if (vx1 && vy1) acc += rc[y1 * W + x1] * wse;
      out[c] = acc;
    }
  }
  float* wb = warped + ((size_t)j * B + b) * 3 * HW + pix;
#pragma unroll
  for (int c = 0; c < 3; ++c) wb[(size_t)c * HW] = out[c];
}

// grid: (ceil(HW/256), B). Loops over the J context images so that d_inv_depth needs no atomics.
// ws: double[J*B*12] (zeroed by the caller) receives d(loss)/d(T[:3,:4]).
__global__ void __launch_bounds__(256) view_synthesis_bwd_kernel(const float* __restrict__ d_warped, const float* __restrict__ inv_depth,
                                                                  const float* __restrict__ ref, const float* __restrict__ K,
                                                                  const float* __restrict__ refK, const float* __restrict__ T,
                                                                  float* __restrict__ d_inv_depth, double* __restrict__ ws,
                                                                  int J, int B, int H, int W, int pad_mode) {
  __shared__ float redT[4][12];
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const bool active = pix < HW;
  Cam cam;
  load_cam(K, refK, b, cam);
  const int v = active ? pix / W : 0, u = active ? pix - v * W : 0;
  const float rho = active ? inv_depth[(size_t)b * HW + pix] : 1.f;
  float g_rho = 0.f;
  for (int j = 0; j < J; ++j) {
    float Tm[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Tm[i] = T[((size_t)j * B + b) * 16 + i];
    float gT[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gT[i] = 0.f;
    if (active) {
      Proj q;
      project_pixel(cam, Tm, rho, u, v, H, W, q);
      float gix = 0.f, giy = 0.f;
      float mx, my;
      q.ix = pad_coordinate(q.ix, W, pad_mode, mx);
      q.iy = pad_coordinate(q.iy, H, pad_mode, my);
      if (q.ix > -1.f && q.ix < (float)W && q.iy > -1.f && q.iy < (float)H) {
        const float fx0 = floorf(q.ix), fy0 = floorf(q.iy);
        const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
        const float ax = q.ix - fx0, ay = q.iy - fy0;
        const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
        const float* rb = ref + ((size_t)j * B + b) * 3 * HW;
        const float* gb = d_warped + ((size_t)j * B + b) * 3 * HW + pix;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float* rc = rb + (size_t)c * HW;
          const float g = gb[(size_t)c * HW];
          const float nw = (vx0 && vy0) ? rc[y0 * W + x0] : 0.f;
          const float ne = (vx1 && vy0) ? rc[y0 * W + x1] : 0.f;
          const float sw = (vx0 && vy1) ? rc[y1 * W + x0] : 0.f;
          const float se = (vx1 && vy1) ? rc[y1 * W + x1] : 0.f;
          // d out / d ix = (ne - nw)(1-ay) + (se - sw) ay ;  d out / d iy = (sw - nw)(1-ax) + (se - ne) ax
          gix += g * ((ne - nw) * (1.f - ay) + (se - sw) * ay);
          giy += g * ((sw - nw) * (1.f - ax) + (se - ne) * ax);
        }
      }
      gix *= mx;       // d(padded coordinate) / d(coordinate)
      giy *= my;
      // ix = p.x / z, iy = p.y / z (the (W-1)/2 factors of normalise/un-normalise cancel)
      const float iz = 1.f / q.z;
      float gp[3];
      gp[0] = gix * iz;
      gp[1] = giy * iz;
      gp[2] = (q.p[2] >= 1e-5f) ? -(gix * q.p[0] + giy * q.p[1]) * iz * iz : 0.f;
      float gXr[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) gXr[i] = cam.rk[0 + i] * gp[0] + cam.rk[3 + i] * gp[1] + cam.rk[6 + i] * gp[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        gT[4 * i + 0] = gXr[i] * q.X[0];
        gT[4 * i + 1] = gXr[i] * q.X[1];
        gT[4 * i + 2] = gXr[i] * q.X[2];
        gT[4 * i + 3] = gXr[i];
      }
      float gd = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float gX = Tm[0 + k] * gXr[0] + Tm[4 + k] * gXr[1] + Tm[8 + k] * gXr[2];
        gd += q.ray[k] * gX;
      }
      if (rho >= 1e-6f) g_rho += -gd * q.d * q.d;
    }
    // 12 pose-gradient partials: wave shuffles (fp32), one LDS stage across the 4 waves, then one fp64 slot per (matrix, pixel
    // block, entry) -- view_synthesis_bwd_finish_kernel adds the pixel blocks in a fixed order (round 3: fp64 atomics)
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        float v = gT[i];
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
        if (lane == 0) redT[wave][i] = v;
      }
      __syncthreads();
      if (threadIdx.x < 12) {
        const int i = threadIdx.x;
        const double s = (double)redT[0][i] + (double)redT[1][i] + (double)redT[2][i] + (double)redT[3][i];
        ws[(((size_t)j * B + b) * gridDim.x + blockIdx.x) * 12 + i] = s;
      }
      __syncthreads();
    }
  }
  if (active) d_inv_depth[(size_t)b * HW + pix] = g_rho;
}

// dT[m] (4x4, last row zero) = sum over the pixel blocks' partials of matrix m: one wave per matrix, lane l adds blocks l, l + 64,
// ... in order and the 64 lane sums meet in a fixed shuffle tree -- bit-reproducible
__global__ void __launch_bounds__(64) view_synthesis_bwd_finish_kernel(const double* __restrict__ part, float* __restrict__ dT, int nblk) {
  const int m = blockIdx.x, lane = threadIdx.x;
  double s[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = 0.0;
  for (int p = lane; p < nblk; p += 64) {
    const double* q = part + ((size_t)m * nblk + p) * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] += q[i];
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    double v = s[i];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
    if (lane == 0) dT[m * 16 + i] = (float)v;
  }
  if (lane < 4) dT[m * 16 + 12 + lane] = 0.f;
}

// ---- SSIM / L1 photometric terms -------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) {
  // nn.ReflectionPad2d(1) index map, then clamped so far-outside tile padding never reads out of bounds
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  if (i < 0) i = 0;
  if (i >= n) i = n - 1;
  return i;
}

struct WinStats { float mx, my, sxx, syy, sxy; };

// 3x3 mean statistics of (x, y) around LDS position `c` (row stride `S`)
__device__ __forceinline__ WinStats win_stats(const float* xs, const float* ys, int c, int S) {
  float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const float a = xs[c + dy * S + dx], b = ys[c + dy * S + dx];
      sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
    }
  WinStats w;
  const float inv9 = 1.f / 9.f;
  w.mx = sx * inv9; w.my = sy * inv9; w.sxx = sxx * inv9; w.syy = syy * inv9; w.sxy = sxy * inv9;
  return w;
}

struct SsimTerms { float ssim, A1, A2, B1, B2; };

__device__ __forceinline__ SsimTerms ssim_terms(const WinStats& w, float C1, float C2) {
  SsimTerms t;
  const float mxmy = w.mx * w.my, mx2 = w.mx * w.mx, my2 = w.my * w.my;
  const float sig_x = w.sxx - mx2, sig_y = w.syy - my2, sig_xy = w.sxy - mxmy;
  t.A1 = 2.f * mxmy + C1;
  t.A2 = 2.f * sig_xy + C2;
  t.B1 = mx2 + my2 + C1;
  t.B2 = sig_x + sig_y + C2;
  t.ssim = (t.A1 * t.A2) / (t.B1 * t.B2);
  return t;
}

#define PH_T 16           // output tile edge
#define PH_S1 (PH_T + 2)  // with 1-pixel halo
#define PH_S2 (PH_T + 4)  // with 2-pixel halo

// grid: (ceil(W/16), ceil(H/16), B); block 256 = 16x16 pixels. LDS: (1 + 2J) images x 3 ch x 18x18.
// clip_loss > 0 (multiview_photometric_loss.py:214-219: every candidate map is clamped at mean + clip*std of ITSELF, a
// float, so no gradient flows through the statistics) needs those statistics first: with `stats` != null the kernel only
// accumulates sum / sum-of-squares per candidate (double[2*ncand]); a finish kernel turns them into thresholds `clip_thr`
// [ncand] for the real pass, which clamps and records which candidates were clamped in the per-pixel byte
// (min: argmin | clamped << 7; mean: bit mask of clamped candidates) so that backward can zero their gradient.
__global__ void __launch_bounds__(256) photometric_fwd_kernel(const float* __restrict__ warped, const float* __restrict__ ref,
                                                               const float* __restrict__ target, double* __restrict__ loss_sum,
                                                               uint8_t* __restrict__ argmin, int J, int B, int H, int W,
                                                               float ssim_w, float C1, float C2, int automask, int reduce_op,
                                                               const float* __restrict__ clip_thr, double* __restrict__ stats) {
  PNSFM_DYN_SMEM(float, smem);
  __shared__ double red[4];
  const int HW = H * W;
  // logical tile (x, y, image): the physical workgroup index (x fastest) goes round robin over the 8 XCDs, so horizontally adjacent
  // 16-pixel tiles -- two per 128-byte line -- and the tiles sharing halo rows would each pull the same lines into a different L2
  // (r04_traffic.json: 3.9x the algorithmic bytes); one contiguous range of the tile order per XCD keeps neighbours on one L2
  const unsigned Lb = pnsfm_xcd_logical_block((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
  const int lbx = (int)(Lb % gridDim.x), lby = (int)((Lb / gridDim.x) % gridDim.y);
  const int b = (int)(Lb / (gridDim.x * gridDim.y));
  const int tx0 = lbx * PH_T, ty0 = lby * PH_T;
  const int plane = PH_S1 * PH_S1;
  // image slots: 0 = target, 1+2j = warped[j], 2+2j = ref[j]
  const int nimg = 1 + 2 * J;
  for (int e = threadIdx.x; e < nimg * 3 * plane; e += 256) {
    const int img = e / (3 * plane);
    int r = e - img * 3 * plane;
    const int c = r / plane;
    r -= c * plane;
    const int ly = r / PH_S1, lx = r - ly * PH_S1;
    const int gy = reflect_idx(ty0 + ly - 1, H), gx = reflect_idx(tx0 + lx - 1, W);
    const float* src;
    if (img == 0) src = target + (size_t)b * 3 * HW;
    else {
      const int j = (img - 1) >> 1;
      src = (((img - 1) & 1) ? ref : warped) + ((size_t)j * B + b) * 3 * HW;
    }
    float v = 0.f;
    if (img == 0 || !(((img - 1) & 1) && !automask)) v = src[(size_t)c * HW + gy * W + gx];
    smem[e] = v;
  }
  __syncthreads();
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  const int gy = ty0 + ly, gx = tx0 + lx;
  const bool valid = gy < H && gx < W;
  const int cpos = (ly + 1) * PH_S1 + lx + 1;
  float best = 0.f, sum = 0.f;
  int best_i = 0, ncand = 0, clamp_mask = 0;
  float cl[6];
  const float* tgt = smem;
  for (int j = 0; j < J; ++j) {
    for (int id = 0; id < (automask ? 2 : 1); ++id) {
      const float* img = smem + (1 + 2 * j + id) * 3 * plane;
      float ssim_acc = 0.f, l1_acc = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* xs = img + c * plane;
        const float* ys = tgt + c * plane;
        const WinStats w = win_stats(xs, ys, cpos, PH_S1);
        const SsimTerms t = ssim_terms(w, C1, C2);
        ssim_acc += fminf(fmaxf((1.f - t.ssim) * 0.5f, 0.f), 1.f);
        l1_acc += fabsf(xs[cpos] - ys[cpos]);
      }
      float l = ssim_w * (ssim_acc / 3.f) + (1.f - ssim_w) * (l1_acc / 3.f);
      cl[ncand] = l;
      if (clip_thr != nullptr && l > clip_thr[ncand]) { l = clip_thr[ncand]; clamp_mask |= 1 << ncand; }
      if (ncand == 0 || l < best) { best = l; best_i = ncand; }
      sum += l;
      ncand++;
    }
  }
  if (stats != nullptr) {   // statistics pre-pass of clip_loss: nothing else is produced
    for (int c = 0; c < ncand; ++c) {
      const double v = valid ? (double)cl[c] : 0.0;
      const double s1 = block_sum_256d(v, red);
      const double s2 = block_sum_256d(v * v, red);
      if (threadIdx.x == 0) { atomicAdd(&stats[2 * c], s1); atomicAdd(&stats[2 * c + 1], s2); }
    }
    return;
  }
  float contrib = 0.f;
  if (valid) {
    contrib = reduce_op == 0 ? best : sum / (float)ncand;
    if (reduce_op == 0) argmin[(size_t)b * HW + gy * W + gx] = (uint8_t)(best_i | (((clamp_mask >> best_i) & 1) << 7));
    else if (clip_thr != nullptr) argmin[(size_t)b * HW + gy * W + gx] = (uint8_t)clamp_mask;
  }
  // stage 1 of the pixel mean: one partial per workgroup (loss_sum = the launch's partial buffer); sum_partials_kernel adds
  // them in a fixed order -- no atomics, the loss is bit-reproducible
  const double s = block_sum_256d((double)contrib, red);
  if (threadIdx.x == 0) loss_sum[Lb] = s;
}

// grid as forward. LDS: (1 + J) images x 3 ch x 20x20  +  J x 3 ch x 3 coefficient planes x 18x18.
// d ssim_p / d x_q = alpha_p + beta_p * y_q + gamma_p * x_q for q in the 3x3 window of p,
// so the gradient at q is a (reflect-aware) 3x3 gather of three coefficient planes.
__global__ void __launch_bounds__(256) photometric_bwd_kernel(const float* __restrict__ warped, const float* __restrict__ target,
                                                               const uint8_t* __restrict__ argmin, float* __restrict__ d_warped,
                                                               float grad_scale, int J, int B, int H, int W, float ssim_w,
                                                               float C1, float C2, int automask, int reduce_op, int clip,
                                                               const float* __restrict__ gdev) {
  PNSFM_DYN_SMEM(float, smem);
  if (gdev) grad_scale *= gdev[0];        // upstream gradient of the scalar loss, still on the device (no `d * g` pass over d_warped)
  const int HW = H * W;
  // logical tile order: one contiguous range per XCD (see photometric_fwd_kernel)
  const unsigned Lb = pnsfm_xcd_logical_block((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x * gridDim.y * gridDim.z);
  const int lbx = (int)(Lb % gridDim.x), lby = (int)((Lb / gridDim.x) % gridDim.y);
  const int b = (int)(Lb / (gridDim.x * gridDim.y));
  const int tx0 = lbx * PH_T, ty0 = lby * PH_T;
  const int plane2 = PH_S2 * PH_S2, plane1 = PH_S1 * PH_S1;
  float* imgs = smem;                           // [(1+J)][3][20*20]; slot 0 = target, 1+j = warped[j]
  float* coef = smem + (1 + J) * 3 * plane2;    // [J][3 ch][3 (alpha,beta,gamma)][18*18]
  for (int e = threadIdx.x; e < (1 + J) * 3 * plane2; e += 256) {
    const int img = e / (3 * plane2);
    int r = e - img * 3 * plane2;
    const int c = r / plane2;
    r -= c * plane2;
    const int ly = r / PH_S2, lx = r - ly * PH_S2;
    const int gy = reflect_idx(ty0 + ly - 2, H), gx = reflect_idx(tx0 + lx - 2, W);
    const float* src = img == 0 ? target + (size_t)b * 3 * HW : warped + ((size_t)(img - 1) * B + b) * 3 * HW;
    imgs[e] = src[(size_t)c * HW + gy * W + gx];
  }
  __syncthreads();
  const int ncand = J * (automask ? 2 : 1);
  // ---- coefficient planes at every p of the 18x18 region
  for (int e = threadIdx.x; e < plane1; e += 256) {
    const int ly = e / PH_S1, lx = e - ly * PH_S1;
    const int py = ty0 + ly - 1, px = tx0 + lx - 1;
    const bool inside = py >= 0 && py < H && px >= 0 && px < W;
    const int cpos = (ly + 1) * PH_S2 + lx + 1;
    int sel = -1;
    if (inside && (reduce_op == 0 || clip)) sel = (int)argmin[(size_t)b * HW + py * W + px];
    for (int j = 0; j < J; ++j) {
      const int cand = j * (automask ? 2 : 1);
      float up = 0.f;
      // min: the byte is argmin | clamped << 7 (a clamped winner has no gradient); mean: a bit mask of clamped candidates
      if (inside) up = reduce_op == 0 ? (sel == cand ? grad_scale : 0.f)
                                      : ((clip && ((sel >> cand) & 1)) ? 0.f : grad_scale / (float)ncand);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float ca = 0.f, cb = 0.f, cg = 0.f;
        if (up != 0.f) {
          const float* xs = imgs + ((1 + j) * 3 + c) * plane2;
          const float* ys = imgs + c * plane2;
          const WinStats w = win_stats(xs, ys, cpos, PH_S2);
          const SsimTerms t = ssim_terms(w, C1, C2);
          const float L = (1.f - t.ssim) * 0.5f;
          if (L >= 0.f && L <= 1.f) {
            const float den = t.B1 * t.B2;
            const float k = up * ssim_w * (1.f / 3.f) * (-0.5f) / (9.f * den);
            ca = k * (2.f * w.my * t.A2 - 2.f * t.A1 * w.my - t.ssim * (2.f * w.mx * t.B2 - 2.f * t.B1 * w.mx));
            cb = k * (2.f * t.A1);
            cg = k * (-2.f * t.ssim * t.B1);
          }
        }
        float* cp = coef + ((j * 3 + c) * 3) * plane1;
        cp[0 * plane1 + e] = ca;
        cp[1 * plane1 + e] = cb;
        cp[2 * plane1 + e] = cg;
      }
    }
  }
  __syncthreads();
  // ---- gather at q
  const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
  const int qy = ty0 + ly, qx = tx0 + lx;
  const bool active = qy < H && qx < W;
  if (!active) return;
  const int q2 = (ly + 2) * PH_S2 + lx + 2;
  int selq = -1;
  if (active && (reduce_op == 0 || clip)) selq = (int)argmin[(size_t)b * HW + qy * W + qx];
  for (int j = 0; j < J; ++j) {
    const int cand = j * (automask ? 2 : 1);
    const float uq = reduce_op == 0 ? (selq == cand ? grad_scale : 0.f)
                                    : ((clip && ((selq >> cand) & 1)) ? 0.f : grad_scale / (float)ncand);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* cp = coef + ((j * 3 + c) * 3) * plane1;
      float sa = 0.f, sb = 0.f, sg = 0.f;
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        const int py = qy + dy;
        if (py < 0 || py >= H) continue;
        const float my = 1.f + ((py == 0 && qy == 1) ? 1.f : 0.f) + ((py == H - 1 && qy == H - 2) ? 1.f : 0.f);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const int px = qx + dx;
          if (px < 0 || px >= W) continue;
          const float mx = 1.f + ((px == 0 && qx == 1) ? 1.f : 0.f) + ((px == W - 1 && qx == W - 2) ? 1.f : 0.f);
          const float m = my * mx;
          const int e = (ly + 1 + dy) * PH_S1 + lx + 1 + dx;
          sa += m * cp[0 * plane1 + e];
          sb += m * cp[1 * plane1 + e];
          sg += m * cp[2 * plane1 + e];
        }
      }
      const float xq = imgs[((1 + j) * 3 + c) * plane2 + q2];
      const float yq = imgs[c * plane2 + q2];
      float g = sa + sb * yq + sg * xq;
      const float diff = xq - yq;
      const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
      g += uq * (1.f - ssim_w) * (1.f / 3.f) * sgn;
      d_warped[(((size_t)j * B + b) * 3 + c) * HW + qy * W + qx] = g;
    }
  }
}

// ---- smoothness ------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_weight(const float* img, size_t HW, int i0, int i1) {
  const float s = fabsf(img[i0] - img[i1]) + fabsf(img[HW + i0] - img[HW + i1]) + fabsf(img[2 * HW + i0] - img[2 * HW + i1]);
  return expf(-(s / 3.f));
}

// grid: (ceil(HW/256), B)
__global__ void __launch_bounds__(256) smoothness_fwd_kernel(const float* __restrict__ inv, const float* __restrict__ image,
                                                              double* __restrict__ sums, int H, int W) {
  __shared__ double red[4];
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  float sx = 0.f, sy = 0.f;
  if (pix < HW) {
    const int y = pix / W, x = pix - y * W;
    const float* ib = inv + (size_t)b * HW;
    const float* im = image + (size_t)b * 3 * HW;
    const float r = ib[pix];
    if (x + 1 < W) sx = fabsf((r - ib[pix + 1]) * edge_weight(im, HW, pix, pix + 1));
    if (y + 1 < H) sy = fabsf((r - ib[pix + W]) * edge_weight(im, HW, pix, pix + W));
  }
  const double tx = block_sum_256d((double)sx, red);
  const double ty = block_sum_256d((double)sy, red);
  if (threadIdx.x == 0) {      // stage 1: partials [block][2]; sum_partials_kernel finishes
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    sums[2 * blk] = tx;
    sums[2 * blk + 1] = ty;
  }
}

__device__ __forceinline__ float sgnf(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) smoothness_bwd_kernel(const float* __restrict__ inv, const float* __restrict__ image,
                                                              float* __restrict__ d_inv, float gx, float gy, int H, int W) {
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= HW) return;
  const int y = pix / W, x = pix - y * W;
  const float* ib = inv + (size_t)b * HW;
  const float* im = image + (size_t)b * 3 * HW;
  const float r = ib[pix];
  float g = 0.f;
  if (x + 1 < W) { const float w = edge_weight(im, HW, pix, pix + 1); g += gx * sgnf((r - ib[pix + 1]) * w) * w; }
  if (x >= 1)    { const float w = edge_weight(im, HW, pix - 1, pix); g -= gx * sgnf((ib[pix - 1] - r) * w) * w; }
  if (y + 1 < H) { const float w = edge_weight(im, HW, pix, pix + W); g += gy * sgnf((r - ib[pix + W]) * w) * w; }
  if (y >= 1)    { const float w = edge_weight(im, HW, pix - W, pix); g -= gy * sgnf((ib[pix - W] - r) * w) * w; }
  d_inv[(size_t)b * HW + pix] = g;
}

// ---- L1-only photometric loss (ssim_loss_weight == 0) with the 'min' reduce op and / or clipping (round 4).  Without the SSIM term
// the reference keeps every candidate as a THREE-channel map (multiview_photometric_loss.py:205-213): clipping takes mean / std over
// B*3*H*W elements of each candidate (:214-219) and 'min' runs over the concatenated channels of all candidates (:243-244), i.e. over
// (candidate, channel) pairs -- not what the SSIM kernels' per-pixel channel mean computes.  No 3x3 window is involved, so these are
// plain per-pixel kernels.  Candidate order as in the reference: for each context j: warped_j, then (automask) the unwarped ref_j.
//   stats != null : statistics pre-pass of clip_loss -- per-block partials [block][ncand][2] of (sum, sum of squares), nothing else
//   rec           : per pixel, 'min': index (candidate * 3 + channel) | clamped << 7; 'mean': bit mask of the clamped pairs
__global__ void __launch_bounds__(256) l1cand_fwd_kernel(const float* __restrict__ warped, const float* __restrict__ ref,
                                                          const float* __restrict__ target, const float* __restrict__ thr,
                                                          double* __restrict__ part, int* __restrict__ rec, double* __restrict__ stats,
                                                          int J, int B, int H, int W, int automask, int reduce_op) {
  __shared__ double red[4];
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const bool valid = pix < HW;
  const int ncand = J * (automask ? 2 : 1);
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  float t[3] = {0.f, 0.f, 0.f};
  if (valid)
    for (int ch = 0; ch < 3; ++ch) t[ch] = target[((size_t)b * 3 + ch) * HW + pix];
  float best = 0.f, sum = 0.f;
  int best_i = 0, best_clamped = 0, mask = 0;
  for (int c = 0; c < ncand; ++c) {
    const int j = automask ? c >> 1 : c;
    const float* img = ((automask && (c & 1)) ? ref : warped) + ((size_t)j * B + b) * 3 * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int ch = 0; ch < 3; ++ch) {
      float v = valid ? fabsf(img[(size_t)ch * HW + pix] - t[ch]) : 0.f;
      if (stats != nullptr) { s1 += (double)v; s2 += (double)v * (double)v; continue; }
      int cl = 0;
      if (thr != nullptr && v > thr[c]) { v = thr[c]; cl = 1; }
      const int idx = c * 3 + ch;
      if (idx == 0 || v < best) { best = v; best_i = idx; best_clamped = cl; }
      sum += v;
      mask |= cl << idx;
    }
    if (stats != nullptr) {
      const double a = block_sum_256d(s1, red);
      const double q = block_sum_256d(s2, red);
      if (threadIdx.x == 0) { stats[(blk * ncand + c) * 2] = a; stats[(blk * ncand + c) * 2 + 1] = q; }
    }
  }
  if (stats != nullptr) return;
  float contrib = 0.f;
  if (valid) {
    contrib = reduce_op == 0 ? best : sum / (float)(3 * ncand);
    rec[(size_t)b * HW + pix] = reduce_op == 0 ? (best_i | (best_clamped << 7)) : mask;
  }
  const double sblk = block_sum_256d((double)contrib, red);
  if (threadIdx.x == 0) part[blk] = sblk;
}

// thresholds of the clipped candidates from the per-block statistics: thr[c] = mean + clip * std (unbiased) over n elements
__global__ void __launch_bounds__(256) l1cand_thr_kernel(const double* __restrict__ stats, int nblk, int ncand, double n, float clip,
                                                          float* __restrict__ thr) {
  __shared__ double red[4];
  for (int c = 0; c < ncand; ++c) {
    double a = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += stats[((size_t)i * ncand + c) * 2]; q += stats[((size_t)i * ncand + c) * 2 + 1]; }
    const double s1 = block_sum_256d(a, red);
    const double s2 = block_sum_256d(q, red);
    if (threadIdx.x == 0) {
      const double mean = s1 / n;
      double var = (s2 - n * mean * mean) / (n - 1.0);
      if (var < 0.0) var = 0.0;
      thr[c] = (float)(mean + (double)clip * sqrt(var));
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) l1cand_bwd_kernel(const float* __restrict__ warped, const float* __restrict__ target,
                                                          const int* __restrict__ rec, float* __restrict__ d_warped, float grad_scale,
                                                          const float* __restrict__ gdev, int J, int B, int H, int W, int automask,
                                                          int reduce_op) {
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pix >= HW) return;
  if (gdev) grad_scale *= gdev[0];
  const int ncand = J * (automask ? 2 : 1);
  const int r = rec[(size_t)b * HW + pix];
  for (int j = 0; j < J; ++j) {
    const int c = automask ? 2 * j : j;
    for (int ch = 0; ch < 3; ++ch) {
      const size_t o = (((size_t)j * B + b) * 3 + ch) * HW + pix;
      const float diff = warped[o] - target[((size_t)b * 3 + ch) * HW + pix];
      const int idx = c * 3 + ch;
      float g;
      if (reduce_op == 0) g = ((r & 127) == idx && !(r >> 7)) ? grad_scale : 0.f;
      else g = ((r >> idx) & 1) ? 0.f : grad_scale / (float)(3 * ncand);
      d_warped[o] = g * sgnf(diff);
    }
  }
}

// ---- smoothness of the MEAN-NORMALISED inverse depth, normalisation fused (round 4).  The reference normalises on the host side of
// the kernel boundary: inv / inv.mean(2, True).mean(3, True).clamp(min=1e-6) (multiview_photometric_loss.py:269-271) -- four ATen
// launches forward and ~eight backward per scale around two tiny kernels.  Here:
//   forward : sn_mean_kernel (per-sample partial sums of d)  ->  sn_fwd_kernel (every block adds its sample's partials in a fixed
//             order, m = max(mean, 1e-6), smoothness of d / m, partial |Sx|, |Sy| sums; block 0 of a sample stores m)  ->
//             sum_partials_scaled_kernel (loss = sum|Sx|/nx + sum|Sy|/ny as a float)
//   backward: with n = d / m and g_n = dL/dn:  dL/dd_i = g_n,i / m - [mean > 1e-6] * (sum_j g_n,j d_j) / (m^2 HW)
//             sn_bwd_kernel writes g_n and per-block partials of sum g_n d;  sn_bwd_apply_kernel finishes in place.
__global__ void __launch_bounds__(256) sn_mean_kernel(const float* __restrict__ inv, double* __restrict__ part, int HW) {
  __shared__ double red[4];
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const double v = pix < HW ? (double)inv[(size_t)blockIdx.y * HW + pix] : 0.0;
  const double t = block_sum_256d(v, red);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// m_b = clamp(mean of sample b, 1e-6): the partials are added in index order by every block alike (identical bits everywhere)
__device__ __forceinline__ float sn_sample_mean(const double* __restrict__ mpart, int nblk, int HW, double* red, bool* clamped) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) acc += mpart[i];
  const double tot = block_sum_256d(acc, red);
  __syncthreads();
  const float mean = (float)(tot / (double)HW);
  if (clamped) *clamped = !(mean > 1e-6f);
  return mean > 1e-6f ? mean : 1e-6f;
}

__global__ void __launch_bounds__(256) sn_fwd_kernel(const float* __restrict__ inv, const float* __restrict__ image,
                                                      const double* __restrict__ mpart, double* __restrict__ sums,
                                                      float* __restrict__ mean_out, int H, int W) {
  __shared__ double red[4];
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const float m = sn_sample_mean(mpart + (size_t)b * gridDim.x, (int)gridDim.x, HW, red, nullptr);
  if (blockIdx.x == 0 && threadIdx.x == 0) mean_out[b] = m;
  float sx = 0.f, sy = 0.f;
  if (pix < HW) {
    const int y = pix / W, x = pix - y * W;
    const float* ib = inv + (size_t)b * HW;
    const float* im = image + (size_t)b * 3 * HW;
    const float r = ib[pix] / m;
    if (x + 1 < W) sx = fabsf((r - ib[pix + 1] / m) * edge_weight(im, HW, pix, pix + 1));
    if (y + 1 < H) sy = fabsf((r - ib[pix + W] / m) * edge_weight(im, HW, pix, pix + W));
  }
  const double tx = block_sum_256d((double)sx, red);
  const double ty = block_sum_256d((double)sy, red);
  if (threadIdx.x == 0) {
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    sums[2 * blk] = tx;
    sums[2 * blk + 1] = ty;
  }
}

__global__ void __launch_bounds__(256) sn_bwd_kernel(const float* __restrict__ inv, const float* __restrict__ image,
                                                      const float* __restrict__ mean, float* __restrict__ g_n, double* __restrict__ part,
                                                      float gx, float gy, int H, int W) {
  __shared__ double red[4];
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const float m = mean[b];
  float g = 0.f, d = 0.f;
  if (pix < HW) {
    const int y = pix / W, x = pix - y * W;
    const float* ib = inv + (size_t)b * HW;
    const float* im = image + (size_t)b * 3 * HW;
    d = ib[pix];
    const float r = d / m;
    if (x + 1 < W) { const float w = edge_weight(im, HW, pix, pix + 1); g += gx * sgnf((r - ib[pix + 1] / m) * w) * w; }
    if (x >= 1)    { const float w = edge_weight(im, HW, pix - 1, pix); g -= gx * sgnf((ib[pix - 1] / m - r) * w) * w; }
    if (y + 1 < H) { const float w = edge_weight(im, HW, pix, pix + W); g += gy * sgnf((r - ib[pix + W] / m) * w) * w; }
    if (y >= 1)    { const float w = edge_weight(im, HW, pix - W, pix); g -= gy * sgnf((ib[pix - W] / m - r) * w) * w; }
    g_n[(size_t)b * HW + pix] = g;
  }
  const double t = block_sum_256d((double)g * (double)d, red);
  if (threadIdx.x == 0) part[(size_t)b * gridDim.x + blockIdx.x] = t;
}

__global__ void __launch_bounds__(256) sn_bwd_apply_kernel(float* __restrict__ d_inv, const float* __restrict__ mean,
                                                            const double* __restrict__ part, const float* __restrict__ gdev, int HW) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  double acc = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) acc += part[(size_t)b * gridDim.x + i];
  const double dot = block_sum_256d(acc, red);
  const float m = mean[b];
  // m == 1e-6 means the clamp was active (a mean that is exactly 1e-6 un-clamped differs from it by the clamp's own tie rule only)
  const float corr = m > 1e-6f ? (float)(dot / ((double)m * (double)m * (double)HW)) : 0.f;
  const float g = gdev ? gdev[0] : 1.f;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix < HW) {
    float* p = d_inv + (size_t)b * HW + pix;
    *p = g * (*p / m - corr);
  }
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_view_synthesis_forward_pad(const float* inv_depth, const float* ref, const float* K, const float* refK,
                                     const float* T, float* warped, int J, int B, int H, int W, int padding_mode,
                                     void* stream) {
  if (J < 1 || B < 1 || H < 2 || W < 2) { set_error("view_synthesis_forward: bad shape"); return -1; }
  if (padding_mode < 0 || padding_mode > 2) { set_error("view_synthesis_forward: padding_mode must be 0 (zeros), 1 (border) or 2 (reflection)"); return -1; }
  dim3 grid(ceil_div(H * W, 256), B, J);
  PNSFM_LAUNCH(view_synthesis_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, inv_depth, ref, K, refK, T, warped, B, H, W,
               padding_mode);
  return check_launch("view_synthesis_forward");
}

int pnsfm_view_synthesis_forward(const float* inv_depth, const float* ref, const float* K, const float* refK, const float* T,
                                 float* warped, int J, int B, int H, int W, void* stream) {
  return pnsfm_view_synthesis_forward_pad(inv_depth, ref, K, refK, T, warped, J, B, H, W, 0, stream);
}

int pnsfm_view_synthesis_backward(const float* d_warped, const float* inv_depth, const float* ref, const float* K,
                                  const float* refK, const float* T, float* d_inv_depth, float* dT, double* ws, int J, int B,
                                  int H, int W, void* stream) {
  return pnsfm_view_synthesis_backward_pad(d_warped, inv_depth, ref, K, refK, T, d_inv_depth, dT, ws, J, B, H, W, 0, stream);
}

int pnsfm_view_synthesis_backward_pad(const float* d_warped, const float* inv_depth, const float* ref, const float* K,
                                      const float* refK, const float* T, float* d_inv_depth, float* dT, double* ws, int J,
                                      int B, int H, int W, int padding_mode, void* stream) {
  if (J < 1 || B < 1 || H < 2 || W < 2) { set_error("view_synthesis_backward: bad shape"); return -1; }
  if (padding_mode < 0 || padding_mode > 2) { set_error("view_synthesis_backward: bad padding_mode"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  (void)ws;     // (round 3's zero-filled atomics target; the per-block partials now live in the stream's scratch buffer)
  dim3 grid(ceil_div(H * W, 256), B);
  ScratchLease lease(s, (size_t)J * B * grid.x * 12 * sizeof(double));
  double* const part = lease.as<double>();
  if (!part) return -1;
  PNSFM_LAUNCH(view_synthesis_bwd_kernel, grid, dim3(256), 0, s, d_warped, inv_depth, ref, K, refK, T, d_inv_depth, part, J, B,
               H, W, padding_mode);
  int e = check_launch("view_synthesis_backward");
  if (e) return e;
  PNSFM_LAUNCH(view_synthesis_bwd_finish_kernel, dim3(J * B), dim3(64), 0, s, (const double*)part, dT, (int)grid.x);
  return check_launch("view_synthesis_backward_finish");
}

__global__ void photometric_clip_finish_kernel(const double* __restrict__ stats, float* __restrict__ thr, int ncand, double n,
                                               float clip) {
  const int c = threadIdx.x;
  if (c < ncand) {
    const double mean = stats[2 * c] / n;
    double var = (stats[2 * c + 1] - n * mean * mean) / (n - 1.0);     // torch.std: unbiased
    if (var < 0.0) var = 0.0;
    thr[c] = (float)(mean + (double)clip * sqrt(var));
  }
}

// per-workgroup partial sums of a two-stage reduction live in the stream's scratch buffer (api.hip: ScratchLease)

int pnsfm_photometric_forward_clip(const float* warped, const float* ref, const float* target, double* loss_sum,
                                   uint8_t* argmin, int J, int B, int H, int W, float ssim_weight, float C1, float C2,
                                   int automask, int reduce_op, float clip_loss, double* stats_ws, float* thr_ws, void* stream) {
  if (J < 1 || J > 3 || H < 3 || W < 3) { set_error("photometric_forward: bad shape (J=%d H=%d W=%d; J<=3)", J, H, W); return -1; }
  if (!(ssim_weight > 0.f)) { set_error("photometric_forward: ssim_weight must be > 0"); return -1; }
  if (automask && reduce_op != 0) { set_error("photometric_forward: automask requires the 'min' reduce op"); return -1; }
  if (!(clip_loss > 0.f) || !stats_ws || !thr_ws) { set_error("photometric_forward_clip: clip_loss must be > 0 with scratch buffers"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const int ncand = J * (automask ? 2 : 1);
  int e = (int)hipMemsetAsync(stats_ws, 0, 12 * sizeof(double), s);
  if (e) { set_error("photometric_forward: memset failed"); return e; }
  dim3 grid(ceil_div(W, PH_T), ceil_div(H, PH_T), B);
  const int nblk = (int)(grid.x * grid.y * grid.z);
  ScratchLease lease(s, (size_t)nblk * sizeof(double));
  double* part = lease.as<double>();
  if (!part) return -1;
  const size_t smem = (size_t)(1 + 2 * J) * 3 * PH_S1 * PH_S1 * sizeof(float);
  PNSFM_LAUNCH(photometric_fwd_kernel, grid, dim3(256), smem, s, warped, ref, target, part, argmin, J, B, H, W, ssim_weight,
               C1, C2, automask, reduce_op, (const float*)nullptr, stats_ws);
  PNSFM_LAUNCH(photometric_clip_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)stats_ws, thr_ws, ncand,
               (double)B * H * W, clip_loss);
  PNSFM_LAUNCH(photometric_fwd_kernel, grid, dim3(256), smem, s, warped, ref, target, part, argmin, J, B, H, W, ssim_weight,
               C1, C2, automask, reduce_op, (const float*)thr_ws, (double*)nullptr);
  PNSFM_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nblk, 1, loss_sum);
  return check_launch("photometric_forward_clip");
}

int pnsfm_photometric_forward(const float* warped, const float* ref, const float* target, double* loss_sum, uint8_t* argmin,
                              int J, int B, int H, int W, float ssim_weight, float C1, float C2, int automask, int reduce_op,
                              void* stream) {
  if (J < 1 || J > 3 || H < 3 || W < 3) { set_error("photometric_forward: bad shape (J=%d H=%d W=%d; J<=3)", J, H, W); return -1; }
  // ssim_weight == 0 is the reference's L1-only loss, whose per-CHANNEL maps only coincide with this kernel's channel
  // mean under the 'mean' reduce without clipping (multiview_photometric_loss.py:205-219, 238-246)
  if (!(ssim_weight >= 0.f) || (ssim_weight == 0.f && reduce_op == 0)) {
    set_error("photometric_forward: ssim_weight must be > 0 (or == 0 with the 'mean' reduce op)");
    return -1;
  }
  if (automask && reduce_op != 0) { set_error("photometric_forward: automask requires the 'min' reduce op"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(W, PH_T), ceil_div(H, PH_T), B);
  const int nblk = (int)(grid.x * grid.y * grid.z);
  ScratchLease lease(s, (size_t)nblk * sizeof(double));
  double* part = lease.as<double>();
  if (!part) return -1;
  const size_t smem = (size_t)(1 + 2 * J) * 3 * PH_S1 * PH_S1 * sizeof(float);
  PNSFM_LAUNCH(photometric_fwd_kernel, grid, dim3(256), smem, s, warped, ref, target, part, argmin, J, B, H, W, ssim_weight,
               C1, C2, automask, reduce_op, (const float*)nullptr, (double*)nullptr);
  PNSFM_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nblk, 1, loss_sum);
  return check_launch("photometric_forward");
}

static int photometric_backward_impl(const float* warped, const float* target, const uint8_t* argmin, float* d_warped,
                                     float grad_scale, int J, int B, int H, int W, float ssim_weight, float C1, float C2,
                                     int automask, int reduce_op, int clip, void* stream, const float* gdev = nullptr) {
  if (J < 1 || J > 3 || H < 3 || W < 3) { set_error("photometric_backward: bad shape (J<=3)"); return -1; }
  dim3 grid(ceil_div(W, PH_T), ceil_div(H, PH_T), B);
  const size_t smem = ((size_t)(1 + J) * 3 * PH_S2 * PH_S2 + (size_t)J * 9 * PH_S1 * PH_S1) * sizeof(float);
  PNSFM_LAUNCH(photometric_bwd_kernel, grid, dim3(256), smem, (hipStream_t)stream, warped, target, argmin, d_warped,
               grad_scale, J, B, H, W, ssim_weight, C1, C2, automask, reduce_op, clip, gdev);
  return check_launch("photometric_backward");
}

int pnsfm_photometric_backward(const float* warped, const float* target, const uint8_t* argmin, float* d_warped,
                               float grad_scale, int J, int B, int H, int W, float ssim_weight, float C1, float C2,
                               int automask, int reduce_op, void* stream) {
  return photometric_backward_impl(warped, target, argmin, d_warped, grad_scale, J, B, H, W, ssim_weight, C1, C2, automask,
                                   reduce_op, 0, stream);
}

int pnsfm_photometric_backward_clip(const float* warped, const float* target, const uint8_t* argmin, float* d_warped,
                                    float grad_scale, int J, int B, int H, int W, float ssim_weight, float C1, float C2,
                                    int automask, int reduce_op, void* stream) {
  return photometric_backward_impl(warped, target, argmin, d_warped, grad_scale, J, B, H, W, ssim_weight, C1, C2, automask,
                                   reduce_op, 1, stream);
}

int pnsfm_smoothness_forward(const float* inv_norm, const float* image, double* sums, int B, int H, int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(H * W, 256), B);
  const int nblk = (int)(grid.x * grid.y);
  ScratchLease lease(s, (size_t)2 * nblk * sizeof(double));
  double* part = lease.as<double>();
  if (!part) return -1;
  PNSFM_LAUNCH(smoothness_fwd_kernel, grid, dim3(256), 0, s, inv_norm, image, part, H, W);
  PNSFM_LAUNCH(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nblk, 2, sums);
  return check_launch("smoothness_forward");
}

int pnsfm_photometric_forward_mean(const float* warped, const float* ref, const float* target, float* loss_mean, uint8_t* argmin,
                                   int J, int B, int H, int W, float ssim_weight, float C1, float C2, int automask, int reduce_op,
                                   void* stream) {
  if (J < 1 || J > 3 || H < 3 || W < 3) { set_error("photometric_forward: bad shape (J=%d H=%d W=%d; J<=3)", J, H, W); return -1; }
  if (!(ssim_weight >= 0.f) || (ssim_weight == 0.f && reduce_op == 0)) {
    set_error("photometric_forward: ssim_weight must be > 0 (or == 0 with the 'mean' reduce op)");
    return -1;
  }
  if (automask && reduce_op != 0) { set_error("photometric_forward: automask requires the 'min' reduce op"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(W, PH_T), ceil_div(H, PH_T), B);
  const int nblk = (int)(grid.x * grid.y * grid.z);
  ScratchLease lease(s, (size_t)nblk * sizeof(double));
  double* part = lease.as<double>();
  if (!part) return -1;
  const size_t smem = (size_t)(1 + 2 * J) * 3 * PH_S1 * PH_S1 * sizeof(float);
  PNSFM_LAUNCH(photometric_fwd_kernel, grid, dim3(256), smem, s, warped, ref, target, part, argmin, J, B, H, W, ssim_weight,
               C1, C2, automask, reduce_op, (const float*)nullptr, (double*)nullptr);
  PNSFM_LAUNCH(sum_partials_scaled_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nblk, 1, 1.0 / ((double)B * H * W), 0.0,
               (double*)nullptr, loss_mean);
  return check_launch("photometric_forward_mean");
}

int pnsfm_photometric_backward_dev(const float* warped, const float* target, const uint8_t* argmin, float* d_warped,
                                   float grad_scale, const float* upstream, int J, int B, int H, int W, float ssim_weight, float C1,
                                   float C2, int automask, int reduce_op, int clip, void* stream) {
  return photometric_backward_impl(warped, target, argmin, d_warped, grad_scale, J, B, H, W, ssim_weight, C1, C2, automask,
                                   reduce_op, clip ? 1 : 0, stream, upstream);
}

int pnsfm_photometric_l1_forward(const float* warped, const float* ref, const float* target, float* loss_mean, int* rec, int J, int B,
                                 int H, int W, int automask, int reduce_op, float clip_loss, void* stream) {
  if (J < 1 || J > 3 || H < 1 || W < 1 || B < 1) { set_error("photometric_l1_forward: bad shape (J=%d H=%d W=%d; J<=3)", J, H, W); return -1; }
  if (automask && reduce_op != 0) { set_error("photometric_l1_forward: automask requires the 'min' reduce op"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const int ncand = J * (automask ? 2 : 1);
  dim3 grid(ceil_div(H * W, 256), B);
  const int nblk = (int)(grid.x * grid.y);
  // scratch: [nblk] loss partials | [nblk][ncand][2] statistics | thresholds
  const size_t nd = (size_t)nblk + (size_t)nblk * ncand * 2 + 8;
  ScratchLease lease(s, nd * sizeof(double));
  double* part = lease.as<double>();
  if (!part) return -1;
  double* stats = part + nblk;
  float* thr = reinterpret_cast<float*>(stats + (size_t)nblk * ncand * 2);
  const bool clip = clip_loss > 0.f;
  if (clip) {
    PNSFM_LAUNCH(l1cand_fwd_kernel, grid, dim3(256), 0, s, warped, ref, target, (const float*)nullptr, part, (int*)nullptr, stats, J, B, H, W,
                 automask, reduce_op);
    PNSFM_LAUNCH(l1cand_thr_kernel, dim3(1), dim3(256), 0, s, (const double*)stats, nblk, ncand, (double)B * 3.0 * H * W, clip_loss, thr);
  }
  PNSFM_LAUNCH(l1cand_fwd_kernel, grid, dim3(256), 0, s, warped, ref, target, clip ? (const float*)thr : (const float*)nullptr, part, rec,
               (double*)nullptr, J, B, H, W, automask, reduce_op);
  PNSFM_LAUNCH(sum_partials_scaled_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nblk, 1, 1.0 / ((double)B * H * W), 0.0,
               (double*)nullptr, loss_mean);
  return check_launch("photometric_l1_forward");
}

int pnsfm_photometric_l1_backward(const float* warped, const float* target, const int* rec, float* d_warped, float grad_scale,
                                  const float* upstream, int J, int B, int H, int W, int automask, int reduce_op, void* stream) {
  if (J < 1 || J > 3 || H < 1 || W < 1 || B < 1) { set_error("photometric_l1_backward: bad shape"); return -1; }
  PNSFM_LAUNCH(l1cand_bwd_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, warped, target, rec, d_warped, grad_scale,
               upstream, J, B, H, W, automask, reduce_op);
  return check_launch("photometric_l1_backward");
}

int pnsfm_smoothness_norm_forward(const float* inv_depth, const float* image, float* loss, float* mean, int B, int H, int W,
                                  void* stream) {
  if (B < 1 || H < 2 || W < 2) { set_error("smoothness_norm_forward: bad shape"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(H * W, 256), B);
  const int nblk = (int)(grid.x * grid.y);
  ScratchLease lease(s, (size_t)3 * nblk * sizeof(double));
  double* mpart = lease.as<double>();
  if (!mpart) return -1;
  double* part = mpart + nblk;
  PNSFM_LAUNCH(sn_mean_kernel, grid, dim3(256), 0, s, inv_depth, mpart, H * W);
  PNSFM_LAUNCH(sn_fwd_kernel, grid, dim3(256), 0, s, inv_depth, image, (const double*)mpart, part, mean, H, W);
  PNSFM_LAUNCH(sum_partials_scaled_kernel, dim3(1), dim3(256), 0, s, (const double*)part, nblk, 2, 1.0 / ((double)B * H * (W - 1)),
               1.0 / ((double)B * (H - 1) * W), (double*)nullptr, loss);
  return check_launch("smoothness_norm_forward");
}

int pnsfm_smoothness_norm_backward(const float* inv_depth, const float* image, const float* mean, const float* upstream,
                                   float* d_inv_depth, int B, int H, int W, void* stream) {
  if (B < 1 || H < 2 || W < 2) { set_error("smoothness_norm_backward: bad shape"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(H * W, 256), B);
  ScratchLease lease(s, (size_t)grid.x * grid.y * sizeof(double));
  double* part = lease.as<double>();
  if (!part) return -1;
  PNSFM_LAUNCH(sn_bwd_kernel, grid, dim3(256), 0, s, inv_depth, image, mean, d_inv_depth, part, (float)(1.0 / ((double)B * H * (W - 1))),
               (float)(1.0 / ((double)B * (H - 1) * W)), H, W);
  PNSFM_LAUNCH(sn_bwd_apply_kernel, grid, dim3(256), 0, s, d_inv_depth, mean, (const double*)part, upstream, H * W);
  return check_launch("smoothness_norm_backward");
}

int pnsfm_smoothness_backward(const float* inv_norm, const float* image, float* d_inv_norm, float gx, float gy, int B, int H,
                              int W, void* stream) {
  PNSFM_LAUNCH(smoothness_bwd_kernel, dim3(ceil_div(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, inv_norm, image,
               d_inv_norm, gx, gy, H, W);
  return check_launch("smoothness_backward");
}

}  // extern "C"
