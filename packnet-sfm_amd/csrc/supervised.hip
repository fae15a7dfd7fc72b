// supervised.hip -- supervised inverse-depth losses of the semi-supervised models, one scale per call.
//
//   /root/reference/packnet_sfm/losses/supervised_loss.py:11-88 (BerHuLoss, SilogLoss, get_loss_func: l1 / mse / berhu /
//   silog / abs_rel) and :138-149 (calculate_loss: 'sparse-*' methods keep only pixels with gt > 0).
//
// The reference gathers the valid pixels with boolean indexing (a device->host sync for the output size) and then runs
// 3-8 ATen reductions; here one scale is a streaming reduction (+ one more pass for BerHu, whose threshold is a global
// max) and the backward pass is one elementwise kernel driven by a few device-side scalars -- no sync, no gather.
// HBM-bound: 8 B/pixel forward (pred + gt), 12 B/pixel backward.
//
//   d = pred - gt, n = number of (valid) pixels
//   l1      mean |d|                                  d/dpred = sign(d) / n
//   mse     mean d^2                                  2 d / n
//   abs_rel mean |d| / pred                           (sign(d) pred - |d|) / pred^2 / n
//   berhu   c = 0.2 * max(d);  (sum |d| + sum_{|d|>c} d^2) / (n + #{|d|>c})      (sign(d) + [|d|>c] 2 d) / (n + n2)
//   silog   l = log pred - log gt;  10 sqrt(mean l^2 - 0.85 mean(l)^2)           10 (l - 0.85 mean l) / (sqrt(.) n pred)
// An empty selection gives NaN, like torch.mean of an empty tensor.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

enum { SUP_L1 = 0, SUP_MSE = 1, SUP_ABS_REL = 2, SUP_BERHU = 3, SUP_SILOG = 4 };

__device__ __forceinline__ float sup_wave_sum(float v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  return v;
}
__device__ __forceinline__ float sup_wave_max(float v) {
  for (int d = 32; d >= 1; d >>= 1) { const float o = __shfl_down(v, d); v = o > v ? o : v; }
  return v;
}

// ws (double[8 + kSupMaxBlocks]): [0] sum a, [1] sum b, [2] count, [4..6] written by the finish kernel (loss, c0, c1),
// [8 + j] = max(d) seen by block j of BerHu's first pass (every block writes its slot: no initialisation, no atomics).
// pass = 0: everything except BerHu's thresholded sums; pass = 1 (BerHu only): c = thr * max_j ws[8+j], then
// a = |d| (+ d^2 above c), b = #{|d| > c}.
constexpr int kSupMaxBlocks = 1024;

__device__ float sup_block_max_of_partials(const double* ws, int nblk) {
  __shared__ float smax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -INFINITY;
  for (int j = tid; j < nblk; j += 256) { const float v = (float)ws[8 + j]; m = v > m ? v : m; }
  m = sup_wave_max(m);
  if (lane == 0) smax[wave] = m;
  __syncthreads();
  float r = smax[0];
  for (int w = 1; w < 4; ++w) r = smax[w] > r ? smax[w] : r;
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256) supervised_reduce_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                                 double* __restrict__ ws, size_t n, int method, int sparse,
                                                                 int pass, float berhu_thr) {
  __shared__ float red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float a = 0.f, b = 0.f, cnt = 0.f, mx = -INFINITY;
  float c = 0.f;
  if (method == SUP_BERHU && pass == 1) c = berhu_thr * sup_block_max_of_partials(ws, (int)gridDim.x);
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) {
    const float p = pred[i], g = gt[i];
    if (sparse && !(g > 0.f)) continue;
    const float d = p - g, ad = fabsf(d);
    cnt += 1.f;
    if (method == SUP_L1) a += ad;
    else if (method == SUP_MSE) a += d * d;
    else if (method == SUP_ABS_REL) a += ad / p;
    else if (method == SUP_SILOG) { const float l = logf(p) - logf(g); a += l * l; b += l; }
    else if (pass == 0) mx = d > mx ? d : mx;
    else { a += ad; if (ad > c) { a += d * d; b += 1.f; } }
  }
  a = sup_wave_sum(a); b = sup_wave_sum(b); cnt = sup_wave_sum(cnt); mx = sup_wave_max(mx);
  if (lane == 0) { red[wave][0] = a; red[wave][1] = b; red[wave][2] = cnt; red[wave][3] = mx; }
  __syncthreads();
  if (tid == 0) {
    double sa = 0, sb = 0, sc = 0;
    float m = -INFINITY;
    for (int w = 0; w < 4; ++w) { sa += red[w][0]; sb += red[w][1]; sc += red[w][2]; m = red[w][3] > m ? red[w][3] : m; }
    if (!(method == SUP_BERHU && pass == 0)) { atomicAdd(&ws[0], sa); atomicAdd(&ws[1], sb); }
    if (pass == 0) {
      atomicAdd(&ws[2], sc);
      if (method == SUP_BERHU) ws[8 + blockIdx.x] = (double)m;
    }
  }
}

__global__ void supervised_finish_kernel(double* __restrict__ ws, float* __restrict__ loss, int method, float berhu_thr,
                                         float silog_ratio, float silog_ratio2, int nblk) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double a = ws[0], b = ws[1], n = ws[2];
  double L, c0 = 0, c1 = 0;
  if (method == SUP_BERHU) {
    L = a / (n + b);
    c0 = 1.0 / (n + b);
    float m = -INFINITY;
    for (int j = 0; j < nblk; ++j) { const float v = (float)ws[8 + j]; m = v > m ? v : m; }
    c1 = (double)(berhu_thr * m);
  } else if (method == SUP_SILOG) {
    const double m = b / n, var = a / n - (double)silog_ratio2 * m * m;
    const double sq = sqrt(var);
    L = sq * silog_ratio;
    c0 = (double)silog_ratio / (sq * n);
    c1 = (double)silog_ratio2 * m;
  } else {
    L = a / n;
    c0 = (method == SUP_MSE ? 2.0 : 1.0) / n;
  }
  ws[4] = L; ws[5] = c0; ws[6] = c1;
  *loss = (float)L;
}

__global__ void __launch_bounds__(256) supervised_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                              const double* __restrict__ ws, const float* __restrict__ gout,
                                                              float* __restrict__ dpred, size_t n, int method, int sparse) {
  const float c0 = (float)ws[5] * gout[0], c1 = (float)ws[6];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float p = pred[i], g = gt[i];
    float r = 0.f;
    if (!(sparse && !(g > 0.f))) {
      const float d = p - g, ad = fabsf(d);
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      if (method == SUP_L1) r = sg * c0;
      else if (method == SUP_MSE) r = d * c0;
      else if (method == SUP_ABS_REL) r = (sg * p - ad) / (p * p) * c0;
      else if (method == SUP_BERHU) r = (sg + (ad > c1 ? 2.f * d : 0.f)) * c0;
      else r = (logf(p) - logf(g) - c1) * c0 / p;
    }
    dpred[i] = r;
  }
}

static int sup_grid(size_t n) {
  size_t g = (n + 255) / 256;
  if (g > kSupMaxBlocks) g = kSupMaxBlocks;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_supervised_loss_forward(const float* pred, const float* gt, float* loss, double* ws, size_t n, int method,
                                  int sparse, void* stream) {
  if (method < 0 || method > 4) { set_error("supervised_loss: unknown method %d", method); return -1; }
  hipStream_t s = (hipStream_t)stream;
  int e = (int)hipMemsetAsync(ws, 0, 8 * sizeof(double), s);
  if (e) { set_error("supervised_loss: memset failed"); return e; }
  const float thr = 0.2f, ratio = 10.f, ratio2 = 0.85f;   // BerHuLoss(threshold=0.2), SilogLoss(ratio=10, ratio2=0.85)
  const int nblk = sup_grid(n);
  PNSFM_LAUNCH(supervised_reduce_kernel, dim3(nblk), dim3(256), 0, s, pred, gt, ws, n, method, sparse, 0, thr);
  if (method == SUP_BERHU)
    PNSFM_LAUNCH(supervised_reduce_kernel, dim3(nblk), dim3(256), 0, s, pred, gt, ws, n, method, sparse, 1, thr);
  PNSFM_LAUNCH(supervised_finish_kernel, dim3(1), dim3(64), 0, s, ws, loss, method, thr, ratio, ratio2, nblk);
  return check_launch("supervised_loss_forward");
}

int pnsfm_supervised_loss_backward(const float* pred, const float* gt, const double* ws, const float* grad_out, float* dpred,
                                   size_t n, int method, int sparse, void* stream) {
  if (method < 0 || method > 4) { set_error("supervised_loss: unknown method %d", method); return -1; }
  PNSFM_LAUNCH(supervised_bwd_kernel, dim3(sup_grid(n)), dim3(256), 0, (hipStream_t)stream, pred, gt, ws, grad_out, dpred, n,
               method, sparse);
  return check_launch("supervised_loss_backward");
}

}  // extern "C"
