// groupnorm.hip -- GroupNorm(G) + {identity, ELU, ReLU}, optional residual add in front; forward + backward.
//
// Replaces torch.nn.GroupNorm(16, C) + nn.ELU(inplace=True) of Conv2D
//   (/root/reference/packnet_sfm/networks/layers/packnet/layers01.py:31-32,36-37), the residual form
//   `activ(normalize(x_out + shortcut))` of ResidualConv (:61-62,72) and GroupNorm+ReLU of PoseNet's conv_gn
//   (/root/reference/packnet_sfm/networks/pose/PoseNet.py:28-34).
// HBM-bound: forward reads x (+res) twice and writes y once (12 or 20 B/element); statistics are
// accumulated in fp64 (sum, sum of squares) so that var = E[x^2]-E[x]^2 is safe.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

__device__ __forceinline__ double block_sum_256(double v, double* red) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// stats[((b*G+g)*nsplit + s)*2 + {0,1}] = {sum, sumsq} over slice s of the group's contiguous (C/G)*HW elements
// (one slot per block: no zero-fill, no atomics, deterministic; the consumer adds the <= 64 partials)
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        double* __restrict__ stats, long n_per_group, int nsplit) {
  __shared__ double red[4];
  const int bg = blockIdx.x, s = blockIdx.y;
  const long per = ((n_per_group + nsplit - 1) / nsplit + 3) & ~3L;
  const long beg = s * per;
  long end = beg + per;
  if (end > n_per_group) end = n_per_group;
  const float* xp = x + (size_t)bg * n_per_group;
  const float* rp = res ? res + (size_t)bg * n_per_group : nullptr;
  double s1 = 0.0, s2 = 0.0;
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    float v = xp[i];
    if (rp) v += rp[i];
    s1 += (double)v;
    s2 += (double)v * (double)v;
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    stats[((size_t)bg * nsplit + s) * 2 + 0] = s1;
    stats[((size_t)bg * nsplit + s) * 2 + 1] = s2;
  }
}

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == 1) return z > 0.f ? z : expm1f(z);
  if (act == 2) return z > 0.f ? z : 0.f;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == 1) return z > 0.f ? 1.f : expf(z);
  if (act == 2) return z > 0.f ? 1.f : 0.f;
  return 1.f;
}

// grid: (B*C, chunks over HW)
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const double* __restrict__ stats, float* __restrict__ y,
                                                        float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                        int C, int HW, int G, float eps, int act, int chunk, int nsplit) {
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const int cpg = C / G, g = c / cpg;
  const double n = (double)cpg * (double)HW;
  double t1 = 0.0, t2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    t1 += stats[((size_t)(b * G + g) * nsplit + k) * 2 + 0];
    t2 += stats[((size_t)(b * G + g) * nsplit + k) * 2 + 1];
  }
  const double m = t1 / n;
  double var = t2 / n - m * m;
  if (var < 0.0) var = 0.0;
  const float mean = (float)m;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (blockIdx.y == 0 && threadIdx.x == 0 && c == g * cpg) {
    mean_out[b * G + g] = mean;
    rstd_out[b * G + g] = rstd;
  }
  const float ga = gamma[c], be = beta[c];
  const size_t base = (size_t)bc * HW;
  const int beg = blockIdx.y * chunk;
  int end = beg + chunk;
  if (end > HW) end = HW;
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    float v = x[base + i];
    if (res) v += res[base + i];
    const float z = (v - mean) * rstd * ga + be;
    y[base + i] = act_fwd(z, act);
  }
}

// red[((b*C+c)*nchunk + chunk)*2 + {0,1}] = { sum dz, sum dz * xhat } over one chunk of HW;  dz = dy * act'(z)
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ res, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, double* __restrict__ red_ws,
                                                             int C, int HW, int G, int act, int chunk) {
  __shared__ double red[4];
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const int g = c / (C / G);
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const float ga = gamma[c], be = beta[c];
  const size_t base = (size_t)bc * HW;
  const int beg = blockIdx.y * chunk;
  int end = beg + chunk;
  if (end > HW) end = HW;
  double s1 = 0.0, s2 = 0.0;
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    float v = x[base + i];
    if (res) v += res[base + i];
    const float xh = (v - mu) * rs;
    const float dz = dy[base + i] * act_grad(xh * ga + be, act);
    s1 += (double)dz;
    s2 += (double)dz * (double)xh;
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    red_ws[((size_t)bc * gridDim.y + blockIdx.y) * 2 + 0] = s1;
    red_ws[((size_t)bc * gridDim.y + blockIdx.y) * 2 + 1] = s2;
  }
}

// dx = rstd * (dz*gamma - mean_g(dz*gamma) - xhat * mean_g(dz*gamma*xhat))
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ res, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const double* __restrict__ red_ws,
                                                            float* __restrict__ dx, int C, int HW, int G, int act, int chunk) {
  const int bc = blockIdx.x;
  const int b = bc / C, c = bc - b * C;
  const int cpg = C / G, g = c / cpg;
  double A = 0.0, Bq = 0.0;
  const int nchunk = gridDim.y;
  for (int k = 0; k < cpg; ++k) {
    const int cc = g * cpg + k;
    const double gk = (double)gamma[cc];
    double r1 = 0.0, r2 = 0.0;
    for (int j = 0; j < nchunk; ++j) {
      r1 += red_ws[(((size_t)b * C + cc) * nchunk + j) * 2 + 0];
      r2 += red_ws[(((size_t)b * C + cc) * nchunk + j) * 2 + 1];
    }
    A += gk * r1;
    Bq += gk * r2;
  }
  const double n = (double)cpg * (double)HW;
  const float mA = (float)(A / n), mB = (float)(Bq / n);
  const float mu = mean[b * G + g], rs = rstd[b * G + g];
  const float ga = gamma[c], be = beta[c];
  const size_t base = (size_t)bc * HW;
  const int beg = blockIdx.y * chunk;
  int end = beg + chunk;
  if (end > HW) end = HW;
  for (int i = beg + threadIdx.x; i < end; i += 256) {
    float v = x[base + i];
    if (res) v += res[base + i];
    const float xh = (v - mu) * rs;
    const float dz = dy[base + i] * act_grad(xh * ga + be, act);
    dx[base + i] = rs * (dz * ga - mA - xh * mB);
  }
}

__global__ void __launch_bounds__(256) gn_bwd_params_kernel(const double* __restrict__ red_ws, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, int B, int C, int nchunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < nchunk; ++j) {
      s1 += red_ws[(((size_t)b * C + c) * nchunk + j) * 2 + 0];
      s2 += red_ws[(((size_t)b * C + c) * nchunk + j) * 2 + 1];
    }
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
}

static int pick_chunk(int HW, int rows) {
  // aim for >= ~1024 blocks overall, chunks a multiple of 256 elements, at least 1024 elements each
  int want = ceil_div(1024, rows);
  if (want < 1) want = 1;
  if (want > PNSFM_GN_MAX_SPLIT) want = PNSFM_GN_MAX_SPLIT;
  int chunk = ceil_div(HW, want);
  if (chunk < 1024) chunk = 1024;
  chunk = round_up(chunk, 256);
  return chunk;
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_groupnorm_act_forward(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                float* mean, float* rstd, double* stats_ws, int B, int C, int HW, int G, float eps,
                                int act, void* stream) {
  if (C % G != 0 || B <= 0 || HW <= 0) { set_error("groupnorm_forward: bad shape C=%d G=%d", C, G); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const long npg = (long)(C / G) * HW;
  int nsplit = ceil_div(1024, B * G);
  const int max_split = (int)((npg + 2047) / 2048);
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit > PNSFM_GN_MAX_SPLIT) nsplit = PNSFM_GN_MAX_SPLIT;
  if (nsplit < 1) nsplit = 1;
  PNSFM_LAUNCH(gn_stats_kernel, dim3(B * G, nsplit), dim3(256), 0, s, x, res, stats_ws, npg, nsplit);
  int e = check_launch("gn_stats");
  if (e) return e;
  const int chunk = pick_chunk(HW, B * C);
  PNSFM_LAUNCH(gn_apply_kernel, dim3(B * C, ceil_div(HW, chunk)), dim3(256), 0, s, x, res, gamma, beta,
               (const double*)stats_ws, y, mean, rstd, C, HW, G, eps, act, chunk, nsplit);
  return check_launch("gn_apply");
}

int pnsfm_groupnorm_act_backward(const float* dy, const float* x, const float* res, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, float* dx, float* dgamma,
                                 float* dbeta, double* red_ws, int B, int C, int HW, int G, int act, void* stream) {
  if (C % G != 0 || B <= 0 || HW <= 0) { set_error("groupnorm_backward: bad shape C=%d G=%d", C, G); return -1; }
  hipStream_t s = (hipStream_t)stream;
  int e = 0;
  const int chunk = pick_chunk(HW, B * C);
  dim3 grid(B * C, ceil_div(HW, chunk));
  if ((int)grid.y > PNSFM_GN_MAX_SPLIT) { set_error("groupnorm_backward: too many chunks"); return -1; }
  PNSFM_LAUNCH(gn_bwd_reduce_kernel, grid, dim3(256), 0, s, dy, x, res, gamma, beta, mean, rstd, red_ws, C, HW, G, act, chunk);
  e = check_launch("gn_bwd_reduce");
  if (e) return e;
  PNSFM_LAUNCH(gn_bwd_apply_kernel, grid, dim3(256), 0, s, dy, x, res, gamma, beta, mean, rstd, (const double*)red_ws, dx,
               C, HW, G, act, chunk);
  e = check_launch("gn_bwd_apply");
  if (e) return e;
  PNSFM_LAUNCH(gn_bwd_params_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, (const double*)red_ws, dgamma, dbeta, B, C, (int)grid.y);
  return check_launch("gn_bwd_params");
}

}  // extern "C"
