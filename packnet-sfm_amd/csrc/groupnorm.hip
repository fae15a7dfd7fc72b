// groupnorm.hip -- GroupNorm(G) + {identity, ELU, ReLU}, optional residual add in front; forward + backward.
//
// Replaces torch.nn.GroupNorm(16, C) + nn.ELU(inplace=True) of Conv2D
//   (/root/reference/packnet_sfm/networks/layers/packnet/layers01.py:31-32,36-37), the residual form
//   `activ(normalize(x_out + shortcut))` of ResidualConv (:61-62,72) and GroupNorm+ReLU of PoseNet's conv_gn
//   (/root/reference/packnet_sfm/networks/pose/PoseNet.py:28-34).
// HBM-bound: forward reads x (+res) twice and writes y once (12 or 20 B/element), backward reads dy, x (+res) twice and
// writes dx once (24 B/element).  Statistics are accumulated in fp64 (sum, sum of squares) so that var = E[x^2]-E[x]^2 is safe.
//
// Layout of the work (round 2: the round-1 kernels moved 4 bytes per lane per load and gave a workgroup 1024 elements; they
// ran at 8-26 % of HBM bandwidth and were launch-bound on the 24x80 and smaller maps):
//   * every load/store is a float4 when HW % 4 == 0 (all PackNet01 shapes), scalar otherwise;
//   * a workgroup owns ROWS x one pixel chunk, ROWS = channels (b, c) handled by 256/T threads groups of T lanes each, so
//     that small maps (6x20: 30 float4 per channel) still give every lane work -- T = threads per channel row (power of 2);
//   * per-(sample, group) statistics and per-(sample, channel) gradient sums are one slot per workgroup row/chunk: no zero
//     fill, no atomics, deterministic; the consumer adds the <= PNSFM_GN_MAX_SPLIT partials;
//   * round 3: TWO launches per direction instead of three / four.  The bookkeeping kernels of round 2 (gn_finish: mean / rstd
//     of every group from the slots; gn_bwd_group: group means of the gradient sums, dgamma / dbeta) moved 0 bytes and cost
//     4.8-7.3 us of launch + dependency latency each, 108 of them per training step (0.65 ms): every workgroup of the apply
//     kernels now adds the few partial slots of its own rows itself (all of them in the same order: identical bits), and the
//     rows of sample 0 / chunk 0 also write mean, rstd, dgamma, dbeta.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

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

struct GnGeom {
  int T;        // lanes per channel row (power of two, <= 256)
  int rows;     // channel rows per workgroup = 256 / T
  int chunk;    // elements of a row handled by one workgroup (multiple of 4*T, or HW)
  int nchunk;   // chunks per row
};

// sum over the T lanes of a row (T a power of two <= 64: shuffles inside the wave; T > 64: through LDS)
__device__ __forceinline__ double row_sum(double v, int T, double* red) {
  const int tid = threadIdx.x;
  if (T <= 64) {
    for (int d = T >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
  }
  // T = 128 or 256: lanes of a row span T/64 waves
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int wave = tid >> 6, wpr = T >> 6;          // waves per row
  __syncthreads();
  if ((tid & 63) == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  const int w0 = (wave / wpr) * wpr;
  for (int k = 0; k < wpr; ++k) s += red[w0 + k];
  return s;
}

// ---- forward statistics: stats[((bg)*nslot + slot)*2 + {0,1}] = {sum, sumsq} of one (channel row, chunk) piece of group bg;
//      nslot = cpg * nchunk (channels per group x chunks)
template <bool VEC>
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        double* __restrict__ stats, int BC, int C, int HW, int G, GnGeom g) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int r = tid / g.T, l = tid - r * g.T;
  const int bc = blockIdx.x * g.rows + r;
  const int beg = blockIdx.y * g.chunk;
  int end = beg + g.chunk;
  if (end > HW) end = HW;
  double s1 = 0.0, s2 = 0.0;
  if (bc < BC) {
    const float* xp = x + (size_t)bc * HW;
    const float* rp = res ? res + (size_t)bc * HW : nullptr;
    if (VEC) {
      for (int i = beg + 4 * l; i < end; i += 4 * g.T) {
        float4 v = *reinterpret_cast<const float4*>(xp + i);
        if (rp) { const float4 q = *reinterpret_cast<const float4*>(rp + i); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        s1 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        s2 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
      }
    } else {
      for (int i = beg + l; i < end; i += g.T) {
        float v = xp[i];
        if (rp) v += rp[i];
        s1 += (double)v;
        s2 += (double)v * (double)v;
      }
    }
  }
  s1 = row_sum(s1, g.T, red);
  s2 = row_sum(s2, g.T, red);
  if (l == 0 && bc < BC) {
    const int cpg = C / G;
    const int b = bc / C, c = bc - b * C;
    const int gi = c / cpg, cl = c - gi * cpg;
    const size_t slot = ((size_t)(b * G + gi) * (cpg * g.nchunk) + (size_t)cl * g.nchunk + blockIdx.y) * 2;
    stats[slot] = s1;
    stats[slot + 1] = s2;
  }
}

// pass 2: every row adds the partial slots of its group (lanes of the row share the slots, fp64, fixed order), derives
// mean / rstd exactly like every other row of that group, and normalises + activates its chunk.  The row of the group's first
// channel in chunk 0 stores mean / rstd for the backward pass.
template <bool VEC>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const double* __restrict__ stats, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out, float* __restrict__ y, int BC, int C, int HW,
                                                        int G, int act, double n, float eps, GnGeom g, int nslot) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int r = tid / g.T, l = tid - r * g.T;
  const int bc_raw = blockIdx.x * g.rows + r;
  const bool valid = bc_raw < BC;
  const int bc = valid ? bc_raw : BC - 1;           // surplus rows of the last workgroup shadow a real one (barriers stay uniform)
  const int cpg = C / G;
  const int b = bc / C, c = bc - b * C;
  const int gi = c / cpg;
  // (nslot: partial {sum, sumsq} pairs per (sample, group) -- cpg * nchunk from gn_stats_kernel, or whatever the producing conv
  // kernel's epilogue wrote: pnsfm_conv2d_forward_gn)
  const double* sp = stats + (size_t)(b * G + gi) * nslot * 2;
  double t1 = 0.0, t2 = 0.0;
  for (int k = l; k < nslot; k += g.T) { t1 += sp[2 * k]; t2 += sp[2 * k + 1]; }
  t1 = row_sum(t1, g.T, red);
  t2 = row_sum(t2, g.T, red);
  const double m = t1 / n;
  double var = t2 / n - m * m;
  if (var < 0.0) var = 0.0;
  const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (!valid) return;
  if (l == 0 && c == gi * cpg && blockIdx.y == 0) { mean_out[b * G + gi] = mu; rstd_out[b * G + gi] = rs; }
  const float sc = rs * gamma[c], sh = beta[c] - mu * sc;         // z = v * sc + sh
  const size_t base = (size_t)bc * HW;
  const int beg = blockIdx.y * g.chunk;
  int end = beg + g.chunk;
  if (end > HW) end = HW;
  if (VEC) {
    for (int i = beg + 4 * l; i < end; i += 4 * g.T) {
      float4 v = *reinterpret_cast<const float4*>(x + base + i);
      if (res) { const float4 q = *reinterpret_cast<const float4*>(res + base + i); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      float4 o;
      o.x = act_fwd(fmaf(v.x, sc, sh), act); o.y = act_fwd(fmaf(v.y, sc, sh), act);
      o.z = act_fwd(fmaf(v.z, sc, sh), act); o.w = act_fwd(fmaf(v.w, sc, sh), act);
      *reinterpret_cast<float4*>(y + base + i) = o;
    }
  } else {
    for (int i = beg + l; i < end; i += g.T) {
      float v = x[base + i];
      if (res) v += res[base + i];
      y[base + i] = act_fwd(fmaf(v, sc, sh), act);
    }
  }
}

// ---- backward, pass 1: red[((b*C+c)*nchunk + chunk)*2 + {0,1}] = { sum dz, sum dz * xhat };  dz = dy * act'(z)
template <bool VEC>
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ res, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, double* __restrict__ red_ws,
                                                             int BC, int C, int HW, int G, int act, GnGeom g) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int r = tid / g.T, l = tid - r * g.T;
  const int bc = blockIdx.x * g.rows + r;
  double s1 = 0.0, s2 = 0.0;
  if (bc < BC) {
    const int b = bc / C, c = bc - b * C;
    const int gi = c / (C / G);
    const float mu = mean[b * G + gi], rs = rstd[b * G + gi];
    const float ga = gamma[c], be = beta[c];
    const size_t base = (size_t)bc * HW;
    const int beg = blockIdx.y * g.chunk;
    int end = beg + g.chunk;
    if (end > HW) end = HW;
    if (VEC) {
      for (int i = beg + 4 * l; i < end; i += 4 * g.T) {
        float4 v = *reinterpret_cast<const float4*>(x + base + i);
        if (res) { const float4 q = *reinterpret_cast<const float4*>(res + base + i); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        const float4 d = *reinterpret_cast<const float4*>(dy + base + i);
        const float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = (vv[k] - mu) * rs;
          const float dz = dd[k] * act_grad(fmaf(xh, ga, be), act);
          s1 += (double)dz;
          s2 += (double)dz * (double)xh;
        }
      }
    } else {
      for (int i = beg + l; i < end; i += g.T) {
        float v = x[base + i];
        if (res) v += res[base + i];
        const float xh = (v - mu) * rs;
        const float dz = dy[base + i] * act_grad(fmaf(xh, ga, be), act);
        s1 += (double)dz;
        s2 += (double)dz * (double)xh;
      }
    }
  }
  s1 = row_sum(s1, g.T, red);
  s2 = row_sum(s2, g.T, red);
  if (l == 0 && bc < BC) {
    red_ws[((size_t)bc * g.nchunk + blockIdx.y) * 2 + 0] = s1;
    red_ws[((size_t)bc * g.nchunk + blockIdx.y) * 2 + 1] = s2;
  }
}

// ---- backward, pass 2: dx = rstd * (dz*gamma - mean_g(dz*gamma) - xhat * mean_g(dz*gamma*xhat)).  Every row first adds the
// pass-1 slots of ITS group (channels of the group x chunks, weighted by gamma: fp64, lanes of the row share the work, the same
// order in every row of the group); the rows of sample 0 in chunk 0 also finish dgamma / dbeta of their channel (sum of the
// channel's slots over the batch) -- what gn_bwd_group_kernel / gn_bwd_params_kernel did in two further launches.
template <bool VEC>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ res, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const double* __restrict__ red_ws,
                                                            float* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int B, int C, int HW, int G, int act,
                                                            double n, GnGeom g) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int r = tid / g.T, l = tid - r * g.T;
  const int BC = B * C;
  const int bc_raw = blockIdx.x * g.rows + r;
  const bool valid = bc_raw < BC;
  const int bc = valid ? bc_raw : BC - 1;
  const int cpg = C / G;
  const int b = bc / C, c = bc - b * C;
  const int gi = c / cpg;
  double A = 0.0, Bq = 0.0;
  for (int idx = l; idx < cpg * g.nchunk; idx += g.T) {
    const int k = idx / g.nchunk;                    // channel of the group, chunk
    const int c2 = gi * cpg + k;
    const double* p = red_ws + (((size_t)b * C + c2) * g.nchunk + (idx - k * g.nchunk)) * 2;
    const double ga2 = (double)gamma[c2];
    A += ga2 * p[0];
    Bq += ga2 * p[1];
  }
  A = row_sum(A, g.T, red);
  Bq = row_sum(Bq, g.T, red);
  // channel totals over the batch (only the rows of sample 0, chunk 0 keep them; cheap enough to stay unconditional, which
  // keeps row_sum's barriers uniform)
  double s1 = 0.0, s2 = 0.0;
  for (int idx = l; idx < B * g.nchunk; idx += g.T) {
    const int b2 = idx / g.nchunk;
    const double* p = red_ws + (((size_t)b2 * C + c) * g.nchunk + (idx - b2 * g.nchunk)) * 2;
    s1 += p[0];
    s2 += p[1];
  }
  s1 = row_sum(s1, g.T, red);
  s2 = row_sum(s2, g.T, red);
  if (!valid) return;
  if (l == 0 && b == 0 && blockIdx.y == 0) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
  const float mA = (float)(A / n), mB = (float)(Bq / n);
  const float mu = mean[b * G + gi], rs = rstd[b * G + gi];
  const float ga = gamma[c], be = beta[c];
  const size_t base = (size_t)bc * HW;
  const int beg = blockIdx.y * g.chunk;
  int end = beg + g.chunk;
  if (end > HW) end = HW;
  if (VEC) {
    for (int i = beg + 4 * l; i < end; i += 4 * g.T) {
      float4 v = *reinterpret_cast<const float4*>(x + base + i);
      if (res) { const float4 q = *reinterpret_cast<const float4*>(res + base + i); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
      const float4 d = *reinterpret_cast<const float4*>(dy + base + i);
      const float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w};
      float oo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (vv[k] - mu) * rs;
        const float dz = dd[k] * act_grad(fmaf(xh, ga, be), act);
        oo[k] = rs * (dz * ga - mA - xh * mB);
      }
      *reinterpret_cast<float4*>(dx + base + i) = make_float4(oo[0], oo[1], oo[2], oo[3]);
    }
  } else {
    for (int i = beg + l; i < end; i += g.T) {
      float v = x[base + i];
      if (res) v += res[base + i];
      const float xh = (v - mu) * rs;
      const float dz = dy[base + i] * act_grad(fmaf(xh, ga, be), act);
      dx[base + i] = rs * (dz * ga - mA - xh * mB);
    }
  }
}

// Work split: T lanes per channel row so that a lane moves >= 4 float4 per pass when the map allows it; rows of one
// workgroup are consecutive (b, c) channels; a row is cut into chunks only when B*C rows alone cannot give ~4 workgroups per CU.
static GnGeom gn_geom(int BC, int HW, bool vec, int max_chunks) {
  GnGeom g;
  const int units = vec ? HW / 4 : HW;                 // loads per row
  int T = 256;
  while (T > 1 && units < 4 * T) T >>= 1;             // >= 4 loads per lane, down to 1 lane per row for tiny maps
  if (T > 256) T = 256;
  g.T = T;
  g.rows = 256 / T;
  const int row_blocks = ceil_div(BC, g.rows);
  int want = ceil_div(1024, row_blocks);               // ~1024 workgroups overall
  if (want < 1) want = 1;
  if (want > max_chunks) want = max_chunks;
  const int per_pass = (vec ? 4 : 1) * T;              // elements one pass of the row's lanes covers
  int chunk = round_up(ceil_div(HW, want), per_pass);
  const int min_chunk = 8 * per_pass;                  // >= 8 passes per lane before a row is cut
  if (chunk < min_chunk) chunk = min_chunk;
  if (chunk >= HW) chunk = HW;
  g.chunk = chunk;
  g.nchunk = ceil_div(HW, chunk);
  return g;
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

// workspace (doubles) the forward / backward entry points need for a [B, C, HW] tensor with G groups
size_t pnsfm_groupnorm_ws_doubles(int B, int C, int G) {
  // forward: 2 per (b, c, chunk); backward: 2 per (b, c, chunk) + 2 per (b, c) + 2 floats per (b, g) (rounded up)
  return (size_t)2 * B * C * PNSFM_GN_MAX_SPLIT + (size_t)2 * B * C + (size_t)B * G + 16;
}

int pnsfm_groupnorm_act_forward(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                float* mean, float* rstd, double* stats_ws, int B, int C, int HW, int G, float eps,
                                int act, void* stream) {
  if (C % G != 0 || B <= 0 || HW <= 0) { set_error("groupnorm_forward: bad shape C=%d G=%d", C, G); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (HW % 4 == 0);
  const int BC = B * C, cpg = C / G;
  const GnGeom g = gn_geom(BC, HW, vec, PNSFM_GN_MAX_SPLIT);
  dim3 grid(ceil_div(BC, g.rows), g.nchunk);
  if (vec) PNSFM_LAUNCH((gn_stats_kernel<true>), grid, dim3(256), 0, s, x, res, stats_ws, BC, C, HW, G, g);
  else PNSFM_LAUNCH((gn_stats_kernel<false>), grid, dim3(256), 0, s, x, res, stats_ws, BC, C, HW, G, g);
  int e = check_launch("gn_stats");
  if (e) return e;
  const double n = (double)cpg * (double)HW;
  if (vec) PNSFM_LAUNCH((gn_apply_kernel<true>), grid, dim3(256), 0, s, x, res, gamma, beta, (const double*)stats_ws, mean, rstd, y, BC, C, HW, G, act, n, eps, g, cpg * g.nchunk);
  else PNSFM_LAUNCH((gn_apply_kernel<false>), grid, dim3(256), 0, s, x, res, gamma, beta, (const double*)stats_ws, mean, rstd, y, BC, C, HW, G, act, n, eps, g, cpg * g.nchunk);
  return check_launch("gn_apply");
}

// Second half of the forward pass alone: the statistics were left behind by the producing convolution (pnsfm_conv2d_forward_gn:
// stats[(b G + g)][nslot][2] doubles).  One launch per layer instead of two, and y is read once instead of twice.
int pnsfm_groupnorm_act_apply(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                              const double* stats, int nslot, int B, int C, int HW, int G, float eps, int act, void* stream) {
  if (C % G != 0 || B <= 0 || HW <= 0 || nslot <= 0) { set_error("groupnorm_apply: bad shape C=%d G=%d nslot=%d", C, G, nslot); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (HW % 4 == 0);
  const int BC = B * C;
  const GnGeom g = gn_geom(BC, HW, vec, PNSFM_GN_MAX_SPLIT);
  dim3 grid(ceil_div(BC, g.rows), g.nchunk);
  const double n = (double)(C / G) * (double)HW;
  const float* res = nullptr;
  if (vec) PNSFM_LAUNCH((gn_apply_kernel<true>), grid, dim3(256), 0, s, x, res, gamma, beta, stats, mean, rstd, y, BC, C, HW, G, act, n, eps, g, nslot);
  else PNSFM_LAUNCH((gn_apply_kernel<false>), grid, dim3(256), 0, s, x, res, gamma, beta, stats, mean, rstd, y, BC, C, HW, G, act, n, eps, g, nslot);
  return check_launch("gn_apply");
}

int pnsfm_groupnorm_act_backward(const float* dy, const float* x, const float* res, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, float* dx, float* dgamma,
                                 float* dbeta, double* red_ws, int B, int C, int HW, int G, int act, void* stream) {
  if (C % G != 0 || B <= 0 || HW <= 0) { set_error("groupnorm_backward: bad shape C=%d G=%d", C, G); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (HW % 4 == 0);
  const int BC = B * C;
  const GnGeom g = gn_geom(BC, HW, vec, PNSFM_GN_MAX_SPLIT);
  dim3 grid(ceil_div(BC, g.rows), g.nchunk);
  int e = 0;
  if (vec) PNSFM_LAUNCH((gn_bwd_reduce_kernel<true>), grid, dim3(256), 0, s, dy, x, res, gamma, beta, mean, rstd, red_ws, BC, C, HW, G, act, g);
  else PNSFM_LAUNCH((gn_bwd_reduce_kernel<false>), grid, dim3(256), 0, s, dy, x, res, gamma, beta, mean, rstd, red_ws, BC, C, HW, G, act, g);
  e = check_launch("gn_bwd_reduce");
  if (e) return e;
  const double n = (double)(C / G) * (double)HW;
  if (vec) PNSFM_LAUNCH((gn_bwd_apply_kernel<true>), grid, dim3(256), 0, s, dy, x, res, gamma, beta, mean, rstd, (const double*)red_ws, dx, dgamma, dbeta, B, C, HW, G, act, n, g);
  else PNSFM_LAUNCH((gn_bwd_apply_kernel<false>), grid, dim3(256), 0, s, dy, x, res, gamma, beta, mean, rstd, (const double*)red_ws, dx, dgamma, dbeta, B, C, HW, G, act, n, g);
  return check_launch("gn_bwd_apply");
}

}  // extern "C"
