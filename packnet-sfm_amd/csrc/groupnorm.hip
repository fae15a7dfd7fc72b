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

// exp(z) - 1 for z <= 0 (round 6: ocml's expm1f is ~30 instructions and the one-launch kernels below are VALU-bound -- a slab's
// workgroup does all of its arithmetic on ONE CU): a degree-7 Taylor polynomial on (-1/4, 0] (truncation < 4e-10 of |z|) and the
// hardware exponential minus one below that (result in [-1, -0.22): the exponential's 1-2 ulp are < 2e-7 of it).
__device__ __forceinline__ float gn_expm1_neg(float z) {
  if (z > -0.25f) {
    float p = 1.f / 5040.f;
    p = fmaf(p, z, 1.f / 720.f);
    p = fmaf(p, z, 1.f / 120.f);
    p = fmaf(p, z, 1.f / 24.f);
    p = fmaf(p, z, 1.f / 6.f);
    p = fmaf(p, z, 0.5f);
    p = fmaf(p, z, 1.f);
    return p * z;
  }
  return __expf(z) - 1.f;
}
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == 1) return z > 0.f ? z : gn_expm1_neg(z);
  if (act == 2) return z > 0.f ? z : 0.f;
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == 1) return z > 0.f ? 1.f : __expf(z);
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
  // (nslot: partial {sum, sumsq} pairs per (sample, group): cpg * nchunk from gn_stats_kernel)
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

// ---- round 6: ONE launch per direction when a (sample, group) slab fits one workgroup's reach ----------------------------------
// The channels of a group are consecutive in NCHW, so a slab is ONE contiguous run of n = (C / G) * HW floats.  On the maps of
// <= 48x160 (every layer of PackNet01 below the 96x320 level and all of PoseNet: ~40 of the step's GroupNorms) a slab is <= 240 KB and
// the two-launch forms above are latency: 8-15 us per launch for a few MB.  Here a workgroup of 1024 (256) threads owns a slab:
//   forward:  the slab is read ONCE into registers (NV float4 per thread), summed (fp32 over 16 elements, fp64 beyond), reduced across
//             the workgroup (shuffles + LDS), normalised, activated and stored -- one launch instead of two;
//   backward: blocks [0, B*G*S) own a slab: pass 1 forms A = sum gamma dz and Bq = sum gamma dz xhat of their slab, pass 2 re-reads x and
//             dy (L2-resident: the same workgroup touched them a moment ago) and writes dx; the blocks behind them own a few CHANNELS
//             each and form dgamma / dbeta over all samples -- every per-element quantity they need (mean, rstd, gamma, beta, x, dy) is
//             known without the slab sums, so nothing is handed from workgroup to workgroup: no atomics, no fences, a fixed order.
// A slab's arithmetic sits on ONE compute unit and is VALU-bound there (first build: 19.5 us forward / 35-45 us backward on the 48x160
// maps against 21 / 32 for the two-launch form, rocprofv3), hence (i) the cheap exponentials above, per-channel scale / shift from LDS,
// incremental channel indices, fp32 partial sums; (ii) S workgroups per slab on the larger slabs: each forms the slab's sums itself
// (identical order: identical bits) and normalises 1 / S of it -- the S parts of a slab land on one XCD (shared L2).
// Results agree with the two-launch form to summation order; each form is deterministic.  pnsfm_set_gn_fused(0) / PNSFM_GN_FUSED=0
// keeps the two-launch form everywhere.
constexpr int GNF_MAXCPG = 128;

__device__ __forceinline__ void gnf_block_sum2(double& a, double& b, double (*red)[16]) {
  for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
  const int nw = (int)blockDim.x >> 6, w = (int)threadIdx.x >> 6;
  if (nw == 1) return;
  __syncthreads();                      // (a previous use of `red` is over)
  if ((threadIdx.x & 63) == 0) { red[0][w] = a; red[1][w] = b; }
  __syncthreads();
  double sa = 0.0, sb = 0.0;
  for (int k = 0; k < nw; ++k) { sa += red[0][k]; sb += red[1][k]; }
  a = sa; b = sb;
}

// logical (slab, part) of a block: with S parts per slab and a slab count that is a multiple of 8, the parts of a slab are 8 blocks
// apart -- workgroups are dealt to the 8 XCDs round-robin, so they share an L2
__device__ __forceinline__ void gnf_slab_part(int blk, int S, int nslab, int& slab, int& part) {
  if (S > 1 && (nslab & 7) == 0) {
    const int q = blk / (8 * S), r = blk - q * (8 * S);
    part = r >> 3;
    slab = q * 8 + (r & 7);
  } else {
    slab = blk / S;
    part = blk - slab * S;
  }
}

// dq / dr: T / HW4 and T % HW4 (the float4 index advances by T per k: channel += dq, position += dr with one carry)
template <int NV, int S>
__global__ void __launch_bounds__(1024) gn_fused_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             float* __restrict__ y, int C, int HW, int G, int act, float eps, int nslab,
                                                             int dq, int dr) {
  __shared__ double red[2][16];
  __shared__ float lsc[GNF_MAXCPG], lsh[GNF_MAXCPG];
  const int tid = threadIdx.x, T = blockDim.x;
  int bg, part;
  gnf_slab_part((int)blockIdx.x, S, nslab, bg, part);
  const int cpg = C / G, gi = bg % G;
  const int HW4 = HW >> 2, n4 = cpg * HW4;
  const size_t base4 = (size_t)bg * n4;
  const float4* xp = reinterpret_cast<const float4*>(x) + base4;
  const float4* rp = res ? reinterpret_cast<const float4*>(res) + base4 : nullptr;
  float4 v[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + k * T;
    v[k] = i < n4 ? xp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (rp) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = tid + k * T;
      if (i < n4) { const float4 q = rp[i]; v[k].x += q.x; v[k].y += q.y; v[k].z += q.z; v[k].w += q.w; }
    }
  }
  double s1 = 0.0, s2 = 0.0;
  {
    float f1 = 0.f, f2 = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      f1 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      f2 += fmaf(v[k].x, v[k].x, v[k].y * v[k].y) + fmaf(v[k].z, v[k].z, v[k].w * v[k].w);
      if ((k & 3) == 3 || k == NV - 1) { s1 += (double)f1; s2 += (double)f2; f1 = 0.f; f2 = 0.f; }
    }
  }
  gnf_block_sum2(s1, s2, red);
  const double n = (double)cpg * (double)HW;
  const double m = s1 / n;
  double var = s2 / n - m * m;
  if (var < 0.0) var = 0.0;
  const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (tid == 0 && part == 0) { mean_out[bg] = mu; rstd_out[bg] = rs; }
  if (tid < cpg) {
    const float sc = rs * gamma[gi * cpg + tid];
    lsc[tid] = sc;
    lsh[tid] = beta[gi * cpg + tid] - mu * sc;
  }
  __syncthreads();
  float4* yp = reinterpret_cast<float4*>(y) + base4;
  int ch = tid / HW4, pos = tid - ch * HW4;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int i = tid + k * T;
    // part p normalises the k-range [p NV / S, (p + 1) NV / S)  (NV % S == 0 whenever S > 1)
    if ((S == 1 || k / (NV / S) == part) && i < n4) {
      const float sc = lsc[ch], sh = lsh[ch];
      float4 o;
      o.x = act_fwd(fmaf(v[k].x, sc, sh), act); o.y = act_fwd(fmaf(v[k].y, sc, sh), act);
      o.z = act_fwd(fmaf(v[k].z, sc, sh), act); o.w = act_fwd(fmaf(v[k].w, sc, sh), act);
      yp[i] = o;
    }
    ch += dq; pos += dr;
    if (pos >= HW4) { pos -= HW4; ++ch; }
  }
}

// TPC: threads per channel of a parameter block (power of two, 64 <= TPC <= blockDim); S: workgroups per slab
__global__ void __launch_bounds__(1024) gn_fused_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const float* __restrict__ res, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                             int HW, int G, int act, int TPC, int S, int dq, int dr) {
  __shared__ double red[2][16];
  __shared__ float lga[GNF_MAXCPG], lbe[GNF_MAXCPG];
  const int tid = threadIdx.x, T = blockDim.x;
  const int cpg = C / G, HW4 = HW >> 2;
  const int nslab = B * G;
  if ((int)blockIdx.x < nslab * S) {
    int bg, part;
    gnf_slab_part((int)blockIdx.x, S, nslab, bg, part);
    const int gi = bg % G;
    const int n4 = cpg * HW4;
    const size_t base4 = (size_t)bg * n4;
    const float4* xp = reinterpret_cast<const float4*>(x) + base4;
    const float4* rp = res ? reinterpret_cast<const float4*>(res) + base4 : nullptr;
    const float4* dp = reinterpret_cast<const float4*>(dy) + base4;
    const float mu = mean[bg], rs = rstd[bg];
    const float nmr = -mu * rs;                    // xhat = v * rs + nmr
    if (tid < cpg) { lga[tid] = gamma[gi * cpg + tid]; lbe[tid] = beta[gi * cpg + tid]; }
    __syncthreads();
    double A = 0.0, Bq = 0.0;
    {
      int ch = tid / HW4, pos = tid - ch * HW4;
      for (int i0 = tid; i0 < n4; i0 += 4 * T) {
        float4 v[4], d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * T;
          if (i < n4) { v[u] = xp[i]; d[u] = dp[i]; }
        }
        if (rp) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * T;
            if (i < n4) { const float4 q = rp[i]; v[u].x += q.x; v[u].y += q.y; v[u].z += q.z; v[u].w += q.w; }
          }
        }
        float fa = 0.f, fb = 0.f;                  // fp32 over the 16 elements of this trip, fp64 beyond
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * T;
          if (i < n4) {
            const float ga = lga[ch], be = lbe[ch];
            const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, dd[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float xh = fmaf(vv[k], rs, nmr);
              const float dz = dd[k] * act_grad(fmaf(xh, ga, be), act);
              a1 += dz;
              a2 = fmaf(dz, xh, a2);
            }
            fa = fmaf(ga, a1, fa);
            fb = fmaf(ga, a2, fb);
          }
          ch += dq; pos += dr;
          if (pos >= HW4) { pos -= HW4; ++ch; }
        }
        A += (double)fa;
        Bq += (double)fb;
      }
    }
    gnf_block_sum2(A, Bq, red);
    const double n = (double)cpg * (double)HW;
    const float mA = (float)(A / n), mB = (float)(Bq / n);
    float4* op = reinterpret_cast<float4*>(dx) + base4;
    // this part's share of the slab: a contiguous range of float4 (whole multiples of T except the last)
    const int per = ((n4 + S - 1) / S + T - 1) / T * T;
    const int lo = part * per;
    int hi = lo + per;
    if (hi > n4) hi = n4;
    for (int i0 = lo + tid; i0 < hi; i0 += 4 * T) {
      float4 v[4], d[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * T;
        if (i < hi) { v[u] = xp[i]; d[u] = dp[i]; }
      }
      if (rp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * T;
          if (i < hi) { const float4 q = rp[i]; v[u].x += q.x; v[u].y += q.y; v[u].z += q.z; v[u].w += q.w; }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * T;
        if (i < hi) {
          const int ch = i / HW4;
          const float ga = lga[ch], be = lbe[ch];
          const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, dd[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
          float oo[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xh = fmaf(vv[k], rs, nmr);
            const float dz = dd[k] * act_grad(fmaf(xh, ga, be), act);
            oo[k] = rs * (dz * ga - mA - xh * mB);
          }
          op[i] = make_float4(oo[0], oo[1], oo[2], oo[3]);
        }
      }
    }
    return;
  }
  // ---- parameter block: channels c0 + r, r = tid / TPC; TPC lanes walk the channel's B x HW elements; dbeta = sum dz, dgamma = sum dz xhat
  const int CPB = T / TPC;
  const int r = tid / TPC, l = tid - r * TPC;
  const int c_raw = ((int)blockIdx.x - nslab * S) * CPB + r;
  const bool valid = c_raw < C;
  const int c = valid ? c_raw : C - 1;          // surplus rows shadow a real channel: the barriers below stay uniform
  const int gi = c / cpg;
  const float ga = gamma[c], be = beta[c];
  double s1 = 0.0, s2 = 0.0;
  const int per = B * HW4;
  for (int e0 = l; e0 < per; e0 += 4 * TPC) {
    float4 v[4], d[4];
    float rs[4], nmr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * TPC;
      if (e < per) {
        const int b = e / HW4, i = e - b * HW4;
        const size_t o4 = ((size_t)b * C + c) * HW4 + i;
        v[u] = reinterpret_cast<const float4*>(x)[o4];
        d[u] = reinterpret_cast<const float4*>(dy)[o4];
        if (res) { const float4 q = reinterpret_cast<const float4*>(res)[o4]; v[u].x += q.x; v[u].y += q.y; v[u].z += q.z; v[u].w += q.w; }
        rs[u] = rstd[b * G + gi];
        nmr[u] = -mean[b * G + gi] * rs[u];
      }
    }
    float f1 = 0.f, f2 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * TPC;
      if (e < per) {
        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, dd[4] = {d[u].x, d[u].y, d[u].z, d[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = fmaf(vv[k], rs[u], nmr[u]);
          const float dz = dd[k] * act_grad(fmaf(xh, ga, be), act);
          f1 += dz;
          f2 = fmaf(dz, xh, f2);
        }
      }
    }
    s1 += (double)f1;
    s2 += (double)f2;
  }
  // sum over the TPC lanes of the row: inside the wave, then across the row's waves through LDS
  for (int d = 32; d >= 1; d >>= 1) { s1 += __shfl_xor(s1, d); s2 += __shfl_xor(s2, d); }
  if (TPC > 64) {
    const int w = tid >> 6, wpr = TPC >> 6;
    if ((tid & 63) == 0) { red[0][w] = s1; red[1][w] = s2; }
    __syncthreads();
    const int w0 = (w / wpr) * wpr;
    double a = 0.0, b2 = 0.0;
    for (int k = 0; k < wpr; ++k) { a += red[0][w0 + k]; b2 += red[1][w0 + k]; }
    s1 = a; s2 = b2;
  }
  if (valid && l == 0) { dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
}

static int g_gn_fused = -1;      // -1: read PNSFM_GN_FUSED on first use (default on)
static bool gn_fused_on() {
  if (g_gn_fused < 0) { const char* e = getenv("PNSFM_GN_FUSED"); g_gn_fused = (e && e[0] == '0') ? 0 : 1; }
  return g_gn_fused == 1;
}
// slabs a workgroup holds: float4 rows, <= 16 float4 per thread of a 1024-thread workgroup (64 K floats = 256 KB)
static bool gn_fused_ok(int C, int HW, int G) {
  if (!gn_fused_on() || HW % 4 != 0 || C / G > GNF_MAXCPG) return false;
  const long n4 = (long)(C / G) * (HW / 4);
  return n4 <= 16L * 1024;
}
static int gn_fused_threads(long n4) { return n4 > 2048 ? 1024 : 256; }
// workgroups per slab: the normalisation (exponentials) is most of a slab's arithmetic; 4 on the slabs of >= 4 float4 per thread of a
// 1024-thread workgroup
static int gn_fused_parts(long n4, int T) {
  const long nv = (n4 + T - 1) / T;
  return (T == 1024 && nv >= 4) ? 4 : 1;
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
  if (gn_fused_ok(C, HW, G)) {
    const long n4 = (long)cpg * (HW / 4);
    const int T = gn_fused_threads(n4);
    const int nv = (int)ceil_div_sz((size_t)n4, (size_t)T);
    const int S = gn_fused_parts(n4, T);
    const int HW4 = HW / 4, dq = T / HW4, dr = T % HW4, nslab = B * G;
    const dim3 grid(nslab * S), block(T);
#define PNSFM_GNF(NVv, Sv) PNSFM_LAUNCH((gn_fused_fwd_kernel<NVv, Sv>), grid, block, 0, s, x, res, gamma, beta, mean, rstd, y, C, HW, G, act, eps, nslab, dq, dr)
    if (nv <= 1) PNSFM_GNF(1, 1);
    else if (nv <= 2) PNSFM_GNF(2, 1);
    else if (nv <= 4) { if (S == 4) PNSFM_GNF(4, 4); else if (S == 2) PNSFM_GNF(4, 2); else PNSFM_GNF(4, 1); }
    else if (nv <= 8) { if (S == 4) PNSFM_GNF(8, 4); else if (S == 2) PNSFM_GNF(8, 2); else PNSFM_GNF(8, 1); }
    else { if (S == 4) PNSFM_GNF(16, 4); else if (S == 2) PNSFM_GNF(16, 2); else PNSFM_GNF(16, 1); }
#undef PNSFM_GNF
    return check_launch("gn_fused_fwd");
  }
  const GnGeom g = gn_geom(BC, HW, vec, PNSFM_GN_MAX_SPLIT);
  dim3 grid(ceil_div(BC, g.rows), g.nchunk);
  // stats_ws == null: the partial sums live in the stream's scratch buffer (they are dead when this call's second launch has run)
  ScratchLease lease(s, stats_ws ? 0 : pnsfm_groupnorm_ws_doubles(B, C, G) * sizeof(double));
  if (!stats_ws) { stats_ws = lease.as<double>(); if (!stats_ws) return -1; }
  if (vec) PNSFM_LAUNCH((gn_stats_kernel<true>), grid, dim3(256), 0, s, x, res, stats_ws, BC, C, HW, G, g);
  else PNSFM_LAUNCH((gn_stats_kernel<false>), grid, dim3(256), 0, s, x, res, stats_ws, BC, C, HW, G, g);
  int e = check_launch("gn_stats");
  if (e) return e;
  const double n = (double)cpg * (double)HW;
  if (vec) PNSFM_LAUNCH((gn_apply_kernel<true>), grid, dim3(256), 0, s, x, res, gamma, beta, (const double*)stats_ws, mean, rstd, y, BC, C, HW, G, act, n, eps, g, cpg * g.nchunk);
  else PNSFM_LAUNCH((gn_apply_kernel<false>), grid, dim3(256), 0, s, x, res, gamma, beta, (const double*)stats_ws, mean, rstd, y, BC, C, HW, G, act, n, eps, g, cpg * g.nchunk);
  return check_launch("gn_apply");
}

int pnsfm_groupnorm_act_backward(const float* dy, const float* x, const float* res, const float* gamma,
                                 const float* beta, const float* mean, const float* rstd, float* dx, float* dgamma,
                                 float* dbeta, double* red_ws, int B, int C, int HW, int G, int act, void* stream) {
  if (C % G != 0 || B <= 0 || HW <= 0) { set_error("groupnorm_backward: bad shape C=%d G=%d", C, G); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (HW % 4 == 0);
  const int BC = B * C;
  if (gn_fused_ok(C, HW, G)) {
    const long n4 = (long)(C / G) * (HW / 4);
    const int T = gn_fused_threads(n4);
    // threads per channel of the parameter blocks: <= 8 float4 per lane where the block allows it
    int TPC = 64;
    while (TPC < T && (long)B * (HW / 4) > 8L * TPC) TPC <<= 1;
    const int nparam = ceil_div(C, T / TPC);
    const int S = gn_fused_parts(n4, T);
    const int HW4 = HW / 4;
    PNSFM_LAUNCH(gn_fused_bwd_kernel, dim3(B * G * S + nparam), dim3(T), 0, s, dy, x, res, gamma, beta, mean, rstd, dx, dgamma, dbeta, B, C, HW, G,
                 act, TPC, S, T / HW4, T % HW4);
    return check_launch("gn_fused_bwd");
  }
  const GnGeom g = gn_geom(BC, HW, vec, PNSFM_GN_MAX_SPLIT);
  dim3 grid(ceil_div(BC, g.rows), g.nchunk);
  ScratchLease lease(s, red_ws ? 0 : pnsfm_groupnorm_ws_doubles(B, C, G) * sizeof(double));
  if (!red_ws) { red_ws = lease.as<double>(); if (!red_ws) return -1; }
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

/* 1 (default): slabs a workgroup can hold run the one-launch kernels; 0: the two-launch form everywhere.  Returns the previous setting. */
int pnsfm_set_gn_fused(int on) {
  const int prev = gn_fused_on() ? 1 : 0;
  g_gn_fused = on ? 1 : 0;
  return prev;
}

}  // extern "C"
