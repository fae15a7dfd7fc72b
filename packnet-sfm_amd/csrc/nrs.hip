// nrs.hip -- Neural-Ray-Surface projection: softmax-weighted expectation of pixel coordinates over a 41x41 candidate patch.
//
// Replaces the core of GenericCamera.project (/root/reference/packnet_sfm/geometry/camera_generic.py:86-208): for every
// (half-resolution) pixel p the reference gathers the ray-surface vectors of a 41x41 window of candidate pixels (window
// translated to stay inside the image, :127-145), forms the logits d_p . r_q (:173-183), applies a temperature softmax
// (:185-188) and takes the expectation of the candidate coordinates (:190-192).  It materialises the [3, HW, 1681] patch
// tensor (743 MB at 192x192), the logits and the softmax; here nothing but the [H, W] results touches HBM:
//   forward : one wave per pixel walks its 1681 candidates (27 per lane), the ray-surface window of an 8x8 pixel tile is staged
//             in LDS once (48x48x3 floats), max / sum / weighted sums by wave shuffles; saves (max, sum) per pixel;
//   backward: d(direction) by the same walk (gather); d(ray surface) as a GATHER too -- for a candidate pixel q the pixels p
//             whose window contains q form a contiguous rectangle (rows [q-20, q+20], extended to the border when the window
//             of the border rows was translated), so every output element has exactly one writer: no atomics, deterministic.
// Bound: L2/LDS-resident gather + VALU (exp), ~20 flop per (pixel, candidate); the arrays are a few MB.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

#define NRS_SIDE 20
#define NRS_K 41

__device__ __forceinline__ int nrs_clampc(int v, int n) {      // centre of the (translated) window of pixel coordinate v
  return v < NRS_SIDE ? NRS_SIDE : (v > n - 1 - NRS_SIDE ? n - 1 - NRS_SIDE : v);
}

__device__ __forceinline__ float wave_max(float v) {
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// dir, ray: [3][h][w]; coords: [h][w][2] = (E[row], E[col]); stat: [h][w][2] = (max logit/T, sum exp)
// MODE 0: forward.  MODE 1: backward w.r.t. dir (gcoords [h][w][2] in; gdir [3][h][w] out).
template <int MODE>
__global__ void __launch_bounds__(256) nrs_walk_kernel(const float* __restrict__ dir, const float* __restrict__ ray,
                                                       float* __restrict__ coords, float* __restrict__ stat,
                                                       const float* __restrict__ gcoords, float* __restrict__ gdir, int h, int w,
                                                       float inv_temp) {
  constexpr int TS = 8, TW = TS + 2 * NRS_SIDE;             // 8x8 pixel tile, 48x48 candidate window
  __shared__ float rs[3][TW * TW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.y * TS, j0 = blockIdx.x * TS;
  const int hw = h * w;
  // window of the tile = union of its pixels' windows
  const int r_lo = nrs_clampc(i0, h) - NRS_SIDE, c_lo = nrs_clampc(j0, w) - NRS_SIDE;
  const int i1 = i0 + TS - 1 < h ? i0 + TS - 1 : h - 1, j1 = j0 + TS - 1 < w ? j0 + TS - 1 : w - 1;
  const int r_hi = nrs_clampc(i1, h) + NRS_SIDE, c_hi = nrs_clampc(j1, w) + NRS_SIDE;
  const int nr = r_hi - r_lo + 1, nc = c_hi - c_lo + 1;
  for (int e = tid; e < nr * nc; e += 256) {
    const int r = e / nc, c = e - r * nc;
    const int g = (r_lo + r) * w + c_lo + c;
    rs[0][r * TW + c] = ray[g];
    rs[1][r * TW + c] = ray[hw + g];
    rs[2][r * TW + c] = ray[2 * hw + g];
  }
  __syncthreads();
  for (int pp = wave; pp < TS * TS; pp += 4) {
    const int i = i0 + pp / TS, j = j0 + pp % TS;
    if (i >= h || j >= w) continue;                          // wave-uniform
    const int p = i * w + j;
    const float d0 = dir[p], d1 = dir[hw + p], d2 = dir[2 * hw + p];
    const int wi = nrs_clampc(i, h) - NRS_SIDE, wj = nrs_clampc(j, w) - NRS_SIDE;     // window origin
    const int base = (wi - r_lo) * TW + (wj - c_lo);
    float z[27];
    float m = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int k = lane + 64 * t;
      if (k < NRS_K * NRS_K) {
        const int di = k / NRS_K, dj = k - di * NRS_K;
        const int o = base + di * TW + dj;
        z[t] = (d0 * rs[0][o] + d1 * rs[1][o] + d2 * rs[2][o]) * inv_temp;
        m = fmaxf(m, z[t]);
      } else {
        z[t] = -3.0e38f;
      }
    }
    if (MODE == 0) {
      m = wave_max(m);
      float s = 0.f, sr = 0.f, sc = 0.f;
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const int k = lane + 64 * t;
        if (k < NRS_K * NRS_K) {
          const int di = k / NRS_K, dj = k - di * NRS_K;
          const float e = expf(z[t] - m);
          s += e;
          sr += e * (float)(wi + di);
          sc += e * (float)(wj + dj);
        }
      }
      s = wave_sum(s); sr = wave_sum(sr); sc = wave_sum(sc);
      if (lane == 0) {
        coords[2 * p] = sr / s;
        coords[2 * p + 1] = sc / s;
        stat[2 * p] = m;
        stat[2 * p + 1] = s;
      }
    } else {
      const float mm = stat[2 * p], inv_s = 1.f / stat[2 * p + 1];
      const float er = coords[2 * p], ec = coords[2 * p + 1];
      const float gr = gcoords[2 * p], gc = gcoords[2 * p + 1];
      float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int t = 0; t < 27; ++t) {
        const int k = lane + 64 * t;
        if (k < NRS_K * NRS_K) {
          const int di = k / NRS_K, dj = k - di * NRS_K;
          const int o = base + di * TW + dj;
          const float wgt = expf(z[t] - mm) * inv_s;
          const float dl = wgt * (gr * ((float)(wi + di) - er) + gc * ((float)(wj + dj) - ec)) * inv_temp;   // dL/dlogit
          a0 += dl * rs[0][o];
          a1 += dl * rs[1][o];
          a2 += dl * rs[2][o];
        }
      }
      a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
      if (lane == 0) { gdir[p] = a0; gdir[hw + p] = a1; gdir[2 * hw + p] = a2; }
    }
  }
}

// backward w.r.t. the ray surface: one wave per candidate pixel q gathers over the pixels p whose window contains q
__global__ void __launch_bounds__(256) nrs_gray_kernel(const float* __restrict__ dir, const float* __restrict__ ray,
                                                       const float* __restrict__ coords, const float* __restrict__ stat,
                                                       const float* __restrict__ gcoords, float* __restrict__ gray, int h, int w,
                                                       float inv_temp) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wave;
  const int hw = h * w;
  if (q >= hw) return;
  const int qi = q / w, qj = q - qi * w;
  const float r0 = ray[q], r1 = ray[hw + q], r2 = ray[2 * hw + q];
  // rows p_i with |centre(p_i) - q_i| <= 20: contiguous, reaching the border when the border rows' windows were translated
  const int ilo = qi <= 2 * NRS_SIDE ? 0 : qi - NRS_SIDE, ihi = qi >= h - 1 - 2 * NRS_SIDE ? h - 1 : qi + NRS_SIDE;
  const int jlo = qj <= 2 * NRS_SIDE ? 0 : qj - NRS_SIDE, jhi = qj >= w - 1 - 2 * NRS_SIDE ? w - 1 : qj + NRS_SIDE;
  const int ni = ihi - ilo + 1, nj = jhi - jlo + 1;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int e = lane; e < ni * nj; e += 64) {
    const int pi = ilo + e / nj, pj = jlo + e % nj;
    if (abs(nrs_clampc(pi, h) - qi) > NRS_SIDE || abs(nrs_clampc(pj, w) - qj) > NRS_SIDE) continue;   // (always false; kept as the definition)
    const int p = pi * w + pj;
    const float d0 = dir[p], d1 = dir[hw + p], d2 = dir[2 * hw + p];
    const float z = (d0 * r0 + d1 * r1 + d2 * r2) * inv_temp;
    const float wgt = expf(z - stat[2 * p]) / stat[2 * p + 1];
    const float dl = wgt * (gcoords[2 * p] * ((float)qi - coords[2 * p]) + gcoords[2 * p + 1] * ((float)qj - coords[2 * p + 1])) * inv_temp;
    a0 += dl * d0;
    a1 += dl * d1;
    a2 += dl * d2;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
  if (lane == 0) { gray[q] = a0; gray[hw + q] = a1; gray[2 * hw + q] = a2; }
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_nrs_project_forward(const float* dir, const float* ray, float* coords, float* stat, int h, int w, float temperature,
                              void* stream) {
  if (h < NRS_K || w < NRS_K) { set_error("nrs_project: the map must be at least 41x41 (got %dx%d)", h, w); return -1; }
  if (!(temperature > 0.f)) { set_error("nrs_project: temperature must be > 0"); return -1; }
  dim3 grid(ceil_div(w, 8), ceil_div(h, 8));
  PNSFM_LAUNCH((nrs_walk_kernel<0>), grid, dim3(256), 0, (hipStream_t)stream, dir, ray, coords, stat, (const float*)nullptr,
               (float*)nullptr, h, w, 1.0f / temperature);
  return check_launch("nrs_project_forward");
}

int pnsfm_nrs_project_backward(const float* dir, const float* ray, const float* coords, const float* stat, const float* gcoords,
                               float* gdir, float* gray, int h, int w, float temperature, void* stream) {
  if (h < NRS_K || w < NRS_K) { set_error("nrs_project: the map must be at least 41x41 (got %dx%d)", h, w); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const float inv_temp = 1.0f / temperature;
  if (gdir) {
    dim3 grid(ceil_div(w, 8), ceil_div(h, 8));
    PNSFM_LAUNCH((nrs_walk_kernel<1>), grid, dim3(256), 0, s, dir, ray, const_cast<float*>(coords), const_cast<float*>(stat), gcoords,
                 gdir, h, w, inv_temp);
    int e = check_launch("nrs_project_backward (direction)");
    if (e) return e;
  }
  if (gray) {
    PNSFM_LAUNCH(nrs_gray_kernel, dim3(ceil_div(h * w, 4)), dim3(256), 0, s, dir, ray, coords, stat, gcoords, gray, h, w, inv_temp);
    return check_launch("nrs_project_backward (ray surface)");
  }
  return 0;
}

}  // extern "C"
