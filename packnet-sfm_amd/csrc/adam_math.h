// adam_math.h -- the per-element Adam update shared by adam_flat_kernel (elementwise.hip), adam_segments_kernel and the fused
// Adam + weight re-pack kernel (conv2d_bx3.h): ONE definition, so the three produce bit-identical parameters and moments.
// torch.optim.Adam(amsgrad=False) as configured by the reference (packnet_sfm/models/model_wrapper.py:128-166).
//   hp[0] = step (as float, exact up to 2^24), hp[1] = lr, hp[2] = beta1, hp[3] = beta2, hp[4] = eps, hp[5] = weight decay,
//   hp[6] = gradient scale, hp[7] = 1 - beta1, hp[8] = 1 - beta2 (rounded from the host's double arithmetic like torch does)
#pragma once

struct AdamCoef {
  float beta1, beta2, eps, wd, gscale, omb1, omb2, rsqrt_bc2, step_size;
};

__device__ __forceinline__ AdamCoef adam_coef(const float* __restrict__ hp) {
  AdamCoef c;
  const float step = hp[0], lr = hp[1];
  c.beta1 = hp[2]; c.beta2 = hp[3]; c.eps = hp[4]; c.wd = hp[5]; c.gscale = hp[6]; c.omb1 = hp[7]; c.omb2 = hp[8];
  const float bc1 = (float)(1.0 - pow((double)c.beta1, (double)step));          // once per thread, in double like the host formula
  c.rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)c.beta2, (double)step)));
  c.step_size = lr / bc1;
  return c;
}

__device__ __forceinline__ void adam_update(const AdamCoef& c, float g, float& p, float& m, float& v) {
  float gi = g * c.gscale;
  if (c.wd != 0.f) gi = fmaf(c.wd, p, gi);
  m = c.beta1 * m + c.omb1 * gi;
  v = c.beta2 * v + c.omb2 * gi * gi;
  p -= c.step_size * (m / (sqrtf(v) * c.rsqrt_bc2 + c.eps));
}
