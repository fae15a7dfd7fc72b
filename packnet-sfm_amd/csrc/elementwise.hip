// elementwise.hip -- InvDepth activation and the flat Adam update (pure HBM streaming kernels).
//
//  invdepth_act : y = sigmoid(x) / min_depth      /root/reference/packnet_sfm/networks/layers/packnet/layers01.py:119-122
//  adam_step    : torch.optim.Adam (amsgrad=False) on one flat parameter group, as configured by
//                 /root/reference/packnet_sfm/models/model_wrapper.py:128-149 and stepped at
//                 /root/reference/packnet_sfm/trainers/horovod_trainer.py:93 (28 B/parameter of HBM traffic).
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

__global__ void __launch_bounds__(256) invdepth_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                            float inv_min_depth) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = inv_min_depth / (1.f + expf(-x[i]));
}

// y = s*sig  =>  dy/dx = s*sig*(1-sig) = y * (1 - y/s)
__global__ void __launch_bounds__(256) invdepth_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            float* __restrict__ dx, size_t n, float min_depth) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float yy = y[i];
    dx[i] = dy[i] * yy * (1.f - yy * min_depth);
  }
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float beta1, float beta2,
                                                    float eps, float wd, float gscale, float bc1, float rsqrt_bc2) {
  const float step_size = lr / bc1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

static int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_invdepth_act_forward(const float* x, float* y, size_t n, float min_depth, void* stream) {
  if (min_depth <= 0.f) { set_error("invdepth_act: min_depth must be > 0"); return -1; }
  PNSFM_LAUNCH(invdepth_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, 1.0f / min_depth);
  return check_launch("invdepth_act_forward");
}

int pnsfm_invdepth_act_backward(const float* dy, const float* y, float* dx, size_t n, float min_depth, void* stream) {
  PNSFM_LAUNCH(invdepth_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n, min_depth);
  return check_launch("invdepth_act_backward");
}

int pnsfm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, float grad_scale, int step, void* stream) {
  if (step < 1) { set_error("adam_step: step must be >= 1"); return -1; }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  PNSFM_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
               beta1, beta2, eps, weight_decay, grad_scale, (float)bc1, (float)(1.0 / sqrt(bc2)));
  return check_launch("adam_step");
}

}  // extern "C"
