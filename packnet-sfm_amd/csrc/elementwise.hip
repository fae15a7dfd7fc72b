// elementwise.hip -- InvDepth activation and the flat Adam update (pure HBM streaming kernels).
//
//  invdepth_act : y = sigmoid(x) / min_depth      /root/reference/packnet_sfm/networks/layers/packnet/layers01.py:119-122
//  pose_vec2mat : [tx,ty,tz,rx,ry,rz] -> 4x4 rigid transform, R = Rx*Ry*Rz (euler)
//                 /root/reference/packnet_sfm/geometry/pose_utils.py:8-52 (euler2mat, pose_vec2mat) and
//                 /root/reference/packnet_sfm/geometry/pose.py:40-46 (Pose.from_vec); ~85 tiny ATen launches
//                 (slices, sin/cos, stacks, bmm, cat + their backward) become one launch each way.
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

// R = Rx(rx) * Ry(ry) * Rz(rz):
//   row0 = [ cy cz,            -cy sz,             sy    ]
//   row1 = [ cx sz + sx sy cz,  cx cz - sx sy sz, -sx cy ]
//   row2 = [ sx sz - cx sy cz,  sx cz + cx sy sz,  cx cy ]
__device__ __forceinline__ void euler_rows(float rx, float ry, float rz, float* R) {
  const float sx = sinf(rx), cx = cosf(rx), sy = sinf(ry), cy = cosf(ry), sz = sinf(rz), cz = cosf(rz);
  R[0] = cy * cz;                 R[1] = -cy * sz;                R[2] = sy;
  R[3] = cx * sz + sx * sy * cz;  R[4] = cx * cz - sx * sy * sz;  R[5] = -sx * cy;
  R[6] = sx * sz - cx * sy * cz;  R[7] = sx * cz + cx * sy * sz;  R[8] = cx * cy;
}

__global__ void __launch_bounds__(64) pose_vec2mat_fwd_kernel(const float* __restrict__ vec, float* __restrict__ mat, int N) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const float* v = vec + n * 6;
  float R[9];
  euler_rows(v[3], v[4], v[5], R);
  float* m = mat + n * 16;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    m[i * 4 + 0] = R[i * 3 + 0];
    m[i * 4 + 1] = R[i * 3 + 1];
    m[i * 4 + 2] = R[i * 3 + 2];
    m[i * 4 + 3] = v[i];
  }
  m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
}

// dR/drx: row1 -> -row2, row2 -> row1 (row0 -> 0);  dR/drz: col0 -> col1, col1 -> -col0 (col2 -> 0);
// dR/dry = Rx * N with N = dRy/dry * Rz = [[-sy cz, sy sz, cy], [0,0,0], [-cy cz, cy sz, -sy]]: rows N0, -sx N2, cx N2.
__global__ void __launch_bounds__(64) pose_vec2mat_bwd_kernel(const float* __restrict__ vec, const float* __restrict__ dmat,
                                                               float* __restrict__ dvec, int N) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const float* v = vec + n * 6;
  const float* g = dmat + n * 16;
  float R[9];
  euler_rows(v[3], v[4], v[5], R);
  const float sx = sinf(v[3]), cx = cosf(v[3]), sy = sinf(v[4]), cy = cosf(v[4]), sz = sinf(v[5]), cz = cosf(v[5]);
  float G[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) G[i * 3 + j] = g[i * 4 + j];
  float drx = 0.f, dry = 0.f, drz = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) drx += -G[3 + j] * R[6 + j] + G[6 + j] * R[3 + j];
  const float N0[3] = {-sy * cz, sy * sz, cy}, N2[3] = {-cy * cz, cy * sz, -sy};
#pragma unroll
  for (int j = 0; j < 3; ++j) dry += G[j] * N0[j] + G[3 + j] * (-sx * N2[j]) + G[6 + j] * (cx * N2[j]);
#pragma unroll
  for (int i = 0; i < 3; ++i) drz += G[i * 3 + 0] * R[i * 3 + 1] - G[i * 3 + 1] * R[i * 3 + 0];
  float* d = dvec + n * 6;
  d[0] = g[3]; d[1] = g[7]; d[2] = g[11];
  d[3] = drx; d[4] = dry; d[5] = drz;
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

// Device-resident optimizer state (hipGraph-replayable: nothing the kernel needs is a launch argument that changes):
//   hp[0] = step (as float, exact up to 2^24), hp[1] = lr, hp[2] = beta1, hp[3] = beta2, hp[4] = eps, hp[5] = weight decay,
//   hp[6] = gradient scale (e.g. 1/world when gradients were sum-reduced), hp[7] = 1 - beta1, hp[8] = 1 - beta2 (rounded from
//   the host's double arithmetic like torch.optim.Adam does: 1.f - 0.999f is 4.7e-5 off)
__global__ void adam_tick_kernel(float* __restrict__ hp) { hp[0] += 1.f; }

// float4 streaming Adam: 28 B/parameter of HBM traffic, bias corrections computed per thread from the device-side step
__global__ void __launch_bounds__(256) adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, size_t n, const float* __restrict__ hp) {
  const float step = hp[0], lr = hp[1], beta1 = hp[2], beta2 = hp[3], eps = hp[4], wd = hp[5], gscale = hp[6];
  const float omb1 = hp[7], omb2 = hp[8];
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));          // once per thread, in double like the host formula
  const float rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
  const float step_size = lr / bc1;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float ga[4] = {gg.x, gg.y, gg.z, gg.w}, pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gi = ga[k] * gscale;
      if (wd != 0.f) gi = fmaf(wd, pa[k], gi);
      ma[k] = beta1 * ma[k] + omb1 * gi;
      va[k] = beta2 * va[k] + omb2 * gi * gi;
      pa[k] -= step_size * (ma[k] / (sqrtf(va[k]) * rsqrt_bc2 + eps));
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
  }
  // tail (n % 4 elements)
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = beta1 * m[i] + omb1 * gi;
    const float vi = beta2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * (mi / (sqrtf(vi) * rsqrt_bc2 + eps));
  }
}

static int grid_for(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- batched 4-D region operations: ONE launch for a list of (copy | add | zero) operations on strided 4-D windows of fp32 tensors.
// The collapsed packing block (layers01.py: PackLayerConv3d) gathers border strips, selects rows of them, pastes results into the
// interior result and scatters / accumulates the matching gradients: ~33 slice-assignments, cats, fills and strided adds per block and
// step on ATen (profiles/r04: ~0.5 ms per step in launches of 4-30 us).  Each autograd Function now issues one launch of this kernel.
// dst / src addresses are base + sum_i idx_i * stride_i (element strides, any layout); blockIdx.y = operation, grid-stride over elements.
struct RegionOp {
  const float* src;      // null for zero
  float* dst;
  unsigned n[4];         // extents (n[3] in float4 units when vec)
  long long ss[4], ds[4];
  int op;                // 0 copy, 1 add (dst += src), 2 zero
  int vec;               // innermost dimension contiguous, 16-byte aligned and a multiple of 4 on both sides: float4 accesses
};
struct RegionOps { RegionOp o[PNSFM_MAX_REGION_OPS]; };

__global__ void __launch_bounds__(256) region_ops_kernel(RegionOps ops) {
  const RegionOp& r = ops.o[blockIdx.y];
  const unsigned total = r.n[0] * r.n[1] * r.n[2] * r.n[3];        // (the entry point refuses windows of >= 2^31 elements)
  for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
    unsigned t = e;
    const unsigned i3 = t % r.n[3]; t /= r.n[3];
    const unsigned i2 = t % r.n[2]; t /= r.n[2];
    const unsigned i1 = t % r.n[1];
    const unsigned i0 = t / r.n[1];
    if (r.vec) {
      float4* d = reinterpret_cast<float4*>(r.dst + i0 * r.ds[0] + i1 * r.ds[1] + i2 * r.ds[2]) + i3;
      if (r.op == 2) { *d = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
      const float4 v = reinterpret_cast<const float4*>(r.src + i0 * r.ss[0] + i1 * r.ss[1] + i2 * r.ss[2])[i3];
      if (r.op == 1) { float4 o = *d; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; *d = o; }
      else *d = v;
    } else {
      float* d = r.dst + i0 * r.ds[0] + i1 * r.ds[1] + i2 * r.ds[2] + i3 * r.ds[3];
      if (r.op == 2) { *d = 0.f; continue; }
      const float v = r.src[i0 * r.ss[0] + i1 * r.ss[1] + i2 * r.ss[2] + i3 * r.ss[3]];
      *d = r.op == 1 ? *d + v : v;
    }
  }
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

int pnsfm_pose_vec2mat_forward(const float* vec, float* mat, int N, void* stream) {
  if (N <= 0) { set_error("pose_vec2mat: bad N"); return -1; }
  PNSFM_LAUNCH(pose_vec2mat_fwd_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, vec, mat, N);
  return check_launch("pose_vec2mat_forward");
}

int pnsfm_pose_vec2mat_backward(const float* vec, const float* dmat, float* dvec, int N, void* stream) {
  if (N <= 0) { set_error("pose_vec2mat: bad N"); return -1; }
  PNSFM_LAUNCH(pose_vec2mat_bwd_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, vec, dmat, dvec, N);
  return check_launch("pose_vec2mat_backward");
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


int pnsfm_adam_flat_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float* hp, void* stream) {
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) {
    set_error("adam_flat_step: buffers must be 16-byte aligned");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  PNSFM_LAUNCH(adam_tick_kernel, dim3(1), dim3(1), 0, s, hp);
  PNSFM_LAUNCH(adam_flat_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, (const float*)hp);
  return check_launch("adam_flat_step");
}

int pnsfm_region_ops(const void* ops_host, int n_ops, void* stream) {
  if (n_ops < 1 || n_ops > PNSFM_MAX_REGION_OPS) { set_error("region_ops: 1..%d operations per launch (got %d)", PNSFM_MAX_REGION_OPS, n_ops); return -1; }
  RegionOps ops;
  long long most = 0;
  const pnsfm_region_op* in = static_cast<const pnsfm_region_op*>(ops_host);
  for (int i = 0; i < n_ops; ++i) {
    RegionOp& r = ops.o[i];
    r.src = in[i].src; r.dst = in[i].dst; r.op = in[i].op;
    long long total = 1;
    for (int k = 0; k < 4; ++k) {
      if (in[i].n[k] < 1) { set_error("region_ops: empty extent in operation %d", i); return -1; }
      r.n[k] = (unsigned)in[i].n[k]; r.ss[k] = in[i].src_stride[k]; r.ds[k] = in[i].dst_stride[k];
      total *= in[i].n[k];
    }
    if (total >= (1LL << 31)) { set_error("region_ops: window of operation %d has >= 2^31 elements", i); return -1; }
    if (r.op < 0 || r.op > 2 || !r.dst || (r.op != 2 && !r.src)) { set_error("region_ops: bad operation %d", i); return -1; }
    // float4 path: unit innermost strides, extent and every outer stride a multiple of 4, 16-byte aligned bases
    bool vec = (r.n[3] % 4 == 0) && r.ds[3] == 1 && (((uintptr_t)r.dst) & 15) == 0;
    for (int k = 0; k < 3 && vec; ++k) vec = r.ds[k] % 4 == 0;
    if (r.op != 2) {
      vec = vec && r.ss[3] == 1 && (((uintptr_t)r.src) & 15) == 0;
      for (int k = 0; k < 3 && vec; ++k) vec = r.ss[k] % 4 == 0;
    }
    r.vec = vec ? 1 : 0;
    if (vec) { r.n[3] /= 4; total /= 4; }
    if (total > most) most = total;
  }
  long long gx = (most + 255) / 256;
  if (gx > 2048) gx = 2048;
  PNSFM_LAUNCH(region_ops_kernel, dim3((unsigned)gx, (unsigned)n_ops), dim3(256), 0, (hipStream_t)stream, ops);
  return check_launch("region_ops");
}

}  // extern "C"
