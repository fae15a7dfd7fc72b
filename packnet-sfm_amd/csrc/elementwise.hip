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
#include <cstring>
#include "../../include/pnsfm.h"
#include "adam_math.h"

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

// float4 streaming Adam: 28 B/parameter of HBM traffic, bias corrections computed per thread from the device-side step (adam_math.h)
__global__ void __launch_bounds__(256) adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, size_t n, const float* __restrict__ hp) {
  const AdamCoef c = adam_coef(hp);
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float ga[4] = {gg.x, gg.y, gg.z, gg.w}, pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_update(c, ga[k], pa[k], ma[k], va[k]);
    reinterpret_cast<float4*>(p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(va[0], va[1], va[2], va[3]);
  }
  // tail (n % 4 elements)
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_update(c, g[i], pi, mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// The same update over a LIST of segments of the arenas in one launch (round 5: everything that is not a split-bf16 conv weight --
// those are updated by adam_pack_table_kernel, conv2d_bx3.h, which writes their packed copies from the registers the update left them
// in).  One workgroup per <= 1024 elements of a segment (binary search over the segments' first blocks, like the pack table).
struct AdamSeg {
  float* p;
  const float* g;
  float* m;
  float* v;
  const float* hp;
  unsigned n;
  int blk0;
};
__global__ void __launch_bounds__(256) adam_segments_kernel(const AdamSeg* __restrict__ segs, int nseg) {
  int lo = 0, hi = nseg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const AdamSeg sg = segs[lo];
  const unsigned e0 = ((unsigned)blockIdx.x - (unsigned)sg.blk0) * 1024u + threadIdx.x * 4u;
  if (e0 >= sg.n) return;
  const AdamCoef c = adam_coef(sg.hp);
  if (e0 + 4u <= sg.n) {           // (segments start on 16-byte boundaries of the arenas: FlatAdam aligns every parameter)
    const float4 gg = *reinterpret_cast<const float4*>(sg.g + e0);
    float4 pp = *reinterpret_cast<float4*>(sg.p + e0), mm = *reinterpret_cast<float4*>(sg.m + e0), vv = *reinterpret_cast<float4*>(sg.v + e0);
    float ga[4] = {gg.x, gg.y, gg.z, gg.w}, pa[4] = {pp.x, pp.y, pp.z, pp.w}, ma[4] = {mm.x, mm.y, mm.z, mm.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_update(c, ga[k], pa[k], ma[k], va[k]);
    *reinterpret_cast<float4*>(sg.p + e0) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    *reinterpret_cast<float4*>(sg.m + e0) = make_float4(ma[0], ma[1], ma[2], ma[3]);
    *reinterpret_cast<float4*>(sg.v + e0) = make_float4(va[0], va[1], va[2], va[3]);
  } else {
    for (unsigned i = e0; i < sg.n; ++i) {
      float pi = sg.p[i], mi = sg.m[i], vi = sg.v[i];
      adam_update(c, sg.g[i], pi, mi, vi);
      sg.p[i] = pi; sg.m[i] = mi; sg.v[i] = vi;
    }
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

// ---- nearest-neighbour up-sampling by an integer factor s (the reference's F.interpolate(mode='nearest'): every predicted scale
// brought to full resolution, models/model_utils.py:163-180 via utils/image.py:148-176, and the inverse depth handed to the next
// iconv block, nn.Upsample in networks/depth/PackNet01.py:87-89,150,159,168): y[n, oy, ox] = x[n, oy / s, ox / s].  One thread per FOUR output columns
// (the host only takes maps whose output width is a multiple of 4).
__global__ void __launch_bounds__(256) upsample_nearest_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w,
                                                                   int s, unsigned total4) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total4) return;
  const unsigned W4 = (unsigned)(w * s) >> 2, Ho = (unsigned)(h * s);
  const unsigned q = i % W4, t = i / W4;
  const unsigned oy = t % Ho, n = t / Ho;
  const float* row = x + ((size_t)n * h + oy / s) * w;
  const unsigned ox = 4u * q;
  float4 v;
  v.x = row[ox / s]; v.y = row[(ox + 1) / s]; v.z = row[(ox + 2) / s]; v.w = row[(ox + 3) / s];
  reinterpret_cast<float4*>(y)[i] = v;
}

// its gradient: dx[n, iy, ix] = sum of the s x s block of dy, rows then columns in ascending order (one thread per input element)
__global__ void __launch_bounds__(256) upsample_nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int h, int w,
                                                                   int s, unsigned total) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned ix = i % (unsigned)w, t = i / (unsigned)w;
  const unsigned iy = t % (unsigned)h, n = t / (unsigned)h;
  const size_t Wo = (size_t)w * s;
  const float* p = dy + ((size_t)n * h * s + (size_t)iy * s) * Wo + (size_t)ix * s;
  float acc = 0.f;
  for (int a = 0; a < s; ++a, p += Wo) {
    if (s == 2) {
      const float2 v = *reinterpret_cast<const float2*>(p);
      acc += v.x; acc += v.y;
    } else if ((s & 3) == 0) {
      for (int b = 0; b < s; b += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + b);
        acc += v.x; acc += v.y; acc += v.z; acc += v.w;
      }
    } else {
      for (int b = 0; b < s; ++b) acc += p[b];
    }
  }
  dx[i] = acc;
}

// ---- the scalar tail of MultiViewPhotometricLoss.forward (reference losses/multiview_photometric_loss.py: reduce_photometric_loss
// :248-252, calc_smoothness_loss :275-280, the in-place sum :337-338): per-scale
// photometric means P[i] and smoothness terms S[i] -> out[0] = loss = mean_i P[i] + weight * mean_i (S[i] / 2^i), out[1] = the
// weighted smoothness term, out[2] = the photometric term, in the reference's operation order (one thread; 2n values).
struct LossTerms {
  const float* p[8];
  const float* s[8];
};

__global__ void loss_combine_fwd_kernel(LossTerms t, int n, int ns, float weight, float* __restrict__ out) {
  float photo = 0.f;
  for (int i = 0; i < n; ++i) photo = photo + t.p[i][0];
  photo = photo / (float)n;
  float smooth = 0.f;
  for (int i = 0; i < ns; ++i) smooth = smooth + t.s[i][0] / (float)(1 << i);
  if (ns > 0) smooth = weight * (smooth / (float)ns);
  out[0] = ns > 0 ? photo + smooth : photo;
  out[1] = smooth;
  out[2] = photo;
}

// d(loss)/dP[i] = g / n -> dout[i]; d(loss)/dS[i] = ((g * weight) / ns) / 2^i -> dout[8 + i]   (g: device scalar)
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, int n, int ns, float weight, float* __restrict__ dout) {
  const int i = threadIdx.x;
  if (i < n) dout[i] = g[0] / (float)n;
  if (i < ns) dout[8 + i] = ((g[0] * weight) / (float)ns) / (float)(1 << i);
}

// ---- bias of the composed packing convolution (networks/layers/packnet/layers01.py of this package: _conv_collapsed; the reference
// runs Conv3d then Conv2d, layers01.py:243-246): bias_eff[co] = b2[co] + sum_f b3[f] * Ssum[co][f], Ssum[co][f] = sum of the
// f-th block (blk contiguous floats) of row co of the Conv2d weight.  One workgroup per co; Ssum is kept for the gradient.
__global__ void __launch_bounds__(256) pack_bias_eff_kernel(const float* __restrict__ W2, const float* __restrict__ b2,
                                                            const float* __restrict__ b3, float* __restrict__ Ssum,
                                                            float* __restrict__ bias_eff, int d, int blk) {
  __shared__ float part[4];
  __shared__ float sums[8];
  const int co = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int f = 0; f < d; ++f) {
    const float* p = W2 + ((size_t)co * d + f) * blk;
    float acc = 0.f;
    for (int e = threadIdx.x; e < blk; e += 256) acc += p[e];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[f] = ((part[0] + part[1]) + part[2]) + part[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int f = 0; f < d; ++f) {
      Ssum[(size_t)co * d + f] = sums[f];
      acc += sums[f] * b3[f];
    }
    bias_eff[co] = (b2 ? b2[co] : 0.f) + acc;
  }
}

// gradients of that bias: db3[f] = sum_co g[co] * Ssum[co][f] (one wave per f, ascending co per lane, fixed-order lane tree)
__global__ void __launch_bounds__(64) pack_bias_eff_db3_kernel(const float* __restrict__ g, const float* __restrict__ Ssum,
                                                               float* __restrict__ db3, int C, int d) {
  const int f = blockIdx.x;
  float acc = 0.f;
  for (int co = threadIdx.x; co < C; co += 64) acc += g[co] * Ssum[(size_t)co * d + f];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (threadIdx.x == 0) db3[f] = acc;
}

// ... and the Conv2d weight's: dW2[co][f-block][ky][kx] = full[co][f-block][ky + 1][kx + 1] (the composition's gradient, computed on
// the zero-ring-padded (k+2)^2 taps) + g[co] * b3[f] (the bias path).  rows = C * d * D tap planes of k x k.
__global__ void __launch_bounds__(256) pack_dw2_finish_kernel(const float* __restrict__ full, const float* __restrict__ g,
                                                              const float* __restrict__ b3, float* __restrict__ dW2, int d, int D,
                                                              int k, unsigned total) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned kk = (unsigned)(k * k), plane = i / kk, tap = i - plane * kk;
  const unsigned ky = tap / (unsigned)k, kx = tap - ky * (unsigned)k;
  const unsigned co = plane / (unsigned)(d * D), f = (plane / (unsigned)D) % (unsigned)d;
  const unsigned k2 = (unsigned)(k + 2);
  dW2[i] = full[((size_t)plane * k2 + ky + 1) * k2 + kx + 1] + g[co] * b3[f];
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

// A slice of the arenas (round 5: FlatAdam updates bucket by bucket underneath the backward pass).  tick != 0 advances the group's
// step counter first -- the FIRST slice of a step does that; the others read the same hp[0].
int pnsfm_adam_flat_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float* hp, int tick,
                           void* stream) {
  if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0) {
    set_error("adam_flat_update: buffers must be 16-byte aligned");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  if (tick) PNSFM_LAUNCH(adam_tick_kernel, dim3(1), dim3(1), 0, s, hp);
  if (n) PNSFM_LAUNCH(adam_flat_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, (const float*)hp);
  return check_launch("adam_flat_update");
}

size_t pnsfm_adam_seg_bytes(void) { return sizeof(AdamSeg); }
int pnsfm_adam_seg_fill(void* seg_host, float* p, const float* g, float* m, float* v, const float* hp, size_t n, int first_block) {
  if (!seg_host || !p || !g || !m || !v || !hp || n == 0 || n > 0xffffffffu) { set_error("adam_seg_fill: bad segment"); return -1; }
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) { set_error("adam_seg_fill: segments must start on 16-byte boundaries"); return -1; }
  AdamSeg sg = {p, g, m, v, hp, (unsigned)n, first_block};
  memcpy(seg_host, &sg, sizeof(sg));
  return (int)((n + 1023) / 1024);
}
int pnsfm_adam_segments(const void* segs_dev, int nseg, int total_blocks, void* stream) {
  if (nseg <= 0 || total_blocks <= 0) return 0;
  if (!segs_dev) { set_error("adam_segments: null table"); return -1; }
  PNSFM_LAUNCH(adam_segments_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const AdamSeg*>(segs_dev), nseg);
  return check_launch("adam_segments");
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

int pnsfm_upsample_nearest_forward(const float* x, float* y, int N, int h, int w, int s, void* stream) {
  if (N < 1 || h < 1 || w < 1 || s < 1 || (w * s) % 4 != 0) { set_error("upsample_nearest_forward: output width must be a multiple of 4 (N=%d h=%d w=%d s=%d)", N, h, w, s); return -1; }
  const size_t total = (size_t)N * h * s * w * s;
  if (total >= (1ull << 32)) { set_error("upsample_nearest_forward: >= 2^32 output elements"); return -1; }
  const unsigned total4 = (unsigned)(total / 4);
  PNSFM_LAUNCH(upsample_nearest_fwd_kernel, dim3((total4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, h, w, s, total4);
  return check_launch("upsample_nearest_forward");
}

int pnsfm_upsample_nearest_backward(const float* dy, float* dx, int N, int h, int w, int s, void* stream) {
  if (N < 1 || h < 1 || w < 1 || s < 1 || (w * s) % 4 != 0) { set_error("upsample_nearest_backward: output width must be a multiple of 4 (N=%d h=%d w=%d s=%d)", N, h, w, s); return -1; }
  if ((size_t)N * h * s * w * s >= (1ull << 32)) { set_error("upsample_nearest_backward: >= 2^32 output elements"); return -1; }
  const unsigned total = (unsigned)((size_t)N * h * w);
  PNSFM_LAUNCH(upsample_nearest_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, dx, h, w, s, total);
  return check_launch("upsample_nearest_backward");
}

int pnsfm_loss_combine_forward(const float* const* photometric, int n, const float* const* smoothness, int ns, float weight,
                               float* out3, void* stream) {
  if (n < 1 || n > 8 || ns < 0 || ns > 8 || !photometric || (ns > 0 && !smoothness)) { set_error("loss_combine_forward: 1..8 photometric and 0..8 smoothness terms (got %d, %d)", n, ns); return -1; }
  LossTerms t = {};
  for (int i = 0; i < n; ++i) t.p[i] = photometric[i];
  for (int i = 0; i < ns; ++i) t.s[i] = smoothness[i];
  PNSFM_LAUNCH(loss_combine_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t, n, ns, weight, out3);
  return check_launch("loss_combine_forward");
}

int pnsfm_loss_combine_backward(const float* g, int n, int ns, float weight, float* dout16, void* stream) {
  if (n < 1 || n > 8 || ns < 0 || ns > 8) { set_error("loss_combine_backward: 1..8 photometric and 0..8 smoothness terms (got %d, %d)", n, ns); return -1; }
  PNSFM_LAUNCH(loss_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, n, ns, weight, dout16);
  return check_launch("loss_combine_backward");
}

int pnsfm_pack_bias_eff_forward(const float* W2, const float* b2, const float* b3, float* Ssum, float* bias_eff, int C, int d,
                                int blk, void* stream) {
  if (C < 1 || d < 1 || d > 8 || blk < 1) { set_error("pack_bias_eff_forward: bad sizes (C=%d d=%d blk=%d)", C, d, blk); return -1; }
  PNSFM_LAUNCH(pack_bias_eff_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, W2, b2, b3, Ssum, bias_eff, d, blk);
  return check_launch("pack_bias_eff_forward");
}

int pnsfm_pack_bias_eff_backward(const float* g, const float* Ssum, const float* b3, const float* dWeff_full, float* db3, float* dW2,
                                 int C, int d, int D, int k, void* stream) {
  if (C < 1 || d < 1 || d > 8 || D < 1 || k < 1) { set_error("pack_bias_eff_backward: bad sizes (C=%d d=%d D=%d k=%d)", C, d, D, k); return -1; }
  if (db3) {
    PNSFM_LAUNCH(pack_bias_eff_db3_kernel, dim3(d), dim3(64), 0, (hipStream_t)stream, g, Ssum, db3, C, d);
    if (int rc = check_launch("pack_bias_eff_backward (db3)")) return rc;
  }
  if (dW2) {
    const size_t total = (size_t)C * d * D * k * k;
    if (total >= (1ull << 32)) { set_error("pack_bias_eff_backward: >= 2^32 weight elements"); return -1; }
    PNSFM_LAUNCH(pack_dw2_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dWeff_full, g, b3,
                 dW2, d, D, k, (unsigned)total);
    if (int rc = check_launch("pack_bias_eff_backward (dW2)")) return rc;
  }
  return 0;
}

}  // extern "C"
