// calib.hip -- box calibration kernels for bench.py's `calibration` block (round 6).
//
// Leases of the same MI355X SKU differ by +-3-5 % in what the matrix pipe and the memory system sustain (power-limited clocks), which
// is more than a round of kernel work moves; the bench line therefore carries two figures measured on THIS box right before the
// timed region:
//   * pnsfm_calib_mfma: the bare six-product v_mfma_f32_32x32x16_bf16 stream of the split-bf16 arithmetic (csrc/conv2d_bx3.h) with
//     every operand in registers -- four independent accumulator tiles per wave, pieces walked (l,h) (h,l) (m,m) (m,h) (h,m) (h,h),
//     four waves per workgroup, no memory traffic: the rate the conv kernels would reach if nothing but their MFMAs existed
//     (tools/micro/bf16x3_check.hip measures 1838 TFLOP/s bf16 = 306 fp32-equivalent on the builder's boxes);
//   * pnsfm_calib_copy: a float4 streaming copy (read n + write n floats): the HBM rate a bandwidth-bound kernel of this library sees.
// Nothing on the training step calls these.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

__global__ void __launch_bounds__(256) calib_mfma_kernel(float* __restrict__ out, int iters) {
  pnsfm_u32x4 a[3], b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[i][j] = 0x3f803f80u + threadIdx.x * (i + 1) + j;
      b[i][j] = 0x3f803f80u + threadIdx.x * 3 + i + j;
    }
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#define PNSFM_CAL(x, y) _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[t] = pnsfm_mfma_bf16(a[x], b[y], acc[t])
    PNSFM_CAL(2, 0); PNSFM_CAL(0, 2); PNSFM_CAL(1, 1); PNSFM_CAL(1, 0); PNSFM_CAL(0, 1); PNSFM_CAL(0, 0);
#undef PNSFM_CAL
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_calib_mfma(float* sink, int blocks, int iters, void* stream) {
  if (!sink || blocks <= 0 || iters <= 0) { set_error("calib_mfma: bad arguments"); return -1; }
  PNSFM_LAUNCH(calib_mfma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
  return check_launch("calib_mfma");
}

int pnsfm_calib_copy(const float* src, float* dst, size_t n_floats, void* stream) {
  if (!src || !dst || n_floats % 4 != 0) { set_error("calib_copy: needs two buffers and a multiple of 4 floats"); return -1; }
  if (n_floats == 0) return 0;
  PNSFM_LAUNCH(calib_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src),
               reinterpret_cast<float4*>(dst), n_floats / 4);
  return check_launch("calib_copy");
}

}  // extern "C"
