// Does the achievable v_mfma_f32_32x32x2_f32 rate depend on the DATA?  The chip clocks to its power budget
// (MI355X_MICROARCH.md "DVFS give-back"), and mfma_peak.hip feeds the matrix pipe near-constant operands.  Here every lane
// holds 8 A and 8 B operands loaded from a buffer filled with (a) zeros, (b) ones, (c) uniform random floats in [-1, 1)
// and loops over them from registers (no LDS, no memory traffic in the loop) -> TFLOP/s per data kind, 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k(const float* __restrict__ data, float* out, int iters) {
  const int tid = threadIdx.x;
  float a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = data[(j * 256 + tid) & 4095]; b[j] = data[((j + 8) * 256 + tid * 7) & 4095]; }
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + m) & 7], b[(j + 2 * m) & 7], acc[m], 0, 0, 0);
    if ((it & 63) == 63)   // keep the accumulators bounded (random data would otherwise overflow to inf: different toggling)
#pragma unroll
      for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] *= 1.0e-3f;
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  float *d, *out;
  hipMalloc(&d, 4096 * 4);
  const int iters = 2048;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bpc : {1, 2}) {
    const int grid = 256 * bpc;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    for (int kind = 0; kind < 4; ++kind) {
      std::vector<float> h(4096);
      srand(1);
      for (auto& v : h) v = kind == 0 ? 0.f : (kind == 1 ? 1.f : (kind == 2 ? (float)rand() / RAND_MAX * 2.f - 1.f : (float)(rand() % 8) * 0.125f));
      hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, out, 64);
      float best = 1e30f, tot = 0;
      for (int rep = 0; rep < 6; ++rep) {            // ~6 back-to-back launches: lets the clock settle under the load
        hipEventRecord(e0, 0);
        for (int q = 0; q < 8; ++q) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= 8; tot = ms; if (ms < best) best = ms;
      }
      const double flops = (double)grid * 4 * iters * 32.0 * (2.0 * 32 * 32 * 2);
      printf("data %-8s  blocks/CU %d  best %.3f ms %6.1f TFLOP/s   last %6.1f TFLOP/s\n",
             kind == 0 ? "zeros" : kind == 1 ? "ones" : kind == 2 ? "random" : "lowent", bpc, best, flops / best * 1e-9, flops / tot * 1e-9);
    }
    hipFree(out);
  }
  return 0;
}
