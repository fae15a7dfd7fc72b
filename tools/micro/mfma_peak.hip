// Calibration microbenchmark: what does v_mfma_f32_32x32x2_f32 sustain on this chip (a) alone, (b) fed by LDS reads
// in the pattern of conv2d_mfma_kernel (per k-step: MT A-reads + NT B-reads, MT*NT MFMAs), at various occupancies?
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT, int NT, int MODE>   // MODE 0: registers only; 1: operands from LDS each k-step (batched by 4)
__global__ void __launch_bounds__(256) k_mfma(float* out, int iters, int lds_floats) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < lds_floats; i += 256) smem[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[MT][NT];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  float a0 = (float)lane * 0.001f, b0 = (float)(lane & 3);
  const float* wb = smem + (lane >> 5) * 64 + (lane & 31);
  const float* pb = smem + 4096 + (lane >> 5) * 400 + (lane & 31);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m][n], 0, 0, 0);
    } else {
      float av[4][MT], bv[4][NT];
      const int o = (it & 7) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int m = 0; m < MT; ++m) av[j][m] = wb[(o + j) * 128 + m * 32];
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[j][n] = pb[(o + j) * 800 + n * 40];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][m], bv[j][n], acc[m][n], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MT, int NT, int MODE>
void run(const char* name, int blocks_per_cu, size_t smem) {
  float* out;
  const int grid = 256 * blocks_per_cu, iters = 4096;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_mfma<MT, NT, MODE>), dim3(grid), dim3(256), smem, 0, out, 16, (int)(smem / 4));
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_mfma<MT, NT, MODE>), dim3(grid), dim3(256), smem, 0, out, iters, (int)(smem / 4));
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * MT * NT * (2.0 * 32 * 32 * 2);
  printf("%-28s blocks/CU %d  smem %6zu  %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, smem, ms, flops / ms * 1e-9);
  hipFree(out);
}

// sustained mode: back-to-back launches of the register-only MFMA loop for `seconds`, TFLOP/s per 0.25 s window
// (does the clock hold under a sustained fp32-MFMA load, or does power management pull it down?)
static void sustained(double seconds) {
  float* out;
  const int grid = 256 * 2, iters = 4096;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const double flops = (double)grid * 4 * iters * 4.0 * 2 * 2 * (2.0 * 32 * 32 * 2);
  double t = 0;
  while (t < seconds) {
    hipEventRecord(e0, 0);
    int n = 0;
    for (; n < 40; ++n) hipLaunchKernelGGL((k_mfma<2, 2, 0>), dim3(grid), dim3(256), 40000, 0, out, iters, 10000);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    t += ms * 1e-3;
    printf("t=%5.2fs  %7.1f TFLOP/s\n", t, n * flops / ms * 1e-9);
  }
  hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 1) { sustained(atof(argv[1])); return 0; }
  for (int b : {1, 2, 3, 4}) run<2, 2, 0>("regs only  MT2 NT2", b, 40000);
  for (int b : {1, 2, 3, 4}) run<2, 2, 1>("LDS-fed    MT2 NT2", b, 40000);
  for (int b : {1, 2, 4, 6}) run<2, 1, 1>("LDS-fed    MT2 NT1", b, 24000);
  for (int b : {1, 2, 4}) run<2, 1, 0>("regs only  MT2 NT1", b, 24000);
  return 0;
}
