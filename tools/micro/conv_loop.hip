// Microbenchmark replicating the tap loop of conv2d_mfma_kernel (LDS-slab variant) without any global traffic, to find
// which part of the loop structure costs MFMA throughput.  FEAT bits: 1 per-tap barrier, 2 per-tap LDS slab store,
// 4 runtime tap->(ky,kx) division, 8 software-pipelined LDS reads (next batch issued before this batch's MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MT, int NT, int FEAT>
__global__ void __launch_bounds__(256) k_loop(float* out, int nchunks, int KS, int PW, int PS, int CI) {
  extern __shared__ float smem[];
  constexpr int BM = 32 * MT;
  float* patch = smem;
  float* wbuf = smem + ((CI * PS + 3) & ~3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
  for (int i = tid; i < CI * PS + 2 * CI * BM + 8; i += 256) smem[i] = (float)(i & 15) * 0.0625f;
  __syncthreads();
  int boff[NT];
  for (int nt = 0; nt < NT; ++nt) boff[nt] = (wave * NT + nt) * PW + l32;
  f32x16 acc[MT][NT];
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  const int KK = KS * KS, ksteps = CI >> 1;
  const int wrow = tid / (BM / 4), wc4 = tid - wrow * (BM / 4);
  float4 wreg = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int c = 0; c < nchunks; ++c) {
    int ky = 0, kx = 0;
    for (int tap = 0; tap < KK; ++tap) {
      const int cur = tap & 1;
      if (FEAT & 4) { ky = tap / KS; kx = tap - ky * KS; }
      const float* wb = wbuf + cur * CI * BM + half * BM + l32;
      const float* pb = patch + half * PS + ky * PW + kx;
      if (FEAT & 8) {
        float av[2][4][MT], bv[2][4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[0][j][mt] = wb[j * 2 * BM + mt * 32];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bv[0][j][nt] = pb[j * 2 * PS + boff[nt]];
        }
#pragma unroll
        for (int kb = 0; kb < 8; kb += 4) {
          const int s = (kb >> 2) & 1;
          if (kb + 4 < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) av[s ^ 1][j][mt] = wb[(kb + 4 + j) * 2 * BM + mt * 32];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) bv[s ^ 1][j][nt] = pb[(kb + 4 + j) * 2 * PS + boff[nt]];
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][j][mt], bv[s][j][nt], acc[mt][nt], 0, 0, 0);
        }
      } else {
        for (int kb = 0; kb < ksteps; kb += 4) {
          float av[4][MT], bv[4][NT];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[j][mt] = wb[(kb + j) * 2 * BM + mt * 32];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[j][nt] = pb[(kb + j) * 2 * PS + boff[nt]];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][mt], bv[j][nt], acc[mt][nt], 0, 0, 0);
        }
      }
      if (!(FEAT & 4)) { if (++kx == KS) { kx = 0; ++ky; } }
      if (FEAT & 2) { if (wrow < CI) *reinterpret_cast<float4*>(wbuf + (cur ^ 1) * CI * BM + wrow * BM + wc4 * 4) = wreg; }
      if (FEAT & 1) __syncthreads();
    }
  }
  float s = 0.f;
  for (int m = 0; m < MT; ++m) for (int n = 0; n < NT; ++n) for (int r = 0; r < 16; ++r) s += acc[m][n][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MT, int NT, int FEAT>
void run(const char* name, int grid, int KS) {
  const int CI = 16, PH = 4 * NT + KS - 1, PW = 32 + KS - 1, PS = PH * PW, BM = 32 * MT;
  const size_t smem = ((size_t)CI * PS + 4 + 2 * CI * BM) * 4;
  float* out;
  hipMalloc(&out, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nchunks = 4;
  hipLaunchKernelGGL((k_loop<MT, NT, FEAT>), dim3(grid), dim3(256), smem, 0, out, 1, KS, PW, PS, CI);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_loop<MT, NT, FEAT>), dim3(grid), dim3(256), smem, 0, out, nchunks, KS, PW, PS, CI);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * nchunks * KS * KS * 8.0 * MT * NT * (2.0 * 32 * 32 * 2);
  printf("%-44s grid %5d k%d smem %6zu  %8.3f ms  %7.1f TFLOP/s\n", name, grid, KS, smem, ms, flops / ms * 1e-9);
  hipFree(out);
}

int main() {
  for (int ks : {7, 3}) {
    run<2, 2, 0>("NT2 plain", 1920, ks);
    run<2, 2, 4>("NT2 +div", 1920, ks);
    run<2, 2, 1>("NT2 +barrier", 1920, ks);
    run<2, 2, 3>("NT2 +barrier+slabstore", 1920, ks);
    run<2, 2, 7>("NT2 +barrier+slabstore+div", 1920, ks);
    run<2, 2, 8>("NT2 pipelined reads", 1920, ks);
    run<2, 2, 15>("NT2 pipelined +barrier+slab+div", 1920, ks);
    run<2, 2, 0>("NT2 plain, 2304 blocks (9/CU)", 2304, ks);
    run<2, 1, 0>("NT1 plain", 3840, ks);
    run<2, 1, 7>("NT1 +barrier+slabstore+div", 3840, ks);
    run<2, 1, 15>("NT1 pipelined +barrier+slab+div", 3840, ks);
  }
  return 0;
}
