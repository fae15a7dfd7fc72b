// gfx950 checks behind the split-bf16 convolution (csrc/conv2d_bx3.hip):
//  (1) operand layout of v_mfma_f32_32x32x16_bf16: lane l supplies A[m = l&31][k = 8*(l>>5) + i], B[k = 8*(l>>5) + i][n = l&31],
//      i = 0..7 (element i in bits [16*(i&1) .. ] of dword i>>1); D as the f32 forms: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31;
//  (2) accuracy of an fp32 GEMM done as 6 bf16 products of exact 3-way round-to-nearest splits (hh, hm, mh, hl, lh, mm) against
//      v_mfma_f32_32x32x2_f32 and against an fp64 reference;
//  (3) issue rate of the 6-product stream.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// exact 3-way split, round-to-nearest pieces (v_cvt_pk_bf16_f32); pieces returned in the HIGH 16 bits
__device__ __forceinline__ unsigned rne(float v) { f32x2 p = {0.f, v}; return __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2)) & 0xffff0000u; }
__device__ __forceinline__ void split3(float v, unsigned& h, unsigned& m, unsigned& l) {
  h = rne(v);
  const float r1 = v - __uint_as_float(h);
  m = rne(r1);
  const float r2 = r1 - __uint_as_float(m);
  l = rne(r2);
}
__device__ __forceinline__ unsigned pack2(unsigned even, unsigned odd) { return (even >> 16) | (odd & 0xffff0000u); }

// C[32][32] = A[32][K] * B[K][32], K multiple of 16
__global__ void gemm_bx3(const float* A, const float* B, float* C, float* C32, int K, int nprod) {
  const int l = threadIdx.x, m = l & 31, kh = l >> 5;
  f32x16 acc = {0}, acc32 = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    u32x4 ah, am, al, bh, bm, bl;
    for (int i = 0; i < 8; i += 2) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3(A[m * K + k0 + 8 * kh + i], h0, m0, l0);
      split3(A[m * K + k0 + 8 * kh + i + 1], h1, m1, l1);
      ah[i >> 1] = pack2(h0, h1); am[i >> 1] = pack2(m0, m1); al[i >> 1] = pack2(l0, l1);
      split3(B[(k0 + 8 * kh + i) * 32 + m], h0, m0, l0);
      split3(B[(k0 + 8 * kh + i + 1) * 32 + m], h1, m1, l1);
      bh[i >> 1] = pack2(h0, h1); bm[i >> 1] = pack2(m0, m1); bl[i >> 1] = pack2(l0, l1);
    }
#define MF(a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0)
    if (nprod >= 6) { MF(al, bh); MF(ah, bl); MF(am, bm); }
    if (nprod >= 3) { MF(am, bh); MF(ah, bm); }
    MF(ah, bh);
    for (int kk = 0; kk < 16; kk += 2)
      acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * K + k0 + kk + kh], B[(k0 + kk + kh) * 32 + m], acc32, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
    C[row * 32 + m] = acc[r];
    C32[row * 32 + m] = acc32[r];
  }
}

__global__ void __launch_bounds__(256) rate(float* out, int iters, int mode) {
  u32x4 a[3], b[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) { a[i][j] = 0x3f803f80u + threadIdx.x * (i + 1) + j; b[i][j] = 0x3f803f80u + threadIdx.x * 3 + i + j; }
  f32x16 acc[4] = {{0}, {0}, {0}, {0}};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#define MR(x, y) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[x]), __builtin_bit_cast(bf16x8, b[y]), acc[t], 0, 0, 0)
      if (mode == 0) { MR(2, 0); MR(0, 2); MR(1, 1); MR(1, 0); MR(0, 1); MR(0, 0); }     // dependent chain per tile
    }
    if (mode == 1) {                                                                     // tiles interleaved
#define MQ(x, y) for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[x]), __builtin_bit_cast(bf16x8, b[y]), acc[t], 0, 0, 0)
      MQ(2, 0); MQ(0, 2); MQ(1, 1); MQ(1, 0); MQ(0, 1); MQ(0, 0);
    }
  }
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int K = 4096;
  std::vector<float> A(32 * K), B(K * 32);
  srand(5);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (1.f + (rand() % 7));
  float *dA, *dB, *dC, *dC32;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096); hipMalloc(&dC32, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<double> ref(1024, 0.0), mag(1024, 0.0);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j)
      for (int k = 0; k < K; ++k) { ref[i * 32 + j] += (double)A[i * K + k] * B[k * 32 + j]; mag[i * 32 + j] += fabs((double)A[i * K + k] * B[k * 32 + j]); }
  for (int nprod : {1, 3, 6}) {
    hipLaunchKernelGGL(gemm_bx3, dim3(1), dim3(64), 0, 0, dA, dB, dC, dC32, K, nprod);
    float C[1024], C32[1024];
    hipMemcpy(C, dC, 4096, hipMemcpyDeviceToHost); hipMemcpy(C32, dC32, 4096, hipMemcpyDeviceToHost);
    double e = 0, e32 = 0;
    for (int i = 0; i < 1024; ++i) { e = fmax(e, fabs(C[i] - ref[i]) / mag[i]); e32 = fmax(e32, fabs(C32[i] - ref[i]) / mag[i]); }
    printf("K=%d  %d bf16 products: max |err| / sum|a||b| = %.3e   (f32 MFMA: %.3e; 2^-24 = %.3e)\n", K, nprod, e, e32, ldexp(1.0, -24));
  }
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, out, 100, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(1024), dim3(256), 0, 0, out, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 1024.0 * 4 * iters * 24;
    printf("mode %d: %.2f ms, %.1f TF bf16 (%.1f TF fp32-equivalent at 6 products)\n", mode, ms, mfma * 32 * 32 * 16 * 2 / ms / 1e9,
           mfma * 32 * 32 * 16 * 2 / 6 / ms / 1e9);
  }
  return 0;
}
