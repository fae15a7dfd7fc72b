// Semantics check of buffer_load_dword{,x4} ... lds (LDS-DMA with buffer addressing) on gfx950:
//  (1) does an out-of-range lane WRITE ZERO to its LDS slot (or leave it untouched)?  (2) is an exec-masked lane left untouched?
//  (3) is a 4-byte-aligned (not 16-byte-aligned) LDS base fine for the dword form?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* g, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 64) smem[i] = 7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)g, (short)0, n * 4, 0x00020000);
  int voff = tid * 4;
  if (tid % 4 == 1) voff = 0x7ffffff0;          // out of range
  if (tid % 4 != 2)                              // lanes == 2 mod 4: masked off
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 1), 4, voff, 0, 0, 0);   // misaligned base (+4 B)
  int voff16 = tid * 16;
  if (tid % 4 == 1) voff16 = 0x7ffffff0;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 256), 16, voff16, 0, 0, 0);
  __syncthreads();
  for (int i = tid; i < 1024; i += 64) out[i] = smem[i];
}
int main() {
  float *g, *o;
  hipMalloc(&g, 4096 * 4); hipMalloc(&o, 1024 * 4);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 100.f + i;
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, g, o, 4096);
  float r[1024]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("dword form, base smem+1: slots 0..9: "); for (int i = 0; i < 10; ++i) printf("%g ", r[i]); printf("\n");
  printf("  expect [7 (untouched slot 0), 100 (lane0), 0-or-7 (lane1 OOB), 7 (lane2 masked), 103, 104, OOB, 7, 107 ...]\n");
  printf("dwordx4 form at smem+256: "); for (int i = 256; i < 256 + 16; ++i) printf("%g ", r[i]); printf("\n");
  return 0;
}
