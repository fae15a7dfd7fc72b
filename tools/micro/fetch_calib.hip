// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the conv kernels use
// (MI355X_MICROARCH.md, HBM: FETCH_SIZE reports 1/2 of the bytes of a 16-B/lane coalesced read; other widths are uncalibrated).
// Each kernel moves a KNOWN number of bytes:
//   read_dword   : 4 B/lane raw buffer loads, 64 consecutive floats per wave instruction   (the conv patch staging)
//   read_dwordx4 : 16 B/lane global loads, 1 KB per wave instruction
//   read_ldsdma  : 16 B/lane buffer_load ... lds (LDS-DMA), 1 KB per wave instruction      (the conv weight stream)
//   write_dword  : 4 B/lane stores, 32 lanes = one 128-B run (the conv epilogue: two 128-B runs per wave instruction)
//   write_dwordx4: 16 B/lane stores
// run under:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -- ./fetch_calib   (and again with WRITE_SIZE);
// tools/pmc_calib.py turns the two CSVs into byte-per-counter factors.   hipcc --offload-arch=gfx950 -O2 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ float llvm_raw_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");

__global__ void __launch_bounds__(256) read_dword(const float* __restrict__ src, float* __restrict__ sink, size_t n_per_block) {
  const float* p = src + (size_t)blockIdx.x * n_per_block;
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)(n_per_block * 4));
  r[3] = 0x00020000;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < n_per_block; i += 256) s += llvm_raw_buffer_load_f32(r, (int)(i * 4), 0, 0);
  if (s == 1.2345f) sink[0] = s;
}
__global__ void __launch_bounds__(256) read_dwordx4(const float4* __restrict__ src, float* __restrict__ sink, size_t n4_per_block) {
  const float4* p = src + (size_t)blockIdx.x * n4_per_block;
  float s = 0.f;
  for (size_t i = threadIdx.x; i < n4_per_block; i += 256) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 1.2345f) sink[0] = s;
}
__global__ void __launch_bounds__(256) read_ldsdma(const float* __restrict__ src, float* __restrict__ sink, size_t n_per_block) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 1024];       // 4 waves x 4 KB ring
  const float* p = src + (size_t)blockIdx.x * n_per_block;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), (short)0, (int)(n_per_block * 4), 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* dst = lds + wave * 1024;
  // each wave instruction moves 1 KB (256 floats); the workgroup advances 4 KB per step
  for (size_t i = (size_t)wave * 256; i < n_per_block; i += 1024)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + ((i >> 10) & 3) * 256), 16, (int)((i + lane * 4) * 4), 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (lds[threadIdx.x] == 1.2345f) sink[0] = 1.f;
}
__global__ void __launch_bounds__(256) write_dword(float* __restrict__ dst, size_t n_per_block, size_t row_stride) {
  // lanes 0-31 write 128 contiguous bytes of one row, lanes 32-63 the same columns of a row `row_stride` floats away (epilogue shape)
  float* p = dst + (size_t)blockIdx.x * n_per_block;
  const int half = (threadIdx.x >> 5) & 1, l32 = threadIdx.x & 31, w = threadIdx.x >> 6;
  for (size_t i = (size_t)w * 32; i < n_per_block / 2; i += 128) p[(size_t)half * (n_per_block / 2) + i + l32] = 1.f;
  (void)row_stride;
}
__global__ void __launch_bounds__(256) write_dwordx4(float4* __restrict__ dst, size_t n4_per_block) {
  float4* p = dst + (size_t)blockIdx.x * n4_per_block;
  for (size_t i = threadIdx.x; i < n4_per_block; i += 256) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const size_t bytes = (size_t)2 << 30;          // 2 GiB: far beyond the 256 MiB Infinity Cache
  const int blocks = 4096;
  float *src, *dst, *sink;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 64);
  hipMemset(src, 0, bytes); hipMemset(dst, 0, bytes);
  const size_t n = bytes / 4 / blocks;           // floats per block (multiple of 1024)
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_dword, dim3(blocks), dim3(256), 0, 0, (const float*)src, sink, n);
    hipLaunchKernelGGL(read_dwordx4, dim3(blocks), dim3(256), 0, 0, (const float4*)src, sink, n / 4);
    hipLaunchKernelGGL(read_ldsdma, dim3(blocks), dim3(256), 0, 0, (const float*)src, sink, n);
    hipLaunchKernelGGL(write_dword, dim3(blocks), dim3(256), 0, 0, dst, n, (size_t)0);
    hipLaunchKernelGGL(write_dwordx4, dim3(blocks), dim3(256), 0, 0, (float4*)dst, n / 4);
    hipDeviceSynchronize();
  }
  printf("true_bytes_per_launch %zu\n", bytes);
  return 0;
}
