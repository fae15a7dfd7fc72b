// pnsfm_common.h -- shared declarations for the gfx950 kernels of the PackNet-SfM hot path.
//
// Everything in csrc/ is written for ONE target: MI355X / gfx950 / wave64. The only
// other way these sources are ever compiled is -DPNSFM_EMU with the host clang++ and
// tests/emu/hipemu.h, which exists so kernel index math can be checked on a machine
// with no GPU (test infrastructure; never a product path).
#pragma once

#ifdef PNSFM_EMU
#include "hipemu.h"
#define PNSFM_DYN_SMEM(T, name) T* name = reinterpret_cast<T*>(hipemu::g_block->dyn_smem)
#define PNSFM_LAUNCH(kern, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), (shmem), [=]() { (kern)(__VA_ARGS__); })
typedef hipemu::f32x16 f32x16;
static inline f32x16 pnsfm_mfma_32x32x2(float a, float b, f32x16 c) { return hipemu::mfma_f32_32x32x2f32(a, b, c); }
// LDS-DMA: lane l of the wave copies 4 bytes from its own global address to lds_wave_base[l] (emulated synchronously)
static inline void pnsfm_glds4(const float* src, float* lds_wave_base) { lds_wave_base[hipemu::my_lane()] = *src; }
#else
#include <hip/hip_runtime.h>
#define PNSFM_DYN_SMEM(T, name) extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
  T* name = reinterpret_cast<T*>(name##_raw)
#define PNSFM_LAUNCH(kern, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kern, (grid), (block), (shmem), (stream), __VA_ARGS__)
typedef float f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x2_f32: exact-f32 matrix FMA, 64 cycles/SIMD, D(32x32) += A(32x2) * B(2x32).
__device__ __forceinline__ f32x16 pnsfm_mfma_32x32x2(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// global_load_lds_dword: asynchronous global -> LDS copy that bypasses the VGPRs.  Every active lane supplies its own
// global address; the LDS destination is wave-uniform base (M0) + lane*4.  Completion is tracked by vmcnt; a
// __syncthreads() drains it.
__device__ __forceinline__ void pnsfm_glds4(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
#endif

#include <cstddef>
#include <cstdint>

namespace pnsfm {

// error plumbing shared by every entry point (api.hip owns the storage)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline size_t ceil_div_sz(size_t a, size_t b) { return (a + b - 1) / b; }

// --- conv2d implicit-GEMM geometry (shared by forward / backward-data / packers) -----------
// GEMM view: M = output channels, N = output pixels, K = (tap, input channel).
struct ConvGeom {
  int MT;             // 32-row MFMA tiles per wave along M (block M tile BM = 32*MT)
  int NT;             // 32-pixel MFMA tiles per wave along N (block covers 4 waves * NT * 32 pixels)
  int CI;             // input channels staged per K-chunk (even, <= 16)
  int mode;           // 0: 2-D pixel tile (4*NT rows x 32 cols), 1: linear run of 128*NT pixels
  int tiles_x, tiles_per_img;
  int PH, PW;         // staged input patch (rows, cols) incl. halo
  int KP, MP;         // padded K-channels / M-channels of the packed weight
  int nchunks, splitK;
  int DMA;            // 1: input patch double-buffered in LDS and fetched by LDS-DMA (global_load_lds)
  size_t smem_bytes;
};
ConvGeom conv_geom(int B, int Cin, int Cout, int H, int W, int ks, int S = 1);   // H, W: output size; S: stride
// padded dims of a packed weight [ks*ks][KP][MP] for a conv with K-channels `Kc`, M-channels `Mc`
int conv_pack_KP(int Kc);
int conv_pack_MP(int Mc);
int conv_pick_MT(int Mc);

// profiling of the dominant kernels with events on the launch stream (see api.hip)
void prof_begin(int kind, double flops, hipStream_t stream, const int* meta = nullptr);
void prof_end(int kind, hipStream_t stream);

}  // namespace pnsfm
