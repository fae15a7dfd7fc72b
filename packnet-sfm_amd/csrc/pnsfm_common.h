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
typedef hipemu::u32x4 pnsfm_u32x4;
static inline f32x16 pnsfm_mfma_bf16(pnsfm_u32x4 a, pnsfm_u32x4 b, f32x16 c) { return hipemu::mfma_f32_32x32x16_bf16(a, b, c); }
typedef hipemu::f32x4 f32x4;
static inline f32x4 pnsfm_mfma_bf16_16(pnsfm_u32x4 a, pnsfm_u32x4 b, f32x4 c) { return hipemu::mfma_f32_16x16x32_bf16(a, b, c); }
static inline unsigned pnsfm_f2u(float v) { unsigned u; memcpy(&u, &v, 4); return u; }
static inline float pnsfm_u2f(unsigned u) { float v; memcpy(&v, &u, 4); return v; }
// v_cvt_pk_bf16_f32: two fp32 -> two bf16 (round to nearest even), `lo` in bits [0,16), `hi` in bits [16,32)
static inline unsigned pnsfm_bf16_rne(float v) { const unsigned u = pnsfm_f2u(v); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }
static inline unsigned pnsfm_cvt_pk_bf16(float lo, float hi) { return pnsfm_bf16_rne(lo) | (pnsfm_bf16_rne(hi) << 16); }
// LDS-DMA: lane l of the wave copies 4 bytes from its own global address to lds_wave_base[l] (emulated synchronously)
static inline void pnsfm_glds4(const float* src, float* lds_wave_base) { lds_wave_base[hipemu::my_lane()] = *src; }
// 16-byte LDS-DMA: lane l copies 4 consecutive floats from its own global address to lds_wave_base[4*l .. 4*l+3]
static inline void pnsfm_glds16(const float* src, float* lds_wave_base) {
  float* d = lds_wave_base + 4 * hipemu::my_lane();
  d[0] = src[0]; d[1] = src[1]; d[2] = src[2]; d[3] = src[3];
}
// LDS-DMA with buffer addressing (buffer_load_dword{,x4} ... lds): lane l copies 4 / 16 bytes from base + voff_bytes to
// lds_wave_base[l] / [4l..4l+3]; an out-of-range lane (voff + size > bytes) writes ZEROS (checked on gfx950:
// tools/micro/blds_check.hip), which is how zero padding and the ragged last K-chunk are produced without a single compare.
#define PNSFM_DMA_INVALID 0x7ffffff0u
struct pnsfm_dma_buf { const char* base; unsigned bytes; };
static inline pnsfm_dma_buf pnsfm_make_dma_buf(const void* base, long bytes) {
  return pnsfm_dma_buf{(const char*)base, bytes <= 0 ? 0u : (unsigned)bytes};
}
static inline void pnsfm_dma4(const pnsfm_dma_buf& b, unsigned voff, float* lds_wave_base) {
  const bool ok = voff <= 0x7fffffffu && (unsigned long long)voff + 4u <= b.bytes;
  lds_wave_base[hipemu::my_lane()] = ok ? *reinterpret_cast<const float*>(b.base + voff) : 0.f;
}
static inline void pnsfm_dma16(const pnsfm_dma_buf& b, unsigned voff, float* lds_wave_base) {
  float* d = lds_wave_base + 4 * hipemu::my_lane();
  for (int i = 0; i < 4; ++i) {
    const unsigned o = voff + 4u * i;
    const bool ok = voff <= 0x7fffffffu && (unsigned long long)o + 4u <= b.bytes;
    d[i] = ok ? *reinterpret_cast<const float*>(b.base + o) : 0.f;
  }
}
#define PNSFM_UNIFORM(i) (i)
static inline void pnsfm_dma_wait() {}
// workgroup barriers with explicit counter waits (conv2d_bx3pp.h): the emulator's barrier is a full one either way
#define PNSFM_BARRIER_ALL() __syncthreads()
#define PNSFM_BARRIER_LDS() __syncthreads()
#define PNSFM_SCHED_FENCE() do {} while (0)
// buffer resource: loads whose per-lane byte offset is >= `bytes` return 0 (see the device version below)
struct pnsfm_buf { const char* base; unsigned bytes; };
static inline pnsfm_buf pnsfm_make_buf(const void* base, unsigned bytes) { return pnsfm_buf{(const char*)base, bytes}; }
static inline float pnsfm_buf_load(const pnsfm_buf& b, unsigned voff_bytes, unsigned soff_bytes) {
  if (voff_bytes + 4u > b.bytes || voff_bytes > 0x7fffffffu) return 0.f;
  return *reinterpret_cast<const float*>(b.base + (size_t)soff_bytes + voff_bytes);
}
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
// v_mfma_f32_32x32x16_bf16: D(32x32) += A(32x16) * B(16x32), bf16 operands (8 per lane: A[m = l&31][k = 8*(l>>5) + i],
// B[k = 8*(l>>5) + i][n = l&31]), f32 accumulate, 32 cycles/SIMD (8 passes) -- 16x the MAC rate of the f32 form.
typedef unsigned pnsfm_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pnsfm_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 pnsfm_mfma_bf16(pnsfm_u32x4 a, pnsfm_u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pnsfm_bf16x8, a), __builtin_bit_cast(pnsfm_bf16x8, b), c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16: D(16x16) += A(16x32) * B(32x16): lane l holds A[m = l&15][k = 8*(l>>4) + i], B[k = 8*(l>>4) + i][n = l&15]
// and D[row = 4*(l>>4) + r][col = l&15]; 16 cycles/SIMD (4 passes): the same MAC rate with a quarter of the accumulator registers.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 pnsfm_mfma_bf16_16(pnsfm_u32x4 a, pnsfm_u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pnsfm_bf16x8, a), __builtin_bit_cast(pnsfm_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pnsfm_f2u(float v) { return __float_as_uint(v); }
__device__ __forceinline__ float pnsfm_u2f(unsigned u) { return __uint_as_float(u); }
// v_cvt_pk_bf16_f32: two fp32 -> two bf16 (round to nearest even), `lo` in bits [0,16), `hi` in bits [16,32)
typedef __bf16 pnsfm_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pnsfm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pnsfm_cvt_pk_bf16(float lo, float hi) {
  const pnsfm_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pnsfm_bf16x2));
}
// global_load_lds_dword: asynchronous global -> LDS copy that bypasses the VGPRs.  Every active lane supplies its own
// global address; the LDS destination is wave-uniform base (M0) + lane*4.  Completion is tracked by vmcnt; a
// __syncthreads() drains it.
__device__ __forceinline__ void pnsfm_glds4(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// global_load_lds_dwordx4: the 16-byte form (1 KiB per wave instruction); both addresses must be 16-byte aligned.
__device__ __forceinline__ void pnsfm_glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// LDS-DMA with buffer addressing: buffer_load_dword{,x4} v_off, s[rsrc], 0 offen lds.  The descriptor (base, num_records)
// lives in SGPRs and is re-based by scalar code; a lane needs ONE 32-bit offset register and no per-element compare: an
// out-of-range lane (voff + size > bytes) writes ZEROS to its LDS slot (checked on gfx950: tools/micro/blds_check.hip).
struct pnsfm_dma_buf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ pnsfm_dma_buf pnsfm_make_dma_buf(const void* base, long bytes) {
  pnsfm_dma_buf b;
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, bytes <= 0 ? 0 : (int)bytes, 0x00020000);
  return b;
}
__device__ __forceinline__ void pnsfm_dma4(const pnsfm_dma_buf& b, unsigned voff, float* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, (int)voff, 0, 0, 0);
}
__device__ __forceinline__ void pnsfm_dma16(const pnsfm_dma_buf& b, unsigned voff, float* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, 0, 0, 0);
}
#define PNSFM_DMA_INVALID 0x7ffffff0u     // a voffset that is out of range for every descriptor
// s_waitcnt vmcnt(0): every LDS-DMA (and load) this wave issued has landed.  REQUIRED before the __syncthreads() that
// publishes DMA'd data to the other waves: hipcc's own wait insertion only guarantees a wave sees ITS OWN copies, and was
// seen to leave `s_waitcnt vmcnt(1)` in front of such a barrier (conv2d_bx3_kernel<2,2>: intermittent stale weight slabs).
__device__ __forceinline__ void pnsfm_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt 0, expcnt 7, lgkmcnt 15
// tell the compiler a value is wave-uniform (moves it to an SGPR)
#define PNSFM_UNIFORM(i) __builtin_amdgcn_readfirstlane(i)
// Workgroup barriers with EXPLICIT counter waits.  __syncthreads() drains vmcnt(0) -- every global load and LDS-DMA the wave has in
// flight -- in front of s_barrier; a wave that has just ISSUED loads whose data nobody needs before a later barrier only has to
// publish its LDS writes: PNSFM_BARRIER_LDS waits lgkmcnt(0) alone.  PNSFM_BARRIER_ALL is the full form (this wave's LDS-DMA has
// landed too).  The "memory" clobber keeps the compiler from moving LDS / global accesses across the barrier.
#define PNSFM_BARRIER_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define PNSFM_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// nothing is scheduled across this point (hipcc otherwise sinks a batch of LDS reads into the MFMA batch that follows it)
#define PNSFM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Raw buffer loads (buffer_load_dword v, v_off, s[rsrc], s_off offen): the address is base + s_off + v_off with a
// wave-uniform base and s_off, so a stencil's 72 neighbour loads need 9 offset VGPRs instead of 72 64-bit pointers, and
// the hardware range check (v_off + 4 > num_records -> returns 0, no memory access) implements zero padding: invalid
// lanes just carry an out-of-range v_off.  Callers pass s_off = 0 (one descriptor per uniform sub-tensor): how the scalar
// offset enters the range check differs between ISA documents, so it is not relied upon.
typedef int pnsfm_i32x4 __attribute__((ext_vector_type(4)));
struct pnsfm_buf { pnsfm_i32x4 rsrc; };
__device__ float pnsfm_llvm_raw_buffer_load_f32(pnsfm_i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ __forceinline__ pnsfm_buf pnsfm_make_buf(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  pnsfm_buf b;
  // the base is wave-uniform: keep the descriptor in SGPRs
  b.rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffu));
  b.rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));   // stride 0
  b.rsrc[2] = __builtin_amdgcn_readfirstlane((int)bytes);                              // num_records (bytes)
  b.rsrc[3] = 0x00020000;                                                              // gfx9 raw dword buffer
  return b;
}
__device__ __forceinline__ float pnsfm_buf_load(const pnsfm_buf& b, unsigned voff_bytes, unsigned soff_bytes) {
  return pnsfm_llvm_raw_buffer_load_f32(b.rsrc, (int)voff_bytes, (int)soff_bytes, 0);
}
#endif

#include <cstddef>
#include <cstdint>

namespace pnsfm {

// error plumbing shared by every entry point (api.hip owns the storage)
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// grow-only device scratch of the given (device, stream) (api.hip): valid until the next scratch_get on the same stream; null on
// failure (also when the stream is being captured into a hipGraph: the library's two-stage reductions are not capturable).
void* scratch_get(hipStream_t stream, size_t bytes);
struct ScratchLease {
  void* p;
  ScratchLease(hipStream_t s, size_t bytes) : p(nullptr) { if (bytes) p = scratch_get(s, bytes); }
  ScratchLease(const ScratchLease&) = delete;
  ScratchLease& operator=(const ScratchLease&) = delete;
  template <class T> T* as() const { return static_cast<T*>(p); }
};

// Block order of the conv kernels.  Workgroup p of a 1-D grid runs on XCD p % 8 (observed on MI355X; relied on for speed only) and
// every XCD has its own 4 MB L2, so round-robin neighbours share nothing: each XCD is handed a CONTIGUOUS range of the logical
// block order instead, and the kernels order their logical blocks so that neighbours share operands (all output-channel tiles of
// one pixel tile; all channel tiles of one pixel split).  mode 0: logical = physical (round 2's order).
#ifdef PNSFM_EMU
static inline unsigned pnsfm_xcd_logical_block(unsigned p, unsigned n) {
#else
__device__ __forceinline__ unsigned pnsfm_xcd_logical_block(unsigned p, unsigned n) {
#endif
  const unsigned xcd = p & 7u, slot = p >> 3, full = n >> 3, rem = n & 7u;
  return xcd * full + (xcd < rem ? xcd : rem) + slot;
}
int block_map_mode();      // api.hip: PNSFM_BLOCK_MAP = 0 (x fastest, round robin) | 1 (operand-sharing order) | 2 (+ XCD ranges, default)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline size_t ceil_div_sz(size_t a, size_t b) { return (a + b - 1) / b; }

// Multi-source input of a convolution: the K channels are the concatenation of up to three NCHW tensors (x [C0 channels], x1 [C1],
// x2 [the rest]) -- the decoder's torch.cat((unpack, skip, upsampled disparity), 1) without the copy.  C0 and C0 + C1 must be
// multiples of the kernels' channel granule (16 forward, 32 / 64 weight gradient) so that a K chunk never straddles two sources.
struct ConvSrc {
  const float* x1;
  const float* x2;
  int C0, C1;
};
// every source but the LAST ends on a `granule`-channel boundary (two sources: C0 + C1 == Cin is the ragged end of the channels,
// handled like a single tensor's ragged end by the buffer range check -- not a boundary)
static inline bool conv_src_aligned(const ConvSrc& ms, int Cin, int granule) {
  return ms.C0 % granule == 0 && (ms.C0 + ms.C1 == Cin || (ms.C0 + ms.C1) % granule == 0);
}

// --- conv2d implicit-GEMM geometry (shared by forward / backward-data / packers) -----------
// GEMM view: M = output channels, N = output pixels, K = (tap, input channel).
struct ConvGeom {
  int MT;             // 32-row MFMA tiles per wave along M (block M tile BM = 32*MT)
  int NT;             // 32-pixel MFMA tiles per wave along N (block covers 4 waves * NT * 32 pixels)
  int CI;             // input channels staged per K-chunk (even, <= 16)
  int mode;           // 0: 2-D pixel tile (4*NT rows x 32 cols), 1: linear run of 128*NT pixels, 2: TW x TH rectangle (split-bf16 kernels)
  int TW, TH;         // mode 2: tile width / height in output pixels (TW * TH <= 128 * NT)
  int tiles_x, tiles_per_img;
  int PH, PW;         // staged input patch (rows, cols) incl. halo
  int KP, MP;         // padded K-channels / M-channels of the packed weight
  int nchunks, splitK;
  int stem;           // variant 0 only: 1 = the 3-channel 5x5 stem kernel (conv2d_stem5_kernel) takes the launch
  int DMA;            // 0: patch staged through registers; 1: patch double-buffered in LDS, fetched by LDS-DMA;
                      // 2: fully pipelined kernel -- patch AND per-kernel-row weight slabs double-buffered by LDS-DMA
                      // 3..5: split-bf16 kernels (fp32 rebuilt from 6 bf16 MFMA products; conv2d_bx3.h)
  int G;              // DMA >= 2: taps per weight stage
  int PB;             // DMA >= 3 (split-bf16 variants, conv2d_bx3.h): patch buffers in LDS
  size_t smem_bytes;
};
ConvGeom conv_geom(int B, int Cin, int Cout, int H, int W, int ks, int S = 1);   // H, W: output size; S: stride
// padded dims of a packed weight [ks*ks][KP][MP] for a conv with K-channels `Kc`, M-channels `Mc`
int conv_pack_KP(int Kc);
int conv_pack_MP(int Mc);
int conv_pick_MT(int Mc);
bool conv_bx3_supported(int Kc, int ks);   // shapes the split-bf16 forward / backward-data kernels take

// tap-major weight-gradient kernel (conv2d_wgrad2.hip); the generic one lives in conv2d.hip and the autotuner picks
bool wgrad2_supported(int Cin, int Cout, int H, int W, int ks);
int wgrad2_total_tiles(int B, int H, int W);
int wgrad2_base_blocks(int Cin, int Cout, int ks);
int enqueue_wgrad2(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W, int ks,
                   int split, hipStream_t stream);

// second stage of the pixel-split f32 weight-gradient kernels (conv2d.hip): out0[n0] | out1[n1] = sum of Z slabs, in slab order
int launch_sum_slabs(const float* ws, size_t zstride, int Z, float* out0, size_t n0, float* out1, size_t n1, hipStream_t s);

// split-bf16 weight-gradient kernel (conv2d_wgrad3.hip)
bool wgrad3_supported(int Cin, int Cout, int H, int W, int ks);
bool wgrad3_nt2_ok(int Cin, int ks);
bool wgrad3_fits(int B, int Cin, int Cout, int H, int W);
int wgrad3_total_tiles(int B, int H, int W);
int wgrad3_WM(int Cout, int want);          // co tiles per workgroup (want: 0 = the most the layer fills, or 1 / 2 / 4)
int wgrad3_base_blocks(int Cin, int Cout, int ks, int NT, int WM);
int enqueue_wgrad3(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W, int ks,
                   int split, int NT, int WM, hipStream_t stream, const ConvSrc* ms = nullptr);

// nine-taps-per-workgroup 3x3 weight gradient on the 16x16x32 MFMA (conv2d_wgrad4.hip).  cfg = WCI (ci tiles of 16 channels per
// workgroup: 1 | 2) | TG << 4 (tile width in 8-pixel groups, 0 = library choice) | TR << 8 (tile rows, 0 = library choice)
bool wgrad4_supported(int Cin, int Cout, int H, int W, int ks);
int wgrad4_TG(int W, int want);
int wgrad4_TR(int H, int want);
int wgrad4_total_tiles(int B, int H, int W, int TG, int TR);
int wgrad4_base_blocks(int Cin, int Cout, int WCI);
int enqueue_wgrad4(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W, int split,
                   int cfg, hipStream_t stream, const ConvSrc* ms = nullptr);

// more than 64 KB of dynamic LDS needs hipFuncSetAttribute once per kernel AND device: `mask` (one static per kernel
// instantiation) remembers the devices already done (api.hip).  0 on success.
int ensure_lds_limit(const void* kernel, unsigned long long* mask, int bytes, const char* what);

// profiling of the dominant kernels with events on the launch stream (see api.hip)
void prof_begin(int kind, double flops, hipStream_t stream, const int* meta = nullptr);
void prof_end(int kind, hipStream_t stream);
bool stream_capturing(hipStream_t stream);   // hipGraph capture in progress on `stream`

}  // namespace pnsfm
