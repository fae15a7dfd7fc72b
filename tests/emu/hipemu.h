// hipemu.h -- TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// A tiny host-side emulator of the subset of the HIP execution model that the
// kernels under packnet-sfm_amd/csrc use, so that the *same kernel sources* can be
// compiled with the host clang++ (-DPNSFM_EMU) and their index math / tiling /
// MFMA fragment mapping checked on a machine without a GPU (this container).
//
//  * every GPU thread is a ucontext fiber; a workgroup's fibers are scheduled
//    round-robin inside ONE OS thread; workgroups run one after the other;
//  * __syncthreads() and the wave collectives (MFMA, shuffles) are generation
//    barriers that yield to the scheduler until every participant has arrived;
//  * v_mfma_f32_32x32x2_f32 is modelled with the gfx950 fragment layout
//    (A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31)
//    as a k-ordered fmaf chain, which is what the hardware computes bit-for-bit.
//
// The emulated library is only ever loaded by tests (tests/emu/build_emu.py);
// packnet_sfm.hip._lib refuses to load anything but the gfx950 build.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WaveState {
  int count = 0;
  unsigned gen = 0;
  float a[64], b[64];
  float a8[64][8], b8[64][8];
  double d[64];
  int alive = 0;
};

struct BlockState {
  int nthreads = 0;
  int alive = 0;
  int bar_count = 0;
  unsigned bar_gen = 0;
  std::vector<WaveState> waves;
  ucontext_t main_ctx;
  std::vector<ucontext_t> ctx;
  std::vector<char*> stacks;
  std::vector<char> done;
  int cur = 0;
  std::function<void()> body;
  char* dyn_smem = nullptr;
  dim3 bdim;
};

inline BlockState* g_block = nullptr;

inline void yield() { swapcontext(&g_block->ctx[g_block->cur], &g_block->main_ctx); }

inline int linear_tid() {
  return (int)(threadIdx.x + g_block->bdim.x * (threadIdx.y + g_block->bdim.y * threadIdx.z));
}

inline void syncthreads() {
  BlockState& B = *g_block;
  unsigned gen = B.bar_gen;
  if (++B.bar_count >= B.alive) {
    B.bar_count = 0;
    B.bar_gen++;
  } else {
    while (B.bar_gen == gen) yield();
  }
}

inline void wave_barrier(WaveState& W) {
  unsigned gen = W.gen;
  if (++W.count >= W.alive) {
    W.count = 0;
    W.gen++;
  } else {
    while (W.gen == gen) yield();
  }
}

inline WaveState& my_wave() { return g_block->waves[linear_tid() >> 6]; }
inline int my_lane() { return linear_tid() & 63; }

inline f32x16 mfma_f32_32x32x2f32(float a, float b, f32x16 c) {
  WaveState& W = my_wave();
  int l = my_lane();
  W.a[l] = a;
  W.b[l] = b;
  wave_barrier(W);
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    acc = fmaf(W.a[row], W.b[col], acc);            // k = 0
    acc = fmaf(W.a[row + 32], W.b[col + 32], acc);  // k = 1
    c[r] = acc;
  }
  wave_barrier(W);
  return c;
}

// v_mfma_f32_32x32x16_bf16: lane l supplies A[m = l&31][k = 8*(l>>5) + i] and B[k = 8*(l>>5) + i][n = l&31], i = 0..7, as
// bf16 pairs packed in 4 dwords (element i in bits [16*(i&1), +16) of dword i>>1); D layout as the f32 forms.  The 16
// products of a row/column pair are exact in fp32; the hardware's internal summation order is not documented, so the
// model sums them (and the accumulator) in double and rounds once -- tests that go through it compare with a tolerance.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
inline f32x16 mfma_f32_32x32x16_bf16(u32x4 a, u32x4 b, f32x16 c) {
  WaveState& W = my_wave();
  int l = my_lane();
  for (int i = 0; i < 8; ++i) {
    unsigned ua = (i & 1) ? (a[i >> 1] & 0xffff0000u) : (a[i >> 1] << 16);
    unsigned ub = (i & 1) ? (b[i >> 1] & 0xffff0000u) : (b[i >> 1] << 16);
    memcpy(&W.a8[l][i], &ua, 4);
    memcpy(&W.b8[l][i], &ub, 4);
  }
  wave_barrier(W);
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double acc = c[r];
    for (int h = 0; h < 2; ++h)
      for (int i = 0; i < 8; ++i) acc += (double)W.a8[row + 32 * h][i] * (double)W.b8[col + 32 * h][i];
    c[r] = (float)acc;
  }
  wave_barrier(W);
  return c;
}

// v_mfma_f32_16x16x32_bf16: lane l supplies A[m = l&15][k = 8*(l>>4) + i] and B[k = 8*(l>>4) + i][n = l&15], i = 0..7 (packed as
// above); lane l holds D[row = 4*(l>>4) + r][col = l&15], r = 0..3.  Summed in double and rounded once, like the 32x32 model.
typedef float f32x4 __attribute__((ext_vector_type(4)));
inline f32x4 mfma_f32_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c) {
  WaveState& W = my_wave();
  int l = my_lane();
  for (int i = 0; i < 8; ++i) {
    unsigned ua = (i & 1) ? (a[i >> 1] & 0xffff0000u) : (a[i >> 1] << 16);
    unsigned ub = (i & 1) ? (b[i >> 1] & 0xffff0000u) : (b[i >> 1] << 16);
    memcpy(&W.a8[l][i], &ua, 4);
    memcpy(&W.b8[l][i], &ub, 4);
  }
  wave_barrier(W);
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    double acc = c[r];
    for (int h = 0; h < 4; ++h)
      for (int i = 0; i < 8; ++i) acc += (double)W.a8[row + 16 * h][i] * (double)W.b8[col + 16 * h][i];
    c[r] = (float)acc;
  }
  wave_barrier(W);
  return c;
}

template <class T>
inline T shfl_generic(T v, int src_lane) {
  WaveState& W = my_wave();
  int l = my_lane();
  W.d[l] = (double)v;
  wave_barrier(W);
  T out = (src_lane >= 0 && src_lane < 64) ? (T)W.d[src_lane] : v;
  wave_barrier(W);
  return out;
}

inline void fiber_entry() {
  BlockState& B = *g_block;
  B.body();
  // thread exit: it no longer takes part in barriers
  int tid = B.cur;
  B.done[tid] = 1;
  B.alive--;
  WaveState& W = B.waves[tid >> 6];
  W.alive--;
  if (B.alive > 0 && B.bar_count >= B.alive && B.bar_count > 0) { B.bar_count = 0; B.bar_gen++; }
  if (W.alive > 0 && W.count >= W.alive && W.count > 0) { W.count = 0; W.gen++; }
  swapcontext(&B.ctx[tid], &B.main_ctx);
}

inline void run_block(dim3 grid, dim3 block, dim3 bidx, size_t shmem, const std::function<void()>& body) {
  static const size_t STACK = 256 * 1024;
  BlockState B;
  B.bdim = block;
  B.nthreads = (int)(block.x * block.y * block.z);
  B.alive = B.nthreads;
  int nw = (B.nthreads + 63) / 64;
  B.waves.resize(nw);
  for (int w = 0; w < nw; ++w) B.waves[w].alive = std::min(64, B.nthreads - 64 * w);
  B.ctx.resize(B.nthreads);
  B.stacks.resize(B.nthreads);
  B.done.assign(B.nthreads, 0);
  B.body = body;
  std::vector<char> smem(shmem + 64);
  B.dyn_smem = (char*)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
  g_block = &B;
  gridDim = grid;
  blockDim = block;
  blockIdx = bidx;
  for (int t = 0; t < B.nthreads; ++t) {
    B.stacks[t] = (char*)malloc(STACK);
    getcontext(&B.ctx[t]);
    B.ctx[t].uc_stack.ss_sp = B.stacks[t];
    B.ctx[t].uc_stack.ss_size = STACK;
    B.ctx[t].uc_link = &B.main_ctx;
    makecontext(&B.ctx[t], (void (*)())fiber_entry, 0);
  }
  int remaining = B.nthreads;
  long spins = 0;
  while (remaining > 0) {
    remaining = 0;
    for (int t = 0; t < B.nthreads; ++t) {
      if (B.done[t]) continue;
      B.cur = t;
      threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
      swapcontext(&B.main_ctx, &B.ctx[t]);
      if (!B.done[t]) remaining++;
    }
    if (++spins > 50000000L) { fprintf(stderr, "hipemu: deadlock suspected\n"); abort(); }
  }
  for (int t = 0; t < B.nthreads; ++t) free(B.stacks[t]);
  g_block = nullptr;
}

template <class F>
inline void launch(dim3 grid, dim3 block, size_t shmem, F f) {
  std::function<void()> body = f;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) run_block(grid, block, dim3(x, y, z), shmem, body);
}

}  // namespace hipemu

#define __syncthreads() hipemu::syncthreads()
template <class T> inline T __shfl_down(T v, unsigned d) { return hipemu::shfl_generic(v, hipemu::my_lane() + (int)d); }
template <class T> inline T __shfl_up(T v, unsigned d) { return hipemu::shfl_generic(v, hipemu::my_lane() - (int)d); }
template <class T> inline T __shfl_xor(T v, int m) { return hipemu::shfl_generic(v, hipemu::my_lane() ^ m); }
template <class T> inline T __shfl(T v, int src) { return hipemu::shfl_generic(v, src); }
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __fdividef(float a, float b) { return a / b; }
inline int hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline int hipMallocAsync(void** p, size_t n, hipStream_t) { *p = malloc(n); return *p ? 0 : 2; }
inline int hipFreeAsync(void* p, hipStream_t) { free(p); return 0; }
inline int hipGetLastError() { return 0; }
inline const char* hipGetErrorString(int) { return "emu"; }
