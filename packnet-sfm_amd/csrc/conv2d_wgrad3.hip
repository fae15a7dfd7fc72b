// conv2d_wgrad3.hip -- weight gradient of a stride-1 KxK convolution on the bf16 matrix pipe (split-bf16 arithmetic).
//
// Replaces the autograd weight gradient of nn.Conv2d in the reference's Conv2D / ResidualConv / Pack / Unpack blocks
//   (/root/reference/packnet_sfm/networks/layers/packnet/layers01.py:28-36, 57-60, 235-246, 274-281):
//   dW[co][ci][ky][kx] = sum_{b, y, x} dY[b][co][y][x] * X[b][ci][y + ky - P][x + kx - P]
// with the arithmetic of conv2d_bx3.h: every fp32 operand is split EXACTLY into three bf16 pieces (h, m, l) and the product
// is rebuilt from the 6 piece products hh, hm, mh, hl, lh, mm with fp32 accumulation (v_mfma_f32_32x32x16_bf16): fp32-class
// error at 16/6 the MAC rate of v_mfma_f32_32x32x2_f32.
//
// GEMM view PER TAP: M = co, N = ci, K = pixels; one MFMA k-step = 16 consecutive pixels of an image row.
//   * a workgroup owns ONE KERNEL ROW ky (logical block x = (ci tile, ky)): its waves accumulate the KS taps of that row for a
//     32(co) x 32*NT(ci) tile each, KS*NT accumulator tiles;  4 waves = WM co tiles x WK pixel shares (WM = 4, 2, 1 for
//     Cout >= 97, >= 33, smaller): all waves share one X patch in LDS;
//   * A operand (dY): lane (co = l&31, half = l>>5) needs 8 consecutive pixels -- 32 contiguous bytes of the NCHW tensor: read
//     straight from global memory (buffer loads, up to a whole tile ahead), split in registers, reused by the KS*NT*6 MFMAs of
//     the k-step; dY never goes through LDS.  The bias gradient is the running sum of the same registers;
//   * B operand (X): the patch rows for this ky (TR = 4 rows x 32 + 16 columns, 8 columns of halo either side so that every
//     8-pixel group is 16-byte aligned) are staged per pixel tile: fp32 rows -> registers (issued before the previous tile's
//     MFMAs) -> 3 bf16 pieces -> LDS [piece][ci][row][col] (channel stride 8 * odd elements: conflict-free ds_read_b128).
//     A lane reads the aligned 8-pixel block and its two neighbours (3 ds_read_b128 per piece) and builds the KS shifted
//     operands in registers: an even shift is a register renaming, an odd shift one v_alignbit_b32 per dword;
//   * the WK pixel shares of a workgroup are summed through LDS; pixel tiles are split over the third logical block index and the partial
//     tensors of a split launch go to a scratch buffer ([split][ky][co][kx][ci]: coalesced stores) that a second kernel sums
//     in a fixed order -- no atomics anywhere: the gradient is bit-reproducible run to run.
// Widths that are a multiple of 4 but not of 8 (20, 4: the upper half of an 8-pixel group may lie past the row end) run a
// variant that range-checks the two halves separately; tiles are 4 rows x 32 columns, or 4 x 16 where that wastes fewer columns (W = 40, 80, 20).  k = 3, 5, 7.
// Roofline: MFMA-bound: 2*Cout*Cin*K*K*B*H*W algorithmic flop against 2500/6 TFLOP/s (bf16 dense peak / 6 products).
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

struct Wgrad3Args {
  const float* x1;   // multi-source input (ConvSrc, pnsfm_common.h): channels [C0, C01) live in x1, [C01, Cin) in x2
  const float* x2;
  int C0, C01;       // C0 = C01 = Cin for a single source
  const float* x;    // [B][Cin][H][W]  (multi-source: [B][C0][H][W])
  const float* dy;   // [B][Cout][H][W]
  float* dw;         // [Cout][Cin][KS][KS]   written directly when the launch has ONE pixel split ...
  float* dbias;      // [Cout] or null
  float* ws;         // ... else partial sums [split][KS(ky)][COP][KS(kx)][CIP] (+ [split][COP] bias partials at ws_bias),
  float* ws_bias;    //     reduced by wgrad3_reduce_kernel: no atomics, a fixed summation order
  int COP, CIP;      // padded channel extents of the workspace (whole workgroup tiles)
  int B, Cin, Cout, H, W;
  int tiles_x, tiles_per_img, total_tiles, tiles_per_split;
  int ci_tiles;      // logical grid x = ci_tiles * KS
  int gx, gy, bmap;  // 1-D launch: ci tiles * KS, co groups, block order (pnsfm_common.h: block_map_mode)
};
#ifdef PNSFM_WG_ABLATE          // compile-time what-if mask (see conv2d_wgrad4.hip)
#define PNSFM_WG_ABL(a) (PNSFM_WG_ABLATE)
#else
#define PNSFM_WG_ABL(a) 0
#endif

#ifdef PNSFM_EMU
static inline unsigned w3_alignbit16(unsigned hi, unsigned lo) { return (lo >> 16) | (hi << 16); }
#else
__device__ __forceinline__ unsigned w3_alignbit16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
#endif

// 8 consecutive fp32 values -> three 16-byte bf16 pieces (see conv2d_bx3.h: exact, round-to-nearest pieces)
__device__ __forceinline__ void w3_split8(const float (&v)[8], pnsfm_u32x4& H, pnsfm_u32x4& M, pnsfm_u32x4& L) {
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const unsigned h = pnsfm_cvt_pk_bf16(v[i], v[i + 1]);
    const float r0 = v[i] - pnsfm_u2f(h << 16), r1 = v[i + 1] - pnsfm_u2f(h & 0xffff0000u);
    const unsigned m = pnsfm_cvt_pk_bf16(r0, r1);
    const float s0 = r0 - pnsfm_u2f(m << 16), s1 = r1 - pnsfm_u2f(m & 0xffff0000u);
    H[i >> 1] = h;
    M[i >> 1] = m;
    L[i >> 1] = pnsfm_cvt_pk_bf16(s0, s1);
  }
}

// OCC = 3 (3x3, one ci tile per wave only): register budget of three workgroups per CU (<= 168 VGPRs: shorter dY ring, one patch tile in
// flight) -- the third wave per SIMD covers the staging / barrier phases the other two leave the matrix pipe idle in.
template <int KS, int NT, int WM, int TCv, int OCC = 2>
struct Wgrad3Geom {
  static constexpr int P = KS / 2, KK = KS * KS;
  static constexpr int WK = 4 / WM;                      // pixel shares of a workgroup
  static constexpr int TR = 4, TC = TCv;                 // pixel tile: 4 rows x 32 (16) columns = 8 (4) k-steps
  static constexpr int SEG = TC / 16;                    // k-steps per tile row
  static constexpr int RS = TC + 16;                     // patch row: 8 halo + TC + 8 halo elements
  static constexpr int CS = TR * RS + 8;                 // channel stride (elements): 200 = 8 * 25 / 136 = 8 * 17 -> conflict-free b128
  static constexpr int NCI = 32 * NT;
  static constexpr int PIECE = NCI * CS;                 // elements of one piece plane
  static constexpr int SMEM = 3 * PIECE * 2;             // bytes
  static constexpr int ITEMS = NCI * TR * (RS / 8);      // (channel, row, 8-column group) items of the patch
  static constexpr int NIT = ITEMS / 256;                // per thread (ITEMS = 768 * NT or 512 * NT)
  static constexpr int KSTEPS = TR * SEG;                // 8 or 4
  static constexpr int KPW = KSTEPS / WK;                // k-steps of a tile per wave
  // dY fragments in flight per wave (8 registers each): a whole tile ahead where the accumulators leave room -- the loads
  // are issued RD k-steps before use, which is what hides the global-memory latency when only one workgroup fits a CU
  static constexpr int RDW = (KS >= 7 || NT == 2 || (OCC == 3 && TCv == 32)) ? 2 : ((KS == 5 || OCC == 3) ? 4 : 8);
  static constexpr int RD = 2 * KPW < RDW ? 2 * KPW : RDW;      // up to TWO tiles ahead (RD divides 2 * KPW)
  // patch prefetch depth in tiles: a 3x3 tile with one ci tile per wave is only 1-2 us of MFMAs -- less than the latency of
  // the loads issued at its start -- so those kernels keep two tiles of raw patch data in flight
  static constexpr int PDX = (KS == 3 && NT == 1 && KPW <= 4 && OCC != 3) ? 2 : 1;
};

// MASKED: W % 8 == 4 -- the upper half of an 8-pixel group may lie past the end of an image row: the two 4-pixel halves are
// range-checked separately (an out-of-row half gets an out-of-range buffer offset and reads as zero)
template <int KS, int NT, int WM, int TCv, bool MASKED, int OCC = 2>
__global__ void __launch_bounds__(256, OCC) conv2d_wgrad3_kernel(Wgrad3Args a) {
  using Gm = Wgrad3Geom<KS, NT, WM, TCv, OCC>;
  constexpr int P = Gm::P, KK = Gm::KK, WK = Gm::WK, TR = Gm::TR, TC = Gm::TC, RS = Gm::RS, CS = Gm::CS, NCI = Gm::NCI, SEG = Gm::SEG;
  constexpr int PIECE = Gm::PIECE, NIT = Gm::NIT, KPW = Gm::KPW, RD = Gm::RD, PDX = Gm::PDX;
  PNSFM_DYN_SMEM(unsigned char, smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int wm = wave % WM, wk = wave / WM;
  const int H = a.H, W = a.W, HW = H * W;
  // logical block ((ci tile, kernel row), co group, pixel split): first index fastest -- the workgroups of one pixel split read the
  // same dY and X -- and a contiguous range of that order per XCD (pnsfm_common.h)
  const unsigned Lb = a.bmap == 2 ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const int bx = (int)(Lb % (unsigned)a.gx), by = (int)((Lb / (unsigned)a.gx) % (unsigned)a.gy), bz = (int)(Lb / (unsigned)(a.gx * a.gy));
  const int cit = bx / KS, ky = bx - cit * KS;
  const int ci0 = cit * NCI;
  const int co0 = (by * WM + wm) * 32;
  const int t_begin = bz * a.tiles_per_split;
  int t_end = t_begin + a.tiles_per_split;
  if (t_end > a.total_tiles) t_end = a.total_tiles;

  f32x16 acc[NT][KS];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][k][r] = 0.f;
  float bsum = 0.f;
  const bool do_bias = a.dbias != nullptr && bx == 0;      // ci tile 0, kernel row 0

  // ---- tile cursors.  Index arithmetic is kept out of the loop: a tile's origin (image, first row, first column) is advanced
  // incrementally in scalar registers for the current tile and the two behind it (the prefetch targets), and every per-lane
  // byte offset is split into a loop-invariant lane part and a per-tile scalar part (ONE descriptor per tensor: rows of
  // padded channels past Cin / Cout read the next image's data or zero, and feed accumulator rows that are never stored).
  struct Cur { int b, y0, x0; };
  const int tiles_y = a.tiles_per_img / a.tiles_x;
  auto advance = [&](Cur& c) {
    c.x0 += TC;
    if (c.x0 >= a.tiles_x * TC) {
      c.x0 = 0;
      c.y0 += TR;
      if (c.y0 >= tiles_y * TR) { c.y0 = 0; ++c.b; }
    }
  };
  Cur cur[3];
  {
    const int b = t_begin / a.tiles_per_img, tt = t_begin - b * a.tiles_per_img, ty = tt / a.tiles_x;
    cur[0].b = b; cur[0].y0 = ty * TR; cur[0].x0 = (tt - ty * a.tiles_x) * TC;
    cur[1] = cur[0]; advance(cur[1]);
    cur[2] = cur[1]; advance(cur[2]);
  }
  // the tensor this workgroup's ci tile lives in (tiles never straddle two sources: the entry point checks the granule)
  const float* xs = a.x;
  int Cs = a.C0, lci0 = ci0;
  if (ci0 >= a.C0) {
    if (ci0 < a.C01) { xs = a.x1; Cs = a.C01 - a.C0; lci0 = ci0 - a.C0; }
    else { xs = a.x2; Cs = a.Cin - a.C01; lci0 = ci0 - a.C01; }
  }
  const pnsfm_buf xbuf = pnsfm_make_buf(xs, (unsigned)((size_t)a.B * Cs * HW * 4));
  const pnsfm_buf dybuf = pnsfm_make_buf(a.dy, (unsigned)((size_t)a.B * a.Cout * HW * 4));

  // patch items of this thread: (channel, row, 8-column group); LDS slot and lane part of the global offset are fixed
  int it_lds[NIT], it_ry[NIT], it_gx[NIT], it_lane[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int e = it * 256 + tid;
    const int ci = e / (TR * (RS / 8));
    const int rem = e - ci * (TR * (RS / 8));
    const int r = rem / (RS / 8), g = rem - r * (RS / 8);
    it_lds[it] = (ci * CS + r * RS + 8 * g) * 2;
    it_ry[it] = r + ky - P;                       // image row = y0 + it_ry
    it_gx[it] = 8 * g - 8;                        // image column = x0 + it_gx
    it_lane[it] = ((lci0 + ci) * HW + it_ry[it] * W + it_gx[it]) * 4;
  }
  float raw[PDX][NIT][8];
  auto load_patch = [&](float (&rw)[NIT][8], const Cur& c) {
    const int sbase = (c.b * Cs * HW + c.y0 * W + c.x0) * 4;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int yy = c.y0 + it_ry[it], xx = c.x0 + it_gx[it];
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const unsigned off = ok ? (unsigned)(sbase + it_lane[it]) : PNSFM_DMA_INVALID;
      // MASKED (W % 8 == 4): the upper 4 pixels of the group may lie past the row end -- they get their own range check
      const unsigned off4 = (MASKED && xx + 4 >= W) ? PNSFM_DMA_INVALID : off + 16u;
#pragma unroll
      for (int u = 0; u < 8; ++u) rw[it][u] = pnsfm_buf_load(xbuf, u < 4 ? off + 4u * u : off4 + 4u * (u - 4), 0);
    }
  };
  auto write_patch = [&](const float (&rw)[NIT][8]) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      pnsfm_u32x4 Hh, Mm, Ll;
      w3_split8(rw[it], Hh, Mm, Ll);
      unsigned char* d = smem + it_lds[it];
      *reinterpret_cast<pnsfm_u32x4*>(d) = Hh;
      *reinterpret_cast<pnsfm_u32x4*>(d + PIECE * 2) = Mm;
      *reinterpret_cast<pnsfm_u32x4*>(d + 2 * PIECE * 2) = Ll;
    }
  };

  // ---- A operand: dY[co0 + l32][8 pixels] of k-step q (row q / SEG, columns 16 * (q % SEG) + 8 * half ..) straight from global
  float araw[RD][8];
  const int a_lane = ((co0 + l32) * HW + 8 * half) * 4;
  auto load_a = [&](float (&dst)[8], const Cur& c, int q) {
    const int yy = c.y0 + q / SEG, xs = c.x0 + 16 * (q % SEG);
    const int xx = xs + 8 * half;
    const bool ok = yy < H && xx < W;
    const unsigned off = ok ? (unsigned)(a_lane + (c.b * a.Cout * HW + yy * W + xs) * 4) : PNSFM_DMA_INVALID;
    const unsigned off4 = (MASKED && xx + 4 >= W) ? PNSFM_DMA_INVALID : off + 16u;
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[u] = pnsfm_buf_load(dybuf, u < 4 ? off + 4u * u : off4 + 4u * (u - 4), 0);
  };

  const unsigned char* const bbase = smem + (size_t)(l32 * CS + 8 + 8 * half) * 2;   // + nt*32*CS*2 + piece + row/col of the k-step

  // one k-step.  7x7 (and the two-ci-tile / three-workgroups-per-CU builds): the B pieces are walked l, m, h -- one 24-element window (prev, cur, next 8-pixel blocks: 12 dwords) in
  // registers at a time instead of all three (the 112 accumulator registers of 7 taps leave no room for 36 + 12: that form
  // spilled) -- and each shifted operand feeds the A pieces it pairs with, so every accumulator still sees its smallest products
  // first: (h,l) | (m,m) (h,m) | (l,h) (m,h) (h,h) [A piece, B piece]; +4.5 % on the two 7x7 layers.  3x3 / 5x5 keep all three
  // windows and finish one tap (6 MFMAs on one accumulator) at a time: the piece-major form measured 4-8 % slower on 5x5.
  constexpr int abl = PNSFM_WG_ABL(a);
  auto kstep = [&](const pnsfm_u32x4 (&A)[3], int q) {
    const int koff = ((q / SEG) * RS + 16 * (q % SEG)) * 2;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (KS >= 7 || NT == 2 || OCC == 3) {      // the register-tight instantiations
#pragma unroll
        for (int s = 2; s >= 0; --s) {
          unsigned Wd[12];
          const unsigned char* p = bbase + (size_t)(nt * 32 * CS + s * PIECE) * 2 + koff;
          const pnsfm_u32x4 c = *reinterpret_cast<const pnsfm_u32x4*>(p);
#pragma unroll
          for (int d = 0; d < 4; ++d) Wd[4 + d] = c[d];
          if (KS > 1) {
            if (abl & 2) {
#pragma unroll
              for (int d = 0; d < 4; ++d) { Wd[d] = c[(d + 1) & 3]; Wd[8 + d] = c[(d + 2) & 3]; }
            } else {
            const pnsfm_u32x4 pv = *reinterpret_cast<const pnsfm_u32x4*>(p - 16);
            const pnsfm_u32x4 nx = *reinterpret_cast<const pnsfm_u32x4*>(p + 16);
#pragma unroll
            for (int d = 0; d < 4; ++d) { Wd[d] = pv[d]; Wd[8 + d] = nx[d]; }
            }
          }
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const int sh = kx - P;                         // element shift of this tap
            pnsfm_u32x4 Bv;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              if ((sh & 1) == 0) Bv[d] = Wd[4 + d + sh / 2];
              else {
                const int lo = 4 + d + (sh - 1) / 2;     // (sh - 1) is even: exact division also for negative shifts
                Bv[d] = (abl & 4) ? (Wd[lo + 1] ^ Wd[lo]) : w3_alignbit16(Wd[lo + 1], Wd[lo]);
              }
            }
            if (abl & 16) { acc[nt][kx][0] += __builtin_bit_cast(float, Bv[0] ^ Bv[1] ^ Bv[2] ^ Bv[3] ^ A[0][0] ^ A[1][1] ^ A[2][2]); continue; }
#pragma unroll
            for (int sa = 2 - s; sa >= 0; --sa) acc[nt][kx] = pnsfm_mfma_bf16(A[sa], Bv, acc[nt][kx]);
          }
        }
      } else {
        // window of 24 elements (prev, cur, next 8-pixel blocks) of each piece: 12 dwords
        unsigned Wd[3][12];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const unsigned char* p = bbase + (size_t)(nt * 32 * CS + s * PIECE) * 2 + koff;
          const pnsfm_u32x4 c = *reinterpret_cast<const pnsfm_u32x4*>(p);
#pragma unroll
          for (int d = 0; d < 4; ++d) Wd[s][4 + d] = c[d];
          if (KS > 1) {
            if (abl & 2) {
#pragma unroll
              for (int d = 0; d < 4; ++d) { Wd[s][d] = c[(d + 1) & 3]; Wd[s][8 + d] = c[(d + 2) & 3]; }
            } else {
            const pnsfm_u32x4 pv = *reinterpret_cast<const pnsfm_u32x4*>(p - 16);
            const pnsfm_u32x4 nx = *reinterpret_cast<const pnsfm_u32x4*>(p + 16);
#pragma unroll
            for (int d = 0; d < 4; ++d) { Wd[s][d] = pv[d]; Wd[s][8 + d] = nx[d]; }
            }
          }
        }
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int sh = kx - P;                         // element shift of this tap
          pnsfm_u32x4 Bv[3];
#pragma unroll
          for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              if ((sh & 1) == 0) Bv[s][d] = Wd[s][4 + d + sh / 2];
              else {
                const int lo = 4 + d + (sh - 1) / 2;     // (sh - 1) is even: exact division also for negative shifts
                Bv[s][d] = (abl & 4) ? (Wd[s][lo + 1] ^ Wd[s][lo]) : w3_alignbit16(Wd[s][lo + 1], Wd[s][lo]);
              }
            }
          if (abl & 16) { acc[nt][kx][0] += __builtin_bit_cast(float, Bv[0][0] ^ Bv[1][1] ^ Bv[2][2] ^ Bv[0][3] ^ Bv[1][2] ^ Bv[2][0] ^ A[0][0] ^ A[1][1] ^ A[2][2]); continue; }
          // smallest terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
          acc[nt][kx] = pnsfm_mfma_bf16(A[2], Bv[0], acc[nt][kx]);
          acc[nt][kx] = pnsfm_mfma_bf16(A[0], Bv[2], acc[nt][kx]);
          acc[nt][kx] = pnsfm_mfma_bf16(A[1], Bv[1], acc[nt][kx]);
          acc[nt][kx] = pnsfm_mfma_bf16(A[1], Bv[0], acc[nt][kx]);
          acc[nt][kx] = pnsfm_mfma_bf16(A[0], Bv[1], acc[nt][kx]);
          acc[nt][kx] = pnsfm_mfma_bf16(A[0], Bv[0], acc[nt][kx]);
        }
      }
    }
  };

  // ---- main loop over this split's pixel tiles, unrolled by two so that ring slots are compile-time:
  //   patch: raw[par % PDX] holds tile t's rows; once written to LDS it is refilled with tile t + PDX;
  //   dY:    k-step i of the tile with parity par sits in ring slot (par * KPW + i) % RD and is replaced, as soon as it has been
  //          split, by the fragment RD k-steps further down this wave's sequence (same tile, next tile or the one after).
#pragma unroll
  for (int p = 0; p < PDX; ++p)
    if (t_begin + p < t_end) load_patch(raw[p], cur[p]);
#pragma unroll
  for (int L = 0; L < RD; ++L)
    if (t_begin + L / KPW < t_end) load_a(araw[L], cur[L / KPW], wk + WK * (L % KPW));
  for (int t0 = t_begin; t0 < t_end; t0 += 2) {
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int t = t0 + par;
      if (t < t_end) {
        if (!(abl & 8) || t == t_begin) {
        __syncthreads();           // every wave is done with the previous tile's patch
        write_patch(raw[par % PDX]);
        __syncthreads();
        if (t + PDX < t_end) load_patch(raw[par % PDX], cur[PDX]);
        }
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          const int slot = (par * KPW + i) % RD;
          pnsfm_u32x4 A[3];
          if (abl & 1) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              A[0][d] = __builtin_bit_cast(unsigned, araw[slot][d]);
              A[1][d] = __builtin_bit_cast(unsigned, araw[slot][4 + d]);
              A[2][d] = __builtin_bit_cast(unsigned, araw[slot][(d + 2) & 7]);
            }
          } else
          w3_split8(araw[slot], A[0], A[1], A[2]);
          if (do_bias) {
#pragma unroll
            for (int u = 0; u < 8; ++u) bsum += araw[slot][u];
          }
          const int dt = (i + RD) / KPW, ni = (i + RD) % KPW;       // dt <= 2
          if (t + dt < t_end && !(abl & 32)) load_a(araw[slot], cur[dt], wk + WK * ni);
          kstep(A, wk + WK * i);
        }
        cur[0] = cur[1]; cur[1] = cur[2]; advance(cur[2]);
      }
    }
  }

  // ---- epilogue.  The WK pixel shares of the workgroup are summed through LDS (one tap at a time: (WK-1)*WM*NT tiles of
  // 4 KB), then every output element has exactly ONE writer in this launch: plain stores, no atomics.
  // D row = (r&3) + 8*(r>>2) + 4*half -> co, col = l32 -> ci
  if (do_bias) bsum += __shfl_xor(bsum, 32);
  if (WK > 1) {
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();           // the patch is dead
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      if (wk > 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((((wk - 1) * WM + wm) * NT + nt) * 16 + r) * 64 + lane] = acc[nt][kx][r];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int j = 1; j < WK; ++j)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][kx][r] += red[((((j - 1) * WM + wm) * NT + nt) * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
    if (do_bias) {
      if (wk > 0 && half == 0) red[((wk - 1) * WM + wm) * 32 + l32] = bsum;
      __syncthreads();
      if (wk == 0)
        for (int j = 1; j < WK; ++j) bsum += red[((j - 1) * WM + wm) * 32 + l32];
    }
  }
  if (wk == 0) {
    if (a.ws == nullptr) {      // the only pixel split: reference layout directly
      const size_t N = (size_t)a.Cin * KK;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ci = ci0 + nt * 32 + l32;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < a.Cout && ci < a.Cin) a.dw[(size_t)co * N + (size_t)ci * KK + ky * KS + kx] = acc[nt][kx][r];
          }
      }
      if (do_bias && half == 0 && co0 + l32 < a.Cout) a.dbias[co0 + l32] = bsum;
    } else {                    // partial sums, ci fastest: 128 contiguous bytes per half-wave
      float* w = a.ws + ((size_t)bz * KS + ky) * a.COP * KS * a.CIP;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int ci = ci0 + nt * 32 + l32;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            w[((size_t)co * KS + kx) * a.CIP + ci] = acc[nt][kx][r];
          }
      }
      if (do_bias && half == 0) a.ws_bias[(size_t)bz * a.COP + co0 + l32] = bsum;
    }
  }
}

// second stage of a pixel-split launch: dW[co][ci][ky][kx] = sum over the Z partial tensors, in a fixed order (deterministic).
// The partials are [z][ky][COP][kx][CIP] with ci fastest: an item = (tap, 4-channel column) is one 16-byte load per partial.
// Round 6: a block owns one co and `ncol` columns (items = KK * ncol <= 256) and its 256 threads are ZG = 256 / items GROUPS that share
// the z range -- group g adds partials g, g + ZG, ... (all loads of a thread independent: <= 8 per thread where the shape allows) --
// and the groups' sums meet in LDS in ascending group order; the result goes through an LDS tile [ci][tap] so that dW leaves as one
// contiguous (ci, tap) run per block.  History: round 2 128-byte runs with 4-byte loads (0.72 ms per step); round 3-5 one thread per
// item walking ALL z with eight loads in flight and 64-256 blocks per launch: latency-bound, 21 us for the four 64 -> 64 @ 96x320
// layers (Z = 69), 77 us for 129 -> 64 @ 192x640 (Z = 237), 0.61 ms per step (profiles/r05_step_breakdown.txt).
__global__ void __launch_bounds__(256) wgrad3_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ ws_bias,
                                                            float* __restrict__ dw, float* __restrict__ dbias, int Z, int KS,
                                                            int COP, int CIP, int Cin, int Cout, int ncol, int ZG) {
  __shared__ float4 part[256];
  __shared__ float tile[4 * 256 + 4];
  const int KK = KS * KS, items = KK * ncol, CIB = 4 * ncol;
  const int co = blockIdx.x, ci0 = blockIdx.y * CIB;
  const int t = threadIdx.x;
  const int zg = t / items, item = t - zg * items;
  const int tap = item / ncol, j = item - tap * ncol;
  const int ci = ci0 + 4 * j;
  const size_t zstride = (size_t)KS * COP * KS * CIP;
  const bool live = zg < ZG && ci < CIP;
  if (live) {
    const int ky = tap / KS, kx = tap - ky * KS;
    const float* p = ws + (((size_t)ky * COP + co) * KS + kx) * CIP + ci;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = zg;
    for (; z + 3 * ZG < Z; z += 4 * ZG) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(z + u * ZG) * zstride);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += v[u].x; s1 += v[u].y; s2 += v[u].z; s3 += v[u].w; }
    }
    for (; z < Z; z += ZG) {
      const float4 a = *reinterpret_cast<const float4*>(p + (size_t)z * zstride);
      s0 += a.x; s1 += a.y; s2 += a.z; s3 += a.w;
    }
    part[t] = make_float4(s0, s1, s2, s3);
  }
  __syncthreads();
  if (zg == 0 && ci < CIP) {
    float4 s = part[item];
    for (int g = 1; g < ZG; ++g) { const float4 q = part[g * items + item]; s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w; }
    float* tl = tile + (4 * j) * KK + tap;
    tl[0] = s.x; tl[KK] = s.y; tl[2 * KK] = s.z; tl[3 * KK] = s.w;
  }
  __syncthreads();
  int nci = Cin - ci0;
  if (nci > CIB) nci = CIB;
  float* out = dw + ((size_t)co * Cin + ci0) * KK;
  for (int e = t; e < nci * KK; e += 256) out[e] = tile[e];
  if (dbias && blockIdx.y == 0 && t < 64) {       // the bias partials: 64 lanes share the z range, a fixed shuffle tree
    float sum = 0.f;
    for (int z = t; z < Z; z += 64) sum += ws_bias[(size_t)z * COP + co];
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if (t == 0) dbias[co] = sum;
  }
}

// the second stage on its own (conv2d_wgrad4.hip writes the same partial-tensor layout)
int launch_wgrad3_reduce(const float* ws, const float* ws_bias, float* dw, float* dbias, int Z, int KS, int COP, int CIP, int Cin,
                         int Cout, hipStream_t s) {
  // z groups so that a thread adds <= ~8 partials; columns per block from what is left of the 256 threads (1x1 layers: <= 32 columns)
  const int KK = KS * KS;
  int zg_want = ceil_div(Z, 8);
  if (zg_want > 256 / KK) zg_want = 256 / KK;
  if (zg_want < 1) zg_want = 1;
  int ncol = (256 / zg_want) / KK;
  if (ncol < 1) ncol = 1;
  if (ncol > 32) ncol = 32;
  if (ncol > ceil_div(CIP, 4)) ncol = ceil_div(CIP, 4);
  int ZG = 256 / (KK * ncol);
  if (ZG > Z) ZG = Z;
  PNSFM_LAUNCH(wgrad3_reduce_kernel, dim3(Cout, ceil_div(Cin, 4 * ncol)), dim3(256), 0, s, ws, ws_bias, dw, dbias, Z, KS, COP, CIP,
               Cin, Cout, ncol, ZG);
  return check_launch("conv2d_backward_weight (split-bf16, reduction)");
}

bool wgrad3_supported(int Cin, int Cout, int H, int W, int ks) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) return false;
  return W % 4 == 0 && Cin >= 16 && Cout >= 16 && H >= 1;      // rows of 4-pixel groups (16-byte aligned)
}
bool wgrad3_fits(int B, int Cin, int Cout, int H, int W) {      // one buffer descriptor per tensor, 31-bit byte offsets
  return (size_t)B * Cin * H * W * 4 < (1ull << 31) && (size_t)B * Cout * H * W * 4 < (1ull << 31);
}
// tile width: 32 columns unless 16 wastes fewer (W = 40: 64 vs 48 columns per row) or the rows need per-element masking
static int wgrad3_tc(int W) {
  if (W % 8 != 0) return 16;
  return round_up(W, 16) < round_up(W, 32) ? 16 : 32;
}
// co tiles per workgroup: the most the layer can fill (4, 2, 1 for Cout >= 97, >= 33, smaller) unless the caller asks for fewer
// -- fewer co tiles = more pixel shares per workgroup (WK = 4 / WM) and more workgroups: parallelism for the low-resolution
// layers WITHOUT a pixel split (no partial tensors, no second kernel)
int wgrad3_WM(int Cout, int want) {
  want &= 7;
  const int most = Cout > 96 ? 4 : (Cout > 32 ? 2 : 1);
  return (want == 1 || want == 2 || want == 4) && want < most ? want : most;
}
int wgrad3_total_tiles(int B, int H, int W) { return B * ceil_div(W, wgrad3_tc(W)) * ceil_div(H, 4); }
int wgrad3_base_blocks(int Cin, int Cout, int ks, int NT, int WM) {
  return ceil_div(Cin, 32 * NT) * ks * ceil_div(ceil_div(Cout, 32), wgrad3_WM(Cout, WM));
}
bool wgrad3_nt2_ok(int Cin, int ks) { return ks <= 3 && Cin > 32; }

template <int KS, int NT, int WM, int TC, bool MASKED, int OCC = 2>
static int launch_wgrad3(const Wgrad3Args& a, dim3 grid, hipStream_t s) {
  using Gm = Wgrad3Geom<KS, NT, WM, TC, OCC>;
#ifndef PNSFM_EMU
  static unsigned long long raised = 0;      // one bit per device
  if (Gm::SMEM > 64 * 1024 &&
      ensure_lds_limit(reinterpret_cast<const void*>(&conv2d_wgrad3_kernel<KS, NT, WM, TC, MASKED, OCC>), &raised, 160 * 1024,
                       "conv2d_backward_weight"))
    return -1;
#endif
  PNSFM_LAUNCH((conv2d_wgrad3_kernel<KS, NT, WM, TC, MASKED, OCC>), grid, dim3(256), (size_t)Gm::SMEM, s, a);
  return check_launch("conv2d_backward_weight (split-bf16)");
}

int enqueue_wgrad3(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W, int ks,
                   int split, int NT, int WMwant, hipStream_t s, const ConvSrc* ms) {
  if (!wgrad3_supported(Cin, Cout, H, W, ks)) { set_error("conv2d_backward_weight (split-bf16): unsupported shape"); return -1; }
  if ((size_t)B * Cin * H * W * 4 >= (1ull << 31) || (size_t)B * Cout * H * W * 4 >= (1ull << 31)) {
    set_error("conv2d_backward_weight (split-bf16): tensor too large for 32-bit buffer offsets");
    return -1;
  }
  if (NT != 2 || !wgrad3_nt2_ok(Cin, ks)) NT = 1;
  if (ms && NT == 2 && !conv_src_aligned(*ms, Cin, 64)) NT = 1;     // a 64-channel tile would straddle two tensors
  // two instantiations do not fit their register budget without spilling (3x3, four co tiles per workgroup, 32-column tiles: two
  // ci tiles per wave, and the three-workgroups-per-CU build): those requests run the one-ci-tile / two-workgroup build instead
  const bool tight = ks == 3 && wgrad3_WM(Cout, WMwant & 7) == 4 && wgrad3_tc(W) == 32 && W % 8 == 0;
  if (tight) { NT = 1; WMwant &= 7; }
  if (ms && !conv_src_aligned(*ms, Cin, 32)) {
    set_error("conv2d_backward_weight (split-bf16): the input tensors must end on 32-channel boundaries");
    return -1;
  }
  Wgrad3Args a;
  a.x = x; a.dy = dy; a.dw = dw; a.dbias = dbias;
  a.x1 = ms ? ms->x1 : nullptr; a.x2 = ms ? ms->x2 : nullptr;
  a.C0 = ms ? ms->C0 : Cin; a.C01 = ms ? ms->C0 + ms->C1 : Cin;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  const int tc = wgrad3_tc(W);
  const bool masked = W % 8 != 0;
  a.tiles_x = ceil_div(W, tc);
  a.tiles_per_img = a.tiles_x * ceil_div(H, 4);
  a.total_tiles = B * a.tiles_per_img;
  if (split < 1) split = 1;
  if (split > a.total_tiles) split = a.total_tiles;
  a.tiles_per_split = ceil_div(a.total_tiles, split);
  const int splitP = ceil_div(a.total_tiles, a.tiles_per_split);
  a.ci_tiles = ceil_div(Cin, 32 * NT);
  const bool occ3 = (WMwant & 8) != 0 && ks == 3 && NT == 1 && !masked;      // WM | 8: the three-workgroups-per-CU build
  WMwant &= 7;
  const int WM = wgrad3_WM(Cout, WMwant);
  const int co_groups = ceil_div(ceil_div(Cout, 32), WM);
  a.COP = co_groups * WM * 32;
  a.CIP = a.ci_tiles * 32 * NT;
  a.ws = nullptr; a.ws_bias = nullptr;
  // pixel-split launch: partial tensors in the stream's scratch buffer (api.hip), summed by wgrad3_reduce_kernel
  const size_t part = (size_t)ks * a.COP * ks * a.CIP;
  ScratchLease lease(s, splitP > 1 ? ((size_t)splitP * (part + a.COP)) * sizeof(float) : 0);
  if (splitP > 1) {
    if (!lease.p) return -1;
    a.ws = lease.as<float>();
    a.ws_bias = a.ws + (size_t)splitP * part;
  }
  a.gx = a.ci_tiles * ks; a.gy = co_groups; a.bmap = block_map_mode();
  dim3 grid(a.ci_tiles * ks * co_groups * splitP);
  int rc = 0;
#define PNSFM_W3T(KSv, NTv, WMv)                                                    \
  do {                                                                              \
    if (masked) rc = launch_wgrad3<KSv, NTv, WMv, 16, true>(a, grid, s);            \
    else if (tc == 16) rc = launch_wgrad3<KSv, NTv, WMv, 16, false>(a, grid, s);    \
    else rc = launch_wgrad3<KSv, NTv, WMv, 32, false>(a, grid, s);                  \
  } while (0)
#define PNSFM_W3(KSv, NTv)                                                \
  do {                                                                    \
    if (WM == 4) PNSFM_W3T(KSv, NTv, 4);                                  \
    else if (WM == 2) PNSFM_W3T(KSv, NTv, 2);                             \
    else PNSFM_W3T(KSv, NTv, 1);                                          \
  } while (0)
  if (occ3) {
    if (tc == 16) { if (WM == 4) rc = launch_wgrad3<3, 1, 4, 16, false, 3>(a, grid, s); else if (WM == 2) rc = launch_wgrad3<3, 1, 2, 16, false, 3>(a, grid, s); else rc = launch_wgrad3<3, 1, 1, 16, false, 3>(a, grid, s); }
    else { if (WM == 4) { set_error("conv2d_backward_weight (split-bf16): unreachable build"); rc = -1; } else if (WM == 2) rc = launch_wgrad3<3, 1, 2, 32, false, 3>(a, grid, s); else rc = launch_wgrad3<3, 1, 1, 32, false, 3>(a, grid, s); }
  }
  else if (ks == 1) { if (NT == 2) PNSFM_W3(1, 2); else PNSFM_W3(1, 1); }
  else if (ks == 3 && NT == 2 && WM == 4) {      // (the 32-column tile of this shape was re-routed above)
    if (masked) rc = launch_wgrad3<3, 2, 4, 16, true>(a, grid, s);
    else if (tc == 16) rc = launch_wgrad3<3, 2, 4, 16, false>(a, grid, s);
    else { set_error("conv2d_backward_weight (split-bf16): unreachable build"); rc = -1; }
  }
  else if (ks == 3) { if (NT == 2) { if (WM == 2) PNSFM_W3T(3, 2, 2); else PNSFM_W3T(3, 2, 1); } else PNSFM_W3(3, 1); }
  else if (ks == 5) PNSFM_W3(5, 1);
  else PNSFM_W3(7, 1);
#undef PNSFM_W3T
#undef PNSFM_W3
  if (a.ws) {
    if (!rc) rc = launch_wgrad3_reduce(a.ws, a.ws_bias, dw, dbias, splitP, ks, a.COP, a.CIP, Cin, Cout, s);
  }
  return rc;
}

}  // namespace pnsfm
