// conv2d_wgrad4.hip -- weight gradient of a stride-1 3x3 convolution, ALL NINE TAPS per workgroup (split-bf16 arithmetic).
//
// Same contract and arithmetic as conv2d_wgrad3.hip (the autograd weight gradient of nn.Conv2d in the reference's Conv2D /
// ResidualConv / Pack / Unpack blocks, /root/reference/packnet_sfm/networks/layers/packnet/layers01.py:28-36, 57-60, 235-246,
// 274-281):   dW[co][ci][ky][kx] = sum_{b, y, x} dY[b][co][y][x] * X[b][ci][y + ky - 1][x + kx - 1]
// every fp32 operand split EXACTLY into three bf16 pieces, the product rebuilt from the 6 piece products with fp32 accumulation.
//
// Why a second kernel.  wgrad3 gives a workgroup ONE kernel row: a 3x3 layer re-reads and re-splits every dY fragment three
// times (once per row) and spends 44 VALU operations of splitting on 18 MFMAs -- its 3x3 launches run at ~120 TFLOP/s where
// the 7x7 ones (42 MFMAs per split) reach 200.  Nine taps of a 32 x 32 tile are 144 accumulator registers, too many next to
// the prefetch registers, so this kernel uses the 16x16x32 MFMA (4 accumulator registers per tile, same MAC rate):
//   * a wave owns 32 co x 16 ci x 9 taps = 18 accumulator tiles (72 registers); one k-step = 32 pixels = 4 groups of 8
//     consecutive pixels of an image row: 2 dY fragments (split once, 88 VALU) feed 108 MFMAs;
//   * the pixel tile is TR rows x TG groups, groups numbered row-major -- a k-step's 4 groups may sit in different rows, so
//     tiles 3 or 5 groups wide (W = 20, 40, 80, ...) waste nothing but the masked half group of W % 8 == 4;
//   * A operand (dY): lane (co = l&15 [+16], group = l>>4) reads 32 contiguous bytes of the NCHW tensor straight from global
//     memory, one or two k-steps ahead; the bias gradient is the running sum of the same registers;
//   * B operand (X): the patch (TR + 2 rows, 8 columns of halo either side) is staged like wgrad3's: fp32 rows -> registers
//     (issued before the previous tile's MFMAs) -> 3 bf16 pieces -> LDS [piece][ci][row][col], channel stride 8 * odd.  A lane
//     reads its aligned 8-pixel block once per (kernel row, piece) plus the two neighbouring dwords and builds the three
//     shifted operands with one v_alignbit_b32 per dword;
//   * the 4 waves of a workgroup are WCO co tiles x WCI ci tiles (no pixel shares: nothing to reduce inside a workgroup);
//     pixel tiles are split over the launch's third logical dimension and the partial tensors of a split launch go to the stream's scratch buffer in
//     wgrad3's layout ([split][ky][co][kx][ci]) for wgrad3_reduce_kernel: no atomics, bit-reproducible.
// Roofline: MFMA-bound: 2*Cout*Cin*9*B*H*W algorithmic flop against 2500/6 TFLOP/s (bf16 dense peak / 6 products).
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

struct Wgrad4Args {
  const float* x1;   // multi-source input (ConvSrc, pnsfm_common.h): channels [C0, C01) live in x1, [C01, Cin) in x2
  const float* x2;
  int C0, C01;       // C0 = C01 = Cin for a single source
  const float* x;    // [B][Cin][H][W]  (multi-source: [B][C0][H][W])
  const float* dy;   // [B][Cout][H][W]
  float* dw;         // [Cout][Cin][3][3]   written directly when the launch has ONE pixel split ...
  float* dbias;      // [Cout] or null
  float* ws;         // ... else partial sums [split][ky][COP][kx][CIP] (+ [split][COP] bias partials at ws_bias)
  float* ws_bias;
  int COP, CIP;      // padded channel extents of the workspace (whole workgroup tiles)
  int B, Cin, Cout, H, W;
  int tiles_x, tiles_per_img, total_tiles, tiles_per_split;
  int gx, gy, bmap;  // 1-D launch: ci tiles, co groups, block order (pnsfm_common.h: block_map_mode)
};
// what-if builds (tools/r6/wgrad_ablate.py; results wrong by construction): -DPNSFM_WG_ABLATE=<mask>, a COMPILE-TIME constant (a run-time
// switch changed hipcc's register allocation: the 7x7 build ran 2x slower with every switch off) -- 1 no dY split, 2 no neighbour LDS
// reads, 4 no shifted operands, 8 patch staged for the first tile only, 16 no MFMAs, 32 dY loaded once
#ifdef PNSFM_WG_ABLATE
#define PNSFM_WG_ABL(a) (PNSFM_WG_ABLATE)
#else
#define PNSFM_WG_ABL(a) 0
#endif

#ifdef PNSFM_EMU
static inline unsigned w4_alignbit16(unsigned hi, unsigned lo) { return (lo >> 16) | (hi << 16); }
#else
__device__ __forceinline__ unsigned w4_alignbit16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
#endif

// 8 consecutive fp32 values -> three 16-byte bf16 pieces (conv2d_bx3.h: exact, round-to-nearest pieces)
__device__ __forceinline__ void w4_split8(const float (&v)[8], pnsfm_u32x4& H, pnsfm_u32x4& M, pnsfm_u32x4& L) {
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

template <int WCI, int TG, int TR>
struct Wgrad4Geom {
  static constexpr int WCO = 4 / WCI;                    // co tiles (32 channels) x ci tiles (16 channels) of the 4 waves
  static constexpr int NCI = 16 * WCI, NCO = 32 * WCO;
  static constexpr int TC = 8 * TG;                      // tile columns
  static constexpr int PR = TR + 2;                      // patch rows (one halo row either side)
  static constexpr int RS = TC + 16;                     // patch row: 8 halo + TC + 8 halo elements
  static constexpr int CS0 = PR * RS;
  static constexpr int CS = ((CS0 / 8) & 1) ? CS0 : CS0 + 8;      // channel stride (elements) = 8 * odd: conflict-free b128
  static constexpr int PIECE = NCI * CS;                 // elements of one piece plane
  static constexpr int SMEM = 3 * PIECE * 2;             // bytes
  static constexpr int IPC = PR * (RS / 8);              // (row, 8-column group) items per channel
  static constexpr int ITEMS = NCI * IPC;
  static constexpr int NIT = (ITEMS + 255) / 256;        // per thread
  static constexpr int G = TR * TG;                      // 8-pixel groups of a tile
  static constexpr int KSTEPS = (G + 3) / 4;             // 4 groups = 32 pixels per MFMA k-step
  // dY k-steps in flight (16 registers each): two where the ring slots stay compile-time without unrolling the tile loop -- the
  // 4-group tiles of the high-resolution layers, whose dY comes from HBM; the low-resolution ones are L2-resident
  static constexpr int RD = (KSTEPS % 2 == 0) ? 2 : 1;
};

// MASKED: W % 8 == 4 -- the upper half of an 8-pixel group may lie past the end of an image row: the two 4-pixel halves are
// range-checked separately (an out-of-row half gets an out-of-range buffer offset and reads as zero)
template <int WCI, int TG, int TR, bool MASKED>
__global__ void __launch_bounds__(256, 2) conv2d_wgrad4_kernel(Wgrad4Args a) {
  using Gm = Wgrad4Geom<WCI, TG, TR>;
  constexpr int RD = Gm::RD;
  constexpr int WCO = Gm::WCO, NCI = Gm::NCI, TC = Gm::TC, PR = Gm::PR, RS = Gm::RS, CS = Gm::CS, PIECE = Gm::PIECE;
  constexpr int IPC = Gm::IPC, ITEMS = Gm::ITEMS, NIT = Gm::NIT, G = Gm::G, KSTEPS = Gm::KSTEPS;
  PNSFM_DYN_SMEM(unsigned char, smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), l16 = lane & 15, j = lane >> 4;
  const int wci = wave % WCI, wco = wave / WCI;
  const int H = a.H, W = a.W, HW = H * W;
  // logical block (ci tile, co group, pixel split): ci tile fastest -- the workgroups of one pixel split read the same dY and X
  // -- and a contiguous range of that order per XCD (pnsfm_common.h)
  const unsigned Lb = a.bmap == 2 ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
  const int bx = (int)(Lb % (unsigned)a.gx), by = (int)((Lb / (unsigned)a.gx) % (unsigned)a.gy), bz = (int)(Lb / (unsigned)(a.gx * a.gy));
  const int ci0 = bx * NCI;
  const int co0 = (by * WCO + wco) * 32;
  const int t_begin = bz * a.tiles_per_split;
  int t_end = t_begin + a.tiles_per_split;
  if (t_end > a.total_tiles) t_end = a.total_tiles;

  f32x4 acc[2][3][3];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][ky][kx][r] = 0.f;
  float bsum0 = 0.f, bsum1 = 0.f;
  const bool do_bias = a.dbias != nullptr && bx == 0 && wci == 0;

  // ---- tile cursors (as wgrad3): origin of the current tile and the two behind it, advanced in scalar registers
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
    int e = it * 256 + tid;
    const bool live = e < ITEMS;
    if (!live) e = ITEMS - 1;
    const int ci = e / IPC;
    const int rem = e - ci * IPC;
    const int r = rem / (RS / 8), g = rem - r * (RS / 8);
    it_lds[it] = live ? (ci * CS + r * RS + 8 * g) * 2 : -1;
    it_ry[it] = r - 1;                            // image row = y0 + it_ry
    it_gx[it] = live ? 8 * g - 8 : (1 << 24);     // image column = x0 + it_gx (a dead item is out of every image)
    it_lane[it] = ((lci0 + ci) * HW + it_ry[it] * W + 8 * g - 8) * 4;
  }
  float raw[NIT][8];
  auto load_patch = [&](const Cur& c) {
    const int sbase = (c.b * Cs * HW + c.y0 * W + c.x0) * 4;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int yy = c.y0 + it_ry[it], xx = c.x0 + it_gx[it];
      const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const unsigned off = ok ? (unsigned)(sbase + it_lane[it]) : PNSFM_DMA_INVALID;
      const unsigned off4 = (MASKED && xx + 4 >= W) ? PNSFM_DMA_INVALID : off + 16u;
#pragma unroll
      for (int u = 0; u < 8; ++u) raw[it][u] = pnsfm_buf_load(xbuf, u < 4 ? off + 4u * u : off4 + 4u * (u - 4), 0);
    }
  };
  auto write_patch = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (ITEMS % 256 != 0 && it == NIT - 1 && it_lds[it] < 0) continue;
      pnsfm_u32x4 Hh, Mm, Ll;
      w4_split8(raw[it], Hh, Mm, Ll);
      unsigned char* d = smem + it_lds[it];
      *reinterpret_cast<pnsfm_u32x4*>(d) = Hh;
      *reinterpret_cast<pnsfm_u32x4*>(d + PIECE * 2) = Mm;
      *reinterpret_cast<pnsfm_u32x4*>(d + 2 * PIECE * 2) = Ll;
    }
  };

  // ---- k-step geometry of this lane: group 4q + j of the tile (row-major) -> tile row / column; groups past the tile's last
  // one (G % 4 != 0) get a column outside every image (dY reads as zero) and the LDS address of group 0
  int a_row[KSTEPS], a_col[KSTEPS];
#pragma unroll
  for (int q = 0; q < KSTEPS; ++q) {
    const int g = 4 * q + j;
    const int row = g / TG;
    a_row[q] = g < G ? row : 0;
    a_col[q] = g < G ? 8 * (g - row * TG) : (1 << 24);
  }
  // A operand: dY[co0 + 16 s + l16][8 pixels of the lane's group] straight from global memory
  float araw[RD][2][8];
  const int a_lane = (co0 + l16) * HW * 4;
  auto load_a = [&](float (&araw)[2][8], const Cur& c, int q) {
    const int yy = c.y0 + a_row[q], xx = c.x0 + a_col[q];
    const bool ok = yy < H && xx < W;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned off = ok ? (unsigned)(a_lane + s * 16 * HW * 4 + (c.b * a.Cout * HW + yy * W + xx) * 4) : PNSFM_DMA_INVALID;
      const unsigned off4 = (MASKED && xx + 4 >= W) ? PNSFM_DMA_INVALID : off + 16u;
#pragma unroll
      for (int u = 0; u < 8; ++u) araw[s][u] = pnsfm_buf_load(dybuf, u < 4 ? off + 4u * u : off4 + 4u * (u - 4), 0);
    }
  };

  const unsigned char* const bbase = smem + (size_t)((wci * 16 + l16) * CS + 8) * 2;

  // one k-step: pieces of B in the order l, m, h so that every accumulator sees its smallest products first:
  // (h,l) | (m,m) (h,m) | (l,h) (m,h) (h,h)   [A piece, B piece]
  constexpr int abl = PNSFM_WG_ABL(a);
  auto kstep = [&](const pnsfm_u32x4 (&A)[2][3], int q) {
    const int boff = (a_row[q] * RS + (a_col[q] & 0xffff)) * 2;
#pragma unroll
    for (int sb = 2; sb >= 0; --sb) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const unsigned char* p = bbase + boff + (ky * RS + sb * PIECE) * 2;
        const pnsfm_u32x4 c = *reinterpret_cast<const pnsfm_u32x4*>(p);
        const unsigned pv = (abl & 2) ? c[1] : *reinterpret_cast<const unsigned*>(p - 4);      // elements -2, -1
        const unsigned nx = (abl & 2) ? c[2] : *reinterpret_cast<const unsigned*>(p + 16);     // elements 8, 9
        pnsfm_u32x4 Bt[3];
        if (abl & 4) { Bt[0] = c; Bt[2] = c; Bt[0][0] ^= pv; Bt[2][3] ^= nx; }
        else {
        Bt[0][0] = w4_alignbit16(c[0], pv);   Bt[0][1] = w4_alignbit16(c[1], c[0]);
        Bt[0][2] = w4_alignbit16(c[2], c[1]); Bt[0][3] = w4_alignbit16(c[3], c[2]);
        Bt[2][0] = w4_alignbit16(c[1], c[0]); Bt[2][1] = w4_alignbit16(c[2], c[1]);
        Bt[2][2] = w4_alignbit16(c[3], c[2]); Bt[2][3] = w4_alignbit16(nx, c[3]);
        }
        Bt[1] = c;
        if (abl & 16) { acc[0][ky][0][0] += __builtin_bit_cast(float, Bt[0][0] ^ Bt[2][3] ^ A[0][0][0] ^ A[1][2][3] ^ A[0][1][1] ^ A[1][1][2] ^ A[1][0][0] ^ A[0][2][3]); continue; }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int sa = 2 - sb; sa >= 0; --sa)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[s][ky][kx] = pnsfm_mfma_bf16_16(A[s][sa], Bt[kx], acc[s][ky][kx]);
      }
    }
  };

  // ---- main loop over this split's pixel tiles: the patch of tile t + 1 and the dY fragments of the next RD k-steps are in
  // flight (registers) while the MFMAs of the current one run (RD divides KSTEPS: the ring slots are compile-time)
#pragma unroll
  for (int L = 0; L < RD; ++L)
    if (t_begin < t_end) load_a(araw[L], cur[0], L);
  if (t_begin < t_end) load_patch(cur[0]);
  for (int t = t_begin; t < t_end; ++t) {
    if (!(abl & 8) || t == t_begin) {
    __syncthreads();           // every wave is done with the previous tile's patch
    write_patch();
    __syncthreads();
    if (t + 1 < t_end) load_patch(cur[1]);
    }
#pragma unroll
    for (int q = 0; q < KSTEPS; ++q) {
      const int slot = q % RD;
      pnsfm_u32x4 A[2][3];
      if (abl & 1) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            A[s2][0][d] = __builtin_bit_cast(unsigned, araw[slot][s2][d]);
            A[s2][1][d] = __builtin_bit_cast(unsigned, araw[slot][s2][4 + d]);
            A[s2][2][d] = __builtin_bit_cast(unsigned, araw[slot][s2][(d + 2) & 7]);
          }
      } else {
      w4_split8(araw[slot][0], A[0][0], A[0][1], A[0][2]);
      w4_split8(araw[slot][1], A[1][0], A[1][1], A[1][2]);
      }
      if (do_bias) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { bsum0 += araw[slot][0][u]; bsum1 += araw[slot][1][u]; }
      }
      const int dt = (q + RD) / KSTEPS, nq = (q + RD) % KSTEPS;      // dt <= 1 (RD <= KSTEPS)
      if (t + dt < t_end && !(abl & 32)) load_a(araw[slot], cur[dt], nq);
      kstep(A, q);
    }
    cur[0] = cur[1]; cur[1] = cur[2]; advance(cur[2]);
  }

  // ---- epilogue: D row = 4 * j + r -> co, column = l16 -> ci; every output element has exactly ONE writer in this launch
  const int ci = ci0 + wci * 16 + l16;
  if (a.ws) {
    float* ws = a.ws + (size_t)bz * 9 * a.COP * a.CIP;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * s + 4 * j + r;
            ws[(((size_t)ky * a.COP + co) * 3 + kx) * a.CIP + ci] = acc[s][ky][kx][r];
          }
  } else if (ci < a.Cin) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + 16 * s + 4 * j + r;
        if (co < a.Cout) {
          float* o = a.dw + ((size_t)co * a.Cin + ci) * 9;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) o[ky * 3 + kx] = acc[s][ky][kx][r];
        }
      }
  }
  if (do_bias) {
    bsum0 += __shfl_xor(bsum0, 16); bsum0 += __shfl_xor(bsum0, 32);
    bsum1 += __shfl_xor(bsum1, 16); bsum1 += __shfl_xor(bsum1, 32);
    if (j == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int co = co0 + 16 * s + l16;
        const float v = s ? bsum1 : bsum0;
        if (a.ws_bias) a.ws_bias[(size_t)bz * a.COP + co] = v;
        else if (co < a.Cout) a.dbias[co] = v;
      }
    }
  }
}

// conv2d_wgrad3.hip: sums the partial tensors of a pixel-split launch in a fixed order
int launch_wgrad3_reduce(const float* ws, const float* ws_bias, float* dw, float* dbias, int Z, int KS, int COP, int CIP, int Cin,
                         int Cout, hipStream_t s);

bool wgrad4_supported(int Cin, int Cout, int H, int W, int ks) {
  return ks == 3 && W % 4 == 0 && Cin >= 16 && Cout >= 16 && H >= 1;      // rows of 4-pixel groups (16-byte aligned)
}
// tile width in 8-pixel groups: 3 (W <= 24), else 4 or 5 -- whichever wastes fewer columns (the caller may ask for the other)
int wgrad4_TG(int W, int want) {
  if (W <= 24) return 3;
  if (want == 4 || want == 5) return want;
  return round_up(W, 40) < round_up(W, 32) ? 5 : 4;
}
int wgrad4_TR(int H, int want) { return (want == 6 || (want == 0 && H % 4 != 0 && H % 6 == 0)) ? 6 : 4; }
int wgrad4_total_tiles(int B, int H, int W, int TG, int TR) { return B * ceil_div(W, 8 * TG) * ceil_div(H, TR); }
int wgrad4_base_blocks(int Cin, int Cout, int WCI) { return ceil_div(Cin, 16 * WCI) * ceil_div(Cout, 32 * (4 / WCI)); }

template <int WCI, int TG, int TR, bool MASKED>
static int launch_wgrad4(const Wgrad4Args& a, dim3 grid, hipStream_t s) {
  using Gm = Wgrad4Geom<WCI, TG, TR>;
#ifndef PNSFM_EMU
  static unsigned long long raised = 0;      // one bit per device
  if (Gm::SMEM > 64 * 1024 &&
      ensure_lds_limit(reinterpret_cast<const void*>(&conv2d_wgrad4_kernel<WCI, TG, TR, MASKED>), &raised, 160 * 1024,
                       "conv2d_backward_weight"))
    return -1;
#endif
  PNSFM_LAUNCH((conv2d_wgrad4_kernel<WCI, TG, TR, MASKED>), grid, dim3(256), (size_t)Gm::SMEM, s, a);
  return check_launch("conv2d_backward_weight (split-bf16, nine taps)");
}

// cfg: WCI (1 | 2) | TG << 4 | TR << 8 (0: library choice)
int enqueue_wgrad4(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W, int split,
                   int cfg, hipStream_t s, const ConvSrc* ms) {
  if (!wgrad4_supported(Cin, Cout, H, W, 3)) { set_error("conv2d_backward_weight (nine taps): unsupported shape"); return -1; }
  if ((size_t)B * Cin * H * W * 4 >= (1ull << 31) || (size_t)B * Cout * H * W * 4 >= (1ull << 31)) {
    set_error("conv2d_backward_weight (nine taps): tensor too large for 32-bit buffer offsets");
    return -1;
  }
  if (ms && !conv_src_aligned(*ms, Cin, 32)) {
    set_error("conv2d_backward_weight (nine taps): the input tensors must end on 32-channel boundaries");
    return -1;
  }
  const int WCI = (cfg & 15) == 1 ? 1 : 2;
  const int TG = wgrad4_TG(W, (cfg >> 4) & 15);
  const int TR = TG == 3 ? wgrad4_TR(H, (cfg >> 8) & 15) : 4;      // 6-row tiles exist for the 3-group (W <= 24) tiles only
  const bool masked = W % 8 != 0;
  Wgrad4Args a;
  a.x = x; a.dy = dy; a.dw = dw; a.dbias = dbias;
  a.x1 = ms ? ms->x1 : nullptr; a.x2 = ms ? ms->x2 : nullptr;
  a.C0 = ms ? ms->C0 : Cin; a.C01 = ms ? ms->C0 + ms->C1 : Cin;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.tiles_x = ceil_div(W, 8 * TG);
  a.tiles_per_img = a.tiles_x * ceil_div(H, TR);
  a.total_tiles = B * a.tiles_per_img;
  if (split < 1) split = 1;
  if (split > a.total_tiles) split = a.total_tiles;
  a.tiles_per_split = ceil_div(a.total_tiles, split);
  const int splitP = ceil_div(a.total_tiles, a.tiles_per_split);
  const int ci_tiles = ceil_div(Cin, 16 * WCI), co_groups = ceil_div(Cout, 32 * (4 / WCI));
  a.COP = co_groups * 32 * (4 / WCI);
  a.CIP = ci_tiles * 16 * WCI;
  a.ws = nullptr; a.ws_bias = nullptr;
  const size_t part = (size_t)9 * a.COP * a.CIP;
  ScratchLease lease(s, splitP > 1 ? ((size_t)splitP * (part + a.COP)) * sizeof(float) : 0);
  if (splitP > 1) {
    if (!lease.p) return -1;
    a.ws = lease.as<float>();
    a.ws_bias = a.ws + (size_t)splitP * part;
  }
  a.gx = ci_tiles; a.gy = co_groups; a.bmap = block_map_mode();
  dim3 grid(ci_tiles * co_groups * splitP);
  int rc = 0;
#define PNSFM_W4(WCIv, TGv, TRv)                                                  \
  do {                                                                            \
    if (masked) rc = launch_wgrad4<WCIv, TGv, TRv, true>(a, grid, s);             \
    else rc = launch_wgrad4<WCIv, TGv, TRv, false>(a, grid, s);                   \
  } while (0)
#define PNSFM_W4G(WCIv)                                                           \
  do {                                                                            \
    if (TG == 3 && TR == 6) PNSFM_W4(WCIv, 3, 6);                                 \
    else if (TG == 3) PNSFM_W4(WCIv, 3, 4);                                       \
    else if (TG == 5) PNSFM_W4(WCIv, 5, 4);                                       \
    else PNSFM_W4(WCIv, 4, 4);                                                    \
  } while (0)
  if (WCI == 1) PNSFM_W4G(1); else PNSFM_W4G(2);
#undef PNSFM_W4G
#undef PNSFM_W4
  if (a.ws && !rc) rc = launch_wgrad3_reduce(a.ws, a.ws_bias, dw, dbias, splitP, 3, a.COP, a.CIP, Cin, Cout, s);
  return rc;
}

}  // namespace pnsfm
