// conv2d_wgrad2.hip -- weight gradient of a stride-1 KxK convolution, tap-major formulation (K in {1, 3, 5}).
//
// Replaces the autograd weight gradient of nn.Conv2d in the reference's Conv2D / ResidualConv / Pack / Unpack blocks
//   (/root/reference/packnet_sfm/networks/layers/packnet/layers01.py:28-36, 57-60, 235-246, 274-281) for the layers whose
//   image width is a multiple of 8 -- every PackNet01 layer from 192x640 down to 12x40; the generic kernel in conv2d.hip
//   keeps the rest (7x7, stride 2, 6x20) and the runtime autotuner times both where both apply.
//
//   dW[co][ci][ky][kx] = sum_{b, y, x} dY[b][co][y][x] * X[b][ci][y + ky - P][x + kx - P]
//
// GEMM view PER TAP: M = co, N = ci, K = pixels.  conv2d.hip's kernel makes (ci, tap) the N dimension, so a lane's B
// operand is a gather through an offset table and the dY fragment is re-read for every 32 (ci, tap) columns.  Here a
// wave owns a 32(co) x 32(ci) tile for ALL taps of a 3x3 (or one kernel row of a 5x5): TG accumulator tiles in AGPRs.
//   * per 32-pixel K-segment the 16 dY fragments are read from LDS ONCE and reused by every tap (dY stays in registers);
//   * the X fragment of (tap, k-step) is patch[ci = lane][pixel + tap offset]: the pixel and tap offsets are compile-time
//     immediates of ds_read_b32 (tile geometry is a template parameter), lanes differ only by the channel stride PS, which
//     is odd -> conflict-free; no offset table, no address arithmetic in the loop;
//   * a workgroup = 2 x 2 waves = 64 co x 64 ci sharing one dY tile and one X halo patch of 64 pixels; both are
//     DOUBLE-BUFFERED in LDS and filled by LDS-DMA (global_load_lds), issued in slices between the taps of the previous
//     tile, so staging never stalls the matrix pipe (one wave per SIMD saturates v_mfma_f32_32x32x2_f32: 64 cycles each);
//   * pixel tiles are split over blockIdx.z when (Cout/64)*(Cin/64) workgroups cannot fill 256 CUs; partial dW meet with
//     fp32 atomics in a zero-filled buffer (un-split: plain stores).
// Tile shapes: the 32 pixels of a K-segment are SR rows x FC columns with FC = 32, 16 or 8 (the largest that divides W), so
// 24x80 and 12x40 feature maps tile exactly in x instead of falling back to linear pixel runs with full-width patches.
// Roofline: MFMA-bound, 2*Cout*Cin*K*K*B*H*W flop against 157.3 TFLOP/s; LDS traffic 0.6 ds_read_b32 per MFMA.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

// out-of-image / padded elements are fetched from here (device variables are per translation unit without -fgpu-rdc)
static __device__ __attribute__((aligned(16))) float w2_zero_page[64];

struct Wgrad2Args {
  const float* x;    // [B][Cin][H][W]
  const float* dy;   // [B][Cout][H][W]
  float* dw;         // [Cout][Cin][KS][KS]
  float* dbias;      // [Cout] or null
  int B, Cin, Cout, H, W;
  int tiles_x, tiles_per_img, total_tiles, tiles_per_split, splitP;
  int ci_tiles;      // gridDim.x = ci_tiles * (tap groups: KS for a 5x5, 1 otherwise)
  size_t zstride;    // pixel-split launch: floats between the partial [dw | dbias] slabs of consecutive splits (0: un-split)
};

template <int KS, int FC>
struct Wgrad2Geom {
  static constexpr int P = KS / 2, KK = KS * KS;
  static constexpr int TG = (KS == 3) ? 9 : KS;       // taps a workgroup accumulates: the whole 3x3, else one kernel row
  static constexpr int SR = 32 / FC;                    // rows of a 32-pixel K-segment
  static constexpr int TR = 2 * SR;                     // rows of a 64-pixel tile (two segments stacked)
  static constexpr int PH = TR + KS - 1, PW = FC + KS - 1, PSR = PH * PW;
  static constexpr int PS = PSR | 1;                    // odd channel stride: lanes (= channels) hit distinct banks
  static constexpr int DS = 65;                         // dY row stride (64 pixels + 1)
  static constexpr int BUF = 64 * DS + 64 * PS;         // floats per LDS buffer
  static constexpr int NPI = (PSR + 63) / 64;           // DMA instructions per patch channel
};

template <int KS, int FC>
__global__ void __launch_bounds__(256, 1) conv2d_wgrad2_kernel(Wgrad2Args a) {
  using Gm = Wgrad2Geom<KS, FC>;
  constexpr int P = Gm::P, KK = Gm::KK, TG = Gm::TG, SR = Gm::SR, TR = Gm::TR, PH = Gm::PH, PW = Gm::PW, PSR = Gm::PSR;
  constexpr int PS = Gm::PS, DS = Gm::DS, BUF = Gm::BUF, NPI = Gm::NPI;
  PNSFM_DYN_SMEM(float, smem);

  const int tid = threadIdx.x;
  // `wave` lives in an SGPR so that LDS-DMA destinations and buffer descriptors derived from it are provably uniform
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int H = a.H, W = a.W, HW = H * W;
  const int ci_t = blockIdx.x % a.ci_tiles, tgi = blockIdx.x / a.ci_tiles;
  const int ci0 = ci_t * 64, co0 = blockIdx.y * 64;
  const int ky0 = (KS == 5) ? tgi : 0;                 // kernel row of this workgroup (5x5 only)

  f32x16 acc[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int t_begin = blockIdx.z * a.tiles_per_split;
  int t_end = t_begin + a.tiles_per_split;
  if (t_end > a.total_tiles) t_end = a.total_tiles;
  const bool do_bias = a.dbias != nullptr && blockIdx.x == 0;
  float bsum = 0.f;

  // per-lane constants of the patch DMA: element e = lane + 64*k of a channel's PH x PW patch
  int pr[NPI], pc[NPI];
#pragma unroll
  for (int k = 0; k < NPI; ++k) {
    const int e = lane + 64 * k;
    pr[k] = e / PW;
    pc[k] = e - pr[k] * PW;
  }
  // dY DMA: lane = pixel of the tile (segment-major: p = s*32 + r*FC + c)
  const int dyr = (lane >> 5) * SR + (lane & 31) / FC, dyc = (lane & 31) % FC;

  // The DMA work of one tile: per wave 16 dY rows (one instruction each) + 16 channels x NPI patch instructions, all
  // buffer_load ... lds: the byte offsets of a lane's elements inside a channel image depend on the TILE only (computed once
  // per tile: TileOff), the channel goes into the scalar descriptor, and out-of-image elements / channels >= C carry an
  // out-of-range offset / an empty descriptor so that the hardware writes the zeros.  `i` is a compile-time constant at
  // every call site (fully unrolled loops).
  constexpr int DMA_PER_WAVE = 16 + 16 * NPI;
  struct TileOff { int b; unsigned dyv; unsigned pv[NPI]; };
  auto tile_off = [&](int tt) -> TileOff {
    TileOff o;
    o.b = tt / a.tiles_per_img;
    const int t = tt - o.b * a.tiles_per_img;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int y0 = ty * TR, x0 = tx * FC;
    const int yd = y0 + dyr;
    o.dyv = yd < H ? (unsigned)(yd * W + x0 + dyc) * 4u : PNSFM_DMA_INVALID;
#pragma unroll
    for (int k = 0; k < NPI; ++k) {
      const int yy = y0 - P + pr[k], xx = x0 - P + pc[k];
      const bool ok = lane + 64 * k < PSR && yy >= 0 && yy < H && xx >= 0 && xx < W;
      o.pv[k] = ok ? (unsigned)(yy * W + xx) * 4u : PNSFM_DMA_INVALID;
    }
    return o;
  };
  auto issue_one = [&](const TileOff& o, float* buf, int i) {
    float* dys = buf;
    float* patch = buf + 64 * DS;
    if (i < 16) {
      const int m = wave + 4 * i;                      // dY row (output channel) of this instruction
      const pnsfm_dma_buf src = pnsfm_make_dma_buf(a.dy + (unsigned)((o.b * a.Cout + co0 + m) * HW), (co0 + m) < a.Cout ? (long)HW * 4 : 0);
      pnsfm_dma4(src, o.dyv, dys + m * DS);
    } else {
      const int q = i - 16;
      const int cil = wave + 4 * (q / NPI), k = q % NPI;
      const pnsfm_dma_buf src = pnsfm_make_dma_buf(a.x + (unsigned)((o.b * a.Cin + ci0 + cil) * HW), (ci0 + cil) < a.Cin ? (long)HW * 4 : 0);
      if (lane + 64 * k < PSR) pnsfm_dma4(src, o.pv[k], patch + cil * PS + 64 * k);
    }
  };

  if (t_begin < t_end) {
    const TileOff o = tile_off(t_begin);
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) issue_one(o, smem, i);
  }
  int cur = 0;
  for (int tt = t_begin; tt < t_end; ++tt) {
    pnsfm_dma_wait();
    __syncthreads();    // this wave's DMA has landed, everyone's has, and the other buffer is free again
    float* buf = smem + cur * BUF;
    float* nbuf = smem + (cur ^ 1) * BUF;
    const TileOff onext = tile_off(tt + 1 < t_end ? tt + 1 : tt);
    const float* dys = buf;
    const float* patch = buf + 64 * DS;
    if (do_bias) {      // bias gradient rides along: 4 threads per dY row, 16 pixels each (tile already in LDS)
      const int m = tid >> 2, q = tid & 3;
      const float* row = dys + m * DS + q * 16;
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) sacc += row[j];
      bsum += sacc;
    }
    const float* Ab = dys + (32 * wm + l32) * DS + half;
    const float* Bb = patch + (32 * wn + l32) * PS + half + ky0 * PW;
    // 2 segments x TG taps = NGROUP groups of 16 MFMAs.  Software pipeline (everything below is fully unrolled, so the
    // buffer parities are compile-time): the 16 X fragments of group g+1 are read from LDS BEFORE group g's MFMAs are
    // issued, both segments' dY fragments are read at the top of the tile, and a slice of the NEXT tile's DMA is issued in
    // front of each group (unconditionally -- past the last tile it re-fetches that tile into the idle buffer -- so that no
    // branch splits the schedule).
    constexpr int NGROUP = 2 * TG;
    // the next tile's DMA is issued during the first two thirds of this tile's groups: the barrier that ends the tile drains
    // vmcnt(0), and a copy issued in the last groups would expose its latency there
    constexpr int NISSUE = (2 * NGROUP + 2) / 3;
    constexpr int PER_GROUP = (DMA_PER_WAVE + NISSUE - 1) / NISSUE;
    float av[2][16], bv[2][16];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg)
#pragma unroll
      for (int j = 0; j < 16; ++j) av[sg][j] = Ab[sg * 32 + 2 * j];
    auto load_b = [&](int gi, float (&dst)[16]) {
      const int sg = gi / TG, tp = gi % TG;
      const int ky = (KS == 3) ? tp / 3 : 0, kx = (KS == 3) ? tp % 3 : tp;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = (2 * j) / FC, c = (2 * j) % FC;
        dst[j] = Bb[(sg * SR + r + ky) * PW + c + kx];
      }
    };
    load_b(0, bv[0]);
#pragma unroll
    for (int gi = 0; gi < NGROUP; ++gi) {
      // The group's share of the next tile's DMA is spread BETWEEN its MFMAs (one instruction + its ~10 address
      // instructions every 16/PER_GROUP MFMAs hides in the 64-cycle shadow of the matrix instruction; issued as one clump
      // it would leave the pipe idle), and the X fragments of group gi+1 are read half-way through.
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if ((j * PER_GROUP) / 16 != ((j + 1) * PER_GROUP) / 16 || PER_GROUP >= 16) {
#pragma unroll
          for (int u = (j * PER_GROUP) / 16; u < ((j + 1) * PER_GROUP) / 16; ++u) {
            const int i = gi * PER_GROUP + u;
            if (i < DMA_PER_WAVE) issue_one(onext, nbuf, i);
          }
        }
        if (j == 8 && gi + 1 < NGROUP) load_b(gi + 1, bv[(gi + 1) & 1]);
        acc[gi % TG] = pnsfm_mfma_32x32x2(av[gi / TG][j], bv[gi & 1][j], acc[gi % TG]);
      }
    }
    cur ^= 1;
  }
  pnsfm_dma_wait();
  __syncthreads();      // the redundant prefetch behind the last tile must land before this workgroup's LDS is released

  // ---- epilogue.  MFMA D layout: row (co) = (r&3) + 8*(r>>2) + 4*half, col (ci) = l32; a lane owns the TG taps of 16
  // (co, ci) pairs, i.e. 16 runs of TG consecutive floats of dW[co][ci][tap] that are 32*TG floats apart between lanes.
  // Storing (or atomically adding) them directly costs 16 partial cache lines per wave instruction -- measured 3-4x the
  // whole kernel's time when the pixel split makes them atomics.  Each wave therefore transposes its tile through LDS (free
  // by now), 16 co rows at a time: row-major [co][ci*TG + tap] in LDS, then every wave instruction covers 64 CONSECUTIVE
  // floats of one dW row.
  {
    constexpr int RW = 32 * TG;                       // floats of one co row of this wave's tile
    float* tb = smem + wave * (16 * RW);
    const int ci_w = ci0 + 32 * wn;                   // first input channel of this wave's tile
    const int ncol = (a.Cin - ci_w < 32 ? a.Cin - ci_w : 32) * TG;      // valid floats per row (<= 0: nothing)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int r = 8 * h + rr;
        const int lr = (r & 3) + 4 * half + 8 * ((r >> 2) & 1);         // co row inside the 16-row pass
#pragma unroll
        for (int tp = 0; tp < TG; ++tp) tb[lr * RW + l32 * TG + tp] = acc[tp][r];
      }
      __syncthreads();
      const int co_b = co0 + 32 * wm + 16 * h;
      for (int row = 0; row < 16; ++row) {
        const int co = co_b + row;
        if (co >= a.Cout) break;
        float* drow = a.dw + (size_t)blockIdx.z * a.zstride + ((size_t)co * a.Cin + ci_w) * KK + ky0 * KS;
        for (int col = lane; col < ncol; col += 64) {
          // KS == 5: a row group covers TG = 5 of the 25 taps of each channel -> runs of 5 floats, 25 apart
          const int dcol = (KS == 5) ? (col / TG) * KK + (col % TG) : col;
          const float v = tb[row * RW + col];
          drow[dcol] = v;        // (pixel-split launch: slab blockIdx.z of the workspace; sum_slabs_kernel adds the slabs)
        }
      }
    }
  }
  if (do_bias) {
    bsum += __shfl_down(bsum, 2);
    bsum += __shfl_down(bsum, 1);
    const int m = tid >> 2;
    if ((tid & 3) == 0 && co0 + m < a.Cout) {
      a.dbias[(size_t)blockIdx.z * a.zstride + co0 + m] = bsum;
    }
  }
}

static int wgrad2_fc(int W) { return W % 32 == 0 ? 32 : (W % 16 == 0 ? 16 : (W % 8 == 0 ? 8 : 0)); }

bool wgrad2_supported(int Cin, int Cout, int H, int W, int ks) {
  if (ks != 1 && ks != 3 && ks != 5) return false;
  if (wgrad2_fc(W) == 0) return false;
  return Cin >= 16 && Cout >= 16 && H >= 1;
}

// total 64-pixel tiles of the launch (the unit the pixel split divides)
int wgrad2_total_tiles(int B, int H, int W) {
  const int fc = wgrad2_fc(W);
  if (!fc) return 0;
  const int TR = 64 / fc;
  return B * (W / fc) * ceil_div(H, TR);
}

// workgroups of one pixel split
int wgrad2_base_blocks(int Cin, int Cout, int ks) { return ceil_div(Cin, 64) * (ks == 5 ? 5 : 1) * ceil_div(Cout, 64); }

template <int KS, int FC>
static int launch_wgrad2(Wgrad2Args a, dim3 grid, hipStream_t s) {
  using Gm = Wgrad2Geom<KS, FC>;
  const size_t smem = 2 * (size_t)Gm::BUF * sizeof(float);
#ifndef PNSFM_EMU
  static unsigned long long raised = 0;      // one bit per device
  if (smem > 64 * 1024 && ensure_lds_limit(reinterpret_cast<const void*>(&conv2d_wgrad2_kernel<KS, FC>), &raised, 160 * 1024,
                                           "conv2d_backward_weight"))
    return -1;
#endif
  PNSFM_LAUNCH((conv2d_wgrad2_kernel<KS, FC>), grid, dim3(256), smem, s, a);
  return check_launch("conv2d_backward_weight (tap-major)");
}

int enqueue_wgrad2(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W, int ks,
                   int split, hipStream_t s) {
  if (!wgrad2_supported(Cin, Cout, H, W, ks)) { set_error("conv2d_backward_weight (tap-major): unsupported shape"); return -1; }
  const int fc = wgrad2_fc(W);
  Wgrad2Args a;
  a.x = x; a.dy = dy; a.dw = dw; a.dbias = dbias;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.tiles_x = W / fc;
  a.tiles_per_img = a.tiles_x * ceil_div(H, 64 / fc);
  a.total_tiles = B * a.tiles_per_img;
  const size_t N = (size_t)Cin * ks * ks;
  {
    // whole [dw | dbias] slabs per split in the grow-only scratch: at most 128 MB of them per launch (see conv2d.hip: wgrad_impl)
    const size_t max_split = ((size_t)128 << 20) / (((size_t)Cout * N + Cout) * sizeof(float));
    if (max_split >= 1 && (size_t)split > max_split) split = (int)max_split;
  }
  if (split < 1) split = 1;
  if (split > a.total_tiles) split = a.total_tiles;
  a.tiles_per_split = ceil_div(a.total_tiles, split);
  a.splitP = ceil_div(a.total_tiles, a.tiles_per_split);
  a.ci_tiles = ceil_div(Cin, 64);
  // pixel-split launch: partial [dw | dbias] slabs in the stream's scratch buffer, added in split order by sum_slabs_kernel
  const size_t slab = (size_t)Cout * N + Cout;
  ScratchLease lease(s, a.splitP > 1 ? (size_t)a.splitP * slab * sizeof(float) : 0);
  a.zstride = 0;
  if (a.splitP > 1) {
    if (!lease.p) return -1;
    a.dw = lease.as<float>();
    a.dbias = dbias ? a.dw + (size_t)Cout * N : nullptr;
    a.zstride = slab;
  }
  dim3 grid(a.ci_tiles * (ks == 5 ? 5 : 1), ceil_div(Cout, 64), a.splitP);
  int rc = 0;
#define PNSFM_W2(KSv)                                                     \
  do {                                                                    \
    if (fc == 32) rc = launch_wgrad2<KSv, 32>(a, grid, s);                \
    else if (fc == 16) rc = launch_wgrad2<KSv, 16>(a, grid, s);           \
    else rc = launch_wgrad2<KSv, 8>(a, grid, s);                          \
  } while (0)
  if (ks == 1) PNSFM_W2(1);
  else if (ks == 3) PNSFM_W2(3);
  else PNSFM_W2(5);
#undef PNSFM_W2
  if (!rc && a.splitP > 1) rc = launch_sum_slabs(a.dw, slab, a.splitP, dw, (size_t)Cout * N, dbias, (size_t)Cout, s);
  return rc;
}

}  // namespace pnsfm
