// conv2d_bx3.h -- fp32 convolution on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 pieces and
// the product is rebuilt from 6 of the 9 piece products.  Included by conv2d.hip (shares ConvArgs / the epilogue / the tuner).
//
// Why: v_mfma_f32_32x32x2_f32 caps the chip at 157 TFLOP/s; v_mfma_f32_32x32x16_bf16 does 16x the MACs per cycle.  With
//   a = a_h + a_m + a_l   (a_h = bf16(a), a_m = bf16(a - a_h), a_l = a - a_h - a_m, round to nearest: 8 + 8 + 8 significand
//   bits, all three exactly representable in bf16, the sum exact; |a_m| <= 2^-9 |a|, |a_l| <= 2^-18 |a|)
//   a*b = a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_l b_h + a_m b_m) + O(2^-26 |a||b|)
// six bf16 MFMAs (fp32 accumulate in the matrix pipe) deliver the product to ~2^-26 relative -- below the rounding of a
// single fp32 FMA -- at 16/6 = 2.7x the MAC rate of the f32 instruction.  Measured on MI355X
// (tools/micro/bf16x3_check.hip, K = 4096 dot products against fp64): max error / sum|a||b| = 9.8e-8 for the 6-product
// form vs 1.5e-7 for v_mfma_f32_32x32x2_f32 itself (whose K-long fmaf chain rounds 16x more often), 3.9e-4 for plain bf16;
// sustained 6-product rate 293-319 TFLOP/s fp32-equivalent.  This is NOT a reduced-precision mode: parity tests hold it to
// the tolerance of the f32 kernels (tests/test_gpu_parity.py::test_conv2d_bx3_*).
//
// Implicit-GEMM structure (same tiles as conv2d_mfma_kernel: block = (32*MT output channels) x (4 waves * NT * 32 pixels)):
//   * K is walked in chunks of 16 input channels = ONE bf16 MFMA k-step per tap; the chunk's halo patch lives in LDS as
//     three bf16 planes [pixel][16 channels] (32 B per pixel and plane), so the B operand of a lane (pixel n = l&31,
//     channels 8*(l>>5) .. +7) is ONE ds_read_b128 per plane, conflict-free (consecutive pixels = consecutive 32 B);
//     the fp32 NCHW input is read through a buffer descriptor (zero padding / ragged channels = hardware range check),
//     split in registers (4 VALU ops per element) and written with ds_write_b128; the NEXT chunk's loads are issued before
//     the last stage of the current chunk so their latency hides behind its MFMAs;
//   * weights are pre-split at pack time (pack_bx3_kernel) into the exact LDS image of the A operand:
//       [32-row m-block][16-channel chunk][tap][piece h,m,l][k-half][32 rows][8 bf16] = 3072 B per (m-block, chunk, tap)
//     so a block's whole weight stream is linear in memory and arrives by 16-byte LDS-DMA (buffer_load_dwordx4 ... lds),
//     double-buffered in stages of G taps: one barrier per G taps;
//   * per tap and wave: 3*(MT+NT) ds_read_b128 feed 6*MT*NT MFMAs (MT = NT = 2: 12 reads, 24 MFMAs = 768 matrix cycles),
//     fragments of tap j+1 are read before the MFMAs of tap j are issued.
// LDS per block = PB*3*PS*32 B (patch) + 2*G*MT*3072 B (weights) + 256 B (the tile's bias values); the host picks G (and PB) so that two blocks share a CU
// where possible (the second block's math covers this block's staging).
#pragma once
#include "adam_math.h"

#define PNSFM_BX3_SLAB 3072      // bytes per (32-row m-block, 16-channel chunk, tap)
#define PNSFM_BX3_MAXIT 4        // patch items (pixel, 8-channel half) a thread prefetches in registers (PS <= 512 pixels); larger patches are staged in rounds

// 8 consecutive-k fp32 values -> the three 16-byte operand pieces.  Exact 3-way split with round-to-nearest pieces:
//   h = bf16(v), m = bf16(v - h), l = v - h - m   (v - h and v - h - m are exact in fp32; l has <= 8 significant bits, so
//   the last conversion is exact too: v == h + m + l).  |m| <= 2^-9 |v|, |l| <= 2^-18 |v|, signs mixed.
// 11 VALU instructions per PAIR of values: 3 v_cvt_pk_bf16_f32, 4 unpacks (shift / mask), 4 subtractions.
__device__ __forceinline__ void bx3_split8(const float (&v)[8], pnsfm_u32x4& H, pnsfm_u32x4& M, pnsfm_u32x4& L) {
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

// OCC = workgroups the register budget is sized for (__launch_bounds__): 2 (<= 256 VGPRs) or 3 (<= 168: one fragment set instead of
// the tap-ahead pair, three patch items in flight) -- three waves per SIMD cover each other's staging / DMA-issue / barrier phases
// (35-50 % of a wave's cycles, tools/bx3_trace.py) where two leave the matrix pipe idle; the host gives such launches <= 53 KB of LDS.
template <int MT, int NT, int OCC>
__global__ void __launch_bounds__(256, OCC) conv2d_bx3_kernel(ConvArgs a) {
  PNSFM_DYN_SMEM(unsigned char, smem);
  constexpr int BM = 32 * MT, MAXIT = (MT * NT == 4 || OCC == 3) ? PNSFM_BX3_MAXIT - 1 : PNSFM_BX3_MAXIT;   // (2,2): acc + fragments leave fewer registers
  const int PS = a.PH * a.PW;
  const int planeB = a.pstride;                  // bytes of one piece plane of the patch
  const int halfB = planeB >> 1;                 // ... which is two half planes: channels 0-7 and 8-15 of the chunk, [pixel][8 ch]
  constexpr bool hp = true;                      // half-plane patch layout (round 2's [pixel][half][8 ch] layout and its A/B switch are gone)
  const int tapB = hp ? 16 : 32;                 // bytes between horizontally adjacent pixels of a plane
  const int patchB = 3 * planeB;
  const int G = a.G;
  const int stageB = G * MT * PNSFM_BX3_SLAB;
  unsigned char* const wbuf0 = smem + a.PB * patchB;
  float* const lds_bias = reinterpret_cast<float*>(wbuf0 + 2 * stageB);      // 64 floats behind the weight stages

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int P = a.KS >> 1, KK = a.KS * a.KS;
  const int H = a.H, W = a.W, HW = H * W;
  const int S = a.S, Hi = a.Hi, Wi = a.Wi, HWi = Hi * Wi;

  // logical block (pixel tile, output-channel tile, K split) of this workgroup: 1-D launch, output-channel tile fastest -- the
  // tiles that read the same input patch are neighbours -- and a contiguous range of that order per XCD (pnsfm_common.h)
  unsigned bx, by, bz;
  {
    // (bmap 3, round 4: weight-heavy layers -- 512 -> 512 @ 12x40: 14 MB of split weights against 4 MB of input -- put the PIXEL tile
    // fastest inside an XCD's range instead, so that the weights of an (output-channel tile, K split) stay in that XCD's 4 MB L2 for all
    // the pixel tiles; with the output-channel tile fastest every pixel tile re-fetched the whole weight stream from memory)
    const unsigned Lb = a.bmap >= 2 ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
    if (a.bmap == 0 || a.bmap == 3) { bx = Lb % (unsigned)a.gx; const unsigned q = Lb / (unsigned)a.gx; by = q % (unsigned)a.gy; bz = q / (unsigned)a.gy; }
    else { by = Lb % (unsigned)a.gy; const unsigned q = Lb / (unsigned)a.gy; bx = q % (unsigned)a.gx; bz = q / (unsigned)a.gx; }
  }
  const int b = (int)bx / a.tiles_per_img;
  const int t = (int)bx - b * a.tiles_per_img;
  const int co0 = (int)by * BM;
  const int c_begin = (int)bz * a.chunks_per_split;
  int c_end = c_begin + a.chunks_per_split;
  if (c_end > a.nchunks) c_end = a.nchunks;

  // ---- pixel-tile geometry (as conv2d_mfma_kernel)
  int py0, px0;
  int boff[NT], oy[NT], ox[NT];
  bool pvalid[NT];
  if (a.mode == 0) {
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int y0 = ty * 4 * NT, x0 = tx * 32;
    py0 = y0 * S - P;
    px0 = x0 * S - P;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int row = wave * NT + nt;
      oy[nt] = y0 + row;
      ox[nt] = x0 + l32;
      pvalid[nt] = oy[nt] < H;
      boff[nt] = (row * a.PW + l32) * S;
    }
  } else if (a.mode == 2) {
    // TW x TH rectangle (16 x 8*NT, or a band of whole rows): slot p of the tile is pixel (p / TW, p % TW)
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int y0 = ty * a.TH, x0 = tx * a.TW;
    py0 = y0 * S - P;
    px0 = x0 * S - P;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int p = (wave * NT + nt) * 32 + l32;
      const int row = p / a.TW, col = p - row * a.TW;
      pvalid[nt] = row < a.TH && y0 + row < H && x0 + col < W;
      oy[nt] = pvalid[nt] ? y0 + row : y0;
      ox[nt] = pvalid[nt] ? x0 + col : x0;
      boff[nt] = pvalid[nt] ? (row * a.PW + col) * S : 0;
    }
  } else {
    const int n0 = t * 128 * NT;
    const int r0 = n0 / W;
    py0 = r0 * S - P;
    px0 = -P;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + (wave * NT + nt) * 32 + l32;
      pvalid[nt] = n < HW;
      const int yy = pvalid[nt] ? n / W : r0;
      oy[nt] = yy;
      ox[nt] = pvalid[nt] ? n - yy * W : 0;
      boff[nt] = ((yy - r0) * a.PW + ox[nt]) * S;
    }
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- patch staging.  Item e = (channel half e / PS, pixel e % PS of the patch); its three 16-byte pieces go to the half plane
  // of each piece plane, 16 bytes per pixel: the B operand of 16 consecutive lanes (consecutive pixels) is then 256 contiguous
  // bytes -- every LDS bank once.  (Round 2's [pixel][half] layout put them 32 bytes apart: a 2-way bank conflict on every B
  // fragment, the 33 % "conflict cycles" of profiles/r02_sq_waits.json.)  A thread's items sit at the same pixel for every chunk: the byte offset inside a channel image is computed
  // once; out-of-image pixels carry an out-of-range offset (the buffer load returns 0 for them, as for channels >= Cin).
  // (multi-source input: the decoder's concatenations are never materialised -- chunk c reads whichever tensor holds its channels)
  const int nitems = 2 * PS;
  const int nit = (nitems + 255) >> 8;
  const bool prefetch = nit <= MAXIT;
  auto item_off = [&](int e) -> unsigned {
    const int hi = hp ? (e >= PS ? 1 : 0) : (e & 1), pix = hp ? e - hi * PS : e >> 1;
    const int r = pix / a.PW, cc = pix - r * a.PW;
    const int yy = py0 + r, xx = px0 + cc;
    const bool ok = e < nitems && yy >= 0 && yy < Hi && xx >= 0 && xx < Wi;
    return ok ? (unsigned)((hi * 8 * HWi + yy * Wi + xx) * 4) : PNSFM_DMA_INVALID;
  };
  auto item_lds = [&](int e) -> int { return (hp && e >= PS) ? halfB + (e - PS) * 16 : e * 16; };
  unsigned gv[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) gv[it] = item_off(it * 256 + tid);
  float raw[MAXIT][8];
  auto chunk_buf = [&](int c) -> pnsfm_buf {       // descriptor over the channels from 16c to the end of THEIR source tensor, this image
    int ci0 = c * 16;
    const float* src = a.x;
    int Cs = a.C0;
    if (ci0 >= a.C0) {                             // wave-uniform
      if (ci0 < a.C01) { src = a.x1; Cs = a.C01 - a.C0; ci0 -= a.C0; }
      else { src = a.x2; Cs = a.Cin - a.C01; ci0 -= a.C01; }
    }
    const long rem = (long)(Cs - ci0) * HWi * 4;
    return pnsfm_make_buf(src + ((size_t)b * Cs + ci0) * HWi, (unsigned)(rem > 0 ? rem : 0));
  };
  auto load_items = [&](int c) {
    const pnsfm_buf buf = chunk_buf(c);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < nit) {
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[it][u] = pnsfm_buf_load(buf, gv[it] + (unsigned)(u * HWi * 4), 0);
      }
  };
  auto write_items = [&](unsigned char* patch) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < nit) {
        const int e = it * 256 + tid;
        pnsfm_u32x4 Hh, Mm, Ll;
        bx3_split8(raw[it], Hh, Mm, Ll);
        if (e < nitems) {
          unsigned char* d = patch + item_lds(e);
          *reinterpret_cast<pnsfm_u32x4*>(d) = Hh;
          *reinterpret_cast<pnsfm_u32x4*>(d + planeB) = Mm;
          *reinterpret_cast<pnsfm_u32x4*>(d + 2 * planeB) = Ll;
        }
      }
  };
  // patches too large for the register prefetch (stride-2 layers, 7x7 with NT = 2): staged in rounds at the chunk boundary
  auto stage_sync = [&](int c, unsigned char* patch) {
    const pnsfm_buf buf = chunk_buf(c);
    for (int e0 = tid; e0 < nitems; e0 += 256) {
      const unsigned off = item_off(e0);
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pnsfm_buf_load(buf, off + (unsigned)(u * HWi * 4), 0);
      pnsfm_u32x4 Hh, Mm, Ll;
      bx3_split8(v, Hh, Mm, Ll);
      unsigned char* d = patch + item_lds(e0);
      *reinterpret_cast<pnsfm_u32x4*>(d) = Hh;
      *reinterpret_cast<pnsfm_u32x4*>(d + planeB) = Mm;
      *reinterpret_cast<pnsfm_u32x4*>(d + 2 * planeB) = Ll;
    }
  };

  // ---- weight stream: stage = G taps of one chunk, MT slabs per tap; one DMA instruction moves 1 KB (one piece of one slab)
  const pnsfm_dma_buf wdesc = pnsfm_make_dma_buf(a.wp, (long)(a.MP / 32) * a.nchunks * KK * PNSFM_BX3_SLAB);
  const int mb0 = (int)by * MT;
  // A stage's slabs sit in LDS as [m tile][tap][piece] -- for one m tile the G taps of a chunk are ONE contiguous run of the packed
  // weight stream (G * 3 KB), so a wave's share of the stage is a contiguous range of 1-KB pieces: source and destination advance by
  // 1024 per instruction and the loop around the DMA is a handful of scalar operations (it used to decode (slab, piece, tap, tile)
  // per instruction: ~25 scalar instructions and 250-300 cycles per DMA, 18-24 % of a wave's time -- tools/bx3_trace.py).
  const unsigned wbase0 = (unsigned)(mb0 * a.nchunks * KK) * PNSFM_BX3_SLAB + lane * 16;
  const unsigned wmtstride = (unsigned)(a.nchunks * KK) * PNSFM_BX3_SLAB;
  auto issue_weights = [&](int c, int tap0, unsigned char* dst) {
    int gcn = KK - tap0;
    if (gcn > G) gcn = G;
    const int run = 3 * gcn, total = MT * run;
    const int per = (total + 3) >> 2;
    int q = wave * per;
    int qe = q + per;
    if (qe > total) qe = total;
    const unsigned src0 = wbase0 + (unsigned)(c * KK + tap0) * PNSFM_BX3_SLAB;
    for (; q < qe; ++q) {
      const int mt = (MT == 2 && q >= run) ? 1 : 0;
      const int r = q - mt * run;
      pnsfm_dma16(wdesc, src0 + mt * wmtstride + (unsigned)r * 1024u, reinterpret_cast<float*>(dst + (mt * G * 3 + r) * 1024));
    }
  };
  // per-lane operand addresses
  unsigned baddr[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) baddr[nt] = hp ? (unsigned)boff[nt] * 16u + (unsigned)(half * halfB) : (unsigned)boff[nt] * 32u + half * 16u;
  const unsigned aaddr = half * 512u + l32 * 16u;

  struct Frag { pnsfm_u32x4 A[MT][3], B[NT][3]; };
  auto load_frag = [&](Frag& f, const unsigned char* wst, int tl, const unsigned char* patch, int tapoffB) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        f.A[mt][s] = *reinterpret_cast<const pnsfm_u32x4*>(wst + ((mt * G + tl) * 3 + s) * 1024 + aaddr);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        f.B[nt][s] = *reinterpret_cast<const pnsfm_u32x4*>(patch + s * planeB + baddr[nt] + tapoffB);
  };
  auto mma = [&](const Frag& f) {
    // smallest terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); tiles interleaved so consecutive MFMAs are independent
#define PNSFM_BX3_P(sa, sb)                                                                       \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                             \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                           \
        acc[mt][nt] = pnsfm_mfma_bf16(f.A[mt][sa], f.B[nt][sb], acc[mt][nt])
    PNSFM_BX3_P(2, 0); PNSFM_BX3_P(0, 2); PNSFM_BX3_P(1, 1); PNSFM_BX3_P(1, 0); PNSFM_BX3_P(0, 1); PNSFM_BX3_P(0, 0);
#undef PNSFM_BX3_P
  };

#ifdef PNSFM_PIPE_TRACE
  // debug build (tools/bx3_trace.py): cycles of this wave in {stage wait, DMA / load issue, MFMA loop, chunk-end staging}
  long long tr_wait = 0, tr_issue = 0, tr_mma = 0, tr_stage = 0, tr_load = 0;
  const long long tr_start = __builtin_readcyclecounter();
#define PNSFM_TR(acc_, expr) do { const long long t0_ = __builtin_readcyclecounter(); expr; acc_ += __builtin_readcyclecounter() - t0_; } while (0)
#else
#define PNSFM_TR(acc_, expr) do { expr; } while (0)
#endif
#ifdef PNSFM_BX3_ABLATE
  // timing experiments (tools/bx3_ablate.py; results are WRONG): 1 no weight DMA in the loop, 2 no patch loads / split / ds_write at
  // chunk ends, 4 no stage barriers, 8 fragments read once per stage, 16 no MFMAs, 32 no epilogue stores
  const int AB = a.ablate;
#define PNSFM_AB(bit) (AB & (bit))
#else
#define PNSFM_AB(bit) 0
#endif
  // ---- prologue: first chunk's patch and first stage's weights
  const int SG = (KK + G - 1) / G;               // stages per chunk
  issue_weights(c_begin, 0, wbuf0);
  if (a.bias != nullptr && a.splitK == 1 && tid < BM) lds_bias[tid] = a.bias[co0 + tid < a.Cout ? co0 + tid : a.Cout - 1];   // published by the first stage barrier
  if (prefetch) { load_items(c_begin); write_items(smem); }
  else stage_sync(c_begin, smem);

#ifdef PNSFM_PIPE_TRACE
  const long long tr_loop = __builtin_readcyclecounter();
#endif
  int pcur = 0, stage = 0;
  for (int c = c_begin; c < c_end; ++c) {
    unsigned char* const patch = smem + pcur * patchB;
    const bool more = c + 1 < c_end;
    for (int sg = 0; sg < SG; ++sg, ++stage) {
      const int tap0 = sg * G;
      const int gcount = (KK - tap0 < G) ? KK - tap0 : G;
      const bool last = sg + 1 == SG;
      PNSFM_TR(tr_wait, pnsfm_dma_wait();    // this wave's weight DMA for the stage has landed ...
               if (!PNSFM_AB(4)) __syncthreads());             // ... and so has everyone's; the patch is visible, the previous stage is consumed
      unsigned char* const wst = wbuf0 + (stage & 1) * stageB;
      unsigned char* const wnext = wbuf0 + ((stage & 1) ^ 1) * stageB;
      // (handing this DMA / load burst out in slices between the taps' MFMA batches was measured and is slower: 124 vs 134 img/s --
      // an LDS-DMA issued inside the MFMA stream stalls the wave longer than the same instruction in a burst)
      PNSFM_TR(tr_issue, {
        if (!PNSFM_AB(1)) {
          if (!last) issue_weights(c, tap0 + G, wnext);
          else if (more) issue_weights(c + 1, 0, wnext);
        }
        if (last && more && prefetch && !PNSFM_AB(2)) PNSFM_TR(tr_load, load_items(c + 1));      // (inside the issue bracket: also counted there)
      });
#ifdef PNSFM_PIPE_TRACE
      const long long tr_m0 = __builtin_readcyclecounter();
#endif

      int ky = tap0 / a.KS, kx = tap0 - ky * a.KS;
      auto tapoff = [&]() -> int {
        const int o = (ky * a.PW + kx) * tapB;
        if (++kx == a.KS) { kx = 0; ++ky; }
        return o;
      };
      if (OCC == 3) {          // one fragment set: the other two waves of the SIMD hide the LDS latency
        Frag f0;
        for (int j = 0; j < gcount; ++j) {
          load_frag(f0, wst, j, patch, tapoff());
          mma(f0);
        }
      } else {
        // Two fragment sets: the reads of tap j+1 are issued before the MFMAs of tap j.  hipcc's wait insertion merges the "f1 was
        // prefetched" and "last tap" paths in front of mma(f0) and waits lgkmcnt(3..0) there, i.e. also for the prefetch it has just
        // issued.  Round 4 peeled the loop so that the steady state waits lgkmcnt(12) (ISA checked) -- and measured it 2 % SLOWER
        // over the step's layer mix in a same-box A/B (7x7 216 -> 211 TFLOP/s, 128 -> 128 @ 48x160 146 -> 138; only 256 -> 256 @ 24x80
        // gained, 132 -> 145): MFMAs that start while the partner wave's and this wave's LDS reads are still streaming are issued
        // less densely than a batch behind a drained counter.  The merged form stays.
        Frag f0, f1;
        load_frag(f0, wst, 0, patch, tapoff());
        if (PNSFM_AB(8)) load_frag(f1, wst, 0, patch, 0);
        for (int j = 0; j < gcount; j += 2) {
          if (j + 1 < gcount && !PNSFM_AB(8)) load_frag(f1, wst, j + 1, patch, tapoff());
          if (!PNSFM_AB(16)) mma(f0);
          if (j + 1 < gcount) {
            if (j + 2 < gcount && !PNSFM_AB(8)) load_frag(f0, wst, j + 2, patch, tapoff());
            if (!PNSFM_AB(16)) mma(f1);
          }
        }
      }

#ifdef PNSFM_PIPE_TRACE
      tr_mma += __builtin_readcyclecounter() - tr_m0;
      const long long tr_s0 = __builtin_readcyclecounter();
#endif
      if (last && more && !PNSFM_AB(2)) {
        if (a.PB == 2) {
          // the other patch buffer was last read two chunks ago: write the next chunk's patch while the other waves finish
          if (prefetch) write_items(smem + (pcur ^ 1) * patchB);
          else stage_sync(c + 1, smem + (pcur ^ 1) * patchB);
        } else {
          __syncthreads();   // every wave is done with this chunk's patch
          if (prefetch) write_items(patch);
          else stage_sync(c + 1, patch);
        }
      }
#ifdef PNSFM_PIPE_TRACE
      tr_stage += __builtin_readcyclecounter() - tr_s0;
#endif
    }
    if (a.PB == 2) pcur ^= 1;
  }
#ifdef PNSFM_PIPE_TRACE
  const long long tr_epi = __builtin_readcyclecounter();
#endif

  if (!PNSFM_AB(32) || acc[0][0][0] == 1.2345f) conv_epilogue<MT, NT>(a, acc, b, co0, half, oy, ox, pvalid, (int)bz, lds_bias, t, wave);
#undef PNSFM_AB
#ifdef PNSFM_PIPE_TRACE
  if (a.trace && lane == 0) {
    const long long tr_end = __builtin_readcyclecounter();
    long long* tt = a.trace + ((size_t)blockIdx.x * 4 + wave) * 8;
    tt[0] = tr_wait; tt[1] = tr_issue; tt[2] = tr_mma; tt[3] = tr_stage; tt[4] = tr_end - tr_start; tt[5] = tr_loop - tr_start;
    tt[6] = tr_end - tr_epi; tt[7] = stage + 4096 * tr_load;
  }
#endif
#undef PNSFM_TR
}

// ---- weight packer for the bx3 kernels: fp32 [Cout][Cin][k][k] -> the split LDS images described above.
//   forward : M = Cout, K = Cin :  A[m][k][tap] = w[m][k][tap]
//   backward: M = Cin,  K = Cout:  A[m][k][tap] = w[k][m][KK-1-tap]       (taps flipped: dX = conv(dY, rot180(W)^T))
// One block per (32-row m-block, 16-channel chunk); taps go through LDS -- all 9 of a 3x3 at once, one kernel row at a time for
// 5x5 / 7x7 -- so that global reads are runs of (channel, tap) and every output tap is written as one contiguous 3072-byte
// slab.  KS is a template parameter: the index arithmetic of the copy loops divides by compile-time constants only.
template <int KS>
__device__ __forceinline__ void pack_bx3_block(float (*tile)[16][33], const float* __restrict__ w, unsigned char* __restrict__ wp_fwd,
                                               unsigned char* __restrict__ wp_bwd, int Cin, int Cout, int nchF, int nchB, int nf,
                                               int blk) {
  constexpr int KK = KS * KS, TSEG = (KS == 3) ? 9 : KS;
  const bool fwd = blk < nf;
  if (!fwd) blk -= nf;
  const int nch = fwd ? nchF : nchB;
  const int mb = blk / nch, ch = blk - mb * nch;
  unsigned char* out = (fwd ? wp_fwd : wp_bwd) + (size_t)(mb * nch + ch) * KK * PNSFM_BX3_SLAB;
  const int m0 = mb * 32, k0 = ch * 16;
  const int Mn = fwd ? Cout : Cin, Kn = fwd ? Cin : Cout;
  for (int t0 = 0; t0 < KK; t0 += TSEG) {
    if (fwd) {
      // for a row m: (k, tap) is contiguous in w
      for (int e = threadIdx.x; e < 32 * 16 * TSEG; e += 256) {
        const int m = e / (16 * TSEG), r = e - m * (16 * TSEG), k = r / TSEG, tt = r - k * TSEG;
        float v = 0.f;
        if (m0 + m < Mn && k0 + k < Kn) v = w[((size_t)(m0 + m) * Cin + k0 + k) * KK + t0 + tt];
        tile[tt][k][m] = v;
      }
    } else {
      // for a K row (an output channel of w): (m = ci, tap) is contiguous in w; output tap t reads source tap KK-1-t
      for (int e = threadIdx.x; e < 16 * 32 * TSEG; e += 256) {
        const int k = e / (32 * TSEG), r = e - k * (32 * TSEG), m = r / TSEG, tt = r - m * TSEG;
        float v = 0.f;
        if (m0 + m < Mn && k0 + k < Kn) v = w[((size_t)(k0 + k) * Cin + m0 + m) * KK + (KK - 1 - (t0 + tt))];
        tile[tt][k][m] = v;
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < TSEG * 64; e += 256) {
      const int tt = e >> 6, kh = (e >> 5) & 1, m = e & 31;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = tile[tt][kh * 8 + i][m];
      pnsfm_u32x4 Hh, Mm, Ll;
      bx3_split8(v, Hh, Mm, Ll);
      unsigned char* o = out + (size_t)(t0 + tt) * PNSFM_BX3_SLAB + kh * 512 + m * 16;
      *reinterpret_cast<pnsfm_u32x4*>(o) = Hh;
      *reinterpret_cast<pnsfm_u32x4*>(o + 1024) = Mm;
      *reinterpret_cast<pnsfm_u32x4*>(o + 2048) = Ll;
    }
    __syncthreads();
  }
}

template <int KS>
__global__ void __launch_bounds__(256) pack_bx3_kernel(const float* __restrict__ w, unsigned char* __restrict__ wp_fwd,
                                                       unsigned char* __restrict__ wp_bwd, int Cin, int Cout, int nchF,
                                                       int nchB, int nf) {
  __shared__ float tile[(KS == 3) ? 9 : KS][16][33];
  pack_bx3_block<KS>(tile, w, wp_fwd, wp_bwd, Cin, Cout, nchF, nchB, nf, (int)blockIdx.x);
}

// One launch for MANY weights (every conv layer of a model right after the optimizer step): a device-resident table of
// {weight, packed buffers, geometry, first block}, a binary search per workgroup.  The ~100 per-layer launches this replaces cost
// ~10 us each whatever their size (two dependent memory round trips behind a launch).
struct PackItem {
  const float* w;
  unsigned char* pf;
  unsigned char* pb;
  int Cin, Cout, ks, nchF, nchB, nf, blk0, nblk;
};
__global__ void __launch_bounds__(256) pack_bx3_table_kernel(const PackItem* __restrict__ tab, int n) {
  __shared__ float tile[9][16][33];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackItem it = tab[lo];
  const int blk = (int)blockIdx.x - it.blk0;
  if (blk >= it.nblk) return;
  if (it.ks == 3) pack_bx3_block<3>(tile, it.w, it.pf, it.pb, it.Cin, it.Cout, it.nchF, it.nchB, it.nf, blk);
  else if (it.ks == 1) pack_bx3_block<1>(tile, it.w, it.pf, it.pb, it.Cin, it.Cout, it.nchF, it.nchB, it.nf, blk);
  else if (it.ks == 5) pack_bx3_block<5>(tile, it.w, it.pf, it.pb, it.Cin, it.Cout, it.nchF, it.nchB, it.nf, blk);
  else pack_bx3_block<7>(tile, it.w, it.pf, it.pb, it.Cin, it.Cout, it.nchF, it.nchB, it.nf, blk);
}

// ---- Adam update + re-pack of the split-bf16 conv weights in ONE pass (round 5).  adam_flat_kernel moved 28 B
// per parameter and pack_bx3_table_kernel then read every weight TWICE (forward image, backward-data image) and wrote 2 x 6 B: 48 B
// per parameter and two launches between backward and the next forward, where nothing overlaps them.  Here a workgroup owns a
// 32 (co) x 32 (ci) super-tile of one weight: it reads p, g, m, v once (16 B), applies the update of adam_math.h (the same inline
// function adam_flat_kernel uses: bit-identical parameters and moments), writes p, m, v back (12 B) and, from the updated values it
// holds in LDS, both packed images (12 B): 40 B per parameter, one launch.  Taps go through the tile like in pack_bx3_block (all 9
// of a 3x3 at once, one kernel row at a time for 5x5 / 7x7); a super-tile covers two 16-channel chunks of the forward image's
// m-block (co tile) and two of the backward image's (ci tile).
struct AdamPackItem {
  float* w;            // the parameter's slice of the parameter arena, [Cout][Cin][k][k]
  const float* g;      // ... of the gradient arena
  float* m;            // ... of the exp_avg arena
  float* v;            // ... of the exp_avg_sq arena
  const float* hp;     // the group's device-resident hyper-parameters (adam_math.h)
  unsigned char* pf;
  unsigned char* pb;
  int Cin, Cout, ks, nchF, nchB, citiles, blk0, nblk;
};

template <int KS>
__device__ __forceinline__ void adam_pack_block(float (*tile)[32][33], const AdamPackItem& it, int blk) {
  constexpr int KK = KS * KS, TSEG = (KS == 3) ? 9 : KS;
  const int cot = blk / it.citiles, cit = blk - cot * it.citiles;
  const int co0 = cot * 32, ci0 = cit * 32;
  const int Cin = it.Cin, Cout = it.Cout;
  const AdamCoef c = adam_coef(it.hp);
  for (int t0 = 0; t0 < KK; t0 += TSEG) {
    // update: for a row co the (ci, tap) elements of the segment are runs of TSEG floats, KK apart (one run of 32 * 9 for a 3x3)
    // four elements per thread and pass, all 16 loads issued before the first store: the arenas are plain pointers (the stores of one
    // element could alias the loads of the next as far as the compiler knows), and one element at a time left 4 loads in flight per
    // lane -- 4.8 TB/s where the flat kernel's float4 stream reaches 5.2.  (32 * 32 * TSEG is a multiple of 4 * 256 for TSEG 1, 5, 7, 9.)
    static_assert((32 * 32 * TSEG) % (4 * 256) == 0, "adam_pack_block: element count per segment");
    for (int e0 = threadIdx.x; e0 < 32 * 32 * TSEG; e0 += 4 * 256) {
      size_t idx[4];
      bool ok[4];
      int lco[4], lci[4], ltt[4];
      float pv[4], gv[4], mv[4], vv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 256 * u;
        const int co = e / (32 * TSEG), r = e - co * (32 * TSEG), ci = r / TSEG, tt = r - ci * TSEG;
        lco[u] = co; lci[u] = ci; ltt[u] = tt;
        ok[u] = co0 + co < Cout && ci0 + ci < Cin;
        idx[u] = ok[u] ? ((size_t)(co0 + co) * Cin + ci0 + ci) * KK + t0 + tt : (size_t)0;      // (element 0: loaded, never stored)
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { pv[u] = it.w[idx[u]]; gv[u] = it.g[idx[u]]; mv[u] = it.m[idx[u]]; vv[u] = it.v[idx[u]]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) adam_update(c, gv[u], pv[u], mv[u], vv[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) { it.w[idx[u]] = pv[u]; it.m[idx[u]] = mv[u]; it.v[idx[u]] = vv[u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) tile[ltt[u]][lco[u]][lci[u]] = ok[u] ? pv[u] : 0.f;
    }
    __syncthreads();
    // forward image: m-block = co tile, chunks 2 cit and 2 cit + 1:  A[m = co][k = ci][tap]
    for (int e = threadIdx.x; e < TSEG * 128; e += 256) {
      const int tt = e >> 7, ch = (e >> 6) & 1, kh = (e >> 5) & 1, mrow = e & 31;
      const int chunk = 2 * cit + ch;
      if (chunk < it.nchF) {
        float vals[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) vals[i] = tile[tt][mrow][ch * 16 + kh * 8 + i];
        pnsfm_u32x4 Hh, Mm, Ll;
        bx3_split8(vals, Hh, Mm, Ll);
        unsigned char* o = it.pf + ((size_t)(cot * it.nchF + chunk) * KK + t0 + tt) * PNSFM_BX3_SLAB + kh * 512 + mrow * 16;
        *reinterpret_cast<pnsfm_u32x4*>(o) = Hh;
        *reinterpret_cast<pnsfm_u32x4*>(o + 1024) = Mm;
        *reinterpret_cast<pnsfm_u32x4*>(o + 2048) = Ll;
      }
    }
    // backward-data image: m-block = ci tile, chunks 2 cot and 2 cot + 1:  A[m = ci][k = co][KK - 1 - tap]
    for (int e = threadIdx.x; e < TSEG * 128; e += 256) {
      const int tt = e >> 7, ch = (e >> 6) & 1, kh = (e >> 5) & 1, mrow = e & 31;
      const int chunk = 2 * cot + ch;
      if (chunk < it.nchB) {
        float vals[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) vals[i] = tile[tt][ch * 16 + kh * 8 + i][mrow];
        pnsfm_u32x4 Hh, Mm, Ll;
        bx3_split8(vals, Hh, Mm, Ll);
        unsigned char* o = it.pb + ((size_t)(cit * it.nchB + chunk) * KK + (KK - 1 - (t0 + tt))) * PNSFM_BX3_SLAB + kh * 512 + mrow * 16;
        *reinterpret_cast<pnsfm_u32x4*>(o) = Hh;
        *reinterpret_cast<pnsfm_u32x4*>(o + 1024) = Mm;
        *reinterpret_cast<pnsfm_u32x4*>(o + 2048) = Ll;
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) adam_pack_table_kernel(const AdamPackItem* __restrict__ tab, int n) {
  __shared__ float tile[9][32][33];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const AdamPackItem it = tab[lo];
  const int blk = (int)blockIdx.x - it.blk0;
  if (blk >= it.nblk) return;
  if (it.ks == 3) adam_pack_block<3>(tile, it, blk);
  else if (it.ks == 1) adam_pack_block<1>(tile, it, blk);
  else if (it.ks == 5) adam_pack_block<5>(tile, it, blk);
  else adam_pack_block<7>(tile, it, blk);
}
