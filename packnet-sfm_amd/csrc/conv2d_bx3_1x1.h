// conv2d_bx3_1x1.h -- 1x1 convolution on the split-bf16 arithmetic WITHOUT LDS (round 6; tuner variant 8).  Included by conv2d.hip.
//
// The 1x1 layers are the shortcuts of the residual blocks (/root/reference/packnet_sfm/networks/layers/packnet/layers01.py:57-60) and
// their backward-data: 23 launches per training step at 192x640, 0.5 ms, all of them 19-29 us whatever their size -- 256 -> 256 on
// 4 x 24x80 is 1 GFLOP over 16 MB (4 us of either) and took 28.  conv2d_bx3_kernel walks K in 16-channel chunks and per chunk does
// global load -> split -> ds_write -> barrier -> 6 MT NT MFMAs of ONE tap: a dependent memory round trip and a barrier for 200-800
// cycles of matrix work, sixteen times in a row.
//
// Without a halo nothing is shared between the pixels of a tile, so nothing has to go through LDS:
//   * B operand: lane (pixel n = l & 31, k-half l >> 5) loads its 8 channels of its own pixel straight from NCHW (per channel the 32
//     lanes of a half read 128 contiguous bytes; ragged channels / pixels = the buffer descriptor's range check) and splits them
//     in registers -- the fragment layout of v_mfma_f32_32x32x16_bf16 as it comes;
//   * A operand: the packed weight stream IS the fragment image (conv2d_bx3.h: [m-block][chunk][piece][k-half][32 rows][8 bf16]), so
//     a lane loads its 16 bytes per piece from global memory; the four waves of a workgroup read the same 3 KB per chunk (L1);
//   * no barrier, no LDS, and D chunks of both operands in flight per wave (D = 2 / 3 / 4 for the (2,2) / two-tile / (1,1) wave
//     tile): the memory latency is paid once per D chunks instead of once per chunk, and with no LDS the occupancy is the
//     register file's.
// Same six piece products per chunk in the same order as conv2d_bx3_kernel: bit-identical results for the same K split
// (tests/test_gpu_round6.py::test_conv1x1_kernel_*).  Workgroup = 4 waves x NT x 32 consecutive pixels of ONE image x 32 MT output
// channels; 1-D launch, output-channel tile fastest inside an XCD's contiguous range (the tiles that read the same pixels share an L2).
#pragma once

template <int MT, int NT>
__global__ void __launch_bounds__(256) conv1x1_bx3_kernel(ConvArgs a) {
  constexpr int BM = 32 * MT;
  constexpr int D = MT * NT == 4 ? 2 : (MT * NT == 2 ? 3 : 4);      // chunks in flight
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int HW = a.H * a.W;
  unsigned bx, by, bz;
  {
    const unsigned Lb = a.bmap >= 2 ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
    by = Lb % (unsigned)a.gy;
    const unsigned q = Lb / (unsigned)a.gy;
    bx = q % (unsigned)a.gx;
    bz = q / (unsigned)a.gx;
  }
  const int b = (int)bx / a.tiles_per_img;
  const int t = (int)bx - b * a.tiles_per_img;
  const int co0 = (int)by * BM;
  const int c_begin = (int)bz * a.chunks_per_split;
  int c_end = c_begin + a.chunks_per_split;
  if (c_end > a.nchunks) c_end = a.nchunks;

  int oy[NT], ox[NT];
  bool pvalid[NT];
  unsigned xoff[NT];      // byte offset of (channel 8 * half, this lane's pixel) in the image's [Cin][HW] slice
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = t * 128 * NT + (wave * NT + nt) * 32 + l32;
    pvalid[nt] = p < HW;
    oy[nt] = 0;
    ox[nt] = pvalid[nt] ? p : 0;
    xoff[nt] = pvalid[nt] ? (unsigned)((8 * half * HW + p) * 4) : PNSFM_DMA_INVALID;
  }
  // channels >= Cin (the ragged last chunk) are past the descriptor's range and read as zero
  const pnsfm_buf xbuf = pnsfm_make_buf(a.x + (size_t)b * a.Cin * HW, (unsigned)((size_t)a.Cin * HW * 4));
  const unsigned char* const wl = reinterpret_cast<const unsigned char*>(a.wp) + ((size_t)(co0 / 32) * a.nchunks) * PNSFM_BX3_SLAB +
                                  half * 512 + l32 * 16;
  const size_t wmt = (size_t)a.nchunks * PNSFM_BX3_SLAB;
  const unsigned cstep = 16u * (unsigned)HW * 4u, istep = (unsigned)HW * 4u;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  struct Raw { float x[NT][8]; pnsfm_u32x4 A[MT][3]; };
  Raw raw[D];
  auto load = [&](Raw& r, int c) {
    const unsigned co = (unsigned)c * cstep;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 8; ++i) r.x[nt][i] = pnsfm_buf_load(xbuf, xoff[nt] + co + (unsigned)i * istep, 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        r.A[mt][s] = *reinterpret_cast<const pnsfm_u32x4*>(wl + mt * wmt + (size_t)c * PNSFM_BX3_SLAB + s * 1024);
  };
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (c_begin + d < c_end) load(raw[d], c_begin + d);

  for (int c = c_begin; c < c_end; c += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (c + d < c_end) {
        pnsfm_u32x4 Bp[NT][3], Ap[MT][3];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bx3_split8(raw[d].x[nt], Bp[nt][0], Bp[nt][1], Bp[nt][2]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int s = 0; s < 3; ++s) Ap[mt][s] = raw[d].A[mt][s];
        if (c + d + D < c_end) load(raw[d], c + d + D);
        // smallest terms first, as conv2d_bx3_kernel: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)   [A piece, B piece]
#define PNSFM_1X1_P(sa, sb)                                                                       \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                         \
          _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                       \
            acc[mt][nt] = pnsfm_mfma_bf16(Ap[mt][sa], Bp[nt][sb], acc[mt][nt])
        PNSFM_1X1_P(2, 0); PNSFM_1X1_P(0, 2); PNSFM_1X1_P(1, 1); PNSFM_1X1_P(1, 0); PNSFM_1X1_P(0, 1); PNSFM_1X1_P(0, 0);
#undef PNSFM_1X1_P
      }
    }
  }

  conv_epilogue<MT, NT, true>(a, acc, b, co0, half, oy, ox, pvalid, (int)bz, nullptr);
}
