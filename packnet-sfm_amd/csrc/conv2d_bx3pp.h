// conv2d_bx3pp.h -- the split-bf16 implicit GEMM of conv2d_bx3.h as a PING-PONG workgroup (round 5; variant 7 of the tuner).
//
// What conv2d_bx3_kernel loses (tools/bx3_trace.py, tools/bx3_ablate.py): a wave spends 40-55 % of its cycles
// OUTSIDE the MFMA loop -- issuing the next weight stage's LDS-DMA (8-14 %), loading / splitting / writing the next patch (10-17 %),
// waiting at the stage barrier -- and the only thing that covers those phases is whatever the co-resident workgroup's wave on the
// same SIMD happens to be doing.  Two independent workgroups drift into the same phase as often as not (both compute: they share the
// matrix pipe at half rate each; both stage: the pipe idles): the matrix pipe is busy 47.6 % of the SIMD-cycles of that kernel.
//
// Here the two waves of a SIMD belong to ONE 512-thread workgroup and alternate by construction.  The workgroup owns TWO pixel
// tiles of the same output-channel tile; waves 0-3 ("group 0") compute the first, waves 4-7 ("group 1") the second.  Time is cut
// into half-steps separated by workgroup barriers; in half-step h group (h & 1) runs the MFMAs of one weight stage on its tile and
// the other group does everything that is NOT matrix work for its own next stage:
//
//     half-step     2s (A)                      2s + 1 (B)                   2s + 2 (A)
//     group 0       MFMA stage s                stage patch / DMA / loads    MFMA stage s + 1
//     group 1       stage patch / DMA / loads   MFMA stage s                 stage patch / DMA / loads
//
//   * both groups read the SAME weight stage (one LDS-DMA stream per workgroup: half the weight traffic and half the DMA
//     instructions per MFMA of the two-workgroup form); weight stages live in a ring of THREE buffers and stage s + 2 is fetched
//     during stage s -- half of its 1-KB pieces by each group in its own staging half-step -- so that nobody ever waits for a
//     DMA it has just issued: a wave drains vmcnt only at the END of its compute half-step, a whole half-step after it issued;
//   * a group's patch buffer is private to it and single: the chunk's last MFMA half-step is followed by the group's own staging
//     half-step, in which the next chunk's patch (fetched into registers one staging half-step earlier, as in conv2d_bx3_kernel)
//     is split and written in place;
//   * the barrier that ends a staging half-step waits for LDS writes only (PNSFM_BARRIER_LDS): the global loads and DMA pieces the
//     group has just issued stay in flight across it; the computing group arrives at that barrier one tap EARLY, with the last
//     tap's fragments already in registers, and issues that tap's MFMAs behind it -- underneath them the released partner waits
//     for its weights and reads its first fragments, so the pipe never sees the head of a half-step;
//   * group 0 finishes one half-step before group 1 and stores its tile underneath group 1's last MFMAs.
// LDS: 2 patch buffers (one per group) + 3 weight stages of G taps + the tile's bias values <= 160 KB; one workgroup per CU,
// 256 VGPRs per wave.  Tiles, fragment layouts, the weight stream and the arithmetic are conv2d_bx3.h's (bit-identical results for
// the same chunk order: the accumulation order inside a tile does not change).
#pragma once
#include <type_traits>

template <int MT, int NT, int G>
__global__ void __launch_bounds__(512) conv2d_bx3pp_kernel(ConvArgs a) {
  PNSFM_DYN_SMEM(unsigned char, smem);
#ifdef PNSFM_PIPE_TRACE
  const long long tr_start = __builtin_readcyclecounter();
#endif
  constexpr int BM = 32 * MT, MAXIT = (MT * NT == 4) ? PNSFM_BX3_MAXIT - 1 : PNSFM_BX3_MAXIT;
  const int PS = a.PH * a.PW;
  const int planeB = a.pstride;                  // bytes of one piece plane of a patch: two half planes [channels 0-7 | 8-15][pixel][8 ch]
  const int halfB = planeB >> 1;
  const int patchB = 3 * planeB;
  const int stageB = G * MT * PNSFM_BX3_SLAB;
  unsigned char* const wbuf0 = smem + 2 * patchB;
  float* const lds_bias = reinterpret_cast<float*>(wbuf0 + 3 * stageB);

  const int tid = threadIdx.x;
  const int lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wave8 = PNSFM_UNIFORM(tid >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3, gtid = tid & 255;      // group (0 | 1), wave inside the group, thread inside the group
  unsigned char* const mypatch = smem + grp * patchB;
  const int P = a.KS >> 1, KK = a.KS * a.KS;
  const int H = a.H, W = a.W, HW = H * W;
  const int S = a.S, Hi = a.Hi, Wi = a.Wi, HWi = Hi * Wi;

  // logical block (pair of pixel tiles, output-channel tile, K split): as conv2d_bx3_kernel, a.gx = number of tile PAIRS
  unsigned bx, by, bz;
  {
    const unsigned Lb = a.bmap >= 2 ? pnsfm_xcd_logical_block(blockIdx.x, gridDim.x) : blockIdx.x;
    if (a.bmap == 0 || a.bmap == 3) { bx = Lb % (unsigned)a.gx; const unsigned q = Lb / (unsigned)a.gx; by = q % (unsigned)a.gy; bz = q / (unsigned)a.gy; }
    else { by = Lb % (unsigned)a.gy; const unsigned q = Lb / (unsigned)a.gy; bx = q % (unsigned)a.gx; bz = q / (unsigned)a.gx; }
  }
  // tiles are numbered over the whole batch: a pair may straddle two images; an odd total leaves the last workgroup's group 1 idle
  // (it still takes part in every barrier and fetches its half of the weight stages)
  const int Tall = a.B * a.tiles_per_img;
  int T = 2 * (int)bx + grp;
  const bool tvalid = T < Tall;
  if (!tvalid) T = Tall - 1;
  const int b = T / a.tiles_per_img;
  const int t = T - b * a.tiles_per_img;
  const int co0 = (int)by * BM;
  const int c_begin = (int)bz * a.chunks_per_split;
  int c_end = c_begin + a.chunks_per_split;
  if (c_end > a.nchunks) c_end = a.nchunks;

  // ---- pixel-tile geometry of this group's tile (as conv2d_bx3_kernel)
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
      pvalid[nt] = tvalid && oy[nt] < H;
      boff[nt] = (row * a.PW + l32) * S;
    }
  } else if (a.mode == 2) {
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    const int y0 = ty * a.TH, x0 = tx * a.TW;
    py0 = y0 * S - P;
    px0 = x0 * S - P;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int p = (wave * NT + nt) * 32 + l32;
      const int row = p / a.TW, col = p - row * a.TW;
      pvalid[nt] = tvalid && row < a.TH && y0 + row < H && x0 + col < W;
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
      pvalid[nt] = tvalid && n < HW;
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

  // ---- patch staging by the 256 threads of the group (items, offsets and the LDS image are conv2d_bx3_kernel's half-plane layout)
  const int nitems = 2 * PS;
  const int nit = (nitems + 255) >> 8;
  const bool prefetch = nit <= MAXIT;
  auto item_off = [&](int e) -> unsigned {
    const int hi = e >= PS ? 1 : 0, pix = e - hi * PS;
    const int r = pix / a.PW, cc = pix - r * a.PW;
    const int yy = py0 + r, xx = px0 + cc;
    const bool ok = tvalid && e < nitems && yy >= 0 && yy < Hi && xx >= 0 && xx < Wi;
    return ok ? (unsigned)((hi * 8 * HWi + yy * Wi + xx) * 4) : PNSFM_DMA_INVALID;
  };
  auto item_lds = [&](int e) -> int { return e >= PS ? halfB + (e - PS) * 16 : e * 16; };
  unsigned gv[MAXIT];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) gv[it] = item_off(it * 256 + gtid);
  float raw[MAXIT][8];
  auto chunk_buf = [&](int c) -> pnsfm_buf {
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
  auto write_items = [&]() {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < nit) {
        const int e = it * 256 + gtid;
        pnsfm_u32x4 Hh, Mm, Ll;
        bx3_split8(raw[it], Hh, Mm, Ll);
        if (e < nitems) {
          unsigned char* d = mypatch + item_lds(e);
          *reinterpret_cast<pnsfm_u32x4*>(d) = Hh;
          *reinterpret_cast<pnsfm_u32x4*>(d + planeB) = Mm;
          *reinterpret_cast<pnsfm_u32x4*>(d + 2 * planeB) = Ll;
        }
      }
  };
  // patches too large for the register prefetch: loaded, split and written in rounds inside the staging half-step
  auto stage_sync = [&](int c) {
    const pnsfm_buf buf = chunk_buf(c);
    for (int e0 = gtid; e0 < nitems; e0 += 256) {
      const unsigned off = item_off(e0);
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pnsfm_buf_load(buf, off + (unsigned)(u * HWi * 4), 0);
      pnsfm_u32x4 Hh, Mm, Ll;
      bx3_split8(v, Hh, Mm, Ll);
      unsigned char* d = mypatch + item_lds(e0);
      *reinterpret_cast<pnsfm_u32x4*>(d) = Hh;
      *reinterpret_cast<pnsfm_u32x4*>(d + planeB) = Mm;
      *reinterpret_cast<pnsfm_u32x4*>(d + 2 * planeB) = Ll;
    }
  };

  // ---- weight stream: this group's HALF of a stage (G taps of one chunk, MT slabs per tap, 1-KB pieces; layout [m tile][tap][piece]
  // in LDS as in conv2d_bx3_kernel).  Group 1 takes the first half of the pieces, group 0 the second; four waves share a half.
  const pnsfm_dma_buf wdesc = pnsfm_make_dma_buf(a.wp, (long)(a.MP / 32) * a.nchunks * KK * PNSFM_BX3_SLAB);
  const int mb0 = (int)by * MT;
  const unsigned wbase0 = (unsigned)(mb0 * a.nchunks * KK) * PNSFM_BX3_SLAB + lane * 16;
  const unsigned wmtstride = (unsigned)(a.nchunks * KK) * PNSFM_BX3_SLAB;
  // A stage always has G tap slots: the slots of a chunk's ragged last stage that lie beyond its KK taps are fetched with an
  // out-of-range offset, which makes the LDS-DMA write ZEROS there (pnsfm_common.h) -- the compute half-step runs all G taps.
  auto issue_weights_half = [&](int c, int tap0, unsigned char* dst) {
    int gcn = KK - tap0;
    if (gcn > G) gcn = G;
    constexpr int run = 3 * G, total = MT * run;
    constexpr int hsplit = total >> 1;
    const int q0 = grp ? 0 : hsplit, q1 = grp ? hsplit : total;
    const int per = (q1 - q0 + 3) >> 2;
    int q = q0 + wave * per;
    int qe = q + per;
    if (qe > q1) qe = q1;
    const unsigned src0 = wbase0 + (unsigned)(c * KK + tap0) * PNSFM_BX3_SLAB;
    for (; q < qe; ++q) {
      const int mt = (MT == 2 && q >= run) ? 1 : 0;
      const int r = q - mt * run;
      const unsigned off = r < 3 * gcn ? src0 + mt * wmtstride + (unsigned)r * 1024u : PNSFM_DMA_INVALID;
      pnsfm_dma16(wdesc, off, reinterpret_cast<float*>(dst + (mt * G * 3 + r) * 1024));
    }
  };
  // per-lane operand addresses
  unsigned baddr[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) baddr[nt] = (unsigned)boff[nt] * 16u + (unsigned)(half * halfB);
  const unsigned aaddr = half * 512u + l32 * 16u;

  struct Frag { pnsfm_u32x4 A[MT][3], B[NT][3]; };
  auto load_frag = [&](Frag& f, const unsigned char* wst, int tl, int tapoffB) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        f.A[mt][s] = *reinterpret_cast<const pnsfm_u32x4*>(wst + ((mt * G + tl) * 3 + s) * 1024 + aaddr);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int s = 0; s < 3; ++s)
        f.B[nt][s] = *reinterpret_cast<const pnsfm_u32x4*>(mypatch + s * planeB + baddr[nt] + tapoffB);
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

  // ---- prologue: weight stages 0 and 1 (both groups fetch their halves), the first chunk's patch, the tile's bias values
  const int SG = (KK + G - 1) / G;               // stages per chunk
  const int NS = (c_end - c_begin) * SG;         // stages of this workgroup
  issue_weights_half(c_begin, 0, wbuf0);
  if (NS > 1) {
    if (SG > 1) issue_weights_half(c_begin, G, wbuf0 + stageB);
    else issue_weights_half(c_begin + 1, 0, wbuf0 + stageB);
  }
  if (a.bias != nullptr && a.splitK == 1 && tid < BM) lds_bias[tid] = a.bias[co0 + tid < a.Cout ? co0 + tid : a.Cout - 1];
  if (prefetch) { load_items(c_begin); write_items(); }
  else stage_sync(c_begin);
  PNSFM_BARRIER_ALL();
  // group 0 computes stage 0 next: the loads a staging half-step would have issued for it (the next chunk's patch, if stage 0 is its
  // chunk's last) go out here, behind the barrier, and land underneath its MFMAs
  if (grp == 0 && prefetch && SG == 1 && c_begin + 1 < c_end) load_items(c_begin + 1);

#ifdef PNSFM_PIPE_TRACE
  // debug build (tools/pp_trace.py): cycles of this wave in {compute half-steps, of which in front of the first MFMA batch; staging
  // half-steps; at the barrier behind a compute / a staging half-step}
  long long tr_comp = 0, tr_head = 0, tr_stage = 0, tr_bwc = 0, tr_bws = 0;
  const long long tr_loop = __builtin_readcyclecounter();
#define PNSFM_TRC(acc_, expr) do { const long long t0_ = __builtin_readcyclecounter(); expr; acc_ += __builtin_readcyclecounter() - t0_; } while (0)
#else
#define PNSFM_TRC(acc_, expr) do { expr; } while (0)
#endif
  // this group's NEXT compute stage (index, chunk, stage inside the chunk, ring slot) and the next stage it fetches weights for
  int cs = 0, cc = c_begin, csg = 0, cslot = 0;
  int ds = 2, dc = c_begin + 2 / SG, dsg = 2 % SG, dslot = 2;
  const int nhalf = 2 * NS;
  for (int h = 0; h < nhalf; ++h) {
    if ((h & 1) == grp) {
      // ---------------- compute half-step: the MFMAs of stage cs on this group's tile
#ifdef PNSFM_PIPE_TRACE
      const long long tr_c0 = __builtin_readcyclecounter();
#endif
      const unsigned char* const wst = wbuf0 + cslot * stageB;
      const int tap0 = csg * G;
      const int gcount = (KK - tap0 < G) ? KK - tap0 : G;
      int ky = tap0 / a.KS, kx = tap0 - ky * a.KS;
      auto tapoff = [&]() -> int {
        const int o = (ky * a.PW + kx) * 16;
        if (++kx == a.KS) { kx = 0; ++ky; }
        return o;
      };
      // The partner wave of this SIMD issues no MFMAs during this half-step, so THIS wave has to keep the pipe fed by itself.  Taps per
      // stage is a template parameter and a stage is ONE straight-line block: the 3 (MT + NT) reads of tap j + 1 go out in front of
      // the 6 MT NT MFMAs of tap j (scheduling barriers keep hipcc from sinking them into the MFMA batch, where every MFMA would
      // wait for a read issued one instruction earlier) and have landed when that batch has been issued.  A chunk's ragged last
      // stage (KK % G taps; 1 of 13 stages of a 7x7 at G = 4) runs the same code: the slabs of its missing taps were fetched with
      // out-of-range offsets, i.e. ZERO-filled by the LDS-DMA (issue_weights_half), so their MFMAs add nothing: 2 of the 51 tap slots
      // of a 7x7 chunk at G = 3.  (ONE code path keeps the accumulators in place.  Measured alternatives, ISA checked: a second arm for
      // the ragged stage -- a generic loop, or a static one-tap block -- costs 32 v_mov_b64 of accumulator shuffling per stage at the
      // merge, plus spills in the (2,2) tile; uniform branches around the leading taps of one block make hipcc guard every fragment
      // read with a wait for the previous one, a ~250-cycle hole per tap.)
      // The barrier that ends this half-step sits IN FRONT of the last tap's MFMAs: that tap's fragments are in registers by then
      // (read underneath the previous tap's batch), so behind the barrier this wave issues 6 MT NT MFMAs that touch no LDS while the
      // partner -- released one MFMA batch early -- waits for its DMA, reads its first fragments and queues its own MFMAs behind
      // these: the head of every compute half-step (~600 cycles: vmcnt wait + 3 (MT + NT) reads + their latency), the barrier's
      // release latency and the pipe's drain are hidden under matrix work instead of leaving the pipe idle (tools/pp_trace.py).
      {
        int toff[G];
#pragma unroll
        for (int j = 0; j < G; ++j) toff[j] = j < gcount ? tapoff() : 0;
        Frag f0, f1;
        load_frag(f0, wst, 0, toff[0]);
        PNSFM_SCHED_FENCE();
#ifdef PNSFM_PIPE_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tr_head += __builtin_readcyclecounter() - tr_c0;
#endif
#pragma unroll
        for (int j = 0; j < G; j += 2) {
          if (j + 1 < G) { load_frag(f1, wst, j + 1, toff[j + 1]); PNSFM_SCHED_FENCE(); }
          if (j == G - 1 && h + 1 < nhalf) { PNSFM_TRC(tr_bwc, PNSFM_BARRIER_ALL()); PNSFM_SCHED_FENCE(); }
          mma(f0);
          PNSFM_SCHED_FENCE();
          if (j + 1 < G) {
            if (j + 2 < G) { load_frag(f0, wst, j + 2, toff[j + 2]); PNSFM_SCHED_FENCE(); }
            if (j + 1 == G - 1 && h + 1 < nhalf) { PNSFM_TRC(tr_bwc, PNSFM_BARRIER_ALL()); PNSFM_SCHED_FENCE(); }
            mma(f1);
            PNSFM_SCHED_FENCE();
          }
        }
      }
      ++cs;
      if (++csg == SG) { csg = 0; ++cc; }
      if (++cslot == 3) cslot = 0;
#ifdef PNSFM_PIPE_TRACE
      tr_comp += __builtin_readcyclecounter() - tr_c0;     // (includes the barrier wait, which is also counted in tr_bwc)
#endif
    } else {
      // ---------------- staging half-step: everything that is not matrix work for this group's next compute stage (cs)
#ifdef PNSFM_PIPE_TRACE
      const long long tr_s0 = __builtin_readcyclecounter();
#endif
      if (cs < NS) {
        if (csg == 0 && cc > c_begin) {            // first stage of a new chunk: its patch replaces the old one in place
          if (prefetch) write_items();
          else stage_sync(cc);
        }
        // (loads first, DMA second: hipcc guards every LDS read against a pending LDS-DMA of the same wave -- it cannot tell the
        // fragment reads from the stage being fetched -- so the next compute half-step opens with a vmcnt wait for everything issued
        // here; issued at the START of this half-step, the global loads have the partner's whole MFMA batch to land in)
        if (prefetch && csg == SG - 1 && cc + 1 < c_end) load_items(cc + 1);    // consumed by the staging half-step after the next compute
        if (ds < NS) issue_weights_half(dc, dsg * G, wbuf0 + dslot * stageB);
        ++ds;
        if (++dsg == SG) { dsg = 0; ++dc; }
        if (++dslot == 3) dslot = 0;
      }
#ifdef PNSFM_PIPE_TRACE
      tr_stage += __builtin_readcyclecounter() - tr_s0;
#endif
      if (h + 1 < nhalf) PNSFM_TRC(tr_bws, PNSFM_BARRIER_LDS());      // publishes the patch; loads and DMA stay in flight
    }
  }

#ifdef PNSFM_PIPE_TRACE
  const long long tr_epi = __builtin_readcyclecounter();
#endif
  conv_epilogue<MT, NT>(a, acc, b, co0, half, oy, ox, pvalid, (int)bz, lds_bias, tvalid ? t : -1, wave);
#ifdef PNSFM_PIPE_TRACE
  if (a.trace && lane == 0) {
    const long long tr_end = __builtin_readcyclecounter();
    long long* tt = a.trace + ((size_t)blockIdx.x * 8 + wave8) * 8;
    tt[0] = tr_comp; tt[1] = tr_head; tt[2] = tr_stage; tt[3] = tr_bwc; tt[4] = tr_end - tr_start; tt[5] = tr_loop - tr_start;
    tt[6] = tr_end - tr_epi; tt[7] = tr_bws * 4096 + NS * 2 + grp;
  }
#endif
#undef PNSFM_TRC
}
