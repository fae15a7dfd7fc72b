// conv2d.hip -- stride-1, zero-padded KxK convolution as an exact-fp32 MFMA implicit GEMM for gfx950.
//
// Replaces nn.ConstantPad2d(k//2) + nn.Conv2d(stride=1) of the reference's Conv2D / ResidualConv /
// InvDepth / PackLayerConv3d / UnpackLayerConv3d blocks
//   (/root/reference/packnet_sfm/networks/layers/packnet/layers01.py:28-36, 57-60, 115-121, 235-246, 274-281)
// and their autograd (dgrad = same kernel on a tap-flipped / transposed packed weight; wgrad below).
//
// GEMM view (forward):  Y[M = co][N = pixel] = sum_{tap, ci} Wp[tap][ci][co] * X[ci][pixel + off(tap)]
//   * one workgroup = 4 wave64; block tile = (32*MT output channels) x (4 waves * NT * 32 pixels);
//   * the input halo patch for CI (<=16) channels is staged ONCE in LDS and reused by all k*k taps
//     (and by every output channel of the tile) -- NCHW rows are read as coalesced row segments;
//   * per tap, a [CI][BM] weight slab (M fastest -> conflict-free ds_read_b32 for the MFMA A operand)
//     is double-buffered through registers while the previous tap's MFMAs run;
//   * v_mfma_f32_32x32x2_f32 (A: lane l holds W[m = l&31][k = l>>5], B: lane l holds X[k = l>>5][n = l&31]),
//     i.e. two input channels per instruction, exact fp32 (bitwise an fmaf chain) -- no TF32/bf16 shortcut;
//   * K (= taps x channels) can be split across blockIdx.z for the low-resolution layers whose pixel
//     count cannot fill 256 CUs (pack4/pack5: 480 pixels, K = 147456); the splits' partial outputs are summed in a
//     fixed order by a second kernel (conv_splitk_reduce_kernel; round 4 -- fp32 atomics before).
// Roofline: MFMA-bound. 2*Cout*Cin*k*k*B*H*W flop per launch against the 157.3 TFLOP/s fp32 matrix peak.
#include "pnsfm_common.h"
#include <cstring>
#include "../../include/pnsfm.h"

#include <algorithm>
#include <array>
#include <dlfcn.h>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <cstdio>

namespace pnsfm {

int conv_pick_MT(int Mc) { return (round_up(Mc, 64) == round_up(Mc, 32)) ? 2 : 1; }
int conv_pack_MP(int Mc) { return round_up(Mc, 32 * conv_pick_MT(Mc)); }
// K rows of a packed weight are padded (with zeros) to whole 16-channel K-chunks: the LDS-DMA of a weight slab then needs
// no validity test for the ragged last chunk
int conv_pack_KP(int Kc) { return round_up(Kc, 16); }

// ---- arithmetic of the forward / backward-data kernels ------------------------------------------------------------------
// 0: v_mfma_f32_32x32x2_f32 everywhere; 1 (default): fp32 rebuilt from exact bf16 splits on the bf16 matrix pipe
// (conv2d_bx3.h) for the shapes it supports.  The PACKED WEIGHT LAYOUT follows from (mode, shape), so weights packed under
// one mode must be re-packed after a switch (the Python side bumps its weight epoch in hip.functional.set_conv_math).
static int g_conv_math = -1;
static int conv_math() {
  if (g_conv_math < 0) {
    const char* e = getenv("PNSFM_CONV_MATH");
    g_conv_math = (e && (e[0] == 'f' || e[0] == '0')) ? 0 : 1;
  }
  return g_conv_math;
}
// K-channels >= 16 (one bf16 MFMA k-step is 16 channels).  1x1 layers (the residual shortcuts) joined in round 3: on the f32
// kernels they ran at 21-35 TFLOP/s -- one tap of 64-cycle MFMAs per staged chunk.
bool conv_bx3_supported(int Kc, int ks) { return Kc >= 16 && (ks >= 3 || ks == 1); }
static bool conv_use_bx3(int Kc, int ks) { return conv_math() == 1 && conv_bx3_supported(Kc, ks); }

static const size_t kMaxSmem = 64 * 1024;        // register-staged / patch-DMA variants (default dynamic-LDS limit)
static const size_t kMaxSmemPipe = 160 * 1024;   // pipelined variant: all of a CDNA4 CU's LDS (needs hipFuncSetAttribute)
static const size_t kPipeTwoBlocks = 80 * 1024;  // ... but prefer a K-chunk that lets two workgroups share the CU
static const size_t kPipeThreeBlocks = 53 * 1024;  // variant 6 of the split-bf16 kernels: three workgroups per CU

// Geometry of the forward / backward-data kernel for a FIXED choice of (NT, K-split); false if it does not fit.
// tm (split-bf16 kernels only): 0 = the classic tilings (32-wide 2-D tiles when W % 32 == 0, else linear runs whose patch spans
// whole rows), 1 = 16-wide rectangles (16 x 8*NT), 2 = row bands (W x floor(128*NT / W)).  On maps whose width is not a multiple
// of 32 (80, 40, 20) the linear runs drag (W + k - 1)-wide patch rows along: 410 staged pixels for 128 outputs at W = 80, too many
// for the register prefetch with NT = 2 -- the rectangles bring that to 180 / 324 and make NT = 2 usable there.
static bool conv_geom_fixed(int B, int Cin, int Cout, int H, int W, int ks, int NT, int want_split, ConvGeom& g, int DMA = 0,
                            int S = 1, int forceMT = 0, int tm = 0) {
  g.DMA = DMA;
  g.stem = 0;
  g.TW = g.TH = 0;
  if (tm != 0 && DMA < 3) return false;
  const int HW = H * W;
  g.MT = conv_pick_MT(Cout);
  // 32-row M tiles instead of 64 double the block count of a low-resolution layer WITHOUT split-K atomics; the packed
  // weight layout is the same as long as its padded M extent is a multiple of 64 anyway
  if (forceMT == 1 && g.MT == 2) g.MT = 1;
  const int BM = 32 * g.MT;
  g.MP = conv_pack_MP(Cout);
  g.KP = conv_pack_KP(Cin);
  g.CI = Cin <= 8 ? 8 : 16;    // K-chunk: multiple of 8 channels = whole batches of 4 MFMA k-steps
  g.mode = (W % 32 == 0) ? 0 : 1;
  g.NT = NT;
  if (tm == 1 || tm == 2) {
    // (on 32-multiple widths the default tiles already are 32 x 4*NT rectangles; the 16 x 8*NT ones are offered there for the 5x5 /
    // 7x7 layers only: 22 x 22 staged pixels instead of 14 x 38 for 256 outputs of a 7x7, which also fits the register prefetch)
    if (W % 32 == 0 && (tm == 2 || ks < 5)) return false;
    g.TW = tm == 1 ? 16 : W;
    g.TH = tm == 1 ? 8 * NT : (128 * NT) / W;
    if (g.TH < 1 || (tm == 1 && W < 16)) return false;
    if (g.TH > H) g.TH = H;
    // useful share of the tile grid: slots that hold a pixel of the image (a bad quantisation loses to the linear runs)
    const double fill = (double)H * W / ((double)ceil_div(H, g.TH) * ceil_div(W, g.TW) * 128.0 * NT);
    if (fill < 0.6) return false;
    g.mode = 2;
    g.tiles_x = ceil_div(W, g.TW);
    g.tiles_per_img = g.tiles_x * ceil_div(H, g.TH);
    g.PH = (g.TH - 1) * S + ks;
    g.PW = (g.TW - 1) * S + ks;
  } else if (g.mode == 0) {
    g.tiles_x = W / 32;
    g.tiles_per_img = g.tiles_x * ceil_div(H, 4 * NT);
    g.PH = (4 * NT - 1) * S + ks;
    g.PW = 31 * S + ks;
  } else {
    const int tile_px = 128 * NT;
    g.tiles_x = 0;
    g.tiles_per_img = ceil_div(HW, tile_px);
    int rows = (tile_px + W - 2) / W + 1;
    if (rows > H) rows = H;
    g.PH = (rows - 1) * S + ks;
    g.PW = (W - 1) * S + ks;
  }
  g.G = 0;
  g.PB = 1;
  if (DMA >= 3) {
    // split-bf16 variants (conv2d_bx3.h): 16-channel chunks; LDS = PB patch buffers of 3 planes + 2 weight stages of G taps.
    //   3: one patch buffer, 4: two patch buffers -- G the largest of {all taps (<= 9), one kernel row, 4, 3, 2, 1} that lets TWO
    //   workgroups share a CU;  5: two patch buffers and a whole kernel row per stage, one workgroup per CU if need be
    if (!conv_bx3_supported(Cin, ks)) return false;
    g.CI = 16;
    g.KP = round_up(Cin, 16);
    g.nchunks = g.KP / 16;
    g.PB = (DMA == 3 || DMA == 6 || DMA == 7) ? 1 : 2;
    if (DMA == 8) {
      // 1x1 without LDS (conv2d_bx3_1x1.h): runs of 128 NT consecutive pixels of one image, no patch, no weight stages
      if (ks != 1 || S != 1 || tm != 0) return false;
      if ((size_t)g.KP * HW * 4 >= (1ull << 31)) return false;      // 32-bit buffer offsets inside an image
      g.mode = 1;
      g.tiles_x = 0;
      g.tiles_per_img = ceil_div(HW, 128 * NT);
      g.PH = g.PW = 0;
      g.PB = 0;
      g.G = 1;
      g.smem_bytes = 0;
      if (want_split < 1) want_split = 1;
      if (want_split > g.nchunks) want_split = g.nchunks;
      const int cps8 = ceil_div(g.nchunks, want_split);
      g.splitK = ceil_div(g.nchunks, cps8);
      return true;
    }
    if (DMA == 6 && g.MT * NT > 2) return false;           // three workgroups per CU: <= 168 VGPRs only without the (2,2) tile
    const int KK = ks * ks;
    const size_t plane = (size_t)round_up(g.PH * g.PW * 32, 1024);
    const size_t patch = (size_t)g.PB * 3 * plane;
    // (+ 256 B: the M tile's 64 bias values, staged in the prologue for the epilogue)
    auto smem_bx3 = [&](int G) -> size_t {
      // 7 = ping-pong workgroup (conv2d_bx3pp.h): one patch buffer per group, a ring of three weight stages, one workgroup per CU
      if (DMA == 7) return 2 * patch + 3 * (size_t)G * g.MT * 3072 + 256;
      return patch + 2 * (size_t)G * g.MT * 3072 + 256;
    };
    const int cand[6] = {KK <= 9 ? KK : ks, ks, 4, 3, 2, 1};
    g.G = 0;
    if (DMA == 7) {
      // stages of <= 4 taps: the two groups alternate per stage, and a staging half-step (patch split + DMA issue) should not be
      // much shorter than the compute half-step it hides under; PNSFM_PP_G overrides (lab)
      // taps per stage: a compute half-step should not be shorter than the staging half-step it hides (patch split + DMA issue +
      // loads: 1 000 - 2 000 cycles, tools/pp_trace.py) -- 2 300 cycles are 3 taps of the (2,2) tile, 6 of (2,1) / (1,2), 12 of (1,1).
      // Built for G in {1, 2, 3, 4, 6, 9}; PNSFM_PP_G overrides the first choice (lab).
      static const int ppG = [] { const char* e = getenv("PNSFM_PP_G"); return e && e[0] ? atoi(e) : 0; }();
      const int per_tap = g.MT * NT;                     // 6 x this many MFMAs per tap and wave
      // (the (2,2) tile takes 3: its G = 4 and G = 6 builds spill ~30 VGPRs -- hipcc -Rpass-analysis=kernel-resource-usage -- and are
      // only reachable through PNSFM_PP_G)
      const int want = ppG > 0 ? ppG : (per_tap >= 4 ? 3 : (per_tap == 2 ? 6 : 9));
      const int cpp[6] = {9, 6, 4, 3, 2, 1};
      for (int i = 0; i < 6 && !g.G; ++i) {
        if (per_tap >= 4 && ppG <= 0 && (cpp[i] == 4 || cpp[i] == 6)) continue;
        if (cpp[i] <= want && cpp[i] <= KK && smem_bx3(cpp[i]) <= kMaxSmemPipe) g.G = cpp[i];
      }
      if (!g.G) return false;
    } else if (DMA == 5) {
      if (smem_bx3(cand[0]) <= kMaxSmemPipe) g.G = cand[0];
    } else if (DMA == 6) {
      for (int i = 0; i < 6 && !g.G; ++i)
        if (cand[i] <= KK && smem_bx3(cand[i]) <= kPipeThreeBlocks) g.G = cand[i];
      if (!g.G) return false;
    } else {
      for (int i = 0; i < 6 && !g.G; ++i)
        if (cand[i] <= KK && smem_bx3(cand[i]) <= kPipeTwoBlocks) g.G = cand[i];
    }
    for (int i = 0; i < 6 && !g.G; ++i)
      if (cand[i] <= KK && smem_bx3(cand[i]) <= kMaxSmemPipe) g.G = cand[i];
    if (!g.G) return false;
    g.smem_bytes = smem_bx3(g.G);
    if (want_split < 1) want_split = 1;
    if (want_split > g.nchunks) want_split = g.nchunks;
    const int cps3 = ceil_div(g.nchunks, want_split);
    g.splitK = ceil_div(g.nchunks, cps3);
    return true;
  }
  if (DMA == 2) {
    // pipelined variant: LDS = 2 patch buffers + 2 weight-slab buffers of one kernel row (G = ks taps) each.  The K-chunk
    // is the largest of {32 (1x1 only), 16, 8} channels that still lets two workgroups share a CU's 160 KB.
    g.G = (ks == 1) ? 1 : ks;
    auto smem_pipe = [&](int CI) -> size_t {
      return (2 * (size_t)round_up(CI * g.PH * g.PW, 64) + 2 * (size_t)g.G * CI * BM) * sizeof(float);
    };
    g.CI = Cin <= 8 ? 8 : 16;
    while (smem_pipe(g.CI) > kPipeTwoBlocks && g.CI > 8) g.CI /= 2;
    g.smem_bytes = smem_pipe(g.CI);
    g.nchunks = ceil_div(Cin, g.CI);
    if (g.PH * g.PW > 1024) return false;          // a channel's patch is fetched as <= 4 DMA groups of 256 elements
    if (want_split < 1) want_split = 1;
    if (want_split > g.nchunks) want_split = g.nchunks;
    const int cps2 = ceil_div(g.nchunks, want_split);
    g.splitK = ceil_div(g.nchunks, cps2);
    return g.smem_bytes <= kMaxSmemPipe;
  }
  // shrink the channel chunk if the halo patch of a very wide image does not fit
  auto smem_for = [&](int CI) -> size_t {
    const size_t patch = (size_t)CI * g.PH * g.PW;
    const size_t slabs = 2 * (size_t)CI * BM;
    if (DMA) return (2 * (size_t)round_up((int)patch, 64) + slabs) * sizeof(float);
    return (patch + 4 + slabs) * sizeof(float);
  };
  while (smem_for(g.CI) > kMaxSmem && g.CI > 8) g.CI /= 2;
  g.smem_bytes = smem_for(g.CI);
  g.nchunks = ceil_div(Cin, g.CI);
  // the depth networks' stem (3 -> C, 5x5, stride 1) on 32-wide tiles: its own kernel under variant 0 (conv2d_stem5_kernel below)
  if (DMA == 0 && Cin == 3 && ks == 5 && S == 1 && g.mode == 0) {
    g.stem = 1;
    g.smem_bytes = ((size_t)76 * BM + (size_t)3 * g.PH * g.PW) * sizeof(float);
  }
  if (want_split < 1) want_split = 1;
  if (want_split > g.nchunks) want_split = g.nchunks;
  const int cps = ceil_div(g.nchunks, want_split);
  g.splitK = ceil_div(g.nchunks, cps);
  return g.smem_bytes <= kMaxSmem;
}

// Default (un-tuned) choice: two pixel tiles per wave when that still gives >= 4 blocks per CU; split K only when the
// grid cannot give every CU ~6 blocks and each split keeps >= 2 channel chunks.  The runtime autotuner below
// (the analogue of the reference's `cudnn.benchmark = True`, trainers/horovod_trainer.py:19) refines this per shape.
static int g_default_dma = 0;   // un-tuned default variant (see pnsfm_set_conv_variant)
static int g_default_bx3 = 3;   // ... of the split-bf16 kernels (3..5)

ConvGeom conv_geom(int B, int Cin, int Cout, int H, int W, int ks, int S) {
  ConvGeom g;
  int NT = 2;
  int DA = g_default_dma;
  if (conv_use_bx3(Cin, ks)) {
    // un-tuned split-bf16 default: one patch buffer, two workgroups per CU; K split only when the tiles cannot fill the chip
    const int D3 = g_default_bx3;
    if (!conv_geom_fixed(B, Cin, Cout, H, W, ks, 2, 1, g, D3, S) || (long)B * g.tiles_per_img * (g.MP / (32 * g.MT)) < 512) NT = 1;
    if (conv_geom_fixed(B, Cin, Cout, H, W, ks, NT, 1, g, D3, S)) {
      const long blocks3 = (long)B * g.tiles_per_img * (g.MP / (32 * g.MT));
      int split3 = 1;
      if (blocks3 < 400 && g.nchunks >= 4) {
        split3 = (int)((512 + blocks3 - 1) / blocks3);
        if (split3 > g.nchunks / 2) split3 = g.nchunks / 2;
        if (split3 < 1) split3 = 1;
      }
      conv_geom_fixed(B, Cin, Cout, H, W, ks, NT, split3, g, D3, S);
      return g;
    }
    NT = 2;
  }
  if (!conv_geom_fixed(B, Cin, Cout, H, W, ks, 2, 1, g, DA, S) ||
      (long)B * g.tiles_per_img * (g.MP / (32 * g.MT)) < 1024) NT = 1;
  conv_geom_fixed(B, Cin, Cout, H, W, ks, NT, 1, g, DA, S);
  const long blocks = (long)B * g.tiles_per_img * (g.MP / (32 * g.MT));
  int split = 1;
  if (DA == 2) {
    // pipelined variant: one wave per SIMD already runs at the matrix rate, so K is split only when the tiles cannot
    // give (nearly) every CU one workgroup
    if (blocks < 200 && g.nchunks >= 4) {
      split = (int)((256 + blocks - 1) / blocks);
      if (split > g.nchunks / 2) split = g.nchunks / 2;
      if (split < 1) split = 1;
    }
  } else if (blocks < 4 * 256 && g.nchunks >= 4) {
    split = (int)((6 * 256 + blocks - 1) / blocks);
    if (split > g.nchunks / 2) split = g.nchunks / 2;
    if (split < 1) split = 1;
  }
  if (!conv_geom_fixed(B, Cin, Cout, H, W, ks, NT, split, g, DA, S) && DA) conv_geom_fixed(B, Cin, Cout, H, W, ks, NT, split, g, 0, S);
  return g;
}

// ---- runtime autotuning ---------------------------------------------------------------------------------------------
// First call for a new (kind, shape): time the candidate (NT, split) configurations on the caller's stream with
// hipEvents (this synchronises -- it happens during warm-up only) and remember the fastest.  PNSFM_AUTOTUNE=0 disables.
static int g_autotune = -1;
static std::map<std::array<int, 7>, std::array<int, 2>> g_tuned;    // database lines + the autotuner's own results
static std::map<std::array<int, 7>, std::array<int, 2>> g_pinned;   // pnsfm_tune_set: honoured in every build and mode
static std::mutex g_tune_mu;

// Tuning database (the analogue of MIOpen's user find-db): PNSFM_TUNE_DB=<file> loads earlier decisions at start-up
// and appends new ones, so that a later process (a profiling run, a resumed training job) launches no candidates.
// One text line per decision: kind B Cin Cout H W ks  cfg split.
static std::string g_tune_db;
static int g_shipped_entries = 0;   // entries read from the shipped tuned_gfx950.db (0: none / user database in use)

static int tune_db_load(const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  char line[256];
  int k[7], v[2], n = 0;
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#') continue;
    if (sscanf(line, "%d %d %d %d %d %d %d %d %d", &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &v[0], &v[1]) == 9) {
      g_tuned[{k[0], k[1], k[2], k[3], k[4], k[5], k[6]}] = {v[0], v[1]};
      ++n;
    }
  }
  fclose(f);
  return n;
}

static void tune_db_append(const std::array<int, 7>& k, const std::array<int, 2>& v) {
  if (g_tune_db.empty()) return;
  FILE* f = fopen(g_tune_db.c_str(), "a");
  if (!f) return;
  fprintf(f, "%d %d %d %d %d %d %d %d %d\n", k[0], k[1], k[2], k[3], k[4], k[5], k[6], v[0], v[1]);
  fclose(f);
}

// PNSFM_TUNE_LOG=<file>: every candidate the autotuner times is appended as
//   kind B Cin Cout H W ks | variant/config split ms       (kind 0/1 forward/backward-data (+10*stride), 2 weight gradient)
// -- the whole configuration landscape of a training step from one ordinary run.
static std::string g_tune_log;
static void tune_log(int kind, const std::array<int, 7>& k, int cfg, int split, float ms) {
  if (g_tune_log.empty()) return;
  FILE* f = fopen(g_tune_log.c_str(), "a");
  if (!f) return;
  fprintf(f, "%d %d %d %d %d %d %d | %d %d %.4f\n", k[0], k[1], k[2], k[3], k[4], k[5], k[6], cfg, split, ms);
  fclose(f);
  (void)kind;
}

static int g_wgrad_variant = -1;  // un-tuned weight-gradient kernel: -1 library default (split-bf16 where the arithmetic mode and the
                                  // shape allow, else generic), 0 generic, 1 tap-major, 2 split-bf16 (pnsfm_set_wgrad_variant)

static bool autotune_enabled() {
#ifdef PNSFM_EMU
  return false;
#else
  if (g_autotune < 0) {
    const char* e = getenv("PNSFM_AUTOTUNE");
    g_autotune = (e && e[0] == '0') ? 0 : 1;
    const char* tl = getenv("PNSFM_TUNE_LOG");
    if (tl && tl[0]) g_tune_log = tl;
    const char* db = getenv("PNSFM_TUNE_DB");
    if (db && db[0] && g_autotune == 1) {
      g_tune_db = db;
      std::lock_guard<std::mutex> lk(g_tune_mu);
      tune_db_load(db);
    } else if (g_autotune == 1 && !(db && db[0])) {
      // no user database: start from the decisions shipped next to the library (tuned_gfx950.db: the autotuner's own output
      // for the bench / reference configurations on an MI355X, read-only; shapes it does not list are timed as usual;
      // PNSFM_TUNE_DB=<file> replaces it, PNSFM_TUNE_DB= (empty) or a missing file means start from nothing)
      Dl_info info;
      if (!db && dladdr(reinterpret_cast<const void*>(&tune_db_load), &info) && info.dli_fname) {
        std::string path(info.dli_fname);
        const size_t slash = path.rfind('/');
        path = (slash == std::string::npos ? std::string(".") : path.substr(0, slash)) + "/tuned_gfx950.db";
        std::lock_guard<std::mutex> lk(g_tune_mu);
        g_shipped_entries = tune_db_load(path.c_str());
      }
    }
  }
  return g_autotune == 1;
#endif
}

// (g_tune_mu held)  Decision for a shape: a pinned entry always wins; database / autotuner entries count only while autotuning is
// on -- a test that switches it off and selects an un-tuned default variant gets exactly that variant, and the database
// survives such a test untouched (the setters used to clear the whole map, shipped decisions included: ADVICE r02).
static const std::array<int, 2>* tune_lookup(const std::array<int, 7>& key, bool tune) {
  auto p = g_pinned.find(key);
  if (p != g_pinned.end()) return &p->second;
  if (!tune) return nullptr;
  auto it = g_tuned.find(key);
  return it == g_tuned.end() ? nullptr : &it->second;
}

struct ConvArgs {
  const float* x;     // [B][Cin][H][W]   (multi-source: [B][C0][H][W], followed in K by x1 [B][C1][H][W] and x2 [B][Cin-C0-C1][H][W])
  const float* x1;
  const float* x2;
  int C0, C01;        // channels of x, of x + x1 (C0 = C01 = Cin for a single source)
  const float* wp;    // [KK][KP][MP]
  const float* bias;  // [Cout] or null
  float* y;           // [B][Cout][H][W]
  float* ws;          // K-split launches: [splitK][B][Cout][H][W] partial outputs (stream scratch), summed by conv_splitk_reduce_kernel
  int B, Cin, Cout, H, W, KS;   // H, W: OUTPUT size
  int S, Hi, Wi;                // stride (1 | 2) and INPUT size (Hi = H, Wi = W when S == 1)
  int CI, mode, tiles_x, tiles_per_img, PH, PW, KP, MP, nchunks, chunks_per_split, splitK;
  int TW, TH;         // mode 2 (split-bf16 kernels): tile width / height in output pixels
  int pstride;        // DMA variants: floats between the two patch buffers
  int G;              // pipelined / bx3 variants: taps per weight stage
  int PB;             // bx3 variants: patch buffers in LDS (1 | 2)
  int playout;        // bx3 variants: patch layout in LDS (1: half planes, conflict-free B fragments; 0: round 2's)
  int gx, gy, bmap;   // bx3 variants (1-D launch): pixel tiles, output-channel tiles, block order (pnsfm_common.h: block_map_mode)
  // round 5: y = conv + bias + addend -- the other gradient of a tensor with two consumers (skip connection, 1x1 shortcut), added where
  // the backward-data result is in registers instead of by a separate elementwise pass (3 passes over the tensor and a launch each).
  // [B][Cout][H][W] with `addend_bs` floats between samples (a channel slice of a wider tensor is fine).  null: off.
  const float* addend;
  size_t addend_bs;
#ifdef PNSFM_BX3_ABLATE
  int ablate;         // debug build only (tools/bx3_ablate.py): what-if switches of conv2d_bx3_kernel -- results are wrong
#endif
  float invPW, invPS;
#ifdef PNSFM_PIPE_TRACE
  long long* trace;   // debug build only (tools/pipe_trace.py): per wave {barrier wait, stage compute, prologue, epilogue} cycles
  int trace_flags;    // 1: skip the in-loop DMA (timing experiment: results are wrong)
#endif
};

// out-of-image / padded elements are read from here: selecting the POINTER (address | zero page) keeps every staging
// load unconditional; hipcc turns `ok ? x[off] : 0` into a branch around the load plus s_waitcnt vmcnt(0) per element
__device__ __attribute__((aligned(16))) float pnsfm_zero_page[64];


// ---- epilogue shared by the forward / backward-data kernels: D row = (r&3) + 8*(r>>2) + 4*half, col = l32.
// The 16*MT bias values a lane needs are fetched up front in ONE batch (the store loop used to fetch each one right
// before its store and wait for it: 32 dependent L2 round trips per workgroup).
// paths are separate loops so the compiler can stream the stores.
template <int MT, int NT, bool ADD = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MT][NT], int b, int co0, int half,
                                              const int (&oy)[NT], const int (&ox)[NT], const bool (&pvalid)[NT], int bz,
                                              const float* lds_bias = nullptr, int gn_tile = -1, int gn_wave = 0) {
  const int HW = a.H * a.W;
  // K-split launches: every split stores its partial tile into its own slab of the workspace (plain stores; no zero-fill, no
  // atomics) and conv_splitk_reduce_kernel adds the slabs in split order, bias included -- the result does not depend on which
  // workgroup finishes first (round 3 met the splits with fp32 atomics in a zero-filled y: 97 fills per step, and two runs of
  // one step differed in the last bits)
  const bool split = a.splitK > 1;
  float* yb = split ? a.ws + ((size_t)bz * a.B + b) * a.Cout * HW : a.y + (size_t)b * a.Cout * HW;
  if (a.bias != nullptr && !split) {
    float bval[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, co = co0 + row;
        // (split-bf16 kernels: the tile's bias values were staged in LDS by the prologue -- 32 L2 round trips at the very end of a
        // workgroup, in front of its stores, otherwise)
        bval[mt][r] = lds_bias ? lds_bias[row] : a.bias[co < a.Cout ? co : a.Cout - 1];
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] += bval[mt][r];
  }
  int poff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) poff[nt] = oy[nt] * a.W + ox[nt];
  if (ADD && a.addend != nullptr && !split) {       // (ADD = false: the f32 kernels, which no addend launch runs -- enqueue_conv)
    // one M tile's 16 x NT values per batch: the loads of a batch are in flight together (a uniform branch: launches without an
    // addend skip it); padded rows / pixels read element 0 of the sample's slice and are never stored
    const float* ab = a.addend + (size_t)b * a.addend_bs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float av[16][NT];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) av[r][nt] = ab[(co < a.Cout && pvalid[nt]) ? (size_t)co * HW + poff[nt] : (size_t)0];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] += av[r][nt];
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if (co < a.Cout && pvalid[nt]) yb[(size_t)co * HW + poff[nt]] = acc[mt][nt][r];
    }
  (void)gn_tile; (void)gn_wave;       // (round 5's GroupNorm statistics from this epilogue are gone: measured neutral twice)
}

// second stage of a K-split launch: y = bias + sum over the splits' slabs, in split order (bit-reproducible) [+ addend, last]
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int Z, int Cout, int HW, size_t total,
                                                                 const float* __restrict__ addend, size_t addend_bs) {
  const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= total) return;
  const size_t chw = (size_t)Cout * HW;
  if ((HW & 3) == 0) {                 // (then total % 4 == 0 and the four elements share a channel)
    float4 s = *reinterpret_cast<const float4*>(ws + i4);
    for (int z = 1; z < Z; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (size_t)z * total + i4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (bias) { const float bv = bias[(i4 / HW) % Cout]; s.x += bv; s.y += bv; s.z += bv; s.w += bv; }
    if (addend) {
      const size_t bb = i4 / chw;
      const float* ap = addend + bb * addend_bs + (i4 - bb * chw);       // (a slice's start need not be 16-byte aligned)
      s.x += ap[0]; s.y += ap[1]; s.z += ap[2]; s.w += ap[3];
    }
    *reinterpret_cast<float4*>(y + i4) = s;
  } else {
    for (size_t i = i4; i < i4 + 4 && i < total; ++i) {
      float s = ws[i];
      for (int z = 1; z < Z; ++z) s += ws[(size_t)z * total + i];
      if (bias) s += bias[(i / HW) % Cout];
      if (addend) { const size_t bb = i / chw; s += addend[bb * addend_bs + (i - bb * chw)]; }
      y[i] = s;
    }
  }
}

#include "conv2d_bx3.h"
#include "conv2d_bx3pp.h"
#include "conv2d_bx3_1x1.h"

// DMA = 0: the halo patch of a channel chunk is staged through registers (8 loads in flight per thread) between two
//          barriers;
// DMA = 1: the patch is DOUBLE-BUFFERED in LDS and the next chunk's patch is fetched with global_load_lds (LDS-DMA: no
//          VGPR round trip, no ds_write pass), issued in slices between the taps of the current chunk so that address
//          generation interleaves with the MFMAs and the fetch latency hides behind them -- this is what lets a grid
//          that is resident in a single round (the low-resolution layers) overlap staging with math.
// The autotuner times both per layer shape.
template <int MT, int NT, bool DMA>
__global__ void __launch_bounds__(256) conv2d_mfma_kernel(ConvArgs a) {
  PNSFM_DYN_SMEM(float, smem);
  constexpr int BM = 32 * MT;
  const int PS = a.PH * a.PW;
  const int ptotal = a.CI * PS;
  // LDS: [patch buffer(s)] [2 weight slabs of CI x BM]
  float* wbuf = smem + (DMA ? 2 * a.pstride : ((ptotal + 3) & ~3));

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
  const int P = a.KS >> 1, KK = a.KS * a.KS;
  const int H = a.H, W = a.W, HW = H * W;
  const int S = a.S, Hi = a.Hi, Wi = a.Wi, HWi = Hi * Wi;

  const int b = blockIdx.x / a.tiles_per_img;
  const int t = blockIdx.x - b * a.tiles_per_img;
  const int co0 = blockIdx.y * BM;
  const int c_begin = blockIdx.z * a.chunks_per_split;
  int c_end = c_begin + a.chunks_per_split;
  if (c_end > a.nchunks) c_end = a.nchunks;

  // ---- pixel-tile geometry: where this lane's output pixels are, and where they sit in the patch
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

  const float* xb = a.x + (size_t)b * a.Cin * HWi;
  // element `idx` of the patch of the chunk starting at channel ci0 -> its global address, or the zero page
  auto patch_src = [&](int ci0, int idx) -> const float* {
    const int cil = (int)(((float)idx + 0.5f) * a.invPS);
    const int e = idx - cil * PS;
    const int r = (int)(((float)e + 0.5f) * a.invPW);
    const int cc = e - r * a.PW;
    const int yy = py0 + r, xx = px0 + cc, ci = ci0 + cil;   // input coordinates
    const bool ok = idx < ptotal && ci < a.Cin && yy >= 0 && yy < Hi && xx >= 0 && xx < Wi;
    return ok ? xb + ((size_t)ci * HWi + yy * Wi + xx) : pnsfm_zero_page;
  };
  // weight slab loader geometry: slab = [CI][BM] floats, one float4 per thread (rows past the packed K extent are zero)
  const int wrow = tid / (BM / 4), wc4 = tid - wrow * (BM / 4);
  const bool wact = wrow < a.CI;
  const size_t tap_stride = (size_t)a.KP * a.MP;
  const int ksteps = a.CI >> 1;
  const int nld = (ptotal + 255) >> 8;           // DMA groups of 256 elements (64 per wave)
  const int per_tap = (nld + KK - 1) / KK;
  int dma_cur = 0;

  if constexpr (DMA) {   // first chunk's patch
    for (int ld = 0; ld < nld; ++ld) {
      const int idx = ld * 256 + tid;
      const float* src = patch_src(c_begin * a.CI, idx);
      if (idx < ptotal) pnsfm_glds4(src, smem + ld * 256 + wave * 64);
    }
  }

  for (int c = c_begin; c < c_end; ++c) {
    const int ci0 = c * a.CI;
    float* patch = smem + (DMA ? dma_cur * a.pstride : 0);
    if constexpr (!DMA) {
      __syncthreads();  // every wave is done reading the previous chunk's patch
      for (int base = tid; base < ptotal; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *patch_src(ci0, base + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (base + u * 256 < ptotal) patch[base + u * 256] = v[u];
      }
    }
    // ---- tap 0 weight slab
    const bool wok = wact && (ci0 + wrow) < a.KP;
    const float* wsrc = wok ? a.wp + ((size_t)(ci0 + wrow) * a.MP + co0 + wc4 * 4) : pnsfm_zero_page;
    const size_t wstep = wok ? tap_stride : 0;
    float4 wreg = *reinterpret_cast<const float4*>(wsrc);
    if (wact) *reinterpret_cast<float4*>(wbuf + wrow * BM + wc4 * 4) = wreg;
    if constexpr (DMA) pnsfm_dma_wait();
    __syncthreads();   // patch (DMA: landed, see pnsfm_dma_wait) and slab 0 visible; previous chunk fully consumed

    const bool dma_next = DMA && (c + 1 < c_end);
    float* dma_dst = smem + (dma_cur ^ 1) * a.pstride;
    int ld_next = 0;
    int ky = 0, kx = 0;
    for (int tap = 0; tap < KK; ++tap) {
      const int cur = tap & 1;
      if (tap + 1 < KK) wreg = *reinterpret_cast<const float4*>(wsrc + (size_t)(tap + 1) * wstep);
      const float* wb = wbuf + cur * a.CI * BM + half * BM + l32;
      const float* pb = patch + half * PS + ky * a.PW + kx;
      if (++kx == a.KS) { kx = 0; ++ky; }
      // batches of 4 k-steps (8 channels).  Software-pipelined: the LDS reads of batch 1 are issued BEFORE the MFMAs of
      // batch 0, so their latency hides under 4*MT*NT MFMAs instead of stalling the wave
      float av[2][4][MT], bv[2][4][NT];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[0][j][mt] = wb[j * 2 * BM + mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[0][j][nt] = pb[j * 2 * PS + boff[nt]];
      }
      if (ksteps == 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[1][j][mt] = wb[(4 + j) * 2 * BM + mt * 32];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bv[1][j][nt] = pb[(4 + j) * 2 * PS + boff[nt]];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = pnsfm_mfma_32x32x2(av[0][j][mt], bv[0][j][nt], acc[mt][nt]);
      if (ksteps == 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = pnsfm_mfma_32x32x2(av[1][j][mt], bv[1][j][nt], acc[mt][nt]);
      }
      if constexpr (DMA) {   // a slice of the NEXT chunk's patch rides behind this tap's MFMAs
        if (dma_next) {
          for (int i = 0; i < per_tap && ld_next < nld; ++i, ++ld_next) {
            const int idx = ld_next * 256 + tid;
            const float* src = patch_src(ci0 + a.CI, idx);
            if (idx < ptotal) pnsfm_glds4(src, dma_dst + ld_next * 256 + wave * 64);
          }
        }
      }
      if (tap + 1 < KK && wact) *reinterpret_cast<float4*>(wbuf + (cur ^ 1) * a.CI * BM + wrow * BM + wc4 * 4) = wreg;
      __syncthreads();
    }
    if constexpr (DMA) dma_cur ^= 1;
  }

  conv_epilogue<MT, NT, false>(a, acc, b, co0, half, oy, ox, pvalid, (int)blockIdx.z, nullptr, t, wave);
}


// ---- the stem: 3 input channels, 5x5, stride 1, 32-wide pixel tiles (round 6) ---------------------------------------------
// conv2d_mfma_kernel stages >= 8 channels per chunk and meets a barrier per tap: on the 3 -> 64 stem at 192x640 that is 4 k-steps of
// v_mfma_f32_32x32x2_f32 per tap for 1.5 k-steps of channels and 25 barriers for 200 MFMAs -- 145 us for 4.7 GFLOP (32 TFLOP/s), the
// first kernel of every step.  Here K is the flat (tap, channel) index: 75 -> 38 k-steps; the tile's WHOLE weight matrix
// (76 x 32 MT floats) and the 3-channel patch are staged once, and the k loop is 38 straight-line steps without a barrier: per k-step
// MT + NT ds_read_b32 for MT NT MFMAs, the patch offsets of the two k of a step compile-time constants.  The sum runs over the same
// non-zero terms in the same order as the generic kernel's (tap-major, channel inside): the same bits.
template <int MT, int NT>
__global__ void __launch_bounds__(256) conv2d_stem5_kernel(ConvArgs a) {
  constexpr int KS = 5, CIN = 3, K = KS * KS * CIN, KSTEPS = (K + 1) / 2, BM = 32 * MT;
  constexpr int PH = 4 * NT + KS - 1, PW = 32 + KS - 1, PS = PH * PW, PTOT = CIN * PS;
  PNSFM_DYN_SMEM(float, smem);
  float* const wl = smem;                        // [2 * KSTEPS][BM]: row k = tap * 3 + ci (row 75: zeros)
  float* const patch = smem + 2 * KSTEPS * BM;   // [CIN][PH][PW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
  const int H = a.H, W = a.W, HW = H * W;
  const int b = blockIdx.x / a.tiles_per_img;
  const int t = blockIdx.x - b * a.tiles_per_img;
  const int co0 = blockIdx.y * BM;
  const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
  const int y0 = ty * 4 * NT, x0 = tx * 32;
  const int py0 = y0 - KS / 2, px0 = x0 - KS / 2;
  int boff[NT], oy[NT], ox[NT];
  bool pvalid[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wave * NT + nt;
    oy[nt] = y0 + row;
    ox[nt] = x0 + l32;
    pvalid[nt] = oy[nt] < H;
    boff[nt] = row * PW + l32;
  }

  // ---- stage the patch (through registers: all loads in flight together) and the weight matrix
  {
    const float* xb = a.x + (size_t)b * CIN * HW;
    constexpr int NP = (PTOT + 255) / 256;
    float pv[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = i * 256 + tid;
      const int ci = idx / PS, e = idx - ci * PS, r = e / PW, c = e - r * PW;
      const int yy = py0 + r, xx = px0 + c;
      const bool ok = idx < PTOT && yy >= 0 && yy < H && xx >= 0 && xx < W;
      pv[i] = *(ok ? xb + ((size_t)ci * HW + yy * W + xx) : pnsfm_zero_page);
    }
    constexpr int NW = (2 * KSTEPS * (BM / 4) + 255) / 256;
    float4 wv[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = i * 256 + tid;
      const int k = idx / (BM / 4), c4 = idx - k * (BM / 4);
      const int tap = k / CIN, ci = k - tap * CIN;
      const bool ok = k < K;
      wv[i] = *reinterpret_cast<const float4*>(ok ? a.wp + (((size_t)tap * a.KP + ci) * a.MP + co0 + c4 * 4) : pnsfm_zero_page);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i)
      if (i * 256 + tid < PTOT) patch[i * 256 + tid] = pv[i];
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (i * 256 + tid < 2 * KSTEPS * (BM / 4)) *reinterpret_cast<float4*>(wl + (size_t)(i * 256 + tid) * 4) = wv[i];
  }
  __syncthreads();

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- 38 k-steps; lane half h of step s holds k = 2 s + h -> (tap, ci) -> patch offset ci PS + ky PW + kx (k = 75: any, its weights are 0)
  const float* const wb = wl + half * BM + l32;
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const int k0 = 2 * s, k1 = 2 * s + 1 < K ? 2 * s + 1 : 0;
    const int o0 = (k0 % CIN) * PS + ((k0 / CIN) / KS) * PW + (k0 / CIN) % KS;
    const int o1 = (k1 % CIN) * PS + ((k1 / CIN) / KS) * PW + (k1 / CIN) % KS;
    const float* pb = patch + (half ? o1 : o0);
    float av[MT], bv[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) av[mt] = wb[s * 2 * BM + mt * 32];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = pb[boff[nt]];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = pnsfm_mfma_32x32x2(av[mt], bv[nt], acc[mt][nt]);
  }

  conv_epilogue<MT, NT, false>(a, acc, b, co0, half, oy, ox, pvalid, 0, nullptr, t, wave);
}

// ---- pipelined variant (DMA == 2) ------------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 occupies a SIMD's matrix pipe for 64 cycles, so ONE wave per SIMD saturates it as long as that
// wave never waits: the kernel is organised so that nothing it waits for is on the critical path.
//   * LDS holds two halo patches (K-chunk c and c+1) and two weight slabs (stage s and s+1); a stage is one kernel ROW
//     of taps (G = ks taps x CI channels x BM output channels), so there is ONE barrier per kernel row instead of one per
//     tap, and no ds_write at all: both operands arrive by LDS-DMA (global_load_lds, 16 bytes per lane for the weights);
//   * at the top of stage s (after its barrier) the wave issues the DMA for stage s+1's slab and a slice of chunk c+1's
//     patch, then runs stage s's MFMAs; the next barrier's vmcnt(0) finds those copies long finished;
//   * operand fragments for the next batch of 4 k-steps are read from LDS before the current batch's MFMAs are issued.
// With the data movement off the critical path the low-resolution layers no longer need a K-split across workgroups
// (fills + fp32 atomics) to hide staging latency: 240 tiles on 256 CUs run at the matrix rate from one wave per SIMD.
template <int MT, int NT>
__global__ void __launch_bounds__(256) conv2d_pipe_kernel(ConvArgs a) {
  PNSFM_DYN_SMEM(float, smem);
  constexpr int BM = 32 * MT;
  const int PS = a.PH * a.PW;
  const int ptotal = a.CI * PS;
  const int G = a.G;
  const int wslab = G * a.CI * BM;              // floats per weight slab (a multiple of 256)
  float* const wbuf0 = smem + 2 * a.pstride;

  const int tid = threadIdx.x;
  // `wave` in an SGPR: LDS-DMA destinations (M0) and buffer descriptors derived from it must be provably wave-uniform,
  // otherwise hipcc wraps every DMA in a waterfall loop
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int P = a.KS >> 1, KK = a.KS * a.KS;
  const int H = a.H, W = a.W, HW = H * W;
  const int S = a.S, Hi = a.Hi, Wi = a.Wi, HWi = Hi * Wi;

  const int b = blockIdx.x / a.tiles_per_img;
  const int t = blockIdx.x - b * a.tiles_per_img;
  const int co0 = blockIdx.y * BM;
  const int c_begin = blockIdx.z * a.chunks_per_split;
  int c_end = c_begin + a.chunks_per_split;
  if (c_end > a.nchunks) c_end = a.nchunks;

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

  const float* xb = a.x + (size_t)b * a.Cin * HWi;
  const int SG = KK / G;                          // stages (kernel rows) per K-chunk
  const int nstage = (c_end - c_begin) * SG;
  const size_t tap_stride = (size_t)a.KP * a.MP;
  const int cilog = a.CI == 16 ? 4 : 3;

  // ---- LDS-DMA with buffer addressing: per DMA instruction the wave spends a handful of SCALAR instructions (re-based
  // descriptor, M0) and one buffer_load ... lds; nothing per element.
  // Patch of one channel = PS floats = NG groups of 256 (64 per wave).  A thread's elements sit at the same (row, col) of
  // the patch for every channel, so their byte offsets inside a channel image are computed ONCE; out-of-image elements
  // carry an out-of-range offset and the hardware writes zeros for them (also for a channel >= Cin: empty descriptor).
  constexpr int MAXG = 4;
  const int NG = (PS + 255) >> 8;
  unsigned pv[MAXG];
#pragma unroll
  for (int gg = 0; gg < MAXG; ++gg) {
    const int e = gg * 256 + tid;
    const int r = e / a.PW, cc = e - r * a.PW;
    const int yy = py0 + r, xx = px0 + cc;
    const bool ok = e < PS && yy >= 0 && yy < Hi && xx >= 0 && xx < Wi;
    pv[gg] = ok ? (unsigned)(yy * Wi + xx) * 4u : PNSFM_DMA_INVALID;
  }
  auto issue_channel = [&](int ci, int cil, float* dst) {        // channel `ci` of the image -> row `cil` of the patch buffer
    const pnsfm_dma_buf buf = pnsfm_make_dma_buf(xb + (unsigned)(ci * HWi), ci < a.Cin ? (long)HWi * 4 : 0);
#pragma unroll
    for (int gg = 0; gg < MAXG; ++gg)
      if (gg < NG && gg * 256 + tid < PS) pnsfm_dma4(buf, pv[gg], dst + cil * PS + gg * 256 + wave * 64);
  };
  // Weight slab of a stage = G taps x CI rows x BM floats, LDS layout [tap][ci][BM]; one 16-byte DMA piece = 256 floats =
  // 256/BM consecutive rows of one tap.  Lane part of the offset is constant; (tap, first row) go into the descriptor.
  constexpr int RPP = 256 / BM;                    // rows per piece
  const unsigned wv = (unsigned)((lane / (BM / 4)) * a.MP + (lane % (BM / 4)) * 4) * 4u;
  const int wpieces = wslab >> 8;
  const long wbytes = (long)KK * (long)tap_stride * 4;
  auto issue_wpiece = [&](int c, int g, int p, float* dst) {
    const int R = p * RPP;                          // first slab row of the piece
    const unsigned off = (unsigned)(g * G + (R >> cilog)) * (unsigned)tap_stride + (unsigned)((c * a.CI + (R & (a.CI - 1))) * a.MP + co0);
    const pnsfm_dma_buf buf = pnsfm_make_dma_buf(a.wp + off, wbytes - (long)off * 4);
    pnsfm_dma16(buf, wv, dst + (p << 8));
  };

#ifdef PNSFM_PIPE_TRACE
  long long tr_bar = 0, tr_comp = 0;
  const long long tr_start = __builtin_readcyclecounter();
#endif
  // prologue: chunk 0's patch and stage 0's slab
  if (nstage > 0) {
    for (int cil = 0; cil < a.CI; ++cil) issue_channel(c_begin * a.CI + cil, cil, smem);
    for (int p = wave; p < wpieces; p += 4) issue_wpiece(c_begin, 0, p, wbuf0);
  }
#ifdef PNSFM_PIPE_TRACE
  const long long tr_loop = __builtin_readcyclecounter();
#endif
  const int cps = (a.CI + SG - 1) / SG;            // patch channels prefetched per stage
  int c = c_begin, g = 0, pcur = 0;
  for (int s = 0; s < nstage; ++s) {
#ifdef PNSFM_PIPE_TRACE
    const long long tr0 = __builtin_readcyclecounter();
#endif
    pnsfm_dma_wait();
    __syncthreads();   // this wave's DMA has landed and so has the others': stage s's operands are in LDS, and
                       // everyone is done with stage s-1 (its slab buffer, and at a chunk boundary its patch buffer, are free)
#ifdef PNSFM_PIPE_TRACE
    const long long tr1 = __builtin_readcyclecounter();
    tr_bar += tr1 - tr0;
#endif
    float* const patch = smem + pcur * a.pstride;
    const float* const wcur = wbuf0 + (s & 1) * wslab;
    // DMA work of this stage, handed out in slices between the MFMA batches below: the slab of stage s+1 and this
    // stage's share of chunk c+1's patch channels
    const bool more = s + 1 < nstage;
    const int gn = (g + 1 == SG) ? 0 : g + 1, cn = (gn == 0) ? c + 1 : c;
    float* const wnext = wbuf0 + ((s + 1) & 1) * wslab;
    float* const pnext = smem + (pcur ^ 1) * a.pstride;
    int wp_next = more ? wave : wpieces;            // next weight piece of this wave
    int ch_next = g * cps;                          // next patch channel (local) ...
    const int ch_end = (more && c + 1 < c_end) ? ((ch_next + cps < a.CI) ? ch_next + cps : a.CI) : ch_next;
    // ---- stage s: G taps x CI/8 batches of 4 k-steps; fragments of batch q+1 are read before batch q's MFMAs
    const int nb = a.CI >> 3, nq = G * nb;
    // front-loaded: everything is issued during the FIRST HALF of the stage's batches, because the barrier that ends the
    // stage drains vmcnt(0) -- a copy issued in the last batch would expose its whole latency (~1-2 us) there
    const int nq_issue = nq > 1 ? nq / 2 : 1;
    const int wpb = ((wpieces + 3) / 4 + nq_issue - 1) / nq_issue, cpb = (cps + nq_issue - 1) / nq_issue;
    auto issue_slice = [&]() {
#ifdef PNSFM_PIPE_TRACE
      if (a.trace_flags & 1) { wp_next = wpieces; ch_next = ch_end; return; }
#endif
      for (int i = 0; i < wpb && wp_next < wpieces; ++i, wp_next += 4) issue_wpiece(cn, gn, wp_next, wnext);
      for (int i = 0; i < cpb && ch_next < ch_end; ++i, ++ch_next) issue_channel((c + 1) * a.CI + ch_next, ch_next, pnext);
    };
    int ky = (g * G) / a.KS, kx = (g * G) - ky * a.KS;
    const float* wb = wcur + half * BM + l32;
    const float* pb = patch + half * PS + ky * a.PW + kx;
    float av[2][4][MT], bv[2][4][NT];
    auto load = [&](int buf, const float* wq, const float* pq) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[buf][j][mt] = wq[j * 2 * BM + mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[buf][j][nt] = pq[j * 2 * PS + boff[nt]];
      }
    };
    auto mma = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = pnsfm_mfma_32x32x2(av[buf][j][mt], bv[buf][j][nt], acc[mt][nt]);
    };
    // pointers of batch q: wq = wb + (tap_in_row * CI + 8*jb) * BM, pq = pb(tap) + 8*jb*PS
    int jb = 0;
    const float* wq = wb;
    const float* pq = pb;
    auto advance = [&]() {
      if (++jb == nb) {                            // next tap of the row
        jb = 0;
        wq += 8 * BM;
        if (++kx == a.KS) { kx = 0; ++ky; }
        pb = patch + half * PS + ky * a.PW + kx;
        pq = pb;
      } else {
        wq += 8 * BM;
        pq += 8 * PS;
      }
    };
    load(0, wq, pq);
    int q = 0;
    for (; q + 2 <= nq; q += 2) {
      advance();
      issue_slice();
      load(1, wq, pq);
      mma(0);
      issue_slice();
      if (q + 2 < nq) {
        advance();
        load(0, wq, pq);
      }
      mma(1);
    }
    if (q < nq) { issue_slice(); mma(0); }         // odd number of batches: the last one sits in buffer 0
    // whatever the slices did not cover (short stages)
    for (; wp_next < wpieces; wp_next += 4) issue_wpiece(cn, gn, wp_next, wnext);
    for (; ch_next < ch_end; ++ch_next) issue_channel((c + 1) * a.CI + ch_next, ch_next, pnext);
#ifdef PNSFM_PIPE_TRACE
    tr_comp += __builtin_readcyclecounter() - tr1;
#endif
    if (++g == SG) { g = 0; ++c; pcur ^= 1; }
  }
#ifdef PNSFM_PIPE_TRACE
  const long long tr_epi = __builtin_readcyclecounter();
#endif

  conv_epilogue<MT, NT, false>(a, acc, b, co0, half, oy, ox, pvalid, (int)blockIdx.z, nullptr, t, wave);
#ifdef PNSFM_PIPE_TRACE
  if (a.trace && lane == 0) {
    const long long tr_end = __builtin_readcyclecounter();
    long long* t = a.trace + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 4 + wave) * 6;
    t[0] = tr_bar; t[1] = tr_comp; t[2] = tr_loop - tr_start; t[3] = tr_end - tr_epi; t[4] = tr_end - tr_start; t[5] = nstage;
  }
#endif
}

#ifdef PNSFM_BX3_ABLATE
static int g_ablate = 0, g_smem_pad = 0;
extern "C" int pnsfm_debug_set_ablate(int f) { g_ablate = f; return 0; }
extern "C" int pnsfm_debug_set_smem_pad(int bytes) { g_smem_pad = bytes; return 0; }   // extra LDS per workgroup: fewer workgroups per CU
#endif
#ifdef PNSFM_PIPE_TRACE
static long long* g_trace_buf = nullptr;
static int g_trace_flags = 0;
extern "C" int pnsfm_debug_set_trace(void* p) { g_trace_buf = (long long*)p; return 0; }
extern "C" int pnsfm_debug_set_trace_flags(int f) { g_trace_flags = f; return 0; }
#endif

// extras of a launch's epilogue: the addend of ConvArgs (round 5; null: none).  (The GroupNorm statistics that rode along here in round 5
// -- pnsfm_conv2d_forward_gn -- were measured neutral twice and removed in round 6.)
struct ConvGnOut {
  const float* addend;
  size_t addend_bs;
};

static int enqueue_conv(const ConvGeom& g, const float* x, const float* wp, const float* bias, float* y, int B, int Cin,
                        int Cout, int H, int W, int ks, hipStream_t stream, const char* what, int S, int Hi, int Wi,
                        const ConvSrc* ms = nullptr, ConvGnOut* gn = nullptr) {
  ConvArgs a;
  a.addend = gn ? gn->addend : nullptr; a.addend_bs = gn ? gn->addend_bs : 0;
  a.x = x; a.wp = wp; a.bias = bias; a.y = y;
  a.x1 = ms ? ms->x1 : nullptr; a.x2 = ms ? ms->x2 : nullptr;
  a.C0 = ms ? ms->C0 : Cin; a.C01 = ms ? ms->C0 + ms->C1 : Cin;
  if (ms && g.DMA < 3) { set_error("%s: several input tensors need the split-bf16 kernels", what); return -1; }
  if (ms && g.DMA == 8) { set_error("%s: the LDS-free 1x1 kernel reads one input tensor", what); return -1; }
  if (a.addend && g.DMA < 3) { set_error("%s: an addend needs the split-bf16 kernels (>= 16 channels of dy, 'bx3' arithmetic)", what); return -1; }
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.KS = ks;
  a.S = S; a.Hi = Hi; a.Wi = Wi;
  a.CI = g.CI; a.mode = g.mode; a.tiles_x = g.tiles_x; a.tiles_per_img = g.tiles_per_img;
  a.TW = g.TW; a.TH = g.TH;
  a.PH = g.PH; a.PW = g.PW; a.KP = g.KP; a.MP = g.MP; a.nchunks = g.nchunks;
  a.chunks_per_split = ceil_div(g.nchunks, g.splitK); a.splitK = g.splitK;
  a.invPW = 1.0f / (float)g.PW;
  a.invPS = 1.0f / (float)(g.PH * g.PW);
  a.pstride = round_up(g.CI * g.PH * g.PW, 64);
  a.G = g.G;
  a.PB = g.PB;
  if (g.DMA >= 3) a.pstride = round_up(g.PH * g.PW * 32, 1024);   // bytes of one bf16 piece plane of the patch
#ifdef PNSFM_PIPE_TRACE
  a.trace = g_trace_buf;
  a.trace_flags = g_trace_flags;
#endif
  // K-split launch: partial outputs in the stream's scratch buffer (api.hip), summed in split order by conv_splitk_reduce_kernel
  const size_t out_elems = (size_t)B * Cout * H * W;
  ScratchLease lease(stream, g.splitK > 1 ? (size_t)g.splitK * out_elems * sizeof(float) : 0);
  a.ws = nullptr;
  if (g.splitK > 1) {
    if (!lease.p) return -1;
    a.ws = lease.as<float>();
  }
  dim3 grid(g.DMA == 7 ? ceil_div(B * g.tiles_per_img, 2) : B * g.tiles_per_img, g.MP / (32 * g.MT), g.splitK);   // (7: pairs of pixel tiles)
  a.gx = (int)grid.x; a.gy = (int)grid.y; a.bmap = block_map_mode();
  if (a.bmap == 2 && g.DMA >= 3) {
    // weight-heavy launch (the packed split weights outweigh the input tensor): pixel tile fastest inside an XCD's range (conv2d_bx3.h)
    // PNSFM_BLOCK_MAP_WEIGHTS = 0 (off) | r: pixel tile fastest when weights > r x input (default 1)
    static const double wratio = [] { const char* e = getenv("PNSFM_BLOCK_MAP_WEIGHTS"); return e && e[0] ? atof(e) : 1.0; }();
    const bool wmap = wratio > 0.0;
    const double wbytes = 6.0 * Cout * (double)Cin * ks * ks, xbytes = 4.0 * B * (double)Cin * Hi * Wi * wratio;
    if (wmap && wbytes > xbytes && grid.x > 1) a.bmap = 3;
  }
  a.playout = 1;
#ifdef PNSFM_BX3_ABLATE
  a.ablate = g_ablate;
#endif
  const dim3 grid1(grid.x * grid.y * grid.z);      // split-bf16 kernels: 1-D launch, block order decoded in the kernel
#define PNSFM_CONV_DISPATCH(DMAv)                                                                                 \
  do {                                                                                                             \
    if (g.MT == 2 && g.NT == 2) PNSFM_LAUNCH((conv2d_mfma_kernel<2, 2, DMAv>), grid, dim3(256), g.smem_bytes, stream, a);      \
    else if (g.MT == 2 && g.NT == 1) PNSFM_LAUNCH((conv2d_mfma_kernel<2, 1, DMAv>), grid, dim3(256), g.smem_bytes, stream, a); \
    else if (g.MT == 1 && g.NT == 2) PNSFM_LAUNCH((conv2d_mfma_kernel<1, 2, DMAv>), grid, dim3(256), g.smem_bytes, stream, a); \
    else PNSFM_LAUNCH((conv2d_mfma_kernel<1, 1, DMAv>), grid, dim3(256), g.smem_bytes, stream, a);                 \
  } while (0)
  if (g.DMA >= 3) {
#ifndef PNSFM_EMU
#define PNSFM_BX3_ATTR(MTv, NTv)                                                                                   \
    do {                                                                                                           \
      static unsigned long long done = 0; /* one bit per device */                                                 \
      if (g.smem_bytes > 64 * 1024 &&                                                                              \
          ensure_lds_limit(reinterpret_cast<const void*>(&conv2d_bx3_kernel<MTv, NTv, 2>), &done, (int)kMaxSmemPipe, what)) \
        return -1;                                                                                                 \
    } while (0)
#else
#define PNSFM_BX3_ATTR(MTv, NTv) do {} while (0)
#endif
    if (g.DMA == 8) {         // 1x1 without LDS (conv2d_bx3_1x1.h)
      if (g.MT == 2 && g.NT == 2) PNSFM_LAUNCH((conv1x1_bx3_kernel<2, 2>), grid1, dim3(256), 0, stream, a);
      else if (g.MT == 2 && g.NT == 1) PNSFM_LAUNCH((conv1x1_bx3_kernel<2, 1>), grid1, dim3(256), 0, stream, a);
      else if (g.MT == 1 && g.NT == 2) PNSFM_LAUNCH((conv1x1_bx3_kernel<1, 2>), grid1, dim3(256), 0, stream, a);
      else PNSFM_LAUNCH((conv1x1_bx3_kernel<1, 1>), grid1, dim3(256), 0, stream, a);
    }
    else if (g.DMA == 7) {         // ping-pong workgroup: 512 threads, up to 160 KB of LDS, taps per stage a template parameter (conv2d_bx3pp.h)
#ifndef PNSFM_EMU
#define PNSFM_PP_ATTR(MTv, NTv, Gv)                                                                                \
      do {                                                                                                         \
        static unsigned long long done = 0; /* one bit per device */                                               \
        if (g.smem_bytes > 64 * 1024 &&                                                                            \
            ensure_lds_limit(reinterpret_cast<const void*>(&conv2d_bx3pp_kernel<MTv, NTv, Gv>), &done, (int)kMaxSmemPipe, what)) \
          return -1;                                                                                               \
      } while (0)
#else
#define PNSFM_PP_ATTR(MTv, NTv, Gv) do {} while (0)
#endif
#define PNSFM_PP_LAUNCH(MTv, NTv, Gv) do { PNSFM_PP_ATTR(MTv, NTv, Gv); PNSFM_LAUNCH((conv2d_bx3pp_kernel<MTv, NTv, Gv>), grid1, dim3(512), g.smem_bytes, stream, a); } while (0)
#define PNSFM_PP_G(MTv, NTv)                                                                                       \
      do {                                                                                                         \
        if (g.G == 9) PNSFM_PP_LAUNCH(MTv, NTv, 9);                                                                \
        else if (g.G == 6) PNSFM_PP_LAUNCH(MTv, NTv, 6);                                                           \
        else if (g.G == 4) PNSFM_PP_LAUNCH(MTv, NTv, 4);                                                           \
        else if (g.G == 3) PNSFM_PP_LAUNCH(MTv, NTv, 3);                                                           \
        else if (g.G == 2) PNSFM_PP_LAUNCH(MTv, NTv, 2);                                                           \
        else PNSFM_PP_LAUNCH(MTv, NTv, 1);                                                                         \
      } while (0)
      if (a.playout == 0) { set_error("%s: the ping-pong kernel needs the half-plane patch layout", what); return -1; }
      if (g.G != 1 && g.G != 2 && g.G != 3 && g.G != 4 && g.G != 6 && g.G != 9) { set_error("%s: the ping-pong kernel is built for 1, 2, 3, 4, 6 or 9 taps per stage", what); return -1; }
      if (g.MT == 2 && g.NT == 2) PNSFM_PP_G(2, 2);
      else if (g.MT == 2 && g.NT == 1) PNSFM_PP_G(2, 1);
      else if (g.MT == 1 && g.NT == 2) PNSFM_PP_G(1, 2);
      else PNSFM_PP_G(1, 1);
#undef PNSFM_PP_G
#undef PNSFM_PP_LAUNCH
#undef PNSFM_PP_ATTR
    }
    else if (g.DMA == 6) {         // three workgroups per CU (<= 53 KB of LDS each: no opt-in needed)
      if (g.MT == 2) PNSFM_LAUNCH((conv2d_bx3_kernel<2, 1, 3>), grid1, dim3(256), g.smem_bytes, stream, a);
      else if (g.NT == 2) PNSFM_LAUNCH((conv2d_bx3_kernel<1, 2, 3>), grid1, dim3(256), g.smem_bytes, stream, a);
      else PNSFM_LAUNCH((conv2d_bx3_kernel<1, 1, 3>), grid1, dim3(256), g.smem_bytes, stream, a);
    }
    else if (g.MT == 2 && g.NT == 2) { PNSFM_BX3_ATTR(2, 2); PNSFM_LAUNCH((conv2d_bx3_kernel<2, 2, 2>), grid1, dim3(256), g.smem_bytes, stream, a); }
    else if (g.MT == 2 && g.NT == 1) { PNSFM_BX3_ATTR(2, 1); PNSFM_LAUNCH((conv2d_bx3_kernel<2, 1, 2>), grid1, dim3(256), g.smem_bytes, stream, a); }
    else if (g.MT == 1 && g.NT == 2) { PNSFM_BX3_ATTR(1, 2); PNSFM_LAUNCH((conv2d_bx3_kernel<1, 2, 2>), grid1, dim3(256), g.smem_bytes, stream, a); }
    else { PNSFM_BX3_ATTR(1, 1); PNSFM_LAUNCH((conv2d_bx3_kernel<1, 1, 2>), grid1, dim3(256), g.smem_bytes, stream, a); }
#undef PNSFM_BX3_ATTR
  } else if (g.DMA == 2) {
#ifndef PNSFM_EMU
    // more than 64 KB of dynamic LDS needs an explicit opt-in per kernel (once)
#define PNSFM_PIPE_ATTR(MTv, NTv)                                                                                  \
    do {                                                                                                           \
      static unsigned long long done = 0; /* one bit per device */                                                 \
      if (g.smem_bytes > 64 * 1024 &&                                                                              \
          ensure_lds_limit(reinterpret_cast<const void*>(&conv2d_pipe_kernel<MTv, NTv>), &done, (int)kMaxSmemPipe, what)) \
        return -1;                                                                                                 \
    } while (0)
#else
#define PNSFM_PIPE_ATTR(MTv, NTv) do {} while (0)
#endif
    if (g.MT == 2 && g.NT == 2) { PNSFM_PIPE_ATTR(2, 2); PNSFM_LAUNCH((conv2d_pipe_kernel<2, 2>), grid, dim3(256), g.smem_bytes, stream, a); }
    else if (g.MT == 2 && g.NT == 1) { PNSFM_PIPE_ATTR(2, 1); PNSFM_LAUNCH((conv2d_pipe_kernel<2, 1>), grid, dim3(256), g.smem_bytes, stream, a); }
    else if (g.MT == 1 && g.NT == 2) { PNSFM_PIPE_ATTR(1, 2); PNSFM_LAUNCH((conv2d_pipe_kernel<1, 2>), grid, dim3(256), g.smem_bytes, stream, a); }
    else { PNSFM_PIPE_ATTR(1, 1); PNSFM_LAUNCH((conv2d_pipe_kernel<1, 1>), grid, dim3(256), g.smem_bytes, stream, a); }
#undef PNSFM_PIPE_ATTR
  } else if (g.DMA) PNSFM_CONV_DISPATCH(true);
  else if (g.stem) {
    if (g.MT == 2 && g.NT == 2) PNSFM_LAUNCH((conv2d_stem5_kernel<2, 2>), grid, dim3(256), g.smem_bytes, stream, a);
    else if (g.MT == 2 && g.NT == 1) PNSFM_LAUNCH((conv2d_stem5_kernel<2, 1>), grid, dim3(256), g.smem_bytes, stream, a);
    else if (g.MT == 1 && g.NT == 2) PNSFM_LAUNCH((conv2d_stem5_kernel<1, 2>), grid, dim3(256), g.smem_bytes, stream, a);
    else PNSFM_LAUNCH((conv2d_stem5_kernel<1, 1>), grid, dim3(256), g.smem_bytes, stream, a);
  }
  else PNSFM_CONV_DISPATCH(false);
#undef PNSFM_CONV_DISPATCH
  int rc = check_launch(what);
  if (!rc && g.splitK > 1) {
    PNSFM_LAUNCH(conv_splitk_reduce_kernel, dim3((unsigned)ceil_div_sz(out_elems, 1024)), dim3(256), 0, stream, (const float*)a.ws, bias, y,
                 g.splitK, Cout, H * W, out_elems, a.addend, a.addend_bs);
    rc = check_launch(what);
  }
  return rc;
}

#ifndef PNSFM_EMU
// time back-to-back executions of fn() on `stream`; returns ms per execution (or < 0 on error).  `reps` executions are the
// minimum; short kernels are repeated until ~0.4 ms have been measured (<= 12 executions): with 2 executions of a 50 us
// kernel the run-to-run noise (+-10 %) used to pick different configurations from one process to the next.
template <class F>
static float time_on_stream(hipStream_t stream, int reps, F fn) {
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0) { if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1.f; }
  if (fn() != 0) return -1.f;                     // warm-up (also faults in code objects)
  float total = 0.f;
  int done = 0;
  while (true) {
    if (hipEventRecord(e0, stream) != hipSuccess) return -1.f;
    for (int i = 0; i < reps; ++i) if (fn() != 0) return -1.f;
    if (hipEventRecord(e1, stream) != hipSuccess) return -1.f;
    if (hipEventSynchronize(e1) != hipSuccess) return -1.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.f;
    total += ms;
    done += reps;
    if (total >= 0.4f || done >= 12) break;
  }
  return total / done;
}
#endif

// configuration of the calling thread's most recent forward / backward-data launch (pnsfm_conv2d_last_config)
static thread_local std::array<int, 8> g_last_conv = {-1, 0, 0, 0, 0, 0, 0, 0};

static int launch_conv(const float* x, const float* wp, const float* bias, float* y, int B, int Cin, int Cout,
                       int H, int W, int ks, hipStream_t stream, const char* what, int kind_tag, int S = 1, int Hi = 0,
                       int Wi = 0, const ConvSrc* ms = nullptr, ConvGnOut* gn = nullptr) {
  if (S == 1) { Hi = H; Wi = W; }
  if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) { set_error("%s: bad shape", what); return -1; }
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("%s: unsupported kernel size %d", what, ks); return -1; }
  if (S != 1 && S != 2) { set_error("%s: unsupported stride %d", what, S); return -1; }
  if (ks == 1 && S == 1 && (H * W) % 32 == 0 && W % 32 != 0) {
    // no halo: any pixel order works, so a map whose width is not a multiple of 32 is tiled as 32-wide rows of the flattened
    // image (whole 2-D tiles instead of linear runs whose patches span full-width rows)
    H = (H * W) / 32; W = 32; Hi = H; Wi = W;
  }
  ConvGeom g = conv_geom(B, Cin, Cout, H, W, ks, S);
  if (g.smem_bytes > (g.DMA >= 2 ? kMaxSmemPipe : kMaxSmem)) { set_error("%s: image too wide for the LDS halo patch (W=%d)", what, W); return -1; }
  {
    // tuned / pinned configuration of this shape: the autotuner's result, a PNSFM_TUNE_DB line or pnsfm_tune_set (tests pin
    // configurations the un-tuned heuristics would not pick; that works in every build, the timing search needs a GPU)
    const bool bx3 = conv_use_bx3(Cin, ks);
    const std::array<int, 7> key = {kind_tag + 10 * S + (bx3 ? 100 : 0), B, Cin, Cout, H, W, ks};
    const bool tune = autotune_enabled();        // (first call: reads the environment and loads PNSFM_TUNE_DB under the lock)
    std::lock_guard<std::mutex> lk(g_tune_mu);
    const std::array<int, 2>* dec = tune_lookup(key, tune);
#ifndef PNSFM_EMU
    // (a shape first seen inside a hipGraph capture cannot be timed -- timing synchronises: un-tuned default, not cached)
    if (!dec && tune && !stream_capturing(stream)) {
      static const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
      float best_ms = 1e30f;
      std::array<int, 2> best = {g.NT | (g.DMA << 4), g.splitK};
      const int nMT = (conv_pick_MT(Cout) == 2) ? 2 : 1;
      const int nTM = (bx3 && (W % 32 != 0 || ks >= 5)) ? (W % 32 != 0 ? 3 : 2) : 1;       // tile modes (rectangles / row bands) exist for the split-bf16 kernels
      // PNSFM_PP: which launches may take the ping-pong workgroup (variant 7): bit 0 forward, bit 1 backward-data (default 3: both)
      static const int pp_mask = [] { const char* e = getenv("PNSFM_PP"); return (e && e[0]) ? atoi(e) : 3; }();
      const bool pp_on = ((pp_mask >> (kind_tag & 1)) & 1) != 0;
      // LDS plans: f32 0..2, split-bf16 3..8 (6 = three workgroups per CU, 7 = ping-pong workgroup, 8 = 1x1 without LDS)
      int vars[6], nVar = 0;
      if (bx3) {
        for (int v = 3; v <= 6; ++v) vars[nVar++] = v;
        if (pp_on) vars[nVar++] = 7;
        if (ks == 1 && S == 1 && !ms) vars[nVar++] = 8;
      } else {
        for (int v = 0; v <= 2; ++v) vars[nVar++] = v;
      }
      for (int cfgt = 0; cfgt < 2 * nVar * nMT * nTM; ++cfgt) {
        const int cfg = cfgt % (2 * nVar * nMT), tm = cfgt / (2 * nVar * nMT);
        const int NT = 2 - (cfg & 1), DA = vars[(cfg >> 1) % nVar], fMT = cfg / (2 * nVar);
        int last_split = -1;
        for (int want : kSplits) {
          ConvGeom c;
          if (!conv_geom_fixed(B, Cin, Cout, H, W, ks, NT, want, c, DA, S, fMT, tm)) break;
          if (c.splitK == last_split) continue;
          last_split = c.splitK;
          const long blocks = (long)B * c.tiles_per_img * (c.MP / (32 * c.MT)) * c.splitK;
          if (c.splitK > 1 && blocks > 24L * 256 * 4) break;      // already far more blocks than the chip holds
          const float tms = time_on_stream(stream, 2, [&]() { return enqueue_conv(c, x, wp, bias, y, B, Cin, Cout, H, W, ks, stream, what, S, Hi, Wi, ms); });
          tune_log(0, key, NT | (DA << 4) | (fMT << 8) | (tm << 9), c.splitK, tms);
          if (tms > 0.f && tms < best_ms) { best_ms = tms; best = {NT | (DA << 4) | (fMT << 8) | (tm << 9), c.splitK}; }
        }
      }
      dec = &g_tuned.emplace(key, best).first->second;
      tune_db_append(key, best);
    }
#endif
    if (dec) {
      ConvGeom t;
      const int DA = ((*dec)[0] >> 4) & 15;
      if ((DA >= 3) == bx3 &&
          conv_geom_fixed(B, Cin, Cout, H, W, ks, (*dec)[0] & 15, (*dec)[1], t, DA, S, ((*dec)[0] >> 8) & 1, ((*dec)[0] >> 9) & 3))
        g = t;
    }
  }
#ifdef PNSFM_BX3_ABLATE
  if (g.DMA >= 3 && g.DMA != 8 && g.smem_bytes + (size_t)g_smem_pad <= kMaxSmemPipe) g.smem_bytes += (size_t)g_smem_pad;
#endif
  const double flops = 2.0 * Cout * (double)Cin * ks * ks * (double)B * H * W;   // useful flops (output pixels)
  const int meta[9] = {B, Cin, Cout, H, W, ks, g.splitK, (int)(B * g.tiles_per_img * (g.MP / (32 * g.MT)) * g.splitK), g.DMA};
  g_last_conv = {g.DMA, g.NT, g.MT, g.G, g.splitK, g.mode == 2 ? (g.TW == 16 ? 1 : 2) : 0, meta[7], (int)g.smem_bytes};
  prof_begin(0, flops, stream, meta);
  const int rc = enqueue_conv(g, x, wp, bias, y, B, Cin, Cout, H, W, ks, stream, what, S, Hi, Wi, ms, gn);
  prof_end(0, stream);
  return rc;
}

// ---- weight packers -----------------------------------------------------------------------------
// forward:  wp[tap][ci][co]            = w[co][ci][tap]                (K = Cin,  M = Cout)
// backward: wp[tap][co][ci]            = w[co][ci][KK-1-tap]           (K = Cout, M = Cin; taps flipped)
// Both go through LDS so that global reads (tap fastest) and writes (M fastest) are both contiguous.
__device__ __forceinline__ void pack_fwd_block(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                                int KK, int KP, int MP, int bx, int by, float* tile) {
  const int ci = bx;       // < KP
  const int co0 = by * 64;
  const int n = 64 * KK;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int col = e / KK, tap = e - col * KK;  // read order: tap fastest
    const int co = co0 + col;
    float v = 0.f;
    if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * KK + tap];
    tile[col * KK + tap] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += 256) {
    const int tap = e >> 6, col = e & 63;  // write order: co fastest
    const int co = co0 + col;
    if (co < MP) wp[((size_t)tap * KP + ci) * MP + co] = tile[col * KK + tap];
  }
}

__device__ __forceinline__ void pack_bwd_block(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                                int KK, int KP, int MP, int bx, int by, float* tile) {
  const int co = bx;       // < KP (K dimension = Cout)
  const int ci0 = by * 64;
  const int n = 64 * KK;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int col = e / KK, tap = e - col * KK;
    const int ci = ci0 + col;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * KK + tap];  // contiguous over (ci, tap)
    tile[col * KK + tap] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += 256) {
    const int tap = e >> 6, col = e & 63;
    const int ci = ci0 + col;
    if (ci < MP) wp[((size_t)(KK - 1 - tap) * KP + co) * MP + ci] = tile[col * KK + tap];
  }
}

// One launch packs both layouts: the first nf = KPf * ceil(MPf/64) blocks write the forward layout, the following
// KPb * ceil(MPb/64) blocks the backward layout (nf = 0 / no further blocks when a pointer is null).
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp_fwd,
                                                            float* __restrict__ wp_bwd, int Cin, int Cout, int KK, int KPf,
                                                            int MPf, int KPb, int MPb, int nf) {
  __shared__ float tile[64 * 49];
  int blk = blockIdx.x;
  if (blk < nf) {
    pack_fwd_block(w, wp_fwd, Cin, Cout, KK, KPf, MPf, blk % KPf, blk / KPf, tile);
  } else {
    blk -= nf;
    pack_bwd_block(w, wp_bwd, Cin, Cout, KK, KPb, MPb, blk % KPb, blk / KPb, tile);
  }
}

// second stage of the pixel-split f32 weight-gradient kernels (this file and conv2d_wgrad2.hip): out = sum over Z slabs.  A slab is
// [n0 floats -> out0 | n1 floats -> out1] (dw, then dbias).  The slabs of an output are shared out to ZG = 256 / OPB thread groups
// (group g adds slabs g, g + ZG, ... with four independent partial sums) that meet in LDS in a fixed order: bit-reproducible, and
// Z = 640 slabs (the 3 -> 64 stem at 192x640) are 10-40 dependent steps instead of 640 (the serial form took 41 us per launch).
template <int OPB>
__global__ void __launch_bounds__(256) sum_slabs_kernel(const float* __restrict__ ws, size_t zstride, int Z, float* __restrict__ out0,
                                                        size_t n0, float* __restrict__ out1, size_t n1) {
  constexpr int ZG = 256 / OPB;
  __shared__ float red[ZG][OPB];
  const int ol = threadIdx.x % OPB, zg = threadIdx.x / OPB;
  const size_t i = (size_t)blockIdx.x * OPB + ol;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n0 + n1) {
    const float* p = ws + i;
    int z = zg;
    for (; z + 3 * ZG < Z; z += 4 * ZG) {
      s0 += p[(size_t)z * zstride];
      s1 += p[(size_t)(z + ZG) * zstride];
      s2 += p[(size_t)(z + 2 * ZG) * zstride];
      s3 += p[(size_t)(z + 3 * ZG) * zstride];
    }
    for (; z < Z; z += ZG) s0 += p[(size_t)z * zstride];
  }
  red[zg][ol] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (zg == 0 && i < n0 + n1) {
    float s = red[0][ol];
#pragma unroll
    for (int g = 1; g < ZG; ++g) s += red[g][ol];
    if (i < n0) out0[i] = s; else out1[i - n0] = s;
  }
}
int launch_sum_slabs(const float* ws, size_t zstride, int Z, float* out0, size_t n0, float* out1, size_t n1, hipStream_t s) {
  if (!out1) n1 = 0;
  const size_t n = n0 + n1;
  // few outputs (a 64 x 75 stem gradient): 16 per workgroup, 16 slab groups each, so that the launch still covers the chip
  if (n < (size_t)64 * 512) PNSFM_LAUNCH((sum_slabs_kernel<16>), dim3((unsigned)ceil_div_sz(n, 16)), dim3(256), 0, s, ws, zstride, Z, out0, n0, out1, n1);
  else PNSFM_LAUNCH((sum_slabs_kernel<64>), dim3((unsigned)ceil_div_sz(n, 64)), dim3(256), 0, s, ws, zstride, Z, out0, n0, out1, n1);
  return check_launch("conv2d_backward_weight (slab sum)");
}

// ---- weight gradient of the stem (3 input channels, 5x5, stride 1; round 6) ------------------------------------------------------------
// conv2d_wgrad_kernel walks the pixels two at a time on v_mfma_f32_32x32x2_f32 with a 4 x 32-column N tile of which 75 columns exist:
// 111 us for the 3 -> 64 stem at 4 x 192x640 (43 TFLOP/s) -- the LAST kernel of every backward pass, nothing left to hide it under.
// Here the same GEMM (M = co, N = (ci, ky, kx) = 75 -> 96 columns, K = pixels) runs on the split-bf16 arithmetic (exact 3-way bf16
// split, six piece products, fp32 accumulate: conv2d_bx3.h), 16 pixels per k-step:
//   * pixel tile = 4 rows x 64 columns = 16 k-steps; the 3-channel patch (8 x 68 floats per channel) is staged fp32 in LDS once per
//     tile, prefetched in registers under the previous tile's MFMAs;
//   * A (dY): lane (co = l & 31, half = l >> 5) loads 8 consecutive pixels of its channel row straight from global memory -- every
//     dY element is read exactly once by the launch -- one tile ahead, k-step by k-step;
//   * B (X): lane (n = l & 31 -> (ci, ky, kx), half) reads its 8 pixels at the tap's shift from the LDS patch (8 ds_read_b32:
//     the shift kx breaks every alignment) and splits them;
//   * the 4 waves are MT co tiles x 4 / MT pixel parts (rows of the tile); every wave holds its co tile x all 96 columns (48
//     accumulator registers); the parts meet in LDS at the end in a fixed order;
//   * pixel tiles are split over blockIdx.y; a split launch writes [dW | dbias] slabs for sum_slabs_kernel (no atomics).
// Needs W % 8 == 0 (a lane's 8 pixels never straddle the image edge).  Roofline: the 126 MB of dY (27 us); MFMA-bound 15 us.
struct StemWgradArgs {
  const float* x;    // [B][3][H][W]
  const float* dy;   // [B][Cout][H][W]
  float* dw;         // [Cout][75] (one split) or the first slab of the workspace
  float* dbias;      // [Cout] / the slab's bias part / null
  size_t zstride;    // floats between the slabs of two splits (0: one split)
  int B, Cout, H, W;
  int tiles_x, tiles_per_img, total_tiles, tiles_per_split;
};

template <int MT>
__global__ void __launch_bounds__(256) conv2d_wgrad_stem5_kernel(StemWgradArgs a) {
  constexpr int KS = 5, CIN = 3, NCOL = CIN * KS * KS;       // 75 columns
  constexpr int TR = 4, TC = 64, PH = TR + KS - 1, PWs = TC + KS - 1, RS = 72, PS = PH * RS;      // patch rows of 68 (+4 pad) floats
  constexpr int NPH = 4 / MT, RPP = TR / NPH, KPP = RPP * (TC / 16);      // pixel parts, rows per part, k-steps per part and tile
  constexpr int NPV = (CIN * PH * PWs + 255) / 256;
  __shared__ float patch[CIN * PS];
  __shared__ float meet[(4 - MT) * 49 * 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int mt = wave % MT, ph = wave / MT;
  const int H = a.H, W = a.W, HW = H * W;
  const int co = blockIdx.x * 32 * MT + 32 * mt + l32;
  const int bz = blockIdx.y;
  const int t_begin = bz * a.tiles_per_split;
  int t_end = t_begin + a.tiles_per_split;
  if (t_end > a.total_tiles) t_end = a.total_tiles;

  const pnsfm_buf dybuf = pnsfm_make_buf(a.dy, (unsigned)((size_t)a.B * a.Cout * HW * 4));
  const pnsfm_buf xbuf = pnsfm_make_buf(a.x, (unsigned)((size_t)a.B * CIN * HW * 4));

  // column n = 32 nt + l32 -> (ci, ky, kx) -> offset inside the patch (columns >= 75 are never stored: any in-range address)
  int nbase[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int n = 32 * nt + l32;
    const int ci = n / (KS * KS), tap = n - ci * KS * KS, ky = tap / KS, kx = tap - ky * KS;
    nbase[nt] = n < NCOL ? ci * PS + ky * RS + kx + 8 * half : 8 * half;
  }
  // patch items of this thread: (channel, row, column) of the 3 x 8 x 68 halo patch
  int pv_lds[NPV], pv_r[NPV], pv_c[NPV], pv_ci[NPV];
#pragma unroll
  for (int i = 0; i < NPV; ++i) {
    int e = i * 256 + tid;
    const bool live = e < CIN * PH * PWs;
    if (!live) e = 0;
    const int ci = e / (PH * PWs), rem = e - ci * (PH * PWs), r = rem / PWs, c = rem - r * PWs;
    pv_lds[i] = live ? ci * PS + r * RS + c : -1;
    pv_r[i] = r - KS / 2; pv_c[i] = c - KS / 2; pv_ci[i] = ci;
  }
  struct Cur { int b, y0, x0; };
  auto tile_of = [&](int t) {
    Cur c;
    c.b = t / a.tiles_per_img;
    const int tt = t - c.b * a.tiles_per_img, ty = tt / a.tiles_x;
    c.y0 = ty * TR; c.x0 = (tt - ty * a.tiles_x) * TC;
    return c;
  };
  float pv[NPV];
  auto load_patch = [&](const Cur& c) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) {
      const int yy = c.y0 + pv_r[i], xx = c.x0 + pv_c[i];
      const bool ok = pv_lds[i] >= 0 && yy >= 0 && yy < H && xx >= 0 && xx < W;
      pv[i] = pnsfm_buf_load(xbuf, ok ? (unsigned)((((c.b * CIN + pv_ci[i]) * H + yy) * W + xx) * 4) : PNSFM_DMA_INVALID, 0);
    }
  };
  // dY fragment of k-step j of this wave's part: tile row ph * RPP + j / 4, columns 16 (j % 4) + 8 half .. + 7
  float araw[KPP][8];
  auto load_a = [&](float (&v)[8], const Cur& c, int j) {
    const int yy = c.y0 + ph * RPP + j / 4, xx = c.x0 + 16 * (j % 4) + 8 * half;
    const bool ok = co < a.Cout && yy < H && xx < W;
    const unsigned off = ok ? (unsigned)((((c.b * a.Cout + co) * H + yy) * W + xx) * 4) : PNSFM_DMA_INVALID;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = pnsfm_buf_load(dybuf, off + 4u * u, 0);
  };

  f32x16 acc[3];
#pragma unroll
  for (int nt = 0; nt < 3; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
  float bsum = 0.f;

  if (t_begin < t_end) {
    const Cur c0 = tile_of(t_begin);
    load_patch(c0);
#pragma unroll
    for (int j = 0; j < KPP; ++j) load_a(araw[j], c0, j);
  }
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();           // every wave is done with the previous tile's patch
#pragma unroll
    for (int i = 0; i < NPV; ++i)
      if (pv_lds[i] >= 0) patch[pv_lds[i]] = pv[i];
    __syncthreads();
    const bool more = t + 1 < t_end;
    Cur cn = tile_of(more ? t + 1 : t);
    if (more) load_patch(cn);
#pragma unroll
    for (int j = 0; j < KPP; ++j) {
      pnsfm_u32x4 Ap[3];
      bx3_split8(araw[j], Ap[0], Ap[1], Ap[2]);
#pragma unroll
      for (int u = 0; u < 8; ++u) bsum += araw[j][u];
      if (more) load_a(araw[j], cn, j);
      const int poff = (ph * RPP + j / 4) * RS + 16 * (j % 4);
      pnsfm_u32x4 Bp[3][3];
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) {
        float bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bv[u] = patch[nbase[nt] + poff + u];
        bx3_split8(bv, Bp[nt][0], Bp[nt][1], Bp[nt][2]);
      }
      // smallest terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)   [dY piece, X piece]
#define PNSFM_SW_P(sa_, sb_) _Pragma("unroll") for (int nt = 0; nt < 3; ++nt) acc[nt] = pnsfm_mfma_bf16(Ap[sa_], Bp[nt][sb_], acc[nt])
      PNSFM_SW_P(2, 0); PNSFM_SW_P(0, 2); PNSFM_SW_P(1, 1); PNSFM_SW_P(1, 0); PNSFM_SW_P(0, 1); PNSFM_SW_P(0, 0);
#undef PNSFM_SW_P
    }
  }

  // ---- the pixel parts of a co tile meet: parts 1 .. NPH-1 park their sums in LDS, part 0 adds them in part order
  __syncthreads();
  if (ph > 0) {
    float* m = meet + (size_t)((ph - 1) * MT + mt) * 49 * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m[(nt * 16 + r) * 64] = acc[nt][r];
    m[48 * 64] = bsum;
  }
  __syncthreads();
  if (ph > 0) return;
#pragma unroll
  for (int p = 1; p < NPH; ++p) {
    const float* m = meet + (size_t)((p - 1) * MT + mt) * 49 * 64 + lane;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] += m[(nt * 16 + r) * 64];
    bsum += m[48 * 64];
  }
  // D row = (r & 3) + 8 (r >> 2) + 4 half -> co of this wave's tile, column = l32 -> n: consecutive lanes write consecutive n
  float* const dwp = a.dw + (size_t)bz * a.zstride;
  const int cow = blockIdx.x * 32 * MT + 32 * mt;
#pragma unroll
  for (int nt = 0; nt < 3; ++nt) {
    const int n = 32 * nt + l32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = cow + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n < NCOL && c < a.Cout) dwp[(size_t)c * NCOL + n] = acc[nt][r];
    }
  }
  if (a.dbias != nullptr) {
    const float v = bsum + __shfl_xor(bsum, 32);      // the two 8-pixel halves of a channel row
    if (half == 0 && co < a.Cout) (a.dbias + (size_t)bz * a.zstride)[co] = v;
  }
}

static bool wgrad_stem5_supported(int B, int Cin, int Cout, int H, int W, int ks, int S) {
  return S == 1 && Cin == 3 && ks == 5 && W % 8 == 0 && conv_math() == 1 && (size_t)B * Cout * H * W * 4 < (1ull << 31);
}
static int wgrad_stem5_total_tiles(int B, int H, int W) { return B * ceil_div(H, 4) * ceil_div(W, 64); }
static int enqueue_wgrad_stem5(const float* x, const float* dy, float* dw, float* dbias, int B, int Cout, int H, int W, int split,
                               hipStream_t s) {
  StemWgradArgs a;
  a.x = x; a.dy = dy; a.B = B; a.Cout = Cout; a.H = H; a.W = W;
  a.tiles_x = ceil_div(W, 64);
  a.tiles_per_img = a.tiles_x * ceil_div(H, 4);
  a.total_tiles = B * a.tiles_per_img;
  if (split < 1) split = 1;
  if (split > a.total_tiles) split = a.total_tiles;
  a.tiles_per_split = ceil_div(a.total_tiles, split);
  const int splitP = ceil_div(a.total_tiles, a.tiles_per_split);
  const size_t slab = (size_t)Cout * 75 + Cout;
  ScratchLease lease(s, splitP > 1 ? (size_t)splitP * slab * sizeof(float) : 0);
  a.dw = dw; a.dbias = dbias; a.zstride = 0;
  if (splitP > 1) {
    if (!lease.p) return -1;
    a.dw = lease.as<float>();
    a.dbias = dbias ? a.dw + (size_t)Cout * 75 : nullptr;
    a.zstride = slab;
  }
  const int MT = Cout > 32 ? 2 : 1;
  dim3 grid(ceil_div(Cout, 32 * MT), splitP);
  g_last_conv = {105, 0, MT, 0, splitP, 0, (int)(grid.x * grid.y), 0};      // pnsfm_conv2d_last_config: 105 = the stem's weight gradient
  if (MT == 2) PNSFM_LAUNCH((conv2d_wgrad_stem5_kernel<2>), grid, dim3(256), 0, s, a);
  else PNSFM_LAUNCH((conv2d_wgrad_stem5_kernel<1>), grid, dim3(256), 0, s, a);
  int rc = check_launch("conv2d_backward_weight (stem)");
  if (!rc && splitP > 1) rc = launch_sum_slabs(a.dw, slab, splitP, dw, (size_t)Cout * 75, dbias, (size_t)Cout, s);
  return rc;
}

// ---- backward-weight -------------------------------------------------------------------------------
// dW[co][n] with n = ci*KK + tap (the reference's [Cout][Cin][k][k] layout, contiguous in n):
//   dW[co][n] = sum_{b, pixel} dY[co][pixel] * X[ci(n)][pixel + off(tap(n))]
// GEMM view: M = co (32*MT per block), N = n (4 waves x 32), K = pixels (tiles of PT pixels).
// A operand: dY tile in LDS [BM][PT+1] (lane m = l&31, k = pixel parity l>>5) -- stride PT+1 is conflict-free;
// B operand: the halo patch of the <= NCI input channels this n-range touches; each lane owns one (ci, tap)
// and walks the pixels through a per-tile offset table.  Pixel tiles are split over blockIdx.z; each split stores its partial
// [dW | dbias] into its own slab of the stream's scratch buffer and sum_slabs_kernel adds the slabs in split order (round 4: no
// atomics, no zero-fill -- the gradient is bit-reproducible, as the split-bf16 kernels' always was).
struct WgradArgs {
  const float* x;   // [B][Cin][H][W]
  const float* dy;  // [B][Cout][H][W]
  float* dw;        // [Cout][Cin*KK]
  float* dbias;     // [Cout] or null: bias gradient (row sums of dY), accumulated by the n-tile-0 blocks
  int B, Cin, Cout, H, W, KS;
  int PT, mode, tiles_x, tiles_per_img, PH, PW, NCI, total_tiles, tiles_per_split, splitP;
  int cstride;  // true H*W of dY (channel stride); H, W above are the TILING dims (k=1 flattens the image to 32-wide rows)
  int S, Hi, Wi; // stride and INPUT (x) size; dY / the pixel tiles live on the OUTPUT grid
  int vec4, PTlog;
  float invPW, invPS;
  size_t zstride;  // pixel-split launch: floats between the partial [dw | dbias] slabs of consecutive splits (0: un-split)
};

template <int MT>
__global__ void __launch_bounds__(256) conv2d_wgrad_kernel(WgradArgs a) {
  PNSFM_DYN_SMEM(float, smem);
  constexpr int BM = 32 * MT;
  const int PT = a.PT, DS = PT + 1;
  const int PS = a.PH * a.PW;
  float* dys = smem;                       // [BM][PT+1]
  float* patch = smem + BM * DS;           // [NCI][PS]
  int* poff = reinterpret_cast<int*>(patch + a.NCI * PS);  // [PT]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, l32 = lane & 31;
  const int P = a.KS >> 1, KK = a.KS * a.KS;
  const int H = a.H, W = a.W, HW = a.cstride;
  const int S = a.S, Hi = a.Hi, Wi = a.Wi, HWi = a.S == 1 ? a.cstride : a.Hi * a.Wi;
  const int N = a.Cin * KK;

  const int n0 = blockIdx.x * 128;
  const int co0 = blockIdx.y * BM;
  const int ci_lo = n0 / KK;
  const int n = n0 + wave * 32 + l32;
  const bool nvalid = n < N;
  int lane_b = 0;
  if (nvalid) {
    const int ci_n = n / KK, tap = n - ci_n * KK;
    const int ky = tap / a.KS, kx = tap - ky * a.KS;
    lane_b = (ci_n - ci_lo) * PS + ky * a.PW + kx;
  }

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  const int t_begin = blockIdx.z * a.tiles_per_split;
  int t_end = t_begin + a.tiles_per_split;
  if (t_end > a.total_tiles) t_end = a.total_tiles;
  const bool do_bias = a.dbias != nullptr && blockIdx.x == 0;
  float bsum = 0.f;

  for (int tt = t_begin; tt < t_end; ++tt) {
    const int b = tt / a.tiles_per_img;
    const int t = tt - b * a.tiles_per_img;
    int py0, px0, y0 = 0, x0 = 0, pn0 = 0, r0 = 0;
    if (a.mode == 0) {
      const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
      y0 = ty * (PT >> 5);
      x0 = tx * 32;
      py0 = y0 * S - P;
      px0 = x0 * S - P;
    } else {
      pn0 = t * PT;
      r0 = pn0 / W;
      py0 = r0 * S - P;
      px0 = -P;
    }
    __syncthreads();  // previous tile fully consumed
    // ---- dY tile [BM][PT] (zero for channels / pixels outside the tensor); branch-free, batched loads
    const float* dyb = a.dy + (size_t)b * a.Cout * HW;
    if (a.vec4) {
      // rows of the tile are contiguous in memory (32-pixel row segments in 2-D mode, the whole run in linear mode)
      const int q4 = PT >> 2, q4log = a.PTlog - 2;
      for (int base = tid; base < BM * q4; base += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int g = base + u * 256;
          const int m = g >> q4log, p = (g & (q4 - 1)) << 2;
          int gidx;
          bool ok;
          if (a.mode == 0) { const int yy = y0 + (p >> 5); gidx = yy * W + x0 + (p & 31); ok = yy < H && gidx + 3 < HW; }
          else { gidx = pn0 + p; ok = gidx + 3 < HW; }
          ok = ok && g < BM * q4 && (co0 + m) < a.Cout;
          const float* src = ok ? dyb + ((size_t)(co0 + m) * HW + gidx) : pnsfm_zero_page;
          v[u] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int g = base + u * 256;
          if (g < BM * q4) {
            float* d = dys + (g >> q4log) * DS + ((g & (q4 - 1)) << 2);
            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
          }
        }
      }
    } else {
      for (int base = tid; base < BM * PT; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = base + u * 256;
          const int m = e >> a.PTlog, p = e & (PT - 1);
          int gidx;
          bool ok;
          if (a.mode == 0) { const int yy = y0 + (p >> 5); gidx = yy * W + x0 + (p & 31); ok = yy < H && gidx < HW; }
          else { gidx = pn0 + p; ok = gidx < HW; }
          ok = ok && e < BM * PT && (co0 + m) < a.Cout;
          const float* src = ok ? dyb + ((size_t)(co0 + m) * HW + gidx) : pnsfm_zero_page;
          v[u] = *src;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = base + u * 256;
          if (e < BM * PT) dys[(e >> a.PTlog) * DS + (e & (PT - 1))] = v[u];
        }
      }
    }
    // ---- pixel -> patch offset table
    if (tid < PT) {
      const int p = tid;
      int off;
      if (a.mode == 0) off = (p >> 5) * a.PW + (p & 31);
      else { const int pn = pn0 + p; const bool ok = pn < HW; const int yy = ok ? pn / W : r0; const int xx = ok ? pn - yy * W : 0; off = (yy - r0) * a.PW + xx; }
      poff[p] = off * S;
    }
    // ---- input halo patch for channels ci_lo .. ci_lo+NCI-1 (branch-free, batched)
    const float* xb = a.x + (size_t)b * a.Cin * HWi;
    {
      const int total = a.NCI * PS;
      for (int base = tid; base < total; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * 256;
          const int cil = (int)(((float)idx + 0.5f) * a.invPS);
          const int e = idx - cil * PS;
          const int r = (int)(((float)e + 0.5f) * a.invPW);
          const int cc = e - r * a.PW;
          const int yy = py0 + r, xx = px0 + cc, ci = ci_lo + cil;
          const bool ok = idx < total && ci < a.Cin && yy >= 0 && yy < Hi && xx >= 0 && xx < Wi && yy * Wi + xx < HWi;
          const float* src = ok ? xb + ((size_t)ci * HWi + yy * Wi + xx) : pnsfm_zero_page;
          v[u] = *src;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int idx = base + u * 256;
          if (idx < total) patch[idx] = v[u];
        }
      }
    }
    __syncthreads();
    if (do_bias) {  // bias gradient rides along: 4 threads per dY row, PT/4 pixels each (tile already in LDS)
      const int m = tid >> 2, q = tid & 3, span = PT >> 2;
      if (m < BM) {
        const float* row = dys + m * DS + q * span;
        float sacc = 0.f;
        for (int j = 0; j < span; ++j) sacc += row[j];
        bsum += sacc;
      }
    }
    // ---- K loop over the tile's pixels, 4 k-steps (8 pixels) per batch; the offset-table reads of the NEXT batch
    //      are issued before this batch's MFMAs so the poff -> patch dependent LDS chain is off the critical path
    const float* ab = dys + l32 * DS + half;
    const float* bb = patch + lane_b;
    const int nbatch = PT >> 3;
    int o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = poff[2 * j + half];
    for (int bt = 0; bt < nbatch; ++bt) {
      float bv[4], av[4][MT];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bv[j] = bb[o[j]];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[j][mt] = ab[mt * 32 * DS + 2 * (4 * bt + j)];
      }
      if (bt + 1 < nbatch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = poff[2 * (4 * (bt + 1) + j) + half];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = pnsfm_mfma_32x32x2(av[j][mt], bv[j], acc[mt]);
    }
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout && nvalid) {
        // (pixel-split launch: a.dw / a.dbias point at slab blockIdx.z of the workspace, see enqueue -- plain stores either way)
        a.dw[(size_t)blockIdx.z * a.zstride + (size_t)co * N + n] = acc[mt][r];
      }
    }
  }
  if (do_bias) {
    bsum += __shfl_down(bsum, 2);
    bsum += __shfl_down(bsum, 1);
    const int m = tid >> 2;
    if ((tid & 3) == 0 && m < BM && co0 + m < a.Cout) {
      a.dbias[(size_t)blockIdx.z * a.zstride + co0 + m] = bsum;
    }
  }
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

// floats to allocate for a packed weight.  A shape the split-bf16 kernels take needs 6 bytes per element (three bf16 pieces)
// instead of 4; the size does not depend on the current arithmetic mode, so a mode switch re-packs in place.
size_t pnsfm_conv2d_packed_elems_fwd(int Cin, int Cout, int ks) {
  const size_t n = (size_t)ks * ks * conv_pack_KP(Cin) * conv_pack_MP(Cout);
  return conv_bx3_supported(Cin, ks) ? n + n / 2 : n;
}
size_t pnsfm_conv2d_packed_elems_bwd(int Cin, int Cout, int ks) {
  const size_t n = (size_t)ks * ks * conv_pack_KP(Cout) * conv_pack_MP(Cin);
  return conv_bx3_supported(Cout, ks) ? n + n / 2 : n;
}

int pnsfm_conv2d_pack_weights(const float* w, float* wp_fwd, float* wp_bwd, int Cin, int Cout, int ks, void* stream) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("pack_weights: unsupported kernel size %d", ks); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const int KK = ks * ks;
  if (!wp_fwd && !wp_bwd) return 0;
  const int KPf = conv_pack_KP(Cin), MPf = conv_pack_MP(Cout), KPb = conv_pack_KP(Cout), MPb = conv_pack_MP(Cin);
  // the layout of each direction follows the kernel that will read it (launch_conv takes the same decision)
  const bool bxf = wp_fwd && conv_use_bx3(Cin, ks), bxb = wp_bwd && conv_use_bx3(Cout, ks);
  float* const f32f = bxf ? nullptr : wp_fwd;
  float* const f32b = bxb ? nullptr : wp_bwd;
  if (f32f || f32b) {
    const int nf = f32f ? KPf * ceil_div(MPf, 64) : 0, nb = f32b ? KPb * ceil_div(MPb, 64) : 0;
    PNSFM_LAUNCH(pack_weights_kernel, dim3(nf + nb), dim3(256), 0, s, w, f32f, f32b, Cin, Cout, KK, KPf, MPf, KPb, MPb, nf);
    int e = check_launch("pack_weights");
    if (e) return e;
  }
  if (bxf || bxb) {
    const int nchF = KPf / 16, nchB = KPb / 16;
    const int nf = bxf ? (MPf / 32) * nchF : 0, nb = bxb ? (MPb / 32) * nchB : 0;
    unsigned char* const pf = reinterpret_cast<unsigned char*>(bxf ? wp_fwd : nullptr);
    unsigned char* const pb = reinterpret_cast<unsigned char*>(bxb ? wp_bwd : nullptr);
    if (ks == 1) PNSFM_LAUNCH((pack_bx3_kernel<1>), dim3(nf + nb), dim3(256), 0, s, w, pf, pb, Cin, Cout, nchF, nchB, nf);
    else if (ks == 3) PNSFM_LAUNCH((pack_bx3_kernel<3>), dim3(nf + nb), dim3(256), 0, s, w, pf, pb, Cin, Cout, nchF, nchB, nf);
    else if (ks == 5) PNSFM_LAUNCH((pack_bx3_kernel<5>), dim3(nf + nb), dim3(256), 0, s, w, pf, pb, Cin, Cout, nchF, nchB, nf);
    else PNSFM_LAUNCH((pack_bx3_kernel<7>), dim3(nf + nb), dim3(256), 0, s, w, pf, pb, Cin, Cout, nchF, nchB, nf);
  }
  return check_launch("pack_weights");
}

size_t pnsfm_conv2d_pack_item_bytes(void) { return sizeof(PackItem); }

int pnsfm_conv2d_pack_item_fill(void* item_host, const float* w, float* wp_fwd, float* wp_bwd, int Cin, int Cout, int ks,
                                int first_block) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("pack_item_fill: unsupported kernel size %d", ks); return -1; }
  if (!item_host || !w || !wp_fwd || !wp_bwd) { set_error("pack_item_fill: null pointer"); return -1; }
  // only weights whose BOTH directions run the split-bf16 kernels go through the table (the others keep the per-layer call)
  if (!conv_use_bx3(Cin, ks) || !conv_use_bx3(Cout, ks)) return 0;
  const int KPf = conv_pack_KP(Cin), MPf = conv_pack_MP(Cout), KPb = conv_pack_KP(Cout), MPb = conv_pack_MP(Cin);
  PackItem it;
  it.w = w;
  it.pf = reinterpret_cast<unsigned char*>(wp_fwd);
  it.pb = reinterpret_cast<unsigned char*>(wp_bwd);
  it.Cin = Cin; it.Cout = Cout; it.ks = ks;
  it.nchF = KPf / 16; it.nchB = KPb / 16;
  it.nf = (MPf / 32) * it.nchF;
  it.blk0 = first_block;
  it.nblk = it.nf + (MPb / 32) * it.nchB;
  memcpy(item_host, &it, sizeof(it));
  return it.nblk;
}

size_t pnsfm_adam_pack_item_bytes(void) { return sizeof(AdamPackItem); }

// One item of the fused Adam + re-pack table (adam_pack_table_kernel): w / g / m / v = the parameter's slices of the optimizer's four
// arenas, hp = the group's device-resident hyper-parameters.  Returns the item's workgroup count, 0 when the weight does not take
// the split-bf16 layout in both directions (it stays with the plain update + the per-layer packer).
int pnsfm_adam_pack_item_fill(void* item_host, float* w, const float* g, float* m, float* v, const float* hp, float* wp_fwd,
                              float* wp_bwd, int Cin, int Cout, int ks, int first_block) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("adam_pack_item_fill: unsupported kernel size %d", ks); return -1; }
  if (!item_host || !w || !g || !m || !v || !hp || !wp_fwd || !wp_bwd) { set_error("adam_pack_item_fill: null pointer"); return -1; }
  if (!conv_use_bx3(Cin, ks) || !conv_use_bx3(Cout, ks)) return 0;
  AdamPackItem it;
  it.w = w; it.g = g; it.m = m; it.v = v; it.hp = hp;
  it.pf = reinterpret_cast<unsigned char*>(wp_fwd);
  it.pb = reinterpret_cast<unsigned char*>(wp_bwd);
  it.Cin = Cin; it.Cout = Cout; it.ks = ks;
  it.nchF = conv_pack_KP(Cin) / 16; it.nchB = conv_pack_KP(Cout) / 16;
  // the packed images have conv_pack_MP(.) / 32 m-blocks; a super-tile row / column is one of them
  const int cotiles = conv_pack_MP(Cout) / 32;
  it.citiles = conv_pack_MP(Cin) / 32;
  if (2 * it.citiles < it.nchF || 2 * cotiles < it.nchB) { set_error("adam_pack_item_fill: tile / chunk mismatch"); return -1; }
  it.blk0 = first_block;
  it.nblk = cotiles * it.citiles;
  memcpy(item_host, &it, sizeof(it));
  return it.nblk;
}

int pnsfm_adam_pack_table(const void* table_dev, int n_items, int total_blocks, void* stream) {
  if (n_items <= 0 || total_blocks <= 0) return 0;
  if (!table_dev) { set_error("adam_pack_table: null table"); return -1; }
  PNSFM_LAUNCH(adam_pack_table_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const AdamPackItem*>(table_dev), n_items);
  return check_launch("adam_pack_table");
}

int pnsfm_conv2d_pack_table(const void* table_dev, int n_items, int total_blocks, void* stream) {
  if (n_items <= 0 || total_blocks <= 0) return 0;
  if (!table_dev) { set_error("pack_table: null table"); return -1; }
  PNSFM_LAUNCH(pack_bx3_table_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
               reinterpret_cast<const PackItem*>(table_dev), n_items);
  return check_launch("pack_table");
}

int pnsfm_conv2d_forward(const float* x, const float* wp_fwd, const float* bias, float* y, int B, int Cin, int Cout,
                         int H, int W, int ks, void* stream) {
  return launch_conv(x, wp_fwd, bias, y, B, Cin, Cout, H, W, ks, (hipStream_t)stream, "conv2d_forward", 0);
}

int pnsfm_conv2d_backward_data(const float* dy, const float* wp_bwd, float* dx, int B, int Cin, int Cout, int H,
                               int W, int ks, void* stream) {
  // dX = conv(dY, flipped/transposed W): K-channels = Cout, M-channels = Cin
  return launch_conv(dy, wp_bwd, nullptr, dx, B, Cout, Cin, H, W, ks, (hipStream_t)stream, "conv2d_backward_data", 1);
}

// dx = conv(dY, flipped/transposed W) + addend: the tensor's OTHER gradient (a skip connection's, a 1x1 shortcut's) added in the
// epilogue instead of by autograd's elementwise sum afterwards.  addend: [B][Cin][H][W], `addend_bstride` floats between samples
// (>= Cin*H*W: a channel slice of the wider gradient a multi-source convolution produced); must not overlap dx.
int pnsfm_conv2d_backward_data_add(const float* dy, const float* wp_bwd, float* dx, const float* addend, long long addend_bstride,
                                   int B, int Cin, int Cout, int H, int W, int ks, void* stream) {
  if (!addend) return pnsfm_conv2d_backward_data(dy, wp_bwd, dx, B, Cin, Cout, H, W, ks, stream);
  if (addend_bstride < (long long)Cin * H * W) { set_error("conv2d_backward_data_add: addend_bstride smaller than a sample"); return -1; }
  ConvGnOut ex = {addend, (size_t)addend_bstride};
  return launch_conv(dy, wp_bwd, nullptr, dx, B, Cout, Cin, H, W, ks, (hipStream_t)stream, "conv2d_backward_data_add", 1, 1, 0, 0, nullptr, &ex);
}

static bool conv_ms_ok(int C0, int C1, int C2, int granule, const char* what) {
  if (C0 <= 0 || C1 <= 0 || C2 < 0 || C0 % granule != 0 || (C2 > 0 && (C0 + C1) % granule != 0)) {
    set_error("%s: input tensors of %d / %d / %d channels: every tensor but the last must end on a %d-channel boundary", what, C0, C1,
              C2, granule);
    return false;
  }
  return true;
}

int pnsfm_conv2d_forward_cat(const float* x0, int C0, const float* x1, int C1, const float* x2, int C2, const float* wp_fwd,
                             const float* bias, float* y, int B, int Cout, int H, int W, int ks, void* stream) {
  const int Cin = C0 + C1 + C2;
  if (!conv_ms_ok(C0, C1, C2, 16, "conv2d_forward_cat")) return -1;
  if (!conv_use_bx3(Cin, ks)) { set_error("conv2d_forward_cat: needs the split-bf16 arithmetic (>= 16 channels)"); return -1; }
  const ConvSrc ms = {x1, x2, C0, C1};
  return launch_conv(x0, wp_fwd, bias, y, B, Cin, Cout, H, W, ks, (hipStream_t)stream, "conv2d_forward_cat", 0, 1, 0, 0, &ms);
}

// H, W: size of dY (the conv OUTPUT); x is [B, Cin, Hi, Wi] with Hi = H, Wi = W when S == 1
static int wgrad_impl(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout, int H, int W,
                      int ks, int S, int Hi, int Wi, void* stream, const ConvSrc* ms = nullptr) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("backward_weight: unsupported kernel size %d", ks); return -1; }
  if (S != 1 && S != 2) { set_error("backward_weight: unsupported stride %d", S); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const int KK = ks * ks, N = Cin * KK, HW = H * W;
  const int H0 = H, W0 = W;                                  // true image size (the 1x1 path below re-tiles H, W)
  const bool v2_ok = S == 1 && wgrad2_supported(Cin, Cout, H0, W0, ks);
  // tap-major kernel: split the 64-pixel tiles so that ~every CU gets one workgroup
  auto v2_default_split = [&]() -> int {
    const int base = wgrad2_base_blocks(Cin, Cout, ks), tiles = wgrad2_total_tiles(B, H0, W0);
    int sp = (256 + base / 2) / base;
    if (sp < 1) sp = 1;
    if (sp > tiles) sp = tiles;
    return sp;
  };
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = dw; a.dbias = dbias;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.KS = ks;
  a.cstride = HW;
  a.S = S;
  if (ks == 1 && S == 1) {  // no halo: any pixel order works, so tile the flattened image as 32-wide rows
    a.W = W = 32;
    a.H = H = ceil_div(HW, 32);
    Hi = H; Wi = W;
  }
  a.Hi = Hi; a.Wi = Wi;
  const int MT = conv_pick_MT(Cout), BM = 32 * MT;
  a.mode = (W % 32 == 0) ? 0 : 1;
  a.NCI = 127 / KK + 2;
  if (a.NCI > Cin) a.NCI = Cin;
  size_t smem = 0;
  for (int PT = (KK == 1 ? 64 : 128); PT >= 32; PT >>= 1) {
    a.PT = PT;
    if (a.mode == 0) {
      a.tiles_x = W / 32;
      a.tiles_per_img = a.tiles_x * ceil_div(H, PT / 32);
      a.PH = (PT / 32 - 1) * S + ks;
      a.PW = 31 * S + ks;
    } else {
      a.tiles_x = 0;
      a.tiles_per_img = ceil_div(HW, PT);
      int rows = (PT + W - 2) / W + 1;
      if (rows > H) rows = H;
      a.PH = (rows - 1) * S + ks;
      a.PW = (W - 1) * S + ks;
    }
    smem = ((size_t)BM * (PT + 1) + (size_t)a.NCI * a.PH * a.PW + PT) * sizeof(float);
    if (smem <= kMaxSmem) break;
  }
  if (smem > kMaxSmem) { set_error("backward_weight: image too wide for the LDS halo patch (W=%d)", W); return -1; }
  a.invPW = 1.0f / (float)a.PW;
  a.invPS = 1.0f / (float)(a.PH * a.PW);
  a.vec4 = (HW % 4 == 0) ? 1 : 0;
  a.PTlog = a.PT == 128 ? 7 : (a.PT == 64 ? 6 : 5);
  a.total_tiles = B * a.tiles_per_img;
  const int n_tiles = ceil_div(N, 128), m_tiles = ceil_div(Cout, BM);
  // pixel-tile split: minimise  rounds(blocks / resident slots) x (tiles per block + fixed per-block cost)
  {
    int occ = (int)((160 * 1024) / (smem + 512));
    if (occ > 4) occ = 4;                       // 67 VGPR + 32 AGPR -> 4 waves/SIMD
    if (occ < 1) occ = 1;
    const long slots = 256L * occ, base = (long)n_tiles * m_tiles;
    double best = 1e30;
    a.splitP = 1;
    a.tiles_per_split = a.total_tiles;
    int prev_tps = -1;
    for (int want = 1; want <= a.total_tiles && want <= 4096; ++want) {
      const int tps = ceil_div(a.total_tiles, want);
      if (tps == prev_tps) continue;
      prev_tps = tps;
      const int split = ceil_div(a.total_tiles, tps);
      const double rounds = (double)((base * split + slots - 1) / slots);
      const double cost = rounds * ((double)tps + 0.75) * (split > 1 ? 1.02 : 1.0);
      if (cost < best) { best = cost; a.splitP = split; a.tiles_per_split = tps; }
      if (base * split > 64 * slots) break;
    }
  }
  //  a pixel split keeps `split` whole [dw | dbias] slabs in the stream's grow-only scratch (api.hip), which torch's
  // allocator cannot see or reuse: 75 MB per slab for the pack5 weight.  Splits are capped so that the slabs of one launch stay under
  // kWgradScratchBudget (the split-bf16 kernels, the default arithmetic, keep far smaller partial tensors and are not affected).
  const size_t kWgradScratchBudget = (size_t)128 << 20;
  const int max_split_f32 = (int)std::max<size_t>(1, kWgradScratchBudget / (((size_t)Cout * N + Cout) * sizeof(float)));
  const bool stem_wg = !ms && wgrad_stem5_supported(B, Cin, Cout, H0, W0, ks, S);      // the 3-channel 5x5 stem: its own kernel
  auto enqueue = [&](int split) -> int {
    if (split > max_split_f32) split = max_split_f32;
    if (stem_wg) return enqueue_wgrad_stem5(x, dy, dw, dbias, B, Cout, H0, W0, split, s);
    WgradArgs c = a;
    c.tiles_per_split = ceil_div(a.total_tiles, split);
    c.splitP = ceil_div(a.total_tiles, c.tiles_per_split);
    // un-split: every dw element and every dbias[co] has exactly one writer -> plain stores into dw / dbias.
    // split over pixel tiles: every split writes a whole [dw | dbias] slab of the scratch buffer; sum_slabs_kernel adds them.
    const size_t slab = (size_t)Cout * N + Cout;
    ScratchLease lease(s, c.splitP > 1 ? (size_t)c.splitP * slab * sizeof(float) : 0);
    c.zstride = 0;
    if (c.splitP > 1) {
      if (!lease.p) return -1;
      c.dw = lease.as<float>();
      c.dbias = dbias ? c.dw + (size_t)Cout * N : nullptr;
      c.zstride = slab;
    }
    dim3 grid(n_tiles, m_tiles, c.splitP);
    if (MT == 2) PNSFM_LAUNCH((conv2d_wgrad_kernel<2>), grid, dim3(256), smem, s, c);
    else PNSFM_LAUNCH((conv2d_wgrad_kernel<1>), grid, dim3(256), smem, s, c);
    int rc = check_launch("conv2d_backward_weight");
    if (!rc && c.splitP > 1) rc = launch_sum_slabs(c.dw, slab, c.splitP, dw, (size_t)Cout * N, dbias, (size_t)Cout, s);
    return rc;
  };
  // split-bf16 kernel (conv2d_wgrad3.hip): part of the split arithmetic mode (pnsfm_set_conv_math), the default there
  // split-bf16 kernel: a 1x1 layer has no halo, so its map is handed over as 32-wide rows of the flattened image when that
  // is exact (whole 16/32-column tiles instead of ragged ones for W = 40, 80)
  const bool flat3 = ks == 1 && (H0 * W0) % 32 == 0;
  const int H3 = flat3 ? (H0 * W0) / 32 : H0, W3 = flat3 ? 32 : W0;
  const bool v3_ok = S == 1 && conv_math() == 1 && wgrad3_supported(Cin, Cout, H3, W3, ks) && wgrad3_fits(B, Cin, Cout, H3, W3);
  auto v3_default_split = [&](int NT) -> int {
    const int base = wgrad3_base_blocks(Cin, Cout, ks, NT, 0), tiles = wgrad3_total_tiles(B, H3, W3);
    int split = (2 * 256 + base - 1) / base;            // two workgroups per CU
    if (split > tiles) split = tiles;
    return split < 1 ? 1 : split;
  };
  int variant = (g_wgrad_variant == 1 && v2_ok) ? 1 : 0;
  int split2 = v2_ok ? v2_default_split() : 1;
  int nt3 = wgrad3_nt2_ok(Cin, ks) ? 2 : 1, wm3 = 0;
  int split3 = v3_ok ? v3_default_split(nt3) : 1;
  // nine-taps kernel (conv2d_wgrad4.hip): 3x3 only; chosen by the autotuner / a pinned decision (variant 3)
  const bool v4_ok = v3_ok && ks == 3 && wgrad4_supported(Cin, Cout, H0, W0, ks);
  int split4 = 1, cfg4 = 2;
  if (v3_ok && g_wgrad_variant != 0 && g_wgrad_variant != 1) variant = 2;     // -1 (library default) or 2 (pinned)
  if (ms) {          // several input tensors (ConvSrc): only the split-bf16 kernel reads them
    if (!v3_ok) { set_error("backward_weight: several input tensors need the split-bf16 weight-gradient kernel"); return -1; }
    variant = 2;
  }
  {
    // tuned / pinned decision of this shape (autotuner, PNSFM_TUNE_DB / shipped database, pnsfm_tune_set -- the latter in
    // every build, so tests can pin kernel, split, ci tiles per wave and co tiles per workgroup on the emulator too)
    const std::array<int, 7> key = {2 + 10 * S + (v3_ok ? 100 : 0) + (ms ? 1000 : 0), B, Cin, Cout, a.cstride, W, ks};
    const bool tune = autotune_enabled();
    std::lock_guard<std::mutex> lk(g_tune_mu);
    const std::array<int, 2>* dec = tune_lookup(key, tune);
#ifndef PNSFM_EMU
    // (first seen inside a hipGraph capture: keep the analytic split -- timing would synchronise)
    if (!dec && tune && !stream_capturing(s)) {
      float best_ms = 1e30f;
      int best_split = a.splitP, prev_tps = -1;
      const long base = (long)n_tiles * m_tiles;
      for (int want = 1; want <= a.total_tiles && !ms; want = want < 8 ? want + 1 : (want * 3 + 1) / 2) {
        const int tps = ceil_div(a.total_tiles, want);
        if (tps == prev_tps) continue;
        prev_tps = tps;
        const int split = ceil_div(a.total_tiles, tps);
        if (base * split < 192 && split < a.total_tiles && split < max_split_f32) continue;   // cannot fill the chip: not worth timing
        if ((base * split > 40L * 256 || split > max_split_f32) && split > 1) break;
        const float ms = time_on_stream(s, 2, [&]() { return enqueue(split); });
        tune_log(2, key, 0, split, ms);
        if (ms > 0.f && ms < best_ms) { best_ms = ms; best_split = split; }
      }
      int best_variant = 0;
      if (v2_ok && !ms) {     // tap-major kernel: pixel splits around one workgroup per CU
        const int base2 = wgrad2_base_blocks(Cin, Cout, ks), tiles2 = wgrad2_total_tiles(B, H0, W0);
        int prev = -1;
        for (int want = 1; want <= tiles2; want = want < 4 ? want + 1 : (want * 3 + 1) / 2) {
          const int tps = ceil_div(tiles2, want);
          const int split = ceil_div(tiles2, tps);
          if (split == prev) continue;
          prev = split;
          if ((long)base2 * split < 160 && split < tiles2) continue;      // cannot fill the chip
          if ((long)base2 * split > 6L * 256 && split > 1) break;
          const float ms = time_on_stream(s, 2, [&]() { return enqueue_wgrad2(x, dy, dw, dbias, B, Cin, Cout, H0, W0, ks, split, s); });
          tune_log(2, key, 1, split, ms);
          if (ms > 0.f && ms < best_ms) { best_ms = ms; best_split = split; best_variant = 1; }
        }
      }
      if (v3_ok) {     // split-bf16 kernel: NT in {1, 2} x co tiles per workgroup x pixel splits around two workgroups per CU
        const int tiles3 = wgrad3_total_tiles(B, H3, W3);
        const int wm_most = wgrad3_WM(Cout, 0);
        for (int NT = 1; NT <= (wgrad3_nt2_ok(Cin, ks) ? 2 : 1); ++NT)
          for (int WMv = wm_most; WMv >= 1; WMv >>= 1) {
            const int base3 = wgrad3_base_blocks(Cin, Cout, ks, NT, WMv);
            int prev = -1;
            for (int want = 1; want <= tiles3; want = want < 4 ? want + 1 : (want * 3 + 1) / 2) {
              const int tps = ceil_div(tiles3, want);
              const int split = ceil_div(tiles3, tps);
              if (split == prev) continue;
              prev = split;
              if ((long)base3 * split < 200 && split < tiles3) continue;      // cannot fill the chip
              if ((long)base3 * split > 16L * 256 && split > 1) break;
              if (WMv != wm_most && split > 2) break;     // fewer co tiles per workgroup only pays when it replaces the pixel split
              if (ms && NT == 2 && !conv_src_aligned(*ms, Cin, 64)) continue;
              for (int occ = 0; occ < ((ks == 3 && NT == 1 && W3 % 8 == 0) ? 2 : 1); ++occ) {    // occ 1: the three-workgroups-per-CU build
                const int wmv = WMv | (occ ? 8 : 0);
                const float tms = time_on_stream(s, 2, [&]() { return enqueue_wgrad3(x, dy, dw, dbias, B, Cin, Cout, H3, W3, ks, split, NT, wmv, s, ms); });
                tune_log(2, key, 2 | (NT << 4) | (wmv << 6), split, tms);
                if (tms > 0.f && tms < best_ms) { best_ms = tms; best_split = split; best_variant = 2 | (NT << 4) | (wmv << 6); }
              }
            }
          }
      }
      if (v4_ok) {     // nine-taps kernel: ci tiles per workgroup x tile width x pixel splits around two workgroups per CU
        for (int WCI = 1; WCI <= 2; ++WCI)
          for (int tgi = 0; tgi < (W0 > 24 ? 2 : 1); ++tgi) {
            const int TG = W0 > 24 ? 4 + tgi : 3, TR = TG == 3 ? wgrad4_TR(H0, 0) : 4;
            if (W0 > 24 && round_up(W0, 8 * TG) > round_up(W0, 8 * (9 - TG)) + 8) continue;      // clearly the more wasteful width
            const int tiles4 = wgrad4_total_tiles(B, H0, W0, TG, TR), base4 = wgrad4_base_blocks(Cin, Cout, WCI);
            const int cfg = WCI | (TG << 4) | (TR << 8);
            int prev = -1;
            for (int want = 1; want <= tiles4; want = want < 4 ? want + 1 : (want * 3 + 1) / 2) {
              const int tps = ceil_div(tiles4, want);
              const int split = ceil_div(tiles4, tps);
              if (split == prev) continue;
              prev = split;
              if ((long)base4 * split < 200 && split < tiles4) continue;      // cannot fill the chip
              if ((long)base4 * split > 16L * 256 && split > 1) break;
              const float tms = time_on_stream(s, 2, [&]() { return enqueue_wgrad4(x, dy, dw, dbias, B, Cin, Cout, H0, W0, split, cfg, s, ms); });
              tune_log(2, key, 3 | (cfg << 4), split, tms);
              if (tms > 0.f && tms < best_ms) { best_ms = tms; best_split = split; best_variant = 3 | (cfg << 4); }
            }
          }
      }
      dec = &g_tuned.emplace(key, std::array<int, 2>{best_split, best_variant}).first->second;
      tune_db_append(key, *dec);
    }
#endif
    if (dec) {
      const int d0 = (*dec)[0], d1 = (*dec)[1];
      variant = ((d1 & 15) == 1 && v2_ok) ? 1 : (((d1 & 15) == 2 && v3_ok) ? 2 : (((d1 & 15) == 3 && v4_ok) ? 3 : 0));
      if (ms && variant != 3) variant = 2;
      if (variant == 3) { split4 = d0; cfg4 = d1 >> 4; }
      if (variant == 2 && (d1 & 15) != 2) { /* a pinned decision for another kernel: keep the split-bf16 defaults */ }
      else if (variant == 2) { split3 = d0; nt3 = ((d1 >> 4) & 3) == 2 ? 2 : 1; wm3 = (d1 >> 6) & 15; }
      else if (variant == 1) split2 = d0;
      else {
        int sp = d0;
        if (sp < 1) sp = 1;
        if (sp > a.total_tiles) sp = a.total_tiles;
        a.tiles_per_split = ceil_div(a.total_tiles, sp);
        a.splitP = ceil_div(a.total_tiles, a.tiles_per_split);
      }
    }
  }
  const double flops = 2.0 * Cout * (double)Cin * KK * (double)B * HW;
  if (variant == 3) {
    const int meta[9] = {B, Cin, Cout, a.cstride, W0, ks, split4, wgrad4_base_blocks(Cin, Cout, (cfg4 & 15) == 1 ? 1 : 2) * split4, 4};
    prof_begin(1, flops, s, meta);
    const int rc = enqueue_wgrad4(x, dy, dw, dbias, B, Cin, Cout, H0, W0, split4, cfg4, s, ms);
    prof_end(1, s);
    return rc;
  }
  if (variant == 2) {
    const int meta[9] = {B, Cin, Cout, a.cstride, W0, ks, split3, wgrad3_base_blocks(Cin, Cout, ks, nt3, wm3) * split3, 3};
    prof_begin(1, flops, s, meta);
    const int rc = enqueue_wgrad3(x, dy, dw, dbias, B, Cin, Cout, H3, W3, ks, split3, nt3, wm3, s, ms);
    prof_end(1, s);
    return rc;
  }
  if (variant == 1) {
    const int meta[9] = {B, Cin, Cout, a.cstride, W0, ks, split2, wgrad2_base_blocks(Cin, Cout, ks) * split2, 2};
    prof_begin(1, flops, s, meta);
    const int rc = enqueue_wgrad2(x, dy, dw, dbias, B, Cin, Cout, H0, W0, ks, split2, s);
    prof_end(1, s);
    return rc;
  }
  const int meta[9] = {B, Cin, Cout, a.cstride, W, ks, a.splitP, (int)(n_tiles * m_tiles * a.splitP), 0};
  prof_begin(1, flops, s, meta);
  const int rc = enqueue(a.splitP);
  prof_end(1, s);
  return rc;
}

int pnsfm_conv2d_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout,
                                 int H, int W, int ks, void* stream) {
  return wgrad_impl(x, dy, dw, dbias, B, Cin, Cout, H, W, ks, 1, H, W, stream);
}

int pnsfm_conv2d_backward_weight_cat(const float* x0, int C0, const float* x1, int C1, const float* x2, int C2, const float* dy,
                                     float* dw, float* dbias, int B, int Cout, int H, int W, int ks, void* stream) {
  if (!conv_ms_ok(C0, C1, C2, 32, "conv2d_backward_weight_cat")) return -1;
  const ConvSrc ms = {x1, x2, C0, C1};
  return wgrad_impl(x0, dy, dw, dbias, B, C0 + C1 + C2, Cout, H, W, ks, 1, H, W, stream, &ms);
}

// 1 when pnsfm_conv2d_backward_weight_cat takes these sources (the caller decides ONCE, up front, whether to concatenate for the
// weight gradient instead of finding out from an error), else 0.  No launch, no error state.
int pnsfm_conv2d_cat_wgrad_supported(int C0, int C1, int C2, int Cout, int B, int H, int W, int ks) {
  if (C0 <= 0 || C1 <= 0 || C2 < 0 || C0 % 32 != 0 || (C2 > 0 && (C0 + C1) % 32 != 0)) return 0;
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) return 0;
  const int Cin = C0 + C1 + C2;
  const bool flat3 = ks == 1 && (H * W) % 32 == 0;
  const int H3 = flat3 ? (H * W) / 32 : H, W3 = flat3 ? 32 : W;
  return (conv_math() == 1 && wgrad3_supported(Cin, Cout, H3, W3, ks) && wgrad3_fits(B, Cin, Cout, H3, W3)) ? 1 : 0;
}

int pnsfm_conv2d_forward_strided(const float* x, const float* wp_fwd, const float* bias, float* y, int B, int Cin,
                                 int Cout, int Hin, int Win, int ks, int stride, void* stream) {
  const int P = ks / 2;
  const int Ho = (Hin + 2 * P - ks) / stride + 1, Wo = (Win + 2 * P - ks) / stride + 1;
  return launch_conv(x, wp_fwd, bias, y, B, Cin, Cout, Ho, Wo, ks, (hipStream_t)stream, "conv2d_forward_strided", 0, stride,
                     Hin, Win);
}

int pnsfm_conv2d_backward_weight_strided(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin,
                                         int Cout, int Hin, int Win, int ks, int stride, void* stream) {
  const int P = ks / 2;
  const int Ho = (Hin + 2 * P - ks) / stride + 1, Wo = (Win + 2 * P - ks) / stride + 1;
  return wgrad_impl(x, dy, dw, dbias, B, Cin, Cout, Ho, Wo, ks, stride, Hin, Win, stream);
}

int pnsfm_set_autotune(int on) {
  g_autotune = on ? 1 : 0;
  return 0;
}

int pnsfm_tune_set(const int* key7, int v0, int v1) {
  if (!key7) { set_error("tune_set: null key"); return -1; }
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_pinned[{key7[0], key7[1], key7[2], key7[3], key7[4], key7[5], key7[6]}] = {v0, v1};
  return 0;
}

int pnsfm_set_wgrad_variant(int tap_major) {
  g_wgrad_variant = (tap_major < 0 || tap_major > 2) ? -1 : tap_major;
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_pinned.clear();       // pins only: database / autotuner decisions stay (they are ignored while autotuning is off)
  return 0;
}

int pnsfm_set_conv_variant(int lds_dma) {
  if (lds_dma >= 3) g_default_bx3 = lds_dma > 6 ? 6 : lds_dma;     // 3..6: un-tuned default of the split-bf16 kernels
  else g_default_dma = lds_dma < 0 ? 0 : lds_dma;
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_pinned.clear();       // pins only (see pnsfm_set_wgrad_variant)
  return 0;
}

// arithmetic of the forward / backward-data kernels: 0 = f32 MFMA, 1 = split-bf16 (default); returns the previous mode.
// Packed weights must be re-packed after a switch.
int pnsfm_set_conv_math(int mode) {
  const int prev = conv_math();
  g_conv_math = mode ? 1 : 0;
  return prev;
}
int pnsfm_get_conv_math(void) { return conv_math(); }

int pnsfm_conv2d_last_config(int* out8) {
  if (!out8) { set_error("conv2d_last_config: null pointer"); return -1; }
  for (int i = 0; i < 8; ++i) out8[i] = g_last_conv[i];
  return g_last_conv[0] < 0 ? 1 : 0;
}

int pnsfm_tune_shipped_entries(void) {
  (void)autotune_enabled();      // first call reads the environment / the shipped database
  return g_shipped_entries;
}

}  // extern "C"
