// sparse.hip -- 2-D sparse tensors on the pixel grid: the depth-completion branch of PackNet-SAN.
//
// Replaces, for that branch, the MinkowskiEngine operations the reference calls
//   (/root/reference/packnet_sfm/networks/layers/minkowski_encoder.py:10-131: ME.MinkowskiConvolution(k, stride 1, dimension 2),
//    ME.MinkowskiMaxPooling(3, 2), and /root/reference/packnet_sfm/networks/layers/minkowski.py:33-83: sparsify_depth,
//    densify_features, map_add_features)
// with kernels whose cost is proportional to the number of ACTIVE sites (LiDAR: ~5 % of the pixels), not to the image.
//
// Representation (one per feature level): the active cells of the [B, h, w] grid in ascending cell order,
//   sites [cap]   int32  linear cell index b*h*w + y*w + x of row n (rows >= count are unused)
//   imap  [B*h*w] int32  row of a cell, or -1
//   count [1]     int32  number of rows, ON THE DEVICE (no host round trip after the first level)
//   feats [cap][C] fp32  one feature row per site (channels fastest: a row is one contiguous run); rows >= count are zero
//   nbr   [cap][k*k] int32  row of the neighbour at kernel offset i, or -1 (built once per level and kernel size)
// Kernel offset order (MinkowskiEngine's region iterator, first coordinate fastest): i = (dy + k/2) + k * (dx + k/2).
//
// Kernels
//   sp_count / sp_scan / sp_write   coordinate map: three-pass deterministic compaction of "value > 0" cells
//   sp_pool_cells                   stride-2 coordinate rule: a coarse cell is active iff one of its 2x2 fine cells is
//   sp_neighbors                    neighbour table from imap
//   sparse_conv_kernel              out[n] = sum_i W[i]^T . feats[nbr[n][i]]   -- implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32):
//                                   M = 32 sites per workgroup, N = 128 / 256 output channels (one or two tiles per wave), K = (offset, channel);
//                                   the 32 neighbour rows of an offset are GATHERED (one contiguous run each) into LDS, offsets with
//                                   no active neighbour in the tile are skipped, the weights stream straight from L2.
//                                   Backward-data is the same kernel on the transposed kernel with the offsets mirrored.
//   sparse_wgrad_kernel             dW[i][ci][co] = sum_n feats[nbr[n][i]][ci] * dout[n][co]: M = ci, N = co, K = sites
//   sp_maxpool_fwd/bwd              max over the active cells of the 3x3 window centred on the coarse cell's origin; argmax routing
//   sp_densify / sp_gather          rows -> dense NCHW (zeros elsewhere) and back (also: dense features picked up at the sites)
// Roofline: the convolutions are MFMA-bound on the f32 instruction (157.3 TFLOP/s) at 2*N*Cin*Cout*k*k flop for N active sites --
// against 2*B*h*w*Cin*Cout*k*k for the dense-plus-mask formulation of round 2.
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

// ---- coordinate map ---------------------------------------------------------------------------------------------------
// exclusive prefix sum of one int per thread over the 256 threads of the workgroup (Hillis-Steele through LDS)
__device__ __forceinline__ int block_exclusive_scan(int v, int* lds, int* total) {
  const int t = threadIdx.x;
  lds[t] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int add = t >= d ? lds[t - d] : 0;
    __syncthreads();
    lds[t] += add;
    __syncthreads();
  }
  const int incl = lds[t];
  *total = lds[255];
  __syncthreads();
  return incl - v;
}

// a workgroup covers 1024 consecutive cells, a thread 4 consecutive ones
__global__ void __launch_bounds__(256) sp_count_kernel(const float* __restrict__ src, int ncell, int* __restrict__ block_counts) {
  __shared__ int lds[256];
  const int c0 = blockIdx.x * 1024 + threadIdx.x * 4;
  int n = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) n += (c0 + u < ncell && src[c0 + u] > 0.f) ? 1 : 0;
  int total;
  block_exclusive_scan(n, lds, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

// exclusive scan of the per-workgroup counts (in place) by ONE workgroup; count[0] = number of active cells
__global__ void __launch_bounds__(256) sp_scan_kernel(int* __restrict__ block_counts, int nb, int* __restrict__ count) {
  __shared__ int lds[256];
  int running = 0;
  for (int b0 = 0; b0 < nb; b0 += 256) {
    const int i = b0 + threadIdx.x;
    const int v = i < nb ? block_counts[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, &total);
    if (i < nb) block_counts[i] = running + ex;
    running += total;
  }
  if (threadIdx.x == 0) count[0] = running;
}

__global__ void __launch_bounds__(256) sp_write_kernel(const float* __restrict__ src, int ncell, const int* __restrict__ block_offsets,
                                                        int* __restrict__ imap, int* __restrict__ sites, int cap) {
  __shared__ int lds[256];
  const int c0 = blockIdx.x * 1024 + threadIdx.x * 4;
  bool act[4];
  int n = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) { act[u] = c0 + u < ncell && src[c0 + u] > 0.f; n += act[u] ? 1 : 0; }
  int total;
  int row = block_offsets[blockIdx.x] + block_exclusive_scan(n, lds, &total);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (c0 + u >= ncell) break;
    if (act[u]) {
      imap[c0 + u] = row < cap ? row : -1;          // (rows past the capacity are dropped; the host sized cap from the count)
      if (row < cap) sites[row] = c0 + u;
      ++row;
    } else {
      imap[c0 + u] = -1;
    }
  }
}

// stride-2 coordinate rule: coarse cell (Y, X) is active iff one of the fine cells (2Y..2Y+1, 2X..2X+1) is
// (odd h / w: the coarse grid has ceil(h / 2) x ceil(w / 2) cells -- MinkowskiEngine's floor(c / 2) coordinates of the last odd row /
// column; the children outside the fine grid do not exist)
__global__ void __launch_bounds__(256) sp_pool_cells_kernel(const int* __restrict__ imap, int B, int h, int w, float* __restrict__ mask_out) {
  const int h2 = (h + 1) / 2, w2 = (w + 1) / 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * h2 * w2) return;
  const int b = i / (h2 * w2), rem = i - b * h2 * w2;
  const int Y = rem / w2, X = rem - Y * w2;
  const int* m = imap + (size_t)b * h * w;
  const bool y1 = 2 * Y + 1 < h, x1 = 2 * X + 1 < w;
  const bool any = m[(2 * Y) * w + 2 * X] >= 0 || (x1 && m[(2 * Y) * w + 2 * X + 1] >= 0) || (y1 && m[(2 * Y + 1) * w + 2 * X] >= 0) ||
                   (y1 && x1 && m[(2 * Y + 1) * w + 2 * X + 1] >= 0);
  mask_out[i] = any ? 1.f : 0.f;
}

__global__ void __launch_bounds__(256) sp_neighbors_kernel(const int* __restrict__ imap, const int* __restrict__ sites,
                                                            const int* __restrict__ count, int cap, int h, int w, int ks,
                                                            int* __restrict__ nbr) {
  const int KK = ks * ks, r = ks / 2;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= cap * KK) return;
  const int n = e / KK, i = e - n * KK;
  int v = -1;
  if (n < count[0]) {
    const int cell = sites[n];
    const int b = cell / (h * w), rem = cell - b * h * w;
    const int y = rem / w, x = rem - y * w;
    const int yy = y + (i % ks) - r, xx = x + (i / ks) - r;
    if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = imap[(size_t)b * h * w + yy * w + xx];
  }
  nbr[e] = v;
}

// ---- sparse convolution (forward and backward-data) --------------------------------------------------------------------
// out[n][co] = sum_i sum_ci feats[nbr[n][flip ? KK-1-i : i]][ci] * kern[i][ci][co]
// CK = channels of a gathered LDS tile (one stage): 32, 64 or 128 -- every stage costs a gather round trip and two barriers, so
// wide layers take 128 channels (64 MFMAs per wave) per stage
// grid (site tiles of 32, output-channel groups of 4 * MAXT tiles): a layer with few sites and many channels (the 6x20 level: 480
// sites, 1024 -> 1024) is spread over the chip by its channel groups; the gather is repeated per group, the weights are not.
template <int MAXT, int CK>
__global__ void __launch_bounds__(256) sparse_conv_kernel(const float* __restrict__ feats, const float* __restrict__ kern,
                                                           const int* __restrict__ nbr, const int* __restrict__ count,
                                                           float* __restrict__ out, int cap, int Cin, int Cout, int KK, int flip) {
  __shared__ int nb[32 * 49];
  __shared__ int tapany[49];
  __shared__ float tile[32][CK + 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int n0 = blockIdx.x * 32;
  const int t0 = blockIdx.y * 4 * MAXT;             // first output-channel tile of this workgroup
  int cnt = count[0];
  if (cnt > cap) cnt = cap;
  const int ntile = (Cout + 31) / 32;
  if (n0 >= cnt) {            // rows past the active set: zero (downstream statistics sum whole columns)
    const int c0 = t0 * 32, nc = (Cout - c0 < 128 * MAXT) ? Cout - c0 : 128 * MAXT;
    for (int e = tid; e < 32 * nc; e += 256) {
      const int s = e / nc;
      if (n0 + s < cap) out[(size_t)(n0 + s) * Cout + c0 + (e - s * nc)] = 0.f;
    }
    return;
  }
  if (tid < KK) tapany[tid] = 0;
  __syncthreads();
  for (int e = tid; e < 32 * KK; e += 256) {
    const int s = e / KK, i = e - s * KK;
    const int n = n0 + s;
    const int v = n < cnt ? nbr[(size_t)n * KK + (flip ? KK - 1 - i : i)] : -1;
    nb[s * KK + i] = v;
    if (v >= 0) tapany[i] = 1;            // benign race: every writer stores 1
  }
  __syncthreads();

  f32x16 acc[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int gs = tid >> 3, gq = tid & 7;            // gather: site, 4-channel group
  const bool vec4 = (Cin & 3) == 0;
  // stages = (offset with at least one active neighbour in the tile, 32-channel chunk); the rows of stage s+1 are fetched into
  // registers while the matrix cores work on stage s (one workgroup alone would otherwise wait out every gather)
  auto next_stage = [&](int& tap, int& ci0) {       // advance to the next stage; tap == KK when done
    ci0 += CK;
    if (ci0 >= Cin) {
      ci0 = 0;
      do { ++tap; } while (tap < KK && !tapany[tap]);
    }
  };
  constexpr int NV = CK / 32;                       // float4 groups per thread and stage
  auto fetch = [&](int tap, int ci0, float (&v)[4 * NV]) {
#pragma unroll
    for (int u = 0; u < 4 * NV; ++u) v[u] = 0.f;
    if (tap >= KK) return;
    const int row = nb[gs * KK + tap];
    if (row < 0) return;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int c = ci0 + 32 * u + 4 * gq;
      const float* p = feats + (size_t)row * Cin + c;
      if (vec4 && c + 3 < Cin) {
        const float4 q = *reinterpret_cast<const float4*>(p);
        v[4 * u] = q.x; v[4 * u + 1] = q.y; v[4 * u + 2] = q.z; v[4 * u + 3] = q.w;
      } else {
        if (c < Cin) v[4 * u] = p[0];
        if (c + 1 < Cin) v[4 * u + 1] = p[1];
        if (c + 2 < Cin) v[4 * u + 2] = p[2];
        if (c + 3 < Cin) v[4 * u + 3] = p[3];
      }
    }
  };
  int tap = 0, ci0 = 0;
  while (tap < KK && !tapany[tap]) ++tap;
  float pre[4 * NV];
  fetch(tap, ci0, pre);
  while (tap < KK) {
    __syncthreads();                                // the previous tile has been consumed
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      float* tp = &tile[gs][32 * u + 4 * gq];
      tp[0] = pre[4 * u]; tp[1] = pre[4 * u + 1]; tp[2] = pre[4 * u + 2]; tp[3] = pre[4 * u + 3];
    }
    __syncthreads();
    int kmax = Cin - ci0;
    if (kmax > CK) kmax = CK;
    const float* kbase = kern + ((size_t)tap * Cin + ci0) * Cout;
    next_stage(tap, ci0);
    fetch(tap, ci0, pre);                           // in flight during the MFMAs below
    // a fixed trip count (channels past kmax contribute zeros): the 16 x MAXT weight loads of a chunk are issued back to back
    // instead of one dependent load per MFMA
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int t = t0 + wave + 4 * j;
      if (t >= ntile) continue;                      // wave-uniform
      const int co = t * 32 + l32;
      float b[CK / 2];
#pragma unroll
      for (int q = 0; q < CK / 2; ++q) {
        const int k = 2 * q + half;
        b[q] = (k < kmax && co < Cout) ? kbase[(size_t)k * Cout + co] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < CK / 2; ++q) acc[j] = pnsfm_mfma_32x32x2(tile[l32][2 * q + half], b[q], acc[j]);
    }
  }
  // D row = (r&3) + 8*(r>>2) + 4*half -> site, col = l32 -> output channel: 128 contiguous bytes per half-wave
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int t = t0 + wave + 4 * j;
    if (t >= ntile) continue;
    const int co = t * 32 + l32;
    if (co >= Cout) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (n < cap) out[(size_t)n * Cout + co] = n < cnt ? acc[j][r] : 0.f;
    }
  }
}

// ---- weight gradient: dK[tap][ci][co] += sum_{n in split} feats[nbr[n][tap]][ci] * dout[n][co] ----------------------------
// grid (KK, ci tiles, co groups * Z); a wave owns one 32 x 32 (ci x co) tile and walks its share of the sites two per MFMA
__global__ void __launch_bounds__(256) sparse_wgrad_kernel(const float* __restrict__ feats, const float* __restrict__ dout,
                                                            const int* __restrict__ nbr, const int* __restrict__ count,
                                                            float* __restrict__ dkern, int cap, int Cin, int Cout, int KK, int Z,
                                                            int per_split) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = PNSFM_UNIFORM(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int tap = blockIdx.x, ci0 = blockIdx.y * 32;
  const int cog = blockIdx.z / Z, z = blockIdx.z - cog * Z;
  const int co0 = (cog * 4 + wave) * 32;
  if (co0 >= Cout) return;
  int cnt = count[0];
  if (cnt > cap) cnt = cap;
  const int beg = z * per_split;
  int end = beg + per_split;
  if (end > cnt) end = cnt;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ci = ci0 + l32, co = co0 + l32;
  // 32 sites (16 MFMAs) per trip: the 16 neighbour indices, then the 32 operand loads they address, are in flight together
  constexpr int U = 16;
  for (int n = beg; n < end; n += 2 * U) {
    int row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = n + 2 * u + half;
      row[u] = s < end ? nbr[(size_t)s * KK + tap] : -1;
    }
    float a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = n + 2 * u + half;
      a[u] = (row[u] >= 0 && ci < Cin) ? feats[(size_t)row[u] * Cin + ci] : 0.f;
      b[u] = (s < end && co < Cout) ? dout[(size_t)s * Cout + co] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = pnsfm_mfma_32x32x2(a[u], b[u], acc);
  }
  if (beg >= end) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = ci0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (c < Cin && co < Cout) {
      atomicAdd(dkern + ((size_t)tap * Cin + c) * Cout + co, acc[r]);      // the entry point zero-fills dkern; shares of the sites meet here
    }
  }
}

// ---- max pooling (3, stride 2) ---------------------------------------------------------------------------------------
// out row m = coarse cell (b, Y, X): max over the ACTIVE fine cells (2Y + dy, 2X + dx), dy, dx in {-1, 0, 1}; arg = their row
__global__ void __launch_bounds__(256) sp_maxpool_fwd_kernel(const float* __restrict__ fin, const int* __restrict__ imap_in,
                                                              const int* __restrict__ sites_out, const int* __restrict__ count_out,
                                                              float* __restrict__ fout, int* __restrict__ arg, int cap, int C,
                                                              int h, int w) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)cap * C) return;
  const int m = (int)(e / C), c = (int)(e - (size_t)m * C);
  float best = 0.f;
  int barg = -1;
  if (m < count_out[0]) {
    const int h2 = (h + 1) / 2, w2 = (w + 1) / 2;
    const int cell = sites_out[m];
    const int b = cell / (h2 * w2), rem = cell - b * h2 * w2;
    const int Y = rem / w2, X = rem - Y * w2;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int y = 2 * Y + dy, x = 2 * X + dx;
        if (y < 0 || y >= h || x < 0 || x >= w) continue;
        const int row = imap_in[(size_t)b * h * w + y * w + x];
        if (row < 0) continue;
        const float v = fin[(size_t)row * C + c];
        if (barg < 0 || v > best) { best = v; barg = row; }
      }
  }
  fout[e] = barg >= 0 ? best : 0.f;
  arg[e] = barg;
}

__global__ void __launch_bounds__(256) sp_maxpool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ arg,
                                                              float* __restrict__ din, int cap, int C) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)cap * C) return;
  const int row = arg[e];
  if (row >= 0) atomicAdd(din + (size_t)row * C + (e % C), dout[e]);      // 3x3 windows of stride 2 overlap: several outputs may pick one input
}

// ---- rows <-> dense NCHW -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sp_densify_kernel(const float* __restrict__ feats, const int* __restrict__ imap,
                                                          float* __restrict__ dense, int B, int C, int hw) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)B * C * hw) return;
  const int cell = (int)(e % hw);
  const size_t bc = e / hw;
  const int c = (int)(bc % C), b = (int)(bc / C);
  const int row = imap[(size_t)b * hw + cell];
  dense[e] = row >= 0 ? feats[(size_t)row * C + c] : 0.f;
}

// rows[n][c] = dense[b][c][cell(n)] for n < count, 0 otherwise
__global__ void __launch_bounds__(256) sp_gather_kernel(const float* __restrict__ dense, const int* __restrict__ sites,
                                                         const int* __restrict__ count, float* __restrict__ rows, int cap, int C,
                                                         int hw) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)cap * C) return;
  const int n = (int)(e / C), c = (int)(e - (size_t)n * C);
  float v = 0.f;
  if (n < count[0]) {
    const int cell = sites[n];
    const int b = cell / hw;
    v = dense[((size_t)b * C + c) * hw + (cell - b * hw)];
  }
  rows[e] = v;
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

size_t pnsfm_sparse_compact_ws_ints(int ncell) { return (size_t)ceil_div(ncell, 1024) + 1; }

int pnsfm_sparse_compact(const float* src, int ncell, int* imap, int* sites, int cap, int* count, int* ws, void* stream) {
  if (ncell <= 0 || cap < 0) { set_error("sparse_compact: bad size"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const int nb = ceil_div(ncell, 1024);
  PNSFM_LAUNCH(sp_count_kernel, dim3(nb), dim3(256), 0, s, src, ncell, ws);
  PNSFM_LAUNCH(sp_scan_kernel, dim3(1), dim3(256), 0, s, ws, nb, count);
  PNSFM_LAUNCH(sp_write_kernel, dim3(nb), dim3(256), 0, s, src, ncell, (const int*)ws, imap, sites, cap);
  return check_launch("sparse_compact");
}

int pnsfm_sparse_pool_cells(const int* imap, int B, int h, int w, float* mask_out, void* stream) {
  if (h < 1 || w < 1 || B <= 0) { set_error("sparse_pool_cells: bad grid (h=%d w=%d)", h, w); return -1; }
  const int n = B * ((h + 1) / 2) * ((w + 1) / 2);
  PNSFM_LAUNCH(sp_pool_cells_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, imap, B, h, w, mask_out);
  return check_launch("sparse_pool_cells");
}

int pnsfm_sparse_neighbors(const int* imap, const int* sites, const int* count, int cap, int h, int w, int ks, int* nbr,
                           void* stream) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("sparse_neighbors: kernel size %d", ks); return -1; }
  if (cap == 0) return 0;
  PNSFM_LAUNCH(sp_neighbors_kernel, dim3(ceil_div(cap * ks * ks, 256)), dim3(256), 0, (hipStream_t)stream, imap, sites, count, cap,
               h, w, ks, nbr);
  return check_launch("sparse_neighbors");
}

int pnsfm_sparse_conv(const float* feats, const float* kern, const int* nbr, const int* count, float* out, int cap, int Cin,
                      int Cout, int ks, int flip, void* stream) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("sparse_conv: kernel size %d", ks); return -1; }
  if (Cin <= 0 || Cout <= 0) { set_error("sparse_conv: channels %d -> %d", Cin, Cout); return -1; }
  if (cap == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int KK = ks * ks, ntile = ceil_div(Cout, 32), sblocks = ceil_div(cap, 32);
  // one tile per wave (128 output channels per workgroup) unless the site tiles alone already fill the chip several times over
  const int maxt = (sblocks >= 2048 && ntile > 4) ? 2 : 1;
  const dim3 grid(sblocks, ceil_div(ntile, 4 * maxt));
#define PNSFM_SP_LAUNCH(MT, CKv) PNSFM_LAUNCH((sparse_conv_kernel<MT, CKv>), grid, dim3(256), 0, s, feats, kern, nbr, count, out, cap, Cin, Cout, KK, flip)
  if (Cin > 64) { if (maxt == 1) PNSFM_SP_LAUNCH(1, 128); else PNSFM_SP_LAUNCH(2, 128); }
  else if (Cin > 32) { if (maxt == 1) PNSFM_SP_LAUNCH(1, 64); else PNSFM_SP_LAUNCH(2, 64); }
  else { if (maxt == 1) PNSFM_SP_LAUNCH(1, 32); else PNSFM_SP_LAUNCH(2, 32); }
#undef PNSFM_SP_LAUNCH
  return check_launch("sparse_conv");
}

int pnsfm_sparse_conv_backward_weight(const float* feats, const float* dout, const int* nbr, const int* count, float* dkern,
                                      int cap, int Cin, int Cout, int ks, void* stream) {
  if (ks != 1 && ks != 3 && ks != 5 && ks != 7) { set_error("sparse_conv_backward_weight: kernel size %d", ks); return -1; }
  hipStream_t s = (hipStream_t)stream;
  const int KK = ks * ks;
  const size_t n = (size_t)KK * Cin * Cout;
  int e = (int)hipMemsetAsync(dkern, 0, n * sizeof(float), s);      // split launches accumulate; an empty site list leaves zeros
  if (e) { set_error("sparse_conv_backward_weight: memset failed"); return e; }
  if (cap == 0) return 0;
  const int ciT = ceil_div(Cin, 32), coG = ceil_div(ceil_div(Cout, 32), 4);
  int Z = ceil_div(1024, KK * ciT * coG);           // ~1024 workgroups overall ...
  const int maxZ = ceil_div(cap, 64);               // ... but at least 64 sites per share
  if (Z > maxZ) Z = maxZ;
  if (Z < 1) Z = 1;
  const int per_split = round_up(ceil_div(cap, Z), 32);
  Z = ceil_div(cap, per_split);
  PNSFM_LAUNCH(sparse_wgrad_kernel, dim3(KK, ciT, coG * Z), dim3(256), 0, s, feats, dout, nbr, count, dkern, cap, Cin, Cout, KK, Z,
               per_split);
  return check_launch("sparse_conv_backward_weight");
}

int pnsfm_sparse_maxpool_forward(const float* fin, const int* imap_in, const int* sites_out, const int* count_out, float* fout,
                                 int* arg, int cap, int C, int h, int w, void* stream) {
  if (cap == 0) return 0;
  PNSFM_LAUNCH(sp_maxpool_fwd_kernel, dim3((unsigned)ceil_div_sz((size_t)cap * C, 256)), dim3(256), 0, (hipStream_t)stream, fin, imap_in,
               sites_out, count_out, fout, arg, cap, C, h, w);
  return check_launch("sparse_maxpool_forward");
}

int pnsfm_sparse_maxpool_backward(const float* dout, const int* arg, float* din, int cap_out, int cap_in, int C, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (cap_in > 0) {
    int e = (int)hipMemsetAsync(din, 0, (size_t)cap_in * C * sizeof(float), s);
    if (e) { set_error("sparse_maxpool_backward: memset failed"); return e; }
  }
  if (cap_out == 0) return 0;
  PNSFM_LAUNCH(sp_maxpool_bwd_kernel, dim3((unsigned)ceil_div_sz((size_t)cap_out * C, 256)), dim3(256), 0, s, dout, arg, din, cap_out, C);
  return check_launch("sparse_maxpool_backward");
}

int pnsfm_sparse_densify(const float* feats, const int* imap, float* dense, int B, int C, int hw, void* stream) {
  PNSFM_LAUNCH(sp_densify_kernel, dim3((unsigned)ceil_div_sz((size_t)B * C * hw, 256)), dim3(256), 0, (hipStream_t)stream, feats, imap,
               dense, B, C, hw);
  return check_launch("sparse_densify");
}

int pnsfm_sparse_gather(const float* dense, const int* sites, const int* count, float* rows, int cap, int C, int hw, void* stream) {
  if (cap == 0) return 0;
  PNSFM_LAUNCH(sp_gather_kernel, dim3((unsigned)ceil_div_sz((size_t)cap * C, 256)), dim3(256), 0, (hipStream_t)stream, dense, sites, count,
               rows, cap, C, hw);
  return check_launch("sparse_gather");
}

}  // extern "C"
