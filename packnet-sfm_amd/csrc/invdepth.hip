// invdepth.hip -- the InvDepth head as one fused kernel each way.
//
//   y = sigmoid(conv3x3(zero_pad1(x)) + b) / min_depth,  x:[B,C,H,W] -> y:[B,1,H,W]
//   /root/reference/packnet_sfm/networks/layers/packnet/layers01.py:98-122 (InvDepth: pad, conv1, Sigmoid, / min_depth)
//
// A convolution with ONE output channel has nothing for the matrix cores (a 32-row MFMA tile would be 3 % used: the
// generic implicit-GEMM kernel ran these four layers at 1-3 TFLOP/s); it is a streaming reduction over the input
// channels, bound by reading x once.  Algorithmic bytes: forward C*H*W*4 read + H*W*4 write per image; backward
// C*H*W*4 read (x) + C*H*W*4 write (dx) + small.
//
// forward : a block owns 64 consecutive pixels x 4 channel quarters (lanes along x -> coalesced rows); every thread
//           accumulates 9 taps x C/4 channels with wave-uniform weights (scalar loads), the four quarters meet in LDS.
// backward: dz = dy * y * (1 - y*min_depth) comes from the caller (pnsfm_invdepth_act_backward).  With
//           D_t(q) = dz[qy-ky+1][qx-kx+1] (zero outside the image) both gradients use the SAME nine numbers at input
//           pixel q:   dx[c][q] = sum_t w[c][t] * D_t(q),     dw[c][t] = sum_q x[c][q] * D_t(q),   db = sum_q dz[q]
//           so one kernel reads x once, writes dx once and keeps 8 channels x 9 taps of dw in registers per thread
//           (block reduction through LDS, one fp32 atomic per value per block).
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

__global__ void __launch_bounds__(256) invdepth_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 int C, int H, int W, float inv_min_depth) {
  __shared__ float part[4][64];
  const int tid = threadIdx.x, lane = tid & 63, quarter = tid >> 6;
  const int HW = H * W;
  const int pix = blockIdx.x * 64 + lane;
  const bool active = pix < HW;
  const int py = active ? pix / W : 0, px = active ? pix - py * W : 0;
  const unsigned kOut = 0x7fffffffu;
  unsigned off[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int yy = py + ky - 1, xx = px + kx - 1;
      const bool ok = active && yy >= 0 && yy < H && xx >= 0 && xx < W;
      off[ky * 3 + kx] = ok ? (unsigned)(yy * W + xx) * 4u : kOut;
    }
  const int cq = (C + 3) / 4;
  const int c0 = quarter * cq;
  const int c1 = (c0 + cq < C) ? c0 + cq : C;
  float acc = 0.f;
  for (int c = c0; c < c1; ++c) {
    const int cu = PNSFM_UNIFORM(c);      // the channel is wave-uniform (one quarter per wave): scalar weight loads
    const pnsfm_buf pb = pnsfm_make_buf(x + ((size_t)blockIdx.z * C + cu) * HW, (unsigned)HW * 4u);
    const float* wc = w + cu * 9;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = pnsfm_buf_load(pb, off[t], 0u);
#pragma unroll
    for (int t = 0; t < 9; ++t) acc = fmaf(wc[t], v[t], acc);
  }
  part[quarter][lane] = acc;
  __syncthreads();
  if (quarter == 0 && active) {
    const float z = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane] + bias[0];
    y[(size_t)blockIdx.z * HW + pix] = inv_min_depth / (1.f + expf(-z));
  }
}

// Round 5 (250 MB per launch against 45 MB algorithmic): the kernel above gives every 64-pixel block its own three
// image rows, and consecutive block indices go round robin over the 8 XCDs -- so every row of x is fetched by three different L2s
// (the blocks of rows y-1, y, y+1) plus the partial lines of the one-pixel halo.  The strip form: a block owns R rows x 64 columns;
// per channel a lane loads the R + 2 centre values of its column ONCE and takes the left / right neighbours from the adjacent lanes
// (wave shifts; lanes 0 and 63 fetch the strip's halo column with a second, two-lane load), so x is read (R + 2) / R times instead of
// 3 x 3 times through the texture path and 3+ times from HBM; the logical block order (x strips, then y strips, then images) is cut
// into one contiguous range per XCD, so the strips that share halo rows / halo lines share an L2.  The per-pixel accumulation order
// (channels of a quarter in sequence, taps 0..8, then the four quarters + bias) is the kernel above's: results are bit-identical.
// Used for the maps with enough strips to fill the chip (192x640: R = 8, 96x320: R = 4); the low-resolution heads stay above.
template <int R>
__global__ void __launch_bounds__(256) invdepth_conv_fwd_strip_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                                       int C, int H, int W, int sx_n, int sy_n, float inv_min_depth) {
  __shared__ float part[4][R][64];
  const int tid = threadIdx.x, lane = tid & 63, quarter = tid >> 6;
  const int HW = H * W;
  const unsigned L = pnsfm_xcd_logical_block(blockIdx.x, gridDim.x);
  const int sx = (int)(L % (unsigned)sx_n), sy = (int)((L / (unsigned)sx_n) % (unsigned)sy_n), b = (int)(L / (unsigned)(sx_n * sy_n));
  const int px = sx * 64 + lane, y0 = sy * R;
  const unsigned kOut = 0x7fffffffu;
  // byte offsets inside a channel plane: the lane's own column, and (lanes 0 / 63 only) the column left / right of the strip
  const int hx = lane == 0 ? px - 1 : (lane == 63 ? px + 1 : -1);
  unsigned offc[R + 2], offh[R + 2];
#pragma unroll
  for (int r = 0; r < R + 2; ++r) {
    const int yy = y0 + r - 1;
    const bool rok = yy >= 0 && yy < H;
    offc[r] = (rok && px < W) ? (unsigned)(yy * W + px) * 4u : kOut;
    offh[r] = (rok && hx >= 0 && hx < W) ? (unsigned)(yy * W + hx) * 4u : kOut;
  }
  const int cq = (C + 3) / 4;
  const int c0 = quarter * cq;
  const int c1 = (c0 + cq < C) ? c0 + cq : C;
  float acc[R];
#pragma unroll
  for (int i = 0; i < R; ++i) acc[i] = 0.f;
  for (int c = c0; c < c1; ++c) {
    const int cu = PNSFM_UNIFORM(c);
    const pnsfm_buf pb = pnsfm_make_buf(x + ((size_t)b * C + cu) * HW, (unsigned)HW * 4u);
    const float* wc = w + cu * 9;
    float vc[R + 2], vh[R + 2], vl[R + 2], vr[R + 2];
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      vc[r] = pnsfm_buf_load(pb, offc[r], 0u);
      vh[r] = pnsfm_buf_load(pb, offh[r], 0u);
    }
#pragma unroll
    for (int r = 0; r < R + 2; ++r) {
      const float up = __shfl_up(vc[r], 1), dn = __shfl_down(vc[r], 1);
      vl[r] = lane == 0 ? vh[r] : up;
      vr[r] = lane == 63 ? vh[r] : dn;
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        acc[i] = fmaf(wc[ky * 3 + 0], vl[i + ky], acc[i]);
        acc[i] = fmaf(wc[ky * 3 + 1], vc[i + ky], acc[i]);
        acc[i] = fmaf(wc[ky * 3 + 2], vr[i + ky], acc[i]);
      }
  }
#pragma unroll
  for (int i = 0; i < R; ++i) part[quarter][i][lane] = acc[i];
  __syncthreads();
  // 256 threads finish R x 64 pixels: thread -> (row = tid / 64 + 4 * k, column = lane)
#pragma unroll
  for (int i = quarter; i < R; i += 4) {
    const int yy = y0 + i;
    if (yy < H && px < W) {
      const float z = part[0][i][lane] + part[1][i][lane] + part[2][i][lane] + part[3][i][lane] + bias[0];
      y[(size_t)b * HW + (size_t)yy * W + px] = inv_min_depth / (1.f + expf(-z));
    }
  }
}

constexpr int kIdCh = 8;                    // channels per thread in the backward kernel
constexpr int kIdVals = kIdCh * 9 + 1;      // + the bias sum
constexpr int kIdRow = 256 + 16;            // padded LDS row (conflict-free 16-lane strided reads)
constexpr int kIdPass = 16;

__global__ void __launch_bounds__(256) invdepth_conv_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ dz, float* __restrict__ dx,
                                                                 float* __restrict__ part, int C, int H, int W, int ppt) {
  __shared__ float red[kIdPass * kIdRow];
  const int tid = threadIdx.x;
  const int HW = H * W;
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * kIdCh;
  const pnsfm_buf zb = pnsfm_make_buf(dz + (size_t)b * HW, (unsigned)HW * 4u);
  const unsigned kOut = 0x7fffffffu;
  float acc[kIdCh][9], bsum = 0.f;
#pragma unroll
  for (int c = 0; c < kIdCh; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[c][t] = 0.f;
  const bool want_db = blockIdx.y == 0;
  for (int it = 0; it < ppt; ++it) {
    const int q = (blockIdx.x * ppt + it) * 256 + tid;
    const bool active = q < HW;
    const int qy = active ? q / W : 0, qx = active ? q - qy * W : 0;
    float D[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = qy - ky + 1, xx = qx - kx + 1;
        const bool ok = active && yy >= 0 && yy < H && xx >= 0 && xx < W;
        D[ky * 3 + kx] = pnsfm_buf_load(zb, ok ? (unsigned)(yy * W + xx) * 4u : kOut, 0u);
      }
    if (want_db) bsum += D[4];               // tap (1,1) is dz[q] itself
    const unsigned qoff = active ? (unsigned)q * 4u : kOut;
#pragma unroll
    for (int c = 0; c < kIdCh; ++c) {
      const int ch = c0 + c;                 // wave-uniform
      if (ch < C) {
        const pnsfm_buf xb = pnsfm_make_buf(x + ((size_t)b * C + ch) * HW, (unsigned)HW * 4u);
        const float xv = pnsfm_buf_load(xb, qoff, 0u);
        const float* wc = w + ch * 9;
        float g = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          g = fmaf(wc[t], D[t], g);
          acc[c][t] = fmaf(xv, D[t], acc[c][t]);
        }
        if (active) dx[((size_t)b * C + ch) * HW + q] = g;
      }
    }
  }
  // ---- block reduction of the 73 partials, kIdPass values per pass (same scheme as conv3d_wgrad_kernel)
  const int rv = tid >> 4, rj = tid & 15;
  constexpr int npass = (kIdVals + kIdPass - 1) / kIdPass;
#pragma unroll
  for (int pass = 0; pass < npass; ++pass) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kIdPass; ++u) {
      const int v = pass * kIdPass + u;
      float val = 0.f;
      if (v < kIdCh * 9) val = acc[v / 9][v % 9];
      else if (v == kIdCh * 9) val = bsum;
      red[u * kIdRow + tid] = val;
    }
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[rv * kIdRow + rj + 16 * i];
    s += __shfl_xor(s, 8);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 1);
    if (rj == 0) {
      // one slot per (pixel block, image, channel group, value): plain stores, no zero-fill; invdepth_conv_bwd_finish_kernel adds
      // the pixel blocks in a fixed order (round 3: fp32 atomics into a zero-filled dw / db)
      const int v = pass * kIdPass + rv;
      if (v < kIdVals) part[(((size_t)blockIdx.z * gridDim.x + blockIdx.x) * gridDim.y + blockIdx.y) * kIdVals + v] = s;
    }
  }
}

// dw[ch][t] (and db) = sum over the (image, pixel block) partials of the channel's group, in index order
__global__ void __launch_bounds__(256) invdepth_conv_bwd_finish_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                        float* __restrict__ db, int C, int npart, int cgroups) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o > C * 9) return;
  const int ch = o / 9, t = o - ch * 9;
  const int cg = o < C * 9 ? ch / kIdCh : 0, v = o < C * 9 ? (ch - cg * kIdCh) * 9 + t : kIdCh * 9;
  float s = 0.f;
  for (int p = 0; p < npart; ++p) s += part[((size_t)p * cgroups + cg) * kIdVals + v];
  if (o < C * 9) dw[o] = s; else if (db) *db = s;
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_invdepth_conv_forward(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W,
                                float min_depth, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("invdepth_conv_forward: bad shape"); return -1; }
  if (min_depth <= 0.f) { set_error("invdepth_conv_forward: min_depth must be > 0"); return -1; }
  if ((size_t)H * W * 4 >= 0x7fffffffull) { set_error("invdepth_conv_forward: plane exceeds the 2 GiB buffer window"); return -1; }
  // strips of R rows x 64 columns where they fill the chip (see invdepth_conv_fwd_strip_kernel); PNSFM_INVDEPTH_STRIP = 0: never,
  // 4 / 8: always, with that R (tests)
  const char* e = getenv("PNSFM_INVDEPTH_STRIP");
  const int mode = e ? atoi(e) : 1;
  const int sx_n = ceil_div(W, 64);
  const bool wide = mode == 1 && W >= 64 && sx_n * 64 - W <= 16;        // at most a quarter of the last strip idle
  int R = 0;
  if (mode == 4 || mode == 8) R = mode;
  else if (wide && (long)sx_n * ceil_div(H, 8) * B >= 768) R = 8;
  else if (wide && (long)sx_n * ceil_div(H, 4) * B >= 400) R = 4;
  if (R) {
    const int sy_n = ceil_div(H, R);
    if (R == 8)
      PNSFM_LAUNCH(invdepth_conv_fwd_strip_kernel<8>, dim3(sx_n * sy_n * B), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, C, H, W,
                   sx_n, sy_n, 1.0f / min_depth);
    else
      PNSFM_LAUNCH(invdepth_conv_fwd_strip_kernel<4>, dim3(sx_n * sy_n * B), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, C, H, W,
                   sx_n, sy_n, 1.0f / min_depth);
  } else {
    PNSFM_LAUNCH(invdepth_conv_fwd_kernel, dim3(ceil_div(H * W, 64), 1, B), dim3(256), 0, (hipStream_t)stream, x, w, bias, y,
                 C, H, W, 1.0f / min_depth);
  }
  return check_launch("invdepth_conv_forward");
}

int pnsfm_invdepth_conv_backward(const float* x, const float* w, const float* dz, float* dx, float* dw, float* db, int B,
                                 int C, int H, int W, void* stream) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) { set_error("invdepth_conv_backward: bad shape"); return -1; }
  if ((size_t)H * W * 4 >= 0x7fffffffull) { set_error("invdepth_conv_backward: plane exceeds the 2 GiB buffer window"); return -1; }
  hipStream_t s = (hipStream_t)stream;
  // pixels per thread: up to 16 (amortises the 73-value block reduction) while the grid keeps >= ~2 blocks per CU
  const int HW = H * W, cgroups = ceil_div(C, kIdCh);
  int ppt = 16;
  while (ppt > 1 && (long)ceil_div(HW, 256 * ppt) * cgroups * B < 512) ppt >>= 1;
  const int gx = ceil_div(HW, 256 * ppt);
  ScratchLease lease(s, (size_t)gx * B * cgroups * kIdVals * sizeof(float));
  float* const part = lease.as<float>();
  if (!part) return -1;
  PNSFM_LAUNCH(invdepth_conv_bwd_kernel, dim3(gx, cgroups, B), dim3(256), 0, s, x, w, dz, dx, part, C, H, W, ppt);
  int rc = check_launch("invdepth_conv_backward");
  if (rc) return rc;
  PNSFM_LAUNCH(invdepth_conv_bwd_finish_kernel, dim3(ceil_div(C * 9 + 1, 256)), dim3(256), 0, s, (const float*)part, dw, db,
               C, gx * B, cgroups);
  return check_launch("invdepth_conv_backward (finish)");
}

}  // extern "C"
