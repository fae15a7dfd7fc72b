/* pnsfm.h -- C ABI of libpnsfm_hip.so: the MI355X (gfx950) kernels behind the PackNet-SfM
 * data-parallel training hot path (PackNet01 depth network + multi-view photometric loss).
 *
 * The reference (TRI-ML/packnet-sfm, /root/reference) has no FFI of its own: every op it
 * runs is a torch.nn / torch.nn.functional call.  Each entry point below therefore names
 * the reference *call site* (file:line under /root/reference) whose arithmetic it replaces.
 * The Python modules in packnet-sfm_amd/packnet_sfm/ bind these with ctypes and re-expose
 * them behind the reference's own module API (PackNet01, MultiViewPhotometricLoss, ...).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (unless typed otherwise), NCHW;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, nothing syncs;
 *   - outputs and named workspaces are caller-allocated.  The ONE exception: the two-stage reductions (pixel-split weight
 *     gradients, the loss scalars) keep their partial sums in a grow-only scratch buffer the library hipMalloc's itself, one per
 *     (device, stream), outside the caller's allocator (csrc/api.hip: a few KB .. tens of MB; a buffer that has to grow is
 *     replaced and the old one freed once the stream has passed it).  Worst case per stream: the K-split slabs of the largest
 *     forward / backward-data launch (splits x output bytes; <= 64 MB on PackNet01's shapes) or, under PNSFM_CONV_MATH=f32, the
 *     pixel-split [dw | dbias] slabs of a weight gradient, which the library caps at 128 MB per launch.  Entry points that need this scratch fail with an
 *     error while `stream` is being captured into a hipGraph (no capture path since round 4);
 *   - return value: 0 on success, non-zero on error (pnsfm_last_error() has the message).
 */
#ifndef PNSFM_H
#define PNSFM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int pnsfm_version(void);
const char* pnsfm_last_error(void);
/* "gfx950" for the product build, "emu" for the host-emulated test build (tests/emu). */
const char* pnsfm_build_target(void);

/* ---- 2-D convolution (stride 1, zero pad k/2) as fp32-MFMA implicit GEMM -------------------
 * replaces nn.Conv2d + nn.ConstantPad2d in Conv2D/ResidualConv/InvDepth/Pack/Unpack blocks:
 *   packnet_sfm/networks/layers/packnet/layers01.py:28-36, :57-60, :115-121, :235-246, :274-281
 * Weights are consumed in a packed layout [k*k][KP][MP] (M = output channel fastest) produced by
 * pnsfm_conv2d_pack_weights from the reference layout [Cout][Cin][k][k] (state-dict contract,
 * packnet_sfm/utils/load.py:114-163).  wp_fwd is used by _forward, wp_bwd by _backward_data. */
size_t pnsfm_conv2d_packed_elems_fwd(int Cin, int Cout, int ks);
size_t pnsfm_conv2d_packed_elems_bwd(int Cin, int Cout, int ks);
int pnsfm_conv2d_pack_weights(const float* w, float* wp_fwd /*nullable*/, float* wp_bwd /*nullable*/,
                              int Cin, int Cout, int ks, void* stream);
/* Many weights in ONE launch (a model's conv layers right after the optimizer step; replaces ~100 per-layer launches of ~10 us
 * each).  The caller keeps a table of pnsfm_conv2d_pack_item_bytes()-sized items in DEVICE memory: each item is written on the
 * host by pnsfm_conv2d_pack_item_fill (w: [Cout][Cin][k][k]; wp_fwd / wp_bwd sized by pnsfm_conv2d_packed_elems_*; first_block =
 * sum of the block counts of the items before it), which returns the item's block count -- 0 if the shape does not take the
 * split-bf16 layout in both directions (leave it to pnsfm_conv2d_pack_weights; nothing is written), < 0 on error -- and then
 * copied to the device by the caller.  pnsfm_conv2d_pack_table packs all of them; pointers must stay valid while the table is
 * in use.  Mirrors nothing in the reference (cuDNN keeps its own filter layouts); replaces packnet_sfm/hip/functional.py's
 * lazy per-layer repack for optimizers that update parameters in place. */
size_t pnsfm_conv2d_pack_item_bytes(void);
int pnsfm_conv2d_pack_item_fill(void* item_host, const float* w, float* wp_fwd, float* wp_bwd, int Cin, int Cout, int ks,
                                int first_block);
int pnsfm_conv2d_pack_table(const void* table_dev, int n_items, int total_blocks, void* stream);
int pnsfm_conv2d_forward(const float* x, const float* wp_fwd, const float* bias /*nullable*/, float* y,
                         int B, int Cin, int Cout, int H, int W, int ks, void* stream);
int pnsfm_conv2d_backward_data(const float* dy, const float* wp_bwd, float* dx,
                               int B, int Cin, int Cout, int H, int W, int ks, void* stream);
/* Round 5: dx = backward-data + addend.  A tensor with two consumers (the encoder feature that is also a decoder skip input,
 * PackNet01.py:119-174; the ResidualConv input read by conv1 and the 1x1 shortcut, layers01.py:66-72) receives two gradients that
 * autograd sums with an elementwise pass; here the second consumer's backward-data launch adds the first gradient in its epilogue
 * (or in the second stage of a K-split launch).  addend: [B][Cin][H][W] with addend_bstride floats between samples -- a channel slice
 * of the wider gradient of a multi-source convolution is passed as it is; must not overlap dx.  addend == NULL: plain backward-data. */
int pnsfm_conv2d_backward_data_add(const float* dy, const float* wp_bwd, float* dx, const float* addend, long long addend_bstride,
                                   int B, int Cin, int Cout, int H, int W, int ks, void* stream);
/* dw in the reference layout [Cout][Cin][k][k]; dbias [Cout] (nullable). Both are overwritten.  When the autotuner
 * splits the pixel reduction the outputs are zero-filled first; placing dbias directly behind dw (dbias == dw +
 * Cout*Cin*k*k) lets one fill cover both. */
int pnsfm_conv2d_backward_weight(const float* x, const float* dy, float* dw, float* dbias /*nullable*/,
                                 int B, int Cin, int Cout, int H, int W, int ks, void* stream);

/* The decoder's skip connections: conv(torch.cat((x0, x1[, x2]), 1)) WITHOUT the concatenated tensor (networks/depth/PackNet01.py:138-174:
 * iconv5..iconv1 read cat(unpacked, skip[, upsampled inverse depth])).  The K loop of the split-bf16 kernels walks the three tensors in
 * turn; x0 and x1 must end on 16-channel (forward) / 32-channel (weight gradient) boundaries, x2 (C2 >= 0 channels) may be ragged.
 * wp_fwd is the packed weight of the [Cout][C0+C1+C2][k][k] parameter.  Backward-data is pnsfm_conv2d_backward_data on the whole
 * parameter (its output IS the concatenated gradient; consumers take channel slices).  Non-zero return (message in
 * pnsfm_last_error) when the shape is outside the split kernels' envelope: the caller then concatenates. */
int pnsfm_conv2d_forward_cat(const float* x0, int C0, const float* x1, int C1, const float* x2 /*nullable*/, int C2,
                             const float* wp_fwd, const float* bias /*nullable*/, float* y, int B, int Cout, int H, int W, int ks,
                             void* stream);
/* 1 if pnsfm_conv2d_backward_weight_cat takes sources of C0 / C1 / C2 channels for this layer shape, else 0 (pure query) */
int pnsfm_conv2d_cat_wgrad_supported(int C0, int C1, int C2, int Cout, int B, int H, int W, int ks);
int pnsfm_conv2d_backward_weight_cat(const float* x0, int C0, const float* x1, int C1, const float* x2 /*nullable*/, int C2,
                                     const float* dy, float* dw, float* dbias /*nullable*/, int B, int Cout, int H, int W, int ks,
                                     void* stream);

/* Strided variants (stride 1 or 2, zero pad k/2): PoseNet's stride-2 conv_gn blocks, networks/pose/PoseNet.py:28-34.
 * x:[B,Cin,Hin,Win] -> y / dy:[B,Cout,Ho,Wo] with Ho = (Hin + 2*(k/2) - k)/stride + 1.  Backward-data of a stride-2 conv is
 * pnsfm_conv2d_backward_data applied to dy zero-upsampled onto the input grid (dy[y][x] at (2y, 2x)). */
int pnsfm_conv2d_forward_strided(const float* x, const float* wp_fwd, const float* bias /*nullable*/, float* y,
                                 int B, int Cin, int Cout, int Hin, int Win, int ks, int stride, void* stream);
int pnsfm_conv2d_backward_weight_strided(const float* x, const float* dy, float* dw, float* dbias /*nullable*/,
                                         int B, int Cin, int Cout, int Hin, int Win, int ks, int stride, void* stream);

/* Runtime autotuning of the conv kernels' tiling / split factors (default on; env PNSFM_AUTOTUNE=0 turns it off).
 * The first call for a new shape times the candidate configurations on the caller's stream (it synchronises), like
 * `torch.backends.cudnn.benchmark = True` in the reference (trainers/horovod_trainer.py:19). */
int pnsfm_set_autotune(int on);
/* Number of autotune decisions read from the database shipped next to the library (csrc/tuned_gfx950.db: the autotuner's own
 * output for the benchmark configurations on an MI355X; 0 when PNSFM_TUNE_DB names a user database, when the file is
 * missing or when autotuning is off).  Shapes the database does not list are timed on first use as usual. */
int pnsfm_tune_shipped_entries(void);
/* Un-tuned default of the forward/backward-data kernel: 0 = halo patch staged through registers, 1 = patch double-buffered
 * by LDS-DMA (global_load_lds) issued in slices between the taps, 2 = the fully pipelined kernel (patch AND per-kernel-row
 * weight slabs double-buffered by LDS-DMA, one barrier per kernel row, up to 160 KB of LDS).  The autotuner times all
 * three; this switch exists for tests.  Drops pnsfm_tune_set pins; database / autotuner decisions are kept (and ignored while
 * autotuning is off). */
int pnsfm_set_conv_variant(int lds_dma);
/* Arithmetic of the forward / backward-data convolution kernels (stride 1 and 2, K-channels >= 16, k >= 3; other shapes
 * always use the f32 instruction):
 *   1 (default) fp32 rebuilt on the bf16 matrix pipe: each fp32 operand is split EXACTLY into three bf16 pieces
 *     (8+8+8 significand bits) and the product is summed from 6 of the 9 piece products with fp32 accumulation
 *     (v_mfma_f32_32x32x16_bf16; dropped terms <= 3 * 2^-24 |a||b|) -- measured error against fp64 is BELOW that of the f32
 *     MFMA chain (tools/micro/bf16x3_check.hip; tests/test_gpu_parity.py::test_conv2d_bx3_error_vs_fp64);
 *   0 v_mfma_f32_32x32x2_f32 everywhere (env PNSFM_CONV_MATH=f32).
 * Variants 3..6 of pnsfm_set_conv_variant select the un-tuned LDS plan of the split kernels (3: one patch buffer, 4: two,
 * 5: two + a whole kernel row of weights per stage, 6: <= 53 KB of LDS and <= 168 registers so that THREE workgroups share a CU).  The packed-weight layout follows from (mode, shape):
 * pnsfm_conv2d_packed_elems_* already return the larger of the two sizes, but weights packed under one mode must be packed
 * again after a switch.  Returns the previous mode.  Under mode 1 the weight gradient runs the same split arithmetic
 * (conv2d_wgrad3.hip: stride 1, k in {3,5,7}, W % 4 == 0, >= 16 channels; tests/test_gpu_round3.py holds its error against fp64 to
 * the f32 kernels' class at the step's real reduction lengths); other shapes and mode 0 use the f32-MFMA weight-gradient kernels. */
int pnsfm_set_conv_math(int mode);
int pnsfm_get_conv_math(void);
/* Un-tuned choice of the weight-gradient kernel: 0 = generic ((ci, tap) columns, offset table; conv2d.hip), 1 = tap-major
 * f32 MFMA (dY fragments kept in registers across the taps, LDS-DMA double buffering; conv2d_wgrad2.hip; stride 1,
 * k in {1,3,5}, W % 8 == 0, >= 16 channels), 2 = split-bf16 arithmetic (conv2d_wgrad3.hip: one kernel row per workgroup, dY
 * straight from global memory, shifted X operands built in registers; stride 1, W % 4 == 0, >= 16 channels; only under
 * pnsfm_set_conv_math(1)), -1 = library default (2 where it applies, else 0).  The autotuner times all that apply, and for 3x3
 * layers also kernel 3 (conv2d_wgrad4.hip: all nine taps per workgroup on v_mfma_f32_16x16x32_bf16, tiles 3 / 4 / 5 pixel groups
 * wide so that W = 20 / 40 / 80 fit; same arithmetic, same two-stage reduction), which is reachable through the tuning database /
 * pnsfm_tune_set only.  For tests. */
int pnsfm_set_wgrad_variant(int tap_major);
/* Programmatic entry of the tuning database (what a PNSFM_TUNE_DB line does): key7 = {kind, B, Cin, Cout, H, W, ks} with
 * kind = 0 forward / 1 backward-data / 2 backward-weight, + 10 * stride; for kind 2 the key holds H*W in place of H and the
 * tiling width in place of W (32 for a 1x1 convolution).  forward / backward-data (+ 100 on `kind` for the split-bf16 arithmetic): v0 = NT | variant << 4 | narrow-M << 8,
 * v1 = K-split; backward-weight: v0 = pixel split, v1 = kernel (0 generic, 1 tap-major, 2 | NT << 4 | WM << 6 split-bf16 one
 * kernel row per workgroup, 3 | (WCI | TG << 4 | TR << 8) << 4 split-bf16 nine taps per workgroup).  Forward / backward-data variants:
 * 0..2 f32 stagings, 3..6 split-bf16 LDS plans, 7 ping-pong workgroup, 8 the 1x1 kernel without LDS.  Used by the determinism sweep
 * (tools/conv_config_sweep.py), which checks EVERY configuration the autotuner may pick against the oracle's convolution. */
int pnsfm_tune_set(const int* key7, int v0, int v1);
/* What the calling thread's most recent forward / backward-data launch actually ran: out8 = {variant (0..2 f32 stagings, 3..6
 * split-bf16 LDS plans, 7 ping-pong workgroup, 8 the 1x1 kernel without LDS), pixel tiles per wave NT, M tiles per wave MT, taps per weight stage G, K-split,
 * tile mode (0 classic, 1 16-wide rectangles, 2 row bands), workgroups, LDS bytes}.  A pinned configuration that does not fit
 * a shape silently falls back to the heuristic one; tests that pin a variant assert on this.  A backward-weight call that ran the
 * stem's kernel (3 input channels, 5x5) leaves {105, 0, MT, 0, pixel splits, 0, workgroups, 0}.  Returns 1 when no launch happened yet. */
int pnsfm_conv2d_last_config(int* out8);
/* Tuning database: environment PNSFM_TUNE_DB=<file> loads earlier autotune decisions when the library first tunes and
 * appends new ones (text, one line per layer shape) -- what MIOpen's user find-db does for the reference's cuDNN/MIOpen
 * convolutions.  A process started with a complete database launches no candidate kernels. */

/* ---- device-side input pipeline (uint8 frames; bit-exact to Pillow's libImaging arithmetic) --------------------------------
 * replaces, on the training input path, datasets/transforms.py:11-41 (train_transforms) = datasets/augmentations.py:101-180
 * resize (transforms.Resize, Lanczos), :228-337 duplicate + colour jitter (torchvision adjust_* on PIL images), :185-226 ToTensor.
 * pnsfm_resample8: ONE axis of PIL's separable 8-bit resample (axis 1: width, axis 0: height) over N images in NHWC layout;
 *   kk [out][ksize] int32 fixed-point coefficients (22 fractional bits) and bounds [out][2] = {first input index, count}, computed
 *   by the host exactly as Resample.c precompute_coeffs / normalize_coeffs_8bpc (packnet_sfm/datasets/device_transforms.py).
 * pnsfm_jitter_totensor: per image the <= 4 colour operations in the drawn order, then ToTensor; img NHWC uint8 [N][H][W][3],
 *   ops = N records {int op[4] (0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none); float factor[4]; int hue_add
 *   (= uint8(hue_factor*255)); int enabled; float color[3]; int has_color} (56 bytes; color = the diagonal of the 3x4 matrix of
 *   jittering[4], augmentations.py:266-277, applied last like Image.convert('RGB', matrix)), lsum_ws: N x uint64 scratch;
 *   out / out_orig (nullable): NCHW float32 [N][3][H][W]. */
int pnsfm_resample8(const uint8_t* in, uint8_t* out, const int* kk, const int* bounds, int ksize, int N, int inH, int inW,
                    int outH, int outW, int C, int axis, void* stream);
int pnsfm_jitter_totensor(const uint8_t* img, const void* ops, unsigned long long* lsum_ws, float* out, float* out_orig /*nullable*/,
                          int N, int H, int W, void* stream);

/* ---- Neural-Ray-Surface projection (softmax expectation over a 41x41 candidate patch) --------------------------------------
 * replaces the core of GenericCamera.project, packnet_sfm/geometry/camera_generic.py:127-192 (patch coordinates with the window
 * translated into the image, ray-surface gather, logits d.r / T, softmax, expectation of the candidate coordinates).
 * dir, ray: [3][h][w] (unit directions of the points to project, ray surface, both at the projection resolution, h, w >= 41);
 * coords: [h][w][2] = (expected row, expected column) in pixels; stat: [h][w][2] = (max logit, sum of exponentials) saved for
 * backward.  Backward: gcoords [h][w][2] -> gdir [3][h][w] and gray [3][h][w] (either may be null); gather formulation, no atomics. */
int pnsfm_nrs_project_forward(const float* dir, const float* ray, float* coords, float* stat, int h, int w, float temperature,
                              void* stream);
int pnsfm_nrs_project_backward(const float* dir, const float* ray, const float* coords, const float* stat, const float* gcoords,
                               float* gdir /*nullable*/, float* gray /*nullable*/, int h, int w, float temperature, void* stream);

/* ---- GroupNorm(G) + activation, optional residual add in front -----------------------------
 * replaces torch.nn.GroupNorm(16, C) + nn.ELU(inplace=True): layers01.py:31-32,36-37 and the
 * residual form `activ(normalize(x_out + shortcut))`: layers01.py:61-62,72.
 * act: 0 = identity, 1 = ELU(alpha=1), 2 = ReLU (PoseNet, networks/pose/PoseNet.py:28-34).
 * stats_ws / red_ws: double[pnsfm_groupnorm_ws_doubles(B, C, G)] scratch (per-workgroup partial sums: no zero-fill needed,
 * deterministic); mean/rstd: float[B*G] saved for backward. */
#define PNSFM_GN_MAX_SPLIT 64
size_t pnsfm_groupnorm_ws_doubles(int B, int C, int G);
int pnsfm_groupnorm_act_forward(const float* x, const float* res /*nullable*/, const float* gamma,
                                const float* beta, float* y, float* mean, float* rstd, double* stats_ws,
                                int B, int C, int HW, int G, float eps, int act, void* stream);
/* red_ws: double[pnsfm_groupnorm_ws_doubles(B, C, G)] scratch. dx is the gradient w.r.t. x (and, identically, w.r.t. res). */
int pnsfm_groupnorm_act_backward(const float* dy, const float* x, const float* res /*nullable*/,
                                 const float* gamma, const float* beta, const float* mean, const float* rstd,
                                 float* dx, float* dgamma, float* dbeta, double* red_ws,
                                 int B, int C, int HW, int G, int act, void* stream);
/* Round 6: when a (sample, group) slab -- (C / G) * HW contiguous floats -- is at most 64 K floats and HW % 4 == 0, the two entry
 * points above run ONE launch each (forward: slab in registers, statistics + normalise + activation; backward: slab blocks write dx,
 * channel blocks write dgamma / dbeta; csrc/groupnorm.hip) instead of two; stats_ws / red_ws are then not touched.
 * pnsfm_set_gn_fused(0) (or PNSFM_GN_FUSED=0 in the environment) keeps the two-launch form everywhere; returns the previous setting.
 * stats_ws / red_ws may be null: the two-launch form then keeps its partial sums in the stream's scratch buffer. */
int pnsfm_set_gn_fused(int on);

/* ---- packing / unpacking data movement ------------------------------------------------------
 * space_to_depth == `packing(x, r=2)` layers01.py:126-148 (== F.pixel_unshuffle):
 *   y[b][4c+2i+j][h][w] = x[b][c][2h+i][2w+j];  x:[B,C,H,W] -> y:[B,4C,H/2,W/2]
 * depth_to_space == nn.PixelShuffle(2) layers01.py:275,285:  x:[B,4C,H,W] -> y:[B,C,2H,2W]
 * Each is the other's backward. */
int pnsfm_space_to_depth(const float* x, float* y, int B, int C, int H, int W, void* stream);
/* same, for an x that is a channel slice of a wider tensor (image b starts at x + b*x_batch_stride floats): the
 * gradient that reaches PixelShuffle's backward through torch.cat((unpack, skip), 1) (PackNet01.py:140-172) */
int pnsfm_space_to_depth_strided(const float* x, float* y, int B, int C, int H, int W, size_t x_batch_stride, void* stream);
int pnsfm_depth_to_space(const float* x, float* y, int B, int C, int H, int W, void* stream);

/* ---- Conv3d(1 -> 8, 3x3x3, padding 1) over the (channel, y, x) volume ----------------------
 * replaces self.conv3d in PackLayerConv3d / UnpackLayerConv3d: layers01.py:236-237,241-245,
 * :276-277,280-284.  p:[B,D,H,W] (the unsqueezed single 3-D feature), out:[B,8*D,H,W] with
 * channel f*D+d (the `.view(b, c*d, h, w)` at :244-245).  w3:[8][27] (= [8,1,3,3,3]), b3:[8]. */
int pnsfm_conv3d_1to8_forward(const float* p, const float* w3, const float* b3, float* out,
                              int B, int D, int H, int W, void* stream);
int pnsfm_conv3d_1to8_backward_data(const float* dout, const float* w3, float* dp,
                                    int B, int D, int H, int W, void* stream);
/* dw3:[8*27], db3:[8]; overwritten. ws: double[8*28] scratch. */
int pnsfm_conv3d_1to8_backward_weight(const float* p, const float* dout, float* dw3, float* db3, double* ws,
                                      int B, int D, int H, int W, void* stream);
/* The same three with NF = 4 or 8 feature maps (`d=num_3d_feat` of PackLayerConv3d / UnpackLayerConv3d,
 * layers01.py:213-232,250-268: 8 in PackNet01, 4 in PackNetSlim01.py:39 and PackNetSAN01.py).  out / dout:[B,NF*D,H,W],
 * w3:[NF][27], b3:[NF] (forward: NULL = no bias); ws stays double[8*28]. */
int pnsfm_conv3d_forward(const float* p, const float* w3, const float* b3, float* out,
                         int B, int D, int H, int W, int NF, void* stream);
int pnsfm_conv3d_backward_data(const float* dout, const float* w3, float* dp,
                               int B, int D, int H, int W, int NF, void* stream);
int pnsfm_conv3d_backward_weight(const float* p, const float* dout, float* dw3, float* db3, double* ws,
                                 int B, int D, int H, int W, int NF, void* stream);

/* ---- InvDepth activation: y = sigmoid(x) / min_depth  (layers01.py:119-122) --------------- */
int pnsfm_invdepth_act_forward(const float* x, float* y, size_t n, float min_depth, void* stream);
int pnsfm_invdepth_act_backward(const float* dy, const float* y, float* dx, size_t n, float min_depth,
                                void* stream);

/* ---- InvDepth head, fused: y = sigmoid(conv3x3(zero_pad1(x)) + b) / min_depth, ONE output channel -----------------
 * replaces InvDepth.forward (layers01.py:98-122: ConstantPad2d(1) -> Conv2d(C, 1, 3) -> Sigmoid -> / min_depth) with a
 * streaming channel reduction (no matrix cores for a 1-row GEMM).  x:[B,C,H,W], w:[1][C][3][3], bias:[1], y:[B,1,H,W].
 * backward: dz = dy * y * (1 - y*min_depth) from pnsfm_invdepth_act_backward; one kernel then produces dx:[B,C,H,W],
 * dw:[C*9] and db:[1] (both overwritten; placing db right behind dw saves a fill). */
int pnsfm_invdepth_conv_forward(const float* x, const float* w, const float* bias, float* y, int B, int C, int H, int W,
                                float min_depth, void* stream);
int pnsfm_invdepth_conv_backward(const float* x, const float* w, const float* dz, float* dx, float* dw, float* db, int B,
                                 int C, int H, int W, void* stream);

/* ---- pose vector -> rigid transform: vec:[N,6] = (tx,ty,tz,rx,ry,rz) -> mat:[N,4,4], R = Rx*Ry*Rz, bottom row 0 0 0 1
 * replaces Pose.from_vec(vec, 'euler') (geometry/pose.py:40-46) = pose_vec2mat + euler2mat (geometry/pose_utils.py:8-52).
 * backward: dmat:[N,4,4] -> dvec:[N,6]. */
int pnsfm_pose_vec2mat_forward(const float* vec, float* mat, int N, void* stream);
int pnsfm_pose_vec2mat_backward(const float* vec, const float* dmat, float* dvec, int N, void* stream);

/* ---- supervised inverse-depth loss of the semi-supervised models, ONE scale ---------------------------------------
 * replaces SupervisedLoss.calculate_loss for one scale (losses/supervised_loss.py:138-149) with the loss functions of
 * get_loss_func (:70-84): method 0 = l1 (nn.L1Loss), 1 = mse, 2 = abs_rel (mean |x-y| / x), 3 = berhu (BerHuLoss :11-53,
 * threshold 0.2), 4 = silog (SilogLoss :55-67, ratio 10, ratio2 0.85); sparse != 0 keeps only pixels with gt > 0 (the
 * 'sparse-*' methods).  pred, gt: n floats (same resolution); loss: 1 float; ws: double[8 + 1024] scratch that the
 * backward call reads again.  backward: grad_out: 1 float (dL/dloss) -> dpred: n floats (zero at masked pixels). */
int pnsfm_supervised_loss_forward(const float* pred, const float* gt, float* loss, double* ws, size_t n, int method,
                                  int sparse, void* stream);
int pnsfm_supervised_loss_backward(const float* pred, const float* gt, const double* ws, const float* grad_out,
                                   float* dpred, size_t n, int method, int sparse, void* stream);

/* ---- view synthesis: inv2depth -> Camera.reconstruct -> Camera.project -> grid_sample ------
 * replaces MultiViewPhotometricLoss.warp_ref_image for ONE scale and J context images:
 *   losses/multiview_photometric_loss.py:127-165, utils/depth.py:103-120 (inv2depth),
 *   geometry/camera.py:112-148 (reconstruct), :150-191 (project),
 *   geometry/camera_utils.py:27-59 (view_synthesis: bilinear, zeros, align_corners=True).
 * inv_depth:[B,1,H,W]; ref:[J,B,3,H,W]; K, refK:[B,3,3] already scaled to this resolution;
 * T:[J,B,4,4] target->context rigid transforms (Pose.mat); warped:[J,B,3,H,W]. */
int pnsfm_view_synthesis_forward(const float* inv_depth, const float* ref, const float* K, const float* refK,
                                 const float* T, float* warped, int J, int B, int H, int W, void* stream);
/* d_inv_depth:[B,1,H,W] (sum over J, overwritten); dT:[J,B,4,4] (rows 0..2 filled, row 3 zero).
 * ws: double[J*B*12] scratch (fp64 accumulation of the pose gradient). */
int pnsfm_view_synthesis_backward(const float* d_warped, const float* inv_depth, const float* ref,
                                  const float* K, const float* refK, const float* T,
                                  float* d_inv_depth, float* dT, double* ws, int J, int B, int H, int W, void* stream);
/* The same with grid_sample's other padding modes (camera_utils.py:58-59, `padding_mode` of the loss config):
 * padding_mode 0 = 'zeros' (the two entry points above), 1 = 'border', 2 = 'reflection' (align_corners=True). */
int pnsfm_view_synthesis_forward_pad(const float* inv_depth, const float* ref, const float* K, const float* refK,
                                     const float* T, float* warped, int J, int B, int H, int W, int padding_mode,
                                     void* stream);
int pnsfm_view_synthesis_backward_pad(const float* d_warped, const float* inv_depth, const float* ref,
                                      const float* K, const float* refK, const float* T, float* d_inv_depth, float* dT,
                                      double* ws, int J, int B, int H, int W, int padding_mode, void* stream);

/* ---- photometric loss of one scale: SSIM + L1, automask, min/mean reduce -------------------
 * replaces SSIM() :14-53, MultiViewPhotometricLoss.SSIM :169-186, calc_photometric_loss :188-223
 * and the per-scale body of reduce_photometric_loss :225-253 (clip_loss > 0: the _clip variants below).
 * Candidates per pixel, in the reference's order (:321-334): warped[0], ref[0], warped[1], ref[1], ...
 * (ref[j] entries only when automask != 0).  reduce_op: 0 = 'min', 1 = 'mean'.
 * loss_sum: double[1], receives sum over pixels of the reduced per-pixel loss (caller divides by B*H*W);
 * argmin: uint8[B*H*W] (candidate index chosen per pixel; written for reduce_op==0). */
int pnsfm_photometric_forward(const float* warped, const float* ref, const float* target,
                              double* loss_sum, uint8_t* argmin, int J, int B, int H, int W,
                              float ssim_weight, float C1, float C2, int automask, int reduce_op, void* stream);
/* d_warped:[J,B,3,H,W] = grad_scale * d(loss_sum)/d(warped), overwritten. */
/* Variants that keep scalars on the device (round 4: no ATen launch between these kernels and autograd): forward_mean writes
 * loss_mean float[1] = loss_sum / (B*H*W); backward_dev multiplies grad_scale by upstream[0] (device scalar, nullable) and takes
 * clip = 0 | 1 (the byte layout of pnsfm_photometric_forward / _forward_clip). */
int pnsfm_photometric_forward_mean(const float* warped, const float* ref, const float* target, float* loss_mean, uint8_t* argmin,
                                   int J, int B, int H, int W, float ssim_weight, float C1, float C2, int automask, int reduce_op,
                                   void* stream);
int pnsfm_photometric_backward_dev(const float* warped, const float* target, const uint8_t* argmin, float* d_warped,
                                   float grad_scale, const float* upstream /*nullable*/, int J, int B, int H, int W,
                                   float ssim_weight, float C1, float C2, int automask, int reduce_op, int clip, void* stream);
int pnsfm_photometric_backward(const float* warped, const float* target, const uint8_t* argmin,
                               float* d_warped, float grad_scale, int J, int B, int H, int W,
                               float ssim_weight, float C1, float C2, int automask, int reduce_op, void* stream);
/* clip_loss > 0 (:214-219): every candidate map is clamped at mean + clip_loss * std of itself (torch.std: unbiased; the
 * threshold is a float in the reference, so no gradient flows through it).  Three launches: statistics pass, threshold
 * kernel, clamped pass.  stats_ws: double[12], thr_ws: float[6] scratch.  The per-pixel byte then also records clamping
 * (min: argmin | clamped << 7; mean: bit mask of clamped candidates) and MUST go to pnsfm_photometric_backward_clip. */
int pnsfm_photometric_forward_clip(const float* warped, const float* ref, const float* target,
                                   double* loss_sum, uint8_t* argmin, int J, int B, int H, int W,
                                   float ssim_weight, float C1, float C2, int automask, int reduce_op,
                                   float clip_loss, double* stats_ws, float* thr_ws, void* stream);
int pnsfm_photometric_backward_clip(const float* warped, const float* target, const uint8_t* argmin,
                                    float* d_warped, float grad_scale, int J, int B, int H, int W,
                                    float ssim_weight, float C1, float C2, int automask, int reduce_op, void* stream);

/* ---- edge-aware smoothness of one scale ------------------------------------------------------
 * replaces calc_smoothness utils/depth.py:165-198 (after inv_depths_normalize :146-162) with
 * gradient_x/y utils/image.py:85-113, and the |.|.mean() of calc_smoothness_loss
 * multiview_photometric_loss.py:276-278.  inv_norm:[B,1,H,W] mean-normalised inverse depth,
 * image:[B,3,H,W].  sums: double[2] = { sum |Sx|, sum |Sy| }. */
int pnsfm_smoothness_forward(const float* inv_norm, const float* image, double* sums,
                             int B, int H, int W, void* stream);
/* L1-only photometric loss (ssim_loss_weight == 0) with the 'min' reduce op and / or clip_loss > 0: the reference then reduces and
 * clips per-CHANNEL candidate maps (multiview_photometric_loss.py:205-219, 238-246).  loss_mean: float[1]; rec: int32[B,H,W] for
 * backward ('min': (candidate*3 + channel) | clamped << 7; 'mean': bit mask of clamped (candidate, channel) pairs).
 * (ssim_loss_weight == 0 with 'mean' and no clipping coincides with the SSIM kernels' channel mean: pnsfm_photometric_forward.) */
int pnsfm_photometric_l1_forward(const float* warped, const float* ref, const float* target, float* loss_mean, int* rec,
                                 int J, int B, int H, int W, int automask, int reduce_op, float clip_loss, void* stream);
int pnsfm_photometric_l1_backward(const float* warped, const float* target, const int* rec, float* d_warped, float grad_scale,
                                  const float* upstream /*nullable*/, int J, int B, int H, int W, int automask, int reduce_op,
                                  void* stream);
/* The same with the mean normalisation of the inverse depth fused (multiview_photometric_loss.py:269-271: inv /
 * inv.mean(2, True).mean(3, True).clamp(min=1e-6)): inv_depth:[B,1,H,W] RAW inverse depth.  loss: float[1] = mean|Sx| + mean|Sy|,
 * mean: float[B] = the clamped per-sample means (kept for backward).  Backward overwrites d_inv_depth with upstream[0] (device
 * scalar, nullable = 1) * dloss/d(inv_depth), the normalisation's own gradient path included. */
int pnsfm_smoothness_norm_forward(const float* inv_depth, const float* image, float* loss, float* mean,
                                  int B, int H, int W, void* stream);
int pnsfm_smoothness_norm_backward(const float* inv_depth, const float* image, const float* mean, const float* upstream /*nullable*/,
                                   float* d_inv_depth, int B, int H, int W, void* stream);
/* d_inv_norm = gx * d(sum|Sx|)/d(inv_norm) + gy * d(sum|Sy|)/d(inv_norm), overwritten. */
int pnsfm_smoothness_backward(const float* inv_norm, const float* image, float* d_inv_norm,
                              float gx, float gy, int B, int H, int W, void* stream);

/* ---- batched strided-window operations (one launch for a list of copies / accumulations / zero-fills) -------------------------
 * The collapsed form of PackLayerConv3d (layers01.py:213-247) computes the r-pixel border frame on thin strips: gathering
 * the strips, dropping rows of the Conv3d output, pasting the results into the interior result and the mirror-image gradient
 * scatters are windows of NCHW tensors.  ops_host: HOST array of n_ops (<= PNSFM_MAX_REGION_OPS) descriptors, copied into the kernel
 * arguments; pointers are device pointers, strides in ELEMENTS.  op: 0 dst = src, 1 dst += src, 2 dst = 0 (src ignored).  Operations
 * of one launch must not overlap each other's destinations. */
#define PNSFM_MAX_REGION_OPS 12
typedef struct {
  const float* src;
  float* dst;
  int n[4];
  long long src_stride[4];
  long long dst_stride[4];
  int op;
} pnsfm_region_op;
int pnsfm_region_ops(const void* ops_host, int n_ops, void* stream);

/* ---- nearest-neighbour up-sampling by an integer factor -------------------------------------
 * replaces F.interpolate(mode='nearest') where the reference brings every predicted scale to full resolution
 * (models/model_utils.py:163-180 -> utils/image.py:148-176) and nn.Upsample(scale_factor=2, mode='nearest')
 * (networks/depth/PackNet01.py:87-89,150,159,168).  x: [N, h, w] planes -> y: [N, h*s, w*s], y[n,oy,ox] = x[n,oy/s,ox/s];
 * backward: dx = s x s block sums of dy (rows, then columns, ascending).  w*s must be a multiple of 4. */
int pnsfm_upsample_nearest_forward(const float* x, float* y, int N, int h, int w, int s, void* stream);
int pnsfm_upsample_nearest_backward(const float* dy, float* dx, int N, int h, int w, int s, void* stream);

/* ---- scalar tail of the multi-view photometric loss -----------------------------------------
 * replaces the Python-level sums of losses/multiview_photometric_loss.py:248-252 (mean over scales of the reduced photometric
 * terms), :275-280 (smoothness terms / 2^i, mean, weight) and :337-338 (their sum), in the reference's operation order.
 * photometric / smoothness: HOST arrays of n / ns DEVICE scalar pointers.  out3 = {loss, weighted smoothness, photometric}.
 * backward: g = device scalar d(loss); dout16[i] = d/dP[i] (i < n), dout16[8 + i] = d/dS[i] (i < ns). */
int pnsfm_loss_combine_forward(const float* const* photometric, int n, const float* const* smoothness, int ns, float weight,
                               float* out3, void* stream);
int pnsfm_loss_combine_backward(const float* g, int n, int ns, float weight, float* dout16, void* stream);

/* ---- bias of the composed packing convolution -------------------------------------------------
 * PackLayerConv3d (networks/layers/packnet/layers01.py:243-246) runs Conv3d(1->d) then Conv2d with nothing in between; where this
 * package composes the two the bias of the composed convolution is
 *   bias_eff[co] = b2[co] + sum_f b3[f] * Ssum[co][f],   Ssum[co][f] = sum of block f (blk = D*k*k floats) of row co of W2.
 * forward writes Ssum [C,d] and bias_eff [C].  backward: db3[f] = sum_co g[co]*Ssum[co][f] (db3 may be NULL) and
 * dW2[C, d*D, k, k] = dWeff_full[C, d*D, k+2, k+2] cropped by one tap on every side + g[co]*b3[f] (dW2 may be NULL). */
int pnsfm_pack_bias_eff_forward(const float* W2, const float* b2, const float* b3, float* Ssum, float* bias_eff, int C, int d,
                                int blk, void* stream);
int pnsfm_pack_bias_eff_backward(const float* g, const float* Ssum, const float* b3, const float* dWeff_full, float* db3, float* dW2,
                                 int C, int d, int D, int k, void* stream);

/* ---- Adam over a flat fp32 parameter buffer (torch.optim.Adam semantics, no amsgrad) -------
 * replaces the optimizer.step() of models/model_wrapper.py:128-149 / trainers/horovod_trainer.py:93
 * for one parameter group.  grad_scale multiplies the gradient first (1/world_size after a
 * sum-all-reduce).  step is the 1-based step count used for bias correction. */
int pnsfm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                    int step, void* stream);
/* The same update with the optimizer state on the DEVICE (replayable inside a hipGraph, one launch per parameter group of any
 * size, float4 accesses): hp = float[12] {step, lr, beta1, beta2, eps, weight_decay, grad_scale, 1-beta1, 1-beta2, ...}; the call increments
 * hp[0] and then applies step hp[0].  All four buffers must be 16-byte aligned. */
int pnsfm_adam_flat_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float* hp, void* stream);
/* The same update on a SLICE of the arenas (16-byte aligned start, n floats): FlatAdam (packnet_sfm/rccl/flat_adam.py) updates a
 * parameter group bucket by bucket while backward is still producing the gradients of the buckets behind it.  tick != 0: advance
 * hp[0] (the group's step counter) first -- exactly one slice per group and step does that. */
int pnsfm_adam_flat_update(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float* hp, int tick,
                           void* stream);
/* Round 5: the optimizer tail in two launches for the whole model.  (1) pnsfm_adam_pack_table: the Adam update of every conv weight
 * that takes the split-bf16 layout in both directions AND the re-pack of its forward / backward-data images in one pass (a workgroup
 * per 32 x 32 (co, ci) super-tile: p, g, m, v read once, p, m, v and both images written: 40 B per parameter instead of the 28 + 20
 * of pnsfm_adam_flat_step + pnsfm_conv2d_pack_table).  Items are filled on the host (pnsfm_adam_pack_item_fill returns the item's
 * workgroup count, 0: not eligible) into a table of pnsfm_adam_pack_item_bytes()-sized records copied to the device.
 * (2) pnsfm_adam_segments: the same update over a list of arena segments -- everything the table does not cover.  hp[0] must have
 * been advanced for the step (pnsfm_adam_flat_update(..., n = 0, tick = 1)) before either launch. */
size_t pnsfm_adam_pack_item_bytes(void);
int pnsfm_adam_pack_item_fill(void* item_host, float* w, const float* g, float* m, float* v, const float* hp, float* wp_fwd,
                              float* wp_bwd, int Cin, int Cout, int ks, int first_block);
int pnsfm_adam_pack_table(const void* table_dev, int n_items, int total_blocks, void* stream);
size_t pnsfm_adam_seg_bytes(void);
int pnsfm_adam_seg_fill(void* seg_host, float* p, const float* g, float* m, float* v, const float* hp, size_t n, int first_block);
int pnsfm_adam_segments(const void* segs_dev, int nseg, int total_blocks, void* stream);

/* ---- live timing of the dominant kernels (used by bench.py's roofline block) ---------------
 * When enabled, every launch of kind k is bracketed by hipEvents on its own stream.
 * kinds: 0 = conv2d forward/backward-data MFMA kernel, 1 = conv2d backward-weight MFMA kernel.
 * collect() synchronises the recorded events and returns totals since the last reset. */
/* Stream fork / join: nothing enqueued on `waiter` after this call runs before everything enqueued on `signaler` so far has finished
 * (hipEventRecord + hipStreamWaitEvent on a library-owned, timing-less event).  Host plumbing for the weight-gradient side stream of
 * packnet_sfm/hip/functional.py (replaces the all-reduce-overlap role of horovod's background thread, trainers/horovod_trainer.py). */
int pnsfm_stream_wait_stream(void* waiter, void* signaler);
/* Box calibration for bench.py's `calibration` block (csrc/calib.hip; nothing on the training step calls these): the bare six-product
 * bf16 MFMA stream of the split arithmetic -- blocks x 4 waves x iters x 24 v_mfma_f32_32x32x16_bf16, operands in registers, `sink`
 * receives blocks * 256 floats -- and a float4 streaming copy of n_floats (multiple of 4) floats. */
int pnsfm_calib_mfma(float* sink, int blocks, int iters, void* stream);
int pnsfm_calib_copy(const float* src, float* dst, size_t n_floats, void* stream);
int pnsfm_prof_enable(int on);
int pnsfm_prof_reset(void);
int pnsfm_prof_collect(int kind, double* total_ms, double* total_flops, long long* launches);
/* per-launch table (shape, grid, ms, TFLOP/s) of everything recorded since the last reset, as CSV */
int pnsfm_prof_dump(const char* path);

/* ---- sparse tensors on the pixel grid: the depth-completion branch of PackNet-SAN (csrc/sparse.hip) ----------------------
 * replaces the MinkowskiEngine calls of networks/layers/minkowski_encoder.py:10-131 and minkowski.py:33-83 (ME.MinkowskiConvolution
 * stride 1 / dimension 2 / no bias, ME.MinkowskiMaxPooling(3, 2), sparsify_depth, densify_features, map_add_features) with kernels whose
 * cost follows the number of ACTIVE sites.  A sparse tensor of a [B, h, w] grid is {sites int32 [cap] (linear cell index of row n,
 * ascending), imap int32 [B*h*w] (row of a cell or -1), count int32 [1] ON THE DEVICE, feats fp32 [cap][C]}; rows >= count are
 * unused and hold zeros.  Kernel offsets are numbered i = (dy + k/2) + k * (dx + k/2) (MinkowskiEngine's region iterator). */
size_t pnsfm_sparse_compact_ws_ints(int ncell);
/* coordinate map of the cells with src > 0 (src: a depth map or a 0/1 mask over ncell = B*h*w cells): imap, sites, count. */
int pnsfm_sparse_compact(const float* src, int ncell, int* imap, int* sites, int cap, int* count, int* ws, void* stream);
/* stride-2 coordinate rule: mask_out [B*(h/2)*(w/2)] = 1 where one of the 2x2 fine cells is active (feed it to _compact). */
int pnsfm_sparse_pool_cells(const int* imap, int B, int h, int w, float* mask_out, void* stream);
/* nbr [cap][ks*ks]: row of the neighbour of site n at offset i, or -1 (outside the grid / inactive / n >= count). */
int pnsfm_sparse_neighbors(const int* imap, const int* sites, const int* count, int cap, int h, int w, int ks, int* nbr, void* stream);
/* out[n][co] = sum_i sum_ci feats[nbr[n][flip ? KK-1-i : i]][ci] * kern[i][ci][co]; kern [ks*ks][Cin][Cout] (MinkowskiConvolution's
 * layout).  Backward-data = the same call with the kernel transposed to [ks*ks][Cout][Cin], Cin/Cout swapped and flip = 1.  Exact fp32
 * (v_mfma_f32_32x32x2_f32).  Every row < cap of `out` is written (zeros past count). */
int pnsfm_sparse_conv(const float* feats, const float* kern, const int* nbr, const int* count, float* out, int cap, int Cin, int Cout,
                      int ks, int flip, void* stream);
/* dkern [ks*ks][Cin][Cout] = sum_n feats[nbr[n][i]][ci] * dout[n][co] (overwritten). */
int pnsfm_sparse_conv_backward_weight(const float* feats, const float* dout, const int* nbr, const int* count, float* dkern, int cap,
                                      int Cin, int Cout, int ks, void* stream);
/* ME.MinkowskiMaxPooling(3, 2): fin rows of the fine [B, h, w] grid (imap_in) -> fout rows of the coarse sites (sites_out / count_out of
 * the compacted pool_cells mask); arg [cap][C] = fine row each maximum came from (-1: none) for the backward scatter. */
int pnsfm_sparse_maxpool_forward(const float* fin, const int* imap_in, const int* sites_out, const int* count_out, float* fout, int* arg,
                                 int cap, int C, int h, int w, void* stream);
int pnsfm_sparse_maxpool_backward(const float* dout, const int* arg, float* din, int cap_out, int cap_in, int C, void* stream);
/* rows -> dense [B][C][hw] (zeros at inactive cells; every element written) and dense -> rows (zeros past count). */
int pnsfm_sparse_densify(const float* feats, const int* imap, float* dense, int B, int C, int hw, void* stream);
int pnsfm_sparse_gather(const float* dense, const int* sites, const int* count, float* rows, int cap, int C, int hw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNSFM_H */
