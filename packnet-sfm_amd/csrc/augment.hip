// augment.hip -- device-side input pipeline: PIL-exact Lanczos resize, colour jitter and ToTensor on uint8 frames.
//
// Replaces, for the training input path, what the reference does per sample on the HOST with PIL / torchvision
//   (/root/reference/packnet_sfm/datasets/transforms.py:11-41 train_transforms ->
//    /root/reference/packnet_sfm/datasets/augmentations.py:101-180 resize_sample[_image_and_intrinsics] (transforms.Resize,
//    Image.ANTIALIAS = Lanczos), :228-252 duplicate_sample, :254-337 colorjitter_sample / random_color_jitter_transform
//    (torchvision.transforms.functional adjust_brightness / _contrast / _saturation / _hue on PIL images), :185-226
//    to_tensor_sample (transforms.ToTensor)).
// at >1000 img/s per node the CPU data loader (3 Lanczos resizes + 3 jitters per sample in PIL) becomes the
// bottleneck; here the decoded uint8 frames are uploaded once and everything after runs on the GPU.
//
// This is byte arithmetic and it is BIT-EXACT to Pillow (12.2, the version in this image; the arithmetic of libImaging's
// Resample.c / Blend.c / Convert.c, restated -- tests pin it against PIL itself, exhaustively for the HSV conversions):
//   * resize: separable convolution with PIL's fixed-point coefficients (22 fractional bits, computed on the host exactly
//     as precompute_coeffs / normalize_coeffs_8bpc do), horizontal pass then vertical pass, clip8 after each;
//   * brightness / contrast / saturation = ImageEnhance = Image.blend(degenerate, image, factor): out = in1 + alpha*(in2 - in1)
//     in C float, truncated for 0 <= alpha <= 1, clipped otherwise; degenerate = black / mean grey (int(mean(L) + 0.5)) / L;
//     L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
//   * hue: RGB -> HSV (colorsys in float with double literals, as Convert.c), H += uint8(hue_factor * 255) mod 256, HSV -> RGB;
//   * the four operations run in the per-sample order drawn by random.shuffle; ToTensor = float(v) / 255.
// HBM-bound streaming kernels: 3 B/pixel in, 2 x 12 B/pixel out (jittered + original float32 planes).
#include "pnsfm_common.h"
#include "../../include/pnsfm.h"

namespace pnsfm {

// one pass of PIL's ImagingResample{Horizontal,Vertical}_8bpc over NHWC uint8 images.
// axis 1: in [N][H][Win][C] -> out [N][H][Wout][C];  axis 0: in [N][Hin][W][C] -> out [N][Hout][W][C]
__global__ void __launch_bounds__(256) resample8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        const int* __restrict__ kk, const int* __restrict__ bounds, int ksize,
                                                        int N, int inH, int inW, int outH, int outW, int C, int axis) {
  const size_t total = (size_t)N * outH * outW * C;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c = (int)(e % C);
    size_t r = e / C;
    const int x = (int)(r % outW);
    r /= outW;
    const int y = (int)(r % outH);
    const int n = (int)(r / outH);
    const int o = axis == 1 ? x : y;
    const int xmin = bounds[2 * o], xmax = bounds[2 * o + 1];
    const int* k = kk + (size_t)o * ksize;
    int ss = 1 << 21;                                     // 1 << (PRECISION_BITS - 1)
    if (axis == 1) {
      const uint8_t* row = in + (((size_t)n * inH + y) * inW) * C + c;
      for (int i = 0; i < xmax; ++i) ss += (int)row[(size_t)(xmin + i) * C] * k[i];
    } else {
      const uint8_t* col = in + ((size_t)n * inH * inW + x) * C + c;
      for (int i = 0; i < xmax; ++i) ss += (int)col[(size_t)(xmin + i) * inW * C] * k[i];
    }
    const int v = ss >> 22;
    out[e] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

struct JitterOps {      // per image
  int op[4];            // 0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none -- applied in this order
  float factor[4];      // blend factor (brightness / contrast / saturation)
  int hue_add;          // uint8(hue_factor * 255), added to H modulo 256
  int enabled;          // 0: this image is not jittered at all (colorjitter_sample's `prob` draw failed)
  float color[3];       // round 5: diagonal of the 3x4 'color' matrix of jittering[4] (augmentations.py:266-277), applied last through
  int has_color;        //          PIL's Image.convert('RGB', matrix) arithmetic (Matrix.c): CLIPF((float)((double)(m * in) + 0.5))
};

__device__ __forceinline__ int lum(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ int blend8(int in1, int in2, float alpha) {
  const float t = (float)in1 + alpha * (float)(in2 - in1);
  if (alpha >= 0.f && alpha <= 1.f) return (int)t & 255;           // (UINT8)(float): truncation
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ void hue_shift(int& r, int& g, int& b, int add) {
  // Convert.c rgb2hsv_row
  const int maxc = r > g ? (r > b ? r : b) : (g > b ? g : b), minc = r < g ? (r < b ? r : b) : (g < b ? g : b);
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = (int)((double)h * 255.0);
    uh = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
    us = (int)((double)s * 255.0);
    us = us < 0 ? 0 : (us > 255 ? 255 : us);
  }
  uh = (uh + add) & 255;
  // Convert.c hsv2rgb
  if (us == 0) { r = g = b = uv; return; }
  const double hf = (double)(float)uh * 6.0 / 255.0;
  const int i = (int)floor(hf);
  const float f = (float)(hf - (double)(float)i);
  const float fs = (float)((double)(float)us / 255.0);
  const double vf = (double)(float)uv;
  int p = (int)rint(vf * (1.0 - (double)fs));
  int q = (int)rint(vf * (1.0 - (double)fs * (double)f));
  int t = (int)rint(vf * (1.0 - (double)fs * (1.0 - (double)f)));
  p = p < 0 ? 0 : (p > 255 ? 255 : p);
  q = q < 0 ? 0 : (q > 255 ? 255 : q);
  t = t < 0 ? 0 : (t > 255 ? 255 : t);
  switch (i % 6) {
    case 0: r = uv; g = t; b = p; break;
    case 1: r = q; g = uv; b = p; break;
    case 2: r = p; g = uv; b = t; break;
    case 3: r = p; g = q; b = uv; break;
    case 4: r = t; g = p; b = uv; break;
    default: r = uv; g = p; b = q; break;
  }
}

// apply operations [first, last) of the image's sequence to one pixel; `mean` = grey level of the contrast degenerate
__device__ __forceinline__ void apply_ops(const JitterOps& j, int first, int last, int mean, int& r, int& g, int& b) {
  for (int s = first; s < last; ++s) {
    const int op = j.op[s];
    const float a = j.factor[s];
    if (op == 0) { r = blend8(0, r, a); g = blend8(0, g, a); b = blend8(0, b, a); }
    else if (op == 1) { r = blend8(mean, r, a); g = blend8(mean, g, a); b = blend8(mean, b, a); }
    else if (op == 2) { const int l = lum(r, g, b); r = blend8(l, r, a); g = blend8(l, g, a); b = blend8(l, b, a); }
    else if (op == 3) hue_shift(r, g, b, j.hue_add);
  }
}

// Pillow Matrix.c, RGB -> RGB with a 12-tuple whose off-diagonal entries and offsets are 0 (the only form the reference builds):
//   float v = m * in + 0 + 0 + 0 + 0.5 (the 0.5 is a double literal: float product, double add, rounded back to float);
//   out = v <= 0 ? 0 : v >= 255.0F ? 255 : (UINT8)v.     Checked against Pillow 12.2 itself (tests/test_input_pipeline.py).
__device__ __forceinline__ int color_scale8(int in, float m) {
  const float v = (float)((double)(m * (float)in) + 0.5);
  return v <= 0.f ? 0 : (v >= 255.f ? 255 : (int)v);
}

__device__ __forceinline__ int contrast_pos(const JitterOps& j) {
  for (int s = 0; s < 4; ++s)
    if (j.op[s] == 1) return s;
  return 4;
}

// pass 1: sum of L over each image AFTER the operations that precede the contrast adjustment (its degenerate image is the
// mean grey of the image it is applied to).  lsum[n] must be zero on entry.
__global__ void __launch_bounds__(256) jitter_lsum_kernel(const uint8_t* __restrict__ img, const JitterOps* __restrict__ ops,
                                                          unsigned long long* __restrict__ lsum, int HW) {
  __shared__ unsigned long long red[4];
  const int n = blockIdx.y;
  const JitterOps j = ops[n];
  const int cp = contrast_pos(j);
  unsigned long long acc = 0;
  if (j.enabled && cp < 4) {
    const uint8_t* p = img + (size_t)n * HW * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
      int r = p[3 * i], g = p[3 * i + 1], b = p[3 * i + 2];
      apply_ops(j, 0, cp, 0, r, g, b);
      acc += (unsigned long long)lum(r, g, b);
    }
  }
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(&lsum[n], t);
  }
}

// pass 2: the whole sequence, then ToTensor: NHWC uint8 -> two NCHW float32 tensors (jittered, original)
__global__ void __launch_bounds__(256) jitter_totensor_kernel(const uint8_t* __restrict__ img, const JitterOps* __restrict__ ops,
                                                              const unsigned long long* __restrict__ lsum, float* __restrict__ out,
                                                              float* __restrict__ out_orig, int HW) {
  const int n = blockIdx.y;
  const JitterOps j = ops[n];
  const int cp = contrast_pos(j);
  // ImageStat mean of L, then int(mean + 0.5): exact in double for any image size here
  const int mean = (j.enabled && cp < 4) ? (int)((double)lsum[n] / (double)HW + 0.5) : 0;
  const uint8_t* p = img + (size_t)n * HW * 3;
  float* o = out + (size_t)n * 3 * HW;
  float* oo = out_orig ? out_orig + (size_t)n * 3 * HW : nullptr;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
    int r = p[3 * i], g = p[3 * i + 1], b = p[3 * i + 2];
    if (oo) { oo[i] = (float)r / 255.f; oo[HW + i] = (float)g / 255.f; oo[2 * HW + i] = (float)b / 255.f; }
    if (j.enabled) {
      apply_ops(j, 0, 4, mean, r, g, b);
      if (j.has_color) { r = color_scale8(r, j.color[0]); g = color_scale8(g, j.color[1]); b = color_scale8(b, j.color[2]); }
    }
    o[i] = (float)r / 255.f;
    o[HW + i] = (float)g / 255.f;
    o[2 * HW + i] = (float)b / 255.f;
  }
}

}  // namespace pnsfm

using namespace pnsfm;

extern "C" {

int pnsfm_resample8(const uint8_t* in, uint8_t* out, const int* kk, const int* bounds, int ksize, int N, int inH, int inW,
                    int outH, int outW, int C, int axis, void* stream) {
  if (axis != 0 && axis != 1) { set_error("resample8: axis must be 0 (vertical) or 1 (horizontal)"); return -1; }
  if ((axis == 1 && inH != outH) || (axis == 0 && inW != outW)) { set_error("resample8: one axis per pass"); return -1; }
  const size_t total = (size_t)N * outH * outW * C;
  int grid = (int)((total + 255) / 256);
  if (grid > 65535) grid = 65535;
  if (grid < 1) grid = 1;
  PNSFM_LAUNCH(resample8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, kk, bounds, ksize, N, inH, inW, outH, outW, C, axis);
  return check_launch("resample8");
}

int pnsfm_jitter_totensor(const uint8_t* img, const void* ops, unsigned long long* lsum_ws, float* out, float* out_orig, int N, int H,
                          int W, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  int e = (int)hipMemsetAsync(lsum_ws, 0, (size_t)N * sizeof(unsigned long long), s);
  if (e) { set_error("jitter_totensor: memset failed"); return e; }
  int gx = (HW + 256 * 8 - 1) / (256 * 8);
  if (gx < 1) gx = 1;
  if (gx > 1024) gx = 1024;
  PNSFM_LAUNCH(jitter_lsum_kernel, dim3(gx, N), dim3(256), 0, s, img, (const JitterOps*)ops, lsum_ws, HW);
  e = check_launch("jitter_lsum");
  if (e) return e;
  PNSFM_LAUNCH(jitter_totensor_kernel, dim3(gx, N), dim3(256), 0, s, img, (const JitterOps*)ops, (const unsigned long long*)lsum_ws, out,
               out_orig, HW);
  return check_launch("jitter_totensor");
}

}  // extern "C"
