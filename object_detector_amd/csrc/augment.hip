// K14: device-side pixel augmentation of the training generator (reference check_generator.py:17-18,
// docs/MODEL.md:60-64: Random Erasing "and anything else at hand").  Augmentation PARAMETERS are sampled on the host
// (od_gen.sample_params); this kernel does the pixel work for a whole batch: crop + resize (bilinear, half-pixel
// centres) + horizontal flip + saturation / contrast / brightness + Random Erasing rectangles, uint8 in -> uint8 NHWC out.
// HBM-bound elementwise; one thread per output pixel.  f32 arithmetic in a fixed order (-ffp-contract=off), mirrored
// op for op by oracle/augment.py => bit-exact.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void od_augment_k(const uint8_t* __restrict__ src, const od_aug_params* __restrict__ prm,
                                                    uint8_t* __restrict__ out, int H, int W) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const od_aug_params p = prm[b];
  const uint8_t* img = src + p.src_offset;
  float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
  if (p.flip) u = 1.0f - u;
  // source position in pixel units (half-pixel centres), clamped to the crop rectangle's pixel range
  float sx = (p.crop_x1 + u * (p.crop_x2 - p.crop_x1)) * (float)p.src_w - 0.5f;
  float sy = (p.crop_y1 + v * (p.crop_y2 - p.crop_y1)) * (float)p.src_h - 0.5f;
  sx = fminf(fmaxf(sx, 0.f), (float)(p.src_w - 1));
  sy = fminf(fmaxf(sy, 0.f), (float)(p.src_h - 1));
  const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
  const int x1 = min(x0 + 1, p.src_w - 1), y1 = min(y0 + 1, p.src_h - 1);
  const float fx = sx - (float)x0, fy = sy - (float)y0;
  float c[3];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float p00 = (float)img[((long long)y0 * p.src_w + x0) * 3 + ch];
    const float p01 = (float)img[((long long)y0 * p.src_w + x1) * 3 + ch];
    const float p10 = (float)img[((long long)y1 * p.src_w + x0) * 3 + ch];
    const float p11 = (float)img[((long long)y1 * p.src_w + x1) * 3 + ch];
    const float top = p00 + (p01 - p00) * fx;
    const float bot = p10 + (p11 - p10) * fx;
    c[ch] = top + (bot - top) * fy;
  }
  const float gray = (c[0] * 0.299f + c[1] * 0.587f) + c[2] * 0.114f;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    float t = gray + (c[ch] - gray) * p.saturation;
    t = (t - 127.5f) * p.contrast + 127.5f;
    t = t + p.brightness;
    c[ch] = fminf(fmaxf(floorf(t + 0.5f), 0.f), 255.f);
  }
  const float cu = ((float)x + 0.5f) / (float)W, cv = ((float)y + 0.5f) / (float)H;  // erasing acts on OUTPUT coords
  for (int e = 0; e < p.n_erase; ++e) {
    if (cu >= p.erase[e][0] && cu < p.erase[e][2] && cv >= p.erase[e][1] && cv < p.erase[e][3]) {
      c[0] = (float)p.erase_rgb[e][0];
      c[1] = (float)p.erase_rgb[e][1];
      c[2] = (float)p.erase_rgb[e][2];
    }
  }
  uint8_t* o = out + (((long long)b * H + y) * W + x) * 3;
  o[0] = (uint8_t)c[0];
  o[1] = (uint8_t)c[1];
  o[2] = (uint8_t)c[2];
}

}  // namespace

extern "C" int od_aug_params_bytes(void) { return (int)sizeof(od_aug_params); }

extern "C" int od_augment_batch(od_ctx* ctx, const uint8_t* src, const void* params, uint8_t* out, int B, int H, int W,
                                void* stream) {
  OD_REQUIRE(ctx && src && params && out && B > 0 && B <= 65535 && H > 0 && W > 0, "od_augment_batch: bad argument");
  hipLaunchKernelGGL(od_augment_k, dim3(od_ceil_div(H * W, 256), B), dim3(256), 0, (hipStream_t)stream, src,
                     (const od_aug_params*)params, out, H, W);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
