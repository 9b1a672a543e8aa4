// K3: first Darknet53 layer.  uint8 RGB NHWC in, 3x3 / stride 1 / pad 1 conv 3 -> 32, fused scale/bias/activation,
// f16 NHWC out.  K = 27 is padded to ONE 32-deep v_mfma_f32_16x16x32_f16 step, the A fragment of which is built
// in registers from a uint8 halo tile in LDS (no im2col buffer, no f16 copy of the image in HBM).  The layer is
// HBM-write bound (64 B out per pixel vs 3 B in), so the output mapping is arranged for full-line stores: the
// weight rows are permuted so that every lane ends up holding 8 CONSECUTIVE output channels of one pixel = one 16-B
// store, and a wave store covers 16 pixels x 64 B = 1 KiB contiguous.
//
// Replaces the preprocess (x/255 folded into `scale`) + first Conv2D/BN/LeakyReLU of `ObjectDetector.predict`
// (reference voc_validate.py:27).
#include "common.h"

namespace {

constexpr int TH = 32, TW = 32;              // output pixels per workgroup (TH = 8 / 16 / 32 measured: 69.7 / 61.4 / 58.1 us)
constexpr int LW = TW + 2, LH = TH + 2;      // halo tile
constexpr int ROWB = LW * 3;                 // contiguous source bytes per halo row

__global__ __launch_bounds__(256) void od_conv_first(const uint8_t* __restrict__ x, const f16* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ bias,
                                                     f16* __restrict__ out, int H, int W, int act, float alpha) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[LH * LW * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lq = lane >> 4;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, b = blockIdx.z;

  // halo tile: LH rows of ROWB contiguous bytes -> [row][pixel][4] (4th byte 0)
  for (int i = tid; i < LH * LW; i += 256) tile[i * 4 + 3] = 0;
  for (int i = tid; i < LH * ROWB; i += 256) {
    const int r = i / ROWB, bt = i - r * ROWB;
    const int px = bt / 3, c = bt - px * 3;
    const int gy = y0 - 1 + r, gx = x0 - 1 + px;
    uint8_t v = 0;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = x[((long long)(b * H + gy) * W + gx) * 3 + c];
    tile[(r * LW + px) * 4 + c] = v;
  }

  // weight fragments (A operand: row = output channel, k = 8*lq + j); row r of n-tile t is channel (r/4)*8 + t*4 + r%4
  f16x8 wf[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ch = (l15 >> 2) * 8 + t * 4 + (l15 & 3);
    wf[t] = *(const f16x8*)(w + ch * 32 + lq * 8);
  }
  // LDS byte offsets of this lane's 8 k's relative to the (un-shifted) pixel
  int koff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = lq * 8 + j;
    const int tap = k / 3, c = k - tap * 3;
    const int dy = tap / 3, dx = tap - dy * 3;
    koff[j] = k < 27 ? (dy * LW + dx) * 4 + c : 3;
  }
  float sc[8], bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[lq * 8 + e];
    bi[e] = bias[lq * 8 + e];
  }
  __syncthreads();

#pragma unroll 4
  for (int mt = 0; mt < TH / 2; ++mt) {
    const int yy = wave * (TH / 4) + (mt >> 1);
    const int xx = (mt & 1) * 16 + l15;
    const uint8_t* pbase = tile + (yy * LW + xx) * 4;
    f16x8 xf;
#pragma unroll
    for (int j = 0; j < 8; ++j) xf[j] = (f16)(float)pbase[koff[j]];
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], xf, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], xf, a1, 0, 0, 0);
    const int gy = y0 + yy, gx = x0 + xx;
    if (gy < H && gx < W) {
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v0 = a0[e] * sc[e] + bi[e];
        float v1 = a1[e] * sc[4 + e] + bi[4 + e];
        if (act == OD_ACT_LEAKY) {
          v0 = v0 > 0.f ? v0 : v0 * alpha;
          v1 = v1 > 0.f ? v1 : v1 * alpha;
        } else if (act == OD_ACT_ELU) {
          v0 = v0 > 0.f ? v0 : alpha * expm1f(v0);
          v1 = v1 > 0.f ? v1 : alpha * expm1f(v1);
        }
        h[e] = (f16)v0;
        h[4 + e] = (f16)v1;
      }
      *(f16x8*)(out + ((long long)(b * H + gy) * W + gx) * 32 + lq * 8) = h;
    }
  }
}

}  // namespace

const char* od_conv_first_kernel_name() { return "od_conv_first"; }

extern "C" int od_conv_first_fwd(od_ctx* ctx, const uint8_t* x, const void* w, const float* scale, const float* bias,
                                 void* out, int B, int H, int W, int Cout, int act, float alpha, void* stream) {
  OD_REQUIRE(ctx && x && w && scale && bias && out, "od_conv_first_fwd: null argument");
  OD_REQUIRE(Cout == 32, "od_conv_first_fwd: Cout must be 32 (got %d)", Cout);
  OD_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535, "od_conv_first_fwd: bad dims");
  OD_REQUIRE((long long)B * H * W * 32 < (1LL << 31), "od_conv_first_fwd: tensor too large");
  dim3 grid(od_ceil_div(W, TW), od_ceil_div(H, TH), B);
  hipLaunchKernelGGL(od_conv_first, grid, dim3(256), 0, (hipStream_t)stream, x, (const f16*)w, scale, bias, (f16*)out,
                     H, W, act, alpha);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

// ---- K11 for the first layer: dW[co][tap*3 + c] += sum_pixels dz[pixel][co] * x_u8[shifted pixel][c] * in_scale -----
// Tiny output (32 x 27), huge reduction: VALU kernel, one 8x32-pixel tile per workgroup, halo in LDS, per-workgroup
// partial sums reduced through LDS, one f32 atomic per (co, k) per workgroup.  No dX (the input is the image).
namespace {
constexpr int GTH = 8, GLH = GTH + 2;  // this kernel's own tile: 8 rows (one per 32-thread part) x 32 pixels
__global__ __launch_bounds__(256) void od_conv_first_wgrad(const uint8_t* __restrict__ x, const f16* __restrict__ dz,
                                                           float* __restrict__ dw, int H, int W, float in_scale) {
  // halo tile converted to f32 ONCE ([row][pixel][4], 4th = 0): the inner loop is one 16-B LDS read + 3 FMAs per tap
  // instead of a byte read + convert + FMA per (tap, channel)
  __shared__ __attribute__((aligned(16))) float tile[GLH * LW * 4];
  __shared__ float red[8][32][28];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * GTH, b = blockIdx.z;
  for (int i = tid; i < GLH * LW; i += 256) {
    const int r = i / LW, px = i - r * LW;
    const int gy = y0 - 1 + r, gx = x0 - 1 + px;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
      const uint8_t* src = x + ((long long)(b * H + gy) * W + gx) * 3;
      v[0] = (float)src[0];
      v[1] = (float)src[1];
      v[2] = (float)src[2];
    }
    *(f32x4*)(tile + i * 4) = v;
  }
  __syncthreads();
  const int co = tid & 31, part = tid >> 5;  // part = tile row (8 rows of 32 pixels)
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  const int gy = y0 + part;
  if (gy < H) {
    const int nx = min(TW, W - x0);
    const f16* dzp = dz + ((long long)(b * H + gy) * W + x0) * 32 + co;
    for (int xx = 0; xx < nx; ++xx) {
      const float d = (float)dzp[xx * 32];
      const float* pb = tile + (part * LW + xx) * 4;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const f32x4 v = *(const f32x4*)(pb + ((tap / 3) * LW + tap % 3) * 4);
        acc[tap * 3 + 0] += d * v[0];
        acc[tap * 3 + 1] += d * v[1];
        acc[tap * 3 + 2] += d * v[2];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) red[part][co][k] = acc[k];
  __syncthreads();
  for (int i = tid; i < 32 * 27; i += 256) {
    const int c2 = i / 27, k = i - c2 * 27;
    float s = 0.f;
#pragma unroll
    for (int p2 = 0; p2 < 8; ++p2) s += red[p2][c2][k];
    atomicAdd(dw + c2 * 27 + k, s * in_scale);
  }
}
}  // namespace

extern "C" int od_conv_first_bwd_weight(od_ctx* ctx, const uint8_t* x, const void* dz, float* dw, int B, int H, int W,
                                        int Cout, float in_scale, void* stream) {
  OD_REQUIRE(ctx && x && dz && dw && Cout == 32 && B > 0 && H > 0 && W > 0 && B <= 65535,
             "od_conv_first_bwd_weight: bad argument (Cout must be 32)");
  dim3 grid(od_ceil_div(W, TW), od_ceil_div(H, GTH), B);
  hipLaunchKernelGGL(od_conv_first_wgrad, grid, dim3(256), 0, (hipStream_t)stream, x, (const f16*)dz, dw, H, W, in_scale);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
