// K3: first Darknet53 layer.  uint8 RGB NHWC in, 3x3 / stride 1 / pad 1 conv 3 -> 32, fused scale/bias/activation,
// f16 NHWC out.  K = 27 is padded to ONE 32-deep v_mfma_f32_16x16x32_f16 step, the A fragment of which is built
// in registers from a uint8 halo tile in LDS (no im2col buffer, no f16 copy of the image in HBM).  The layer is
// HBM-write bound (64 B out per pixel vs 3 B in), so the output mapping is arranged for full-line stores: the
// weight rows are permuted so that every lane ends up holding 8 CONSECUTIVE output channels of one pixel = one 16-B
// store, and a wave store covers 16 pixels x 64 B = 1 KiB contiguous.
//
// Replaces the preprocess (x/255 folded into `scale`) + first Conv2D/BN/LeakyReLU of `ObjectDetector.predict`
// (reference voc_validate.py:27).
#include <stdlib.h>

#include "conv_common.h"

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

// ---- K11 for the first layer: dW[co][tap*3 + c] += in_scale * sum_pixels dz[pixel][co] * x_u8[shifted pixel][c] ----------
// Tiny output (32 x 27), huge reduction (B*H*W pixels).  Round 1 ran it on the VALU (352 us at 32 x 320^2: one broadcast LDS
// read + 3 FMAs per tap per pixel per lane).  Now: the image is widened once to f16 with 8 channels per pixel (r, g, b, 0 x 5:
// 16 B, the granule of an LDS-DMA lane), and the GENERIC weight-gradient kernel (conv_wgrad.hip, MFMA, pixel index = k) runs on
// it as a 3x3 conv with Cin = 8, Cout = 32 -> per-split slabs [split][32][72]; a last kernel sums the slabs in ascending order,
// drops the 5 padding channels of every tap and applies in_scale.  No atomics: the result is bit-reproducible.  No dX (the
// input is the image).
namespace {
__global__ __launch_bounds__(256) void od_u8_to_f16x8(const uint8_t* __restrict__ x, f16* __restrict__ out, long long npix) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    const uint8_t* s = x + i * 3;
    f16x8 v = {(f16)(float)s[0], (f16)(float)s[1], (f16)(float)s[2], (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    *(f16x8*)(out + i * 8) = v;
  }
}
// one wave per output element: lane l adds the slabs l, l+64, ... in ascending order, then a fixed butterfly over the
// lanes -- the same summation tree on every run (bit-reproducible), 864 waves instead of one workgroup walking 512 slabs
__global__ __launch_bounds__(256) void od_conv_first_wgrad_finish(const float* __restrict__ slabs, int nsplit,
                                                                  float* __restrict__ dw, float in_scale) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // output element
  const int lane = threadIdx.x & 63;
  if (i >= 32 * 27) return;
  const int co = i / 27, k = i - co * 27;
  const int col = (k / 3) * 8 + (k % 3);
  float s = 0.f;
  for (int sp = lane; sp < nsplit; sp += 64) s += slabs[((long long)sp * 32 + co) * 72 + col];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) dw[i] += s * in_scale;
}

// ---- streaming form (W % 32 == 0): no widened copy of the image, no generic kernel ------------------------------------
// D[co][j] = sum_p dZ[p][co] * X[p][j], j = tap*3 + c (27 of 32 columns): per 32-pixel chunk of an image row ONE 2-KiB
// contiguous piece of dZ (LDS-DMA, then the transposing LDS read forms the MFMA operand: channel l15 of pixels 8*lq .. +7)
// and 16 byte loads per lane of the uint8 image (pixel 8*lq + e shifted by the lane's tap, channel c; the 3 x 34 x 3-byte
// window of a chunk lives in L1), 4 v_mfma_f32_16x16x32_f16.  Every wave walks its own chunks (chunk = wave index + k *
// total waves) with the next chunk's loads in flight, keeps the 32 x 32 f32 sums in 16 registers, the four waves of a
// workgroup are added in order through LDS, one [32][32] slab per workgroup, od_conv_first_wgrad_finish2 adds the slabs in
// a fixed order.  224 -> 95 us at 32 x 320^2 (the widening pass + the generic kernel at 14 % tile use + finish before; the
// byte loads, 16 vector-memory instructions per chunk and wave, are what bounds it now).
typedef __fp16 h4v __attribute__((__vector_size__(4 * sizeof(__fp16))));

__global__ __launch_bounds__(256) void od_conv_first_wgrad_stream(const uint8_t* __restrict__ x, const f16* __restrict__ dz,
                                                                  float* __restrict__ slabs, int H, int W,
                                                                  int chunks_per_row, int nchunks) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 2 * 2048];  // per wave: two 2-KiB dZ chunks; reused for the wave sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int tq = l15 >> 2, tp = l15 & 3;
  char* const mybuf = lds + wave * 4096;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;

  // this lane's two image columns j = fj*16 + l15 -> (dy, dx, c); j >= 27: zero column
  int jdy[2], jdx[2], jc[2];
  bool jok[2];
#pragma unroll
  for (int fj = 0; fj < 2; ++fj) {
    const int j = fj * 16 + l15;
    const int tap = j / 3;
    jok[fj] = j < 27;
    jc[fj] = j - tap * 3;
    jdy[fj] = tap / 3 - 1;
    jdx[fj] = tap % 3 - 1;
  }
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  unsigned char xb[2][8];  // raw bytes of the chunk whose loads are in flight
  auto issue = [&](int chunk, int buf) {
    const int row = chunk / chunks_per_row, x0 = (chunk - row * chunks_per_row) * 32;
    const int b = row / H, y = row - b * H;
    const f16* src = dz + ((long long)row * W + x0) * 32;
    glds16(src + lane * 8, mybuf + buf * 2048);
    glds16(src + 512 + lane * 8, mybuf + buf * 2048 + 1024);
#pragma unroll
    for (int fj = 0; fj < 2; ++fj) {
      const int py = y + jdy[fj];
      const bool rowok = jok[fj] && (unsigned)py < (unsigned)H;
      const uint8_t* rp = x + ((long long)(b * H + py) * W) * 3 + jc[fj];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int px = x0 + 8 * lq + e + jdx[fj];
        unsigned char v = 0;
        if (rowok && (unsigned)px < (unsigned)W) v = rp[px * 3];
        xb[fj][e] = v;
      }
    }
  };

  int chunk = gw;
  int buf = 0;
  if (chunk < nchunks) issue(chunk, 0);
  while (chunk < nchunks) {
    // everything of `chunk` has landed (18 vector-memory operations per chunk; nothing younger is in flight yet)
    wait_vmcnt<0>();
    f16x8 bf[2];
#pragma unroll
    for (int fj = 0; fj < 2; ++fj)
#pragma unroll
      for (int e = 0; e < 8; ++e) bf[fj][e] = (f16)(float)xb[fj][e];
    f16x8 af[2];
    const char* cb = mybuf + buf * 2048;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      h4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4v*)(cb + (8 * lq + tq) * 64 + f * 32 + tp * 8));
      h4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4v*)(cb + (8 * lq + 4 + tq) * 64 + f * 32 + tp * 8));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        af[f][e] = (f16)lo[e];
        af[f][4 + e] = (f16)hi[e];
      }
    }
    const int next = chunk + nw;
    if (next < nchunks) issue(next, buf ^ 1);  // (the transposed reads above are complete before their values are used below;
                                               //  the DMA targets the OTHER buffer)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int fj = 0; fj < 2; ++fj) acc[f][fj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[f], bf[fj], acc[f][fj], 0, 0, 0);
    chunk = next;
    buf ^= 1;
  }
  // ---- the four waves' sums, added in wave order; acc[f][fj][e] = D[f*16 + lq*4 + e][fj*16 + l15]
  wait_vmcnt<0>();
  __syncthreads();
  float* red = (float*)lds;  // [wave][32][32]
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int fj = 0; fj < 2; ++fj)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[(wave * 32 + f * 16 + lq * 4 + e) * 32 + fj * 16 + l15] = acc[f][fj][e];
  __syncthreads();
  float* out = slabs + (long long)blockIdx.x * 1024;
  for (int i = tid; i < 1024; i += 256) out[i] = ((red[i] + red[1024 + i]) + red[2048 + i]) + red[3072 + i];
}

// one wave per output element: lane l adds the slabs l, l+64, ... in ascending order, then a fixed butterfly over the lanes
__global__ __launch_bounds__(256) void od_conv_first_wgrad_finish2(const float* __restrict__ slabs, int nslabs,
                                                                   float* __restrict__ dw, float in_scale) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // output element co*27 + j
  const int lane = threadIdx.x & 63;
  if (i >= 32 * 27) return;
  const int co = i / 27, j = i - co * 27;
  float s = 0.f;
  for (int sp = lane; sp < nslabs; sp += 64) s += slabs[(long long)sp * 1024 + co * 32 + j];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) dw[i] += s * in_scale;
}
}  // namespace

static bool first_wgrad_use_stream(int B, int H, int W) {
  static int stream_ok = -1;
  if (stream_ok < 0) {
    const char* e = getenv("OD_FIRST_WGRAD_STREAM");  // 0 = the widened-copy + generic-kernel path (A/B timing)
    stream_ok = e ? atoi(e) : 1;
  }
  return stream_ok && W % 32 == 0 && (long long)B * H * W < (1LL << 31);
}

static int first_wgrad_stream_wgs(const od_ctx* ctx, int B, int H, int W) {
  const long long nchunks = (long long)B * H * (W / 32);
  long long wgs = 6LL * (ctx->num_cu > 0 ? ctx->num_cu : 256);  // 6 waves per SIMD (64 VGPRs, 16 KiB LDS per workgroup): all resident
  if (wgs * 4 > nchunks) wgs = (nchunks + 3) / 4;
  return (int)wgs;
}

extern "C" size_t od_conv_first_bwd_weight_workspace_bytes(od_ctx* ctx, int B, int H, int W) {
  if (!ctx || B <= 0 || H <= 0 || W <= 0) return 0;
  if (first_wgrad_use_stream(B, H, W)) return (size_t)first_wgrad_stream_wgs(ctx, B, H, W) * 1024 * sizeof(float);
  const size_t x8 = (size_t)B * H * W * 16;
  const int split = od_conv2d_bwd_weight_splits(ctx, B, H, W, 8, 32, 3, 1);
  return x8 + (size_t)split * 32 * 72 * sizeof(float);
}

extern "C" int od_conv_first_bwd_weight(od_ctx* ctx, const uint8_t* x, const void* dz, float* dw, int B, int H, int W,
                                        int Cout, float in_scale, void* workspace, size_t workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && x && dz && dw && workspace && Cout == 32 && B > 0 && H > 0 && W > 0,
             "od_conv_first_bwd_weight: bad argument (Cout must be 32)");
  if (workspace_bytes < od_conv_first_bwd_weight_workspace_bytes(ctx, B, H, W)) {
    od_set_error("od_conv_first_bwd_weight: workspace too small");
    return OD_ERR_WORKSPACE;
  }
  OD_REQUIRE(((uintptr_t)workspace & 15) == 0, "od_conv_first_bwd_weight: workspace must be 16-byte aligned");
  const long long npix = (long long)B * H * W;
  if (first_wgrad_use_stream(B, H, W)) {
    const int wgs = first_wgrad_stream_wgs(ctx, B, H, W);
    hipLaunchKernelGGL(od_conv_first_wgrad_stream, dim3(wgs), dim3(256), 0, (hipStream_t)stream, x, (const f16*)dz,
                       (float*)workspace, H, W, W / 32, (int)(npix / 32));
    OD_CHECK_LAUNCH();
    hipLaunchKernelGGL(od_conv_first_wgrad_finish2, dim3(32 * 27 / 4), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, wgs, dw, in_scale);
    OD_CHECK_LAUNCH();
    return OD_OK;
  }
  f16* x8 = (f16*)workspace;
  float* slabs = (float*)((char*)workspace + (size_t)npix * 16);
  long long blocks = (npix + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(od_u8_to_f16x8, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x8, npix);
  OD_CHECK_LAUNCH();
  int nsplit = 0;
  if (int rc = od_wgrad_slabs_impl(ctx, x8, dz, slabs, B, H, W, 8, 32, 3, 1, stream, &nsplit)) return rc;
  hipLaunchKernelGGL(od_conv_first_wgrad_finish, dim3(32 * 27 / 4), dim3(256), 0, (hipStream_t)stream, slabs, nsplit, dw,
                     in_scale);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
