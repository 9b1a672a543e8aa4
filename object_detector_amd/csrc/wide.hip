// Wide (f32) add paths of the mixed-precision inference plan (net.py precision="mixed"; DESIGN.md §5 "Numerics to
// north_star's tolerance").  Every convolution still runs on f16 MFMA operands; what this kernel adds is
//   * an f32 RESIDUAL STREAM: x32' = x32 + y32, with the f16 operand copy for the next convolution rounded ONCE from the
//     f32 sum (the default plan rounds the stream itself after every block: 23 roundings on the identity path), and
//   * SPLIT OPERANDS: a tensor as an f16 (hi, lo) pair, [M, 2C] = [hi | lo] with lo = f16(v - hi): a convolution over 2C
//     input channels with the weights repeated sees ~22 significant bits of v on plain f16 MFMAs.
// HBM-bound elementwise pass: 8 channels per thread, 16/32-byte accesses, rows in lock step over the whole chip.
#include "common.h"

template <bool RES32>
__global__ __launch_bounds__(256) void od_wide_add_k(const float* __restrict__ y, const void* __restrict__ res,
                                                     float* __restrict__ out32, f16* __restrict__ out16,
                                                     f16* __restrict__ hilo, long long nvec, int G, int up2, int H,
                                                     int W) {
  // vector v = (row r, 8-channel group g); hilo rows are 2*C wide
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
    const f32x4 a0 = *(const f32x4*)(y + v * 8), a1 = *(const f32x4*)(y + v * 8 + 4);
    float s[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const long long r = v / G;
    const int g = (int)(v - r * G);
    if (res) {
      long long rv = v;  // the residual's vector: the same element, or its nearest-neighbour parent on the half-size map
      if (up2) {
        long long pix = r;
        const int x = (int)(pix % W);
        pix /= W;
        const int yy = (int)(pix % H);
        const long long b = pix / H;
        rv = ((b * (H >> 1) + (yy >> 1)) * (W >> 1) + (x >> 1)) * G + g;
      }
      if (RES32) {
        const f32x4 r0 = *(const f32x4*)((const float*)res + rv * 8), r1 = *(const f32x4*)((const float*)res + rv * 8 + 4);
        s[0] += r0.x; s[1] += r0.y; s[2] += r0.z; s[3] += r0.w;
        s[4] += r1.x; s[5] += r1.y; s[6] += r1.z; s[7] += r1.w;
      } else {
        const f16x8 rh = *(const f16x8*)((const f16*)res + rv * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)rh[e];
      }
    }
    if (out32) {
      f32x4 o0 = {s[0], s[1], s[2], s[3]}, o1 = {s[4], s[5], s[6], s[7]};
      *(f32x4*)(out32 + v * 8) = o0;
      *(f32x4*)(out32 + v * 8 + 4) = o1;
    }
    f16x8 hi;
#pragma unroll
    for (int e = 0; e < 8; ++e) hi[e] = (f16)s[e];
    if (out16) *(f16x8*)(out16 + v * 8) = hi;
    if (hilo) {
      f16x8 lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) lo[e] = (f16)(s[e] - (float)hi[e]);
      f16* row = hilo + r * (long long)(16 * G);
      *(f16x8*)(row + g * 8) = hi;
      *(f16x8*)(row + 8 * G + g * 8) = lo;
    }
  }
}

extern "C" int od_wide_add(od_ctx* ctx, const od_wide_desc* d, void* stream) {
  OD_REQUIRE(ctx && d && d->y && d->M > 0 && d->C > 0 && d->C % 8 == 0, "od_wide_add: bad argument (C must be a multiple of 8)");
  OD_REQUIRE(d->out32 || d->out16 || d->out_hilo, "od_wide_add: no output");
  OD_REQUIRE((((uintptr_t)d->y | (uintptr_t)d->res | (uintptr_t)d->out32 | (uintptr_t)d->out16 | (uintptr_t)d->out_hilo) & 15) == 0,
             "od_wide_add: tensors must be 16-byte aligned");
  OD_REQUIRE(!d->res_up2 || (d->res && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0 && d->M % ((long long)d->H * d->W) == 0),
             "od_wide_add: res_up2 needs res, even H and W, and M = B * H * W");
  OD_REQUIRE(!d->res_up2 || (const void*)d->res != (const void*)d->out32, "od_wide_add: res_up2 cannot run in place");
  const long long nvec = d->M * (d->C / 8);
  long long blocks = (nvec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (d->res && d->res_f32)
    hipLaunchKernelGGL(od_wide_add_k<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d->y, d->res, d->out32,
                       (f16*)d->out16, (f16*)d->out_hilo, nvec, d->C / 8, d->res_up2, d->H, d->W);
  else
    hipLaunchKernelGGL(od_wide_add_k<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d->y, d->res, d->out32,
                       (f16*)d->out16, (f16*)d->out_hilo, nvec, d->C / 8, d->res_up2, d->H, d->W);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
