// K11a' (backward-data of the FIRST stride-2 convolution, 64 -> 32 channels): dX [B, 2Hs, 2Ws, 32] from dZ [B, Hs, Ws, 64].
//
// The generic transposed path of conv_mfma.hip orders the GEMM rows by output parity class and skips the taps that are zero
// for a whole tile; on this layer that is 25 600 workgroups of 1-4 K steps each, 64-byte output rows at a 128-byte stride,
// half of every 64-wide n tile empty: 350 us for 315 MB of traffic.  Here the layer is what it is, a streaming kernel:
//
//   persistent workgroups (4 waves, 2 per CU); all nine 64 x 32 weight taps stay in LDS (36 KiB); a tile = 4 x 16 dZ
//   pixels (+ one halo row / column = 5 x 17, read once) -> 8 x 32 output pixels; wave w owns dZ row w of the tile = ONE
//   16-pixel MFMA fragment, and computes the four output parity classes of those pixels from the 1 / 2 / 2 / 4 taps
//   that reach them (36 v_mfma_f32_16x16x32_f16 per tile and wave, every tap's source is an LDS row offset);
//   dZ tiles stream through a 3-deep LDS ring by LDS-DMA with a counted vmcnt (stores and the next tiles' DMAs stay in
//   flight), results leave straight from the accumulators as 16-byte stores (v_permlane16_swap pairs two 16-channel
//   fragments into 8 consecutive channels per lane, as in conv_8ph.hip).
//
// out[Y, X, n] = sum over (ty, tx) with (Y + ty - 1, X + tx - 1) even of dZ[(Y + ty - 1) / 2, (X + tx - 1) / 2, :] . wt[n][(ty*3 + tx)*64 + :]
// -- the same sum, on the same packed weights (od_pack_weights' backward layout), as od_conv2d_fwd(transposed = 1).
//
// Replaces the input-gradient half of the Keras Conv2D backward pass of `b.down1` in the reference's (unseen) training
// loop (docs/MODEL.md:15-17 names the backbone).
#include "conv_common.h"

namespace {

constexpr int T_R = 4, T_C = 16;                 // dZ pixels per tile
constexpr int T_PIX = 96;                        // LDS pixel rows per tile buffer: 5 x 17 = 85, padded to 12 DMA pieces
constexpr int T_ROWB = 128;                      // 64 f16 per pixel
constexpr int T_NBUF = 3;
constexpr int T_WBYTES = 9 * 32 * T_ROWB;        // weights: row = tap * 32 + n
constexpr int T_LDS = T_WBYTES + T_NBUF * T_PIX * T_ROWB;

struct TconvKP {
  const f16* dz;    // [B, Hs, Ws, 64]
  const f16* wt;    // [256][576] packed backward weights (rows 0..31 used)
  const float* scale;
  const float* bias;
  f16* out;         // [B, 2Hs, 2Ws, 32]
  const f16* zero;
  int B, Hs, Ws, Kstride;
  int tiles_x, tiles_per_img, ntiles;
};

__global__ __launch_bounds__(256, 2) void od_tconv_64_32(TconvKP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem;
  char* const tlds = smem + T_WBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- weights: 36 pieces of 8 rows x 128 B; piece q -> rows 8q .. 8q+7 (row = tap*32 + n) -------------------------
  {
    const int r8 = lane >> 3;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int q = wave * 9 + k;
      const int row = q * 8 + r8;
      const int tap = row >> 5, n = row & 31;
      const int lc = (lane & 7) ^ (row & 7);
      glds16(p.wt + ((long long)n * p.Kstride + tap * 64 + lc * 8), wlds + q * 1024);
    }
  }

  // ---- tile staging: 12 pieces of 8 pixel rows; wave w issues pieces w, w+4, w+8 ------------------------------------
  auto stage = [&](int tile, int buf) {
    const int img = tile / p.tiles_per_img, t2 = tile - img * p.tiles_per_img;
    const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;
    const int u0 = ty * T_R, v0 = tx * T_C;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int q = (wave + 4 * k) * 8 + (lane >> 3);  // pixel slot
      const int r = q / 17, c = q - r * 17;
      const int u = u0 + r, v = v0 + c;
      const int lc = (lane & 7) ^ (q & 7);
      const bool ok = q < 85 && u < p.Hs && v < p.Ws;
      const f16* src = ok ? p.dz + ((((long long)img * p.Hs + u) * p.Ws + v) * 64 + lc * 8) : p.zero;
      glds16(src, tlds + buf * (T_PIX * T_ROWB) + (wave + 4 * k) * 1024);
    }
  };

  const int first = blockIdx.x, step = gridDim.x;
  int nmine = first < p.ntiles ? (p.ntiles - first + step - 1) / step : 0;
  if (nmine > 0) stage(first, 0);
  if (nmine > 1) stage(first + step, 1);

  // this lane's 8 output channels after the permlane pairing, and their scale / bias
  const int n8 = (lq & 1) * 16 + (lq >> 1) * 8;
  float sc[8], bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = p.scale[n8 + e];
    bi[e] = p.bias[n8 + e];
  }
  // weight fragment base: row = tap*32 + f*16 + l15, chunk (kh*4 + lq) ^ (l15 & 7)
  const int fw = l15 * T_ROWB + ((lq ^ (l15 & 7)) * 16);
  const int Ho = 2 * p.Hs, Wo = 2 * p.Ws;

  for (int it = 0; it < nmine; ++it) {
    const int tile = first + it * step;
    const int buf = it % T_NBUF;
    // The DMAs of tile `it` were issued two iterations ago; vector memory retires in order, and behind them this wave has
    // issued stores(it-2) [4], the DMAs of tile it+1 [3, if there is one] and stores(it-1) [4]: those may stay in flight.
    if (it < 2) {
      wait_vmcnt<0>();  // weights + the prologue's two tiles
    } else if (it + 1 < nmine) {
      wait_vmcnt<11>();
    } else {
      wait_vmcnt<8>();
    }
    __builtin_amdgcn_s_barrier();
    if (it + 2 < nmine) stage(tile + 2 * step, (it + 2) % T_NBUF);

    const int img = tile / p.tiles_per_img, t2 = tile - img * p.tiles_per_img;
    const int tyy = t2 / p.tiles_x, txx = t2 - tyy * p.tiles_x;
    const int u = tyy * T_R + wave, v0 = txx * T_C;
    const char* tb = tlds + buf * (T_PIX * T_ROWB);

    // source fragments: (dy, dx) in {0,1}^2 -> pixel slot (wave + dy)*17 + l15 + dx, two k halves
    f16x8 xs[2][2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int q = (wave + dy) * 17 + l15 + dx;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
          xs[dy][dx][kh] = *(const f16x8*)(tb + q * T_ROWB + ((((kh * 4 + lq) ^ (q & 7))) * 16));
      }
    f32x4 acc[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int f = 0; f < 2; ++f) acc[c][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // class (py, px): py = 0 -> ty = 1 (dy 0); py = 1 -> ty = 0 (dy 0), ty = 2 (dy 1); same for x
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const int py = ty == 1 ? 0 : 1, px = tx == 1 ? 0 : 1;
        const int dy = ty == 2 ? 1 : 0, dx = tx == 2 ? 1 : 0;
        const int tap = ty * 3 + tx;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const f16x8 wf = *(const f16x8*)(wlds + (tap * 32 + f * 16) * T_ROWB + (fw ^ (kh * 64)));
            acc[py * 2 + px][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xs[dy][dx][kh], acc[py * 2 + px][f], 0, 0, 0);
          }
      }
    od_mfma_results_ready();
    {  // (Hs % 4 == 0 and Ws % 16 == 0: every row / column of the tile exists -> exactly four stores per wave and tile,
       //  which the counted waits above rely on)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[c][0][e], b = acc[c][1][e];
          od_permlane16_swap(a, b);
          o[e] = a;
          o[4 + e] = b;
        }
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)(o[e] * sc[e] + bi[e]);
        const int vv = v0 + l15;
        *(f16x8*)(p.out + ((((long long)img * Ho + (2 * u + py)) * Wo + (2 * vv + px)) * 32 + n8)) = h;
      }
    }
  }
}

}  // namespace

// host side: called by od_conv2d_fwd for transposed = 1, Cin = 64, Cout = 32, linear epilogue, no residual
bool od_tconv_small_supported(const od_conv_desc* d) {
  return d->transposed && d->ksize == 3 && d->stride == 2 && d->Cin == 64 && d->Cout == 32 && d->act == OD_ACT_LINEAR &&
         d->res_mode == OD_RES_NONE && d->out_dtype == OD_DT_F16 && d->out_batch_stride == 0 && d->out_pix_stride == 0 &&
         d->H % T_R == 0 && d->W % T_C == 0 && !d->bn_partials && !d->w2;
}

int od_tconv_small_launch(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run) {
  if (kernel_name) *kernel_name = "od_tconv_64_32";
  if (dry_run) return OD_OK;
  TconvKP p;
  p.dz = (const f16*)d->x;
  p.wt = (const f16*)d->w;
  p.scale = d->scale;
  p.bias = d->bias;
  p.out = (f16*)d->out;
  p.zero = (const f16*)ctx->zero_page;
  p.B = d->B;
  p.Hs = d->H;
  p.Ws = d->W;
  p.Kstride = od_round_up(9 * d->Cin, 64);
  p.tiles_x = d->W / T_C;
  p.tiles_per_img = (d->H / T_R) * p.tiles_x;
  p.ntiles = d->B * p.tiles_per_img;
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int grid = 2 * cus;
  if (grid > p.ntiles) grid = p.ntiles;
  if (int rc = od_ensure_lds(ctx, (const void*)&od_tconv_64_32, (size_t)T_LDS)) return rc;
  hipLaunchKernelGGL(od_tconv_64_32, dim3(grid), dim3(256), T_LDS, stream, p);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
