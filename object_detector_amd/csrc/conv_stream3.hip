// K1s: 3x3 convolutions of the 160x160 / 320x320 maps with 32 <-> 64 channels as streaming kernels (the training step's
// forward / backward-data of `b.down1` and `b.s1.0.b`; inference runs these layers inside od_stem / od_bneck<64>).
//
// Those layers move 160-315 MB for 30 GFLOP: they are HBM-bound, and on the implicit-GEMM table kernels they spend their
// time in per-tile prologues (6 400-12 800 workgroups of 5-9 K steps) -- 78-110 us against 35-70 us of traffic.  Same recipe
// as conv_tconv.hip: persistent workgroups (4 waves, 2 per CU), all nine weight taps resident in LDS (36 KiB either way:
// 9 x 32 x 64 or 9 x 64 x 32 f16), a tile = 4 x 16 OUTPUT pixels whose input window ((4-1)*S + 3) x ((16-1)*S + 3) pixels
// streams through a 3-deep LDS ring by LDS-DMA (counted vmcnt: the stores and the next windows stay in flight), wave w
// owns output row w of the tile = one 16-pixel MFMA fragment, 36 v_mfma_f32_16x16x32_f16 per tile and wave, 16-byte
// stores straight from the accumulators.  Epilogue: scale / bias / activation as od_conv2d_fwd defines them (no residual).
//
// Replaces the Conv2D (+ BatchNormalization + activation) layers `b.down1`, `b.s1.0.b` of the reference's network
// (docs/MODEL.md:15-17) in the layer-by-layer plans, and their input-gradient halves in training.
#include "conv_common.h"

namespace {

struct Stream3KP {
  const f16* x;      // [B, H, W, KC]
  const f16* w;      // packed [256][9*KC (padded to 64)]: row n, k = tap*KC + c
  const float* scale;
  const float* bias;
  f16* out;          // [B, Ho, Wo, NC]
  const f16* zero;
  int H, W, Ho, Wo, Kstride;
  int act;
  float alpha;
  int tiles_x, tiles_per_img, ntiles;
};

template <int KC, int NC, int S>
struct S3Cfg {
  static constexpr int WIN_H = 3 * S + 3, WIN_W = 15 * S + 3;      // input window of a 4 x 16 output tile
  static constexpr int PXB = KC * 2;                                // bytes per pixel row in LDS
  static constexpr int PX_PER_PIECE = 1024 / PXB;                   // pixels per 1-KiB DMA piece
  static constexpr int PIECES = ((WIN_H * WIN_W + PX_PER_PIECE - 1) / PX_PER_PIECE + 3) / 4 * 4;  // multiple of 4 waves
  static constexpr int ND = PIECES / 4;                             // DMA pieces per wave and tile
  static constexpr int SLOTS = PIECES * PX_PER_PIECE;
  static constexpr int WBYTES = 9 * NC * PXB;                       // weights: row = tap*NC + n, KC halfs
  static constexpr int WPIECES = WBYTES / 1024;                     // 36
  static constexpr int NS = NC / 32;                                // 16-byte store instructions per wave and tile
  static constexpr int CH = PXB / 16;                               // 16-byte chunks per pixel row (8 or 4)
  // window ring: three buffers (two tiles ahead) where two workgroups still fit a CU's 160 KiB, else two (one ahead)
  static constexpr int NBUF = (WBYTES + 3 * SLOTS * PXB) * 2 <= 160 * 1024 ? 3 : 2;
  static constexpr int LDS = WBYTES + NBUF * SLOTS * PXB;
};

// 16-byte chunk swizzle inside a pixel row: 128-B rows (8 chunks) XOR the row's low 3 bits, 64-B rows (4 chunks) XOR bits
// 1-2 of the row (two rows share a 128-B bank line) -- the 16 consecutive rows a fragment read touches then cover all banks
template <int CH>
__device__ __forceinline__ int s3_key(int row) {
  return CH == 8 ? (row & 7) : ((row >> 1) & 3);
}

template <int KC, int NC, int S>
__global__ __launch_bounds__(256, 2) void od_conv_stream3(Stream3KP p) {
  using Cf = S3Cfg<KC, NC, S>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wlds = smem;
  char* const tlds = smem + Cf::WBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  constexpr int LPP = Cf::CH;                   // lanes per pixel row inside a DMA piece
  constexpr int PPP = Cf::PX_PER_PIECE;

  // ---- weights: WPIECES pieces of 1 KiB = PPP rows each; row = tap*NC + n ---------------------------------------------
#pragma unroll
  for (int k = 0; k < Cf::WPIECES / 4; ++k) {
    const int q = wave * (Cf::WPIECES / 4) + k;
    const int row = q * PPP + lane / LPP;
    const int tap = row / NC, n = row - tap * NC;
    const int lc = (lane % LPP) ^ s3_key<Cf::CH>(row);
    glds16(p.w + ((long long)n * p.Kstride + tap * KC + lc * 8), wlds + q * 1024);
  }

  auto stage = [&](int tile, int buf) {
    const int img = tile / p.tiles_per_img, t2 = tile - img * p.tiles_per_img;
    const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;
    const int y0 = ty * 4 * S - 1, x0 = tx * 16 * S - 1;  // window origin in the input (pad 1)
#pragma unroll
    for (int k = 0; k < Cf::ND; ++k) {
      const int piece = wave + 4 * k;
      const int q = piece * PPP + lane / LPP;  // pixel slot
      const int r = q / Cf::WIN_W, c = q - r * Cf::WIN_W;
      const int yy = y0 + r, xx = x0 + c;
      const int lc = (lane % LPP) ^ s3_key<Cf::CH>(q);
      const bool ok = q < Cf::WIN_H * Cf::WIN_W && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const f16* src = ok ? p.x + ((((long long)img * p.H + yy) * p.W + xx) * KC + lc * 8) : p.zero;
      glds16(src, tlds + buf * (Cf::SLOTS * Cf::PXB) + piece * 1024);
    }
  };

  const int first = blockIdx.x, step = gridDim.x;
  const int nmine = first < p.ntiles ? (p.ntiles - first + step - 1) / step : 0;
  constexpr int NBUF = Cf::NBUF, AHEAD = NBUF - 1;
  if (nmine > 0) stage(first, 0);
  if (AHEAD > 1 && nmine > 1) stage(first + step, 1);

  // epilogue constants: store instruction s covers channels s*32 .. s*32+31; this lane's 8 channels inside it
  const int n8 = (lq & 1) * 16 + (lq >> 1) * 8;
  float sc[Cf::NS][8], bi[Cf::NS][8];
#pragma unroll
  for (int s = 0; s < Cf::NS; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[s][e] = p.scale[s * 32 + n8 + e];
      bi[s][e] = p.bias[s * 32 + n8 + e];
    }

  for (int it = 0; it < nmine; ++it) {
    const int tile = first + it * step;
    const int buf = it % NBUF;
    // vector memory retires in order.  Three buffers: behind the DMAs of tile `it` this wave has issued stores(it-2) [NS],
    // the DMAs of tile it+1 [ND, if there is one] and stores(it-1) [NS]; two buffers: only stores(it-1).
    if (it < AHEAD) {
      wait_vmcnt<0>();
    } else if (AHEAD == 1) {
      wait_vmcnt<Cf::NS>();
    } else if (it + 1 < nmine) {
      wait_vmcnt<2 * Cf::NS + Cf::ND>();
    } else {
      wait_vmcnt<2 * Cf::NS>();
    }
    __builtin_amdgcn_s_barrier();
    if (it + AHEAD < nmine) stage(tile + AHEAD * step, (it + AHEAD) % NBUF);

    const int img = tile / p.tiles_per_img, t2 = tile - img * p.tiles_per_img;
    const int ty = t2 / p.tiles_x, tx = t2 - ty * p.tiles_x;
    const int oy = ty * 4 + wave, ox0 = tx * 16;
    const char* tb = tlds + buf * (Cf::SLOTS * Cf::PXB);

    f32x4 acc[NC / 16];
#pragma unroll
    for (int f = 0; f < NC / 16; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int q = (wave * S + dy) * Cf::WIN_W + l15 * S + dx;  // this lane's input pixel for the tap
        const int tap = dy * 3 + dx;
#pragma unroll
        for (int kh = 0; kh < KC / 32; ++kh) {
          const f16x8 xv = *(const f16x8*)(tb + q * Cf::PXB + (((kh * 4 + lq) ^ s3_key<Cf::CH>(q)) * 16));
#pragma unroll
          for (int f = 0; f < NC / 16; ++f) {
            const int row = tap * NC + f * 16 + l15;
            const f16x8 wv = *(const f16x8*)(wlds + row * Cf::PXB + (((kh * 4 + lq) ^ s3_key<Cf::CH>(row)) * 16));
            acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, acc[f], 0, 0, 0);
          }
        }
      }
    od_mfma_results_ready();
#pragma unroll
    for (int s = 0; s < Cf::NS; ++s) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = acc[2 * s][e], b = acc[2 * s + 1][e];
        od_permlane16_swap(a, b);
        o[e] = a;
        o[4 + e] = b;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] * sc[s][e] + bi[s][e];
      if (p.act == OD_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = od_leaky(o[e], p.alpha);
      } else if (p.act == OD_ACT_ELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] > 0.f ? o[e] : p.alpha * od_expm1_fast(o[e]);
      }
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (f16)o[e];
      *(f16x8*)(p.out + ((((long long)img * p.Ho + oy) * p.Wo + ox0 + l15) * NC + s * 32 + n8)) = h;
    }
  }
}

struct S3Entry {
  int kc, nc, stride;
  const void* fn;
  const char* name;
  size_t lds;
};
const S3Entry g_s3[] = {
    {64, 32, 1, (const void*)&od_conv_stream3<64, 32, 1>, "od_conv_stream3<64, 32, 1>", (size_t)S3Cfg<64, 32, 1>::LDS},
    {32, 64, 1, (const void*)&od_conv_stream3<32, 64, 1>, "od_conv_stream3<32, 64, 1>", (size_t)S3Cfg<32, 64, 1>::LDS},
    {32, 64, 2, (const void*)&od_conv_stream3<32, 64, 2>, "od_conv_stream3<32, 64, 2>", (size_t)S3Cfg<32, 64, 2>::LDS},
};

const S3Entry* s3_find(const od_conv_desc* d) {
  for (const S3Entry& e : g_s3)
    if (e.kc == d->Cin && e.nc == d->Cout && e.stride == d->stride) return &e;
  return nullptr;
}

}  // namespace

bool od_conv_stream3_supported(const od_conv_desc* d) {
  if (d->transposed || d->ksize != 3 || d->res_mode != OD_RES_NONE || d->out_dtype != OD_DT_F16 || d->out_batch_stride != 0 ||
      d->out_pix_stride != 0 || d->bn_partials || d->w2 || d->H % d->stride || d->W % d->stride)
    return false;
  const int Ho = d->H / d->stride, Wo = d->W / d->stride;
  return s3_find(d) != nullptr && Ho % 4 == 0 && Wo % 16 == 0 && (long long)d->B * d->H * d->W * d->Cin < (1LL << 31);
}

int od_conv_stream3_launch(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run) {
  const S3Entry* e = s3_find(d);
  if (!e) return OD_ERR_INVALID;
  if (kernel_name) *kernel_name = e->name;
  if (dry_run) return OD_OK;
  Stream3KP p;
  p.x = (const f16*)d->x;
  p.w = (const f16*)d->w;
  p.scale = d->scale;
  p.bias = d->bias;
  p.out = (f16*)d->out;
  p.zero = (const f16*)ctx->zero_page;
  p.H = d->H;
  p.W = d->W;
  p.Ho = d->H / d->stride;
  p.Wo = d->W / d->stride;
  p.Kstride = od_round_up(9 * d->Cin, 64);
  p.act = d->act;
  p.alpha = d->alpha;
  p.tiles_x = p.Wo / 16;
  p.tiles_per_img = (p.Ho / 4) * p.tiles_x;
  p.ntiles = d->B * p.tiles_per_img;
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int grid = 2 * cus;
  if (grid > p.ntiles) grid = p.ntiles;
  if (int rc = od_ensure_lds(ctx, e->fn, e->lds)) return rc;
  void* args[] = {&p};
  OD_CHECK_HIP(hipLaunchKernel(e->fn, dim3(grid), dim3(256), args, e->lds, stream));
  return OD_OK;
}
