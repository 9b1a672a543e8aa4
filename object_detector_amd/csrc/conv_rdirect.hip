// K1r: 3x3 stride-2 convolution 64 -> 128 channels on a large map (`b.down2`: 160x160 -> 80x80 at 320^2) with
// ALL weights resident in LDS and the pixel operand read straight from global memory into registers.
//
// The layer moves 157 MB for 30 GFLOP and is HBM-bound; on the table kernels it costs 76-79 us against 35 us of traffic
// (short K: 9 K steps per tile, thousands of workgroups in their prologue).  The streaming kernels of
// conv_stream3.hip keep nine weight taps of a 32 <-> 64-channel layer (36 KiB) next to a window ring; here the nine taps are
// 147 KiB, so there is no room for a window -- and none is needed: a lane's MFMA operand for tap (dy, dx) is 16 contiguous
// bytes of ONE input pixel (8 channels), i.e. one global_load_dwordx4; the 2.25 (stride 2) or 9 (stride 1) loads that touch
// the same pixel hit L1 / L2.  One workgroup per CU (8 waves, 147 KiB of weights loaded once), every wave walks its own
// 16-pixel row segments (item = wave index + k x total waves) with the next item's 18 loads in flight, 144 MFMAs per item
// (weights from LDS, conflict-free ds_read_b128), no barrier after the weights have landed; epilogue = scale / bias /
// activation from the accumulators, 16-byte stores (v_permlane16_swap pairing, as conv_8ph.hip).
//
// Replaces Conv2D + BatchNormalization + LeakyReLU of `b.down2` inside `ObjectDetector.predict` (reference
// voc_validate.py:27; docs/MODEL.md:15-17) and the same layers' forward in training.
#include <stdlib.h>

#include "conv_common.h"

namespace {

struct RdKP {
  const f16* x;      // [B, H, W, KC]
  const f16* w;      // packed [256][9*KC]: row n, k = tap*KC + c
  const float* scale;
  const float* bias;
  f16* out;          // [B, Ho, Wo, NC]
  const f16* res;    // residual [B, Ho, Wo, NC] added after the activation (OD_RES_SAME), or null
  int H, W, Ho, Wo, Kstride;
  int act;
  float alpha;
  int segs_per_row, nitems;
};

// 16-byte chunk swizzle of a weight row: 128-B and longer rows XOR the low 3 chunk bits with the row's low 3 bits, 64-B rows
// (4 chunks) with bits 1-2 of the row; every fragment read (16 consecutive rows, one chunk column) is then conflict-free
template <int CH>
__device__ __forceinline__ int rd_swz(int chunk, int row) {
  return CH >= 8 ? ((chunk & ~7) | ((chunk ^ row) & 7)) : (chunk ^ ((row >> 1) & 3));
}

// KS = 3: the 3x3 layer described above.  KS = 1: the pointwise layers of the 160x160 / 80x80 maps in training (weights
// <= 16 KiB, a pixel is loaded exactly once): pure streams that the table kernels run at 60-75 % of what HBM gives.
template <int KC, int NC, int S, int KS>
__global__ __launch_bounds__(512, 2) void od_conv_rdirect(RdKP p) {
  constexpr int ROWB = KC * 2;             // bytes per weight row (one tap, one output channel)
  constexpr int CH = ROWB / 16;            // 16-byte chunks per row
  constexpr int KH = KC / 32;              // MFMA k steps per tap
  constexpr int NF = NC / 16;              // output-channel fragments
  constexpr int TAPS = KS * KS, PAD = KS / 2;
  constexpr int WROWS = TAPS * NC;
  extern __shared__ __attribute__((aligned(16))) char wlds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  // ---- all weights -> LDS: row r = tap*NC + n, chunk c stored at c ^ (r & 7) (low 3 bits) -----------------------------
  {
    constexpr int LPR = CH;                 // lanes per row in a 1-KiB piece
    constexpr int RPP = 64 / LPR;           // rows per piece
    constexpr int PIECES = (WROWS + RPP - 1) / RPP;  // 144 (3x3, KC = 64, NC = 128)
    for (int q = wave; q < PIECES; q += 8) {
      int r = q * RPP + lane / LPR;
      if (r >= WROWS) r = WROWS - 1;  // (a last, partial piece re-writes the last row with itself)
      const int tap = r / NC, n = r - tap * NC;
      const int lc = rd_swz<CH>(lane % LPR, r);
      glds16(p.w + ((long long)n * p.Kstride + tap * KC + lc * 8), wlds + q * 1024);
    }
  }

  // this lane's operand loads of one item: pixel (oy*S + dy - 1, (ox0 + l15)*S + dx - 1), channels kh*32 + lq*8 .. +7
  auto load_item = [&](int item, f16x8 (&xs)[TAPS][KH]) {
    const int row = item / p.segs_per_row, ox0 = (item - row * p.segs_per_row) * 16;
    const int b = row / p.Ho, oy = row - b * p.Ho;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) {
      const int iy = oy * S + dy - PAD;
      const bool yok = (unsigned)iy < (unsigned)p.H;
      const f16* rowp = p.x + ((long long)(b * p.H + iy) * p.W) * KC + lq * 8;
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const int ix = (ox0 + l15) * S + dx - PAD;
        const bool ok = yok && (unsigned)ix < (unsigned)p.W;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
          if (ok) v = *(const f16x8*)(rowp + (long long)ix * KC + kh * 32);
          xs[dy * KS + dx][kh] = v;
        }
      }
    }
  };

  const int n8 = (lq & 1) * 16 + (lq >> 1) * 8;  // this lane's 8 channels inside a 32-channel store group
  const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  f16x8 xa[TAPS][KH], xb[TAPS][KH];
  int item = gw;
  if (item < p.nitems) load_item(item, xa);
  // scale / bias -> LDS behind the weights (read per item; 64 more registers would not fit beside two operand sets)
  float* const sb = (float*)(wlds + WROWS * ROWB);
  if (tid < NC) {
    sb[tid] = p.scale[tid];
    sb[NC + tid] = p.bias[tid];
  }
  wait_vmcnt<0>();  // every weight piece has landed (a counted wait cannot be used: border items issue fewer operand loads)
  __syncthreads();

  // weight fragment address: row tap*NC + f*16 + l15 (its swizzle key depends on l15 only: NC and 16 are multiples of 16),
  // chunk kh*4 + lq
  int fwk[KH];
#pragma unroll
  for (int kh = 0; kh < KH; ++kh) fwk[kh] = l15 * ROWB + rd_swz<CH>(kh * 4 + lq, l15) * 16;

  auto compute_store = [&](int it, f16x8 (&xs)[TAPS][KH]) {
    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
      for (int kh = 0; kh < KH; ++kh)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const f16x8 wv = *(const f16x8*)(wlds + (tap * NC + f * 16) * ROWB + fwk[kh]);
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xs[tap][kh], acc[f], 0, 0, 0);
        }
    od_mfma_results_ready();
    const int row = it / p.segs_per_row, ox0 = (it - row * p.segs_per_row) * 16;
    const long long ooff = ((long long)row * p.Wo + ox0 + l15) * NC + n8;
    f16* orow = p.out + ooff;
    f16x8 rv[NC / 32];
    if (p.res) {
#pragma unroll
      for (int s = 0; s < NC / 32; ++s) rv[s] = *(const f16x8*)(p.res + ooff + s * 32);
    }
#pragma unroll
    for (int s = 0; s < NC / 32; ++s) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = acc[2 * s][e], b = acc[2 * s + 1][e];
        od_permlane16_swap(a, b);
        o[e] = a;
        o[4 + e] = b;
      }
      const f32x4 s0 = *(const f32x4*)(sb + s * 32 + n8), s1 = *(const f32x4*)(sb + s * 32 + n8 + 4);
      const f32x4 b0 = *(const f32x4*)(sb + NC + s * 32 + n8), b1 = *(const f32x4*)(sb + NC + s * 32 + n8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = o[e] * s0[e] + b0[e];
        o[4 + e] = o[4 + e] * s1[e] + b1[e];
      }
      if (p.act == OD_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = od_leaky(o[e], p.alpha);
      } else if (p.act == OD_ACT_ELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = o[e] > 0.f ? o[e] : p.alpha * od_expm1_fast(o[e]);
      }
      if (p.res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += (float)rv[s][e];
      }
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (f16)o[e];
      *(f16x8*)(orow + s * 32) = h;
    }
  };

  // two items per trip so that the operand sets stay in fixed registers (xa: even items, xb: odd items)
  while (item < p.nitems) {
    const int n1 = item + nw;
    if (n1 < p.nitems) load_item(n1, xb);
    compute_store(item, xa);
    if (n1 >= p.nitems) break;
    const int n2 = n1 + nw;
    if (n2 < p.nitems) load_item(n2, xa);
    compute_store(n1, xb);
    item = n2;
  }
}

// Transposed form: backward-data of `b.down2` (dZ [B, Hs, Ws, 128] -> dX [B, 2Hs, 2Ws, 64]; the packed backward weights, 147 KiB,
// resident).  An item = 16 dZ pixels of one row; the lane's four source pixels (i + {0,1}, j + {0,1}) x 4 k steps are 16
// direct loads; the four output parity classes take 1 / 2 / 2 / 4 of the nine taps (conv_tconv.hip has the derivation):
// 144 MFMAs, eight 16-byte stores per lane.  The generic transposed path runs this layer in 122 us (157 MB of traffic).
template <int KC, int NC>
__global__ __launch_bounds__(512, 2) void od_tconv_rdirect(RdKP p) {
  constexpr int ROWB = KC * 2, CH = ROWB / 16, KH = KC / 32, NF = NC / 16, WROWS = 9 * NC;
  extern __shared__ __attribute__((aligned(16))) char wlds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  {
    constexpr int LPR = CH, RPP = 64 / LPR, PIECES = WROWS / RPP;
    for (int q = wave; q < PIECES; q += 8) {
      const int r = q * RPP + lane / LPR;
      const int tap = r / NC, n = r - tap * NC;
      const int lc = rd_swz<CH>(lane % LPR, r);
      glds16(p.w + ((long long)n * p.Kstride + tap * KC + lc * 8), wlds + q * 1024);
    }
  }
  // here p.H / p.W are the dZ map (Hs, Ws), p.Ho / p.Wo the output map (2Hs, 2Ws); an item = (dZ row, 16-pixel segment)
  auto load_item = [&](int item, f16x8 (&xs)[4][KH]) {
    const int row = item / p.segs_per_row, j0 = (item - row * p.segs_per_row) * 16;
    const int b = row / p.H, i = row - b * p.H;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int u = i + dy, v = j0 + l15 + dx;
        const bool ok = u < p.H && v < p.W;
        const f16* src = p.x + (((long long)(b * p.H + u) * p.W + v) * KC + lq * 8);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          f16x8 val = {0, 0, 0, 0, 0, 0, 0, 0};
          if (ok) val = *(const f16x8*)(src + kh * 32);
          xs[dy * 2 + dx][kh] = val;
        }
      }
  };
  const int n8 = (lq & 1) * 16 + (lq >> 1) * 8;
  const int gw = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  f16x8 xa[4][KH], xb[4][KH];
  int item = gw;
  if (item < p.nitems) load_item(item, xa);
  float* const sb = (float*)(wlds + WROWS * ROWB);
  if (tid < NC) {
    sb[tid] = p.scale[tid];
    sb[NC + tid] = p.bias[tid];
  }
  wait_vmcnt<0>();
  __syncthreads();
  int fwk[KH];
#pragma unroll
  for (int kh = 0; kh < KH; ++kh) fwk[kh] = l15 * ROWB + rd_swz<CH>(kh * 4 + lq, l15) * 16;

  auto compute_store = [&](int it, f16x8 (&xs)[4][KH]) {
    const int row = it / p.segs_per_row, j0 = (it - row * p.segs_per_row) * 16;
    const int b = row / p.H, i = row - b * p.H;
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {  // one parity class at a time: 16 accumulator registers live
      const int py = cls >> 1, px = cls & 1;
      f32x4 acc[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          if ((ty == 1 ? 0 : 1) != py || (tx == 1 ? 0 : 1) != px) continue;
          const int src = (ty == 2 ? 2 : 0) + (tx == 2 ? 1 : 0), tap = ty * 3 + tx;
#pragma unroll
          for (int kh = 0; kh < KH; ++kh)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const f16x8 wv = *(const f16x8*)(wlds + (tap * NC + f * 16) * ROWB + fwk[kh]);
              acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xs[src][kh], acc[f], 0, 0, 0);
            }
        }
      od_mfma_results_ready();
      const long long ooff = (((long long)b * p.Ho + 2 * i + py) * p.Wo + 2 * (j0 + l15) + px) * NC + n8;
#pragma unroll
      for (int s = 0; s < NC / 32; ++s) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[2 * s][e], bq = acc[2 * s + 1][e];
          od_permlane16_swap(a, bq);
          o[e] = a;
          o[4 + e] = bq;
        }
        const f32x4 s0 = *(const f32x4*)(sb + s * 32 + n8), s1 = *(const f32x4*)(sb + s * 32 + n8 + 4);
        const f32x4 b0 = *(const f32x4*)(sb + NC + s * 32 + n8), b1 = *(const f32x4*)(sb + NC + s * 32 + n8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = o[e] * s0[e] + b0[e];
          o[4 + e] = o[4 + e] * s1[e] + b1[e];
        }
        if (p.act == OD_ACT_LEAKY) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = od_leaky(o[e], p.alpha);
        } else if (p.act == OD_ACT_ELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = o[e] > 0.f ? o[e] : p.alpha * od_expm1_fast(o[e]);
        }
        if (p.res) {
          const f16x8 rv = *(const f16x8*)(p.res + ooff + s * 32);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += (float)rv[e];
        }
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)o[e];
        *(f16x8*)(p.out + ooff + s * 32) = h;
      }
    }
  };
  while (item < p.nitems) {
    const int n1 = item + nw;
    if (n1 < p.nitems) load_item(n1, xb);
    compute_store(item, xa);
    if (n1 >= p.nitems) break;
    const int n2 = n1 + nw;
    if (n2 < p.nitems) load_item(n2, xa);
    compute_store(n1, xb);
    item = n2;
  }
}

struct RdEntry {
  int kc, nc, stride, ks;
  const void* fn;
  const char* name;
  size_t lds;
};
#define OD_RD1(KC, NC) \
  { KC, NC, 1, 1, (const void*)&od_conv_rdirect<KC, NC, 1, 1>, "od_conv_rdirect<" #KC ", " #NC ", 1, 1>", (size_t)NC * KC * 2 + 1024 }
const RdEntry g_rd[] = {
    {64, 128, 2, 3, (const void*)&od_conv_rdirect<64, 128, 2, 3>, "od_conv_rdirect<64, 128, 2, 3>", (size_t)9 * 128 * 128 + 1024},
    OD_RD1(64, 32), OD_RD1(32, 64), OD_RD1(128, 64), OD_RD1(64, 128),  // the pointwise layers of stages 1-2 (training)
    // (stride 1 -- b.s2.*.b in training -- is instantiable but not offered: there every pixel is loaded nine times and the
    //  kernel is L1-bound, 60-63 us against 58-59 us on the table kernel; stride 2 loads a pixel 2.25 times: 76 -> 54 us)
};
const RdEntry g_rdt = {128, 64, 2, 3, (const void*)&od_tconv_rdirect<128, 64>, "od_tconv_rdirect<128, 64>",
                       (size_t)9 * 64 * 256 + 1024};
const RdEntry* rd_find(const od_conv_desc* d) {
  if (d->transposed) return (d->Cin == 128 && d->Cout == 64 && d->ksize == 3 && d->stride == 2) ? &g_rdt : nullptr;
  for (const RdEntry& e : g_rd)
    if (e.kc == d->Cin && e.nc == d->Cout && e.stride == d->stride && e.ks == d->ksize) return &e;
  return nullptr;
}

}  // namespace

bool od_conv_rdirect_supported(const od_conv_desc* d) {
  if ((d->res_mode != OD_RES_NONE && d->res_mode != OD_RES_SAME) || d->out_dtype != OD_DT_F16 || d->out_batch_stride != 0 ||
      d->out_pix_stride != 0 || d->bn_partials || d->w2)
    return false;
  if (!d->transposed && (d->H % d->stride || d->W % d->stride)) return false;
  const int Wo = d->transposed ? d->W : d->W / d->stride;  // (transposed: items are 16-pixel segments of the dZ rows)
  if (rd_find(d) == nullptr || Wo % 16 != 0 || (long long)d->B * d->H * d->W * d->Cin >= (1LL << 31)) return false;
  // large maps only: the 147 KiB of weights are loaded once per workgroup (OD_CONV_RDIRECT_MIN_PIXELS overrides: tests)
  long long min_px = d->ksize == 3 ? (1 << 18) : (1 << 16);  // (the pointwise forms load at most 16 KiB of weights)
  if (const char* e = getenv("OD_CONV_RDIRECT_MIN_PIXELS")) min_px = atoll(e);
  return (long long)d->B * d->H * d->W * (d->transposed ? 4 : 1) >= min_px;  // (transposed: the output map is the large one)
}

int od_conv_rdirect_launch(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run) {
  const RdEntry* e = rd_find(d);
  if (!e) return OD_ERR_INVALID;
  if (kernel_name) *kernel_name = e->name;
  if (dry_run) return OD_OK;
  RdKP p;
  p.x = (const f16*)d->x;
  p.w = (const f16*)d->w;
  p.scale = d->scale;
  p.bias = d->bias;
  p.out = (f16*)d->out;
  p.res = d->res_mode == OD_RES_SAME ? (const f16*)d->res : nullptr;
  p.H = d->H;
  p.W = d->W;
  p.Ho = d->transposed ? 2 * d->H : d->H / d->stride;
  p.Wo = d->transposed ? 2 * d->W : d->W / d->stride;
  p.Kstride = od_round_up(d->ksize * d->ksize * d->Cin, 64);
  p.act = d->act;
  p.alpha = d->alpha;
  p.segs_per_row = (d->transposed ? d->W : p.Wo) / 16;
  p.nitems = d->B * (d->transposed ? d->H : p.Ho) * p.segs_per_row;
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  int grid = d->ksize == 1 ? 2 * cus : cus;  // (the 3x3 form holds 147 KiB of LDS: one workgroup per CU)
  if (grid * 8 > p.nitems) grid = od_ceil_div(p.nitems, 8);
  if (int rc = od_ensure_lds(ctx, e->fn, e->lds)) return rc;
  void* args[] = {&p};
  OD_CHECK_HIP(hipLaunchKernel(e->fn, dim3(grid), dim3(512), args, e->lds, stream));
  return OD_OK;
}
