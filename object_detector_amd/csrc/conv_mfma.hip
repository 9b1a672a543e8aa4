// K1/K2: im2col-free implicit-GEMM convolution (3x3 / 1x1, stride 1 / 2, NHWC f16, f32 accumulate) on CDNA4 MFMA.
//
//   C[n, m] = sum_k W[n, k] * X[m, k]      m = (b, ho, wo) output pixel, n = output channel,
//                                          k = (dy*KS + dx)*Cin + cin  (never materialised)
//
// One 256-thread workgroup (4 waves) owns a BM x BN output tile.  Per 64-deep K step it gathers the BM x 64
// activation slice and the BN x 64 weight slice straight into LDS with global_load_lds_dwordx4 (LDS-DMA, 16 B per
// lane, per-lane SOURCE address = the shifted input pixel, or the context's zero page for padding / tails), double
// buffered so the DMA of step k+1 overlaps the MFMAs of step k.  LDS rows are 128 B; 16-B chunk c of row r lives at
// chunk c ^ (r & 7) (swizzle applied on the source side of the DMA and on the ds_read_b128 side), which makes the
// fragment reads bank-conflict free.  v_mfma_f32_16x16x32_f16 runs with the WEIGHTS as the A operand so that each
// lane ends up with 4 consecutive output channels of one pixel; the epilogue applies scale/bias/activation in f32,
// stages the tile through LDS and writes full NHWC lines (16 B per lane) with the residual added in f32 and ONE
// rounding to f16.
//
// Replaces the Conv2D + BatchNormalization + LeakyReLU/ELU (+ Add) layers executed inside
// `ObjectDetector.predict` (reference voc_validate.py:27; docs/MODEL.md:5-21).
#include "common.h"

namespace {

constexpr int BK = 64;        // K elements per step
constexpr int ROW_BYTES = 128;  // BK * sizeof(f16)

struct ConvKP {
  const f16* x;
  const f16* w;
  const float* scale;
  const float* bias;
  const f16* res;
  void* out;
  const f16* zero;
  int H, W, Cin, Ho, Wo, Cout;
  int stride, pad;
  int Kpad, Ktot, M, HoWo;
  int act;
  float alpha;
  int res_mode, out_f32, cin64;
  long long obs, ops;
  int mtiles, ntiles;
};

__device__ __forceinline__ void glds16(const void* gptr, void* lptr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lptr, 16, 0, 0);
}

__device__ __forceinline__ float od_act(float v, int act, float alpha) {
  if (act == OD_ACT_LEAKY) return v > 0.f ? v : v * alpha;
  if (act == OD_ACT_ELU) return v > 0.f ? v : alpha * expm1f(v);
  return v;
}

template <int BM, int BN, int WM, int WN, int KS>
__global__ __launch_bounds__(256, 2) void od_conv_igemm(ConvKP p) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MT = WTM / 16, NT = WTN / 16;
  constexpr int AR = BM / 32, BR = BN / 32;  // glds rounds (32 rows of 128 B per round per workgroup)
  constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int SLD = BN + 4;  // epilogue staging row stride in floats

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of logical tiles,
  // n fastest, so the tiles that re-read the same activation rows / halos hit the same L2.
  int logical;
  {
    const int nt = p.mtiles * p.ntiles;
    const int pid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, loc = pid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = logical / p.ntiles, tn = logical - tm * p.ntiles;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- per-lane gather state --------------------------------------------------------------------------------
  const int rr = tid >> 3;              // row within a 32-row round
  const int lc = (tid & 7) ^ (rr & 7);  // logical 16-B chunk this lane fetches (source-side swizzle)
  int a_base[AR], a_hi0[AR], a_wi0[AR];
#pragma unroll
  for (int rd = 0; rd < AR; ++rd) {
    const int m = m0 + rd * 32 + rr;
    if (m < p.M) {
      const unsigned b = (unsigned)m / (unsigned)p.HoWo;
      const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
      const unsigned ho = pix / (unsigned)p.Wo;
      const unsigned wo = pix - ho * (unsigned)p.Wo;
      a_hi0[rd] = (int)ho * p.stride - p.pad;
      a_wi0[rd] = (int)wo * p.stride - p.pad;
      a_base[rd] = (((int)b * p.H + a_hi0[rd]) * p.W + a_wi0[rd]) * p.Cin;
    } else {
      a_hi0[rd] = -(1 << 24);
      a_wi0[rd] = 0;
      a_base[rd] = 0;
    }
  }
  const f16* wrow = p.w + (long long)(n0 + rr) * p.Kpad + lc * 8;

  auto stage = [&](int ks, int buf) {
    char* abuf = smem + buf * STAGE_BYTES;
    char* bbuf = abuf + A_BYTES;
    const int k0 = ks * BK;
    int dy = 0, dx = 0, cin, kvalid;
    if (KS == 1) {
      cin = k0 + lc * 8;
      kvalid = cin < p.Cin;
    } else if (p.cin64) {
      const int tap = k0 / p.Cin;  // wave-uniform: a 64-deep step never straddles a tap
      dy = tap / 3;
      dx = tap - dy * 3;
      cin = k0 - tap * p.Cin + lc * 8;
      kvalid = 1;
    } else {
      const int k = k0 + lc * 8;
      const int tap = k / p.Cin;
      dy = tap / 3;
      dx = tap - dy * 3;
      cin = k - tap * p.Cin;
      kvalid = k < p.Ktot;
    }
    const int koff = (dy * p.W + dx) * p.Cin + cin;
#pragma unroll
    for (int rd = 0; rd < AR; ++rd) {
      const int hi = a_hi0[rd] + dy, wi = a_wi0[rd] + dx;
      const bool ok = kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const f16* src = ok ? p.x + (a_base[rd] + koff) : p.zero;
      glds16(src, abuf + (rd * 32 + wave * 8) * ROW_BYTES);
    }
#pragma unroll
    for (int rd = 0; rd < BR; ++rd) {
      glds16(wrow + (long long)rd * 32 * p.Kpad + k0, bbuf + (rd * 32 + wave * 8) * ROW_BYTES);
    }
  };

  // ---- main loop ----------------------------------------------------------------------------------------------
  const int wm = wave / WN, wn = wave - wm * WN;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.Kpad / BK;
  stage(0, 0);
  __syncthreads();  // emits vmcnt(0): step-0 DMA landed
  const int swz = l15 & 7;
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) stage(ks + 1, buf ^ 1);
    const char* abuf = smem + buf * STAGE_BYTES;
    const char* bbuf = abuf + A_BYTES;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int coff = ((kh * 4 + lq) ^ swz) * 16;
      f16x8 xa[MT], wb[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        xa[i] = *(const f16x8*)(abuf + (wm * WTM + i * 16 + l15) * ROW_BYTES + coff);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        wb[j] = *(const f16x8*)(bbuf + (wn * WTN + j * 16 + l15) * ROW_BYTES + coff);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();  // next buffer landed (vmcnt(0)) and everyone is done reading this one
  }

  // ---- epilogue: scale/bias/act in f32 -> LDS staging -> full-line stores (+ residual) -------------------------
  float* stg = (float*)smem;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nl = wn * WTN + j * 16 + lq * 4;
    const f32x4 sc = *(const f32x4*)(p.scale + n0 + nl);
    const f32x4 bi = *(const f32x4*)(p.bias + n0 + nl);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int ml = wm * WTM + i * 16 + l15;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = od_act(acc[i][j][e] * sc[e] + bi[e], p.act, p.alpha);
      *(f32x4*)(stg + ml * SLD + nl) = v;
    }
  }
  __syncthreads();

  constexpr int CH = BN / 8;     // 8-channel chunks per row
  constexpr int RPP = 256 / CH;  // rows per pass
  const int c8 = (tid % CH) * 8;
  const int n = n0 + c8;
#pragma unroll
  for (int ps = 0; ps < BM / RPP; ++ps) {
    const int row = ps * RPP + tid / CH;
    const int m = m0 + row;
    if (m < p.M && n < p.Cout) {
      const f32x4 v0 = *(const f32x4*)(stg + row * SLD + c8);
      const f32x4 v1 = *(const f32x4*)(stg + row * SLD + c8 + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const unsigned b = (unsigned)m / (unsigned)p.HoWo;
      const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
      if (p.res_mode != OD_RES_NONE) {
        long long roff;
        if (p.res_mode == OD_RES_SAME) {
          roff = (long long)m * p.Cout + n;
        } else {
          const unsigned ho = pix / (unsigned)p.Wo, wo = pix - ho * (unsigned)p.Wo;
          roff = ((long long)(b * (unsigned)(p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + n;
        }
        const f16x8 r = *(const f16x8*)(p.res + roff);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
      }
      const long long ooff = (long long)b * p.obs + (long long)pix * p.ops + n;
      if (p.out_f32) {
        float* o = (float*)p.out + ooff;
        *(f32x4*)o = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)v[e];
        *(f16x8*)((f16*)p.out + ooff) = h;
      }
    }
  }
}

struct TileCfg {
  int BM, BN;
  const void* k1;  // KS == 1
  const void* k3;  // KS == 3
  const char* name1;
  const char* name3;
};

#define OD_CFG(BM, BN, WM, WN)                                                         \
  {                                                                                    \
    BM, BN, (const void*)&od_conv_igemm<BM, BN, WM, WN, 1>,                            \
        (const void*)&od_conv_igemm<BM, BN, WM, WN, 3>,                                \
        "od_conv_igemm<" #BM "," #BN "," #WM "," #WN ",1>",                            \
        "od_conv_igemm<" #BM "," #BN "," #WM "," #WN ",3>"                             \
  }

const TileCfg g_cfgs[] = {
    OD_CFG(128, 128, 2, 2),
    OD_CFG(128, 64, 2, 2),
    OD_CFG(64, 128, 2, 2),
    OD_CFG(64, 64, 2, 2),
};
constexpr int kNumCfgs = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

size_t cfg_lds_bytes(const TileCfg& c) {
  const size_t pipe = 2u * (size_t)(c.BM + c.BN) * ROW_BYTES;
  const size_t stg = (size_t)c.BM * (c.BN + 4) * sizeof(float);
  return pipe > stg ? pipe : stg;
}

int pick_cfg(const od_ctx* ctx, int M, int Cout) {
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  if (Cout <= 64) {
    return (od_ceil_div(M, 128) >= 2 * cus) ? 1 : 3;
  }
  const long t128 = (long)od_ceil_div(M, 128) * od_ceil_div(Cout, 128);
  if (t128 >= 2L * cus) return 0;
  const long t64 = (long)od_ceil_div(M, 64) * od_ceil_div(Cout, 128);
  if (t64 >= 2L * cus) return 2;
  return 3;
}

}  // namespace

extern "C" int od_conv_num_tile_cfgs(void) { return kNumCfgs; }

extern "C" int od_conv_weight_dims(int cout, int cin, int ksize, int* cout_pad, int* kpad) {
  OD_REQUIRE(cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "od_conv_weight_dims: bad dims");
  if (cout_pad) *cout_pad = od_round_up(cout, 128);
  if (kpad) *kpad = od_round_up(ksize * ksize * cin, BK);
  return OD_OK;
}

int od_conv2d_fwd_impl(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name,
                       bool dry_run) {
  OD_REQUIRE(ctx && d, "od_conv2d_fwd: null ctx/desc");
  OD_REQUIRE(d->x && d->w && d->scale && d->bias && d->out, "od_conv2d_fwd: null tensor");
  OD_REQUIRE(d->ksize == 1 || d->ksize == 3, "od_conv2d_fwd: ksize %d unsupported", d->ksize);
  OD_REQUIRE(d->stride == 1 || d->stride == 2, "od_conv2d_fwd: stride %d unsupported", d->stride);
  OD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "od_conv2d_fwd: bad dims");
  OD_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0, "od_conv2d_fwd: Cin/Cout must be multiples of 8 (got %d/%d)",
             d->Cin, d->Cout);
  OD_REQUIRE(d->res_mode == OD_RES_NONE || d->res, "od_conv2d_fwd: res_mode set but res is null");
  OD_REQUIRE(d->act >= OD_ACT_LINEAR && d->act <= OD_ACT_ELU, "od_conv2d_fwd: bad act");
  const int pad = d->ksize / 2;
  const int Ho = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  const int Wo = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  if (d->res_mode == OD_RES_UP2)
    OD_REQUIRE(Ho % 2 == 0 && Wo % 2 == 0, "od_conv2d_fwd: OD_RES_UP2 needs even output size");
  const long long M64 = (long long)d->B * Ho * Wo;
  OD_REQUIRE(M64 * d->Cout < (1LL << 31) && (long long)d->B * d->H * d->W * d->Cin < (1LL << 31),
             "od_conv2d_fwd: tensor too large for 32-bit element offsets");
  const int M = (int)M64;

  int cfg = d->tile_cfg;
  if (cfg < 0) cfg = pick_cfg(ctx, M, d->Cout);
  OD_REQUIRE(cfg < kNumCfgs, "od_conv2d_fwd: tile_cfg %d out of range", cfg);
  const TileCfg& tc = g_cfgs[cfg];

  ConvKP p;
  p.x = (const f16*)d->x;
  p.w = (const f16*)d->w;
  p.scale = d->scale;
  p.bias = d->bias;
  p.res = (const f16*)d->res;
  p.out = d->out;
  p.zero = (const f16*)ctx->zero_page;
  p.H = d->H;
  p.W = d->W;
  p.Cin = d->Cin;
  p.Ho = Ho;
  p.Wo = Wo;
  p.Cout = d->Cout;
  p.stride = d->stride;
  p.pad = pad;
  p.Ktot = d->ksize * d->ksize * d->Cin;
  p.Kpad = od_round_up(p.Ktot, BK);
  p.M = M;
  p.HoWo = Ho * Wo;
  p.act = d->act;
  p.alpha = d->alpha;
  p.res_mode = d->res_mode;
  p.out_f32 = d->out_dtype == OD_DT_F32;
  p.cin64 = (d->Cin % 64 == 0);
  p.obs = d->out_batch_stride ? d->out_batch_stride : (long long)p.HoWo * d->Cout;
  p.ops = d->out_pix_stride ? d->out_pix_stride : d->Cout;
  p.mtiles = od_ceil_div(M, tc.BM);
  p.ntiles = od_ceil_div(d->Cout, tc.BN);
  // weights/scale/bias are padded to a multiple of 128 output channels, so any BN <= 128 tile stays in bounds.

  if (kernel_name) *kernel_name = d->ksize == 1 ? tc.name1 : tc.name3;
  if (dry_run) return OD_OK;

  const void* fn = d->ksize == 1 ? tc.k1 : tc.k3;
  const size_t lds = cfg_lds_bytes(tc);
  static bool attr_done[kNumCfgs][2] = {};
  if (!attr_done[cfg][d->ksize == 3]) {
    OD_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done[cfg][d->ksize == 3] = true;
  }
  void* args[] = {&p};
  OD_CHECK_HIP(hipLaunchKernel(fn, dim3(p.mtiles * p.ntiles), dim3(256), args, lds, stream));
  return OD_OK;
}

extern "C" int od_conv2d_fwd(od_ctx* ctx, const od_conv_desc* d, void* stream) {
  return od_conv2d_fwd_impl(ctx, d, (hipStream_t)stream, nullptr, false);
}
