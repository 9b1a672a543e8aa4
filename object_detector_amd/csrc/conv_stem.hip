// K3+K1 fused stem: the first two Darknet53 layers in ONE kernel,
//     out = act(bn(conv3x3/s2(act(bn(conv3x3/s1(x_u8))))))        x u8 [B,H,W,3] -> out f16 [B,H/2,W/2,64]
// The 32-channel full-resolution tensor between them (210 MB at batch 32 / 320x320: the largest tensor of the whole
// network, written by one launch and read back by the next) never leaves the CU.
//
// PERSISTENT workgroups (8 waves, one per CU) walk 16x16-pixel output tiles:
//   1. the 35x35 uint8 input window of the NEXT tile is fetched into registers while this tile computes, converted to
//      f16 ONCE per pixel and dropped into the other LDS buffer at the end of the tile ([row][pixel][4 x f16], 4th = 0,
//      zeros outside the image)
//   2. producer = od_conv_first's convolution on the 33x33 window of first-layer pixels with the K axis laid out as
//      k = tap*4 + channel: a lane's 8 k's are two whole window pixels = two 8-byte LDS reads, no per-element convert
//      or pack (the byte-window form spent 28 VALU instructions per fragment on that and made the kernel VALU-issue
//      bound); taps 0..7 fill one 32-deep MFMA, tap 8 a second one (4 of 32 k's used: the matrix pipe is idle here
//      anyway); scale/bias/activation, ONE rounding to f16, 16-byte LDS writes (weight rows permuted so that a lane holds
//      8 consecutive channels); pixels outside the image are written as 0 = the zero padding the stride-2 convolution
//      sees in the two-layer network
//   3. consumer = 3x3 stride 2 from that window.  The window is stored DE-INTERLEAVED (even and odd columns as two
//      planes per row) so that the 16 pixels x0 + 2*l + dx of a fragment are 16 consecutive 64-byte rows: the
//      chunk ^ 3*((row>>2)&1) swizzle is then conflict-free for every tap (a stride-2 walk over interleaved rows is
//      2-way conflicted under every XOR swizzle of this family).  Taps column by column: 9 A fragments serve 3 taps.
//      All nine weight taps stay resident in LDS.
//   4. epilogue from the accumulators (v_permlane16_swap -> 8 consecutive channels per lane, 16-byte stores).
// Same rounding points as od_conv_first_fwd + od_conv2d_fwd.
// Replaces the preprocess + first two Conv2D/BatchNormalization/LeakyReLU layers of `ObjectDetector.predict`
// (reference voc_validate.py:27; docs/MODEL.md:15-17).
#include "conv_common.h"

namespace {

struct StemKP {
  const uint8_t* x;
  const f16* w0;
  const float* s0;
  const float* b0;
  const f16* w3;
  const float* s3;
  const float* b3;
  f16* out;
  int B, H, W;  // input size; output is H/2 x W/2
  int k3stride;
  int act;
  float alpha;
  int tiles_x, tiles_y;
};

// Tile = 16 x 16 output pixels; every wave runs producer then consumer (a two-stage wave pipeline -- 4 producer waves
// feeding 4 consumer waves through two window buffers -- measured SLOWER, 157 vs 133 us: the kernel is VALU-issue bound
// (28 k VALU instructions per SIMD for the first layer alone, profiles/r01/conv_stem_analysis.txt), and one wave per
// SIMD cannot issue them faster than one per 4-5 cycles).
constexpr int S_UW = 35;                       // uint8 window edge
constexpr int S_UBYTES = 9856;                 // 35*35 pixels x 4 f16 = 9800, rounded up to 128
constexpr int S_TW = 33, S_TROW = 34;          // first-layer window edge; LDS rows per window row (17 even + 17 odd slots)
constexpr int S_NTP = S_TW * S_TW;             // 1089 producer pixels = 69 m-fragments (last one ragged)
constexpr int S_NTF = (S_NTP + 15) / 16;
constexpr int S_TBYTES = S_TW * S_TROW * 64;   // 71808
constexpr int S_W3TAP = 64 * 64;               // 64 output channels x 32 middle channels x 2 B
constexpr int S_OFF_U = 0, S_OFF_T = 2 * S_UBYTES, S_OFF_W3 = S_OFF_T + S_TBYTES;
constexpr int S_LDS = S_OFF_W3 + 9 * S_W3TAP;  // 118656
constexpr int S_UROUNDS = (S_UW * S_UW * 3 + 511) / 512;  // byte loads per thread per window

static __device__ __forceinline__ int s_swz(int row) { return 3 * ((row >> 2) & 1); }

// compile-time activation: see conv_bneck.hip bn_act (405 branches and 48 v_exp_f32 in the run-time-enum form of od_stem)
template <int ACT>
static __device__ __forceinline__ float s_act(float v, float alpha) {
  if (ACT == OD_ACT_LEAKY) return od_leaky(v, alpha);
  if (ACT == OD_ACT_ELU) return v > 0.f ? v : alpha * od_expm1_fast(v);
  return v;
}

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t s_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ void s_buffer_dma(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// DBG = 1 (OD_CONV_DEBUG=64): s_memtime stamps of the 6th tile of workgroup 0, waves 0 and 5 (od_debug_stem_stamps)
__device__ unsigned long long g_stem_stamps[2][8];
#define ST_STAMP(k)                                                                         \
  do {                                                                                      \
    if (DBG) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[k])::"memory"); \
  } while (0)

template <int DBG, int ACT = OD_ACT_LEAKY>
__global__ __launch_bounds__(512, 2) void od_stem_k(StemKP p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;

  int tile = blockIdx.x;
  if (tile >= ntiles) return;

  // ---- once per workgroup --------------------------------------------------------------------------------------
  // 3x3 weights: 9 taps x 64 rows x 64 B, chunk-swizzled, buffer-addressed LDS-DMA (whole waves drop out: 256 chunks)
  {
    const __amdgpu_buffer_rsrc_t rs_w3 = s_make_rsrc(p.w3, 256u * (unsigned)p.k3stride * 2u);
    const int n = tid >> 2, pc = tid & 3;
    const int voff = (n * p.k3stride + (pc ^ s_swz(n)) * 8) * 2;
    if (tid < 256) {
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) s_buffer_dma(rs_w3, voff, tap * 64, smem + S_OFF_W3 + tap * S_W3TAP + wave * 1024);
    }
  }
  // input windows: the 4th f16 of every pixel stays the zero that k = tap*4 + 3 of the A fragment reads
  for (int i = tid; i < 2 * S_UBYTES / 4; i += 512) ((int*)(smem + S_OFF_U))[i] = 0;

  // per-thread byte slots of a window: i = r*512 + tid -> (row, byte in row); row = 105 contiguous source bytes
  int u_src[S_UROUNDS], u_dst[S_UROUNDS];
#pragma unroll
  for (int r = 0; r < S_UROUNDS; ++r) {
    const int i = r * 512 + tid;
    const int uy = i / (S_UW * 3), bt = i - uy * (S_UW * 3);
    const int ux = bt / 3, c = bt - ux * 3;
    u_src[r] = i < S_UW * S_UW * 3 ? (uy << 16) | (ux << 8) | c : -1;
    u_dst[r] = (uy * S_UW + ux) * 8 + c * 2;
  }
  uint8_t ubuf[S_UROUNDS];
  auto fetch_u = [&](int t) {  // global -> registers
    const int b = t / tpi;
    const int trem = t - b * tpi;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int gy0 = tyi * 32 - 2, gx0 = txi * 32 - 2;
    const uint8_t* xb = p.x + (long long)b * p.H * p.W * 3;
#pragma unroll
    for (int r = 0; r < S_UROUNDS; ++r) {
      uint8_t v = 0;
      if (u_src[r] >= 0) {
        const int gy = gy0 + (u_src[r] >> 16), gx = gx0 + ((u_src[r] >> 8) & 255);
        if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = xb[((long long)gy * p.W + gx) * 3 + (u_src[r] & 255)];
      }
      ubuf[r] = v;
    }
  };
  auto store_u = [&](int buf) {  // registers -> LDS
#pragma unroll
    for (int r = 0; r < S_UROUNDS; ++r)
      if (u_src[r] >= 0) *(f16*)(smem + S_OFF_U + buf * S_UBYTES + u_dst[r]) = (f16)(float)ubuf[r];
  };

  // first-layer weights (A operand; row r of n-tile t = channel (r/4)*8 + t*4 + r%4, so a lane ends up with 8 consecutive
  // channels) re-gathered from the [32][tap*3 + c] pack into the k = tap*4 + c layout: wfa = taps 0..7, wfb = tap 8
  f16x8 wfa[2], wfb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ch = (l15 >> 2) * 8 + t * 4 + (l15 & 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tap = 2 * lq + (j >> 2), c = j & 3;
      wfa[t][j] = c < 3 ? p.w0[ch * 32 + tap * 3 + c] : (f16)0.f;
      wfb[t][j] = (lq == 0 && j < 3) ? p.w0[ch * 32 + 24 + j] : (f16)0.f;
    }
  }
  // byte offsets of this lane's two taps (and of tap 8) relative to its window pixel
  const int ko0 = (((2 * lq) / 3) * S_UW + (2 * lq) % 3) * 8;
  const int ko1 = (((2 * lq + 1) / 3) * S_UW + (2 * lq + 1) % 3) * 8;
  constexpr int ko8 = (2 * S_UW + 2) * 8;
  float sc0[8], bi0[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc0[e] = p.s0[lq * 8 + e];
    bi0[e] = p.b0[lq * 8 + e];
  }
  const int wm = wave >> 1, wn = wave & 1;
  int boff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (wn * 2 + j) * 16 + l15;
    boff[j] = row * 64 + ((lq ^ s_swz(row)) * 16);
  }

  __syncthreads();  // zero fill done before the first window bytes land
  fetch_u(tile);
  store_u(0);
  wait_vmcnt<0>();
  __syncthreads();

  int cur = 0;
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int tcount = 0;
#pragma unroll 1
  for (; tile < ntiles; tile += (int)gridDim.x) {
    ST_STAMP(0);
    const int b = tile / tpi;
    const int trem = tile - b * tpi;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int y0 = tyi * 16, x0 = txi * 16;  // output tile origin
    const char* uw = smem + S_OFF_U + cur * S_UBYTES;

    // ---- producer: the 33x33 window of first-layer pixels ----------------------------------------------------------
    // Two m-fragments per trip, written out by hand (the optimizer refuses to unroll this loop): the LDS reads and MFMAs of
    // the second fragment are in flight while the VALU epilogue of the first one issues.  With two waves per SIMD the
    // one-fragment form spent ~1.1 k cycles per fragment on ~80 VALU instructions: latency, not VALU throughput.
    typedef f16 f16x4 __attribute__((ext_vector_type(4)));
    auto p_load = [&](int f, int& wp, int& wy, int& wx, f16x8& xf, f16x8& xg) {
      wp = f * 16 + l15;
      const int wpc = wp < S_NTP ? wp : S_NTP - 1;
      wy = wpc / S_TW;
      wx = wpc - wy * S_TW;
      const char* pbase = uw + (wy * S_UW + wx) * 8;
      const f16x4 q0 = *(const f16x4*)(pbase + ko0), q1 = *(const f16x4*)(pbase + ko1);
      const f16x4 q8 = *(const f16x4*)(pbase + ko8);
      xf = f16x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      const f16 z = (f16)0.f;
      xg = lq == 0 ? f16x8{q8[0], q8[1], q8[2], q8[3], z, z, z, z} : f16x8{z, z, z, z, z, z, z, z};
    };
    auto p_mfma = [&](const f16x8& xf, const f16x8& xg, f32x4& a0, f32x4& a1) {
      a0 = f32x4{0.f, 0.f, 0.f, 0.f};
      a1 = f32x4{0.f, 0.f, 0.f, 0.f};
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfa[0], xf, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfa[1], xf, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfb[0], xg, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfb[1], xg, a1, 0, 0, 0);
    };
    auto p_store = [&](int wp, int wy, int wx, const f32x4& a0, const f32x4& a1) {
      if (wp < S_NTP) {
        const int iy = 2 * y0 - 1 + wy, ix = 2 * x0 - 1 + wx;
        const bool inside = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v0 = s_act<ACT>(a0[e] * sc0[e] + bi0[e], p.alpha);
          const float v1 = s_act<ACT>(a1[e] * sc0[4 + e] + bi0[4 + e], p.alpha);
          h[e] = inside ? (f16)v0 : (f16)0.f;
          h[4 + e] = inside ? (f16)v1 : (f16)0.f;
        }
        const int trow = wy * S_TROW + (wx & 1) * 17 + (wx >> 1);  // de-interleaved columns
        *(f16x8*)(smem + S_OFF_T + trow * 64 + ((lq ^ s_swz(trow)) * 16)) = h;
      }
    };
#pragma unroll 1
    for (int f = wave; f < S_NTF; f += 16) {
      int wpA, wyA, wxA, wpB, wyB, wxB;
      f16x8 xfA, xgA, xfB, xgB;
      f32x4 a0A, a1A, a0B, a1B;
      const bool hasB = f + 8 < S_NTF;  // wave-uniform
      p_load(f, wpA, wyA, wxA, xfA, xgA);
      if (hasB) p_load(f + 8, wpB, wyB, wxB, xfB, xgB);
      p_mfma(xfA, xgA, a0A, a1A);
      if (hasB) p_mfma(xfB, xgB, a0B, a1B);
      p_store(wpA, wyA, wxA, a0A, a1A);
      if (hasB) p_store(wpB, wyB, wxB, a0B, a1B);
    }
    ST_STAMP(1);
    __syncthreads();  // window complete; this tile's uint8 window is free
    ST_STAMP(2);

    const int tnext = tile + (int)gridDim.x;
    if (tnext < ntiles) fetch_u(tnext);  // global byte loads in flight during the consumer
    ST_STAMP(3);

    // ---- consumer: 3x3 stride 2 from the window ------------------------------------------------------------------
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      f16x8 xa[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int trow = (8 * wm + k) * S_TROW + (dx & 1) * 17 + (dx >> 1) + l15;
        xa[k] = *(const f16x8*)(smem + S_OFF_T + trow * 64 + ((lq ^ s_swz(trow)) * 16));
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        f16x8 wb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) wb[j] = *(const f16x8*)(smem + S_OFF_W3 + (dy * 3 + dx) * S_W3TAP + boff[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[2 * i + dy], acc[i][j], 0, 0, 0);
      }
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------
    ST_STAMP(4);
    od_mfma_results_ready();
    {
      const int ch = (wn * 2 + (lq & 1)) * 16 + (lq >> 1) * 8;
      float sc[8], bi[8];
      const f32x4 q0 = *(const f32x4*)(p.s3 + ch), q1 = *(const f32x4*)(p.s3 + ch + 4);
      const f32x4 r0 = *(const f32x4*)(p.b3 + ch), r1 = *(const f32x4*)(p.b3 + ch + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = q0[e];
        sc[4 + e] = q1[e];
        bi[e] = r0[e];
        bi[4 + e] = r1[e];
      }
      const int Ho = p.H >> 1, Wo = p.W >> 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ty = wm * 4 + i;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[i][0][e], bq = acc[i][1][e];
          od_permlane16_swap(a, bq);
          v[e] = a;
          v[4 + e] = bq;
        }
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)s_act<ACT>(v[e] * sc[e] + bi[e], p.alpha);
        *(f16x8*)(p.out + ((long long)(b * Ho + y0 + ty) * Wo + x0 + l15) * 64 + ch) = h;
      }
    }
    ST_STAMP(5);
    if (tnext < ntiles) store_u(cur ^ 1);  // waits for the byte loads issued before the consumer
    ST_STAMP(6);
    __syncthreads();                        // next window visible; everyone is done with the first-layer window
    ST_STAMP(7);
    cur ^= 1;
    if (DBG && blockIdx.x == 0 && (wave == 0 || wave == 5) && ++tcount == 6 && lane == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) g_stem_stamps[wave ? 1 : 0][k] = st[k];
    }
  }
}

}  // namespace

extern "C" int od_stem_supported(int H, int W) { return H > 0 && W > 0 && (H % 32) == 0 && (W % 32) == 0; }  // net input rule

const char* od_stem_kernel_name() { return "od_stem_k<0, 1>"; }

extern "C" int od_stem_fwd(od_ctx* ctx, const od_stem_desc* d, void* stream) {
  OD_REQUIRE(ctx && d, "od_stem_fwd: null ctx/desc");
  OD_REQUIRE(d->x && d->w0 && d->scale0 && d->bias0 && d->w3 && d->scale3 && d->bias3 && d->out, "od_stem_fwd: null tensor");
  OD_REQUIRE(od_stem_supported(d->H, d->W), "od_stem_fwd: H and W must be multiples of 32 (got %dx%d)", d->H, d->W);
  OD_REQUIRE(d->B > 0 && (long long)d->B * d->H * d->W * 16 < (1LL << 31), "od_stem_fwd: bad batch / tensor too large");
  OD_REQUIRE(d->act >= OD_ACT_LINEAR && d->act <= OD_ACT_ELU, "od_stem_fwd: bad act");
  OD_REQUIRE(d->act != OD_ACT_LEAKY || (d->alpha >= 0.f && d->alpha <= 1.f), "od_stem_fwd: leaky slope must be in [0, 1]");
  StemKP p;
  p.x = d->x;
  p.w0 = (const f16*)d->w0;
  p.s0 = d->scale0;
  p.b0 = d->bias0;
  p.w3 = (const f16*)d->w3;
  p.s3 = d->scale3;
  p.b3 = d->bias3;
  p.out = (f16*)d->out;
  p.B = d->B;
  p.H = d->H;
  p.W = d->W;
  p.k3stride = od_round_up(9 * 32, 64);
  p.act = d->act;
  p.alpha = d->alpha;
  p.tiles_x = d->W / 32;
  p.tiles_y = d->H / 32;
  int ntiles = d->B * p.tiles_x * p.tiles_y;
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  const int grid = ntiles < cus ? ntiles : cus;
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("OD_CONV_DEBUG");
    dbg = e ? atoi(e) : 0;
  }
  const void* fn = dbg == 64 && d->act == OD_ACT_LEAKY ? (const void*)&od_stem_k<1, OD_ACT_LEAKY>
                   : d->act == OD_ACT_LEAKY            ? (const void*)&od_stem_k<0, OD_ACT_LEAKY>
                   : d->act == OD_ACT_ELU              ? (const void*)&od_stem_k<0, OD_ACT_ELU>
                                                       : (const void*)&od_stem_k<0, OD_ACT_LINEAR>;
  if (int rc = od_ensure_lds(ctx, fn, (size_t)S_LDS)) return rc;
  void* args[] = {&p, &ntiles};
  OD_CHECK_HIP(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(512), args, (size_t)S_LDS, (hipStream_t)stream));
  return OD_OK;
}

// debug only (not part of include/odhip.h)
extern "C" int od_debug_stem_stamps(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stem_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
