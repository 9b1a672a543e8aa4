// K1/K2 (large layers): implicit-GEMM convolution on a 256-wide, 8-wave, ONE-workgroup-per-CU schedule.
//
// The 128x128 kernels of conv_mfma.hip move 15.6 KB from L2 into LDS per MFLOP and pay one barrier per 32 MFMAs per
// wave; they top out at ~35 % of the dense f16 MFMA peak.  This kernel halves the bytes per flop
// (BM x 256 output tile, BM = 160..256) and skews the two waves of every SIMD by half a phase, so that one of them is
// inside a 16-MFMA cluster while its partner issues LDS reads and the LDS-DMA of a later K tile:
//
//   waves 0-3 (wave row 0) and waves 4-7 (wave row 1) own (BM/2) x 64 output sub-tiles.
//   Per 64-deep K tile a wave does 4 phases = the 4 quadrants of its sub-tile, register operands reused:
//     p0: read A-lo (4 m-frags) + B-lo (2 n-frags)   MFMA A-lo x B-lo     stage B-hi of tile t+1
//     p1: read B-hi                                  MFMA A-lo x B-hi     stage A-hi of tile t+1
//     p2: read A-hi (MF1 m-frags)                    MFMA A-hi x B-hi     stage A-lo of tile t+2
//     p3: (B-lo still in registers)                  MFMA A-hi x B-lo     stage B-lo of tile t+2
//   load part = { ds_read_b128 ..., 2 x buffer_load_dwordx4 ... lds, counted s_waitcnt vmcnt }, MFMA part = 16 MFMAs.
//   ONE s_barrier per phase: wave row 0 runs { load part, MFMA part } between two barriers, wave row 1 runs { MFMA part of
//   the previous phase, load part } -- the skew is in program order, not in barrier count.
//   LDS = 2 K-tile buffers x 4 regions (A-lo, A-hi, B-lo, B-hi; 128 rows x 128 B each, chunk ^ (row & 7) swizzle on
//   the DMA source side) = 128 KiB.  A region is re-staged no earlier than two intervals after its last read, and read no
//   earlier than one interval after the counted wait (+ barrier) that retires its DMA -- for both wave rows.
//   vmcnt never drains to 0 in the steady state: every wait leaves the 4 youngest stages (8 DMAs per wave) in flight.
//
// A operand = activations gathered im2col-free (per-lane source = shifted input pixel or the zero page), B operand =
// packed weights; tap walk, padding masks, XCD-aware tile order, split-K slabs and the epilogue are those of
// conv_mfma.hip.  3x3 and 1x1, stride 1 and 2, Cin % 64 == 0.
//
// Replaces the Conv2D + BatchNormalization + LeakyReLU/ELU (+ Add) layers executed inside
// `ObjectDetector.predict` (reference voc_validate.py:27; docs/MODEL.md:5-21).
#include "conv_common.h"

namespace {

constexpr int E_BN = 256, E_BK = 64;
constexpr int E_REGION = 128 * 128;       // one staged region: 128 rows x 64 f16
constexpr int E_BUF = 4 * E_REGION;       // A-lo, A-hi, B-lo, B-hi
constexpr int E_ALO = 0, E_AHI = E_REGION, E_BLO = 2 * E_REGION, E_BHI = 3 * E_REGION;

// DBG & 32: s_memtime stamps of K tile 10, waves 0 and 4 of workgroup 0 (5 per phase: load part start, loads issued,
// DMA wait done, MFMA start, MFMA done); read back with od_debug_e8_stamps
__device__ unsigned long long g_e8_stamps[2][20];

struct TapWalk {  // wave-uniform position of a K tile inside the (tap, cin) axis
  int c0, tap, tapoff, dx;
  int sel_tap;  // tap the cached per-lane offsets (a_sel) were selected for
};

constexpr unsigned E_OOB = 0x80000000u;  // buffer offset beyond any tensor this kernel accepts: the lane reads zeros

// Epilogue straight from the accumulators (no LDS staging, no barrier).  After v_mfma_f32_16x16x32 with the weights as
// the A operand a lane (pixel l15, quad lq) holds channels lq*4..lq*4+3 of every 16-channel fragment.  One
// v_permlane16_swap per element on a PAIR of fragments (j, j+1) leaves every lane with 8 CONSECUTIVE channels of its
// pixel -- even lane rows get fragment j, odd rows fragment j+1 -- so scale/bias/activation/residual run in f32 on a
// 16-byte residual load and end in one 16-byte NHWC store (one rounding to f16), 64 contiguous bytes per pixel per
// instruction.  Same arithmetic and rounding points as conv_epilogue.
// NPAIR = fragment pairs per wave (2 = the 64-channel wave tile of the main GEMM, 1 = the 32-channel one of the fused
// pointwise layer); KEEP: the rounded f16 rows are also returned (keep[i][pr]) for that layer's LDS image; mrows = rows
// of this wave's sub-tile that exist (the fused layer's last row group of a BM < 256 tile is partly empty).
// pre_sc / pre_bi: this lane's 8 scale / bias values already in registers (NF == 2 only): the fused layer's epilogue then
// issues no load at all -- a load here would queue behind the first epilogue's stores (vector memory returns in order).
template <int MT, int NF = 4, bool KEEP = false>
static __device__ __forceinline__ void e8_epilogue_direct(const ConvKP& p, f32x4 (&acc)[MT][NF], int m0w, int n0w, int l15,
                                                          int lq, f16x8 (*keep)[NF / 2] = nullptr, int mrows = MT * 16,
                                                          const float* pre_sc = nullptr, const float* pre_bi = nullptr) {
  // pixel decomposition once per m-fragment (shared by both fragment pairs)
  unsigned pb[MT], ppix[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0w + i * 16 + l15;
    pb[i] = (unsigned)m / (unsigned)p.HoWo;
    ppix[i] = (unsigned)m - pb[i] * (unsigned)p.HoWo;
  }
#pragma unroll
  for (int pr = 0; pr < NF / 2; ++pr) {
    const int n = n0w + (2 * pr + (lq & 1)) * 16 + (lq >> 1) * 8;  // this lane's 8 channels
    const bool nok = n < p.Cout;
    float sc[8], bi[8];
    if (pre_sc) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[e] = pre_sc[e];
        bi[e] = pre_bi[e];
      }
    } else {
      const f32x4 s0 = *(const f32x4*)(p.scale + n), s1 = *(const f32x4*)(p.scale + n + 4);
      const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = s0[e];
        sc[4 + e] = s1[e];
        bi[e] = b0[e];
        bi[4 + e] = b1[e];
      }
    }
    // all residual rows of this pair are requested before the first one is used (MT loads in flight per lane)
    f16x8 rv[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0w + i * 16 + l15;
      f16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
      if (nok && m < p.M && p.res_mode != OD_RES_NONE) {
        long long roff;
        if (p.res_mode == OD_RES_SAME) {
          roff = (long long)m * p.Cout + n;
        } else {
          const unsigned ho = ppix[i] / (unsigned)p.Wo, wo = ppix[i] - ho * (unsigned)p.Wo;
          roff = ((long long)(pb[i] * (unsigned)(p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + n;
        }
        r = *(const f16x8*)(p.res + roff);
      }
      rv[i] = r;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0w + i * 16 + l15;
      const bool ok = nok && m < p.M && i * 16 + l15 < mrows;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = acc[i][2 * pr][e], bq = acc[i][2 * pr + 1][e];
        od_permlane16_swap(a, bq);
        v[e] = a;
        v[4 + e] = bq;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
      if (p.act == OD_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = od_leaky(v[e], p.alpha);
      } else if (p.act == OD_ACT_ELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : p.alpha * od_expm1_fast(v[e]);
      }
      if (p.res_mode != OD_RES_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rv[i][e];
      }
      if (ok) {
        const long long ooff = (long long)pb[i] * p.obs + (long long)ppix[i] * p.ops + n;
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
      if (KEEP) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)v[e];
        keep[i][pr] = h;
      }
    }
  }
}

// DBG (timing ablations only, results are garbage; selected by OD_CONV_DEBUG on the 3x3 / BM = 256 variant):
//   1 = no LDS-DMA, 2 = no fragment reads and no MFMA, 8 = no MFMA, 16 = no fragment reads
// BUF = 1: loaders are buffer_load_dwordx4 ... lds (resource in SGPRs, per-lane byte offset cached per filter tap,
// K position in the scalar offset: no per-DMA address arithmetic, padding = out-of-range offset -> zeros);
// BUF = 0: global_load_lds_dwordx4 with 64-bit per-lane addresses and the zero page (kept for A/B timing).
// PW: the pointwise (1x1) layer that consumes this tile's 256 output channels runs in the epilogue (see the end of the
// kernel); the launch then writes both tensors and the 1x1 layer has no launch of its own.
// SEG: grouped launch -- the m-tile index selects one of up to three input maps (x, out, H, W, M come from the segment
// table; stride 1, no residual): the prediction module shared by the pyramid levels as ONE launch per layer.
template <int KS, int MF1, int DBG = 0, int BUF = 1, bool PW = false, bool SEG = false>
__global__ __launch_bounds__(512, 2) void od_conv_8ph(ConvKP p_in) {
  constexpr int MF0 = 4, MT = MF0 + MF1, WROWS = MT * 16, BM = 2 * WROWS;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  unsigned long long cst[4] = {0, 0, 0, 0};
#define E8_CSTAMP(k)                                                                               \
  do {                                                                                             \
    if (DBG & 32) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cst[k])::"memory"); \
  } while (0)
  E8_CSTAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, lq = lane >> 4;

  int logical;
  {
    const int nt = p_in.mtiles * p_in.ntiles;
    const int pid = p_in.splitk > 1 ? (int)blockIdx.x / p_in.splitk : (int)blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, loc = pid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int tm = logical / p_in.ntiles;
  const int tn = logical - tm * p_in.ntiles;
  ConvKP p_seg;  // SEG only: the parameter block with the map-dependent fields of this tile's segment
  if (SEG) p_seg = p_in;
  if (SEG && p_in.nseg > 1) {  // wave-uniform: everything below sees one ordinary dense map
    int sg = 0;
    if (tm >= p_in.seg_tile0[1]) sg = 1;
    if (p_in.nseg > 2 && tm >= p_in.seg_tile0[2]) sg = 2;
    tm -= p_in.seg_tile0[sg];
    p_seg.x = p_in.seg_x[sg];
    p_seg.out = p_in.seg_out[sg];
    p_seg.H = p_seg.Ho = p_in.seg_H[sg];
    p_seg.W = p_seg.Wo = p_in.seg_W[sg];
    p_seg.HoWo = p_seg.H * p_seg.W;
    p_seg.M = p_in.seg_M[sg];
    p_seg.x_bytes = (unsigned)p_seg.M * (unsigned)p_in.Cin * 2u;
    if (p_in.obs == 0) p_seg.obs = (long long)p_seg.HoWo * p_in.Cout;  // dense output: the batch stride is this map's
  }
  const ConvKP& p = SEG ? p_seg : p_in;
  const int m0 = tm * BM, n0 = tn * E_BN;
  const int nk_all = p.Ktot / E_BK;
  const int ks0 = p.splitk > 1 ? ((int)blockIdx.x % p.splitk) * p.steps_per_split : 0;
  const int nk = p.splitk > 1 ? min(p.steps_per_split, nk_all - ks0) : nk_all;
  if (nk <= 0) return;

  // ---- per-lane staging state: 64 rows x 8 chunks per DMA instruction of the workgroup --------------------------
  const int rr = tid >> 3;
  const int lc = (tid & 7) ^ (rr & 7);
  int a_base[2][2];      // BUF: byte offset of the window-centre pixel (>= 0); else element offset of tap (0,0)
  unsigned a_vmask[2][2];
  unsigned a_sel[2][2];  // BUF: a_base or E_OOB for the tap the walk is at
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + j * WROWS + s * 64 + rr;
      a_vmask[s][j] = 0u;
      a_base[s][j] = 0;
      if (m < p.M && (s == 0 || rr < MF1 * 16)) {
        const unsigned b = (unsigned)m / (unsigned)p.HoWo;
        const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
        const unsigned ho = pix / (unsigned)p.Wo;
        const unsigned wo = pix - ho * (unsigned)p.Wo;
        const int hi0 = (int)ho * p.stride - p.pad, wi0 = (int)wo * p.stride - p.pad;
        a_base[s][j] = BUF ? ((((int)b * p.H + hi0 + p.pad) * p.W + wi0 + p.pad) * p.Cin + lc * 8) * 2
                           : (((int)b * p.H + hi0) * p.W + wi0) * p.Cin + lc * 8;
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) {
          const int hi = hi0 + t / KS, wi = wi0 + t % KS;
          if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) a_vmask[s][j] |= 1u << t;
        }
      }
    }
  int w_off[2][2];  // element offsets of this lane's weight rows
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + (2 * j + (rr >> 5)) * 64 + s * 32 + (rr & 31);
      w_off[s][j] = (n * p.Kstride + lc * 8 + ks0 * E_BK) * (BUF ? 2 : 1);
    }
  // buffer resources: x is addressed from (pad rows + pad pixels) before its start so that the tap offset is >= 0
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x - (long long)p.pad * (p.W + 1) * p.Cin), 0, (int)(p.x_bytes + (unsigned)(p.pad * (p.W + 1) * p.Cin * 2)),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
  auto blds16 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned voff, int soff, char* lptr) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lptr, 16, (int)voff, soff, 0, 0);
  };

  auto walk_init = [&](TapWalk& w, int kt) {
    const int k0 = kt * E_BK;
    if (KS == 3) {
      w.sel_tap = -1;
      w.tap = k0 / p.Cin;
      w.c0 = k0 - w.tap * p.Cin;
      const int dy = w.tap / 3;
      w.dx = w.tap - dy * 3;
      w.tapoff = (dy * p.W + w.dx) * p.Cin;
    } else {
      w.sel_tap = -1;
      w.tap = 0;
      w.c0 = k0;
      w.dx = 0;
      w.tapoff = 0;
    }
  };
  auto walk_next = [&](TapWalk& w) {
    w.c0 += E_BK;
    if (KS == 3 && w.c0 >= p.Cin) {
      w.c0 = 0;
      ++w.tap;
      if (++w.dx == 3) {
        w.dx = 0;
        w.tapoff += (p.W - 2) * p.Cin;
      } else {
        w.tapoff += p.Cin;
      }
    }
  };
  TapWalk walk_lo, walk_hi;  // next A-lo / A-hi tile to stage
  walk_init(walk_lo, ks0);
  walk_init(walk_hi, ks0);

  char* const piece = smem + wave * 1024;  // this wave's 8 rows inside a 64-row DMA round
  auto stage_a = [&](int s, TapWalk& w, int buf) {
    if (DBG & 1) return;
    if ((DBG & 64) && w.tap != 0) {  // ablation: A traffic of an LDS-window kernel (one DMA round per 9 K tiles)
      walk_next(w);
      return;
    }
    const int koff = w.tapoff + w.c0;
    char* dst = piece + buf * E_BUF + (s ? E_AHI : E_ALO);
    if (BUF) {
      if (w.sel_tap != w.tap) {  // wave-uniform: once per filter tap
        w.sel_tap = w.tap;
#pragma unroll
        for (int j = 0; j < 2; ++j) a_sel[s][j] = ((a_vmask[s][j] >> w.tap) & 1u) ? (unsigned)a_base[s][j] : E_OOB;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) blds16(rs_x, a_sel[s][j], koff * 2, dst + j * 8192);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool ok = (a_vmask[s][j] >> w.tap) & 1u;
        const f16* src = ok ? p.x + (a_base[s][j] + koff) : p.zero;
        glds16(src, dst + j * 8192);
      }
    }
    walk_next(w);
  };
  auto stage_b = [&](int s, int t, int buf) {
    if (DBG & 1) return;
    char* dst = piece + buf * E_BUF + (s ? E_BHI : E_BLO);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (BUF) blds16(rs_w, (unsigned)w_off[s][j], t * (E_BK * 2), dst + j * 8192);
      else glds16(p.w + (w_off[s][j] + t * E_BK), dst + j * 8192);
    }
  };

  // PW: the second layer's weights (64 KiB: 4 k-slabs of [128 output channels][64 k], the ring's row format) stream into
  // the K-tile buffer the LAST tile does not use, two DMAs per phase of that tile, so they have landed when the main loop ends
  const int w2buf = ((nk - 1) & 1) ^ 1;
  const __amdgpu_buffer_rsrc_t rs_w2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(PW ? p.w2 : p.w), 0, (int)(PW ? p.w2_bytes : p.w_bytes), 0x00020000);
  auto stage_w2 = [&](int ks) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      blds16(rs_w2, (unsigned)(((h * 64 + rr) * p.K2stride + lc * 8) * 2), ks * (E_BK * 2),
             piece + w2buf * E_BUF + ks * E_REGION + h * 8192);
  };

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: tile 0 complete + A-lo / B-lo of tile 1 -------------------------------------------------------
  stage_a(0, walk_lo, 0);
  stage_b(0, 0, 0);
  stage_b(1, 0, 0);
  stage_a(1, walk_hi, 0);
  if (nk > 1) {
    stage_a(0, walk_lo, 1);
    stage_b(0, 1, 1);
    wait_vmcnt<8>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();

  // fragment read offsets: row (l15) x 128 B, chunk (kh*4 + lq) ^ (l15 & 7)
  const int fa = (wr * 64 + l15) * 128 + ((lq ^ (l15 & 7)) * 16);
  const int fb = (wc * 32 + l15) * 128 + ((lq ^ (l15 & 7)) * 16);

  f16x8 xa[MF0][2], wlo[2][2], whi[2][2];
  constexpr bool kRead = !(DBG & (2 | 16)), kMma = !(DBG & (2 | 8));
  if (!kRead) {
#pragma unroll
    for (int f = 0; f < MF0; ++f)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) xa[f][kh] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) wlo[f][kh] = whi[f][kh] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
  }
  auto ldf = [&](f16x8& dst, const char* src) {
    if (kRead) dst = *(const f16x8*)src;
  };
  auto mma = [&](f32x4& c, const f16x8& a, const f16x8& b) {
    if (kMma) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    } else {
      asm volatile("" ::"v"(a), "v"(b));
    }
  };
  unsigned long long st[16];
#define E8_STAMP(k)                                                     \
  do {                                                                  \
    if (DBG & 32) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[k])::"memory"); \
  } while (0)

  // ---- the four load parts (fragment reads of this phase + LDS-DMA of a later tile + counted wait) and MFMA parts
  auto L0 = [&](int t) {
    const char* cur = smem + (t & 1) * E_BUF;
    E8_STAMP(0);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) ldf(wlo[f][kh], cur + E_BLO + ((fb + f * 2048) ^ (kh * 64)));
#pragma unroll
    for (int f = 0; f < MF0; ++f)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) ldf(xa[f][kh], cur + E_ALO + ((fa + f * 2048) ^ (kh * 64)));
    if (t + 1 < nk) {
      stage_b(1, t + 1, (t & 1) ^ 1);
      wait_vmcnt<8>();  // B-hi of tile t (read in the next phase) has landed
    } else if (PW) {
      stage_w2(0);
      wait_vmcnt<2>();  // everything of tile t has landed, the two W2 pieces stay in flight
    } else {
      wait_vmcnt<0>();
    }
    E8_STAMP(1);
  };
  auto L1 = [&](int t) {
    const char* cur = smem + (t & 1) * E_BUF;
    E8_STAMP(4);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) ldf(whi[f][kh], cur + E_BHI + ((fb + f * 2048) ^ (kh * 64)));
    if (t + 1 < nk) {
      stage_a(1, walk_hi, (t & 1) ^ 1);
      wait_vmcnt<8>();  // A-hi of tile t
    } else if (PW) {
      stage_w2(1);  // (tile t landed in L0)
    } else {
      wait_vmcnt<0>();
    }
    E8_STAMP(5);
  };
  auto L2 = [&](int t) {
    const char* cur = smem + (t & 1) * E_BUF;
    E8_STAMP(8);
#pragma unroll
    for (int f = 0; f < MF1; ++f)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) ldf(xa[f][kh], cur + E_AHI + ((fa + f * 2048) ^ (kh * 64)));
    if (t + 2 < nk) stage_a(0, walk_lo, t & 1);
    if (PW && t + 1 == nk) stage_w2(2);
    E8_STAMP(9);
  };
  auto L3 = [&](int t) {
    E8_STAMP(12);
    if (t + 2 < nk) {
      stage_b(0, t + 2, t & 1);
      wait_vmcnt<8>();  // A-lo and B-lo of tile t+1
    } else if (PW && t + 1 == nk) {
      stage_w2(3);  // (nothing of the main loop is in flight any more; the barrier after the loop waits for W2)
    } else {
      wait_vmcnt<0>();
    }
    E8_STAMP(13);
  };
  auto M = [&](int ph) {  // ph is a literal at every call site
    if (DBG & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    E8_STAMP(ph * 4 + 2);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < ((ph < 2) ? MF0 : MF1); ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (ph == 0) mma(acc[i][j], wlo[j][kh], xa[i][kh]);
          if (ph == 1) mma(acc[i][2 + j], whi[j][kh], xa[i][kh]);
          if (ph == 2) mma(acc[MF0 + i][2 + j], whi[j][kh], xa[i][kh]);
          if (ph == 3) mma(acc[MF0 + i][j], wlo[j][kh], xa[i][kh]);
        }
    __builtin_amdgcn_s_setprio(0);
    E8_STAMP(ph * 4 + 3);
  };
  auto dump_stamps = [&](int t) {
    if ((DBG & 32) && t == 10 && blockIdx.x == 0 && (wave & 3) == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0)
#pragma unroll
        for (int k = 0; k < 16; ++k) g_e8_stamps[wr][k] = st[k];
    }
  };

  E8_CSTAMP(1);
  // ONE barrier per phase.  Wave row 0 runs { load part, MFMA part } between two barriers, wave row 1 runs { MFMA part
  // of the previous phase, load part }: on every SIMD one wave is in its MFMA cluster while its partner issues reads
  // and DMAs, and neither waits for the other in between.  Both rows read, stage and wait for phase P in the same
  // barrier interval, so the LDS hazards are those of an unskewed loop: read one interval after the counted wait,
  // re-stage a region two or more intervals after its last read.
  if (wr == 0) {
    for (int t = 0; t < nk; ++t) {
      L0(t);
      M(0);
      __builtin_amdgcn_s_barrier();
      L1(t);
      M(1);
      __builtin_amdgcn_s_barrier();
      L2(t);
      M(2);
      __builtin_amdgcn_s_barrier();
      L3(t);
      M(3);
      __builtin_amdgcn_s_barrier();
      dump_stamps(t);
    }
  } else {
    L0(0);
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nk; ++t) {
      M(0);
      L1(t);
      __builtin_amdgcn_s_barrier();
      M(1);
      L2(t);
      __builtin_amdgcn_s_barrier();
      M(2);
      L3(t);
      __builtin_amdgcn_s_barrier();
      M(3);
      if (t + 1 < nk) {
        L0(t + 1);
        __builtin_amdgcn_s_barrier();
      }
      dump_stamps(t);
    }
  }
  __syncthreads();
  E8_CSTAMP(2);

  if (PW) {
    // ---- fused pointwise layer: t = act2(scale2 * (y . W2) + bias2) for this tile's BM pixels, y = the f16 rows the
    // epilogue below stores (all 256 channels of a pixel are in this workgroup: Cout == 256, one n tile).
    // LDS: the K-tile buffer the last tile did not use = W2 as 4 k-slabs of [128 out channels][64 k] (128-B rows, the
    // ring's swizzle; streamed in during the last K tile, see stage_w2), the other buffer = the y rows of ONE wave row
    // (WROWS pixels x 4 channel slabs of 64), so the second GEMM runs in two passes.
    const int rg = wave >> 2, cg = wave & 3;  // second GEMM: wave = 64 pixel rows x 32 output channels
    float sc2[8], bi2[8];                      // its scale / bias, loaded BEFORE the first epilogue's stores are queued
    {
      const int n2 = cg * 32 + (lq & 1) * 16 + (lq >> 1) * 8;
      const f32x4 s0 = *(const f32x4*)(p.scale2 + n2), s1 = *(const f32x4*)(p.scale2 + n2 + 4);
      const f32x4 b0 = *(const f32x4*)(p.bias2 + n2), b1 = *(const f32x4*)(p.bias2 + n2 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc2[e] = s0[e];
        sc2[4 + e] = s1[e];
        bi2[e] = b0[e];
        bi2[4 + e] = b1[e];
      }
    }
    f16x8 ykeep[MT][2];
    od_mfma_results_ready();
    e8_epilogue_direct<MT, 4, true>(p, acc, m0 + wr * WROWS, n0 + wc * 64, l15, lq, ykeep);
    ConvKP p2 = p;
    p2.scale = p.scale2;
    p2.bias = p.bias2;
    p2.act = p.act2;
    p2.alpha = p.alpha2;
    p2.res_mode = OD_RES_NONE;
    p2.out = (void*)p.out2;
    p2.out_f32 = 0;
    p2.Cout = p.Cout2;
    p2.obs = (long long)p.HoWo * p.Cout2;
    p2.ops = p.Cout2;
    char* const ybuf = smem + (w2buf ^ 1) * E_BUF;
    const char* const wbuf = smem + w2buf * E_BUF;
    const int fy = (rg * 64 + l15) * 128 + ((lq ^ (l15 & 7)) * 16);
    const int fw = (cg * 32 + l15) * 128 + ((lq ^ (l15 & 7)) * 16);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wr == half) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int row = i * 16 + l15;
            const int chunk = (2 * pr + (lq & 1)) * 2 + (lq >> 1);
            *(f16x8*)(ybuf + wc * E_REGION + row * 128 + ((chunk ^ (row & 7)) * 16)) = ykeep[i][pr];
          }
      }
      // (W2 landed before the __syncthreads that closed the main loop.)  The epilogue's stores keep draining behind the
      // second GEMM: a __syncthreads here would wait for them -> raw barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      f32x4 acc2[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int f = 0; f < 2; ++f) acc2[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          f16x8 wf[2], yf[4];
#pragma unroll
          for (int f = 0; f < 2; ++f) wf[f] = *(const f16x8*)(wbuf + ks * E_REGION + ((fw + f * 2048) ^ (kh * 64)));
#pragma unroll
          for (int i = 0; i < 4; ++i) yf[i] = *(const f16x8*)(ybuf + ks * E_REGION + ((fy + i * 2048) ^ (kh * 64)));
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc2[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f], yf[i], acc2[i][f], 0, 0, 0);
        }
      od_mfma_results_ready();
      e8_epilogue_direct<4, 2>(p2, acc2, m0 + half * WROWS + rg * 64, cg * 32, l15, lq, nullptr, WROWS - rg * 64, sc2, bi2);
      if (half == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave has read the first wave row's y image
      }
    }
  } else if (p.splitk > 1) {
    conv_epilogue<E_BN, 2, 4, MT, 4, 512>(p, smem, acc, m0, n0, tid, wr, wc, l15, lq);  // f32 partial slabs
  } else {
    od_mfma_results_ready();
    e8_epilogue_direct<MT>(p, acc, m0 + wr * WROWS, n0 + wc * 64, l15, lq);
  }
  if (DBG & 32) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    E8_CSTAMP(3);
    if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0)
#pragma unroll
      for (int k = 0; k < 4; ++k) g_e8_stamps[wr][16 + k] = cst[k];
  }
}

struct E8Entry {
  int BM;
  const void* k1;
  const void* k3;
  const char* name1;
  const char* name3;
  const void* k1pw;  // + the consuming pointwise layer in the epilogue
  const void* k3pw;
  const char* name1pw;
  const char* name3pw;
  const void* k3seg;  // grouped launch over several maps (3x3)
  const char* name3seg;
};
#define OD_E8(MF1)                                                                                        \
  {                                                                                                       \
    32 * (4 + MF1), (const void*)&od_conv_8ph<1, MF1>, (const void*)&od_conv_8ph<3, MF1>,                 \
        "od_conv_8ph<1, " #MF1 ", 0, 1, false, false>", "od_conv_8ph<3, " #MF1 ", 0, 1, false, false>",                             \
        (const void*)&od_conv_8ph<1, MF1, 0, 1, true>, (const void*)&od_conv_8ph<3, MF1, 0, 1, true>,     \
        "od_conv_8ph<1, " #MF1 ", 0, 1, true, false>", "od_conv_8ph<3, " #MF1 ", 0, 1, true, false>",                   \
        (const void*)&od_conv_8ph<3, MF1, 0, 1, false, true>, "od_conv_8ph<3, " #MF1 ", 0, 1, false, true>" \
  }
const E8Entry g_e8[] = {OD_E8(4), OD_E8(3), OD_E8(2), OD_E8(1)};  // BM = 256, 224, 192, 160
const void* const g_e8_dbg[][2] = {{(const void*)&od_conv_8ph<3, 4, 1>, "od_conv_8ph<3, 4, dbg1>"},
                                   {(const void*)&od_conv_8ph<3, 4, 2>, "od_conv_8ph<3, 4, dbg2>"},
                                   {(const void*)&od_conv_8ph<3, 4, 8>, "od_conv_8ph<3, 4, dbg8>"},
                                   {(const void*)&od_conv_8ph<3, 4, 16>, "od_conv_8ph<3, 4, dbg16>"},
                                   {(const void*)&od_conv_8ph<3, 4, 32>, "od_conv_8ph<3, 4, dbg32>"},
                                   {(const void*)&od_conv_8ph<3, 4, 64>, "od_conv_8ph<3, 4, dbg64>"},
                                   {(const void*)&od_conv_8ph<3, 4, 0, 0>, "od_conv_8ph<3, 4, glds>"}};
constexpr int kNumE8 = sizeof(g_e8) / sizeof(g_e8[0]);

}  // namespace

int od_conv_8ph_num_cfgs() { return kNumE8; }

bool od_conv_8ph_select(int idx, const ConvKP& p, int ksize, ConvKernelInfo* info, size_t* lds_bytes) {
  if (idx < 0 || idx >= kNumE8) return false;
  if ((p.Cin & 63) != 0 || p.tconv) return false;
  if (p.x_bytes >= 0x7F000000u || p.w_bytes >= 0x7F000000u || p.x_bytes == 0) return false;  // E_OOB must stay out of range
  const E8Entry& e = g_e8[idx];
  const bool pw = p.w2 != nullptr;  // the caller (od_conv2d_fwd) has checked od_conv_8ph_can_fuse_pointwise
  info->fn = ksize == 1 ? (pw ? e.k1pw : e.k1) : (pw ? e.k3pw : e.k3);
  info->name = ksize == 1 ? (pw ? e.name1pw : e.name1) : (pw ? e.name3pw : e.name3);
  if (p.nseg > 1) {
    if (ksize != 3 || pw || p.stride != 1 || p.res_mode != OD_RES_NONE) return false;
    info->fn = e.k3seg;
    info->name = e.name3seg;
  } else if (ksize == 3 && !pw) {
    // ordinary 3x3 launches run the segment-capable instantiation too (its segment table is empty: nseg <= 1), so that the
    // kernel is ONE symbol whether or not a layer is grouped; OD_E8_ONE_SYMBOL=0 keeps the plain instantiation (A/B timing)
    static int one = -1;
    if (one < 0) {
      const char* ev = getenv("OD_E8_ONE_SYMBOL");
      one = ev ? atoi(ev) : 1;
    }
    if (one) {
      info->fn = e.k3seg;
      info->name = e.name3seg;
    }
  }
  if (idx == 0 && ksize == 3 && p.dbg && !pw) {
    const int di = p.dbg == 1 ? 0 : p.dbg == 2 ? 1 : p.dbg == 8 ? 2 : p.dbg == 16 ? 3 : p.dbg == 32 ? 4 : p.dbg == 64 ? 5 : p.dbg == 128 ? 6 : -1;
    if (di >= 0) {
      info->fn = g_e8_dbg[di][0];
      info->name = (const char*)g_e8_dbg[di][1];
    }
  }
  info->BM = e.BM;
  info->BN = E_BN;
  info->threads = 512;
  const size_t epi = (size_t)(e.BM / 2) * (E_BN + 4) * 4;
  *lds_bytes = epi > (size_t)2 * E_BUF ? epi : (size_t)2 * E_BUF;
  return true;
}

// debug only (not part of include/odhip.h): copies the s_memtime stamps of the OD_CONV_DEBUG=32 build to the host
extern "C" int od_debug_e8_stamps(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_e8_stamps), sizeof(unsigned long long) * 40) == hipSuccess ? 0 : -1;
}
