// K1/K2: im2col-free implicit-GEMM convolution (3x3 / 1x1, stride 1 / 2, NHWC f16, f32 accumulate) on CDNA4 MFMA.
//
//   C[n, m] = sum_k W[n, k] * X[m, k]      m = (b, ho, wo) output pixel, n = output channel,
//                                          k = (dy*KS + dx)*Cin + cin  (never materialised)
//
// One workgroup (4 or 8 waves) owns a BM x BN output tile.  Per BK-deep K step it gathers the BM x BK activation
// slice and the BN x BK weight slice straight into LDS with global_load_lds_dwordx4 (LDS-DMA, 16 B per lane, per-lane
// SOURCE address = the shifted input pixel, or the context's zero page for padding / tails).  The LDS is a ring of
// STAGES buffers; loads run STAGES-1 steps ahead of the MFMAs and are retired with a COUNTED s_waitcnt vmcnt(N) + one
// raw s_barrier per step (never vmcnt(0) in the steady state), so the HBM/L2 latency of the gather is covered by
// STAGES-2 whole K steps of matrix work.  LDS images are bank-conflict free for ds_read_b128 fragment reads:
//   BK = 64: 128-B rows, 16-B chunk c of row r at chunk c ^ (r & 7)      (swizzle applied on the DMA source side)
//   BK = 32: per 16-row piece, chunk-major [chunk][row]                  (the DMA lane picks (row, chunk) to match)
// v_mfma_f32_16x16x32_f16 runs with the WEIGHTS as the A operand so that each lane ends up with 4 consecutive output
// channels of one pixel; the epilogue applies scale/bias/activation in f32, stages the tile through LDS and writes full
// NHWC lines (16 B per lane) with the residual added in f32 and ONE rounding to f16.
//
// Replaces the Conv2D + BatchNormalization + LeakyReLU/ELU (+ Add) layers executed inside
// `ObjectDetector.predict` (reference voc_validate.py:27; docs/MODEL.md:5-21).
#include <stdlib.h>
#include <string.h>

#include "conv_common.h"

namespace {

template <int BM, int BN, int BK, int STAGES, int WM, int WN, int SPEC = 0>
struct ConvCfg {
  static constexpr int NT = WM * WN * 64;            // threads of one role (consumers; = loaders when SPEC)
  static constexpr int NTHREADS = NT * (SPEC ? 2 : 1);
  static constexpr int CPR = BK / 8;    // 16-B chunks per row
  static constexpr int RPR = NT / CPR;  // rows covered by one DMA round of the whole workgroup
  static constexpr int AR = BM / RPR, BR = BN / RPR;
  static constexpr int ROWB = BK * 2;
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int WTM = BM / WM, WTN = BN / WN;
  static constexpr int MT = WTM / 16, NTL = WTN / 16;
  static constexpr int SLD = BN + 4;  // epilogue staging row stride (floats)
  static constexpr int PIPE_BYTES = STAGES * STAGE_BYTES;
  static constexpr int EPI_BYTES = WTM * SLD * 4;
  static constexpr int LDS_BYTES = PIPE_BYTES > EPI_BYTES ? PIPE_BYTES : EPI_BYTES;
  static constexpr int LOADS = AR + BR;  // LDS-DMA instructions per wave per K step
  static_assert(BM % RPR == 0 && BN % RPR == 0, "tile must be a whole number of DMA rounds");
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(LOADS * (STAGES - 2 > 0 ? STAGES - 2 : 0) <= 63, "vmcnt field");
};

// UNI: every BK-deep K step lies inside ONE filter tap (Cin % BK == 0; always true for KS == 1 with Cin % BK == 0):
// the tap walk is then wave-uniform scalar state advanced incrementally, and per-row padding validity is a 9-bit mask
// computed once.  UNI = false keeps a per-lane k -> (tap, cin) decomposition for odd channel counts.
// SPEC: wave specialisation.  The workgroup has 2 x WM*WN waves: the first half only runs MFMAs (consumers), the second
// half only issues the LDS-DMA (loaders) -- an LDS-DMA instruction costs its issuing wave ~70 cycles, which otherwise
// comes straight out of the MFMA stream.  Both halves meet at the same per-step barrier.
template <int BM, int BN, int BK, int STAGES, int WM, int WN, int KS, int MINW, bool UNI, int SPEC, bool STATS = false>
__global__ __launch_bounds__(WM* WN * 64 * (SPEC ? 2 : 1), MINW) void od_conv_igemm(ConvKP p) {
  using Cf = ConvCfg<BM, BN, BK, STAGES, WM, WN, SPEC>;
  constexpr int NT = Cf::NT, AR = Cf::AR, BR = Cf::BR, RPR = Cf::RPR, ROWB = Cf::ROWB;
  constexpr int WTM = Cf::WTM, WTN = Cf::WTN, MT = Cf::MT, NTL = Cf::NTL;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid_all = threadIdx.x;
  const int lane = tid_all & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
  const bool is_loader = SPEC ? (wave_all >= WM * WN) : true;
  const bool is_consumer = SPEC ? (wave_all < WM * WN) : true;
  const int tid = SPEC ? (tid_all & (NT - 1)) : tid_all;   // index inside the role
  const int wave = SPEC ? (wave_all >= WM * WN ? wave_all - WM * WN : wave_all) : wave_all;
  const int l15 = lane & 15, lq = lane >> 4;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2); give each XCD a contiguous run of logical tiles,
  // n fastest, so the tiles that re-read the same activation rows / halos hit the same L2.
  int logical;
  {
    const int nt = p.mtiles * p.ntiles;
    const int pid = p.splitk > 1 ? (int)blockIdx.x / p.splitk : (int)blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, loc = pid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = logical / p.ntiles, tn = logical - tm * p.ntiles;
  const int m0 = tm * BM, n0 = tn * BN;
  // split-K: this workgroup's K-step range
  const int nk_all = (p.Ktot + BK - 1) / BK;
  const int ks0 = p.splitk > 1 ? ((int)blockIdx.x % p.splitk) * p.steps_per_split : 0;
  int nk = p.splitk > 1 ? min(p.steps_per_split, nk_all - ks0) : nk_all;
  if (nk <= 0) return;  // uniform for the whole workgroup

  // ---- per-lane gather state --------------------------------------------------------------------------------
  int rr, lc, piece_off;  // row inside a DMA round, logical 16-B chunk fetched, LDS byte offset of this wave's piece
  if (BK == 64) {
    rr = tid >> 3;
    lc = (tid & 7) ^ (rr & 7);
    piece_off = wave * 8 * ROWB;
  } else {
    rr = wave * 16 + (lane & 15);
    lc = lane >> 4;
    piece_off = wave * 16 * ROWB;
  }
  int a_base[AR], a_hi0[AR], a_wi0[AR], a_b[AR];
  unsigned a_vmask[AR];  // UNI: bit t = tap t reads inside the image for this row
#pragma unroll
  for (int rd = 0; rd < AR; ++rd) {
    int m = m0 + rd * RPR + rr;
    a_vmask[rd] = 0u;
    if (p.tconv && m < p.M) m = od_tconv_pixel(p, (unsigned)m, BM);  // rows are grouped by output parity class (see below)
    if (m >= 0 && m < p.M) {
      const unsigned b = (unsigned)m / (unsigned)p.HoWo;
      const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
      const unsigned ho = pix / (unsigned)p.Wo;
      const unsigned wo = pix - ho * (unsigned)p.Wo;
      a_hi0[rd] = (int)ho * p.stride - p.pad;
      a_wi0[rd] = (int)wo * p.stride - p.pad;
      a_base[rd] = (((int)b * p.H + a_hi0[rd]) * p.W + a_wi0[rd]) * p.Cin + lc * 8;
      a_b[rd] = (int)b;
#pragma unroll
      for (int t = 0; t < KS * KS; ++t) {
        const int hi = a_hi0[rd] + t / KS, wi = a_wi0[rd] + t % KS;
        bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        if (p.tconv) ok = ok && !((hi | wi) & 1);  // only the even positions of the zero-upsampled view carry data
        if (ok) a_vmask[rd] |= 1u << t;
      }
    } else {
      a_hi0[rd] = -(1 << 24);
      a_wi0[rd] = 0;
      a_base[rd] = 0;
      a_b[rd] = 0;
    }
  }
  const f16* wrow = p.w + (long long)(n0 + rr) * p.Kstride + lc * 8;

  // Transposed mode (backward-data of a stride-2 conv): an output pixel only receives the taps whose source position in
  // the zero-upsampled view is even -- 1, 2, 2 or 4 of the 9, by the parity of (y, x).  The rows of the GEMM are ordered
  // tile by tile in parity classes (od_tconv_pixel), so a tile's rows share their class and the taps that
  // are zero for EVERY row of the tile are skipped as whole K steps: 2.25 instead of 9 taps on average.  Skipped steps
  // only add exact zeros, so the result is bit-identical to the un-skipped walk.
  unsigned tmask = 0x1FFu;
  if (UNI && KS == 3 && p.tconv && p.splitk <= 1) {
    unsigned um = 0u;
#pragma unroll
    for (int rd = 0; rd < AR; ++rd) um |= a_vmask[rd];
    for (int off = 32; off; off >>= 1) um |= (unsigned)__shfl_xor((int)um, off, 64);
    unsigned* sh = (unsigned*)smem;
    if (tid_all == 0) *sh = 0u;
    __syncthreads();
    if (lane == 0) atomicOr(sh, um);
    __syncthreads();
    tmask = (unsigned)__builtin_amdgcn_readfirstlane((int)*sh);
    __syncthreads();  // smem[0] is part of the ring from here on
    nk = __builtin_popcount(tmask) * (p.Cin / BK);
  }

  // loader state: the NEXT step to stage (steps are staged strictly in order) -- all wave-uniform scalars
  int ld_k0 = ks0 * BK, ld_c0 = ld_k0, ld_tap = 0, ld_tapoff = 0, ld_dx = 0, ld_dy = 0;
  if (UNI && KS == 3) {
    ld_tap = ld_k0 / p.Cin;
    ld_c0 = ld_k0 - ld_tap * p.Cin;
    if (tmask != 0x1FFu && tmask != 0u) {
      ld_tap = __builtin_ctz(tmask);
      ld_c0 = 0;
      ld_k0 = ld_tap * p.Cin;
    }
    ld_dy = ld_tap / 3;
    ld_dx = ld_tap - ld_dy * 3;
    ld_tapoff = (ld_dy * p.W + ld_dx) * p.Cin;
  }

  // SPEC == 2: the loader waves stage through REGISTERS (global_load_dwordx4 -> ds_write_b128) instead of LDS-DMA: the
  // L2 -> LDS-DMA path tops out near 30 B/clk/CU, plain vector loads from L2 reach about twice that
  constexpr int NX = AR + BR;
  f16x8 rtmp[SPEC == 2 ? NX : 1];
  char* rdst[SPEC == 2 ? NX : 1];
  int rn = 0;
  auto xfer = [&](const f16* src, char* lds_piece) {
    if (SPEC == 2) {
      rtmp[rn] = *(const f16x8*)src;
      rdst[rn] = lds_piece + lane * 16;
      ++rn;
    } else {
      glds16(src, lds_piece);
    }
  };
  auto stage = [&](int buf) {
    if (p.dbg & 1) return;
    rn = 0;
    char* abuf = smem + buf * Cf::STAGE_BYTES + piece_off;
    char* bbuf = abuf + Cf::A_BYTES;
    if (UNI) {
      const bool cvalid = (KS == 3) || (ld_c0 + lc * 8 < p.Cin);
      const int koff = ld_tapoff + ld_c0;  // scalar: (dy*W + dx)*Cin + c0
#pragma unroll
      for (int rd = 0; rd < AR; ++rd) {
        const bool ok = cvalid && ((a_vmask[rd] >> ld_tap) & 1u);
        const f16* src;
        if (KS == 3 && p.tconv) {
          const int u = (a_hi0[rd] + ld_dy) >> 1, v = (a_wi0[rd] + ld_dx) >> 1;
          src = ok ? p.x + (((a_b[rd] * p.Hs + u) * p.Ws + v) * p.Cin + lc * 8 + ld_c0) : p.zero;
        } else {
          src = ok ? p.x + (a_base[rd] + koff) : p.zero;
        }
        xfer(src, abuf + rd * RPR * ROWB);
      }
    } else {
      const int k = ld_k0 + lc * 8;
      const int tap = k / p.Cin;
      const int dy = tap / 3, dx = tap - dy * 3;
      const int cin = k - tap * p.Cin;
      const bool kvalid = k < p.Ktot;
      const int koff = (dy * p.W + dx) * p.Cin + cin - lc * 8;
#pragma unroll
      for (int rd = 0; rd < AR; ++rd) {
        const int hi = a_hi0[rd] + dy, wi = a_wi0[rd] + dx;
        const bool ok = kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        const f16* src = ok ? p.x + (a_base[rd] + koff) : p.zero;
        xfer(src, abuf + rd * RPR * ROWB);
      }
    }
#pragma unroll
    for (int rd = 0; rd < BR; ++rd) xfer(wrow + (long long)rd * RPR * p.Kstride + ld_k0, bbuf + rd * RPR * ROWB);
    if (SPEC == 2) {
#pragma unroll
      for (int i = 0; i < NX; ++i) *(f16x8*)rdst[i] = rtmp[i];
    }
    // advance to the next step
    ld_k0 += BK;
    if (UNI) {
      ld_c0 += BK;
      if (KS == 3 && ld_c0 >= p.Cin) {
        ld_c0 = 0;
        ++ld_tap;
        if (++ld_dx == 3) {
          ld_dx = 0;
          ++ld_dy;
          ld_tapoff += (p.W - 2) * p.Cin;
        } else {
          ld_tapoff += p.Cin;
        }
        while (ld_tap < 9 && !((tmask >> ld_tap) & 1u)) {  // taps that are zero for the whole tile (transposed mode)
          ++ld_tap;
          ld_k0 += p.Cin;
          if (++ld_dx == 3) {
            ld_dx = 0;
            ++ld_dy;
          }
        }
      }
    }
  };

  // ---- main loop: STAGES-deep LDS ring, counted vmcnt, one raw barrier per K step ------------------------------
  const int wm = wave / WN, wn = wave - wm * WN;
  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (is_loader) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) stage(s);
  }

  const int swz = l15 & 7;
  int buf = 0;
  for (int ks = 0; ks < nk; ++ks) {
    // retire this step's DMA (issued STAGES-1 steps ago); later steps stay in flight
    if ((p.dbg & 4) || !is_loader) {
      // consumers have no DMA of their own; dbg bit 2: never wait for the DMA (garbage results, timing ablation)
    } else if (STAGES > 2 && ks + (STAGES - 2) < nk) {
      wait_vmcnt<Cf::LOADS*(STAGES > 2 ? STAGES - 2 : 0)>();
    } else {
      wait_vmcnt<0>();
    }
    if (SPEC == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's piece of step ks landed; everyone finished reading step ks-1
    {
      const int nxt = ks + STAGES - 1;
      int nb = buf + STAGES - 1;
      if (nb >= STAGES) nb -= STAGES;
      if (is_loader && nxt < nk) stage(nb);  // overwrites the buffer read at step ks-1
    }
    const char* abuf = smem + buf * Cf::STAGE_BYTES;
    const char* bbuf = abuf + Cf::A_BYTES;
    __builtin_amdgcn_s_setprio(1);
    if (is_consumer && !(p.dbg & 2))
#pragma unroll
    for (int kh = 0; kh < BK / 32; ++kh) {
      f16x8 xa[MT], wb[NTL];
      if (BK == 64) {
        const int coff = ((kh * 4 + lq) ^ swz) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[i] = *(const f16x8*)(abuf + (wm * WTM + i * 16 + l15) * ROWB + coff);
#pragma unroll
        for (int j = 0; j < NTL; ++j) wb[j] = *(const f16x8*)(bbuf + (wn * WTN + j * 16 + l15) * ROWB + coff);
      } else {
        const int coff = lq * 256 + l15 * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) xa[i] = *(const f16x8*)(abuf + (wm * WTM + i * 16) * ROWB + coff);
#pragma unroll
        for (int j = 0; j < NTL; ++j) wb[j] = *(const f16x8*)(bbuf + (wn * WTN + j * 16) * ROWB + coff);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    if (++buf == STAGES) buf = 0;
  }
  __syncthreads();  // all fragment reads done before the ring is reused as epilogue staging

  conv_epilogue<BN, WM, WN, MT, NTL, Cf::NTHREADS, STATS>(p, smem, acc, m0, n0, tid_all, is_consumer ? wm : -1, wn, l15, lq);
}

// split-K finish: out = act(scale * sum_s slab[s] + bias) (+ residual); slabs summed in ascending s (deterministic)
__global__ __launch_bounds__(256) void od_conv_finish(ConvKP p) {
  const long long nvec = (long long)p.M * (p.Cout >> 3);
  const int C8 = p.Cout >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / C8), n = (int)(i - (long long)m * C8) * 8;
    const float* wsp = p.ws + (long long)m * p.Cout + n;
    const long long slab = (long long)p.M * p.Cout;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sidx = 0; sidx < p.splitk; ++sidx) {
      const f32x4 a0 = *(const f32x4*)(wsp + sidx * slab), a1 = *(const f32x4*)(wsp + sidx * slab + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += a0[e];
        v[4 + e] += a1[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = v[e] * p.scale[n + e] + p.bias[n + e];
      if (p.act == OD_ACT_LEAKY) v[e] = od_leaky(v[e], p.alpha);
      else if (p.act == OD_ACT_ELU) v[e] = v[e] > 0.f ? v[e] : p.alpha * od_expm1_fast(v[e]);
    }
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

struct TileCfg {
  int BM, BN, BK, threads;
  size_t lds;
  const void* k1;   // KS == 1 (channel tail masked per lane: any Cin % 8 == 0)
  const void* k3;   // KS == 3, Cin % BK == 0
  const void* k3g;  // KS == 3, any Cin % 8 == 0 (per-lane tap decomposition); may be null
  const char* name1;
  const char* name3;
  const char* name3g;
};

#define OD_STR2(x) #x
#define OD_STR(x) OD_STR2(x)
#define OD_NAME(BM, BN, BK, ST, WM, WN, KS, MINW, UNI)                                                              \
  "od_conv_igemm<" OD_STR(BM) ", " OD_STR(BN) ", " OD_STR(BK) ", " OD_STR(ST) ", " OD_STR(WM) ", " OD_STR(WN) ", " OD_STR(KS) ", " OD_STR(MINW) ", " #UNI ", 0, false>"
#define OD_CFG(BM, BN, BK, ST, WM, WN, MINW)                                                                        \
  {                                                                                                                 \
    BM, BN, BK, WM* WN * 64, (size_t)ConvCfg<BM, BN, BK, ST, WM, WN>::LDS_BYTES,                                    \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 1, MINW, true, 0>,                                      \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 3, MINW, true, 0>, nullptr,                                \
        OD_NAME(BM, BN, BK, ST, WM, WN, 1, MINW, true), OD_NAME(BM, BN, BK, ST, WM, WN, 3, MINW, true), ""          \
  }
#define OD_CFG_S(BM, BN, BK, ST, WM, WN, MINW)                                                                      \
  {                                                                                                                 \
    BM, BN, BK, WM* WN * 128, (size_t)ConvCfg<BM, BN, BK, ST, WM, WN, 1>::LDS_BYTES,                                \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 1, MINW, true, 1>,                                      \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 3, MINW, true, 1>, nullptr,                             \
        "od_conv_igemm<" OD_STR(BM) ", " OD_STR(BN) ", " OD_STR(BK) ", " OD_STR(ST) ", " OD_STR(WM) ", " OD_STR(WN) ", 1, " OD_STR(MINW) ", true, 1, false>", \
        "od_conv_igemm<" OD_STR(BM) ", " OD_STR(BN) ", " OD_STR(BK) ", " OD_STR(ST) ", " OD_STR(WM) ", " OD_STR(WN) ", 3, " OD_STR(MINW) ", true, 1, false>", "" \
  }
#define OD_CFG_S2(BM, BN, BK, ST, WM, WN, MINW)                                                                     \
  {                                                                                                                 \
    BM, BN, BK, WM* WN * 128, (size_t)ConvCfg<BM, BN, BK, ST, WM, WN, 1>::LDS_BYTES,                                \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 1, MINW, true, 2>,                                      \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 3, MINW, true, 2>, nullptr,                             \
        "od_conv_igemm<" OD_STR(BM) ", " OD_STR(BN) ", " OD_STR(BK) ", " OD_STR(ST) ", " OD_STR(WM) ", " OD_STR(WN) ", 1, " OD_STR(MINW) ", true, 2, false>", \
        "od_conv_igemm<" OD_STR(BM) ", " OD_STR(BN) ", " OD_STR(BK) ", " OD_STR(ST) ", " OD_STR(WM) ", " OD_STR(WN) ", 3, " OD_STR(MINW) ", true, 2, false>", "" \
  }
#define OD_CFG_G(BM, BN, BK, ST, WM, WN, MINW)                                                                      \
  {                                                                                                                 \
    BM, BN, BK, WM* WN * 64, (size_t)ConvCfg<BM, BN, BK, ST, WM, WN>::LDS_BYTES,                                    \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 1, MINW, true, 0>,                                      \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 3, MINW, true, 0>,                                      \
        (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, 3, MINW, false, 0>,                                        \
        OD_NAME(BM, BN, BK, ST, WM, WN, 1, MINW, true), OD_NAME(BM, BN, BK, ST, WM, WN, 3, MINW, true),             \
        OD_NAME(BM, BN, BK, ST, WM, WN, 3, MINW, false)                                                             \
  }

//                   BM   BN  BK ST WM WN minwaves/SIMD
// Only what pick_cfg / the bn_partials fallback can select (round 3: the 22 table configs, the LDS-window kernels and the
// persistent window kernel that never won a layer are gone -- profiles/r01/conv_cfg_sweep.txt, profiles/r02/spec64_sweep.txt
// record what they measured).
const TileCfg g_cfgs[] = {
    OD_CFG_G(128, 128, 64, 2, 2, 2, 2),  // 0: generic geometry (any Cin % 8 == 0; 64 KiB, 2 WG/CU)
    OD_CFG_G(128, 64, 64, 2, 2, 2, 2),   // 1
    OD_CFG_G(64, 128, 64, 2, 2, 2, 2),   // 2
    OD_CFG_G(64, 64, 64, 2, 2, 2, 2),    // 3
    OD_CFG_S(128, 128, 64, 2, 2, 2, 4),  // 4: 4 MFMA waves + 4 DMA waves, 64 KiB, 2 WG/CU
    OD_CFG_S(128, 128, 64, 3, 2, 2, 2),  // 5: same, 3-deep ring (96 KiB, 1 WG/CU): few tiles, long K
    OD_CFG(64, 64, 64, 4, 2, 2, 2),      // 6: deep ring for the short-K 1x1 layers (cold L2: latency, not bandwidth)
    OD_CFG_S(64, 128, 64, 3, 2, 2, 4),   // 7: specialised 64 x 128, 72 KiB: small-M layers
};
constexpr int kNumCfgs = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

// STATS instantiations (od_conv_desc.bn_partials: BatchNorm partial sums written by the epilogue) of the table configs the
// training forward pass is given: variant 0 = 1x1, 1 = 3x3 tap-uniform, 2 = 3x3 generic.  nullptr = not instantiated.
#define OD_ST(BM, BN, BK, ST, WM, WN, KS, MINW, UNI, SPEC) (const void*)&od_conv_igemm<BM, BN, BK, ST, WM, WN, KS, MINW, UNI, SPEC, true>
const void* stats_kernel(int cfg, int variant) {
  switch (cfg) {
    case 0: return variant == 0 ? OD_ST(128, 128, 64, 2, 2, 2, 1, 2, true, 0) : variant == 1 ? OD_ST(128, 128, 64, 2, 2, 2, 3, 2, true, 0) : OD_ST(128, 128, 64, 2, 2, 2, 3, 2, false, 0);
    case 1: return variant == 0 ? OD_ST(128, 64, 64, 2, 2, 2, 1, 2, true, 0) : variant == 1 ? OD_ST(128, 64, 64, 2, 2, 2, 3, 2, true, 0) : OD_ST(128, 64, 64, 2, 2, 2, 3, 2, false, 0);
    case 2: return variant == 0 ? OD_ST(64, 128, 64, 2, 2, 2, 1, 2, true, 0) : variant == 1 ? OD_ST(64, 128, 64, 2, 2, 2, 3, 2, true, 0) : OD_ST(64, 128, 64, 2, 2, 2, 3, 2, false, 0);
    case 3: return variant == 0 ? OD_ST(64, 64, 64, 2, 2, 2, 1, 2, true, 0) : variant == 1 ? OD_ST(64, 64, 64, 2, 2, 2, 3, 2, true, 0) : OD_ST(64, 64, 64, 2, 2, 2, 3, 2, false, 0);
    case 4: return variant == 0 ? OD_ST(128, 128, 64, 2, 2, 2, 1, 4, true, 1) : variant == 1 ? OD_ST(128, 128, 64, 2, 2, 2, 3, 4, true, 1) : nullptr;
    case 5: return variant == 0 ? OD_ST(128, 128, 64, 3, 2, 2, 1, 2, true, 1) : variant == 1 ? OD_ST(128, 128, 64, 3, 2, 2, 3, 2, true, 1) : nullptr;
    case 6: return variant == 0 ? OD_ST(64, 64, 64, 4, 2, 2, 1, 2, true, 0) : variant == 1 ? OD_ST(64, 64, 64, 4, 2, 2, 3, 2, true, 0) : nullptr;
    case 7: return variant == 0 ? OD_ST(64, 128, 64, 3, 2, 2, 1, 4, true, 1) : variant == 1 ? OD_ST(64, 128, 64, 3, 2, 2, 3, 4, true, 1) : nullptr;
    default: return nullptr;
  }
}
#undef OD_ST

// Tile choice from the measured table (profiles/r01/conv_cfg_sweep.txt; MI355X, batch-32 Darknet53 shapes).
int pick_cfg(const od_ctx* ctx, int M, int Cin, int Cout, int ksize, bool e8_ok, bool throughput) {
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  const bool spec_ok = (Cin % 64) == 0;  // wave-specialised kernels are tap-uniform only
  if (Cout <= 64) return (ksize == 3 && od_ceil_div(M, 128) >= 8 * cus) ? 1 : 3;
  const long t128 = (long)od_ceil_div(M, 128) * od_ceil_div(Cout, 128);
  if (ksize == 1) {
    if (!spec_ok) return 3;
    // small-M layers, measured IN the network (scripts/sweep_net_cfg.py, profiles/r01/conv_innet_sweep.txt): their input
    // was just written by the previous kernel, every first touch misses L2, so ring depth matters more than in a
    // back-to-back microbenchmark -- 3-deep specialised 64x128 (7) for M <= 16 k, 4-deep 64x64 (6) for long-K 1x1 at M <= 4 k.
    // (Round 2 tried a SPECIALISED 64 x 64 tile -- 4 MFMA + 4 DMA waves -- for these short-K layers: slower than the plain
    // one on every 1x1 shape, 15.9 vs 13.0 us on s3.a; profiles/r02/spec64_sweep.txt.)
    if (M <= 4096) return (M >= 2048 && Cin >= 512) ? 6 : 3;
    if (M <= 16384) return 7;
    if (Cout < 256) return 3;
    // wide 1x1 (neck laterals): fall through to the 128x128 / 8-wave comparison below
    if (t128 < cus) return 4;
  }
  if (!spec_ok) return t128 >= 2L * cus ? 0 : 2;
  if (throughput && e8_ok && ksize == 3 && Cout >= 192 && M >= 2048) {
    // tile_cfg = -2: other launches overlap this one (batches in flight on several streams), so an under-filled grid is
    // not wasted and the figure of merit is CU x time, not time: the 8-wave kernel (one workgroup per CU, half the
    // L2->LDS bytes per flop) then also takes the stage-4 / stage-5 layers (profiles/r01/inflight_sweep.txt: +4.5 % img/s)
    const int nk = od_ceil_div(ksize * ksize * Cin, 64);
    double best = t128 >= cus ? t128 * (7.5 + 1.07 * nk) * 0.5 : t128 * (10.0 + 0.55 * nk);
    int pick = t128 >= cus ? 4 : (Cout <= 256 ? 7 : 5);
    for (int i = 0; i < od_conv_8ph_num_cfgs(); ++i) {
      const int mt = 8 - i, bm = 32 * mt;
      const long tiles = (long)od_ceil_div(M, bm) * od_ceil_div(Cout, 256);
      if (tiles * 3 < cus) continue;  // a grid below a third of the chip gained nothing (stage 5, coarse head levels)
      const double c = (double)tiles * (13.0 + 1.78 * nk * (0.5 + 0.0625 * mt));
      if (c < 0.95 * best) {
        best = c / 0.95;
        pick = kNumCfgs + i;
      }
    }
    return pick;
  }
  if (t128 >= cus) {
    if (Cout == 128) return 2;
    // 128x128 specialised kernel (2 workgroups per CU) vs the 8-wave BM x 256 kernel (1 per CU): whole rounds x
    // (fixed cost + K tiles x cost per tile), constants in us from profiles/r01/conv_8ph_sweep_{320,640}.txt
    const int nk = od_ceil_div(ksize * ksize * Cin, 64);
    const double c13 = (double)od_ceil_div((int)t128, 2 * cus) * (7.5 + 1.07 * nk);
    double best = 0.93 * c13;
    int pick = 4;
    for (int i = 0; e8_ok && i < od_conv_8ph_num_cfgs(); ++i) {
      const int mt = 8 - i, bm = 32 * mt;
      const long tiles = (long)od_ceil_div(M, bm) * od_ceil_div(Cout, 256);
      const double c = (double)((tiles + cus - 1) / cus) * (13.0 + 1.78 * nk * (0.5 + 0.0625 * mt));
      if (c < best) {
        best = c;
        pick = kNumCfgs + i;
      }
    }
    return pick;
  }
  if (Cout <= 256) return M < 2048 ? 3 : 7;       // few, narrow tiles (neck / prediction module on the coarse levels)
  // few tiles, long K.  Up to half a round of 128 x 128 tiles (backward-data of stage 5: M = 3200, Cout = 512, K = 9216) the
  // 64-row specialised tile doubles the workgroups: 55.6 vs 78.8 us (profiles/r02/dgrad_cfg_sweep.txt); above that one
  // deep-ring workgroup per CU
  static int last = -2;
  if (last == -2) {
    const char* e = getenv("OD_PICK_FEW_TILES");  // tuning: force the config of this branch
    last = e ? atoi(e) : -1;
  }
  if (last >= 0) return last;
  return (M >= 2048 && 2 * t128 <= cus) ? 7 : 5;  // (batch-1 maps keep their split-K plan on 5)
}

}  // namespace

extern "C" int od_conv_num_tile_cfgs(void) { return kNumCfgs + od_conv_8ph_num_cfgs(); }

extern "C" int od_conv_weight_dims(int cout, int cin, int ksize, int* cout_pad, int* kpad) {
  OD_REQUIRE(cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "od_conv_weight_dims: bad dims");
  if (cout_pad) *cout_pad = od_round_up(cout, 256);
  if (kpad) *kpad = od_round_up(ksize * ksize * cin, 64);
  return OD_OK;
}

static int od_conv2d_fwd_main(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run,
                              int* mtiles_out, bool* fused_out);

// od_conv_desc.w2 (the pointwise layer that consumes this launch's output): inside the 8-wave kernel's epilogue when the
// selected kernel can do it, otherwise as a second launch right behind the first -- the caller's plan is the same either way
static int od_conv2d_fwd_impl2(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run,
                               int* mtiles_out) {
  bool fused = false;
  if (d && d->nseg > 1) {
    // grouped launch: one launch when the 8-wave kernel takes the layer, else one ordinary launch per segment
    OD_REQUIRE(d->nseg <= 3, "od_conv2d_fwd: nseg %d > 3", d->nseg);
    OD_REQUIRE(d->ksize == 3 && d->stride == 1 && d->res_mode == OD_RES_NONE && !d->w2 && !d->bn_partials && !d->transposed,
               "od_conv2d_fwd: a grouped launch (nseg > 1) is a 3x3 stride-1 layer without residual / w2 / bn_partials / "
               "transposed mode");
    for (int i = 0; i < d->nseg; ++i)
      OD_REQUIRE(d->seg_x[i] && d->seg_out[i] && d->seg_H[i] > 0 && d->seg_W[i] > 0, "od_conv2d_fwd: segment %d is incomplete", i);
    od_conv_desc q = *d;  // the first segment stands in for x / out / H / W in the shared validation
    q.x = d->seg_x[0];
    q.out = d->seg_out[0];
    q.H = d->seg_H[0];
    q.W = d->seg_W[0];
    bool grouped = false;
    int rc = od_conv2d_fwd_main(ctx, &q, stream, kernel_name, dry_run, mtiles_out, &grouped);
    if (rc != OD_OK || grouped) return rc;
    for (int i = 0; i < d->nseg; ++i) {
      q.nseg = 0;
      q.x = d->seg_x[i];
      q.out = d->seg_out[i];
      q.H = d->seg_H[i];
      q.W = d->seg_W[i];
      rc = od_conv2d_fwd_main(ctx, &q, stream, i == 0 ? kernel_name : nullptr, dry_run, nullptr, nullptr);
      if (rc != OD_OK) return rc;
    }
    return OD_OK;
  }
  const int rc = od_conv2d_fwd_main(ctx, d, stream, kernel_name, dry_run, mtiles_out, &fused);
  if (rc != OD_OK || !d->w2 || fused || dry_run) return rc;
  const int pad = d->ksize / 2;
  od_conv_desc q;
  memset(&q, 0, sizeof(q));
  q.x = d->out;
  q.w = d->w2;
  q.scale = d->scale2;
  q.bias = d->bias2;
  q.out = d->out2;
  q.B = d->B;
  q.H = (d->H + 2 * pad - d->ksize) / d->stride + 1;
  q.W = (d->W + 2 * pad - d->ksize) / d->stride + 1;
  q.Cin = d->Cout;
  q.Cout = d->Cout2;
  q.ksize = 1;
  q.stride = 1;
  q.act = d->act2;
  q.alpha = d->alpha2;
  q.res_mode = OD_RES_NONE;
  q.out_dtype = OD_DT_F16;
  q.tile_cfg = d->tile_cfg < 0 ? d->tile_cfg : -1;
  q.splitk = d->splitk;
  q.splitk_workspace = d->splitk_workspace;  // same stream: the first launch's finish kernel is done with it
  q.splitk_workspace_bytes = d->splitk_workspace_bytes;
  return od_conv2d_fwd_main(ctx, &q, stream, nullptr, false, nullptr, nullptr);
}

int od_conv2d_fwd_impl(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name,
                       bool dry_run) {
  return od_conv2d_fwd_impl2(ctx, d, stream, kernel_name, dry_run, nullptr);
}

static int od_conv2d_fwd_rows_impl(od_ctx* ctx, const od_conv_desc* d, int* rows) {
  return od_conv2d_fwd_impl2(ctx, d, nullptr, nullptr, true, rows);
}

static int od_conv2d_fwd_main(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run,
                              int* mtiles_out, bool* fused_out) {
  OD_REQUIRE(ctx && d, "od_conv2d_fwd: null ctx/desc");
  OD_REQUIRE(d->x && d->w && d->scale && d->bias && d->out, "od_conv2d_fwd: null tensor");
  OD_REQUIRE(d->ksize == 1 || d->ksize == 3, "od_conv2d_fwd: ksize %d unsupported", d->ksize);
  OD_REQUIRE(d->stride == 1 || d->stride == 2, "od_conv2d_fwd: stride %d unsupported", d->stride);
  OD_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "od_conv2d_fwd: bad dims");
  OD_REQUIRE(d->Cin % 8 == 0 && d->Cout % 8 == 0, "od_conv2d_fwd: Cin/Cout must be multiples of 8 (got %d/%d)",
             d->Cin, d->Cout);
  OD_REQUIRE(d->res_mode == OD_RES_NONE || d->res, "od_conv2d_fwd: res_mode set but res is null");
  OD_REQUIRE(d->act >= OD_ACT_LINEAR && d->act <= OD_ACT_ELU, "od_conv2d_fwd: bad act");
  OD_REQUIRE(d->act != OD_ACT_LEAKY || (d->alpha >= 0.f && d->alpha <= 1.f), "od_conv2d_fwd: leaky slope must be in [0, 1]");
  const int pad = d->ksize / 2;
  const bool tconv = d->transposed != 0;
  if (tconv)
    OD_REQUIRE(d->ksize == 3 && d->stride == 2 && d->Cin % 64 == 0,
               "od_conv2d_fwd: transposed mode is the backward-data of a 3x3 stride-2 conv (Cin %% 64 == 0)");
  // transposed: a stride-1 conv over the 2x zero-upsampled [B, 2H, 2W, Cin] view of x
  const int Hv = tconv ? 2 * d->H : d->H, Wv = tconv ? 2 * d->W : d->W, stride = tconv ? 1 : d->stride;
  const int Ho = (Hv + 2 * pad - d->ksize) / stride + 1;
  const int Wo = (Wv + 2 * pad - d->ksize) / stride + 1;
  if (d->res_mode == OD_RES_UP2)
    OD_REQUIRE(Ho % 2 == 0 && Wo % 2 == 0, "od_conv2d_fwd: OD_RES_UP2 needs even output size");
  const long long M64 = (long long)d->B * Ho * Wo;
  OD_REQUIRE(M64 * d->Cout < (1LL << 31) && (long long)d->B * d->H * d->W * d->Cin < (1LL << 31),
             "od_conv2d_fwd: tensor too large for 32-bit element offsets");
  const bool grouped = d->nseg > 1;  // (validated by the caller: 3x3, stride 1, plain epilogue; q.H / q.W = segment 0)
  long long Mg = 0;
  for (int i = 0; grouped && i < d->nseg; ++i) Mg += (long long)d->B * d->seg_H[i] * d->seg_W[i];
  OD_REQUIRE(!grouped || Mg * d->Cout < (1LL << 31), "od_conv2d_fwd: grouped launch too large for 32-bit element offsets");
  const int M = grouped ? (int)Mg : (int)M64;

  const bool want_stats = d->bn_partials != nullptr;
  if (want_stats)
    OD_REQUIRE(!tconv && d->out_dtype == OD_DT_F16 && d->act == OD_ACT_LINEAR && d->res_mode == OD_RES_NONE,
               "od_conv2d_fwd: bn_partials needs the raw convolution (f16 output, no activation, no residual, no transposed "
               "gather; scale / bias are NOT applied)");
  if (d->w2) {
    OD_REQUIRE(d->scale2 && d->bias2 && d->out2 && d->Cout2 > 0 && d->Cout2 % 8 == 0,
               "od_conv2d_fwd: w2 needs scale2, bias2, out2 and Cout2 (a multiple of 8)");
    OD_REQUIRE(d->act2 >= OD_ACT_LINEAR && d->act2 <= OD_ACT_ELU, "od_conv2d_fwd: bad act2");
    OD_REQUIRE(d->act2 != OD_ACT_LEAKY || (d->alpha2 >= 0.f && d->alpha2 <= 1.f), "od_conv2d_fwd: leaky slope (alpha2) must be in [0, 1]");
    OD_REQUIRE(d->out_dtype == OD_DT_F16 && !tconv && !want_stats && d->Cout % 8 == 0 &&
                   (d->out_batch_stride == 0 || d->out_batch_stride == (long long)Ho * Wo * d->Cout) &&
                   (d->out_pix_stride == 0 || d->out_pix_stride == d->Cout),
               "od_conv2d_fwd: w2 (the consuming pointwise layer) needs a dense f16 output of the first layer");
  }
  if (tconv && d->tile_cfg < 0 && od_tconv_small_supported(d)) {
    static int allow = -1;
    if (allow < 0) {
      const char* e = getenv("OD_TCONV_SMALL");  // 0 = the generic transposed path (A/B timing)
      allow = e ? atoi(e) : 1;
    }
    if (allow) return od_tconv_small_launch(ctx, d, stream, kernel_name, dry_run);
  }
  if (d->tile_cfg < 0 && od_conv_rdirect_supported(d)) {  // (also the transposed form of b.down2's backward-data)
    static int allow = -1;
    if (allow < 0) {
      const char* e = getenv("OD_CONV_RDIRECT");  // 0 = the table kernels (A/B timing)
      allow = e ? atoi(e) : 1;
    }
    if (allow) return od_conv_rdirect_launch(ctx, d, stream, kernel_name, dry_run);
  }
  if (!tconv && d->tile_cfg < 0 && od_conv_stream3_supported(d)) {
    static int allow = -1;
    if (allow < 0) {
      const char* e = getenv("OD_CONV_STREAM3");  // 0 = the table kernels (A/B timing)
      allow = e ? atoi(e) : 1;
    }
    if (allow) return od_conv_stream3_launch(ctx, d, stream, kernel_name, dry_run);
  }
  int cfg = d->tile_cfg;
  if (cfg < 0)  // (the 8-wave kernel has its own epilogue without the statistics path: not offered when they are asked for)
    cfg = pick_cfg(ctx, M, d->Cin, d->Cout, d->ksize,
                   !want_stats && !tconv && (long long)d->B * d->H * d->W * d->Cin * 2 < 0x7F000000LL, cfg == -2);
  if (want_stats && d->tile_cfg < 0) {
    const int var = d->ksize == 1 ? 0 : ((d->Cin % g_cfgs[cfg].BK) == 0 ? 1 : 2);
    if (!stats_kernel(cfg, var)) {  // same tile shape without the feature the table lacks, else the generic 128 x 128 / 64 x 128
      const int bm = g_cfgs[cfg].BM, bn = g_cfgs[cfg].BN;
      cfg = (d->Cin % 64 == 0) ? ((bm >= 128 && bn >= 128) ? 4 : (bn >= 128 ? 7 : 3)) : ((bm >= 128 && bn >= 128) ? 0 : (bn >= 128 ? 2 : (bm >= 128 ? 1 : 3)));
    }
  }
  const int cfg_e8 = kNumCfgs;
  OD_REQUIRE(cfg < cfg_e8 + od_conv_8ph_num_cfgs(), "od_conv2d_fwd: tile_cfg %d out of range", cfg);
  const bool use_e8 = cfg >= cfg_e8;
  if (grouped && !(use_e8 && d->Cin % 64 == 0)) {  // the table kernels have no segment table: one launch per segment
    if (fused_out) *fused_out = false;
    return OD_OK;
  }
  OD_REQUIRE(!(want_stats && use_e8), "od_conv2d_fwd: bn_partials is supported by the table kernels only (tile_cfg %d)", cfg);
  OD_REQUIRE(!tconv || !use_e8, "od_conv2d_fwd: transposed mode runs on the table kernels only (tile_cfg %d)", cfg);
  TileCfg tc = g_cfgs[use_e8 ? 0 : cfg];

  ConvKP p;
  p.x = (const f16*)d->x;
  p.w = (const f16*)d->w;
  p.scale = d->scale;
  p.bias = d->bias;
  p.res = (const f16*)d->res;
  p.out = d->out;
  p.zero = (const f16*)ctx->zero_page;
  p.H = Hv;
  p.W = Wv;
  p.tconv = tconv;
  p.Hs = d->H;
  p.Ws = d->W;
  p.Cin = d->Cin;
  p.Ho = Ho;
  p.Wo = Wo;
  p.Cout = d->Cout;
  p.stride = stride;
  p.pad = pad;
  p.Ktot = d->ksize * d->ksize * d->Cin;
  p.Kstride = od_round_up(p.Ktot, 64);
  p.M = M;
  p.HoWo = Ho * Wo;
  p.act = d->act;
  p.alpha = d->alpha;
  p.res_mode = d->res_mode;
  p.out_f32 = d->out_dtype == OD_DT_F32;
  p.stats = d->bn_partials;
  p.nseg = grouped ? d->nseg : 0;
  p.w2 = nullptr;  // set below when the selected kernel runs the consuming pointwise layer in its epilogue
  p.scale2 = d->scale2;
  p.bias2 = d->bias2;
  p.out2 = (f16*)d->out2;
  p.Cout2 = d->Cout2;
  p.act2 = d->act2;
  p.alpha2 = d->alpha2;
  p.K2stride = od_round_up(d->Cout, 64);
  p.w2_bytes = (unsigned)((long long)od_round_up(d->Cout2 > 0 ? d->Cout2 : 1, 256) * p.K2stride * 2);
  p.x_bytes = (unsigned)((long long)d->B * d->H * d->W * d->Cin * 2);
  p.w_bytes = (unsigned)((long long)od_round_up(d->Cout, 256) * p.Kstride * 2);
  p.obs = d->out_batch_stride ? d->out_batch_stride : (long long)p.HoWo * d->Cout;
  p.ops = d->out_pix_stride ? d->out_pix_stride : d->Cout;
  {
    static int dbg = -1;
    if (dbg < 0) {
      const char* e = getenv("OD_CONV_DEBUG");
      dbg = e ? atoi(e) : 0;
    }
    p.dbg = dbg;
  }
  ConvKernelInfo e8;
  if (use_e8) {
    // 8-wave / 256-wide schedule (conv_8ph.hip): same launch path (split-K slabs, finish kernel) as the table kernels
    size_t lds = 0;
    if (!od_conv_8ph_select(cfg - cfg_e8, p, d->ksize, &e8, &lds)) {
      od_set_error("od_conv2d_fwd: tile_cfg %d (8-phase kernel) needs Cin %% 64 == 0 and no transposed gather", cfg);
      return OD_ERR_INVALID;
    }
    tc.BM = e8.BM;
    tc.BN = e8.BN;
    tc.BK = 64;
    tc.threads = e8.threads;
    tc.lds = lds;
  }
  p.mtiles = od_ceil_div(M, tc.BM);
  if (grouped) {  // every segment's rows padded to whole m-tiles
    int t0 = 0;
    for (int i = 0; i < 3; ++i) {
      p.seg_tile0[i] = t0;
      p.seg_x[i] = nullptr;
      p.seg_out[i] = nullptr;
      p.seg_H[i] = p.seg_W[i] = p.seg_M[i] = 0;
      if (i < d->nseg) {
        p.seg_x[i] = (const f16*)d->seg_x[i];
        p.seg_out[i] = d->seg_out[i];
        p.seg_H[i] = d->seg_H[i];
        p.seg_W[i] = d->seg_W[i];
        p.seg_M[i] = d->B * d->seg_H[i] * d->seg_W[i];
        t0 += od_ceil_div(p.seg_M[i], tc.BM);
      }
    }
    p.seg_tile0[3] = t0;
    p.mtiles = t0;
    if (!d->out_batch_stride) p.obs = 0;  // dense outputs: the kernel takes every segment's own H * W * Cout
    if (fused_out) *fused_out = true;  // (the caller's "ran as one grouped launch" flag)
  }
  p.Mq = 0;
  if (tconv) {  // rows per parity class padded to whole tiles, classes interleaved tile by tile (od_tconv_pixel)
    p.Mq = M / 4;
    p.mtiles = 4 * od_ceil_div(p.Mq, tc.BM);
    p.M = p.mtiles * tc.BM;
  }
  p.ntiles = od_ceil_div(d->Cout, tc.BN);
  if (mtiles_out) *mtiles_out = p.mtiles;
  // split-K for layers that cannot fill the chip with output tiles (batch-1 inference): every K-range workgroup writes
  // its partial tile to its own slab of the caller's f32 workspace; splitk == 0 lets the library choose
  p.splitk = 1;
  p.steps_per_split = 0;
  p.ws = (float*)d->splitk_workspace;
  if (want_stats && !dry_run) {
    const long long need = (long long)p.mtiles * 2 * d->Cout * 4;
    if (d->bn_partials_bytes < need) {
      od_set_error("od_conv2d_fwd: bn_partials holds %lld bytes, %d rows x 2 x %d channels need %lld", (long long)d->bn_partials_bytes,
                   p.mtiles, d->Cout, need);
      return OD_ERR_WORKSPACE;
    }
  }
  if (d->splitk_workspace && d->splitk != 1 && !tconv && !want_stats && !grouped) {  // transposed mode orders its rows by parity class: no slabs
    const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
    const int tiles = p.mtiles * p.ntiles;
    const int nk = od_ceil_div(p.Ktot, tc.BK);
    static int thr_mul = -1, tgt_mul = -1;  // tuning knobs (OD_SPLITK="thr,tgt"): split when tiles*thr <= CUs, aim at tgt*CUs/2 workgroups
    if (thr_mul < 0) {
      thr_mul = 8;  // measured on MI355X (profiles/r01/splitk_sweep.txt): split only when <= CUs/8 tiles,
      tgt_mul = 1;  // aiming at ~CUs/2 workgroups
      const char* e = getenv("OD_SPLITK");
      if (e) sscanf(e, "%d,%d", &thr_mul, &tgt_mul);
    }
    int sk = d->splitk > 1 ? d->splitk : ((tiles * thr_mul <= cus && nk >= 8) ? od_ceil_div(tgt_mul * cus / 2, tiles) : 1);
    if (sk > nk / 4) sk = nk / 4;  // >= 4 K steps per workgroup
    const long long slab_bytes = (long long)M * d->Cout * 4;
    if ((long long)sk * slab_bytes > (long long)d->splitk_workspace_bytes) sk = (int)(d->splitk_workspace_bytes / slab_bytes);
    if (sk > 1) {
      p.steps_per_split = od_ceil_div(nk, sk);
      p.splitk = od_ceil_div(nk, p.steps_per_split);
    }
  }
  // the consuming pointwise layer inside the epilogue: 8-wave kernel, one n tile holding all 256 channels of a pixel, 128
  // output channels (W2 = 64 KiB of LDS), no split-K slabs
  if (d->w2 && use_e8 && p.splitk == 1 && d->Cout == 256 && d->Cout2 == 128) {
    static int allow = -1;
    if (allow < 0) {
      const char* e = getenv("OD_FUSE_POINTWISE");  // 0 = always two launches (A/B timing)
      allow = e ? atoi(e) : 1;
    }
    if (allow) {
      p.w2 = (const f16*)d->w2;
      size_t lds = 0;
      if (!od_conv_8ph_select(cfg - cfg_e8, p, d->ksize, &e8, &lds)) return OD_ERR_INVALID;
    }
  }
  if (fused_out) *fused_out = grouped || p.w2 != nullptr;
  // weights/scale/bias are padded to a multiple of 256 output channels, so any BN <= 256 tile stays in bounds.

  // kernel variant: 1x1 / 3x3-uniform-tap / 3x3-generic (odd channel counts fall back to a config that has one)
  int variant = d->ksize == 1 ? 0 : ((d->Cin % tc.BK) == 0 ? 1 : 2);
  if (!use_e8 && variant == 2 && !tc.k3g) {
    od_set_error("od_conv2d_fwd: tile_cfg %d needs Cin %% %d == 0 for 3x3 (Cin = %d); use cfg 0-3", cfg, tc.BK, d->Cin);
    return OD_ERR_INVALID;
  }
  const void* fn = use_e8 ? e8.fn : (variant == 0 ? tc.k1 : (variant == 1 ? tc.k3 : tc.k3g));
  if (want_stats) {
    fn = stats_kernel(cfg, variant);
    if (!fn) {
      od_set_error("od_conv2d_fwd: tile_cfg %d has no bn_partials instantiation for this layer (have: 0-7)", cfg);
      return OD_ERR_INVALID;
    }
  }
  if (kernel_name) *kernel_name = use_e8 ? e8.name : (variant == 0 ? tc.name1 : (variant == 1 ? tc.name3 : tc.name3g));
  if (dry_run) return OD_OK;

  if (int rc = od_ensure_lds(ctx, fn, tc.lds)) return rc;
  void* args[] = {&p};
  // the 8-wave kernel's epilogue needs no LDS unless it writes split-K slabs: ask only for the two K-tile buffers then
  // (128 KiB), which leaves room on the CU for a small workgroup of another stream
  const size_t launch_lds = (use_e8 && p.splitk <= 1 && tc.lds > (size_t)128 * 1024) ? (size_t)128 * 1024 : tc.lds;
  OD_CHECK_HIP(hipLaunchKernel(fn, dim3(p.mtiles * p.ntiles * p.splitk), dim3(tc.threads), args, launch_lds, stream));
  if (p.splitk > 1) {
    const long long nvec = (long long)p.M * (p.Cout / 8);
    long long fb = (nvec + 255) / 256;
    if (fb > 2048) fb = 2048;
    hipLaunchKernelGGL(od_conv_finish, dim3((unsigned)fb), dim3(256), 0, stream, p);
    OD_CHECK_LAUNCH();
  }
  return OD_OK;
}

extern "C" int od_conv2d_fwd(od_ctx* ctx, const od_conv_desc* d, void* stream) {
  return od_conv2d_fwd_impl(ctx, d, (hipStream_t)stream, nullptr, false);
}

extern "C" int od_conv2d_fwd_bn_rows(od_ctx* ctx, const od_conv_desc* d) {
  if (!ctx || !d) return -1;
  od_conv_desc q = *d;
  if (!q.bn_partials) q.bn_partials = (float*)(uintptr_t)16;  // dry run: only the tile choice matters
  int rows = 0;
  if (od_conv2d_fwd_rows_impl(ctx, &q, &rows) != OD_OK) return -1;
  return rows;
}
