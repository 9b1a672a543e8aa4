// K1+K2 fused: one Darknet53 residual block of the early stages in ONE kernel,
//     out = x + act(bn3(conv3x3(act(bn1(conv1x1(x))))))            x, out f16 [B,H,W,C]; C/2 middle channels,
// so that the C/2-channel tensor between the two convolutions never leaves the CU and x is read once (it is both the
// 1x1 input and the residual).  Stages 1-2 (C = 64, 128) are HBM/launch bound when run as two layers: 157 + 262 MB of
// traffic and two launches per block at stage 1 instead of 210 MB and one.
//
// One workgroup (8 waves) owns a 16x16-pixel output tile:
//   1. LDS-DMA: the 18x18 input window (zeros outside the image), the 1x1 weights and the 3x3 weights
//      (C = 64: all 9 taps resident, persistent workgroups with two window buffers; C = 128: tap 0, the other taps
//      stream through a 7-slot ring during step 3, see BneckCfg)
//   2. producer: t = act(s1 * (x_window . w1) + b1) for the 324 window pixels on MFMA (21 m-fragments over 8 waves),
//      rounded ONCE to f16 and written to the LDS window of t; window pixels outside the image are written as 0,
//      which is exactly the zero padding the 3x3 convolution sees in the unfused network
//   3. consumer: the 3x3 convolution reads its A fragments straight out of that window (fragment = one 16-pixel tile
//      row, shifted by the tap; taps column by column so that 6 fragments serve 3 taps), waves 4 (tile rows) x 2 (channels)
//   4. epilogue from the accumulators: v_permlane16_swap gives every lane 8 consecutive channels of its pixel;
//      scale/bias/activation in f32, + residual from the x window (LDS, or registers at C = 128), one rounding,
//      16-byte stores.
// LDS rows are chunk-swizzled on the DMA source side (conflict-free ds_read_b128 for every tap shift, brute-forced
// against the gfx950 lane-group bank model): 256-B rows chunk ^ ((row & 7) << 1), 128-B rows chunk ^ (row & 7),
// 64-B rows chunk ^ (3 * ((row >> 2) & 1)).
//
// Same rounding points as the two-layer path (f16 t, f16 out, f32 accumulation and epilogue arithmetic).
// Replaces two Conv2D + BatchNormalization + LeakyReLU layers and the Add of `ObjectDetector.predict`
// (reference voc_validate.py:27; docs/MODEL.md:15-17).
#include <stdlib.h>

#include "conv_common.h"

namespace {

struct BneckKP {
  const f16* x;
  const f16* w1;
  const float* s1;
  const float* b1;
  const f16* w3;
  const float* s3;
  const float* b3;
  f16* out;
  const f16* zero;
  int B, H, W;
  int k1stride, k3stride;
  int act;
  float alpha;
  int tiles_x, tiles_y;
};

template <int CPR>
static __device__ __forceinline__ int bn_swz(int row) {
  return CPR == 16 ? ((row & 7) << 1) : (CPR == 8 ? (row & 7) : 3 * ((row >> 2) & 1));
}

// The activation is a TEMPLATE parameter of the kernels: as a run-time enum tested per element (round 1) the compiler kept
// the dispatch as control flow inside the unrolled epilogues -- 328 branches and 40 v_exp_f32 in od_bneck<64>, the ELU path
// with its divergent v > 0 test laid out for every element even when the network only ever asks for LeakyReLU.
template <int ACT>
static __device__ __forceinline__ float bn_act(float v, float alpha) {
  if (ACT == OD_ACT_LEAKY) return od_leaky(v, alpha);
  if (ACT == OD_ACT_ELU) return v > 0.f ? v : alpha * od_expm1_fast(v);
  return v;
}

// buffer-addressed LDS-DMA (resource in SGPRs, per-lane byte offset, scalar byte offset); kept in a plain __device__
// function: the host pass of hipcc (ROCm 7.2) silently drops a kernel TEMPLATE whose body names the buffer builtins
static __device__ __forceinline__ __amdgpu_buffer_rsrc_t bn_make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
static __device__ __forceinline__ void bn_buffer_dma(__amdgpu_buffer_rsrc_t rs, int voff, int soff, char* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int C>
struct BneckCfg {
  static constexpr int CM = C / 2;
  static constexpr int RBX = 2 * C, RBT = C;          // row bytes of the x window / w1 and of the t window / w3 taps
  static constexpr int CPX = RBX / 16, CPT = RBT / 16;
  static constexpr int WW = 18, NWP = WW * WW, NWF = 21;  // window: 324 pixels = 21 m-fragments (last one ragged)
  static constexpr int XW_BYTES = NWF * 16 * RBX;
  static constexpr int TW_BYTES = NWP * RBT;
  static constexpr int W1_BYTES = CM * RBX;
  static constexpr int W3TAP = C * RBT;
  // C = 64: PERSISTENT workgroups, all weights resident, two x-window buffers (the next tile's window streams in while
  //         this tile is computed).
  // C = 128: one tile per workgroup; the 3x3 taps stream through a 7-slot ring: slot 0 behind the w1 region, slot 1 = the
  //         w1 region (free after the producer), slots 2-6 = the x window (free after the producer: the residual is
  //         prefetched from global memory into registers instead).
  static constexpr bool RESIDENT = (C == 64);
  static constexpr int NXW = RESIDENT ? 2 : 1;
  static constexpr int OFF_XW = 0, OFF_TW = NXW * XW_BYTES, OFF_W1 = OFF_TW + TW_BYTES;
  static constexpr int OFF_W3 = OFF_W1 + W1_BYTES;  // resident: 9 taps; streamed: ring slot 0
  static constexpr int LDS_BYTES = RESIDENT ? OFF_W3 + 9 * W3TAP : OFF_W3 + W3TAP;
  static constexpr int NSLOT = 7;
  static_assert(RESIDENT || (W3TAP == W1_BYTES && 5 * W3TAP <= XW_BYTES), "ring slots must fit the freed regions");
  static constexpr int NF1 = CM / 16, KS1 = C / 32;  // producer: n-fragments, k-steps
  static constexpr int NF3 = C / 16, KS3 = CM / 32;  // consumer: n-fragments, k-steps per tap
  static constexpr int NFW = NF3 / 2;                // n-fragments per consumer wave
  static constexpr int TAP_DMAS = C * CPT / 512;     // LDS-DMA instructions per thread per tap (streamed)
  static __host__ __device__ constexpr int slot_off(int slot) {
    return slot == 0 ? OFF_W3 : (slot == 1 ? OFF_W1 : OFF_XW + (slot - 2) * W3TAP);
  }
};

// DBG = 1 (OD_CONV_DEBUG=32): s_memtime stamps of the 6th tile of workgroup 0, waves 0 and 5 (od_debug_bneck_stamps)
__device__ unsigned long long g_bn_stamps[2][8];

// C = 64 runs 12 waves: waves 0-7 compute, waves 8-11 only issue the NEXT tile's x-window LDS-DMAs (and wait for them).
// Stamps of the 8-wave version: producer 3.8 k, window DMA issue 4.8 k (the issuing waves stall while the memory pipeline
// is full: 41 KB per tile at the CU's HBM share), consumer 2.0 k, epilogue 4.5 k cycles per tile, all in series; with
// loader waves the 4.8 k run beside the other 10.3 k.
template <int C, int DBG = 0, int ACT = OD_ACT_LEAKY>
__global__ __launch_bounds__(C == 64 ? 768 : 512, C == 64 ? 3 : 2) void od_bneck(BneckKP p, int ntiles) {
  using Cf = BneckCfg<C>;
  constexpr bool LOADERS = (C == 64);
  constexpr int CM = Cf::CM, RBX = Cf::RBX, RBT = Cf::RBT, CPX = Cf::CPX, CPT = Cf::CPT, WW = Cf::WW, NWP = Cf::NWP;
  constexpr int NF1 = Cf::NF1, KS1 = Cf::KS1, KS3 = Cf::KS3, NFW = Cf::NFW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int tpi = p.tiles_x * p.tiles_y;

  // ---- LDS-DMA helpers --------------------------------------------------------------------------------------
  // x-window DMA: chunk q of the window image -> (window pixel, 16-byte chunk) is the same for every tile; only the tile
  // origin changes.  rel = element offset relative to the tile's first pixel, wyx = (wy << 8) | wx for the bounds test.
  constexpr int XWT = LOADERS ? 256 : 512;  // threads that issue the window DMAs
  constexpr int XWR = (NWP * CPX + XWT - 1) / XWT;
  const int xtid = LOADERS ? tid - 512 : tid, xwave = LOADERS ? wave - 8 : wave;
  int xw_rel[XWR], xw_wyx[XWR];
#pragma unroll
  for (int r = 0; r < XWR; ++r) {
    const int q = r * XWT + xtid;
    const int wp = q / CPX, pc = q - wp * CPX;
    const int wy = wp / WW, wx = wp - wy * WW;
    xw_rel[r] = ((wy - 1) * p.W + (wx - 1)) * C + (pc ^ bn_swz<CPX>(wp)) * 8;
    xw_wyx[r] = (q >= 0 && q < NWP * CPX && xtid >= 0) ? ((wy << 8) | wx) : -1;
  }
  auto issue_xwin = [&](int tile, int buf) {
    const int b = tile / tpi;
    const int trem = tile - b * tpi;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int y0 = tyi * 16, x0 = txi * 16;
    const f16* org = p.x + ((long long)(b * p.H + y0) * p.W + x0) * C;
#pragma unroll
    for (int r = 0; r < XWR; ++r) {
      if (xw_wyx[r] >= 0) {
        const int y = y0 - 1 + (xw_wyx[r] >> 8), x = x0 - 1 + (xw_wyx[r] & 255);
        const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const f16* src = ok ? org + xw_rel[r] : p.zero;
        glds16(src, smem + Cf::OFF_XW + buf * Cf::XW_BYTES + (r * XWT + xwave * 64) * 16);
      }
    }
  };
  // 3x3 taps: buffer-addressed LDS-DMA (resource in SGPRs, per-lane row offset fixed, tap = scalar offset)
  constexpr int TAPR = (C * CPT + 511) / 512;
  int tap_voff[TAPR];
#pragma unroll
  for (int r = 0; r < TAPR; ++r) {
    const int q = r * 512 + tid;
    const int n = q / CPT, pc = q - n * CPT;
    tap_voff[r] = q < C * CPT ? (n * p.k3stride + (pc ^ bn_swz<CPT>(n)) * 8) * 2 : (int)0x80000000;
  }
  const __amdgpu_buffer_rsrc_t rs_w3 = bn_make_rsrc(p.w3, 256u * (unsigned)p.k3stride * 2u);
  auto load_tap = [&](int tap, int dst_off) {
#pragma unroll
    for (int r = 0; r < TAPR; ++r) {
      char* dst = smem + dst_off + (r * 512 + wave * 64) * 16;
      if ((r + 1) * 512 <= C * CPT) {  // compile-time: every lane of every wave takes part
        bn_buffer_dma(rs_w3, tap_voff[r], tap * CM * 2, dst);
      } else if (r * 512 + wave * 64 < C * CPT) {  // ragged last round: whole waves drop out (C * CPT % 64 == 0)
        bn_buffer_dma(rs_w3, tap_voff[r], tap * CM * 2, dst);
      }
    }
  };

  // ---- once per workgroup: weights ------------------------------------------------------------------------------
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  if (LOADERS && wave >= 8) {
    // ---- loader waves: the window of tile i+1 streams into the other buffer while the compute waves work on tile i.
    // Buffer cur^1 was last read in tile i-1's epilogue, i.e. before that tile's closing barrier.
    issue_xwin(tile, 0);
    wait_vmcnt<0>();
    __syncthreads();  // (prologue barrier of the compute waves)
    int curl = 0;
#pragma unroll 1
    for (; tile < ntiles; tile += (int)gridDim.x) {
      if (tile + (int)gridDim.x < ntiles) issue_xwin(tile + (int)gridDim.x, curl ^ 1);
      __builtin_amdgcn_s_barrier();  // A: t window complete (raw: __syncthreads() would first drain this wave's DMAs and
                                     //    hold the compute waves at A until the next window has landed)
      wait_vmcnt<0>();               // the next window has landed
      __builtin_amdgcn_s_barrier();  // B: tile done
      curl ^= 1;
    }
    return;
  }
  if (!LOADERS) issue_xwin(tile, 0);
#pragma unroll 1
  for (int q = tid; q < CM * CPX + 63; q += 512) {
    if (q < CM * CPX) {
      const int n = q / CPX, pc = q - n * CPX;
      const int lc = pc ^ bn_swz<CPX>(n);
      glds16(p.w1 + (long long)n * p.k1stride + lc * 8, smem + Cf::OFF_W1 + (q - lane) * 16);
    }
  }
  if (Cf::RESIDENT) {
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) load_tap(tap, Cf::OFF_W3 + tap * Cf::W3TAP);
  } else {
    load_tap(0, Cf::slot_off(0));
  }
  wait_vmcnt<0>();
  __syncthreads();

  const int wm = wave >> 1, wn = wave & 1;
  int boff[NFW][KS3];  // per-lane byte offsets of the consumer's B fragments inside one tap image
#pragma unroll
  for (int j = 0; j < NFW; ++j)
#pragma unroll
    for (int ks = 0; ks < KS3; ++ks) {
      const int row = (wn * NFW + j) * 16 + l15;
      boff[j][ks] = row * RBT + (((ks * 4 + lq) ^ bn_swz<CPT>(row)) * 16);
    }

  int cur = 0;
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int tcount = 0;
#define BN_STAMP(k)                                                                              \
  do {                                                                                           \
    if (DBG) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[k])::"memory");      \
  } while (0)
#pragma unroll 1
  for (; tile < ntiles; tile += (int)gridDim.x) {
    BN_STAMP(0);
    const int b = tile / tpi;
    const int trem = tile - b * tpi;
    const int tyi = trem / p.tiles_x, txi = trem - tyi * p.tiles_x;
    const int y0 = tyi * 16, x0 = txi * 16;
    const char* xw = smem + Cf::OFF_XW + cur * Cf::XW_BYTES;

    // ---- producer: t window ---------------------------------------------------------------------------------
    {
      // producer operands (reloaded per tile so that they are dead during the consumer): 1x1 weights, scale / bias
    f16x8 wb1[NF1][KS1];
  #pragma unroll
    for (int j = 0; j < NF1; ++j)
  #pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const int row = j * 16 + l15;
        wb1[j][ks] = *(const f16x8*)(smem + Cf::OFF_W1 + row * RBX + (((ks * 4 + lq) ^ bn_swz<CPX>(row)) * 16));
      }
    float sc1[NF1][4], bi1[NF1][4];
  #pragma unroll
    for (int j = 0; j < NF1; ++j) {
      const f32x4 s = *(const f32x4*)(p.s1 + j * 16 + lq * 4), bb = *(const f32x4*)(p.b1 + j * 16 + lq * 4);
  #pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc1[j][e] = s[e];
        bi1[j][e] = bb[e];
      }
    }
  #pragma unroll 1
      for (int f = wave; f < Cf::NWF; f += 8) {
        const int wp = f * 16 + l15;  // this lane's window pixel (rows >= 324 of the last fragment are scratch)
        f32x4 acc1[NF1];
  #pragma unroll
        for (int j = 0; j < NF1; ++j) acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  #pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
          const f16x8 xa = *(const f16x8*)(xw + wp * RBX + (((ks * 4 + lq) ^ bn_swz<CPX>(wp)) * 16));
  #pragma unroll
          for (int j = 0; j < NF1; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1[j][ks], xa, acc1[j], 0, 0, 0);
        }
        if (wp < NWP) {
          const int wy = wp / WW, wx = wp - wy * WW;
          const bool inside = (unsigned)(y0 - 1 + wy) < (unsigned)p.H && (unsigned)(x0 - 1 + wx) < (unsigned)p.W;
  #pragma unroll
          for (int j = 0; j < NF1; ++j) {
            f16x4 h;
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = bn_act<ACT>(acc1[j][e] * sc1[j][e] + bi1[j][e], p.alpha);
              h[e] = inside ? (f16)v : (f16)0.f;
            }
            const int lc = j * 2 + (lq >> 1);
            *(f16x4*)(smem + Cf::OFF_TW + wp * RBT + ((lc ^ bn_swz<CPT>(wp)) * 16) + (lq & 1) * 8) = h;
          }
        }
      }

    }

    // residual (streamed variant): the x window is about to be recycled as ring slots -> centre pixels to registers
    f16x8 resv[NFW / 2][4];
    if (!Cf::RESIDENT) {
#pragma unroll
      for (int pr = 0; pr < NFW / 2; ++pr) {
        const int ch = (wn * NFW + 2 * pr + (lq & 1)) * 16 + (lq >> 1) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int wrow = (wm * 4 + i + 1) * WW + 1 + l15;
          resv[pr][i] = *(const f16x8*)(xw + wrow * RBX + (((ch >> 3) ^ bn_swz<CPX>(wrow)) * 16));
        }
      }
    }
    BN_STAMP(1);
    __syncthreads();  // t window complete; w1 region and (streamed) x window free
    BN_STAMP(2);

    if (Cf::RESIDENT) {
      if (!LOADERS && tile + (int)gridDim.x < ntiles) issue_xwin(tile + (int)gridDim.x, cur ^ 1);  // next tile's window
    } else {
#pragma unroll
      for (int k = 1; k <= 6; ++k) load_tap((k % 3) * 3 + k / 3, Cf::slot_off(k));  // consumption order, see below
    }

    BN_STAMP(3);
    // ---- consumer: 3x3 from the t window -----------------------------------------------------------------------
    f32x4 acc[4][NFW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NFW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Taps are consumed column by column (dx outer, dy inner): the 6 A fragments of one window column (tile rows
    // wm*4 - 1 .. wm*4 + 4, shifted by dx) are read ONCE and reused by the three dy taps.
    f16x8 xa[6][KS3];
    auto load_col = [&](int dx) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int wrow = (wm * 4 + k) * WW + dx + l15;
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks)
          xa[k][ks] = *(const f16x8*)(smem + Cf::OFF_TW + wrow * RBT + (((ks * 4 + lq) ^ bn_swz<CPT>(wrow)) * 16));
      }
    };
    auto do_tap = [&](int dy, int wbase) {
#pragma unroll
      for (int ks = 0; ks < KS3; ++ks) {
        f16x8 wb[NFW];
#pragma unroll
        for (int j = 0; j < NFW; ++j) wb[j] = *(const f16x8*)(smem + wbase + boff[j][ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NFW; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i + dy][ks], acc[i][j], 0, 0, 0);
      }
    };
    if (Cf::RESIDENT) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        load_col(dx);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) do_tap(dy, Cf::OFF_W3 + (dy * 3 + dx) * Cf::W3TAP);
      }
    } else {
      // consumption order k = dx*3 + dy holds tap (dy*3 + dx).  k = 0 landed before the producer, k = 1..6 are in
      // flight (issued above, in order, slots 1..6); k = 7 / 8 go into slots 0 / 1 once k = 0 / 1 are consumed.  Every
      // counted wait leaves exactly the DMAs issued AFTER the needed tap in flight.
      constexpr int D = Cf::TAP_DMAS;
      load_col(0);
      do_tap(0, Cf::slot_off(0));            // k = 0
      wait_vmcnt<5 * D>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      load_tap(1 * 3 + 2, Cf::slot_off(0));  // k = 7: (dx 2, dy 1)
      do_tap(1, Cf::slot_off(1));            // k = 1
      wait_vmcnt<5 * D>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      load_tap(2 * 3 + 2, Cf::slot_off(1));  // k = 8: (dx 2, dy 2)
      do_tap(2, Cf::slot_off(2));            // k = 2
      wait_vmcnt<5 * D>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      load_col(1);
      do_tap(0, Cf::slot_off(3));            // k = 3
      wait_vmcnt<4 * D>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      do_tap(1, Cf::slot_off(4));            // k = 4
      wait_vmcnt<3 * D>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      do_tap(2, Cf::slot_off(5));            // k = 5
      wait_vmcnt<2 * D>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      load_col(2);
      do_tap(0, Cf::slot_off(6));            // k = 6
      wait_vmcnt<1 * D>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      do_tap(1, Cf::slot_off(0));            // k = 7
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();  // raw: __syncthreads() would drain every DMA in flight
      do_tap(2, Cf::slot_off(1));            // k = 8
    }

    BN_STAMP(4);
    od_mfma_results_ready();
    // ---- epilogue: 8 consecutive channels per lane (permlane16 swap of a fragment pair) + residual ----------------
#pragma unroll
    for (int pr = 0; pr < NFW / 2; ++pr) {
      const int ch = (wn * NFW + 2 * pr + (lq & 1)) * 16 + (lq >> 1) * 8;
      float sc[8], bi[8];
      {
        const f32x4 s0 = *(const f32x4*)(p.s3 + ch), s1 = *(const f32x4*)(p.s3 + ch + 4);
        const f32x4 b0 = *(const f32x4*)(p.b3 + ch), b1 = *(const f32x4*)(p.b3 + ch + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sc[e] = s0[e];
          sc[4 + e] = s1[e];
          bi[e] = b0[e];
          bi[4 + e] = b1[e];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ty = wm * 4 + i;
        f16x8 r;
        if (Cf::RESIDENT) {
          const int wrow = (ty + 1) * WW + 1 + l15;  // centre pixel of the x window
          r = *(const f16x8*)(xw + wrow * RBX + (((ch >> 3) ^ bn_swz<CPX>(wrow)) * 16));
        } else {
          r = resv[pr][i];
        }
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = acc[i][2 * pr][e], bq = acc[i][2 * pr + 1][e];
          od_permlane16_swap(a, bq);
          v[e] = a;
          v[4 + e] = bq;
        }
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)(bn_act<ACT>(v[e] * sc[e] + bi[e], p.alpha) + (float)r[e]);
        *(f16x8*)(p.out + ((long long)(b * p.H + y0 + ty) * p.W + x0 + l15) * C + ch) = h;
      }
    }
    BN_STAMP(5);
    if (Cf::RESIDENT) {
      // the next tile's window was issued BEFORE this tile's (NFW/2)*4 output stores: a counted wait retires the DMA
      // and leaves the stores in flight (vmcnt retires in issue order)
      if (!LOADERS) wait_vmcnt<(NFW / 2) * 4>();  // (with loader waves the compute waves have no DMA of their own in flight)
      BN_STAMP(6);
      __syncthreads();   // everyone is done with the t window and with this tile's x window
      cur ^= 1;
    }
    BN_STAMP(7);
    if (DBG && blockIdx.x == (Cf::RESIDENT ? 0 : 300) && (wave == 0 || wave == 5) && (++tcount == 6 || !Cf::RESIDENT) && lane == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) g_bn_stamps[wave ? 1 : 0][k] = st[k];
    }
  }
}

}  // namespace

extern "C" int od_bottleneck_supported(int H, int W, int C) {
  return (C == 64 || C == 128) && H > 0 && W > 0 && (H % 16) == 0 && (W % 16) == 0;
}

const char* od_bottleneck_kernel_name(int C) { return C == 64 ? "od_bneck<64, 0, 1>" : "od_bneck<128, 0, 1>"; }

extern "C" int od_bottleneck_fwd(od_ctx* ctx, const od_bneck_desc* d, void* stream) {
  OD_REQUIRE(ctx && d, "od_bottleneck_fwd: null ctx/desc");
  OD_REQUIRE(d->x && d->w1 && d->scale1 && d->bias1 && d->w3 && d->scale3 && d->bias3 && d->out,
             "od_bottleneck_fwd: null tensor");
  OD_REQUIRE(od_bottleneck_supported(d->H, d->W, d->C),
             "od_bottleneck_fwd: needs C in {64, 128} and H, W multiples of 16 (got C=%d, %dx%d)", d->C, d->H, d->W);
  OD_REQUIRE(d->B > 0 && (long long)d->B * d->H * d->W * d->C < (1LL << 31), "od_bottleneck_fwd: bad batch / tensor too large");
  OD_REQUIRE(d->act >= OD_ACT_LINEAR && d->act <= OD_ACT_ELU, "od_bottleneck_fwd: bad act");
  OD_REQUIRE(d->act != OD_ACT_LEAKY || (d->alpha >= 0.f && d->alpha <= 1.f), "od_bottleneck_fwd: leaky slope must be in [0, 1]");
  BneckKP p;
  p.x = (const f16*)d->x;
  p.w1 = (const f16*)d->w1;
  p.s1 = d->scale1;
  p.b1 = d->bias1;
  p.w3 = (const f16*)d->w3;
  p.s3 = d->scale3;
  p.b3 = d->bias3;
  p.out = (f16*)d->out;
  p.zero = (const f16*)ctx->zero_page;
  p.B = d->B;
  p.H = d->H;
  p.W = d->W;
  p.k1stride = od_round_up(d->C, 64);
  p.k3stride = od_round_up(9 * (d->C / 2), 64);
  p.act = d->act;
  p.alpha = d->alpha;
  p.tiles_x = d->W / 16;
  p.tiles_y = d->H / 16;
  static int dbg = -1;
  if (dbg < 0) {
    const char* e = getenv("OD_CONV_DEBUG");
    dbg = e ? atoi(e) : 0;
  }
  const void* fn;
  if (d->C == 64) {
    fn = dbg == 32 && d->act == OD_ACT_LEAKY ? (const void*)&od_bneck<64, 1, OD_ACT_LEAKY>
         : d->act == OD_ACT_LEAKY            ? (const void*)&od_bneck<64, 0, OD_ACT_LEAKY>
         : d->act == OD_ACT_ELU              ? (const void*)&od_bneck<64, 0, OD_ACT_ELU>
                                             : (const void*)&od_bneck<64, 0, OD_ACT_LINEAR>;
  } else {
    fn = dbg == 33 && d->act == OD_ACT_LEAKY ? (const void*)&od_bneck<128, 1, OD_ACT_LEAKY>
         : d->act == OD_ACT_LEAKY ? (const void*)&od_bneck<128, 0, OD_ACT_LEAKY>
         : d->act == OD_ACT_ELU ? (const void*)&od_bneck<128, 0, OD_ACT_ELU>
                                : (const void*)&od_bneck<128, 0, OD_ACT_LINEAR>;
  }
  const int lds = d->C == 64 ? BneckCfg<64>::LDS_BYTES : BneckCfg<128>::LDS_BYTES;
  if (int rc = od_ensure_lds(ctx, fn, (size_t)lds)) return rc;
  int ntiles = d->B * p.tiles_x * p.tiles_y;
  // C = 64: persistent workgroups (one per CU) walk the tiles; C = 128: one workgroup per tile
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  const int grid = d->C == 64 ? (ntiles < cus ? ntiles : cus) : ntiles;
  void* args[] = {&p, &ntiles};
  OD_CHECK_HIP(hipLaunchKernel(fn, dim3((unsigned)grid), dim3(d->C == 64 ? 768 : 512), args, (size_t)lds, (hipStream_t)stream));
  return OD_OK;
}

// debug only (not part of include/odhip.h)
extern "C" int od_debug_bneck_stamps(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_bn_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
