// K1 (3x3 / stride 1): direct convolution from an LDS-resident input WINDOW -- the nine filter taps re-read the same
// staged pixels out of LDS instead of re-gathering them from L2 nine times (the implicit-GEMM kernel of conv_mfma.hip is
// L2->LDS bandwidth bound: 32 KiB of DMA per 2.1 MFLOP; this kernel moves ~1/3 of that).
//
// Tile = BM consecutive output pixels of the flattened (b, y, x) order x BN output channels (same M tiling as the
// implicit-GEMM kernel, so small feature maps waste nothing).  Because the layout is NHWC, every input pixel a tile
// needs lies in the flattened range [m0 - W - 1, m0 + BM + W + 1): tap (dy, dx) of output pixel m is input pixel
// m + (dy-1)*W + (dx-1).  For one 64-channel slice that range (BM + 2W + 2 pixels x 128 B) is DMA'd into LDS ONCE and
// serves 9 K-steps; taps that fall outside the image (left/right/top/bottom edges, tile tails) are redirected per
// lane to a zero slot.  Per (slice, tap) step only the BN x 64 weight slice is streamed (3-deep ring, counted vmcnt);
// the next slice's window streams in behind the weights, 1/8 per step, into the other window buffer.
//
// LDS window image: 16-pixel pieces, chunk-major [8 chunks][16 pixels][16 B]: a ds_read_b128 fragment read of 16
// consecutive pixels is bank-conflict free for ANY tap shift, with no swizzle arithmetic, and pixel +16 = +2048 B
// (an immediate offset).  Weights: 128-B rows, chunk c of row r at c ^ (r & 7), as in conv_mfma.hip.
//
// Replaces the 3x3 Conv2D + BatchNormalization + LeakyReLU/ELU (+ Add) layers of `ObjectDetector.predict`
// (reference voc_validate.py:27; docs/MODEL.md:5-21).
#include "conv_common.h"

namespace {

constexpr int PIECE_BYTES = 2048;  // 16 pixels x 64 channels x 2 B
constexpr int WSTAGES = 3;         // weight ring depth
constexpr int WIN_STEPS = 8;       // the next window is issued during taps 0..7 of the current slice

template <int BM, int BN, int WM, int WN>
struct WinCfg {
  static constexpr int NT = WM * WN * 64, NW = WM * WN;
  static constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NTL = WTN / 16;
  static constexpr int WROWS = NT / 8;           // weight rows per DMA round
  static constexpr int BR = BN / WROWS;          // weight DMA rounds = loads per wave per step
  static constexpr int KW = (NW >= 8) ? 1 : 2;   // window DMA slots per wave per step
  static constexpr int MAX_WIN_INSTR = WIN_STEPS * NW * KW;  // 1-KiB DMA instructions available for one window
  static constexpr int WSTAGE_BYTES = BN * 128;
  static constexpr int EPI_BYTES = WTM * (BN + 4) * 4;
  static_assert(BN % WROWS == 0, "BN must be a whole number of weight DMA rounds");
};

template <int BM, int BN, int WM, int WN, int MINW>
__global__ __launch_bounds__(WM* WN * 64, MINW) void od_conv3x3_win(ConvKP p, int np) {
  using Cf = WinCfg<BM, BN, WM, WN>;
  constexpr int NW = Cf::NW, WTM = Cf::WTM, WTN = Cf::WTN, MT = Cf::MT, NTL = Cf::NTL;
  constexpr int BR = Cf::BR, KW = Cf::KW, WROWS = Cf::WROWS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS map: [window 0][window 1][weight ring][zero/dummy piece (1 KiB)]
  const int win_bytes = np * PIECE_BYTES;
  char* const wring = smem + 2 * win_bytes;
  char* const zpiece = wring + WSTAGES * Cf::WSTAGE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  int logical;
  {
    const int nt = p.mtiles * p.ntiles;
    const int pid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, loc = pid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = logical / p.ntiles, tn = logical - tm * p.ntiles;
  const int m0 = tm * BM, n0 = tn * BN;
  const int W = p.W, Cin = p.Cin;
  const int mwin0 = m0 - W - 1;  // flattened input pixel held by window row 0

  // zero slot (also the landing pad of dummy DMA, which only ever writes zeros)
  *(f32x4*)(zpiece + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- DMA helpers ------------------------------------------------------------------------------------------
  // window instruction q (1 KiB): piece q>>1, half q&1; lane = (pixel lane&15, chunk (q&1)*4 + lane>>4)
  const int nwi = np * 2;
  auto win_load = [&](int q, int slice, int wbuf) {
    const bool real = q < nwi;
    const int piece = q >> 1, half = q & 1;
    const int mp = mwin0 + piece * 16 + (lane & 15);
    const bool ok = real && (unsigned)mp < (unsigned)p.M;
    const f16* src = ok ? p.x + ((long long)mp * Cin + slice * 64 + (half * 4 + (lane >> 4)) * 8) : p.zero;
    char* dst = real ? smem + wbuf * win_bytes + piece * PIECE_BYTES + half * 1024 : zpiece;
    glds16(src, dst);
  };
  // weight slice of step (slice, tap): rows n0.., k = tap*Cin + slice*64
  const int wrr = tid >> 3;
  const int wlc = (tid & 7) ^ (wrr & 7);
  const f16* wrow = p.w + (long long)(n0 + wrr) * p.Kstride + wlc * 8;
  auto w_load = [&](int k0, int stage, bool real) {
    char* base = wring + stage * Cf::WSTAGE_BYTES + wave * 8 * 128;
#pragma unroll
    for (int rd = 0; rd < BR; ++rd) {
      const f16* src = real ? wrow + (long long)rd * WROWS * p.Kstride + k0 : p.zero;
      char* dst = real ? base + rd * WROWS * 128 : zpiece;
      glds16(src, dst);
    }
  };

  // ---- per-lane MFMA-side pixel state -----------------------------------------------------------------------
  const int wm = wave / WN, wn = wave - wm * WN;
  unsigned vmask[MT];  // bit t: tap t of this lane's pixel of m-tile i reads inside the image
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * WTM + i * 16 + l15;
    vmask[i] = 0u;
    if (m < p.M) {
      const unsigned b = (unsigned)m / (unsigned)p.HoWo;
      const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
      const int y = (int)(pix / (unsigned)W), x = (int)(pix - (pix / (unsigned)W) * (unsigned)W);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)W) vmask[i] |= 1u << t;
      }
    }
  }
  const int zaddr = (int)(zpiece - smem);
  const int r_lane = wm * WTM + l15;  // window row of m-tile 0 for tap (0,0)

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: window of slice 0, weights of steps 0 and 1 --------------------------------------------------
  const int nslices = Cin >> 6;
  const int nsteps = nslices * 9;
  for (int q = wave; q < nwi; q += NW) win_load(q, 0, 0);
  w_load(0, 0, true);
  w_load(nsteps > 1 ? Cin : 0, 1, nsteps > 1);  // step 1 = (slice 0, tap 1): k0 = Cin
  wait_vmcnt<0>();
  __syncthreads();

  // loader state = step s + 2
  int ld_tap = 2, ld_slice = 0, ld_stage = 2;
  int tap = 0, slice = 0, toff = 0, dxc = 0, wstage = 0, wbuf = 0;
  const int swz = l15 & 7;
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<BR + KW>();            // everything issued two steps ago (weights of THIS step, a window part) landed
    __builtin_amdgcn_s_barrier();     // ... for every wave; and everyone is done reading what gets overwritten now
    if (!(p.dbg & 1)) {
      const bool real = (s + 2) < nsteps;
      w_load(ld_tap * Cin + ld_slice * 64, ld_stage, real);
      const bool wreal = tap < WIN_STEPS && (slice + 1) < nslices;
#pragma unroll
      for (int kk = 0; kk < KW; ++kk) {
        const int q = wreal ? (tap * NW + wave) * KW + kk : nwi;
        win_load(q, slice + 1, wbuf ^ 1);
      }
      if (++ld_tap == 9) {
        ld_tap = 0;
        ++ld_slice;
      }
      if (++ld_stage == WSTAGES) ld_stage = 0;
    }
    const char* wbase = wring + wstage * Cf::WSTAGE_BYTES;
    const char* xbase = smem + wbuf * win_bytes;
    const int r0 = r_lane + toff;  // this lane's window row for m-tile 0 at this tap
    const int xa0 = (r0 >> 4) * PIECE_BYTES + (r0 & 15) * 16 + lq * 256;
    __builtin_amdgcn_s_setprio(1);
    if (!(p.dbg & 2))
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      f16x8 xa[MT], wb[NTL];
      const int coff = ((kh * 4 + lq) ^ swz) * 16;
#pragma unroll
      for (int j = 0; j < NTL; ++j) wb[j] = *(const f16x8*)(wbase + (wn * WTN + j * 16 + l15) * 128 + coff);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const bool ok = (vmask[i] >> tap) & 1u;
        const char* a = ok ? xbase + (xa0 + i * PIECE_BYTES + kh * 1024) : smem + zaddr;
        xa[i] = *(const f16x8*)a;
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    // advance compute state
    if (++wstage == WSTAGES) wstage = 0;
    if (++tap == 9) {
      tap = 0;
      toff = 0;
      dxc = 0;
      ++slice;
      wbuf ^= 1;
    } else if (++dxc == 3) {
      dxc = 0;
      toff += W - 2;
    } else {
      ++toff;
    }
  }
  wait_vmcnt<0>();
  __syncthreads();  // ring / windows are reused as epilogue staging
  conv_epilogue<BN, WM, WN, MT, NTL>(p, smem, acc, m0, n0, tid, wm, wn, l15, lq);
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised form: 4 MFMA waves (2x2 over the BM x BN tile) + 4 DMA waves per workgroup, sized so that TWO
// workgroups share a CU (<= 80 KiB LDS each): the two workgroups drift apart and fill each other's barrier / LDS-latency
// gaps, and the MFMA waves never pay the ~70-cycle issue cost of an LDS-DMA instruction.
//   NWBUF = 2: the next slice's window streams in behind the weights (as above)
//   NWBUF = 1: one window buffer; at a slice change the DMA waves reload it between two barriers (the other workgroup
//              of the CU covers the gap)
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WS, int NWBUF>
struct WinSpecCfg {
  static constexpr int WTM = BM / 2, WTN = BN / 2, MT = WTM / 16, NTL = WTN / 16;
  static constexpr int BR = BN / 32;                 // weight DMA instructions per loader wave per step
  static constexpr int KW = (NWBUF == 2) ? 2 : 0;    // window DMA slots per loader wave per step
  static constexpr int L = BR + KW;
  static constexpr int MAX_WIN_INSTR = (NWBUF == 2) ? WIN_STEPS * 4 * KW : (1 << 20);
  static constexpr int WSTAGE_BYTES = BN * 128;
  static constexpr int EPI_BYTES = WTM * (BN + 4) * 4;
};

template <int BM, int BN, int WS, int NWBUF, int MINW>
__global__ __launch_bounds__(512, MINW) void od_conv3x3_wins(ConvKP p, int np) {
  using Cf = WinSpecCfg<BM, BN, WS, NWBUF>;
  constexpr int WTM = Cf::WTM, WTN = Cf::WTN, MT = Cf::MT, NTL = Cf::NTL, BR = Cf::BR, KW = Cf::KW;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int win_bytes = np * PIECE_BYTES;
  char* const wring = smem + NWBUF * win_bytes;
  char* const zpiece = wring + WS * Cf::WSTAGE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave_all >= 4;
  const int wave = is_loader ? wave_all - 4 : wave_all;
  const int ltid = tid & 255;
  const int l15 = lane & 15, lq = lane >> 4;

  int logical;
  {
    const int nt = p.mtiles * p.ntiles;
    const int pid = blockIdx.x;
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, loc = pid >> 3;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tm = logical / p.ntiles, tn = logical - tm * p.ntiles;
  const int m0 = tm * BM, n0 = tn * BN;
  const int W = p.W, Cin = p.Cin;
  const int mwin0 = m0 - W - 1;
  const int nwi = np * 2;
  const int nslices = Cin >> 6;
  const int nsteps = nslices * 9;

  if (wave_all == 0) *(f32x4*)(zpiece + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  const int wm = wave >> 1, wn = wave & 1;
  f32x4 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (is_loader) {
    // =========================================== DMA waves ===========================================
    auto win_load = [&](int q, int slice, int wbuf) {
      const bool real = q < nwi;
      const int piece = q >> 1, half = q & 1;
      const int mp = mwin0 + piece * 16 + (lane & 15);
      const bool ok = real && (unsigned)mp < (unsigned)p.M;
      const f16* src = ok ? p.x + ((long long)mp * Cin + slice * 64 + (half * 4 + (lane >> 4)) * 8) : p.zero;
      char* dst = real ? smem + wbuf * win_bytes + piece * PIECE_BYTES + half * 1024 : zpiece;
      glds16(src, dst);
    };
    const int wrr = ltid >> 3;
    const int wlc = (ltid & 7) ^ (wrr & 7);
    const f16* wrow = p.w + (long long)(n0 + wrr) * p.Kstride + wlc * 8;
    auto w_load = [&](int k0, int stage, bool real) {
      char* base = wring + stage * Cf::WSTAGE_BYTES + wave * 8 * 128;
#pragma unroll
      for (int rd = 0; rd < BR; ++rd) {
        const f16* src = real ? wrow + (long long)rd * 32 * p.Kstride + k0 : p.zero;
        char* dst = real ? base + rd * 32 * 128 : zpiece;
        glds16(src, dst);
      }
    };
    // prologue: window of slice 0 + weights of the first WS-1 steps
    for (int q = wave; q < nwi; q += 4) win_load(q, 0, 0);
    int ld_tap = 0, ld_slice = 0, ld_stage = 0;
    auto next_w = [&](bool real) {
      w_load(ld_tap * Cin + ld_slice * 64, ld_stage, real);
      if (++ld_tap == 9) {
        ld_tap = 0;
        ++ld_slice;
      }
      if (++ld_stage == WS) ld_stage = 0;
    };
#pragma unroll
    for (int s = 0; s < WS - 1; ++s) next_w(s < nsteps);
    wait_vmcnt<0>();
    __syncthreads();  // matches the consumers' prologue barrier

    int tap = 0, slice = 0, wbuf = 0;
    for (int s = 0; s < nsteps; ++s) {
      if (NWBUF == 1 && tap == 0 && slice > 0) {
        __builtin_amdgcn_s_barrier();  // A: every MFMA wave is done with the old window
        for (int q = wave; q < nwi; q += 4) win_load(q, slice, 0);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();  // B: new window (and every earlier weight stage) landed
      } else {
        wait_vmcnt<(WS - 2) * Cf::L>();
        __builtin_amdgcn_s_barrier();
      }
      next_w((s + WS - 1) < nsteps);
      if (NWBUF == 2) {
        const bool wreal = tap < WIN_STEPS && (slice + 1) < nslices;
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) win_load(wreal ? (tap * 4 + wave) * KW + kk : nwi, slice + 1, wbuf ^ 1);
      }
      if (++tap == 9) {
        tap = 0;
        ++slice;
        wbuf ^= 1;
      }
    }
    wait_vmcnt<0>();
  } else {
    // =========================================== MFMA waves ==========================================
    unsigned vmask[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * WTM + i * 16 + l15;
      vmask[i] = 0u;
      if (m < p.M) {
        const unsigned b = (unsigned)m / (unsigned)p.HoWo;
        const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
        const int y = (int)(pix / (unsigned)W), x = (int)(pix - (pix / (unsigned)W) * (unsigned)W);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
          if ((unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)W) vmask[i] |= 1u << t;
        }
      }
    }
    const int zaddr = (int)(zpiece - smem);
    const int r_lane = wm * WTM + l15;
    const int swz = l15 & 7;
    __syncthreads();  // prologue barrier (window 0 + first weight stages landed)

    int tap = 0, slice = 0, toff = 0, dxc = 0, wstage = 0, wbuf = 0;
    for (int s = 0; s < nsteps; ++s) {
      if (NWBUF == 1 && tap == 0 && slice > 0) {
        __builtin_amdgcn_s_barrier();  // A
        __builtin_amdgcn_s_barrier();  // B
      } else {
        __builtin_amdgcn_s_barrier();
      }
      const char* wbase = wring + wstage * Cf::WSTAGE_BYTES;
      const char* xbase = smem + (NWBUF == 2 ? wbuf * win_bytes : 0);
      const int r0 = r_lane + toff;
      const int xa0 = (r0 >> 4) * PIECE_BYTES + (r0 & 15) * 16 + lq * 256;
      __builtin_amdgcn_s_setprio(1);
      if (!(p.dbg & 2))
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        f16x8 xa[MT], wb[NTL];
        const int coff = ((kh * 4 + lq) ^ swz) * 16;
#pragma unroll
        for (int j = 0; j < NTL; ++j) wb[j] = *(const f16x8*)(wbase + (wn * WTN + j * 16 + l15) * 128 + coff);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const bool ok = (vmask[i] >> tap) & 1u;
          const char* a = ok ? xbase + (xa0 + i * PIECE_BYTES + kh * 1024) : smem + zaddr;
          xa[i] = *(const f16x8*)a;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NTL; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      if (++wstage == WS) wstage = 0;
      if (++tap == 9) {
        tap = 0;
        toff = 0;
        dxc = 0;
        ++slice;
        wbuf ^= 1;
      } else if (++dxc == 3) {
        dxc = 0;
        toff += W - 2;
      } else {
        ++toff;
      }
    }
  }
  __syncthreads();  // ring / window are reused as epilogue staging
  conv_epilogue<BN, 2, 2, MT, NTL, 512>(p, smem, acc, m0, n0, tid, is_loader ? -1 : wm, wn, l15, lq);
}

struct WinEntry {
  int BM, BN, threads, max_win_instr, wstage_bytes, epi_bytes;
  int nwbuf, wstages;
  const void* fn;
  const char* name;
};

#define OD_WSTR2(x) #x
#define OD_WSTR(x) OD_WSTR2(x)
#define OD_WIN(BM, BN, WM, WN, MINW)                                                                         \
  {                                                                                                          \
    BM, BN, WM* WN * 64, WinCfg<BM, BN, WM, WN>::MAX_WIN_INSTR, WinCfg<BM, BN, WM, WN>::WSTAGE_BYTES,        \
        WinCfg<BM, BN, WM, WN>::EPI_BYTES, 2, WSTAGES, (const void*)&od_conv3x3_win<BM, BN, WM, WN, MINW>,   \
        "od_conv3x3_win<" OD_WSTR(BM) ", " OD_WSTR(BN) ", " OD_WSTR(WM) ", " OD_WSTR(WN) ", " OD_WSTR(MINW) ">" \
  }

#define OD_WINS(BM, BN, WS, NWBUF, MINW)                                                                      \
  {                                                                                                          \
    BM, BN, 512, WinSpecCfg<BM, BN, WS, NWBUF>::MAX_WIN_INSTR, WinSpecCfg<BM, BN, WS, NWBUF>::WSTAGE_BYTES,  \
        WinSpecCfg<BM, BN, WS, NWBUF>::EPI_BYTES, NWBUF, WS,                                                 \
        (const void*)&od_conv3x3_wins<BM, BN, WS, NWBUF, MINW>,                                              \
        "od_conv3x3_wins<" OD_WSTR(BM) ", " OD_WSTR(BN) ", " OD_WSTR(WS) ", " OD_WSTR(NWBUF) ">"             \
  }

const WinEntry g_win[] = {
    OD_WIN(256, 128, 4, 2, 2),  // 0: 8 waves, wave tile 64x64
    OD_WIN(128, 128, 2, 2, 1),  // 1: 4 waves
    OD_WIN(128, 128, 4, 2, 2),  // 2: 8 waves, wave tile 32x64
    OD_WIN(256, 128, 2, 2, 1),  // 3: 4 waves, wave tile 128x64
    OD_WIN(128, 256, 2, 4, 2),  // 4: 8 waves, wave tile 64x64, wide N
    OD_WIN(128, 64, 2, 2, 1),   // 5: 4 waves, wave tile 64x32
    OD_WINS(128, 128, 3, 1, 4),  // 6: specialised, 1 window buffer, 3 weight stages, 2 WG/CU when <= 80 KiB
    OD_WINS(128, 128, 2, 1, 4),  // 7: 2 weight stages
    OD_WINS(128, 128, 2, 2, 4),  // 8: 2 window buffers
    OD_WINS(128, 128, 3, 2, 2),  // 9
    OD_WINS(256, 128, 3, 1, 2),  // 10: MFMA wave tile 128x64, 1 WG/CU
    OD_WINS(256, 128, 3, 2, 2),  // 11
};
constexpr int kNumWin = sizeof(g_win) / sizeof(g_win[0]);

}  // namespace

int od_conv_win_num_cfgs() { return kNumWin; }

bool od_conv_win_select(int idx, const ConvKP& p, ConvKernelInfo* info, size_t* lds_bytes) {
  if (idx < 0 || idx >= kNumWin) return false;
  if (p.stride != 1 || p.pad != 1 || (p.Cin & 63) != 0 || p.Ho != p.H || p.Wo != p.W || p.tconv) return false;
  const WinEntry& e = g_win[idx];
  const int np = (e.BM + 2 * p.W + 2 + 15) / 16;
  if (np * 2 > e.max_win_instr) return false;  // window cannot be streamed in 8 steps
  size_t lds = (size_t)e.nwbuf * np * PIECE_BYTES + (size_t)e.wstages * e.wstage_bytes + 1024;
  if ((size_t)e.epi_bytes > lds) lds = e.epi_bytes;
  if (lds > 160 * 1024) return false;
  info->fn = e.fn;
  info->name = e.name;
  info->BM = e.BM;
  info->BN = e.BN;
  info->threads = e.threads;
  *lds_bytes = lds;
  return true;
}

int od_conv_win_np(int idx, int W) { return (g_win[idx].BM + 2 * W + 2 + 15) / 16; }
