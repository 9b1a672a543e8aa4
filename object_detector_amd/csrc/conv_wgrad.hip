// K11 (backward-weight): dW[co][tap*Cin + ci] (dense f32 [Cout][k*k*Cin]) += sum_m dZ[m][co] * X[shift_tap(m)][ci]   (3x3 / 1x1, stride 1 / 2)
//
// GEMM with the PIXEL index as the reduction dimension: out tile = 128 output channels x 128 (tap, ci) columns, each
// workgroup reduces a contiguous slice of the M = B*Ho*Wo pixels (split-K over workgroups) and adds its f32 partial tile
// into dW with global_atomic_add_f32.  Both operands live pixel-major in HBM (NHWC), i.e. k-MINOR for this GEMM, so the
// 32-pixel x 128-channel LDS tiles are read with ds_read_b64_tr_b16 (the CDNA4 transposing LDS read) to form
// v_mfma_f32_16x16x32_f16 fragments: lane (l15, lq) gets channel l15 of pixels 8*lq .. 8*lq+7 from two 4x16 blocks.
// Tiles arrive by LDS-DMA (global_load_lds_dwordx4, per-lane source = the tap-shifted input pixel or the zero page);
// 256-B rows with the 16-B chunk index XORed by ((r&3) | ((r>>3)&1)<<2) << 1 make the transposed reads conflict-free.
//
// Replaces the weight-gradient half of the Keras Conv2D backward pass of the reference's (unseen) training loop
// (SURVEY.md §2.2 K11); results are f32 (loss-scaled), consumed by od_sgd_step.
#include <stdlib.h>

#include "conv_common.h"

namespace {

typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct WgradKP {
  const f16* x;    // [B,H,W,Cin]
  const f16* dz;   // [B,Ho,Wo,Cout]
  float* dw;       // [Cout][Ktot] f32, dense (the layout of the f32 master weights); atomic adds when slabs == null
  float* slabs;    // or: [split][Cout][Ktot] partial sums, plain stores (od_wgrad_reduce_multi adds them in a fixed order)
  const f16* zero;
  int H, W, Cin, Ho, Wo, Cout, ks, stride, pad;
  int Kstride, Ktot, M, HoWo;
  int rtiles, ctiles, split, chunks_per_split;
};

constexpr int KC = 32;          // pixels per K chunk (one MFMA k step)
constexpr int TILE = 128;       // out tile edge
constexpr int ROWB = 256;       // LDS row bytes (128 f16)
constexpr int OPER_BYTES = KC * ROWB;  // 8 KiB per operand per stage
#ifndef OD_WG_NSTAGE
#define OD_WG_NSTAGE 3
#endif
constexpr int NSTAGE = OD_WG_NSTAGE;       // LDS ring: chunk c+3 streams in while chunk c is multiplied (64 KiB, 2 workgroups per CU)

__device__ __forceinline__ int swz_key(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }

__global__ __launch_bounds__(256, 2) void od_conv_wgrad(WgradKP p) {
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * 2 * OPER_BYTES];  // [stage][D | X]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  int bid = blockIdx.x;
  const int sp = bid % p.split;
  bid /= p.split;
  const int ct = bid % p.ctiles, rt = bid / p.ctiles;
  const int co0 = rt * TILE, n0 = ct * TILE;
  const int chunk0 = sp * p.chunks_per_split;
  const int nchunks_total = (p.M + KC - 1) / KC;
  const int nch = min(p.chunks_per_split, nchunks_total - chunk0);
  if (nch <= 0) return;

  // ---- DMA mapping: instruction q (1 KiB) = tile rows 4q .. 4q+3; lane = (row 4q + lane/16, physical chunk lane%16)
  // wave w issues q = w and q = w + 4 for both operands
  const int prow[2] = {4 * wave + (lane >> 4), 4 * (wave + 4) + (lane >> 4)};
  int lchunk[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) lchunk[h] = (lane & 15) ^ swz_key(prow[h]);
  // X operand: this lane's columns (same for both rows when the keys agree; computed per row)
  int xtap_dy[2], xtap_dx[2], xci[2];
  bool xcol_ok[2], dcol_ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = n0 + lchunk[h] * 8;
    const int tap = j / p.Cin;
    xci[h] = j - tap * p.Cin;
    xtap_dy[h] = tap / p.ks;
    xtap_dx[h] = tap - xtap_dy[h] * p.ks;
    xcol_ok[h] = j < p.Ktot;
    dcol_ok[h] = (co0 + lchunk[h] * 8) < p.Cout;
  }

  // pixel coordinates of this lane's two tile rows for the NEXT chunk to be staged, advanced by KC per chunk (no division
  // in the loop: Wo >= 10 here, so a 32-pixel step wraps at most a few rows)
  int sm[2], sb[2], sho[2], swo[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sm[h] = chunk0 * KC + prow[h];
    const unsigned b = (unsigned)sm[h] / (unsigned)p.HoWo;
    const unsigned pix = (unsigned)sm[h] - b * (unsigned)p.HoWo;
    sb[h] = (int)b;
    sho[h] = (int)(pix / (unsigned)p.Wo);
    swo[h] = (int)(pix - (unsigned)sho[h] * (unsigned)p.Wo);
  }
  auto stage = [&](int buf) {  // stages are issued strictly in chunk order
    char* dbuf = smem + buf * 2 * OPER_BYTES;
    char* xbuf = dbuf + OPER_BYTES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = wave + 4 * h;
      const int m = sm[h];
      const bool mok = m < p.M;
      const f16* dsrc = (mok && dcol_ok[h]) ? p.dz + ((long long)m * p.Cout + co0 + lchunk[h] * 8) : p.zero;
      glds16(dsrc, dbuf + q * 1024);
      const int hi = sho[h] * p.stride + xtap_dy[h] - p.pad, wi = swo[h] * p.stride + xtap_dx[h] - p.pad;
      const bool xok = mok && xcol_ok[h] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const f16* xsrc = xok ? p.x + ((((long long)sb[h] * p.H + hi) * p.W + wi) * p.Cin + xci[h]) : p.zero;
      glds16(xsrc, xbuf + q * 1024);
      sm[h] += KC;
      swo[h] += KC;
      while (swo[h] >= p.Wo) {
        swo[h] -= p.Wo;
        if (++sho[h] == p.Ho) {
          sho[h] = 0;
          ++sb[h];
        }
      }
    }
  };

  // ---- MFMA side: wave (wr, wc) owns a 64 x 64 sub-tile ---------------------------------------------------------
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // transposed-read addressing: lane 4q+p of a 16-lane group supplies row q, 4 columns starting at 4p
  const int tq = l15 >> 2, tp = l15 & 3;
  int roff[2];  // byte offset of (row, swizzle) for the two 4-row blocks of this lane's 8 pixels
  int rkey[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 8 * lq + 4 * h + tq;
    roff[h] = r * ROWB;
    rkey[h] = swz_key(r);
  }

  // ring: chunks are staged NSTAGE-1 ahead; a counted vmcnt retires chunk c and leaves the later ones in flight (4 DMAs per
  // chunk per wave), one raw s_barrier per chunk (every wave's piece of chunk c landed; chunk c-1's buffer is free)
#pragma unroll
  for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
    if (s0 < nch) stage(s0);
  int buf = 0, sbuf = NSTAGE - 1;  // buffer of chunk c / of the next chunk to stage
  for (int c = 0; c < nch; ++c, buf = (buf + 1 == NSTAGE ? 0 : buf + 1)) {
    // chunks c+1 .. c+NSTAGE-2 may stay in flight
    if (NSTAGE >= 4 && c + 2 < nch) {
      wait_vmcnt<8>();
    } else if (NSTAGE >= 3 && c + 1 < nch) {
      wait_vmcnt<4>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (c + NSTAGE - 1 < nch) {
      stage(sbuf);
      sbuf = sbuf + 1 == NSTAGE ? 0 : sbuf + 1;
    }
    const char* dbuf = smem + buf * 2 * OPER_BYTES;
    const char* xbuf = dbuf + OPER_BYTES;
    f16x8 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = (wr * 64 + i * 16) / 4 + tp;  // 8-byte unit inside the row (16 channels = 4 units)
      h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(dbuf + roff[0] + (((u >> 1) ^ rkey[0]) * 16) + (u & 1) * 8));
      h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(dbuf + roff[1] + (((u >> 1) ^ rkey[1]) * 16) + (u & 1) * 8));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        af[i][e] = (f16)lo[e];
        af[i][4 + e] = (f16)hi[e];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int u = (wc * 64 + j * 16) / 4 + tp;
      h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(xbuf + roff[0] + (((u >> 1) ^ rkey[0]) * 16) + (u & 1) * 8));
      h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(xbuf + roff[1] + (((u >> 1) ^ rkey[1]) * 16) + (u & 1) * 8));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf[j][e] = (f16)lo[e];
        bf[j][4 + e] = (f16)hi[e];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
  }

  // ---- f32 partial tile -> dW (atomic adds; 16 lanes = 64 contiguous bytes per row) ---------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wc * 64 + j * 16 + l15;
      if (col < p.Ktot) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = co0 + wr * 64 + i * 16 + lq * 4 + e;
          if (co < p.Cout) {
            if (p.slabs) p.slabs[((long long)sp * p.Cout + co) * p.Kstride + col] = acc[i][j][e];
            else atomicAdd(p.dw + (long long)co * p.Kstride + col, acc[i][j][e]);
          }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 8-wave form: 256 output channels x 256 (tap, ci) columns per workgroup, one workgroup per CU.
// The 128 x 128 kernel above moves 16 KiB L2 -> LDS per 1.05 MFLOP (4 LDS-DMA issues per wave per 16 MFMAs: the waves are
// DMA-issue bound, 0.1 of the MFMA peak in the training step); the 256-wide tile halves the bytes per flop and gives
// every wave 32 MFMAs per 4 DMA issues and per barrier.  LDS image: per stage four half-tiles of [32 pixels][128 channels]
// (dZ channels 0-127 / 128-255, X columns 0-127 / 128-255), each with the row layout / swizzle of the kernel above, so the
// transposed fragment reads stay conflict-free.  Wave (wr, wc) = (wave >> 1, wave & 1) owns 64 channels x 128 columns.
// The MFMA takes the X fragment as its first operand: a lane then holds 4 CONSECUTIVE columns of one output channel, and
// the f32 partial tile leaves as 16-byte stores (32 per wave) instead of 128 scalar ones.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int W8_TILE = 256;
constexpr int W8_NSTAGE = 4;
constexpr int W8_STAGE_BYTES = 4 * OPER_BYTES;  // 32 KiB
constexpr int W8_LDS = W8_NSTAGE * W8_STAGE_BYTES;  // 128 KiB

__global__ __launch_bounds__(512, 2) void od_conv_wgrad_w8(WgradKP p) {
  extern __shared__ __attribute__((aligned(16))) char smem8[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;

  int bid = blockIdx.x;
  const int sp = bid % p.split;
  bid /= p.split;
  const int ct = bid % p.ctiles, rt = bid / p.ctiles;
  const int co0 = rt * W8_TILE, n0 = ct * W8_TILE;
  const int chunk0 = sp * p.chunks_per_split;
  const int nchunks_total = (p.M + KC - 1) / KC;
  const int nch = min(p.chunks_per_split, nchunks_total - chunk0);
  if (nch <= 0) return;

  // ---- DMA mapping: wave w fills rows 4w .. 4w+3 of each of the four half-tiles (one 1-KiB instruction each)
  const int prow = 4 * wave + (lane >> 4);
  const int lchunk = (lane & 15) ^ swz_key(prow);
  int xtap_dy[2], xtap_dx[2], xci[2];
  bool xcol_ok[2], dcol_ok[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int j = n0 + h * 128 + lchunk * 8;
    const int tap = j / p.Cin;
    xci[h] = j - tap * p.Cin;
    xtap_dy[h] = tap / p.ks;
    xtap_dx[h] = tap - xtap_dy[h] * p.ks;
    xcol_ok[h] = j < p.Ktot;
    dcol_ok[h] = (co0 + h * 128 + lchunk * 8) < p.Cout;
  }
  int sm = chunk0 * KC + prow, sb, sho, swo;
  {
    const unsigned b = (unsigned)sm / (unsigned)p.HoWo;
    const unsigned pix = (unsigned)sm - b * (unsigned)p.HoWo;
    sb = (int)b;
    sho = (int)(pix / (unsigned)p.Wo);
    swo = (int)(pix - (unsigned)sho * (unsigned)p.Wo);
  }
  auto stage = [&](int buf) {  // stages are issued strictly in chunk order
    char* base = smem8 + buf * W8_STAGE_BYTES + wave * 1024;
    const bool mok = sm < p.M;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f16* dsrc = (mok && dcol_ok[h]) ? p.dz + ((long long)sm * p.Cout + co0 + h * 128 + lchunk * 8) : p.zero;
      glds16(dsrc, base + h * OPER_BYTES);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int hi = sho * p.stride + xtap_dy[h] - p.pad, wi = swo * p.stride + xtap_dx[h] - p.pad;
      const bool xok = mok && xcol_ok[h] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const f16* xsrc = xok ? p.x + ((((long long)sb * p.H + hi) * p.W + wi) * p.Cin + xci[h]) : p.zero;
      glds16(xsrc, base + (2 + h) * OPER_BYTES);
    }
    sm += KC;
    swo += KC;
    while (swo >= p.Wo) {
      swo -= p.Wo;
      if (++sho == p.Ho) {
        sho = 0;
        ++sb;
      }
    }
  };

  // ---- MFMA side ---------------------------------------------------------------------------------------------------
  const int wr = wave >> 1, wc = wave & 1;
  const int d_half = wr >> 1, d_ch0 = (wr & 1) * 64;
  f32x4 acc[8][4];  // [x fragment j][dz fragment i]: element e = column j*16 + lq*4 + e, channel i*16 + l15
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tq = l15 >> 2, tp = l15 & 3;
  int roff[2], rkey[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = 8 * lq + 4 * h + tq;
    roff[h] = r * ROWB;
    rkey[h] = swz_key(r);
  }
  auto frag = [&](const char* tile, int ch) -> f16x8 {  // 16 channels starting at `ch` of a half-tile, this lane's 8 pixels
    const int u = ch / 4 + tp;
    h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) h4*)(tile + roff[0] + (((u >> 1) ^ rkey[0]) * 16) + (u & 1) * 8));
    h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) h4*)(tile + roff[1] + (((u >> 1) ^ rkey[1]) * 16) + (u & 1) * 8));
    f16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[e] = (f16)lo[e];
      f[4 + e] = (f16)hi[e];
    }
    return f;
  };

#pragma unroll
  for (int s0 = 0; s0 < W8_NSTAGE - 1; ++s0)
    if (s0 < nch) stage(s0);
  int buf = 0, sbuf = W8_NSTAGE - 1;
  for (int c = 0; c < nch; ++c, buf = (buf + 1 == W8_NSTAGE ? 0 : buf + 1)) {
    // 4 DMAs per chunk per wave; chunks c+1, c+2 may stay in flight
    if (c + 2 < nch) {
      wait_vmcnt<8>();
    } else if (c + 1 < nch) {
      wait_vmcnt<4>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // every wave's rows of chunk c have landed; chunk c-1's buffer is free
    if (c + W8_NSTAGE - 1 < nch) {
      stage(sbuf);
      sbuf = sbuf + 1 == W8_NSTAGE ? 0 : sbuf + 1;
    }
    const char* st = smem8 + buf * W8_STAGE_BYTES;
    const char* dtile = st + d_half * OPER_BYTES;
    const char* xtile = st + (2 + wc) * OPER_BYTES;
    f16x8 df[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) df[i] = frag(dtile, d_ch0 + i * 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f16x8 xf = frag(xtile, j * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, df[i], acc[j][i], 0, 0, 0);
    }
  }

  // ---- f32 partial tile -> slab (16-byte stores: 4 consecutive columns per lane) or atomic adds into dW
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int co = co0 + wr * 64 + i * 16 + l15;
    if (co >= p.Cout) continue;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = n0 + wc * 128 + j * 16 + lq * 4;
      if (col >= p.Ktot) continue;  // Ktot is a multiple of 8: a group of 4 columns is all in or all out
      if (p.slabs) {
        *(f32x4*)(p.slabs + ((long long)sp * p.Cout + co) * p.Kstride + col) = acc[j][i];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(p.dw + (long long)co * p.Kstride + col + e, acc[j][i][e]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Thin form for the 160x160 / 320x320 maps: Cin = 32, Cout = 64, 3x3 (b.down1, b.s1.0.b): the whole 64 x 288 gradient is ONE
// tile, the reduction runs over 0.8 M pixels, and the layer is a stream of dZ (105 MB) and X (52 / 210 MB).  On the 128 x 128
// kernel above it is 3 column tiles x 171 pixel splits that each re-stage dZ and gather X tap by tap (9 x the bytes) at 37 %
// tile use: 150-165 us.  Here: persistent workgroups (4 waves as 2 x 2: 32 channels x 144 columns per wave = 18 accumulator
// fragments), a chunk = 32 consecutive output pixels of one row: its dZ rows (4 KiB, contiguous) and the three input row
// segments it touches (34 or 65 pixels x 64 B each, contiguous) arrive by LDS-DMA into a 3-deep ring (counted vmcnt, one
// raw barrier per chunk); every tap is a row offset into those segments for the transposing LDS read.  One f32 slab per
// workgroup, summed by od_wgrad_reduce_multi in its fixed order.
// ---------------------------------------------------------------------------------------------------------------------
template <int S>
struct ThinCfg {
  static constexpr int NPX = 31 * S + 3;                  // input pixels per row segment: 34 (stride 1) / 65 (stride 2)
  static constexpr int XPIECES = (NPX + 15) / 16;         // 1-KiB pieces (16 pixels x 64 B) per segment: 3 / 5
  static constexpr int XSLOTS = XPIECES * 16;
  static constexpr int PIECES = (4 + 3 * XPIECES + 3) / 4 * 4;  // dZ 4 + X 3 segments, padded to the 4 waves: 16 / 20
  static constexpr int ND = PIECES / 4;
  static constexpr int BUF = PIECES * 1024;
  static constexpr int LDS = 3 * BUF;
};

template <int S>
__global__ __launch_bounds__(256, 2) void od_conv_wgrad_thin(WgradKP p, int chunks_per_row, int nchunks) {
  using Cf = ThinCfg<S>;
  extern __shared__ __attribute__((aligned(16))) char tsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int tq = l15 >> 2, tp = l15 & 3;
  const int wr = wave >> 1, wc = wave & 1;

  // piece k of a chunk: 0-3 = dZ (8 pixels x 128 B each), then segment dy = 0..2 with XPIECES pieces of 16 pixels x 64 B,
  // then padding pieces (zero page) so that every wave issues ND per chunk
  auto stage = [&](int chunk, int buf) {
    const int row = chunk / chunks_per_row, ox0 = (chunk - row * chunks_per_row) * 32;
    const int b = row / p.Ho, oy = row - b * p.Ho;
    char* base = tsm + buf * Cf::BUF;
#pragma unroll
    for (int k = 0; k < Cf::ND; ++k) {
      const int piece = wave + 4 * k;
      const f16* src = p.zero;
      if (piece < 4) {
        src = p.dz + (((long long)row * p.Wo + ox0) * 64 + piece * 512 + lane * 8);
      } else if (piece < 4 + 3 * Cf::XPIECES) {
        const int sp = piece - 4;
        const int dy = sp / Cf::XPIECES, q = (sp - dy * Cf::XPIECES) * 16 + (lane >> 2);  // pixel slot inside the segment
        const int iy = oy * S + dy - 1, ix = ox0 * S - 1 + q;
        if (q < Cf::NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          src = p.x + ((((long long)b * p.H + iy) * p.W + ix) * 32 + (lane & 3) * 8);
      }
      glds16(src, base + piece * 1024);
    }
  };

  f32x4 acc[2][9];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int first = blockIdx.x, step = gridDim.x;
  const int nmine = first < nchunks ? (nchunks - first + step - 1) / step : 0;
  if (nmine > 0) stage(first, 0);
  if (nmine > 1) stage(first + step, 1);
  for (int it = 0; it < nmine; ++it) {
    // chunk `it` landed: at most the ND pieces of chunk it+1 stay in flight
    if (it + 1 < nmine) {
      wait_vmcnt<Cf::ND>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (it + 2 < nmine) stage(first + (it + 2) * step, (it + 2) % 3);
    const char* db = tsm + (it % 3) * Cf::BUF;  // dZ: [32 pixels][64 channels], 128-B rows
    const char* xb = db + 4096;                 // X: 3 segments of XSLOTS pixels, 64-B rows
    f16x8 af[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int unit = (2 * wr + i) * 4 + tp;  // 8-byte unit inside the 128-B row: 16 channels = 4 units
      h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(db + (8 * lq + tq) * 128 + unit * 8));
      h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(db + (8 * lq + 4 + tq) * 128 + unit * 8));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        af[i][e] = (f16)lo[e];
        af[i][4 + e] = (f16)hi[e];
      }
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int cf = wc * 9 + c;            // column fragment 0..17 = (tap, 16-channel half)
      const int tap = cf >> 1, half = cf & 1;
      const int dy = tap / 3, dx = tap - dy * 3;
      const char* seg = xb + dy * (Cf::XSLOTS * 64);
      h4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(seg + (S * (8 * lq + tq) + dx) * 64 + (half * 4 + tp) * 8));
      h4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
          (__attribute__((address_space(3))) h4*)(seg + (S * (8 * lq + 4 + tq) + dx) * 64 + (half * 4 + tp) * 8));
      f16x8 bf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf[e] = (f16)lo[e];
        bf[4 + e] = (f16)hi[e];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf, acc[i][c], 0, 0, 0);
    }
  }
  // acc[i][c][e] = D[channel (2*wr + i)*16 + lq*4 + e][column (wc*9 + c)*16 + l15]
  float* slab = p.slabs + (long long)blockIdx.x * 64 * 288;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 9; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        slab[((2 * wr + i) * 16 + lq * 4 + e) * 288 + (wc * 9 + c) * 16 + l15] = acc[i][c][e];
}

}  // namespace

// pixel split of the 256-wide kernel: one 512-thread workgroup per CU, one round (tiles x split <= CUs), and at least 12
// chunks per workgroup so that the pipeline fill and the 256-KiB partial-tile store stay a small part of it
static int wgrad_w8_split(int cus, int nchunks, int Cout, int Ktot, int* chunks_per_split) {
  const int tiles = od_ceil_div(Cout, W8_TILE) * od_ceil_div(Ktot, W8_TILE);
  static int target = -1;  // workgroups one launch aims at (OD_WGRAD_W8_WGS; default = every CU)
  if (target < 0) {
    const char* e = getenv("OD_WGRAD_W8_WGS");
    target = e ? atoi(e) : 0;
  }
  if (target > 0) cus = target;
  int split = cus / tiles;
  if (split < 1) split = 1;
  if (split > nchunks / 12) split = nchunks / 12 > 0 ? nchunks / 12 : 1;
  const int cps = od_ceil_div(nchunks, split);
  if (chunks_per_split) *chunks_per_split = cps;
  return od_ceil_div(nchunks, cps);
}

// which kernel a layer takes: the 256-wide tile when it covers the [Cout][Ktot] matrix without much padding AND its one
// round of workgroups fills most of the chip (measured per shape, profiles/r02/wgrad_bench.txt: the 1x1 layers and the
// 10x10 maps with few tiles run 62-72 workgroups of it and are faster on the 128-wide kernel's 480+)
static bool wgrad_use_w8(int cus, int M, int Cout, int Ktot) {
  static int force = -2;
  if (force == -2) {
    const char* e = getenv("OD_WGRAD_W8");  // 0 = never, 1 = whenever possible (tuning)
    force = e ? atoi(e) : -1;
  }
  if (force == 0) return false;
  if (force == 1) return true;
  const long long tiles = (long long)od_ceil_div(Cout, W8_TILE) * od_ceil_div(Ktot, W8_TILE);
  const double eff = (double)Cout * Ktot / (double)(tiles * W8_TILE * W8_TILE);
  int cps = 0;
  const int split = wgrad_w8_split(cus, od_ceil_div(M, KC), Cout, Ktot, &cps);
  static int min_wgs = -1, min_cps = -1;
  if (min_wgs < 0) {
    const char* e = getenv("OD_WGRAD_W8_MIN");
    min_wgs = e ? atoi(e) : 160;
    // pixel chunks per workgroup below which the 128 x 128 kernel is taken.  Standalone the 256-wide kernel wins from ~24
    // chunks; INSIDE the two-stream step it holds a whole CU (128 KiB of LDS, 256 VGPRs x 8 waves) while the other stream's
    // kernels wait for a slot, and only pays with long pixel runs: 32 x 320^2 (31 chunks per workgroup on stage 3) 10.80 ->
    // 10.65 ms without it, 16 x 640^2 (63) 17.5 -> 18.0 ms without it; 60 keeps both (sweep: profiles/r02/wgrad_w8_cps_sweep.txt)
    const char* c = getenv("OD_WGRAD_W8_CPS");
    min_cps = c ? atoi(c) : 60;
  }
  return eff >= 0.74 && tiles * split >= min_wgs && cps >= min_cps;
}

// the thin kernel's shapes (slab output only): 3x3, 32 -> 64 channels, output rows that are whole 32-pixel chunks
static bool wgrad_thin_ok(int Cin, int Cout, int ksize, int stride, int H, int W, int Ho, int Wo) {
  static int allow = -1;
  if (allow < 0) {
    const char* e = getenv("OD_WGRAD_THIN");  // 0 = the 128 x 128 kernel (A/B timing)
    allow = e ? atoi(e) : 1;
  }
  return allow && Cin == 32 && Cout == 64 && ksize == 3 && Wo % 32 == 0 && H == Ho * stride && W == Wo * stride;
}
static int wgrad_thin_grid(const od_ctx* ctx, long long nchunks) {
  long long g = 2LL * (ctx->num_cu > 0 ? ctx->num_cu : 256);
  return (int)(g < nchunks ? g : nchunks);
}

static int wgrad_split(const od_ctx* ctx, int M, int Cout, int Ktot, int* chunks_per_split) {
  const int nchunks = od_ceil_div(M, KC);
  const int cus = ctx->num_cu > 0 ? ctx->num_cu : 256;
  if (wgrad_use_w8(cus, M, Cout, Ktot)) return wgrad_w8_split(cus, nchunks, Cout, Ktot, chunks_per_split);
  const int rtiles = od_ceil_div(Cout, TILE), ctiles = od_ceil_div(Ktot, TILE);
  static int split_mul = -1;  // workgroups per CU the pixel split aims at (OD_WGRAD_SPLIT; every workgroup emits a full
  if (split_mul < 0) {        // 64 KiB f32 tile, so more splits = more partial-sum traffic)
    const char* e = getenv("OD_WGRAD_SPLIT");
    split_mul = e ? atoi(e) : 2;  // measured on the batch-32 step: 4 -> 18.2 ms, 2 -> 17.5 ms, 1 -> 18.7 ms
  }
  int split = od_ceil_div(split_mul * cus, rtiles * ctiles);
  if (split > nchunks) split = nchunks;
  if (split < 1) split = 1;
  const int cps = od_ceil_div(nchunks, split);
  if (chunks_per_split) *chunks_per_split = cps;
  return od_ceil_div(nchunks, cps);
}

static int wgrad_impl(od_ctx* ctx, const void* x, const void* dz, float* dw, float* slabs, int B, int H, int W, int Cin,
                      int Cout, int ksize, int stride, void* stream, int* nsplit) {
  OD_REQUIRE(ctx && x && dz && (dw || slabs), "od_conv2d_bwd_weight: null argument");
  OD_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), "od_conv2d_bwd_weight: bad ksize/stride");
  OD_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0,
             "od_conv2d_bwd_weight: Cin/Cout must be multiples of 8");
  WgradKP p;
  p.x = (const f16*)x;
  p.dz = (const f16*)dz;
  p.dw = dw;
  p.slabs = slabs;
  p.zero = (const f16*)ctx->zero_page;
  p.H = H;
  p.W = W;
  p.Cin = Cin;
  p.ks = ksize;
  p.stride = stride;
  p.pad = ksize / 2;
  p.Ho = (H + 2 * p.pad - ksize) / stride + 1;
  p.Wo = (W + 2 * p.pad - ksize) / stride + 1;
  p.Cout = Cout;
  p.Ktot = ksize * ksize * Cin;
  p.Kstride = p.Ktot;
  const long long M64 = (long long)B * p.Ho * p.Wo;
  OD_REQUIRE(M64 * Cout < (1LL << 31) && (long long)B * H * W * Cin < (1LL << 31), "od_conv2d_bwd_weight: too large");
  p.M = (int)M64;
  p.HoWo = p.Ho * p.Wo;
  if (slabs && wgrad_thin_ok(Cin, Cout, ksize, stride, H, W, p.Ho, p.Wo)) {
    const int nchunks = p.M / 32;
    const int grid = wgrad_thin_grid(ctx, nchunks);
    p.split = grid;
    if (nsplit) *nsplit = grid;
    const void* fn = stride == 1 ? (const void*)&od_conv_wgrad_thin<1> : (const void*)&od_conv_wgrad_thin<2>;
    const size_t lds = stride == 1 ? (size_t)ThinCfg<1>::LDS : (size_t)ThinCfg<2>::LDS;
    if (int rc = od_ensure_lds(ctx, fn, lds)) return rc;
    int cpr = p.Wo / 32, nch = nchunks;
    void* args[] = {&p, &cpr, &nch};
    OD_CHECK_HIP(hipLaunchKernel(fn, dim3(grid), dim3(256), args, lds, (hipStream_t)stream));
    return OD_OK;
  }
  p.split = wgrad_split(ctx, p.M, Cout, p.Ktot, &p.chunks_per_split);
  if (nsplit) *nsplit = p.split;
  if (wgrad_use_w8(ctx->num_cu > 0 ? ctx->num_cu : 256, p.M, Cout, p.Ktot)) {
    p.rtiles = od_ceil_div(Cout, W8_TILE);
    p.ctiles = od_ceil_div(p.Ktot, W8_TILE);
    if (int rc = od_ensure_lds(ctx, (const void*)&od_conv_wgrad_w8, (size_t)W8_LDS)) return rc;
    hipLaunchKernelGGL(od_conv_wgrad_w8, dim3(p.rtiles * p.ctiles * p.split), dim3(512), W8_LDS, (hipStream_t)stream, p);
    OD_CHECK_LAUNCH();
    return OD_OK;
  }
  p.rtiles = od_ceil_div(Cout, TILE);
  p.ctiles = od_ceil_div(p.Ktot, TILE);
  hipLaunchKernelGGL(od_conv_wgrad, dim3(p.rtiles * p.ctiles * p.split), dim3(256), 0, (hipStream_t)stream, p);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

// for conv_first.hip (the first layer's weight gradient runs on this kernel over an f16 x 8-channel copy of the image)
int od_wgrad_slabs_impl(od_ctx* ctx, const void* x, const void* dz, float* slabs, int B, int H, int W, int Cin, int Cout,
                        int ksize, int stride, void* stream, int* nsplit) {
  return wgrad_impl(ctx, x, dz, nullptr, slabs, B, H, W, Cin, Cout, ksize, stride, stream, nsplit);
}

extern "C" int od_conv2d_bwd_weight(od_ctx* ctx, const void* x, const void* dz, float* dw, int B, int H, int W, int Cin,
                                    int Cout, int ksize, int stride, void* stream) {
  return wgrad_impl(ctx, x, dz, dw, nullptr, B, H, W, Cin, Cout, ksize, stride, stream, nullptr);
}

extern "C" int od_conv2d_bwd_weight_splits(od_ctx* ctx, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
  if (!ctx || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2))
    return 0;
  const int pad = ksize / 2;
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  const long long M = (long long)B * Ho * Wo;
  if (wgrad_thin_ok(Cin, Cout, ksize, stride, H, W, Ho, Wo)) return wgrad_thin_grid(ctx, M / 32);  // slabs of od_conv2d_bwd_weight_slabs
  return wgrad_split(ctx, (int)M, Cout, ksize * ksize * Cin, nullptr);
}

extern "C" int od_conv2d_bwd_weight_slabs(od_ctx* ctx, const void* x, const void* dz, float* slabs, int B, int H, int W,
                                          int Cin, int Cout, int ksize, int stride, void* stream) {
  OD_REQUIRE(slabs, "od_conv2d_bwd_weight_slabs: null slabs");
  return wgrad_impl(ctx, x, dz, nullptr, slabs, B, H, W, Cin, Cout, ksize, stride, stream, nullptr);
}

namespace {
// grads[dw_offset + i] = sum over the layer's slabs in a FIXED order (bit-reproducible); blockIdx.y = layer.
// Two levels: the slab range is cut into G contiguous groups (G = a power of two chosen from count and nslabs only, so the
// order never depends on the launch), a thread adds its group's slabs in ascending order with four loads in flight, the G
// partial sums are added in ascending group order through LDS.  G = 1 for the large layers (one thread per four elements
// already fills the chip); the small ones (a 64 -> 32 1x1 layer has 2048 weights and 512 slabs) used to run 512 threads
// through 512 dependent loads each: 49 us, now 8.
__global__ __launch_bounds__(256) void od_wgrad_reduce_k(const od_wgrad_red* __restrict__ tbl, float* __restrict__ grads) {
  __shared__ f32x4 part[256];
  const od_wgrad_red e = tbl[blockIdx.y];
  float* out = grads + e.dw_offset;
  const long long n4 = e.count >> 2;  // counts are multiples of 8 (Cin, Cout % 8 == 0)
  int G = 1;
  while (G < 64 && n4 * G < 65536 && G * 2 <= e.nslabs) G *= 2;
  const int q = 256 / G;              // element quads per workgroup pass
  const int iq = threadIdx.x % q, g = threadIdx.x / q;
  const int per = (e.nslabs + G - 1) / G;
  const int k0 = g * per, k1 = min(k0 + per, e.nslabs);
  for (long long base = (long long)blockIdx.x * q; base < n4; base += (long long)gridDim.x * q) {
    const long long i = base + iq;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4 && k0 < k1) {
      s = *(const f32x4*)(e.slabs + (long long)k0 * e.count + i * 4);
      int k = k0 + 1;
      for (; k + 3 < k1; k += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(e.slabs + (long long)(k + u) * e.count + i * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s[0] += v[u][0];
          s[1] += v[u][1];
          s[2] += v[u][2];
          s[3] += v[u][3];
        }
      }
      for (; k < k1; ++k) {
        const f32x4 v = *(const f32x4*)(e.slabs + (long long)k * e.count + i * 4);
        s[0] += v[0];
        s[1] += v[1];
        s[2] += v[2];
        s[3] += v[3];
      }
    }
    if (G == 1) {
      if (i < n4) *(f32x4*)(out + i * 4) = s;
      continue;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n4) {
      for (int h = 1; h < G; ++h) {
        if (h * per >= e.nslabs) break;  // empty trailing groups
        const f32x4 v = part[h * q + iq];
        s[0] += v[0];
        s[1] += v[1];
        s[2] += v[2];
        s[3] += v[3];
      }
      *(f32x4*)(out + i * 4) = s;
    }
    __syncthreads();
  }
}
}  // namespace

extern "C" int od_wgrad_reduce_multi(od_ctx* ctx, const od_wgrad_red* table, int nlayers, float* grads, void* stream) {
  OD_REQUIRE(ctx && table && grads && nlayers > 0 && nlayers <= 65535, "od_wgrad_reduce_multi: bad argument");
  hipLaunchKernelGGL(od_wgrad_reduce_k, dim3(nlayers == 1 ? 1024 : 256, nlayers), dim3(256), 0, (hipStream_t)stream, table, grads);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
