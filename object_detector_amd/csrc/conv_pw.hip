// K1'' : persistent, wave-specialised LDS-window 3x3 / stride-1 convolution for the large layers.
//
// What the round-1 measurements asked for: fewer L2->LDS bytes per flop than the 128x128 implicit GEMM
// (15.6 KB/MFLOP, 63 % MFMA-bound in cycles), no per-tile launch / prologue / epilogue bubble, MFMA waves that never issue
// a DMA.  One workgroup per CU, 12 waves:
//   waves 0-7  : MFMA waves, 4 (M) x 2 (N) over a 256-pixel x 128-channel tile, 64 x 64 each, fragment reads
//                software-pipelined by k-half, epilogue straight from the accumulators (no LDS staging, no barrier)
//   waves 8-11 : DMA waves: per step the 128 x 64 weight slice (3-deep ring) plus 1/8 of the NEXT 64-channel slice's
//                input window into the other window buffer -- also across tile boundaries, so the next tile's first
//                steps are already in LDS when the MFMA waves finish their epilogue
// The workgroup walks its tiles (persistent); the step sequence (tile, slice, tap) is one continuous pipeline with one
// s_barrier per step and counted vmcnt.  L2->LDS traffic: (16 + 44/9) KB per 4.2 MFLOP step = 5 KB/MFLOP.
// Window image and zero-redirect of out-of-image taps: exactly as conv_win.hip.
//
// Replaces the 3x3 Conv2D + BatchNormalization + LeakyReLU/ELU (+ Add) layers of `ObjectDetector.predict`
// (reference voc_validate.py:27; docs/MODEL.md:5-21).
#include "conv_common.h"

namespace {

constexpr int PW_BM = 256, PW_BN = 128;
constexpr int PW_PIECE = 2048;     // 16 pixels x 64 channels x 2 B
constexpr int PW_WSTAGE = PW_BN * 128;
constexpr int PW_BR = 4;           // weight DMA instructions per loader wave per step (128 rows x 128 B / 4 waves / 1 KiB)
constexpr int PW_KW = 2;           // window DMA slots per loader wave per step (taps 0..7)
constexpr int PW_MT = 4, PW_NT = 4;

template <int WS>
__global__ __launch_bounds__(768, 3) void od_conv3x3_pw(ConvKP p, int np, int ntiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int win_bytes = np * PW_PIECE;
  char* const wring = smem + 2 * win_bytes;
  char* const zpiece = wring + WS * PW_WSTAGE;
  constexpr int L = PW_BR + PW_KW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_loader = wave_all >= 8;
  const int l15 = lane & 15, lq = lane >> 4;
  const int W = p.W, Cin = p.Cin;
  const int nslices = Cin >> 6;
  const int nwi = np * 2;

  // persistent tile walk: workgroup w' (XCD-contiguous relabelling of blockIdx) owns tiles w', w'+G, ...
  const int G = gridDim.x;
  int wq;
  {
    const int pid = blockIdx.x;
    const int q = G >> 3, r = G & 7, xcd = pid & 7, loc = pid >> 3;
    wq = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int nmy = wq < ntiles_total ? (ntiles_total - wq + G - 1) / G : 0;
  if (nmy == 0) return;
  const int total_steps = nmy * nslices * 9;

  if (wave_all == 0) *(f32x4*)(zpiece + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tile_m0 = [&](int k) { return ((wq + k * G) / p.ntiles) * PW_BM; };
  auto tile_n0 = [&](int k) { return ((wq + k * G) % p.ntiles) * PW_BN; };

  if (is_loader) {
    // ================================================ DMA waves ================================================
    const int lw = wave_all - 8;
    const int ltid = tid - 512;
    auto win_load = [&](int q, int k, int slice, int wbuf) {
      const bool real = q < nwi && k < nmy;
      const int piece = q >> 1, half = q & 1;
      const int mp = tile_m0(k) - W - 1 + piece * 16 + (lane & 15);
      const bool ok = real && (unsigned)mp < (unsigned)p.M;
      const f16* src = ok ? p.x + ((long long)mp * Cin + slice * 64 + (half * 4 + (lane >> 4)) * 8) : p.zero;
      char* dst = real ? smem + wbuf * win_bytes + piece * PW_PIECE + half * 1024 : zpiece;
      glds16(src, dst);
    };
    const int wrr = ltid >> 3;
    const int wlc = (ltid & 7) ^ (wrr & 7);
    // weight stream iterator (WS-1 steps ahead of the MFMA waves)
    int wk = 0, wslice = 0, wtap = 0, wstage = 0;
    auto next_w = [&]() {
      const bool real = wk < nmy;
      const f16* wrow = p.w + (long long)(tile_n0(real ? wk : 0) + wrr) * p.Kstride + wlc * 8 + wtap * Cin + wslice * 64;
      char* base = wring + wstage * PW_WSTAGE + lw * 8 * 128;
#pragma unroll
      for (int rd = 0; rd < PW_BR; ++rd) {
        const f16* src = real ? wrow + (long long)rd * 32 * p.Kstride : p.zero;
        char* dst = real ? base + rd * 32 * 128 : zpiece;
        glds16(src, dst);
      }
      if (++wtap == 9) {
        wtap = 0;
        if (++wslice == nslices) {
          wslice = 0;
          ++wk;
        }
      }
      if (++wstage == WS) wstage = 0;
    };
    // prologue: first window + the first WS-1 weight stages
    for (int q = lw; q < nwi; q += 4) win_load(q, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < WS - 1; ++s) next_w();
    wait_vmcnt<0>();
    __syncthreads();

    int k = 0, slice = 0, tap = 0, wbuf = 0;
    for (int gs = 0; gs < total_steps; ++gs) {
      wait_vmcnt<(WS - 2) * L>();
      __builtin_amdgcn_s_barrier();
      next_w();
      {
        // the slice after the current one (possibly the first slice of the next tile)
        int nk = k, ns = slice + 1;
        if (ns == nslices) {
          ns = 0;
          ++nk;
        }
#pragma unroll
        for (int kk = 0; kk < PW_KW; ++kk) {
          const int q = tap < 8 ? (tap * 4 + lw) * PW_KW + kk : nwi;
          win_load(q, nk, ns, wbuf ^ 1);
        }
      }
      if (++tap == 9) {
        tap = 0;
        wbuf ^= 1;
        if (++slice == nslices) {
          slice = 0;
          ++k;
        }
      }
    }
    wait_vmcnt<0>();
    return;
  }

  // ================================================== MFMA waves =================================================
  const int wm = wave_all >> 1, wn = wave_all & 1;
  const int zaddr = (int)(zpiece - smem);
  const int r_lane = wm * 64 + l15;
  const int swz = l15 & 7;
  __syncthreads();  // prologue barrier

  int k = 0, slice = 0, tap = 0, toff = 0, dxc = 0, wstage = 0, wbuf = 0;
  unsigned vmask[PW_MT];
  f32x4 acc[PW_MT][PW_NT];
  auto begin_tile = [&]() {
    const int m0 = tile_m0(k);
#pragma unroll
    for (int i = 0; i < PW_MT; ++i) {
      const int m = m0 + wm * 64 + i * 16 + l15;
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
#pragma unroll
      for (int j = 0; j < PW_NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  begin_tile();

  for (int gs = 0; gs < total_steps; ++gs) {
    __builtin_amdgcn_s_barrier();
    const char* wbase = wring + wstage * PW_WSTAGE;
    const char* xbase = smem + wbuf * win_bytes;
    const int r0 = r_lane + toff;
    const int xa0 = (r0 >> 4) * PW_PIECE + (r0 & 15) * 16 + lq * 256;
    __builtin_amdgcn_s_setprio(1);
    {
      f16x8 xa[2][PW_MT], wb[2][PW_NT];
      auto load_frags = [&](int kh, f16x8* xr, f16x8* wr) {
        const int coff = ((kh * 4 + lq) ^ swz) * 16;
#pragma unroll
        for (int j = 0; j < PW_NT; ++j) wr[j] = *(const f16x8*)(wbase + (wn * 64 + j * 16 + l15) * 128 + coff);
#pragma unroll
        for (int i = 0; i < PW_MT; ++i) {
          const bool ok = (vmask[i] >> tap) & 1u;
          const char* a = ok ? xbase + (xa0 + i * PW_PIECE + kh * 1024) : smem + zaddr;
          xr[i] = *(const f16x8*)a;
        }
      };
      load_frags(0, xa[0], wb[0]);
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        if (kh == 0) load_frags(1, xa[1], wb[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PW_MT; ++i)
#pragma unroll
          for (int j = 0; j < PW_NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[kh][j], xa[kh][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if (++wstage == WS) wstage = 0;
    if (++tap == 9) {
      tap = 0;
      toff = 0;
      dxc = 0;
      wbuf ^= 1;
      if (++slice == nslices) {
        // ---- tile done: epilogue straight from the accumulators (lane = pixel l15 of m-tile i, 4 channels of n-tile j)
        const int m0 = tile_m0(k), n0 = tile_n0(k);
#pragma unroll
        for (int i = 0; i < PW_MT; ++i) {
          const int m = m0 + wm * 64 + i * 16 + l15;
          if (m < p.M) {
            const unsigned b = (unsigned)m / (unsigned)p.HoWo;
            const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
            long long rbase = 0;
            if (p.res_mode == OD_RES_SAME) {
              rbase = (long long)m * p.Cout;
            } else if (p.res_mode == OD_RES_UP2) {
              const unsigned ho = pix / (unsigned)p.Wo, wo = pix - ho * (unsigned)p.Wo;
              rbase = ((long long)(b * (unsigned)(p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout;
            }
            const long long obase = (long long)b * p.obs + (long long)pix * p.ops;
#pragma unroll
            for (int j = 0; j < PW_NT; ++j) {
              const int n = n0 + wn * 64 + j * 16 + lq * 4;
              if (n < p.Cout) {
                const f32x4 sc = *(const f32x4*)(p.scale + n), bi = *(const f32x4*)(p.bias + n);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float t = acc[i][j][e] * sc[e] + bi[e];
                  if (p.act == OD_ACT_LEAKY) t = t > 0.f ? t : t * p.alpha;
                  else if (p.act == OD_ACT_ELU) t = t > 0.f ? t : p.alpha * od_expm1_fast(t);
                  v[e] = t;
                }
                if (p.res_mode != OD_RES_NONE) {
                  const f16x4 r = *(const f16x4*)(p.res + rbase + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] += (float)r[e];
                }
                if (p.out_f32) {
                  *(f32x4*)((float*)p.out + obase + n) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                  f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                  *(f16x4*)((f16*)p.out + obase + n) = h;
                }
              }
            }
          }
        }
        slice = 0;
        ++k;
        if (k < nmy) begin_tile();
      }
    } else if (++dxc == 3) {
      dxc = 0;
      toff += W - 2;
    } else {
      ++toff;
    }
  }
}

}  // namespace

bool od_conv_pw_select(const ConvKP& p, int num_cu, ConvKernelInfo* info, size_t* lds_bytes, int* np_out, int* grid,
                       int* ntiles_total) {
  if (p.stride != 1 || p.pad != 1 || (p.Cin & 63) != 0 || p.Ho != p.H || p.Wo != p.W || p.tconv) return false;
  if ((p.Cout & 3) != 0) return false;
  const int np = (PW_BM + 2 * p.W + 2 + 15) / 16;
  if (np * 2 > 8 * 4 * PW_KW) return false;  // the next window must stream in 8 steps
  constexpr int WS = 3;
  const size_t lds = (size_t)2 * np * PW_PIECE + (size_t)WS * PW_WSTAGE + 1024;
  if (lds > 160 * 1024) return false;
  const int mt = (p.M + PW_BM - 1) / PW_BM, nt = (p.Cout + PW_BN - 1) / PW_BN;
  info->fn = (const void*)&od_conv3x3_pw<WS>;
  info->name = "od_conv3x3_pw<3>";
  info->BM = PW_BM;
  info->BN = PW_BN;
  info->threads = 768;
  *lds_bytes = lds;
  *np_out = np;
  *ntiles_total = mt * nt;
  const int cus = num_cu > 0 ? num_cu : 256;
  *grid = mt * nt < cus ? mt * nt : cus;
  return true;
}
