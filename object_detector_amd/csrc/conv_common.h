// Shared by the convolution translation units (conv_mfma.hip: implicit GEMM; conv_win.hip: LDS-window direct 3x3).
#pragma once
#include "common.h"

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
  int Kstride, Ktot, M, HoWo;
  int act;
  float alpha;
  int res_mode, out_f32;
  long long obs, ops;
  int mtiles, ntiles;
  int splitk, steps_per_split;  // split-K (small-M layers): each workgroup reduces a K range, f32 atomics into ws
  float* ws;                    // [splitk][M][Cout] f32 partial slabs; od_conv_finish sums them and applies the epilogue
  int Mq;             // transposed mode: B*Hs*Ws source positions (rows per parity class)
  int tconv, Hs, Ws;  // transposed (backward-data of a stride-2 conv): x is [B,Hs,Ws,Cin], gathered through a 2x zero-upsampled view
  unsigned x_bytes, w_bytes;  // extents of x / packed w (buffer-addressed loaders: out-of-range lanes read zeros)
  int dbg;  // tuning ablations (OD_CONV_DEBUG): bit0 = skip the DMA, bit1 = skip fragment reads + MFMA
  // training forward (od_conv_desc.bn_partials): per-channel partial sums of the STORED f16 outputs of this tile, row
  // mtile of [mtiles][2][Cout] f32 = (sum z, sum z^2): the BatchNorm statistics pass over z disappears
  float* stats;
  // the pointwise layer that consumes this launch's output, run inside the epilogue (od_conv_desc.w2; 8-wave kernel only):
  // out2[m][0..Cout2) = act2(scale2 * (out[m][0..Cout) . w2) + bias2), f16 dense NHWC
  const f16* w2;
  const float* scale2;
  const float* bias2;
  f16* out2;
  int Cout2, act2;
  float alpha2;
  unsigned w2_bytes;
  int K2stride;
  // grouped launch (od_conv_desc.nseg > 1; 8-wave kernel only): the same layer over nseg input maps of different sizes,
  // m-tiles of segment s = [seg_tile0[s], seg_tile0[s + 1]); a tile never straddles two segments
  int nseg;
  int seg_tile0[4];
  const f16* seg_x[3];
  void* seg_out[3];
  int seg_H[3], seg_W[3], seg_M[3];
};

static __device__ __forceinline__ void glds16(const void* gptr, void* lptr) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lptr, 16, 0, 0);
}

// expm1 for the negative branch of ELU in a conv epilogue: v_exp_f32 away from 0, a degree-6 Taylor polynomial for
// |v| < 0.25 (where exp(v) - 1 would cancel).  Relative error < 1e-6, far below the one f16 rounding that follows;
// libm's expm1f costs ~10x the instructions, which a one-workgroup-per-CU epilogue cannot hide.
static __device__ __forceinline__ float od_expm1_fast(float v) {
  const float poly = v * (1.f + v * (0.5f + v * (1.f / 6 + v * (1.f / 24 + v * (1.f / 120 + v * (1.f / 720))))));
  return v > -0.25f ? poly : __expf(v) - 1.f;
}

// v_permlane16_swap through inline asm (hipcc ROCm 7.2 folds repeated __builtin_amdgcn_permlane16_swap calls of an unrolled
// loop into one).  The compiler's hazard recognizer does not see inside an asm statement: the caller must run
// od_mfma_results_ready() once between the last MFMA that writes the swapped registers and the first swap (a VALU read
// of an MFMA result needs up to 18 wait states); the s_nop pair here covers VALU-write -> permlane-read.
static __device__ __forceinline__ void od_permlane16_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
static __device__ __forceinline__ void od_mfma_results_ready() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// LeakyReLU with a slope in [0, 1] (od_conv2d_fwd validates it): max(v, alpha * v) is the same function as the select,
// one v_mul + one v_max instead of compare + multiply + select.
static __device__ __forceinline__ float od_leaky(float v, float alpha) { return fmaxf(v, v * alpha); }

template <int N>
static __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// transposed mode: GEMM row m' -> linear output pixel index (b*HoWo + y*Wo + x), or -1 for a padding row.  Rows are
// grouped by output parity class tile by tile: tile t of BM rows belongs to class t & 3 and covers positions
// [(t >> 2)*BM, +BM) of the SOURCE grid [B, Hs, Ws]; pixel (y, x) = (2i + (cls >> 1), 2j + (cls & 1)).  The four classes of
// one source block are neighbours in the tile order (same XCD / L2: the dZ block is fetched once, and the two 64..128-B
// halves of an output line are written close together); p.M is the padded row count 4 * ceil(Mq / BM) * BM.
static __device__ __forceinline__ int od_tconv_pixel(const ConvKP& p, unsigned m, unsigned BM) {
  const unsigned t = m / BM, r = m - t * BM;
  const unsigned cls = t & 3u, idx = (t >> 2) * BM + r;
  if (idx >= (unsigned)p.Mq) return -1;
  const unsigned HsWs = (unsigned)(p.Hs * p.Ws);
  const unsigned b = idx / HsWs, q = idx - b * HsWs;
  const unsigned i = q / (unsigned)p.Ws, j = q - i * (unsigned)p.Ws;
  return (int)(b * (unsigned)p.HoWo + (2u * i + (cls >> 1)) * (unsigned)p.Wo + 2u * j + (cls & 1u));
}

// ---- epilogue shared by every conv kernel: accumulators -> LDS staging (one wave-row of the tile at a time) ->
//      scale/bias/act (+ residual) in f32 on full NHWC lines, ONE rounding to f16, 16-B stores.
//      acc[i][j][e] holds pixel (wave-row base + i*16 + l15), channel (wn*WTN + j*16 + lq*4 + e).
template <int BN, int WM, int WN, int MT, int NTL, int NT = WM * WN * 64, bool STATS = false>
static __device__ __forceinline__ void conv_epilogue(const ConvKP& p, char* smem, f32x4 (&acc)[MT][NTL], int m0, int n0,
                                                     int tid, int wm, int wn, int l15, int lq) {
  constexpr int WTM = MT * 16, WTN = NTL * 16, SLD = BN + 4;
  float* stg = (float*)smem;
  constexpr int CH = BN / 8;    // 8-channel chunks per row
  constexpr int RPP = NT / CH;  // rows per store pass
  const int c8 = (tid % CH) * 8;
  const int n = n0 + c8;
  float sc[8], bi[8];
  if (!STATS) {
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
  // STATS instantiations (training forward, od_conv_desc.bn_partials): the launcher guarantees the identity epilogue
  // (scale = 1, bias = 0, no activation, no residual, no split-K), so the 16 scale / bias registers are free for the sums
  float st0[STATS ? 8 : 1], st1[STATS ? 8 : 1];
  if (STATS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) st0[e] = st1[e] = 0.f;
  }
  // residual rows are fetched up front (one 16-B load per store pass, all in flight together) so that their HBM/L2
  // latency overlaps the LDS staging instead of serialising pass after pass
  // a wave-row of the tile (WTM rows) is stored in passes of RPP rows; a workgroup with more threads than one wave-row has
  // 8-channel chunks (512 threads on a 64 x 64 tile: RPP = 64 > WTM = 32) makes ONE pass in which only rows < WTM take part
  constexpr int NPASS = (WTM + RPP - 1) / RPP;
  constexpr bool RAGGED = (WTM % RPP) != 0;
  f16x8 resv[WM][NPASS];
  if (p.res_mode != OD_RES_NONE && p.splitk <= 1) {
#pragma unroll
    for (int wr = 0; wr < WM; ++wr)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        int m = m0 + wr * WTM + ps * RPP + tid / CH;
        f16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
        if (RAGGED && ps * RPP + tid / CH >= WTM) m = p.M;
        if (p.tconv && m < p.M) m = od_tconv_pixel(p, (unsigned)m, WM * MT * 16);
        if (m >= 0 && m < p.M && n < p.Cout) {
          long long roff;
          if (p.res_mode == OD_RES_SAME) {
            roff = (long long)m * p.Cout + n;
          } else {
            const unsigned b = (unsigned)m / (unsigned)p.HoWo;
            const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
            const unsigned ho = pix / (unsigned)p.Wo, wo = pix - ho * (unsigned)p.Wo;
            roff = ((long long)(b * (unsigned)(p.Ho >> 1) + (ho >> 1)) * (p.Wo >> 1) + (wo >> 1)) * p.Cout + n;
          }
          r = *(const f16x8*)(p.res + roff);
        }
        resv[wr][ps] = r;
      }
  }
#pragma unroll
  for (int wr = 0; wr < WM; ++wr) {
    if (wm == wr) {
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const int nl = wn * WTN + j * 16 + lq * 4;
#pragma unroll
        for (int i = 0; i < MT; ++i) *(f32x4*)(stg + (i * 16 + l15) * SLD + nl) = acc[i][j];
      }
    }
    __syncthreads();
    if (p.splitk > 1) {
      // split-K: store this workgroup's partial tile into ITS slab of the f32 workspace (plain 16-B stores, no atomics:
      // od_conv_finish sums the slabs in a fixed order, so the result is bit-reproducible); scale / bias / activation /
      // residual happen there too
      float* slab = p.ws + (long long)((int)blockIdx.x % p.splitk) * p.M * p.Cout;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int row = ps * RPP + tid / CH;
        const int m = (RAGGED && row >= WTM) ? p.M : m0 + wr * WTM + row;
        if (m < p.M && n < p.Cout) {
          float* o = slab + (long long)m * p.Cout + n;
          *(f32x4*)o = *(const f32x4*)(stg + row * SLD + c8);
          *(f32x4*)(o + 4) = *(const f32x4*)(stg + row * SLD + c8 + 4);
        }
      }
      if (wr + 1 < WM) __syncthreads();
      continue;
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + tid / CH;
      int m = (RAGGED && row >= WTM) ? p.M : m0 + wr * WTM + row;
      if (p.tconv && m < p.M) m = od_tconv_pixel(p, (unsigned)m, WM * MT * 16);
      if (m >= 0 && m < p.M && n < p.Cout) {
        const f32x4 v0 = *(const f32x4*)(stg + row * SLD + c8);
        const f32x4 v1 = *(const f32x4*)(stg + row * SLD + c8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (!STATS) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + bi[e];
        }
        if (STATS) {
        } else if (p.act == OD_ACT_LEAKY) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = od_leaky(v[e], p.alpha);
        } else if (p.act == OD_ACT_ELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : p.alpha * od_expm1_fast(v[e]);
        }
        if (!STATS && p.res_mode != OD_RES_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)resv[wr][ps][e];
        }
        const unsigned b = (unsigned)m / (unsigned)p.HoWo;
        const unsigned pix = (unsigned)m - b * (unsigned)p.HoWo;
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
          if (STATS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float q = (float)h[e];  // the stored value: what a separate statistics pass over z would read
              st0[e] += q;
              st1[e] += q * q;
            }
          }
        }
      }
    }
    if (wr + 1 < WM) __syncthreads();
  }
  if (STATS) {
    // fixed-order reduction over the threads that own the same 8-channel chunk (tid % CH): butterfly over the lanes of a
    // wave (CH divides 64), then over the waves through LDS; one partial row per m-tile -> bit-reproducible statistics
    static_assert(CH == 8 || CH == 16 || CH == 32, "BN must be 64, 128 or 256");
#pragma unroll
    for (int off = CH; off < 64; off <<= 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        st0[e] += __shfl_xor(st0[e], off, 64);
        st1[e] += __shfl_xor(st1[e], off, 64);
      }
    }
    __syncthreads();  // the staging rows of the last pass have been read
    float* red = (float*)smem;  // [NT / 64][CH][16]
    const int wv = tid >> 6, ln = tid & 63;
    if (ln < CH) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(wv * CH + ln) * 16 + e] = st0[e];
        red[(wv * CH + ln) * 16 + 8 + e] = st1[e];
      }
    }
    __syncthreads();
    if (tid < CH && n < p.Cout) {
      float a[8], b[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = b[e] = 0.f;
      for (int w2 = 0; w2 < NT / 64; ++w2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          a[e] += red[(w2 * CH + tid) * 16 + e];
          b[e] += red[(w2 * CH + tid) * 16 + 8 + e];
        }
      }
      float* row = p.stats + (long long)(m0 / (WM * WTM)) * 2 * p.Cout;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        row[n + e] = a[e];
        row[p.Cout + n + e] = b[e];
      }
    }
  }
}

// a launchable convolution kernel variant
struct ConvKernelInfo {
  const void* fn;
  const char* name;
  int BM, BN, threads;
};

// conv_8ph.hip: 8-wave, BM x 256 tile, one workgroup per CU, staggered wave groups (3x3 and 1x1)
int od_conv_8ph_num_cfgs();
bool od_conv_8ph_select(int idx, const ConvKP& p, int ksize, ConvKernelInfo* info, size_t* lds_bytes);
// conv_tconv.hip: streaming backward-data kernel of the first stride-2 convolution (dZ 64 channels -> dX 32 channels)
bool od_tconv_small_supported(const od_conv_desc* d);
int od_tconv_small_launch(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run);
// conv_rdirect.hip: 3x3, 64 -> 128 channels on large maps: all weights in LDS, pixel operand straight from global memory
bool od_conv_rdirect_supported(const od_conv_desc* d);
int od_conv_rdirect_launch(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run);
// conv_stream3.hip: streaming 3x3 kernels of the 32 <-> 64-channel layers on the largest maps
bool od_conv_stream3_supported(const od_conv_desc* d);
int od_conv_stream3_launch(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name, bool dry_run);
