// K11 (elementwise part) + K12: training-side kernels around the MFMA convolutions.  All HBM-bound; NHWC f16
// activations / gradients, f32 statistics and parameters.  Reductions are deterministic (per-workgroup partials summed
// in a fixed order), so a training step is bit-reproducible on one GPU.
//
//   od_bn_stats        per-channel mean / rstd of z = conv(x) over (B,H,W) (BatchNorm in training mode), folded into the
//                      (scale, shift) pair of the fused form  y = act(scale*z + shift) (+ residual)
//   od_scale_act       that fused elementwise form (the inference conv epilogue, as its own pass)
//   od_bn_bwd          da = dy * act'(.) ; dgamma = sum(da*xhat), dbeta = sum(da) ; dz = gamma*rstd*(da - dbeta/N - xhat*dgamma/N)
//   od_down2_sum_add   gradient of the nearest 2x upsample in the FPN top-down add
//   od_sgd_step        SGD + momentum + weight decay on f32 masters with per-layer LR multipliers
//                      (docs/MODEL.md:84-90: shared layers x 1/share-count, base network x 1/100), re-packs the f16 copies
//
// reference: the Keras BatchNormalization / activation / optimizer machinery behind the (unseen) `fit` of
// tk.dl.od.ObjectDetector; specified here by docs/MODEL.md:19-21,84-90.
#include <string.h>

#include "common.h"

namespace {

constexpr int ROW_UNROLL = 4;      // rows in flight per thread in the row-walking kernels
constexpr int RED_ROWS_MAX = 8192;  // rows per workgroup in the channel reductions (upper bound)

// rows per workgroup: enough workgroups to fill the chip (>= ~2048), a multiple of the row lanes of one workgroup
static inline int rows_per_wg(long long M, int C) {
  const int G = C >> 3;
  const int lanes = 256 / (G < 256 ? G : 256);
  long long r = (M + 1023) / 1024;  // <= ~1024 partial rows for the final pass
  r = (r + lanes - 1) / lanes * lanes;
  if (r < lanes) r = lanes;
  if (r > RED_ROWS_MAX) r = RED_ROWS_MAX;
  return (int)r;
}

// the channel reductions write one partial row per workgroup and od_chan_final walks those rows: at least 32 KiB of
// tensor per workgroup, so that a small layer does not produce as many bytes of partials as it has data
static inline int rows_per_wg_reduce(long long M, int C) {
  const int G = C >> 3;
  const int lanes = 256 / (G < 256 ? G : 256);
  long long r = (M + 1023) / 1024;
  const long long rmin = (32768 + 2 * C - 1) / (2 * C);
  if (r < rmin) r = rmin;
  r = (r + lanes - 1) / lanes * lanes;
  if (r > RED_ROWS_MAX) r = RED_ROWS_MAX;
  return (int)r;
}

// The activation is a TEMPLATE parameter of the elementwise kernels (round 2): as a run-time enum tested per element the
// compiler kept the dispatch as control flow inside the unrolled loops (207 / 203 / 183 branches in od_scale_act_k /
// od_bn_bwd_apply_k / od_chan_reduce<1>, the libm expm1f / expf paths laid out for every element even for LeakyReLU).
template <int ACT>
__device__ __forceinline__ float act_fwd(float a, float alpha) {
  if (ACT == OD_ACT_LEAKY) return a > 0.f ? a : a * alpha;
  if (ACT == OD_ACT_ELU) return a > 0.f ? a : alpha * expm1f(a);
  return a;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float a, float alpha) {
  if (ACT == OD_ACT_LEAKY) return a > 0.f ? 1.f : alpha;
  if (ACT == OD_ACT_ELU) return a > 0.f ? 1.f : alpha * expf(a);
  return 1.f;
}
#define OD_ACT_SWITCH(act, STMT)                         \
  do {                                                   \
    if ((act) == OD_ACT_LEAKY) {                         \
      constexpr int A = OD_ACT_LEAKY;                    \
      STMT;                                              \
    } else if ((act) == OD_ACT_ELU) {                    \
      constexpr int A = OD_ACT_ELU;                      \
      STMT;                                              \
    } else {                                             \
      constexpr int A = OD_ACT_LINEAR;                   \
      STMT;                                              \
    }                                                    \
  } while (0)

// ---- channel reductions: thread = (8-channel group, row lane); LDS tree over row lanes; one partial row per workgroup
// MODE 0: (sum z, sum z^2)   MODE 1: (sum da, sum da*xhat)
template <int MODE, int ACT>
__global__ __launch_bounds__(256) void od_chan_reduce(const f16* __restrict__ z, const f16* __restrict__ dy,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      long long M, int C, int act, float alpha, int rows_wg,
                                                      float* __restrict__ partials) {
  extern __shared__ float red[];  // [2][256][8]
  const int G = C >> 3;                       // channel groups
  const int tid = threadIdx.x;
  const int lanes = 256 / min(G, 256);        // row lanes per group inside one pass
  const long long r0 = (long long)blockIdx.x * rows_wg;
  const long long r1 = min(r0 + rows_wg, M);
  for (int g0 = 0; g0 < G; g0 += 256) {       // C > 2048 loops (not used by this network)
    const int g = g0 + (lanes > 1 ? tid % G : tid);
    const int rl = lanes > 1 ? tid / G : 0;
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    if (g < G && rl < lanes) {
      float sc[8], sh[8], mu[8], rs[8];
      if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sc[e] = scale[g * 8 + e];
          sh[e] = shift[g * 8 + e];
          mu[e] = mean[g * 8 + e];
          rs[e] = rstd[g * 8 + e];
        }
      }
      // ROW_UNROLL independent 16-B loads per tensor in flight per thread (one load per trip left the kernel at ~1.5 TB/s)
      for (long long r = r0 + rl; r < r1; r += (long long)ROW_UNROLL * lanes) {
        f16x8 zv[ROW_UNROLL], dv[ROW_UNROLL];
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
          const long long rr = r + (long long)u * lanes;
          if (rr < r1) {
            zv[u] = *(const f16x8*)(z + rr * C + g * 8);
            if (MODE == 1) dv[u] = *(const f16x8*)(dy + rr * C + g * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < ROW_UNROLL; ++u) {
          if (r + (long long)u * lanes >= r1) break;
          if (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = (float)zv[u][e];
              s0[e] += v;
              s1[e] += v * v;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float zf = (float)zv[u][e];
              const float da = (float)dv[u][e] * act_grad<ACT>(zf * sc[e] + sh[e], alpha);
              s0[e] += da;
              s1[e] += da * ((zf - mu[e]) * rs[e]);
            }
          }
        }
      }
    }
    // fixed-order sum over the row lanes of each group: butterfly inside the wave when the G groups tile a wave
    // (power-of-two G < 64), then over the remaining holders (waves / row lanes) through LDS
    int hidx = rl, nh = lanes;
    bool holder = g < G && rl < lanes;
    if (lanes > 1 && G < 64 && (G & (G - 1)) == 0) {
      for (int off = G; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s0[e] += __shfl_xor(s0[e], off, 64);
          s1[e] += __shfl_xor(s1[e], off, 64);
        }
      }
      hidx = tid >> 6;
      nh = 4;
      holder = (tid & 63) < G;
    }
    if (holder) {
      float* r0p = red + (hidx * G + (g - g0)) * 8;
      float* r1p = r0p + 256 * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        r0p[e] = s0[e];
        r1p[e] = s1[e];
      }
    }
    __syncthreads();
    if (holder && hidx == 0) {
      for (int l = 1; l < nh; ++l) {
        const float* a = red + (l * G + (g - g0)) * 8;
        const float* b = a + 256 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s0[e] += a[e];
          s1[e] += b[e];
        }
      }
      float* out = partials + (long long)blockIdx.x * 2 * C;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        out[g * 8 + e] = s0[e];
        out[C + g * 8 + e] = s1[e];
      }
    }
    __syncthreads();
  }
}

// fixed-order final sums; MODE 0 -> mean, rstd, scale, shift (and running stats); MODE 1 -> dgamma, dbeta
template <int MODE>
__global__ __launch_bounds__(256) void od_chan_final(const float* __restrict__ partials, int nblocks, int C, float invM,
                                                     float eps, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ o0,
                                                     float* __restrict__ o1, float* __restrict__ o2,
                                                     float* __restrict__ o3, float* __restrict__ run_mean,
                                                     float* __restrict__ run_var, float momentum) {
  // 8 channels per workgroup x 32 partial lanes; fixed summation order (lane-strided partials, then lanes 0..31)
  __shared__ float red[2][32][8];
  const int cl = threadIdx.x & 7, ln = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  float a = 0.f, b = 0.f;
  if (c < C) {
    for (int i = ln; i < nblocks; i += 128) {  // 4 independent loads per sum in flight (was one: ~10 us of latency chain)
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ii = i + 32 * u;
        av[u] = ii < nblocks ? partials[(long long)ii * 2 * C + c] : 0.f;
        bv[u] = ii < nblocks ? partials[(long long)ii * 2 * C + C + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a += av[u];
        b += bv[u];
      }
    }
  }
  red[0][ln][cl] = a;
  red[1][ln][cl] = b;
  __syncthreads();
  if (ln != 0 || c >= C) return;
  a = 0.f;
  b = 0.f;
#pragma unroll
  for (int l = 0; l < 32; ++l) {
    a += red[0][l][cl];
    b += red[1][l][cl];
  }
  if (MODE == 0) {
    const float mu = a * invM;
    const float var = fmaxf(b * invM - mu * mu, 0.f);
    const float rs = rsqrtf(var + eps);
    const float sc = gamma[c] * rs;
    o0[c] = mu;
    o1[c] = rs;
    o2[c] = sc;
    o3[c] = beta[c] - mu * sc;
    if (run_mean) {
      run_mean[c] = momentum * run_mean[c] + (1.f - momentum) * mu;
      run_var[c] = momentum * run_var[c] + (1.f - momentum) * var;
    }
  } else {
    o0[c] += b;  // dgamma (accumulates: the prediction module is shared by three levels)
    o1[c] += a;  // dbeta
    o2[c] = b;   // this call's sums, consumed by the apply pass
    o3[c] = a;
  }
}

// Elementwise passes over [M, C]: thread = (fixed 8-channel group g, row lane); the per-channel constants are loaded
// once into registers and the thread walks rows -- a grid-stride loop would re-load 8 x (2..6) floats per 16 B of data.
template <int ACT>
__global__ __launch_bounds__(256) void od_scale_act_k(const f16* __restrict__ z, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const f16* __restrict__ res,
                                                      f16* __restrict__ y, long long M, int C, int act, float alpha,
                                                      int res_up2, int H, int W, int rows_wg) {
  const int G = C >> 3, tid = threadIdx.x;
  const int lanes = 256 / min(G, 256);
  const int g = lanes > 1 ? tid % G : tid, rl = lanes > 1 ? tid / G : 0;
  if (g >= G || rl >= lanes) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[g * 8 + e];
    sh[e] = shift[g * 8 + e];
  }
  const long long r0 = (long long)blockIdx.x * rows_wg, r1 = min(r0 + rows_wg, M);
  for (long long r = r0 + rl; r < r1; r += (long long)ROW_UNROLL * lanes) {
    f16x8 zv[ROW_UNROLL], rv[ROW_UNROLL];
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) {
      const long long ru = r + (long long)u * lanes;
      if (ru >= r1) break;
      zv[u] = *(const f16x8*)(z + ru * C + g * 8);
      if (res) {
        long long rr = ru;
        if (res_up2) {
          long long pix = ru;
          const int x = (int)(pix % W);
          pix /= W;
          const int yy = (int)(pix % H);
          const long long b = pix / H;
          rr = (b * (H >> 1) + (yy >> 1)) * (W >> 1) + (x >> 1);
        }
        rv[u] = *(const f16x8*)(res + rr * C + g * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) {
      const long long ru = r + (long long)u * lanes;
      if (ru >= r1) break;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_fwd<ACT>((float)zv[u][e] * sc[e] + sh[e], alpha);
      if (res) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rv[u][e];
      }
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
      *(f16x8*)(y + ru * C + g * 8) = o;
    }
  }
}

// dz = gamma*rstd*(da - dbeta/N - xhat*dgamma/N), da = dy*act'(scale*z+shift); bn == 0: dz = da (conv bias layer)
template <int ACT>
__global__ __launch_bounds__(256) void od_bn_bwd_apply_k(const f16* __restrict__ z, const f16* __restrict__ dy,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ sum_dax, const float* __restrict__ sum_da,
                                                         f16* __restrict__ dz, long long M, int C, float invM, int act,
                                                         float alpha, int bn, int rows_wg) {
  const int G = C >> 3, tid = threadIdx.x;
  const int lanes = 256 / min(G, 256);
  const int g = lanes > 1 ? tid % G : tid, rl = lanes > 1 ? tid / G : 0;
  if (g >= G || rl >= lanes) return;
  float sc[8], sh[8], mu[8], rs[8], k1[8], k2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = g * 8 + e;
    // all six pointers are valid for bn == 0 too (the host passes stand-ins): unconditional loads, ONE wait --
    // per-element `bn ? p[c] : 0` compiled to 48 serialised load round trips (a 20 us floor per launch)
    sc[e] = scale[c];
    sh[e] = shift[c];
    mu[e] = mean[c];
    rs[e] = rstd[c];
    k1[e] = sum_da[c] * invM;
    k2[e] = sum_dax[c] * invM;
  }
  const long long r0 = (long long)blockIdx.x * rows_wg, r1 = min(r0 + rows_wg, M);
  for (long long r = r0 + rl; r < r1; r += (long long)ROW_UNROLL * lanes) {
    f16x8 zv[ROW_UNROLL], dv[ROW_UNROLL];
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) {
      const long long ru = r + (long long)u * lanes;
      if (ru >= r1) break;
      zv[u] = *(const f16x8*)(z + ru * C + g * 8);
      dv[u] = *(const f16x8*)(dy + ru * C + g * 8);
    }
#pragma unroll
    for (int u = 0; u < ROW_UNROLL; ++u) {
      const long long ru = r + (long long)u * lanes;
      if (ru >= r1) break;
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float zf = (float)zv[u][e];
        const float da = (float)dv[u][e] * act_grad<ACT>(zf * sc[e] + sh[e], alpha);
        float rr = da;
        if (bn) rr = sc[e] * (da - k1[e] - ((zf - mu[e]) * rs[e]) * k2[e]);  // scale = gamma*rstd
        o[e] = (f16)rr;
      }
      *(f16x8*)(dz + ru * C + g * 8) = o;
    }
  }
}

// d_up[b,y,x,c] (+)= sum over the 2x2 block of d[b,2y..,2x..,c]
__global__ __launch_bounds__(256) void od_down2_sum_add_k(const f16* __restrict__ d, f16* __restrict__ dup, long long nvec,
                                                          int Hh, int Wh, int C8, int accumulate) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const int g = (int)(i % C8);
    long long pix = i / C8;
    const int x = (int)(pix % Wh);
    pix /= Wh;
    const int y = (int)(pix % Hh);
    const long long b = pix / Hh;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (accumulate) {
      const f16x8 a = *(const f16x8*)(dup + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long long si = ((b * (2 * Hh) + 2 * y + (q >> 1)) * (2 * Wh) + 2 * x + (q & 1)) * C8 + g;
      const f16x8 a = *(const f16x8*)(d + si * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)a[e];
    }
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)v[e];
    *(f16x8*)(dup + i * 8) = o;
  }
}

// f16 NHWC elementwise add: a += b
__global__ __launch_bounds__(256) void od_add_f16_k(f16* __restrict__ a, const f16* __restrict__ b, long long nvec) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const f16x8 x = *(const f16x8*)(a + i * 8), y = *(const f16x8*)(b + i * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)x[e] + (float)y[e]);
    *(f16x8*)(a + i * 8) = o;
  }
}

// SGD + momentum on a flat f32 segment; optional f16 re-pack of conv weights into the forward layout [Cout_pad][Kpad]
// and the backward-data layout [Cin_pad][Kpad_t] (taps flipped, in/out channels swapped)
__global__ __launch_bounds__(256) void od_sgd_k(float* __restrict__ w, float* __restrict__ m, const float* __restrict__ g,
                                                long long n, float lr, float momentum, float wd, float inv_scale) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float grad = g[i] * inv_scale + wd * w[i];
    const float mv = momentum * m[i] + grad;
    m[i] = mv;
    w[i] -= lr * mv;
  }
}

// master weights f32 [Cout][k*k*Cin] (k index = tap*Cin + cin) -> forward f16 [Cout_pad][Kpad] and, for the backward-data
// conv, f16 [Cin_pad][Kpad_t] with wt[ci][(k*k-1-tap)*Cout + co] = w[co][tap*Cin + ci]
__global__ __launch_bounds__(256) void od_pack_w_k(const float* __restrict__ w, f16* __restrict__ wf, f16* __restrict__ wt,
                                                   int Cout, int Cin, int taps, int Kpad, int Kpad_t) {
  const long long n = (long long)Cout * taps * Cin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    const long long r = i / Cin;
    const int tap = (int)(r % taps);
    const int co = (int)(r / taps);
    const f16 v = (f16)w[i];
    wf[(long long)co * Kpad + tap * Cin + ci] = v;
    if (wt) wt[(long long)ci * Kpad_t + (taps - 1 - tap) * Cout + co] = v;
  }
}

// multi-tensor forms: blockIdx.y = tensor, blockIdx.x strides over its elements
__global__ __launch_bounds__(256) void od_sgd_multi_k(float* __restrict__ w, float* __restrict__ m, const float* __restrict__ g,
                                                      const od_sgd_seg* __restrict__ segs, float momentum, float inv_scale,
                                                      const int32_t* __restrict__ skip) {
  if (skip && *skip) return;  // a non-finite gradient was found (od_grad_nonfinite): leave weights and momentum untouched
  const od_sgd_seg sg = segs[blockIdx.y];
  float* ws = w + sg.offset;
  float* ms = m + sg.offset;
  const float* gs = g + sg.offset;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < sg.count; i += (long long)gridDim.x * 256) {
    const float grad = gs[i] * inv_scale + sg.weight_decay * ws[i];
    const float mv = momentum * ms[i] + grad;
    ms[i] = mv;
    ws[i] -= sg.lr * mv;
  }
}

// One 64 (Cout) x 64 (Cin) tile of one tap per trip: coalesced f32 reads, forward-layout rows written straight from the
// registers (8-B stores), the backward-data layout (in/out channels swapped) through an LDS transpose as 16-B stores of
// 8 consecutive output channels -- the per-element form scattered 40 M two-byte writes (580 us per step).
__global__ __launch_bounds__(256) void od_pack_multi_k(const float* __restrict__ wflat, const od_pack_layer* __restrict__ layers) {
  __shared__ f16 tile[64][68];  // [co][ci]
  const od_pack_layer L = layers[blockIdx.y];
  const int taps = L.ksize * L.ksize;
  const int Kpad = (taps * L.Cin + 63) / 64 * 64, Kpad_t = (taps * L.Cout + 63) / 64 * 64;
  const float* w = wflat + L.w_offset;
  f16* wf = (f16*)L.w_fwd;
  f16* wt = (f16*)L.w_bwd;
  const int cob = (L.Cout + 63) >> 6, cib = (L.Cin + 63) >> 6;
  const int ntiles = taps * cob * cib;
  const int tid = threadIdx.x;
  const bool vec = ((L.w_offset | L.Cin) & 3) == 0;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int ib = t % cib;
    const int r = t / cib;
    const int ob = r % cob, tap = r / cob;
    const int co0 = ob * 64, ci0 = ib * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = (tid >> 4) + 16 * k, col = (tid & 15) * 4;
      const int co = co0 + row, ci = ci0 + col;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (co < L.Cout && ci < L.Cin) {
        const float* src = w + ((long long)co * taps + tap) * L.Cin + ci;
        if (vec) {
          const f32x4 q = *(const f32x4*)src;
          v[0] = q[0], v[1] = q[1], v[2] = q[2], v[3] = q[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ci + e < L.Cin) v[e] = src[e];
        }
        f16* dst = wf + (long long)co * Kpad + tap * L.Cin + ci;
        if (vec) {
          typedef f16 f16x4 __attribute__((ext_vector_type(4)));
          *(f16x4*)dst = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ci + e < L.Cin) dst[e] = (f16)v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[row][col + e] = (f16)v[e];
    }
    __syncthreads();
    if (wt) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int cr = (tid >> 3) + 32 * k, c8 = (tid & 7) * 8;
        const int ci = ci0 + cr, co = co0 + c8;
        if (ci < L.Cin && co < L.Cout) {  // Cout % 8 == 0: a group of 8 output channels is all-in or all-out
          f16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = tile[c8 + e][cr];
          *(f16x8*)(wt + (long long)ci * Kpad_t + (taps - 1 - tap) * L.Cout + co) = o;
        }
      }
    }
    __syncthreads();
  }
}

// loss gradient w.r.t. pred (f32 [B,P,C], rows of one pyramid level) -> loss-scaled f16 NHWC gradient of that level's
// prediction conv output [B, rows*C] (rows = H*W*8 priors, C = 2+NC+4 -> H*W x 8*C channels)
__global__ __launch_bounds__(256) void od_pred_grad_level_k(const float* __restrict__ g, f16* __restrict__ dz, int B,
                                                            long long img_stride, long long off, long long n_per_img,
                                                            float scale) {
  const long long total = (long long)B * n_per_img;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long b = i / n_per_img, r = i - b * n_per_img;
    dz[i] = (f16)(g[b * img_stride + off + r] * scale);
  }
}

unsigned grid_for(long long nvec) {
  long long b = (nvec + 255) / 256;
  return (unsigned)(b > 256 * 16 ? 256 * 16 : (b < 1 ? 1 : b));
}

}  // namespace

__global__ void od_bn_fold_k(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                             const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ bias,
                             int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(var[c] + eps);  // correctly rounded divide / sqrt, no contraction: = numpy f32
  scale[c] = sc;
  bias[c] = beta[c] - mean[c] * sc;
}

extern "C" int od_bn_fold(od_ctx* ctx, const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          float* scale, float* bias, int C, void* stream) {
  OD_REQUIRE(ctx && gamma && beta && mean && var && scale && bias, "od_bn_fold: null argument");
  OD_REQUIRE(C > 0, "od_bn_fold: C <= 0");
  hipLaunchKernelGGL(od_bn_fold_k, dim3(od_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps,
                     scale, bias, C);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_conv2d_bwd_data(od_ctx* ctx, const void* dz, const void* w_bwd, const void* dx_accumulate, void* dx, int B,
                                  int Ho, int Wo, int Cin, int Cout, int ksize, int stride, void* stream) {
  OD_REQUIRE(ctx && dz && w_bwd && dx, "od_conv2d_bwd_data: null argument");
  OD_REQUIRE(Cin > 0 && Cin <= 2048, "od_conv2d_bwd_data: Cin out of range (1..2048)");
  od_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = dz;
  d.w = w_bwd;
  d.scale = ctx->ones;
  d.bias = (const float*)ctx->zero_page;
  d.res = dx_accumulate;
  d.res_mode = dx_accumulate ? OD_RES_SAME : OD_RES_NONE;
  d.out = dx;
  d.B = B;
  d.H = Ho;
  d.W = Wo;
  d.Cin = Cout;  // the backward-data conv contracts over the forward conv's output channels
  d.Cout = Cin;
  d.ksize = ksize;
  d.stride = stride;
  d.act = OD_ACT_LINEAR;
  d.out_dtype = OD_DT_F16;
  d.tile_cfg = -1;
  d.transposed = stride == 2;
  d.splitk = 1;
  return od_conv2d_fwd_impl(ctx, &d, (hipStream_t)stream, nullptr, false);
}

extern "C" size_t od_bn_workspace_bytes(long long M, int C) {
  if (M <= 0 || C <= 0 || C % 8) return 0;
  const int rw = rows_per_wg_reduce(M, C);
  return (size_t)((M + rw - 1) / rw) * 2 * C * sizeof(float);
}

extern "C" int od_bn_stats(od_ctx* ctx, const void* z, long long M, int C, const float* gamma, const float* beta,
                           float eps, float* mean, float* rstd, float* scale, float* shift, float* run_mean,
                           float* run_var, float momentum, void* workspace, size_t workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && z && gamma && beta && mean && rstd && scale && shift && workspace, "od_bn_stats: null argument");
  OD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "od_bn_stats: C must be a multiple of 8, <= 2048");
  const int rw = rows_per_wg_reduce(M, C);
  const int nblocks = (int)((M + rw - 1) / rw);
  if (workspace_bytes < od_bn_workspace_bytes(M, C)) {
    od_set_error("od_bn_stats: workspace too small");
    return OD_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  hipLaunchKernelGGL((od_chan_reduce<0, OD_ACT_LINEAR>), dim3(nblocks), dim3(256), 2 * 256 * 8 * sizeof(float), s, (const f16*)z,
                     (const f16*)nullptr, nullptr, nullptr, nullptr, nullptr, M, C, 0, 0.f, rw, part);
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_chan_final<0>, dim3(od_ceil_div(C, 8)), dim3(256), 0, s, part, nblocks, C, 1.f / (float)M, eps,
                     gamma, beta, mean, rstd, scale, shift, run_mean, run_var, momentum);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_bn_stats_from_partials(od_ctx* ctx, const float* partials, int rows, long long M, int C, const float* gamma,
                                         const float* beta, float eps, float* mean, float* rstd, float* scale, float* shift,
                                         float* run_mean, float* run_var, float momentum, void* stream) {
  OD_REQUIRE(ctx && partials && gamma && beta && mean && rstd && scale && shift, "od_bn_stats_from_partials: null argument");
  OD_REQUIRE(rows > 0 && M > 0 && C > 0 && C % 8 == 0, "od_bn_stats_from_partials: bad dims");
  hipLaunchKernelGGL(od_chan_final<0>, dim3(od_ceil_div(C, 8)), dim3(256), 0, (hipStream_t)stream, partials, rows, C,
                     1.f / (float)M, eps, gamma, beta, mean, rstd, scale, shift, run_mean, run_var, momentum);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_scale_act(od_ctx* ctx, const void* z, const float* scale, const float* shift, const void* res,
                            int res_mode, void* y, int B, int H, int W, int C, int act, float alpha, void* stream) {
  OD_REQUIRE(ctx && z && scale && shift && y, "od_scale_act: null argument");
  OD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "od_scale_act: bad dims");
  OD_REQUIRE(res_mode == OD_RES_NONE || res, "od_scale_act: res_mode set but res is null");
  OD_REQUIRE(C <= 2048, "od_scale_act: C <= 2048");
  const long long M = (long long)B * H * W;
  const int rw = rows_per_wg(M, C);
  OD_ACT_SWITCH(act, hipLaunchKernelGGL(od_scale_act_k<A>, dim3((unsigned)((M + rw - 1) / rw)), dim3(256), 0, (hipStream_t)stream,
                                        (const f16*)z, scale, shift,
                                        res_mode == OD_RES_NONE ? (const f16*)nullptr : (const f16*)res, (f16*)y, M, C, act,
                                        alpha, res_mode == OD_RES_UP2, H, W, rw));
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_bn_bwd(od_ctx* ctx, const void* z, const void* dy, const float* scale, const float* shift,
                         const float* mean, const float* rstd, long long M, int C, int act, float alpha, int bn,
                         float* dgamma, float* dbeta, void* dz, void* workspace, size_t workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && z && dy && scale && shift && dz && dgamma && dbeta && workspace, "od_bn_bwd: null argument");
  OD_REQUIRE(!bn || (mean && rstd), "od_bn_bwd: bn needs mean/rstd");
  OD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "od_bn_bwd: C must be a multiple of 8, <= 2048");
  const int rw = rows_per_wg_reduce(M, C), rw_apply = rows_per_wg(M, C);
  const int nblocks = (int)((M + rw - 1) / rw);
  const size_t need = od_bn_workspace_bytes(M, C) + 2 * (size_t)C * sizeof(float);
  if (workspace_bytes < need) {
    od_set_error("od_bn_bwd: workspace %zu < %zu", workspace_bytes, need);
    return OD_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* sums = part + (size_t)nblocks * 2 * C;  // [2][C]: this call's sum(da*xhat), sum(da)
  // without BN the "mean/rstd" are not used by the sums we need (dbeta only); pass scale/shift twice to keep pointers valid
  OD_ACT_SWITCH(act, hipLaunchKernelGGL((od_chan_reduce<1, A>), dim3(nblocks), dim3(256), 2 * 256 * 8 * sizeof(float), s,
                                        (const f16*)z, (const f16*)dy, scale, shift, bn ? mean : shift, bn ? rstd : scale, M,
                                        C, act, alpha, rw, part));
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_chan_final<1>, dim3(od_ceil_div(C, 8)), dim3(256), 0, s, part, nblocks, C, 0.f, 0.f,
                     (const float*)nullptr, (const float*)nullptr, dgamma, dbeta, sums, sums + C, (float*)nullptr,
                     (float*)nullptr, 0.f);
  OD_CHECK_LAUNCH();
  OD_ACT_SWITCH(act, hipLaunchKernelGGL(od_bn_bwd_apply_k<A>, dim3((unsigned)((M + rw_apply - 1) / rw_apply)), dim3(256), 0, s,
                                        (const f16*)z, (const f16*)dy, scale, shift, bn ? mean : shift, bn ? rstd : scale,
                                        sums, sums + C, (f16*)dz, M, C, 1.f / (float)M, act, alpha, bn, rw_apply));
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_down2_sum_add(od_ctx* ctx, const void* d, void* dup, int B, int Hh, int Wh, int C, int accumulate,
                                void* stream) {
  OD_REQUIRE(ctx && d && dup && B > 0 && Hh > 0 && Wh > 0 && C > 0 && C % 8 == 0, "od_down2_sum_add: bad argument");
  const long long nvec = (long long)B * Hh * Wh * (C / 8);
  hipLaunchKernelGGL(od_down2_sum_add_k, dim3(grid_for(nvec)), dim3(256), 0, (hipStream_t)stream, (const f16*)d, (f16*)dup,
                     nvec, Hh, Wh, C / 8, accumulate);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_add_f16(od_ctx* ctx, void* a, const void* b, long long n, void* stream) {
  OD_REQUIRE(ctx && a && b && n > 0 && n % 8 == 0, "od_add_f16: n must be a positive multiple of 8");
  hipLaunchKernelGGL(od_add_f16_k, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (f16*)a, (const f16*)b, n / 8);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_sgd_step(od_ctx* ctx, float* w, float* m, const float* g, long long n, float lr, float momentum,
                           float weight_decay, float inv_loss_scale, void* stream) {
  OD_REQUIRE(ctx && w && m && g && n > 0, "od_sgd_step: bad argument");
  hipLaunchKernelGGL(od_sgd_k, dim3(grid_for((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, w, m, g, n, lr, momentum,
                     weight_decay, inv_loss_scale);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_sgd_step_multi(od_ctx* ctx, float* w, float* m, const float* g, const od_sgd_seg* segs, int nseg,
                                 float momentum, float inv_loss_scale, const int32_t* skip_if_nonzero, void* stream) {
  OD_REQUIRE(ctx && w && m && g && segs && nseg > 0 && nseg <= 65535, "od_sgd_step_multi: bad argument");
  hipLaunchKernelGGL(od_sgd_multi_k, dim3(256, nseg), dim3(256), 0, (hipStream_t)stream, w, m, g, segs, momentum,
                     inv_loss_scale, skip_if_nonzero);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

// flag[0] = 1 when any of g[0..n) is Inf or NaN, else 0 (flag is cleared by the first launch, set by the second)
__global__ void od_flag_clear_k(int32_t* flag) { *flag = 0; }
__global__ __launch_bounds__(256) void od_grad_nonfinite_k(const float* __restrict__ g, long long n4, long long n,
                                                           int32_t* __restrict__ flag) {
  bool bad = false;
  const f32x4* g4 = (const f32x4*)g;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const f32x4 v = g4[i];
    // exponent field all ones <=> Inf or NaN
    bad = bad || (__float_as_uint(v.x) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(v.y) & 0x7f800000u) == 0x7f800000u ||
          (__float_as_uint(v.z) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(v.w) & 0x7f800000u) == 0x7f800000u;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4))
    bad = bad || (__float_as_uint(g[4 * n4 + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

extern "C" int od_grad_nonfinite(od_ctx* ctx, const float* g, long long n, int32_t* flag, void* stream) {
  OD_REQUIRE(ctx && g && flag && n > 0, "od_grad_nonfinite: bad argument");
  OD_REQUIRE(((uintptr_t)g & 15) == 0, "od_grad_nonfinite: g must be 16-byte aligned");
  hipLaunchKernelGGL(od_flag_clear_k, dim3(1), dim3(1), 0, (hipStream_t)stream, flag);
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_grad_nonfinite_k, dim3(grid_for(n / 4 / 4 + 1)), dim3(256), 0, (hipStream_t)stream, g, n / 4, n, flag);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

// dst[0..n) = src[0..n) when *flag != 0 (no-op otherwise): the trainer restores the BatchNorm running statistics of a
// step whose update was skipped (the forward pass that overflowed has already folded its batch statistics into them)
__global__ __launch_bounds__(256) void od_copy_if_k(float* __restrict__ dst, const float* __restrict__ src, long long n,
                                                    const int32_t* __restrict__ flag) {
  if (*flag == 0) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int od_copy_if_nonzero(od_ctx* ctx, float* dst, const float* src, long long n, const int32_t* flag, void* stream) {
  OD_REQUIRE(ctx && dst && src && flag && n > 0, "od_copy_if_nonzero: bad argument");
  hipLaunchKernelGGL(od_copy_if_k, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dst, src, n, flag);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

// f32 <-> bf16 (round to nearest even; Inf / NaN preserved) for the bf16 gradient all-reduce payload (BASELINE.json
// configs[4]; SURVEY.md §0.1 "bf16 for the all-reduce payload")
__global__ __launch_bounds__(256) void od_f32_to_bf16_k(const float* __restrict__ src, uint16_t* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const uint32_t u = __float_as_uint(src[i]);
    uint32_t r;
    if ((u & 0x7fffffffu) > 0x7f800000u) r = (u >> 16) | 0x40u;  // NaN stays NaN
    else r = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    dst[i] = (uint16_t)r;
  }
}
__global__ __launch_bounds__(256) void od_bf16_to_f32_k(const uint16_t* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    dst[i] = __uint_as_float((uint32_t)src[i] << 16);
}

extern "C" int od_cast_f32_bf16(od_ctx* ctx, const float* src, void* dst, long long n, void* stream) {
  OD_REQUIRE(ctx && src && dst && n > 0, "od_cast_f32_bf16: bad argument");
  hipLaunchKernelGGL(od_f32_to_bf16_k, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, n);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_cast_bf16_f32(od_ctx* ctx, const void* src, float* dst, long long n, void* stream) {
  OD_REQUIRE(ctx && src && dst && n > 0, "od_cast_bf16_f32: bad argument");
  hipLaunchKernelGGL(od_bf16_to_f32_k, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, dst,
                     n);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_pack_weights_multi(od_ctx* ctx, const float* w, const od_pack_layer* layers, int nlayers, void* stream) {
  OD_REQUIRE(ctx && w && layers && nlayers > 0 && nlayers <= 65535, "od_pack_weights_multi: bad argument");
  // 1024 x 256 threads per layer: the largest layers (4.7 M weights, transposed 2-byte scatter for the backward pack) need the
  // parallelism; the blocks of small layers exit after one test
  hipLaunchKernelGGL(od_pack_multi_k, dim3(288, nlayers), dim3(256), 0, (hipStream_t)stream, w, layers);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_pack_weights(od_ctx* ctx, const float* w, void* w_fwd, void* w_bwd, int Cout, int Cin, int ksize,
                               void* stream) {
  OD_REQUIRE(ctx && w && w_fwd && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "od_pack_weights: bad argument");
  const int taps = ksize * ksize;
  const int Kpad = od_round_up(taps * Cin, 64), Kpad_t = od_round_up(taps * Cout, 64);
  const long long n = (long long)Cout * taps * Cin;
  hipLaunchKernelGGL(od_pack_w_k, dim3(grid_for((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, w, (f16*)w_fwd,
                     (f16*)w_bwd, Cout, Cin, taps, Kpad, Kpad_t);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_pred_grad_to_level(od_ctx* ctx, const float* grad_pred, void* dz, int B, int P, int C, int row_off,
                                     int rows, float loss_scale, void* stream) {
  OD_REQUIRE(ctx && grad_pred && dz && B > 0 && P > 0 && C > 0 && rows > 0 && row_off >= 0 && row_off + rows <= P,
             "od_pred_grad_to_level: bad argument");
  const long long n = (long long)rows * C;
  hipLaunchKernelGGL(od_pred_grad_level_k, dim3(grid_for(((long long)B * n + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                     grad_pred, (f16*)dz, B, (long long)P * C, (long long)row_off * C, n, loss_scale);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
