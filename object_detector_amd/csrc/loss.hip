// K10: detection loss, forward + gradient w.r.t. the prediction tensor, in one elementwise pass (HBM-bound).
// reference docs/MODEL.md:33-52:
//   objectness : 2-class focal loss (alpha_t, gamma)                    [:33-37]
//   class      : softmax + categorical cross-entropy on assigned priors [:39-44]
//   box        : MSE (doc mode) or smooth-L1 (north_star mode) on the corner-form offsets, assigned priors only [:46-52]
// Rows: pred/y f32 [R, 2+NC+4] (R = B*P); y as written by od_assign_anchors (row all-zero = ignore).
// total = (sum_obj + sum_cls + sum_box) / max(1, #assigned rows); grad is d total / d pred.
// Reduction is deterministic: per-workgroup partials, then one workgroup sums them in a fixed order.
#include "common.h"

namespace {

constexpr int LROWS = 256;

__global__ __launch_bounds__(256) void od_loss_count(const float* __restrict__ y, long long R, int C, int* __restrict__ npos) {
  __shared__ int sc;
  if (threadIdx.x == 0) sc = 0;
  __syncthreads();
  int c = 0;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < R; r += (long long)gridDim.x * 256)
    c += y[r * C + 1] == 1.f;
  if (c) atomicAdd(&sc, c);
  __syncthreads();
  if (threadIdx.x == 0 && sc) atomicAdd(npos, sc);
}

__global__ __launch_bounds__(256) void od_loss_rows(const float* __restrict__ pred, const float* __restrict__ y,
                                                    float* __restrict__ grad, long long R, int NC, float alpha,
                                                    float gamma, int box_mode, float w_obj, float w_cls, float w_box,
                                                    const int* __restrict__ npos, float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = NC + 6, tid = threadIdx.x;
  float* sp = sm;                // [LROWS][C] pred -> grad
  float* sy = sm + LROWS * C;    // [LROWS][C]
  __shared__ float red[3][4];
  const long long r0 = (long long)blockIdx.x * LROWS;
  const int nrows = (int)((R - r0) < LROWS ? (R - r0) : LROWS);
  const int nel = nrows * C;
  for (int i = tid * 4; i < nel; i += 256 * 4) {
    if (i + 3 < nel) {
      *(f32x4*)(sp + i) = *(const f32x4*)(pred + r0 * C + i);
      *(f32x4*)(sy + i) = *(const f32x4*)(y + r0 * C + i);
    } else {
      for (int e = i; e < nel; ++e) {
        sp[e] = pred[r0 * C + e];
        sy[e] = y[r0 * C + e];
      }
    }
  }
  __syncthreads();
  const float invn = 1.f / (float)max(1, *npos);
  float l_obj = 0.f, l_cls = 0.f, l_box = 0.f;
  if (tid < nrows) {
    float* p = sp + tid * C;
    const float* t = sy + tid * C;
    const float t0 = t[0], t1 = t[1];
    // ---- objectness: focal loss over softmax(l0, l1) ----
    const float l0 = p[0], l1 = p[1];
    float g0 = 0.f, g1 = 0.f;
    if (t0 + t1 > 0.f) {
      const float m = fmaxf(l0, l1);
      const float lse = m + logf(expf(l0 - m) + expf(l1 - m));
      const float lp0 = l0 - lse, lp1 = l1 - lse;
      const float p0 = expf(lp0), p1 = expf(lp1);
      const bool pos = t1 > 0.5f;
      const float lpt = pos ? lp1 : lp0, pt = pos ? p1 : p0;
      const float a = pos ? alpha : 1.f - alpha;
      const float om = 1.f - pt;
      const float mod = gamma == 2.f ? om * om : powf(om, gamma);
      const float dmod = gamma == 2.f ? 2.f * om : gamma * powf(om, gamma - 1.f);
      l_obj = -a * mod * lpt;
      const float dl = -a * (mod - dmod * pt * lpt);  // d loss / d log p_t
      // d log p_t / d l_j = delta_tj - p_j
      g0 = dl * ((pos ? 0.f : 1.f) - p0);
      g1 = dl * ((pos ? 1.f : 0.f) - p1);
    }
    p[0] = g0 * w_obj * invn;
    p[1] = g1 * w_obj * invn;
    if (t1 > 0.5f) {
      // ---- class: softmax cross-entropy ----
      float mx = p[2];
      for (int c = 1; c < NC; ++c) mx = fmaxf(mx, p[2 + c]);
      float s = 0.f;
      for (int c = 0; c < NC; ++c) s += expf(p[2 + c] - mx);
      const float lse = mx + logf(s);
      for (int c = 0; c < NC; ++c) {
        const float lg = p[2 + c];
        const float q = expf(lg - lse);
        l_cls -= t[2 + c] * (lg - lse);
        p[2 + c] = (q - t[2 + c]) * w_cls * invn;
      }
      // ---- box ----
      for (int e = 0; e < 4; ++e) {
        const float d = p[2 + NC + e] - t[2 + NC + e];
        float l, g;
        if (box_mode == 0) {  // smooth-L1, beta = 1
          const float ad = fabsf(d);
          l = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
          g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        } else {  // MSE over the 4 coordinates (docs/MODEL.md:46-48)
          l = 0.25f * d * d;
          g = 0.5f * d;
        }
        l_box += l;
        p[2 + NC + e] = g * w_box * invn;
      }
    } else {
      for (int c = 2; c < C; ++c) p[c] = 0.f;
    }
  }
  __syncthreads();
  for (int i = tid * 4; i < nel; i += 256 * 4) {
    if (i + 3 < nel) {
      *(f32x4*)(grad + r0 * C + i) = *(const f32x4*)(sp + i);
    } else {
      for (int e = i; e < nel; ++e) grad[r0 * C + e] = sp[e];
    }
  }
  // deterministic block reduction: wave shuffle tree, then 4 wave partials added in order
  float v[3] = {l_obj, l_cls, l_box};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    if ((tid & 63) == 0) red[k][tid >> 6] = v[k];
  }
  __syncthreads();
  if (tid < 3) partials[(long long)blockIdx.x * 3 + tid] = ((red[tid][0] + red[tid][1]) + red[tid][2]) + red[tid][3];
}

__global__ __launch_bounds__(256) void od_loss_final(const float* __restrict__ partials, int nblocks,
                                                     const int* __restrict__ npos, float w_obj, float w_cls, float w_box,
                                                     float* __restrict__ losses) {
  __shared__ float red[3][256];
  const int tid = threadIdx.x;
  float a[3] = {0.f, 0.f, 0.f};
  for (int i = tid; i < nblocks; i += 256)
    for (int k = 0; k < 3; ++k) a[k] += partials[(long long)i * 3 + k];
  for (int k = 0; k < 3; ++k) red[k][tid] = a[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s)
      for (int k = 0; k < 3; ++k) red[k][tid] += red[k][tid + s];
    __syncthreads();
  }
  if (tid == 0) {
    const float invn = 1.f / (float)max(1, *npos);
    losses[0] = red[0][0] * w_obj * invn;
    losses[1] = red[1][0] * w_cls * invn;
    losses[2] = red[2][0] * w_box * invn;
    losses[3] = (losses[0] + losses[1]) + losses[2];
  }
}

}  // namespace

extern "C" size_t od_loss_workspace_bytes(int B, int P) {
  if (B <= 0 || P <= 0) return 0;
  const long long R = (long long)B * P;
  return 256 + (size_t)((R + LROWS - 1) / LROWS) * 3 * sizeof(float);
}

extern "C" int od_loss_fwd_bwd(od_ctx* ctx, const float* pred, const float* y, float* grad, float* losses, int B, int P,
                               int NC, float focal_alpha, float focal_gamma, int box_mode, float w_obj, float w_cls,
                               float w_box, void* workspace, size_t workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && pred && y && grad && losses && workspace, "od_loss_fwd_bwd: null argument");
  OD_REQUIRE(B > 0 && P > 0 && NC > 0 && NC <= 90, "od_loss_fwd_bwd: bad dims");
  OD_REQUIRE(box_mode == 0 || box_mode == 1, "od_loss_fwd_bwd: box_mode 0 = smooth-L1, 1 = MSE");
  const long long R = (long long)B * P;
  const int nblocks = (int)((R + LROWS - 1) / LROWS);
  const size_t need = 256 + (size_t)nblocks * 3 * sizeof(float);
  if (workspace_bytes < need) {
    od_set_error("od_loss_fwd_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
    return OD_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  int* npos = (int*)workspace;
  float* partials = (float*)((char*)workspace + 256);
  OD_CHECK_HIP(hipMemsetAsync(npos, 0, 4, s));
  int cb = (int)((R + 255) / 256);
  if (cb > 2048) cb = 2048;
  hipLaunchKernelGGL(od_loss_count, dim3(cb), dim3(256), 0, s, y, R, NC + 6, npos);
  OD_CHECK_LAUNCH();
  const size_t lds = (size_t)2 * LROWS * (NC + 6) * sizeof(float);
  if (int rc = od_ensure_lds(ctx, (const void*)&od_loss_rows, lds)) return rc;
  hipLaunchKernelGGL(od_loss_rows, dim3(nblocks), dim3(256), lds, s, pred, y, grad, R, NC, focal_alpha, focal_gamma,
                     box_mode, w_obj, w_cls, w_box, npos, partials);
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_loss_final, dim3(1), dim3(256), 0, s, partials, nblocks, npos, w_obj, w_cls, w_box, losses);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
