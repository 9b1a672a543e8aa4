// K9: prior-box assignment + target encoding = `od.pb.encode_truth` (reference check_assign.py:21,25-27).
// Per image: GT boxes x priors IoU, argmax both ways, dense target rows y[P, 2+NC+4]:
//   y[p,0]=1 background | y[p,1]=1 assigned (check_assign.py:25) | y[p,2:2+NC] one-hot class (:26) |
//   y[p,-4:] corner-form regression target, the inverse of decode_locs (:27)
// [BUILD-DEFINED] rule (the reference does not pin it; mirrored op-for-op by oracle/assign.py):
//   1. prior p takes GT g* = argmax_g IoU(g,p) (ties: lowest g) if IoU >= pos_thr; if neg_thr <= IoU < pos_thr the
//      row is "ignore" (all zeros: no objectness loss); otherwise background
//   2. every GT g (ascending g, later wins) force-takes p* = argmax_p IoU(g,p) (ties: lowest p) when that IoU > 0
// HBM-bound elementwise work; IoU uses IEEE f32 division; compiled with -ffp-contract=off => bit-exact vs numpy.
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int GMAX = 128;

__device__ __forceinline__ float iou_f32(const f32x4 a, const f32x4 c) {
  const float ix1 = fmaxf(a[0], c[0]), iy1 = fmaxf(a[1], c[1]);
  const float ix2 = fminf(a[2], c[2]), iy2 = fminf(a[3], c[3]);
  const float iw = fmaxf(ix2 - ix1, 0.f), ih = fmaxf(iy2 - iy1, 0.f);
  const float inter = iw * ih;
  const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
  const float area_c = (c[2] - c[0]) * (c[3] - c[1]);
  const float uni = (area_a + area_c) - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

// pass 1: per prior best GT; per GT best prior (u64 atomicMax on (iou_bits << 32 | ~p))
__global__ __launch_bounds__(256) void od_assign_match(const float* __restrict__ priors, const float* __restrict__ gt_boxes,
                                                       const int* __restrict__ gt_counts, int P, int Gmax,
                                                       int* __restrict__ best_g, float* __restrict__ best_iou,
                                                       u64* __restrict__ gt_best) {
  __shared__ f32x4 sg[GMAX];
  __shared__ u64 sbest[GMAX];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int G = min(gt_counts[b], Gmax);
  for (int g = tid; g < G; g += 256) {
    sg[g] = *(const f32x4*)(gt_boxes + ((long long)b * Gmax + g) * 4);
    sbest[g] = 0ull;
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + tid;
  if (p < P) {
    const f32x4 pr = *(const f32x4*)(priors + (long long)p * 4);
    int bg = -1;
    float bi = 0.f;
    for (int g = 0; g < G; ++g) {
      const float v = iou_f32(sg[g], pr);
      if (v > bi) {
        bi = v;
        bg = g;
      }
      if (v > 0.f) atomicMax(&sbest[g], ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)p));
    }
    best_g[(long long)b * P + p] = bg;
    best_iou[(long long)b * P + p] = bi;
  }
  __syncthreads();
  for (int g = tid; g < G; g += 256)
    if (sbest[g]) atomicMax(&gt_best[(long long)b * Gmax + g], sbest[g]);
}

// pass 2: resolve + encode dense rows; counts positives per image
__global__ __launch_bounds__(256) void od_assign_encode(const float* __restrict__ priors, const float* __restrict__ gt_boxes,
                                                        const int* __restrict__ gt_classes,
                                                        const int* __restrict__ gt_counts, int P, int Gmax, int NC,
                                                        float pos_thr, float neg_thr, float loc_scale,
                                                        const int* __restrict__ best_g, const float* __restrict__ best_iou,
                                                        const u64* __restrict__ gt_best, float* __restrict__ y,
                                                        int* __restrict__ assigned_gt, int* __restrict__ npos) {
  __shared__ unsigned sforce[GMAX];
  __shared__ int scount;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int G = min(gt_counts[b], Gmax);
  for (int g = tid; g < G; g += 256) {
    const u64 k = gt_best[(long long)b * Gmax + g];
    sforce[g] = k ? 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull) : 0xFFFFFFFFu;
  }
  if (tid == 0) scount = 0;
  __syncthreads();
  const int p = blockIdx.x * 256 + tid;
  if (p < P) {
    const int C = NC + 6;
    int g = -1;        // assigned GT
    bool ignore = false;
    const int bg = best_g[(long long)b * P + p];
    const float bi = best_iou[(long long)b * P + p];
    if (bg >= 0 && bi >= pos_thr) g = bg;
    else if (bg >= 0 && bi >= neg_thr) ignore = true;
    for (int q = 0; q < G; ++q)
      if (sforce[q] == (unsigned)p) g = q;  // later GT wins
    float* row = y + ((long long)b * P + p) * C;
    for (int c = 0; c < C; ++c) row[c] = 0.f;
    if (g >= 0) {
      const f32x4 pr = *(const f32x4*)(priors + (long long)p * 4);
      const f32x4 gb = *(const f32x4*)(gt_boxes + ((long long)b * Gmax + g) * 4);
      const float pw = pr[2] - pr[0], ph = pr[3] - pr[1];
      row[1] = 1.f;
      const int cls = gt_classes[(long long)b * Gmax + g];
      if (cls >= 0 && cls < NC) row[2 + cls] = 1.f;
      row[2 + NC + 0] = ((gb[0] - pr[0]) / pw) / loc_scale;
      row[2 + NC + 1] = ((gb[1] - pr[1]) / ph) / loc_scale;
      row[2 + NC + 2] = ((gb[2] - pr[2]) / pw) / loc_scale;
      row[2 + NC + 3] = ((gb[3] - pr[3]) / ph) / loc_scale;
      atomicAdd(&scount, 1);
    } else if (!ignore) {
      row[0] = 1.f;
    }
    if (assigned_gt) assigned_gt[(long long)b * P + p] = g >= 0 ? g : (ignore ? -2 : -1);
  }
  __syncthreads();
  if (tid == 0 && scount) atomicAdd(&npos[b], scount);
}

struct AssignLayout {
  size_t best_g, best_iou, gt_best, total;
};
AssignLayout assign_layout(int B, int P, int Gmax) {
  AssignLayout l;
  size_t o = 0;
  l.gt_best = o;
  o += ((size_t)B * Gmax * 8 + 255) & ~(size_t)255;
  l.best_g = o;
  o += (size_t)B * P * 4;
  l.best_iou = o;
  o += (size_t)B * P * 4;
  l.total = o;
  return l;
}

}  // namespace

extern "C" size_t od_assign_workspace_bytes(int B, int P, int Gmax) {
  if (B <= 0 || P <= 0 || Gmax <= 0) return 0;
  return assign_layout(B, P, Gmax).total;
}

extern "C" int od_assign_anchors(od_ctx* ctx, const float* priors, const float* gt_boxes, const int32_t* gt_classes,
                                 const int32_t* gt_counts, int B, int P, int Gmax, int NC, float pos_thr, float neg_thr,
                                 float loc_scale, float* y, int32_t* assigned_gt, int32_t* npos, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && priors && gt_boxes && gt_classes && gt_counts && y && npos && workspace,
             "od_assign_anchors: null argument");
  OD_REQUIRE(B > 0 && B <= 65535 && P > 0 && NC > 0 && Gmax > 0 && Gmax <= GMAX,
             "od_assign_anchors: bad dims (Gmax <= %d)", GMAX);
  OD_REQUIRE(loc_scale > 0.f && pos_thr >= neg_thr, "od_assign_anchors: bad thresholds");
  const AssignLayout l = assign_layout(B, P, Gmax);
  if (workspace_bytes < l.total) {
    od_set_error("od_assign_anchors: workspace %zu < %zu bytes", workspace_bytes, l.total);
    return OD_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  hipStream_t s = (hipStream_t)stream;
  OD_CHECK_HIP(hipMemsetAsync(ws + l.gt_best, 0, (size_t)B * Gmax * 8, s));
  OD_CHECK_HIP(hipMemsetAsync(npos, 0, (size_t)B * 4, s));
  dim3 grid(od_ceil_div(P, 256), B);
  hipLaunchKernelGGL(od_assign_match, grid, dim3(256), 0, s, priors, gt_boxes, gt_counts, P, Gmax,
                     (int*)(ws + l.best_g), (float*)(ws + l.best_iou), (u64*)(ws + l.gt_best));
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_assign_encode, grid, dim3(256), 0, s, priors, gt_boxes, gt_classes, gt_counts, P, Gmax, NC,
                     pos_thr, neg_thr, loc_scale, (const int*)(ws + l.best_g), (const float*)(ws + l.best_iou),
                     (const u64*)(ws + l.gt_best), y, assigned_gt, npos);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
