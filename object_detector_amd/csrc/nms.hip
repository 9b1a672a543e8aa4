// K8: class-aware greedy NMS on the K best candidates of every image (docs/MODEL.md:78-82: among the confident
// predictions, same-class overlaps keep only the most confident).  Integer / bit work, latency-bound; three launches:
//   sort : one workgroup per image, bitonic sort of the K 64-bit keys in LDS (descending = (conf desc, flat asc)),
//          gather of the candidates' boxes / classes into rank order
//   mask : (K/64 x B) workgroups of 16 waves; lane = row i, each wave walks one 64-column word and builds the suppression bitmask
//          M[i][w] bit j = (j > i && same class && inter > thr * union) -- one u64 per (row, 64-column block)
//   scan : the image's mask is staged in LDS (<= 128 KiB), then ONE wavefront runs the greedy scan: each 64x64 diagonal
//          block is resolved with scalar 64-bit ops on SGPRs (v_readlane; only rows that suppress something take a
//          step), kept rows OR their mask rows into the running `removed` words, 16 LDS reads in flight per lane
// The IoU predicate is division-free f32 arithmetic in a fixed order; this TU is compiled with -ffp-contract=off so
// it is bit-identical to numpy's (oracle/nms.py), which is what makes the kept-index output bit-exact.
#include "post_common.h"

namespace {

typedef unsigned long long u64;

__global__ __launch_bounds__(1024) void od_nms_sort(const float* __restrict__ boxes, const u64* __restrict__ keys,
                                                    const int* __restrict__ counts, int P, int NC, int K, int KP,
                                                    u64* __restrict__ skeys, f32x4* __restrict__ sbox,
                                                    int* __restrict__ scls) {
  __shared__ u64 s[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < KP) s[tid] = tid < K ? keys[(long long)b * K + tid] : 0ull;
  __syncthreads();
  for (int k = 2; k <= KP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int ixj = tid ^ j;
      if (tid < KP && ixj > tid) {
        const u64 a = s[tid], c = s[ixj];
        const bool desc = (tid & k) == 0;
        if (desc ? (a < c) : (a > c)) {
          s[tid] = c;
          s[ixj] = a;
        }
      }
      __syncthreads();
    }
  }
  if (tid < KP) {
    const u64 key = s[tid];
    skeys[(long long)b * KP + tid] = key;
    const int n = counts[b];
    if (tid < n) {
      const unsigned flat = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      const unsigned p = flat / (unsigned)NC;
      const unsigned c = flat - p * (unsigned)NC;
      sbox[(long long)b * KP + tid] = *(const f32x4*)(boxes + ((long long)b * P + p) * 4);
      scls[(long long)b * KP + tid] = (int)c;
    }
  }
}

__device__ __forceinline__ bool suppresses(const f32x4 a, float area_a, const f32x4 c, float thr) {
  const float ix1 = fmaxf(a[0], c[0]), iy1 = fmaxf(a[1], c[1]);
  const float ix2 = fminf(a[2], c[2]), iy2 = fminf(a[3], c[3]);
  const float iw = fmaxf(ix2 - ix1, 0.f), ih = fmaxf(iy2 - iy1, 0.f);
  const float inter = iw * ih;
  const float area_c = (c[2] - c[0]) * (c[3] - c[1]);
  const float uni = (area_a + area_c) - inter;
  return inter > thr * uni;
}

__global__ __launch_bounds__(1024) void od_nms_mask(const f32x4* __restrict__ sbox, const int* __restrict__ scls,
                                                   const int* __restrict__ counts, int KP, int W, float thr, int strict,
                                                   u64* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  f32x4* lb = (f32x4*)sm;                // [KP]
  int* lc = (int*)(sm + (size_t)KP * 16);  // [KP]
  const int b = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = counts[b];
  if (rb * 64 >= n) {  // no valid rows here: the scan never reads these words with a live bit, but keep them defined
    for (int w = rb + wv; w < W; w += 16) mask[((long long)b * KP + rb * 64 + lane) * W + w] = 0ull;
    return;
  }
  for (int i = tid; i < n; i += 1024) {
    lb[i] = sbox[(long long)b * KP + i];
    lc[i] = scls[(long long)b * KP + i];
  }
  __syncthreads();
  const int i = rb * 64 + lane;
  const bool rowok = i < n;
  const f32x4 a = rowok ? lb[i] : f32x4{0.f, 0.f, 0.f, 0.f};
  const int ca = rowok ? lc[i] : -1;
  const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
  for (int w = rb + wv; w < W; w += 16) {
    u64 bits = 0ull;
    const int jend = min(64, n - w * 64);
    for (int jj = 0; jj < jend; ++jj) {
      const int j = w * 64 + jj;  // wave-uniform -> LDS broadcast reads
      const f32x4 c = lb[j];
      const int cc = lc[j];
      const bool hit = rowok && j > i && (strict || cc == ca) && suppresses(a, area_a, c, thr);
      bits |= (u64)hit << jj;
    }
    mask[((long long)b * KP + i) * W + w] = bits;
  }
}

__device__ __forceinline__ u64 readlane64(u64 v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void od_nms_scan(const u64* __restrict__ mask, const u64* __restrict__ skeys,
                                                   const int* __restrict__ counts, int KP, int W, int max_det,
                                                   int* __restrict__ keep_flat, int* __restrict__ keep_count) {
  extern __shared__ __attribute__((aligned(16))) u64 lm[];  // [n][W] mask rows, then 16 kept words
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int n = __builtin_amdgcn_readfirstlane(counts[b]);
  const u64* mb = mask + (long long)b * KP * W;
  for (int i = tid; i < n * W; i += 256) lm[i] = mb[i];
  __syncthreads();
  if (tid >= 64) return;  // the greedy scan itself is one wavefront (no further workgroup barrier below)
  u64* keptw = lm + (size_t)KP * W;

  const int w = lane & 15, g = lane >> 4;
  u64 removed = 0ull;  // lane (g, w): running suppression word w (replicated over g)
#pragma unroll 1
  for (int blk = 0; blk < W; ++blk) {
    u64 alive = 0ull;
    if (blk * 64 < n) {
      const int nv = n - blk * 64;
      const u64 valid = nv >= 64 ? ~0ull : ((1ull << nv) - 1ull);
      alive = ~readlane64(removed, blk) & valid;
      const u64 dg = (lane < nv) ? lm[(size_t)(blk * 64 + lane) * W + blk] : 0ull;
      // resolve the 64x64 diagonal block: only rows that suppress something need a step (scalar loop on SGPRs)
      u64 todo = __ballot(dg != 0ull) & alive;
      while (todo) {
        const int r = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        if ((alive >> r) & 1ull) {
          alive &= ~readlane64(dg, r);
          todo &= alive;
        }
      }
      // kept rows of this block suppress later columns: OR their mask rows (16 LDS reads per lane)
      u64 acc = 0ull;
      const bool wok = w < W && w > blk;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int r = t * 4 + g;
        const bool on = wok && ((alive >> r) & 1ull);
        const u64 v = on ? lm[(size_t)(blk * 64 + r) * W + w] : 0ull;
        acc |= v;
      }
      acc |= __shfl_xor(acc, 16);
      acc |= __shfl_xor(acc, 32);
      removed |= acc;
    }
    if (lane == 0) keptw[blk] = alive;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): keptw stores landed (single wave, program order)
  int base = 0;
  for (int blk = 0; blk < W; ++blk) {
    const u64 kw = keptw[blk];
    const int pos = base + __popcll(kw & ((1ull << lane) - 1ull));
    if (((kw >> lane) & 1ull) && pos < max_det) {
      const u64 key = skeys[(long long)b * KP + blk * 64 + lane];
      keep_flat[(long long)b * max_det + pos] = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
    }
    base += __popcll(kw);
  }
  const int total = min(base, max_det);
  for (int i = total + lane; i < max_det; i += 64) keep_flat[(long long)b * max_det + i] = -1;
  if (lane == 0) keep_count[b] = total;
}

int next_pow2(int v) {
  int p = 64;
  while (p < v) p <<= 1;
  return p;
}

struct NmsLayout {
  size_t skeys, sbox, scls, mask, total;
};
NmsLayout nms_layout(int B, int KP) {
  NmsLayout l;
  size_t o = 0;
  l.skeys = o;
  o += (size_t)B * KP * 8;
  l.sbox = o;
  o += (size_t)B * KP * 16;
  l.scls = o;
  o += (size_t)B * KP * 4;
  l.mask = o;
  o += (size_t)B * KP * (KP / 64) * 8;
  l.total = o;
  return l;
}

}  // namespace

void od_nms_sorted_buffers(void* nms_workspace, int B, int K, unsigned long long** skeys, f32x4** sbox, int** scls, int* KP) {
  const int kp = next_pow2(K);
  const NmsLayout l = nms_layout(B, kp);
  char* ws = (char*)nms_workspace;
  *skeys = (unsigned long long*)(ws + l.skeys);
  *sbox = (f32x4*)(ws + l.sbox);
  *scls = (int*)(ws + l.scls);
  *KP = kp;
}

int od_nms_mask_scan_launch(od_ctx* ctx, void* nms_workspace, const int32_t* counts, int B, int K, float iou_threshold,
                            int strict, int max_det, int32_t* keep_flat, int32_t* keep_count, hipStream_t s) {
  const int KP = next_pow2(K), W = KP / 64;
  const NmsLayout l = nms_layout(B, KP);
  char* ws = (char*)nms_workspace;
  u64* skeys = (u64*)(ws + l.skeys);
  f32x4* sbox = (f32x4*)(ws + l.sbox);
  int* scls = (int*)(ws + l.scls);
  u64* mask = (u64*)(ws + l.mask);
  hipLaunchKernelGGL(od_nms_mask, dim3(W, B), dim3(1024), (size_t)KP * 20, s, sbox, scls, counts, KP, W, iou_threshold,
                     strict, mask);
  OD_CHECK_LAUNCH();
  const size_t scan_lds = (size_t)KP * W * 8 + 16 * 8;
  if (int rc = od_ensure_lds(ctx, (const void*)&od_nms_scan, scan_lds)) return rc;
  hipLaunchKernelGGL(od_nms_scan, dim3(B), dim3(256), scan_lds, s, mask, skeys, counts, KP, W, max_det, keep_flat,
                     keep_count);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" size_t od_nms_workspace_bytes(int B, int K) {
  if (B <= 0 || K <= 0 || K > 1024) return 0;
  return nms_layout(B, next_pow2(K)).total;
}

extern "C" int od_nms(od_ctx* ctx, const float* boxes, const uint64_t* keys, const int32_t* counts, int B, int P, int NC,
                      int K, float iou_threshold, int strict, int max_det, int32_t* keep_flat, int32_t* keep_count,
                      void* workspace, size_t workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && boxes && keys && counts && keep_flat && keep_count && workspace, "od_nms: null argument");
  OD_REQUIRE(B > 0 && B <= 65535 && P > 0 && NC > 0 && K > 0 && K <= 1024 && max_det > 0,
             "od_nms: bad dims (K <= 1024)");
  const int KP = next_pow2(K);
  const NmsLayout l = nms_layout(B, KP);
  if (workspace_bytes < l.total) {
    od_set_error("od_nms: workspace %zu < %zu bytes", workspace_bytes, l.total);
    return OD_ERR_WORKSPACE;
  }
  char* ws = (char*)workspace;
  u64* skeys = (u64*)(ws + l.skeys);
  f32x4* sbox = (f32x4*)(ws + l.sbox);
  int* scls = (int*)(ws + l.scls);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(od_nms_sort, dim3(B), dim3(1024), 0, s, boxes, (const u64*)keys, counts, P, NC, K, KP, skeys, sbox,
                     scls);
  OD_CHECK_LAUNCH();
  return od_nms_mask_scan_launch(ctx, workspace, counts, B, K, iou_threshold, strict, max_det, keep_flat, keep_count, s);
}
