// K7: exact per-image top-K candidate selection over conf[B, N] (N = P*NC), HBM-bound integer/bit work.
//
// Order is total: key = (float_bits(conf) << 32) | (0xFFFFFFFF - flat_index), larger key = better, so a candidate
// with the same confidence but a LOWER flat index wins (oracle/nms.py: sort key (conf desc, flat asc)).  conf <=
// conf_threshold (and NaN) are not candidates.  Selection is a most-significant-digit radix select, never a sort of
// all N values and never approximate:
//   1. hist0     : full pass, 4096-bin histogram of score bits [30:19] per image (LDS atomics -> global atomics)
//   2. select0   : per image, the digit d0 holding the K-th largest; G0 = #candidates above it
//   3. partition : full pass; digit > d0 -> straight to the output, digit == d0 -> candidate list (flat index only)
//   4. refine    : one workgroup per image radix-selects the remaining 19 score bits + index bits inside the
//                  candidate list (LDS histograms; early exit as soon as a bin is taken whole) and appends the winners
// Output keys are an unordered SET of min(K, #candidates) keys per image (K8 sorts them); unused slots are 0.
#include "post_common.h"

namespace {

constexpr int NB = OD_TOPK_NB;
#define score_bits od_score_bits
#define find_digit od_find_digit

__global__ __launch_bounds__(256) void od_topk_hist0(const float* __restrict__ conf, int N, float thr,
                                                     int* __restrict__ hist, int chunk) {
  __shared__ int lh[NB];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < NB; i += 256) lh[i] = 0;
  __syncthreads();
  const float* src = conf + (long long)b * N;
  const int beg = blockIdx.x * chunk;
  const int end = min(beg + chunk, N);
  for (int i = beg + threadIdx.x * 4; i < end; i += 256 * 4) {
    if (i + 3 < end) {
      const f32x4 v = *(const f32x4*)(src + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned sb = score_bits(v[e], thr);
        if (sb) atomicAdd(&lh[(sb >> 19) & (NB - 1)], 1);
      }
    } else {
      for (int e = i; e < end; ++e) {
        const unsigned sb = score_bits(src[e], thr);
        if (sb) atomicAdd(&lh[(sb >> 19) & (NB - 1)], 1);
      }
    }
  }
  __syncthreads();
  int* gh = hist + (long long)b * NB;
  for (int i = threadIdx.x; i < NB; i += 256)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

__global__ __launch_bounds__(64) void od_topk_select0(const int* __restrict__ hist, TopkState* __restrict__ st, int K) {
  const int b = blockIdx.x;
  int above, in_bin;
  const int d = find_digit(hist + (long long)b * NB, NB, K, &above, &in_bin);
  if (threadIdx.x == 0) {
    TopkState s;
    s.d0 = d;  // -1: fewer than K candidates in total
    s.krem = d < 0 ? 0 : K - above;
    s.nout = 0;
    s.ncand = 0;
    st[b] = s;
  }
}

// Block-local compaction: winners / candidates are first collected in LDS (LDS atomics), then the workgroup reserves
// its output range with ONE global atomic per list -- a global atomic per element serialises on 32 addresses.
__global__ __launch_bounds__(256) void od_topk_partition(const float* __restrict__ conf, int N, float thr,
                                                         TopkState* __restrict__ st, unsigned long long* __restrict__ keys,
                                                         unsigned* __restrict__ cand, int K, int chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned sm_u[];
  unsigned* l_out = sm_u;        // [K] flat indices going straight to the output (fewer than K elements of the whole image
                                 //     lie above the d0 bin, or exist at all when d0 == -1)
  unsigned* l_cand = sm_u + K;   // [chunk] flat indices of the d0 bin (all of them, if every score of the chunk is equal)
  __shared__ int n_out, n_cand, base_out, base_cand;
  const int b = blockIdx.y;
  const int d0 = st[b].d0;
  const float* src = conf + (long long)b * N;
  if (threadIdx.x == 0) {
    n_out = 0;
    n_cand = 0;
  }
  __syncthreads();
  const int beg = blockIdx.x * chunk;
  const int end = min(beg + chunk, N);
  // 16-byte loads, four per thread in flight (N and chunk are multiples of 4; one 4-byte load per trip ran this pass at
  // 1.4 TB/s -- 32 us of the 0.2 ms post-processing chain at 32 x 320^2 -- against 4.4 TB/s for the histogram pass)
  for (int i0 = beg + threadIdx.x * 4; i0 < end; i0 += 256 * 4 * 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024;
      v[u] = i < end ? *(const f32x4*)(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 1024;
      if (i >= end) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned sb = score_bits(v[u][e], thr);
        if (!sb) continue;
        const int dg = (int)((sb >> 19) & (NB - 1));
        if (dg > d0) {  // d0 == -1: everything
          l_out[atomicAdd(&n_out, 1)] = (unsigned)(i + e);
        } else if (dg == d0) {
          l_cand[atomicAdd(&n_cand, 1)] = (unsigned)(i + e);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    base_out = n_out ? atomicAdd(&st[b].nout, n_out) : 0;
    base_cand = n_cand ? atomicAdd(&st[b].ncand, n_cand) : 0;
  }
  __syncthreads();
  unsigned long long* ok = keys + (long long)b * K + base_out;
  for (int j = threadIdx.x; j < n_out; j += 256) {
    const unsigned f = l_out[j];
    ok[j] = ((unsigned long long)__float_as_uint(src[f]) << 32) | (unsigned long long)(0xFFFFFFFFu - f);
  }
  unsigned* oc = cand + (long long)b * N + base_cand;
  for (int j = threadIdx.x; j < n_cand; j += 256) oc[j] = l_cand[j];
}

// Refine inside the d0 bin.  Remaining key bits, most significant first: score[18:8], score[7:0], ~flat[31:21],
// ~flat[20:10], ~flat[9:0].
__global__ __launch_bounds__(1024) void od_topk_refine(const float* __restrict__ conf, int N, TopkState* __restrict__ st,
                                                       unsigned long long* __restrict__ keys,
                                                       const unsigned* __restrict__ cand, int K) {
  __shared__ int lh[NB];
  __shared__ int sh_digit, sh_above, sh_inbin;
  const int b = blockIdx.x;
  const TopkState s = st[b];
  if (s.d0 < 0 || s.krem <= 0) return;
  const float* src = conf + (long long)b * N;
  const unsigned* ic = cand + (long long)b * N;
  const int nc = s.ncand;
  int krem = s.krem;
  // 51 low key bits below the first digit: 19 score bits + 32 index bits
  unsigned long long prefix = 0, pmask = 0;  // over the 51-bit sub-key  (score[18:0] << 32 | ~flat)
  const int shifts[5] = {40, 32, 21, 10, 0};
  const int widths[5] = {11, 8, 11, 11, 10};
  bool whole = (nc == krem);  // take the whole bin
  for (int ps = 0; ps < 5 && !whole; ++ps) {
    const int sh = shifts[ps], nbins = 1 << widths[ps];
    for (int i = threadIdx.x; i < NB; i += 1024) lh[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nc; i += 1024) {
      const unsigned f = ic[i];
      const unsigned sb = __float_as_uint(src[f]);
      const unsigned long long sub = ((unsigned long long)(sb & 0x7FFFFu) << 32) | (unsigned long long)(0xFFFFFFFFu - f);
      if ((sub & pmask) == prefix) atomicAdd(&lh[(int)((sub >> sh) & (unsigned long long)(nbins - 1))], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      int above, in_bin;
      const int d = find_digit(lh, NB, krem, &above, &in_bin);  // bins >= nbins are empty
      if (threadIdx.x == 0) {
        sh_digit = d;
        sh_above = above;
        sh_inbin = in_bin;
      }
    }
    __syncthreads();
    prefix |= (unsigned long long)sh_digit << sh;
    pmask |= (unsigned long long)(nbins - 1) << sh;
    krem -= sh_above;
    whole = (sh_inbin == krem);
    __syncthreads();
  }
  // winners: sub-key > prefix on the masked bits, or == prefix (then the whole remaining bin is taken)
  unsigned long long* ok = keys + (long long)b * K;
  for (int i = threadIdx.x; i < nc; i += 1024) {
    const unsigned f = ic[i];
    const unsigned sb = __float_as_uint(src[f]);
    const unsigned long long sub = ((unsigned long long)(sb & 0x7FFFFu) << 32) | (unsigned long long)(0xFFFFFFFFu - f);
    if ((sub & pmask) >= prefix) {
      const int slot = atomicAdd(&st[b].nout, 1);
      if (slot < K) ok[slot] = ((unsigned long long)sb << 32) | (unsigned long long)(0xFFFFFFFFu - f);
    }
  }
}

__global__ void od_topk_counts(const TopkState* __restrict__ st, int* __restrict__ counts, int B, int K) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) counts[b] = min(st[b].nout, K);
}

struct Layout {
  size_t hist_off, state_off, cand_off, total;
};
Layout ws_layout(int B, int N) {
  Layout l;
  l.hist_off = 0;
  size_t o = (size_t)B * NB * sizeof(int);
  l.state_off = o;
  o += ((size_t)B * sizeof(TopkState) + 255) & ~(size_t)255;
  l.cand_off = o;
  o += (size_t)B * N * sizeof(unsigned);
  l.total = o;
  return l;
}

}  // namespace

extern "C" size_t od_topk_workspace_bytes(int B, int N, int K) {
  (void)K;
  if (B <= 0 || N <= 0) return 0;
  return ws_layout(B, N).total;
}

extern "C" int od_topk_scores(od_ctx* ctx, const float* conf, int B, int N, int K, float conf_threshold,
                              uint64_t* keys, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && conf && keys && counts && workspace, "od_topk_scores: null argument");
  OD_REQUIRE(B > 0 && B <= 65535 && N > 0 && K > 0, "od_topk_scores: bad dims");
  OD_REQUIRE(conf_threshold >= 0.f, "od_topk_scores: conf_threshold must be >= 0 (scores are probabilities)");
  OD_REQUIRE(N % 4 == 0, "od_topk_scores: N must be a multiple of 4");
  const Layout l = ws_layout(B, N);
  if (workspace_bytes < l.total) {
    od_set_error("od_topk_scores: workspace %zu < %zu bytes", workspace_bytes, l.total);
    return OD_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int* hist = (int*)(ws + l.hist_off);
  TopkState* st = (TopkState*)(ws + l.state_off);
  unsigned* cand = (unsigned*)(ws + l.cand_off);
  OD_CHECK_HIP(hipMemsetAsync(ws, 0, l.cand_off, s));  // histograms + state
  OD_CHECK_HIP(hipMemsetAsync(keys, 0, (size_t)B * K * sizeof(uint64_t), s));
  // ~8 workgroups per CU in total; chunk is a multiple of 1024 elements so vector loads stay aligned
  int chunks = od_ceil_div(2048, B);
  int chunk = od_round_up(od_ceil_div(N, chunks), 1024);
  if (chunk > 8192) chunk = 8192;  // partition stages 2 x chunk u32 in LDS (<= 64 KiB)
  chunks = od_ceil_div(N, chunk);
  hipLaunchKernelGGL(od_topk_hist0, dim3(chunks, B), dim3(256), 0, s, conf, N, conf_threshold, hist, chunk);
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_topk_select0, dim3(B), dim3(64), 0, s, hist, st, K);
  OD_CHECK_LAUNCH();
  // its own, smaller chunk: K + chunk u32 of LDS (20 KiB at K = 1024) -> eight workgroups per CU
  const int pchunk = chunk > 4096 ? 4096 : chunk;
  const int pchunks = od_ceil_div(N, pchunk);
  const size_t plds = ((size_t)K + pchunk) * sizeof(unsigned);
  OD_REQUIRE(plds <= 160 * 1024, "od_topk_scores: K too large for the partition pass (K + 4096 u32 of LDS)");
  if (int rc = od_ensure_lds(ctx, (const void*)&od_topk_partition, plds)) return rc;
  hipLaunchKernelGGL(od_topk_partition, dim3(pchunks, B), dim3(256), plds, s, conf, N, conf_threshold, st,
                     (unsigned long long*)keys, cand, K, pchunk);
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_topk_refine, dim3(B), dim3(1024), 0, s, conf, N, st, (unsigned long long*)keys, cand, K);
  OD_CHECK_LAUNCH();
  hipLaunchKernelGGL(od_topk_counts, dim3(od_ceil_div(B, 64)), dim3(64), 0, s, st, counts, B, K);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
