// The product path of K5-K8 as ONE call (od_detect): pred -> boxes + exact top-K + NMS kept indices in five launches, without
// materialising the confidence tensor.
//
// Round 2 ran od_head_postprocess (writes conf f32 [B,P,NC]: 43 MB at 32 x 320^2), od_topk_scores (two full passes over conf:
// histogram, partition) and od_nms -- 9 kernels + 2 memsets and 194 MB of HBM traffic per batch.  Here:
//   pass 1  (od_detect_pass1)  reads pred once: confidences in LDS, decoded boxes out, the 4096-bin first-digit histogram of
//           all scores (LDS atomics -> global), and ONE float per prior: its largest confidence (rowmax)
//   pass 2  (od_detect_pass2)  finds the first digit d0 of the K-th key from the histogram, then reads rowmax and recomputes
//           (same code, bit-identical) the confidences of ONLY the priors whose best score reaches the d0 bin -- a fraction of
//           a percent of them -- and partitions those into winners (digit > d0) and the d0-bin candidate list, as 64-bit keys
//   refine  (od_detect_refine_sort)  one workgroup per image: radix-select inside the d0 bin on the remaining 51 key bits,
//           bitonic sort of the K winners in LDS, gather of their boxes / classes for the NMS, counts; re-zeroes the histogram
//   od_nms_mask, od_nms_scan  as in od_nms (nms.hip)
// Same total order, same exact selection, same kept indices as the three-call path (tests compare them bit for bit).
// Compiled with -ffp-contract=off like post.hip / topk.hip / nms.hip.
#include <string.h>

#include "post_common.h"

namespace {

typedef unsigned long long u64;
constexpr int NB = OD_TOPK_NB;
constexpr int DT_ROWS = 256;      // priors per workgroup (one thread each)
constexpr int DT_CAND_CAP = 1024; // d0-bin candidates a workgroup compacts in LDS before falling back to global atomics

// First radix digit.  Confidences are products of two probabilities, so a candidate's bit pattern lies in (bits(thr),
// bits(1.0)]: the 4096 bins are spread over THAT range (dbase = bits(thr), dshift = the smallest shift that fits it) instead of
// over all exponents -- at thr = 0.01 a bin is 2^14 ulps (0.2 % of the value) wide instead of 2^19 (4.4 %): the threshold bin
// holds 22x fewer scores, and far fewer priors have to be looked at again in pass 2.
__device__ __forceinline__ int od_digit0(unsigned sb, unsigned dbase, int dshift) { return (int)((sb - dbase) >> dshift); }

// grid (ceil(P / 256), B).  LDS: rows [256][C] f32 (confidences are computed in place over the class logits) + hist[4096].
__global__ __launch_bounds__(256) void od_detect_pass1(const float* __restrict__ pred, const float* __restrict__ priors,
                                                       float* __restrict__ boxes, float* __restrict__ rowmax,
                                                       float* __restrict__ conf_out, int* __restrict__ hist,
                                                       TopkState* __restrict__ st, int P, int NC, float loc_scale, int clip,
                                                       float thr, unsigned dbase, int dshift) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = NC + 6, tid = threadIdx.x, b = blockIdx.y;
  float* rows = sm;                        // [DT_ROWS][C]
  int* lh = (int*)(sm + DT_ROWS * C);      // [NB]
  const int p0 = blockIdx.x * DT_ROWS;
  const int nrows = min(DT_ROWS, P - p0);
  for (int i = tid; i < NB; i += 256) lh[i] = 0;
  if (blockIdx.x == 0 && tid == 0) {  // pass 2 counts into these with atomics
    TopkState z = {0, 0, 0, 0};
    st[b] = z;
  }
  const long long r0 = (long long)b * P + p0;
  const float* src = pred + r0 * C;  // 16-byte aligned: P is even and p0 a multiple of 256
  const int nel = nrows * C;
  for (int i = tid * 4; i < nel; i += 256 * 4) {
    if (i + 3 < nel) {
      *(f32x4*)(rows + i) = *(const f32x4*)(src + i);
    } else {
      for (int e = i; e < nel; ++e) rows[e] = src[e];
    }
  }
  __syncthreads();
  if (tid < nrows) {
    float* row = rows + tid * C;
    const f32x4 loc = {row[2 + NC], row[3 + NC], row[4 + NC], row[5 + NC]};
    od_row_conf(row, NC, row + 2);
    float mx = 0.f;
    for (int c = 0; c < NC; ++c) {
      const float v = row[2 + c];
      mx = fmaxf(mx, v);
      const unsigned sb = od_score_bits(v, thr);
      if (sb) atomicAdd(&lh[od_digit0(sb, dbase, dshift)], 1);
    }
    rowmax[r0 + tid] = mx;
    const f32x4 pr = *(const f32x4*)(priors + (long long)(p0 + tid) * 4);
    *(f32x4*)(boxes + (r0 + tid) * 4) = od_decode_one(loc, pr, loc_scale, clip);
  }
  __syncthreads();
  int* gh = hist + (long long)b * NB;
  for (int i = tid; i < NB; i += 256)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
  if (conf_out) {  // optional dense confidences (API parity with od_head_postprocess; the product path passes NULL)
    float* dst = conf_out + r0 * NC;
    for (int i = tid; i < nrows * NC; i += 256) {
      const int r = i / NC, c = i - r * NC;
      dst[i] = rows[r * C + 2 + c];
    }
  }
}

// Block-wide (256 threads) search of a 4096-bin GLOBAL histogram for the bin where the count of elements in higher bins first
// reaches >= krem: every thread owns 16 consecutive bins in registers (one round of loads), a suffix scan over the 256
// partial sums (wave shuffles + four partials through LDS) finds the owner, the owner walks its 16 bins.
__device__ __forceinline__ void od_find_digit_256(const int* __restrict__ gh, int krem, int* sh /* [8] LDS */, int* d_out,
                                                  int* above_out) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int v[16];
  int tot = 0;
  const int4* g4 = (const int4*)(gh + tid * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int4 t = g4[q];
    v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
    tot += t.x + t.y + t.z + t.w;
  }
  int suf = tot;  // inclusive suffix sum over the lanes of this wave
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_down(suf, off);
    if (lane + off < 64) suf += o;
  }
  if (lane == 0) sh[wv] = suf;  // the wave's total
  if (tid == 0) {
    sh[4] = -1;
    sh[5] = 0;
  }
  __syncthreads();
  int higher_waves = 0;
  for (int w = wv + 1; w < 4; ++w) higher_waves += sh[w];
  const int incl = suf + higher_waves, higher = incl - tot;  // elements in the bins of this thread and above / strictly above
  if (higher < krem && incl >= krem) {  // exactly one thread
    int run = higher;
#pragma unroll
    for (int q = 15; q >= 0; --q) {
      if (run + v[q] >= krem) {
        sh[4] = tid * 16 + q;
        sh[5] = run;
        break;
      }
      run += v[q];
    }
  }
  __syncthreads();
  *d_out = sh[4];
  *above_out = sh[5];
}

constexpr int DT2_RPT = 4;  // priors per thread in pass 2 (1024 per workgroup)

// grid (ceil(P / 1024), B).  LDS: per-thread row scratch [256][C] + l_out [K] + l_cand [DT_CAND_CAP] keys.
__global__ __launch_bounds__(256) void od_detect_pass2(const float* __restrict__ pred, const float* __restrict__ rowmax,
                                                       const int* __restrict__ hist, TopkState* __restrict__ st,
                                                       u64* __restrict__ keys, u64* __restrict__ cand, int P, int NC, int K,
                                                       float thr, long long cand_stride, unsigned dbase, int dshift) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = NC + 6, tid = threadIdx.x, b = blockIdx.y;
  float* rows = sm;                          // [DT_ROWS][C]
  u64* l_out = (u64*)(sm + DT_ROWS * C);     // [K]   (DT_ROWS * C * 4 bytes is a multiple of 8)
  u64* l_cand = l_out + K;                   // [DT_CAND_CAP]
  __shared__ int n_out, n_cand, n_hot, base_out, base_cand, sh_fd[8];
  __shared__ int hot_list[DT_ROWS * DT2_RPT];
  if (tid == 0) {
    n_out = 0;
    n_cand = 0;
    n_hot = 0;
  }
  int d0, above;
  od_find_digit_256(hist + (long long)b * NB, K, sh_fd, &d0, &above);  // (its barriers also publish n_out / n_cand)
  if (blockIdx.x == 0 && tid == 0) {
    st[b].d0 = d0;  // -1: fewer than K candidates in the whole image -> every candidate is a winner
    st[b].krem = d0 < 0 ? 0 : K - above;
  }
  const int p_base = blockIdx.x * (DT_ROWS * DT2_RPT);
  // the priors whose best score reaches the d0 bin (a few per cent at most) are first COMPACTED into an LDS list and then
  // taken one per thread: walking them where they sit ran every wave through the row code at a few per cent lane occupancy
  float mxv[DT2_RPT];
#pragma unroll
  for (int u = 0; u < DT2_RPT; ++u) {
    const int p = p_base + u * DT_ROWS + tid;
    mxv[u] = p < P ? rowmax[(long long)b * P + p] : 0.f;
  }
#pragma unroll
  for (int u = 0; u < DT2_RPT; ++u) {
    const unsigned sb = od_score_bits(mxv[u], thr);
    if (sb && od_digit0(sb, dbase, dshift) >= d0) hot_list[atomicAdd(&n_hot, 1)] = p_base + u * DT_ROWS + tid;
  }
  __syncthreads();
  const int nh = n_hot;
  for (int e = tid; e < nh; e += 256) {
    const int p = hot_list[e];
    float* row = rows + tid * C;
    const float* src = pred + ((long long)b * P + p) * C;
    for (int c = 0; c < 2 + NC; ++c) row[c] = src[c];
    od_row_conf(row, NC, row + 2);  // the same code as pass 1: bit-identical confidences
    for (int c = 0; c < NC; ++c) {
      const unsigned sbc = od_score_bits(row[2 + c], thr);
      if (!sbc) continue;
      const int dg = od_digit0(sbc, dbase, dshift);
      if (dg < d0) continue;
      const unsigned flat = (unsigned)(p * NC + c);
      const u64 key = ((u64)sbc << 32) | (u64)(0xFFFFFFFFu - flat);
      if (dg > d0) {  // fewer than K of these in the whole image
        l_out[atomicAdd(&n_out, 1)] = key;
      } else {
        const int slot = atomicAdd(&n_cand, 1);
        if (slot < DT_CAND_CAP) {
          l_cand[slot] = key;
        } else {  // a degenerate image (e.g. all scores equal): straight to the global list
          cand[(long long)b * cand_stride + atomicAdd(&st[b].ncand, 1)] = key;
        }
      }
    }
  }
  __syncthreads();
  const int nc = min(n_cand, DT_CAND_CAP);
  if (tid == 0) {
    base_out = n_out ? atomicAdd(&st[b].nout, n_out) : 0;
    base_cand = nc ? atomicAdd(&st[b].ncand, nc) : 0;
  }
  __syncthreads();
  u64* ok = keys + (long long)b * K + base_out;
  for (int j = tid; j < n_out; j += 256) ok[j] = l_out[j];
  u64* oc = cand + (long long)b * cand_stride + base_cand;
  for (int j = tid; j < nc; j += 256) oc[j] = l_cand[j];
}

// One workgroup (1024 threads) per image: refine inside the d0 bin on the sub-key (low dshift bits of score - dbase) << 32 |
// ~flat (<= 51 bits, digits of 11, 8, 11, 11, 10 bits from the top), then sort + gather for the NMS.
__global__ __launch_bounds__(1024) void od_detect_refine_sort(const float* __restrict__ boxes, TopkState* __restrict__ st,
                                                              u64* __restrict__ keys, const u64* __restrict__ cand,
                                                              int* __restrict__ hist, int* __restrict__ counts, int P, int NC,
                                                              int K, int KP, long long cand_stride, u64* __restrict__ skeys,
                                                              f32x4* __restrict__ sbox, int* __restrict__ scls, unsigned dbase,
                                                              int dshift) {
  __shared__ int lh[NB];
  __shared__ u64 s[1024];
  __shared__ int sh_digit, sh_above, sh_inbin, n_win;
  const int b = blockIdx.x, tid = threadIdx.x;
  const TopkState t = st[b];
  const int nout0 = min(t.nout, K);
  if (tid == 0) n_win = 0;
  // winners that pass 2 wrote straight to the output; the rest of the sort array is empty (key 0 sorts last)
  s[tid] = tid < nout0 ? keys[(long long)b * K + tid] : 0ull;
  __syncthreads();
  if (t.d0 >= 0 && t.krem > 0) {
    const u64* ic = cand + (long long)b * cand_stride;
    const int nc = t.ncand;
    int krem = t.krem;
    const unsigned low_mask = (1u << dshift) - 1u;  // score bits below the first digit
    u64 prefix = 0, pmask = 0;
    const int shifts[5] = {40, 32, 21, 10, 0};
    const int widths[5] = {11, 8, 11, 11, 10};
    bool whole = (nc == krem);  // take the whole bin
    for (int ps = 0; ps < 5 && !whole; ++ps) {
      const int sh = shifts[ps], nbins = 1 << widths[ps];
      for (int i = tid; i < NB; i += 1024) lh[i] = 0;
      __syncthreads();
      for (int i = tid; i < nc; i += 1024) {
        const u64 key = ic[i];
        const u64 sub = ((u64)(((unsigned)(key >> 32) - dbase) & low_mask) << 32) | (key & 0xFFFFFFFFull);
        if ((sub & pmask) == prefix) atomicAdd(&lh[(int)((sub >> sh) & (u64)(nbins - 1))], 1);
      }
      __syncthreads();
      if (tid < 64) {
        int above, in_bin;
        const int d = od_find_digit(lh, NB, krem, &above, &in_bin);  // bins >= nbins are empty
        if (tid == 0) {
          sh_digit = d;
          sh_above = above;
          sh_inbin = in_bin;
        }
      }
      __syncthreads();
      prefix |= (u64)sh_digit << sh;
      pmask |= (u64)(nbins - 1) << sh;
      krem -= sh_above;
      whole = (sh_inbin == krem);
      __syncthreads();
    }
    // winners: sub-key > prefix on the masked bits, or == prefix (then the whole remaining bin is taken)
    for (int i = tid; i < nc; i += 1024) {
      const u64 key = ic[i];
      const u64 sub = ((u64)(((unsigned)(key >> 32) - dbase) & low_mask) << 32) | (key & 0xFFFFFFFFull);
      if ((sub & pmask) >= prefix) {
        const int slot = nout0 + atomicAdd(&n_win, 1);
        if (slot < K) s[slot] = key;
      }
    }
    __syncthreads();
  }
  const int n = min(nout0 + n_win, K);
  // bitonic sort, descending = (conf desc, flat asc)
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
    if (tid < K) keys[(long long)b * K + tid] = key;  // the API's key set: sorted here, unused slots 0
    if (tid < n) {
      const unsigned flat = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      const unsigned p = flat / (unsigned)NC;
      const unsigned c = flat - p * (unsigned)NC;
      sbox[(long long)b * KP + tid] = *(const f32x4*)(boxes + ((long long)b * P + p) * 4);
      scls[(long long)b * KP + tid] = (int)c;
    }
  }
  if (tid == 0) counts[b] = n;
  int* gh = hist + (long long)b * NB;  // leave the histogram zeroed for the next call
  for (int i = tid; i < NB; i += 1024) gh[i] = 0;
}

// one workgroup per image: row r of out[b] = {flat index (int bits), conf, x1, y1, x2, y2} of kept detection r, the
// confidence recomputed from pred by the same code as pass 1
__global__ __launch_bounds__(256) void od_gather_det_pred(const float* __restrict__ pred, const float* __restrict__ boxes,
                                                          const int32_t* __restrict__ keep_flat,
                                                          const int32_t* __restrict__ keep_count, int P, int NC, int max_det,
                                                          float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, C = NC + 6;
  const int n = keep_count[b];
  float* o = out + (size_t)b * (1 + 6 * (size_t)max_det);
  if (threadIdx.x == 0) o[0] = __int_as_float(n);
  float* row = sm + threadIdx.x * C;
  for (int r = threadIdx.x; r < max_det; r += 256) {
    float* rec = o + 1 + 6 * (size_t)r;
    if (r < n) {
      const int flat = keep_flat[(size_t)b * max_det + r];
      const int p = flat / NC, c = flat - p * NC;
      const float* src = pred + ((size_t)b * P + p) * C;
      for (int e = 0; e < 2 + NC; ++e) row[e] = src[e];
      od_row_conf(row, NC, row + 2);
      const float* bx = boxes + ((size_t)b * P + p) * 4;
      rec[0] = __int_as_float(flat);
      rec[1] = row[2 + c];
      rec[2] = bx[0];
      rec[3] = bx[1];
      rec[4] = bx[2];
      rec[5] = bx[3];
    } else {
      rec[0] = __int_as_float(-1);
      rec[1] = rec[2] = rec[3] = rec[4] = rec[5] = 0.f;
    }
  }
}

struct DetLayout {
  size_t hist, state, rowmax, cand, total;
  long long cand_stride;
};
DetLayout det_layout(int B, int P, int NC) {
  DetLayout l;
  size_t o = 0;
  l.hist = o;
  o += (size_t)B * NB * sizeof(int);
  l.state = o;
  o += ((size_t)B * sizeof(TopkState) + 255) & ~(size_t)255;
  l.rowmax = o;
  o += (((size_t)B * P * sizeof(float)) + 255) & ~(size_t)255;
  l.cand = o;
  l.cand_stride = (long long)P * NC;  // worst case: every score of an image sits in the d0 bin
  o += (size_t)B * (size_t)l.cand_stride * sizeof(u64);
  l.total = o;
  return l;
}

}  // namespace

extern "C" size_t od_detect_workspace_bytes(int B, int P, int NC, int K) {
  (void)K;
  if (B <= 0 || P <= 0 || NC <= 0) return 0;
  return det_layout(B, P, NC).total;
}

extern "C" int od_detect_workspace_init(od_ctx* ctx, void* workspace, size_t workspace_bytes, int B, int P, int NC, void* stream) {
  OD_REQUIRE(ctx && workspace && B > 0 && P > 0 && NC > 0, "od_detect_workspace_init: bad argument");
  const DetLayout l = det_layout(B, P, NC);
  if (workspace_bytes < l.total) {
    od_set_error("od_detect_workspace_init: workspace %zu < %zu bytes", workspace_bytes, l.total);
    return OD_ERR_WORKSPACE;
  }
  OD_CHECK_HIP(hipMemsetAsync(workspace, 0, l.rowmax, (hipStream_t)stream));  // histograms + state
  return OD_OK;
}

extern "C" int od_detect(od_ctx* ctx, const float* pred, const float* priors, int B, int P, int NC, float loc_scale, int clip,
                         float conf_threshold, int K, float iou_threshold, int strict, int max_det, float* boxes, float* conf,
                         uint64_t* keys, int32_t* counts, int32_t* keep_flat, int32_t* keep_count, void* workspace,
                         size_t workspace_bytes, void* nms_workspace, size_t nms_workspace_bytes, void* stream) {
  OD_REQUIRE(ctx && pred && priors && boxes && keys && counts && keep_flat && keep_count && workspace && nms_workspace,
             "od_detect: null argument");
  OD_REQUIRE(B > 0 && B <= 65535 && P > 0 && P % 2 == 0 && NC > 0 && NC <= 76 && K > 0 && K <= 1024 && max_det > 0,
             "od_detect: bad dims (P even, NC <= 76, K <= 1024)");
  OD_REQUIRE((long long)P * NC < (1LL << 31), "od_detect: P * NC must fit 31 bits");
  OD_REQUIRE(conf_threshold >= 0.f, "od_detect: conf_threshold must be >= 0 (scores are probabilities)");
  const DetLayout l = det_layout(B, P, NC);
  if (workspace_bytes < l.total) {
    od_set_error("od_detect: workspace %zu < %zu bytes", workspace_bytes, l.total);
    return OD_ERR_WORKSPACE;
  }
  if (nms_workspace_bytes < od_nms_workspace_bytes(B, K)) {
    od_set_error("od_detect: NMS workspace %zu < %zu bytes", nms_workspace_bytes, od_nms_workspace_bytes(B, K));
    return OD_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  int* hist = (int*)(ws + l.hist);
  TopkState* st = (TopkState*)(ws + l.state);
  float* rowmax = (float*)(ws + l.rowmax);
  u64* cand = (u64*)(ws + l.cand);
  const int C = NC + 6;
  // first-digit mapping: 4096 bins over (bits(thr), bits(1.0)]
  unsigned dbase;
  memcpy(&dbase, &conf_threshold, 4);
  int dshift = 0;
  while (((0x3F800000u - dbase) >> dshift) >= (unsigned)NB) ++dshift;
  const dim3 grid((unsigned)od_ceil_div(P, DT_ROWS), (unsigned)B);
  const size_t lds1 = (size_t)DT_ROWS * C * 4 + (size_t)NB * 4;
  if (int rc = od_ensure_lds(ctx, (const void*)&od_detect_pass1, lds1)) return rc;
  hipLaunchKernelGGL(od_detect_pass1, grid, dim3(256), lds1, s, pred, priors, boxes, rowmax, conf, hist, st, P, NC, loc_scale,
                     clip, conf_threshold, dbase, dshift);
  OD_CHECK_LAUNCH();
  const size_t lds2 = (size_t)DT_ROWS * C * 4 + ((size_t)K + DT_CAND_CAP) * 8;
  if (int rc = od_ensure_lds(ctx, (const void*)&od_detect_pass2, lds2)) return rc;
  const dim3 grid2((unsigned)od_ceil_div(P, DT_ROWS * DT2_RPT), (unsigned)B);
  hipLaunchKernelGGL(od_detect_pass2, grid2, dim3(256), lds2, s, pred, rowmax, hist, st, (u64*)keys, cand, P, NC, K,
                     conf_threshold, l.cand_stride, dbase, dshift);
  OD_CHECK_LAUNCH();
  u64* skeys;
  f32x4* sbox;
  int* scls;
  int KP;
  od_nms_sorted_buffers(nms_workspace, B, K, &skeys, &sbox, &scls, &KP);
  hipLaunchKernelGGL(od_detect_refine_sort, dim3(B), dim3(1024), 0, s, boxes, st, (u64*)keys, cand, hist, counts, P, NC, K, KP,
                     l.cand_stride, skeys, sbox, scls, dbase, dshift);
  OD_CHECK_LAUNCH();
  return od_nms_mask_scan_launch(ctx, nms_workspace, counts, B, K, iou_threshold, strict, max_det, keep_flat, keep_count, s);
}

extern "C" int od_gather_detections_pred(od_ctx* ctx, const float* pred, const float* boxes, const int32_t* keep_flat,
                                         const int32_t* keep_count, int B, int P, int NC, int max_det, float* out, void* stream) {
  OD_REQUIRE(ctx && pred && boxes && keep_flat && keep_count && out, "od_gather_detections_pred: null argument");
  OD_REQUIRE(B > 0 && P > 0 && NC > 0 && NC <= 76 && max_det > 0, "od_gather_detections_pred: bad dims");
  hipLaunchKernelGGL(od_gather_det_pred, dim3(B), dim3(256), (size_t)256 * (NC + 6) * 4, (hipStream_t)stream, pred, boxes,
                     keep_flat, keep_count, P, NC, max_det, out);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
