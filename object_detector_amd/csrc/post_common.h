// Shared by post.hip (K5/K6), topk.hip (K7) and detect.hip (the fused product path): the confidence arithmetic and the
// radix-select helpers.  Every TU that includes this is compiled with -ffp-contract=off, so the same source is the same
// f32 op sequence everywhere (and the one of oracle/postprocess.py): confidences are bit-identical across kernels.
#pragma once
#include "common.h"

constexpr int OD_TOPK_NB = 4096;  // histogram bins: first radix digit = score bits [30:19]

struct TopkState {  // per image
  int d0;           // first digit of the K-th key; -1 = take every candidate (fewer than K exist)
  int krem;         // how many to take from the d0 bin
  int nout;         // output slots used
  int ncand;        // candidate-list length
};

// conf[c] = sigmoid(l1 - l0) * softmax(classes)[c]  (docs/MODEL.md:54-58), written over out[0..NC); `out` may alias
// row + 2 (in place over the class logits): every logit is read before its slot is written.
__device__ __forceinline__ void od_row_conf(const float* row, int NC, float* out) {
  const float obj = 1.f / (1.f + expf(row[0] - row[1]));
  float mx = row[2];
  for (int c = 1; c < NC; ++c) mx = fmaxf(mx, row[2 + c]);
  float s = 0.f;
  for (int c = 0; c < NC; ++c) {
    const float e = expf(row[2 + c] - mx);
    out[c] = e;
    s += e;
  }
  for (int c = 0; c < NC; ++c) out[c] = obj * (out[c] / s);
}

__device__ __forceinline__ unsigned od_score_bits(float v, float thr) {
  return v > thr ? __float_as_uint(v) : 0u;  // positive floats: bit pattern is monotone in value
}

// boxes = prior + (loc * loc_scale) * [pw, ph, pw, ph], optional clip to [0,1]   (od.pb.decode_locs,
// reference check_assign.py:27: zero offsets decode to the prior itself)
__device__ __forceinline__ f32x4 od_decode_one(f32x4 loc, f32x4 pr, float loc_scale, int clip) {
  const float pw = pr[2] - pr[0], ph = pr[3] - pr[1];
  f32x4 o;
  o[0] = pr[0] + (loc[0] * loc_scale) * pw;
  o[1] = pr[1] + (loc[1] * loc_scale) * ph;
  o[2] = pr[2] + (loc[2] * loc_scale) * pw;
  o[3] = pr[3] + (loc[3] * loc_scale) * ph;
  if (clip) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fminf(fmaxf(o[e], 0.f), 1.f);
  }
  return o;
}

// Wave-level search of an nbins-bin histogram (in LDS or global) for the bin where the count of elements in HIGHER
// bins first reaches >= krem.  Returns digit (uniform) and *above = #elements in bins above it.  One wave.
__device__ __forceinline__ int od_find_digit(const int* hist, int nbins, int krem, int* above, int* in_bin) {
  const int lane = threadIdx.x & 63;
  const int per = nbins / 64;  // bins per lane; lane l owns bins [l*per, (l+1)*per)
  int s = 0;
  for (int i = 0; i < per; ++i) s += hist[lane * per + i];
  // inclusive suffix sum over lanes: suf = sum over lanes >= lane
  int suf = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int o = __shfl_down(suf, off);
    if (lane + off < 64) suf += o;
  }
  const int higher = suf - s;  // elements in lanes above this one
  const bool mine = higher < krem && suf >= krem;
  const unsigned long long bal = __ballot(mine);
  int digit = -1, ab = 0, ib = 0;
  if (bal) {
    const int owner = __ffsll((long long)bal) - 1;
    if (lane == owner) {
      int run = higher;
      for (int i = per - 1; i >= 0; --i) {
        const int c = hist[lane * per + i];
        if (run + c >= krem) {
          digit = lane * per + i;
          ab = run;
          ib = c;
          break;
        }
        run += c;
      }
    }
    digit = __shfl(digit, owner);
    ab = __shfl(ab, owner);
    ib = __shfl(ib, owner);
  }
  *above = ab;
  *in_bin = ib;
  return digit;
}

// nms.hip: the suppression-mask and greedy-scan launches of od_nms on already sorted / gathered candidates
int od_nms_mask_scan_launch(od_ctx* ctx, void* nms_workspace, const int32_t* counts, int B, int K, float iou_threshold,
                            int strict, int max_det, int32_t* keep_flat, int32_t* keep_count, hipStream_t stream);
// nms.hip: where od_nms keeps the sorted keys / gathered boxes / classes inside its workspace (KP = next power of two >= K)
void od_nms_sorted_buffers(void* nms_workspace, int B, int K, unsigned long long** skeys, f32x4** sbox, int** scls, int* KP);
