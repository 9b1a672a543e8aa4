// K4 (standalone), K5, K6: FPN top-down add, head post-process (objectness x class confidence) and prior-box decode.
// All HBM-bound elementwise work; rows are staged through LDS so that every global access is a full-line access.
// Compiled with -ffp-contract=off: the decode arithmetic is the bit-exact f32 sequence of oracle/postprocess.py.
#include "post_common.h"

namespace {

#define decode_one od_decode_one

constexpr int PP_ROWS = 256;  // priors per workgroup

// pred [B*P, C] -> conf [B*P, NC], boxes [B*P, 4];  C = 2 + NC + 4
__global__ __launch_bounds__(256) void od_head_post(const float* __restrict__ pred, const float* __restrict__ priors,
                                                    float* __restrict__ conf, float* __restrict__ boxes,
                                                    long long rows, int P, int NC, float loc_scale, int clip) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = NC + 6;
  const int tid = threadIdx.x;
  const long long r0 = (long long)blockIdx.x * PP_ROWS;
  const int nrows = (int)((rows - r0) < PP_ROWS ? (rows - r0) : PP_ROWS);
  float* sin = sm;                  // [PP_ROWS][C]
  float* sout = sm + PP_ROWS * C;   // [PP_ROWS][NC]
  // coalesced load (r0*C*4 bytes is 16-B aligned because PP_ROWS*C*4 is)
  const int nvec = nrows * C;
  const float* src = pred + r0 * C;
  for (int i = tid * 4; i < nvec; i += 256 * 4) {
    if (i + 3 < nvec) {
      *(f32x4*)(sin + i) = *(const f32x4*)(src + i);
    } else {
      for (int e = i; e < nvec; ++e) sin[e] = src[e];
    }
  }
  __syncthreads();
  if (tid < nrows) {
    const float* row = sin + tid * C;
    od_row_conf(row, NC, sout + tid * NC);
    const long long r = r0 + tid;
    const int p = (int)(r % P);
    const f32x4 loc = {row[2 + NC], row[3 + NC], row[4 + NC], row[5 + NC]};
    const f32x4 pr = *(const f32x4*)(priors + (long long)p * 4);
    *(f32x4*)(boxes + r * 4) = decode_one(loc, pr, loc_scale, clip);
  }
  __syncthreads();
  const int nout = nrows * NC;
  float* dst = conf + r0 * NC;
  for (int i = tid * 4; i < nout; i += 256 * 4) {
    if (i + 3 < nout) {
      *(f32x4*)(dst + i) = *(const f32x4*)(sout + i);
    } else {
      for (int e = i; e < nout; ++e) dst[e] = sout[e];
    }
  }
}

__global__ __launch_bounds__(256) void od_decode(const float* __restrict__ locs, const float* __restrict__ priors,
                                                 float* __restrict__ boxes, long long rows, int P, float loc_scale,
                                                 int clip) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const int p = (int)(r % P);
  const f32x4 loc = *(const f32x4*)(locs + r * 4);
  const f32x4 pr = *(const f32x4*)(priors + (long long)p * 4);
  *(f32x4*)(boxes + r * 4) = decode_one(loc, pr, loc_scale, clip);
}

// out[b,y,x,c] = a[b,y,x,c] + up[b,y/2,x/2,c]; 8 channels (16 B) per thread
__global__ __launch_bounds__(256) void od_up2_add(const f16* __restrict__ a, const f16* __restrict__ up,
                                                  f16* __restrict__ out, long long nvec, int H, int W, int C8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C8);
    long long pix = i / C8;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H);
    const long long b = pix / H;
    const long long ui = ((b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C8 + c;
    const f16x8 va = *(const f16x8*)(a + i * 8);
    const f16x8 vu = *(const f16x8*)(up + ui * 8);
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (f16)((float)va[e] + (float)vu[e]);
    *(f16x8*)(out + i * 8) = o;
  }
}

// one workgroup per image: row r of out[b] = {flat index (int bits), conf, x1, y1, x2, y2} of kept detection r
__global__ __launch_bounds__(256) void od_gather_det(const float* __restrict__ conf, const float* __restrict__ boxes,
                                                     const int32_t* __restrict__ keep_flat,
                                                     const int32_t* __restrict__ keep_count, int P, int NC, int max_det,
                                                     float* __restrict__ out) {
  const int b = blockIdx.x;
  const int n = keep_count[b];
  float* o = out + (size_t)b * (1 + 6 * (size_t)max_det);
  if (threadIdx.x == 0) o[0] = __int_as_float(n);
  for (int r = threadIdx.x; r < max_det; r += 256) {
    float* row = o + 1 + 6 * (size_t)r;
    if (r < n) {
      const int flat = keep_flat[(size_t)b * max_det + r];
      const int p = flat / NC;
      const float* bx = boxes + ((size_t)b * P + p) * 4;
      row[0] = __int_as_float(flat);
      row[1] = conf[(size_t)b * P * NC + flat];
      row[2] = bx[0];
      row[3] = bx[1];
      row[4] = bx[2];
      row[5] = bx[3];
    } else {
      row[0] = __int_as_float(-1);
      row[1] = row[2] = row[3] = row[4] = row[5] = 0.f;
    }
  }
}

}  // namespace

extern "C" int od_gather_detections(od_ctx* ctx, const float* conf, const float* boxes, const int32_t* keep_flat,
                                    const int32_t* keep_count, int B, int P, int NC, int max_det, float* out, void* stream) {
  OD_REQUIRE(ctx && conf && boxes && keep_flat && keep_count && out, "od_gather_detections: null argument");
  OD_REQUIRE(B > 0 && P > 0 && NC > 0 && max_det > 0, "od_gather_detections: bad dims");
  hipLaunchKernelGGL(od_gather_det, dim3(B), dim3(256), 0, (hipStream_t)stream, conf, boxes, keep_flat, keep_count, P, NC,
                     max_det, out);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_head_postprocess(od_ctx* ctx, const float* pred, const float* priors, float* conf, float* boxes,
                                   int B, int P, int NC, float loc_scale, int clip, void* stream) {
  OD_REQUIRE(ctx && pred && priors && conf && boxes, "od_head_postprocess: null argument");
  OD_REQUIRE(B > 0 && P > 0 && NC > 0 && NC <= 76, "od_head_postprocess: bad dims (NC <= 76: 256 rows x (2 NC + 6) floats of LDS)");
  const long long rows = (long long)B * P;
  const size_t lds = (size_t)PP_ROWS * (NC + 6 + NC) * sizeof(float);
  if (int rc = od_ensure_lds(ctx, (const void*)&od_head_post, lds)) return rc;
  const unsigned grid = (unsigned)((rows + PP_ROWS - 1) / PP_ROWS);
  hipLaunchKernelGGL(od_head_post, dim3(grid), dim3(256), lds, (hipStream_t)stream, pred, priors, conf, boxes, rows, P,
                     NC, loc_scale, clip);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_decode_locs(od_ctx* ctx, const float* locs, const float* priors, float* boxes, int N, int P,
                              float loc_scale, int clip, void* stream) {
  OD_REQUIRE(ctx && locs && priors && boxes, "od_decode_locs: null argument");
  OD_REQUIRE(N > 0 && P > 0, "od_decode_locs: bad dims");
  const long long rows = (long long)N * P;
  const unsigned grid = (unsigned)((rows + 255) / 256);
  hipLaunchKernelGGL(od_decode, dim3(grid), dim3(256), 0, (hipStream_t)stream, locs, priors, boxes, rows, P, loc_scale,
                     clip);
  OD_CHECK_LAUNCH();
  return OD_OK;
}

extern "C" int od_upsample2x_add(od_ctx* ctx, const void* a, const void* up, void* out, int B, int H, int W, int C,
                                 void* stream) {
  OD_REQUIRE(ctx && a && up && out, "od_upsample2x_add: null argument");
  OD_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0,
             "od_upsample2x_add: H,W must be even and C a multiple of 8");
  const long long nvec = (long long)B * H * W * (C / 8);
  long long blocks = (nvec + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(od_up2_add, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f16*)a,
                     (const f16*)up, (f16*)out, nvec, H, W, C / 8);
  OD_CHECK_LAUNCH();
  return OD_OK;
}
