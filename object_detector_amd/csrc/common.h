// Internal helpers shared by every translation unit of libodhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <unordered_map>

#include "../../include/odhip.h"

struct od_ctx {
  int device;
  void* zero_page;  // 8 KiB of zeros: source for padded / out-of-range lanes of LDS-DMA gathers; a zero bias vector
  float* ones;      // 2048 x 1.0f: identity scale vector of od_conv2d_bwd_data
  int num_cu;
  // largest dynamic-LDS size already granted to each kernel ON THIS DEVICE (hipFuncAttributeMaxDynamicSharedMemorySize is
  // per device, and one process may hold a context per GPU): see od_ensure_lds
  std::unordered_map<const void*, size_t> lds_attr;
};

// Raise the kernel's dynamic-LDS limit to `lds` bytes on ctx's device unless that was already done through this context.
int od_ensure_lds(od_ctx* ctx, const void* fn, size_t lds);

void od_set_error(const char* fmt, ...);

#define OD_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      od_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));     \
      return OD_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define OD_CHECK_LAUNCH()                                                                    \
  do {                                                                                       \
    hipError_t e_ = hipGetLastError();                                                       \
    if (e_ != hipSuccess) {                                                                  \
      od_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return OD_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define OD_REQUIRE(cond, ...)   \
  do {                          \
    if (!(cond)) {              \
      od_set_error(__VA_ARGS__); \
      return OD_ERR_INVALID;    \
    }                           \
  } while (0)

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int od_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int od_round_up(int a, int b) { return od_ceil_div(a, b) * b; }

// conv launcher (conv_mfma.hip); kernel_name receives the device symbol launched (may be NULL)
int od_conv2d_fwd_impl(od_ctx* ctx, const od_conv_desc* d, hipStream_t stream, const char** kernel_name,
                       bool dry_run);
const char* od_conv_first_kernel_name();
const char* od_bottleneck_kernel_name(int C);
const char* od_stem_kernel_name();

// weight-gradient launcher (conv_wgrad.hip): per-split f32 slabs [split][Cout][k*k*Cin]; *nsplit receives the split count
int od_wgrad_slabs_impl(od_ctx* ctx, const void* x, const void* dz, float* slabs, int B, int H, int W, int Cin, int Cout,
                        int ksize, int stride, void* stream, int* nsplit);
