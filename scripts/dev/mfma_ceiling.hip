// What the chip sustains on v_mfma_f32_16x16x32_f16 with operands in registers (no LDS, no global traffic in the loop):
// the clock it holds under that load on RANDOM data sets the practical ceiling of every MFMA-bound kernel of this
// repository (MI355X_MICROARCH.md "DVFS give-back").  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 scripts/dev/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop(const f16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(t * 8 + i) & 0xFFFFF];
    b[i] = in[(t * 8 + 4 + i) & 0xFFFFF];
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[t] = s;
}

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                  \
      return 1;                                                               \
    }                                                                         \
  } while (0)

template <int WAVES>
static int run(const char* tag, const f16x8* din, float* dout, int cus, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<WAVES>, dim3(cus), dim3(WAVES * 64), 0, 0, din, dout, iters);
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0.f;
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_loop<WAVES>, dim3(cus), dim3(WAVES * 64), 0, 0, din, dout, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double flop = (double)cus * WAVES * iters * 16.0 * (2.0 * 16 * 16 * 32);
  printf("%-34s %d waves/CU  %8.3f ms avg  %7.1f TFLOP/s avg  %7.1f best\n", tag, WAVES, sum / reps, flop / (sum / reps * 1e-3) / 1e12,
         flop / (best * 1e-3) / 1e12);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t n = 1 << 20;
  std::vector<f16x8> h(n);
  srand(1);
  for (size_t i = 0; i < n; ++i)
    for (int e = 0; e < 8; ++e) h[i][e] = (f16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
  f16x8* din;
  float* dout;
  CK(hipMalloc(&din, n * sizeof(f16x8)));
  CK(hipMalloc(&dout, (size_t)cus * 1024 * sizeof(float)));
  const int iters = 40000;  // ~1.5-3 ms per launch: long enough for the clock to settle
  CK(hipMemcpy(din, h.data(), n * sizeof(f16x8), hipMemcpyHostToDevice));
  if (run<4>("random operands", din, dout, cus, iters)) return 1;
  if (run<8>("random operands", din, dout, cus, iters / 2)) return 1;
  CK(hipMemset(din, 0, n * sizeof(f16x8)));
  if (run<4>("zero operands", din, dout, cus, iters)) return 1;
  if (run<8>("zero operands", din, dout, cus, iters / 2)) return 1;
  printf("%d CUs, %d MHz nominal\n", cus, prop.clockRate / 1000);
  return 0;
}
