// Sustained rate of the two exact-fp32 MFMA shapes with nothing else running: every wave issues
// a long stream of independent-accumulator MFMAs on register operands. Same peak on paper
// (256 flop/clk/CU); the question is what the power-limited clock makes of each.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_f32_rate scripts/probes/mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = argc > 2 ? atoi(argv[2]) : 512;     // 256 CUs x 2 workgroups of 4 waves
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 4; ++which) {
    const bool big = (which & 1) == 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (big) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters);
      else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      // per wave per iteration: 16 MFMAs x 4096 flop (32x32x2) or 32 MFMAs x 2048 flop (16x16x4)
      const double flops = (double)blocks * 4 * iters * 16 * 4096.0;
      if (rep == 2)
        printf("{\"mfma\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f}\n", big ? "32x32x2" : "16x16x4",
               ms, flops / ms / 1e9);
    }
  }
  return 0;
}
