// Sustained rate of the two bf16 MFMA shapes with nothing else running (register operands,
// independent accumulators, eight rotating operand registers filled with pseudo-random bf16
// values): what the power-limited clock allows the matrix cores alone. The MFMAs are inline
// asm with the accumulator tied to an AGPR tuple (the compiler otherwise shuffles accumulators
// through v_accvgpr moves and pads with s_nop, which is not what a GEMM inner loop looks like).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_bf16_rate scripts/probes/mfma_bf16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ bf16x8 rnd(unsigned s) {
  bf16x8 v;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    v[i] = (short)(0x3f00 | ((s >> 16) & 0x80ff));     // +-[0.5, 1) with random mantissa
  }
  return v;
}

__global__ __launch_bounds__(256) void k16(float* out, int iters, int zero) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = rnd(threadIdx.x * 8 + i + blockIdx.x * 977);
    b[i] = rnd(threadIdx.x * 8 + i + 4 + blockIdx.x * 131);
    if (zero) { a[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; b[i] = a[i]; }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
                     : "+a"(acc[i * 4 + j]) : "v"(a[i]), "v"(b[j]));
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k32(float* out, int iters, int zero) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) {
    a[i] = rnd(threadIdx.x * 8 + i + blockIdx.x * 977);
    b[i] = rnd(threadIdx.x * 8 + i + 4 + blockIdx.x * 131);
    if (zero) { a[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; b[i] = a[i]; }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                       : "+a"(acc[i * 2 + j]) : "v"(a[i]), "v"(b[j]));
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 100000;
  const int blocks = argc > 2 ? atoi(argv[2]) : 512;
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 6; ++which) {
    const bool big = (which & 1) == 1;
    const int zero = which >= 4;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (big) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, zero);
      else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters, zero);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // per wave per iteration: 16 x 16384 flop (16x16x32) or 8 x 32768 flop (32x32x16)
    const double flops = (double)blocks * 4 * iters * 16 * 16384.0;
    printf("{\"mfma\": \"%s\", \"operands\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f}\n",
           big ? "32x32x16" : "16x16x32", zero ? "zero" : "random", ms, flops / ms / 1e9);
  }
  return 0;
}
