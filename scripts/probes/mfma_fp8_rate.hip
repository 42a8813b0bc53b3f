// Sustained rate of the two block-scaled fp8 (e4m3) MFMA shapes with nothing else running:
// register operands filled with pseudo-random fp8 values (or zeros), unit scales, independent
// AGPR accumulators, inline-asm MFMAs (see mfma_bf16_rate.hip for why). What the power-limited
// clock allows the matrix cores alone in the fp8 GEMM's instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_fp8_rate scripts/probes/mfma_fp8_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__device__ i32x8 rnd(unsigned s) {
  i32x8 v;
  for (int i = 0; i < 8; ++i) {
    unsigned w = 0;
    for (int b = 0; b < 4; ++b) {
      s = s * 1664525u + 1013904223u;
      // e4m3 with exponent field 6 or 7 (|x| in [0.5, 2)), random sign and mantissa
      w |= (((s >> 16) & 0x8f) | 0x30) << (8 * b);
    }
    v[i] = (int)w;
  }
  return v;
}

__global__ __launch_bounds__(256) void k16(float* out, int iters, int zero) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = rnd(threadIdx.x * 8 + i + blockIdx.x * 977);
    b[i] = rnd(threadIdx.x * 8 + i + 4 + blockIdx.x * 131);
    if (zero) { a[i] = i32x8{0, 0, 0, 0, 0, 0, 0, 0}; b[i] = a[i]; }
  }
  const int one = 0x7f7f7f7f;        // e8m0 scale 2^0
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                     : "+a"(acc[i * 4 + j]) : "v"(a[i]), "v"(b[j]), "v"(one));
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k32(float* out, int iters, int zero) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  i32x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) {
    a[i] = rnd(threadIdx.x * 8 + i + blockIdx.x * 977);
    b[i] = rnd(threadIdx.x * 8 + i + 4 + blockIdx.x * 131);
    if (zero) { a[i] = i32x8{0, 0, 0, 0, 0, 0, 0, 0}; b[i] = a[i]; }
  }
  const int one = 0x7f7f7f7f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                       : "+a"(acc[i * 2 + j]) : "v"(a[i]), "v"(b[j]), "v"(one));
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 50000;
  const int blocks = argc > 2 ? atoi(argv[2]) : 512;
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 4; ++which) {
    const bool big = (which & 1) == 1;
    const int zero = which >= 2;
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (big) hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, zero);
      else hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, out, iters, zero);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // per wave per iteration: 16 x 65536 flop (16x16x128) or 8 x 131072 flop (32x32x64)
    const double flops = (double)blocks * 4 * iters * 16 * 65536.0;
    printf("{\"mfma\": \"%s\", \"operands\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f}\n",
           big ? "scale_32x32x64_f8" : "scale_16x16x128_f8", zero ? "zero" : "random e4m3", ms,
           flops / ms / 1e9);
  }
  return 0;
}
