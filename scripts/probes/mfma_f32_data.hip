// What the exact-fp32 MFMA sustains depending on WHAT it multiplies: the same stream of
// independent-accumulator v_mfma_f32_32x32x2_f32 with (0) two smooth per-lane constants,
// (1) one pair of N(0,1) values per lane, (2) eight pairs of N(0,1) values per lane used in
// rotation (operands change from one MFMA to the next, as in a GEMM), (3) eight pairs of random
// bit patterns, (4) as 2 but A and B name the same register.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_f32_data scripts/probes/mfma_f32_data.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k32(const float* __restrict__ src, float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int u = 0; u < 8; ++u) {
    if (MODE == 0) {
      a[u] = threadIdx.x * 1e-3f;
      b[u] = 1.0f + threadIdx.x * 1e-4f;
    } else {
      const int v = (MODE == 1) ? 0 : u;
      a[u] = src[(2 * v) * 256 + threadIdx.x];
      b[u] = src[(2 * v + 1) * 256 + threadIdx.x];
    }
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (MODE == 4)
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, %0" : "+a"(acc[i]) : "v"(a[u]));
        else
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[u]), "v"(b[u]));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10000;
  const int blocks = argc > 2 ? atoi(argv[2]) : 512;
  float *out, *src_n, *src_bits;
  hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  hipMalloc(&src_n, 16 * 256 * sizeof(float));
  hipMalloc(&src_bits, 16 * 256 * sizeof(float));
  std::vector<float> hn(16 * 256);
  std::vector<unsigned> hb(16 * 256);
  srand(1);
  for (int i = 0; i < 16 * 256; ++i) {
    const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    hn[i] = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
    hb[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();
  }
  hipMemcpy(src_n, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(src_bits, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[5] = {"smooth constants", "one N(0,1) pair", "eight N(0,1) pairs in rotation",
                          "eight pairs of random bits", "eight N(0,1) values, A and B one register"};
  for (int pass = 0; pass < 2; ++pass)
    for (int mode = 0; mode < 5; ++mode) {
      float ms = 0.f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        const float* src = mode == 3 ? src_bits : src_n;
        switch (mode) {
          case 0: hipLaunchKernelGGL(k32<0>, dim3(blocks), dim3(256), 0, 0, src, out, iters); break;
          case 1: hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, src, out, iters); break;
          case 2: hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, src, out, iters); break;
          case 3: hipLaunchKernelGGL(k32<3>, dim3(blocks), dim3(256), 0, 0, src, out, iters); break;
          default: hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, src, out, iters); break;
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
      printf("{\"operands\": \"%s\", \"ms\": %.3f, \"tflops\": %.1f}\n", names[mode], ms,
             flops / ms / 1e9);
    }
  return 0;
}
