// Does hardware register 29 (SHADER_CYCLES on newer parts) count shader clocks on gfx950? Two reads
// around a fixed run of MFMAs, next to s_memtime (100 MHz reference clock) around the same run.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ void probe(unsigned long long* out) {
  unsigned c0, c1;
  unsigned long long t0, t1;
  f32x4 acc = {0, 0, 0, 0};
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_getreg_b32 %0, hwreg(29, 0, 20)" : "=s"(c0));
#pragma unroll
  for (int i = 0; i < 256; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  asm volatile("s_getreg_b32 %0, hwreg(29, 0, 20)" : "=s"(c1));
  t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = c0; out[1] = c1; out[2] = t1 - t0; out[3] = (unsigned long long)acc[0];
  }
}
int main() {
  unsigned long long* d; unsigned long long h[4];
  (void)hipMalloc(&d, 32);
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("hwreg29: %llu -> %llu (delta %lld), s_memtime delta %llu (x10 ns)\n", h[0], h[1],
           (long long)((h[1] - h[0]) & 0xfffff), h[2]);
  }
  return 0;
}
