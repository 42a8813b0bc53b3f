// Which XCD does workgroup i of a one-workgroup-per-CU launch run on? Every workgroup stores the XCC_ID
// hardware register (and its CU / SE ids) at its index; the host prints them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
  extern __shared__ char smem[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hwid;
    smem[0] = 1;
  }
  // stay resident for a while so that every workgroup gets a CU of its own
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  unsigned* d;
  hipMalloc(&d, n * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(probe, dim3(n), dim3(512), 131072, 0, d);
  std::vector<unsigned> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("%d xcc=%u cu=%u se=%u%s", i, h[2 * i] & 15, (h[2 * i + 1] >> 8) & 15, (h[2 * i + 1] >> 13) & 7,
           (i % 8 == 7) ? "\n" : "  ");
  return 0;
}
