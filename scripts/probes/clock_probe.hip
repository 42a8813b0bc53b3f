// Effective shader clock of short, sparse kernels: a one-workgroup kernel runs a dependent FMA
// chain of known length and reports shader-clock cycles (s_memtime) and wall time (the constant
// 100 MHz counter) from inside; the host launches it back to back or with idle gaps.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/clock_probe scripts/probes/clock_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ void chain_kernel(float* out, unsigned long long* stamps, int iters) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  float x = (float)threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999f, 0.001f);
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = w1 - w0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const int gap_us = argc > 2 ? atoi(argv[2]) : 0;
  const int launches = argc > 3 ? atoi(argv[3]) : 200;
  float* out; unsigned long long* stamps; unsigned long long host[2];
  hipMalloc(&out, 1024 * sizeof(float));
  hipMalloc(&stamps, 2 * sizeof(unsigned long long));
  double cyc = 0, wall = 0;
  for (int l = 0; l < launches; ++l) {
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, out, stamps, iters);
    hipMemcpy(host, stamps, sizeof(host), hipMemcpyDeviceToHost);
    if (l >= launches / 2) { cyc += (double)host[0]; wall += (double)host[1]; }
    if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
  }
  const int n = launches - launches / 2;
  printf("{\"iters\": %d, \"gap_us\": %d, \"cycles_per_launch\": %.0f, \"wall_ns_per_launch\": %.0f, "
         "\"cycles_per_fma\": %.2f, \"effective_ghz\": %.3f}\n",
         iters, gap_us, cyc / n, wall / n * 10.0, cyc / n / iters, (cyc / n) / (wall / n * 10.0));
  return 0;
}
