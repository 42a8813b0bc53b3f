// How fast can a CU fill LDS from L2-resident data? Three ways, each moving the same 64 KiB per
// trip with 512 threads (8 waves), 16 bytes per lane per instruction:
//   0: LDS-DMA (global_load_lds, 16 B per lane)
//   1: global_load_dwordx4 into registers, then ds_write_b128
//   2: global_load_dwordx4 only (no LDS write): what the load path alone sustains
// and, next to them, 3: the same DMA while the waves also issue ds_read_b128 at the GEMM's rate
// (24 reads per wave per trip). Prints bytes per clock per CU (shader clock from s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/lds_fill_rate scripts/probes/lds_fill_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void global_void;

template <int MODE, bool BARRIER>
__global__ __launch_bounds__(512) void fill(const char* __restrict__ src, int* out, int trips,
                                            unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every workgroup reads the same 256 KiB (L2 hits), a different quarter each trip so that the
  // loads cannot be hoisted out of the loop
  i32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < trips; ++t) {
    char* buf = smem + (t & 1) * 65536;
    const char* base = src + (size_t)(t & 3) * 65536;
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds((global_void*)(base + (wave * 8 + j) * 1024 + lane * 16),
                                         (lds_void*)(buf + (wave * 8 + j) * 1024), 16, 0, 0);
      if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 24; ++j) {
          const i32x4 v = *reinterpret_cast<const i32x4*>(smem + ((t + 1) & 1) * 65536 +
                                                          ((wave * 24 + j) & 63) * 1024 + lane * 16);
          acc += v;
        }
      }
      // at most two trips of DMA in flight: throughput, not latency
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      i32x4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        r[j] = *reinterpret_cast<const i32x4*>(base + (wave * 8 + j) * 1024 + lane * 16);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          *reinterpret_cast<i32x4*>(buf + (wave * 8 + j) * 1024 + lane * 16) = r[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += r[j];
      }
    }
    if (BARRIER) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  const i32x4 v = *reinterpret_cast<const i32x4*>(smem + tid * 16);
  out[blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3] + v[0];
}

int main(int argc, char** argv) {
  const int trips = argc > 1 ? atoi(argv[1]) : 4000;
  const int blocks = 256;
  char* src;
  int* out;
  unsigned long long* cyc;
  (void)hipMalloc(&src, (size_t)4 * 65536);
  (void)hipMemset(src, 1, (size_t)4 * 65536);
  (void)hipMalloc(&out, blocks * 512 * sizeof(int));
  (void)hipMalloc(&cyc, blocks * sizeof(unsigned long long));
  const size_t lds = 2 * 65536;
#define ALLOW(M)                                                                                  \
  (void)hipFuncSetAttribute((const void*)fill<M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
  (void)hipFuncSetAttribute((const void*)fill<M, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  ALLOW(0) ALLOW(1) ALLOW(2) ALLOW(3)
  const char* names[4] = {"lds-dma 16B/lane", "global_load_dwordx4 + ds_write_b128",
                          "global_load_dwordx4 only", "lds-dma + 24 ds_read_b128 per wave"};
  for (int bar = 0; bar < 2; ++bar)
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
#define RUN(M)                                                                                    \
  if (bar) hipLaunchKernelGGL((fill<M, true>), dim3(blocks), dim3(512), lds, 0, src, out, trips, cyc); \
  else hipLaunchKernelGGL((fill<M, false>), dim3(blocks), dim3(512), lds, 0, src, out, trips, cyc);
      switch (mode) {
        case 0: RUN(0) break;
        case 1: RUN(1) break;
        case 2: RUN(2) break;
        default: RUN(3) break;
      }
      (void)hipDeviceSynchronize();
    }
    unsigned long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += (double)h[i];
    mean /= blocks;
    printf("{\"mode\": \"%s\", \"barrier_per_trip\": %d, \"cycles_per_64KiB\": %.1f, \"bytes_per_clk_per_cu\": %.1f}\n",
           names[mode], bar, mean / trips, 65536.0 * trips / mean);
  }
  return 0;
}
