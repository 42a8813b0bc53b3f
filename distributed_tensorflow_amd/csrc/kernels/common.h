// Shared declarations for the gfx950 kernel library (libdtf_hip.so).
//
// Every launcher is a plain C function taking raw device pointers, sizes and the hipStream_t
// to enqueue on, so the Python runtime can call it through ctypes with torch tensors'
// data_ptr() and torch's current stream (which may be a capturing stream: launchers never
// allocate, synchronise or query, so they are HIP-graph capturable).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define DTF_API extern "C" __attribute__((visibility("default")))

// Launchers return 0 on success or a negative code for a rejected argument; a HIP launch
// error is returned as the positive hipError_t.
#define DTF_OK 0
#define DTF_ERR_BAD_ARG (-1)
#define DTF_ERR_UNSUPPORTED (-2)

#define DTF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

// activation codes shared by the fused GEMM epilogue and the elementwise kernels
enum DtfActivation { DTF_ACT_NONE = 0, DTF_ACT_SIGMOID = 1, DTF_ACT_RELU = 2, DTF_ACT_TANH = 3 };

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 in 4 VGPRs
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float dtf_apply_act(float v, int act) {
  switch (act) {
    case DTF_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case DTF_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case DTF_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__device__ __forceinline__ float dtf_bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}

// round-to-nearest-even f32 -> bf16 (plain cast semantics; NaN stays NaN)
__device__ __forceinline__ unsigned short dtf_f32_to_bf16(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// wave64 all-lane sum / max via butterfly shuffles
__device__ __forceinline__ float dtf_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float dtf_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// Remap a linear workgroup id so that each of the 8 XCDs (which receive workgroups
// round-robin) works on a contiguous chunk of the tile grid and neighbouring tiles share an
// L2. Bijective for any grid size.
__device__ __forceinline__ int dtf_xcd_remap(int orig, int nwg) {
  const int nxcd = 8;
  int xcd = orig % nxcd;
  int q = nwg / nxcd, r = nwg % nxcd;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + orig / nxcd;
}
