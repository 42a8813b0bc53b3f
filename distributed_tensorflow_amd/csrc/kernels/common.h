// Shared declarations for the gfx950 kernel library (libdtf_hip.so).
//
// Every launcher is a plain C function taking raw device pointers, sizes and the hipStream_t
// to enqueue on, so the Python runtime can call it through ctypes with torch tensors'
// data_ptr() and torch's current stream (which may be a capturing stream: launchers never
// allocate, synchronise or query, so they are HIP-graph capturable).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

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
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;      // 32 fp8 in 8 VGPRs

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

// two f32 -> two bf16 in one dword (lo | hi << 16) with the hardware convert of gfx950
// (v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays a quiet NaN -- the results of
// dtf_f32_to_bf16 in one instruction instead of five per element)
__device__ __forceinline__ unsigned dtf_f32x2_to_bf16x2(float lo, float hi) {
  typedef float dtf_f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 dtf_bf16x2_t __attribute__((ext_vector_type(2)));
  const dtf_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dtf_bf16x2_t));
}

// f32 -> OCP fp8 e4m3fn (bias 7, 3 mantissa bits, no infinities, max 448), round-to-nearest-
// even, SATURATING: |f| beyond 448 becomes +-448 (0x7e), never NaN. NaN -> 0x7f.
__device__ __forceinline__ unsigned char dtf_f32_to_e4m3(float f) {
  const unsigned int u = __float_as_uint(f);
  const unsigned int sign = (u >> 24) & 0x80u;
  unsigned int a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (unsigned char)(sign | 0x7fu);
  if (a < 0x3c800000u) {                       // |f| < 2^-6: subnormal grid, step 2^-9
    const float q = rintf(__uint_as_float(a) * 512.0f);   // 0..8 (8 = the smallest normal)
    return (unsigned char)(sign | (unsigned int)q);
  }
  a += 0x7ffffu + ((a >> 20) & 1u);            // RNE at mantissa bit 20
  a &= ~0xfffffu;
  const int e = (int)(a >> 23) - 127 + 7;
  const unsigned int m = (a >> 20) & 7u;
  if (e > 15 || (e == 15 && m == 7u)) return (unsigned char)(sign | 0x7eu);
  return (unsigned char)(sign | ((unsigned int)e << 3) | m);
}

__device__ __forceinline__ float dtf_e4m3_to_f32(unsigned char b) {
  const unsigned int sign = ((unsigned int)b & 0x80u) << 24;
  const unsigned int e = (b >> 3) & 15u, m = b & 7u;
  if (e == 15u && m == 7u) return __uint_as_float(sign | 0x7fc00000u);
  if (e == 0u) return __uint_as_float(sign | __float_as_uint((float)m * 0.001953125f));
  return __uint_as_float(sign | ((e + 120u) << 23) | (m << 20));
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

// ---- host side: things a launcher sets up once per device ---------------------------------
// (per device: a process may drive several GPUs, and both the CU count and a kernel's dynamic-LDS
// opt-in belong to the device that is current at the launch)
static inline int dtf_current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  return dev;
}

// Runs BODY once per device (the first launcher call made while that device is current); other
// threads arriving meanwhile wait until the attributes are in place.
struct DtfPerDeviceOnce {
  std::mutex mu;
  unsigned long long done = 0;
};
template <typename F>
static inline void dtf_once_per_device(DtfPerDeviceOnce& once, F&& body) {
  const unsigned long long bit = 1ull << dtf_current_device();
  std::lock_guard<std::mutex> lk(once.mu);
  if (!(once.done & bit)) {
    body();
    once.done |= bit;
  }
}
