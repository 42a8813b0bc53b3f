// Small data-movement and bookkeeping kernels around the GEMMs: strided 2-D block copy (the
// assembly of a sharded product's result blocks, variable shards), in-place variable updates
// (AssignAdd / AssignSub), broadcast / reduction gradients, log-softmax gradient, sign, the
// truncated-normal initialiser and the abs-max reduction fp8 quantisation scales come from.
// All memory-bound; 16 B per lane where alignment allows.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_BLOCKS = 2048;

inline int grid_for(long n, int per_thread) {
  long blocks = (n + (long)BLOCK * per_thread - 1) / ((long)BLOCK * per_thread);
  if (blocks < 1) blocks = 1;
  return (int)(blocks > MAX_BLOCKS ? MAX_BLOCKS : blocks);
}

// dst[r, 0:row_bytes) = src[r, 0:row_bytes) for r < rows; pitches in bytes. VEC = bytes per lane.
template <typename V>
__global__ __launch_bounds__(BLOCK) void copy_2d_kernel(const unsigned char* __restrict__ src,
                                                        unsigned char* __restrict__ dst, long rows,
                                                        long row_units, long src_pitch,
                                                        long dst_pitch) {
  const long n = rows * row_units;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long r = i / row_units, u = i - r * row_units;
    *reinterpret_cast<V*>(dst + r * dst_pitch + u * (long)sizeof(V)) =
        *reinterpret_cast<const V*>(src + r * src_pitch + u * (long)sizeof(V));
  }
}

// dst[r, 0:dst_units) = src[r, 0:src_units) followed by zeros, in units of 16 bytes: rows copied
// into wider, zero-padded rows (a GEMM operand whose K is not a multiple of the kernel's K-tile)
__global__ __launch_bounds__(BLOCK) void pad_rows_kernel(const unsigned char* __restrict__ src,
                                                         unsigned char* __restrict__ dst, long rows,
                                                         long src_units, long dst_units,
                                                         long src_pitch, long dst_pitch) {
  const long n = rows * dst_units;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long r = i / dst_units, u = i - r * dst_units;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (u < src_units) v = *reinterpret_cast<const u32x4*>(src + r * src_pitch + u * 16);
    *reinterpret_cast<u32x4*>(dst + r * dst_pitch + u * 16) = v;
  }
}

// dst[r, c] = src[r, c] for r < rows, c < cols, zero elsewhere, for all r < dst_rows, c < dst_ld
// (elements of 2 or 4 bytes; any alignment): an operand copied into an aligned, zero-padded image
template <typename T>
__global__ __launch_bounds__(BLOCK) void pad_2d_kernel(const T* __restrict__ src,
                                                       T* __restrict__ dst, long rows, long cols,
                                                       long src_ld, long dst_rows, long dst_ld) {
  const long n = dst_rows * dst_ld;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long r = i / dst_ld, c = i - r * dst_ld;
    dst[i] = (r < rows && c < cols) ? src[r * src_ld + c] : (T)0;
  }
}

// var[i] += sign * delta[i % dn]
__global__ __launch_bounds__(BLOCK) void assign_add_kernel(float* __restrict__ var,
                                                           const float* __restrict__ delta, long n,
                                                           long dn, float sign) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    var[i] += sign * delta[dn == n ? i : i % dn];
}

__global__ __launch_bounds__(BLOCK) void sign_kernel(const float* __restrict__ x,
                                                     float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const float v = x[i];
    out[i] = v > 0.f ? 1.f : (v < 0.f ? -1.f : v);      // keeps +-0 and NaN
  }
}

// out[r, c] = (idx[r] == c) ? on : off for `rows` indices and `depth` classes (tf.one_hot on the
// last axis); an index outside [0, depth) yields a row of `off`. WIDE: 64-bit indices.
template <bool WIDE>
__global__ __launch_bounds__(BLOCK) void one_hot_kernel(const void* __restrict__ idx,
                                                        float* __restrict__ out, long rows,
                                                        long depth, float on, float off) {
  const long n = rows * depth;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long r = i / depth, c = i - r * depth;
    const long k = WIDE ? reinterpret_cast<const long*>(idx)[r]
                        : (long)reinterpret_cast<const int*>(idx)[r];
    out[i] = k == c ? on : off;
  }
}

// Gradient of a reduction / broadcast target: out[o, m, i] = g[o, i] * scale, where the output is
// viewed as [outer, mid, inner] and g as [outer, inner] (mid = the reduced extent). Covers
// reduce over all axes (outer = inner = 1), over axis 1 of a matrix (inner = 1), over axis 0
// (outer = 1) and the bias-style broadcast of a vector over rows (outer = 1).
__global__ __launch_bounds__(BLOCK) void broadcast_mid_kernel(const float* __restrict__ g,
                                                              float* __restrict__ out, long outer,
                                                              long mid, long inner, float scale) {
  const long n = outer * mid * inner;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long in = i % inner, o = i / (inner * mid);
    out[i] = g[o * inner + in] * scale;
  }
}

// dx = dy - exp(y) * rowsum(dy), y = log-softmax output; one wave per row
__global__ __launch_bounds__(BLOCK) void log_softmax_grad_kernel(const float* __restrict__ y,
                                                                 const float* __restrict__ dy,
                                                                 float* __restrict__ dx, long rows,
                                                                 long cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long r = blockIdx.x * (long)(BLOCK / 64) + wave; r < rows;
       r += (long)gridDim.x * (BLOCK / 64)) {
    const float* yr = y + r * cols;
    const float* dr = dy + r * cols;
    float s = 0.f;
    for (long c = lane; c < cols; c += 64) s += dr[c];
    s = dtf_wave_sum(s);
    for (long c = lane; c < cols; c += 64) dx[r * cols + c] = dr[c] - expf(yr[c]) * s;
  }
}

// ---- abs-max (fixed-order two-pass; the result is the same bits on every launch) --------------
template <bool SRC_BF16>
__global__ __launch_bounds__(BLOCK) void absmax_partial_kernel(const void* __restrict__ x_, long n,
                                                               float* __restrict__ partial) {
  __shared__ float red[BLOCK / 64];
  float m = 0.f;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const float v = SRC_BF16 ? dtf_bf16_to_f32(reinterpret_cast<const unsigned short*>(x_)[i])
                             : reinterpret_cast<const float*>(x_)[i];
    m = fmaxf(m, fabsf(v));
  }
  m = dtf_wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0];
    for (int i = 1; i < BLOCK / 64; ++i) t = fmaxf(t, red[i]);
    partial[blockIdx.x] = t;
  }
}

// out[0] = max(partials); out[1] = scale = target / max (1 when max == 0); out[2] = 1 / scale
__global__ __launch_bounds__(64) void absmax_final_kernel(const float* __restrict__ partial, int n,
                                                          float target, float* __restrict__ out) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, partial[i]);
  m = dtf_wave_max(m);
  if (threadIdx.x == 0) {
    const float scale = m > 0.f ? target / m : 1.f;
    out[0] = m;
    out[1] = scale;
    out[2] = 1.f / scale;
  }
}

// quantise with the scale read from device memory (no host round trip)
template <bool SRC_BF16>
__global__ __launch_bounds__(BLOCK) void cast_to_fp8_dev_scale_kernel(
    const void* __restrict__ x_, unsigned char* __restrict__ out, long n,
    const float* __restrict__ scale_ptr) {
  const float scale = *scale_ptr;
  const long n4 = n >> 2;
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  for (long i = start; i < n4; i += step) {
    f32x4 v;
    if (SRC_BF16) {
      const bf16x4 h = reinterpret_cast<const bf16x4*>(x_)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = dtf_bf16_to_f32((unsigned short)h[j]);
    } else {
      v = reinterpret_cast<const f32x4*>(x_)[i];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j] * scale, -448.f, 448.f);
    int packed = 0;
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], packed, false);
    packed = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], packed, true);
    reinterpret_cast<int*>(out)[i] = packed;
  }
  for (long i = (n4 << 2) + start; i < n; i += step) {
    const float f = SRC_BF16 ? dtf_bf16_to_f32(reinterpret_cast<const unsigned short*>(x_)[i])
                             : reinterpret_cast<const float*>(x_)[i];
    out[i] = dtf_f32_to_e4m3(__builtin_amdgcn_fmed3f(f * scale, -448.f, 448.f));
  }
}

// ---- truncated normal: N(0,1) restricted to [-2, 2] by inverse-CDF of a Philox uniform --------
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0,
                                             unsigned int k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0;
  const unsigned int hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
  c[0] = hi1 ^ c[1] ^ k0;
  c[1] = lo1;
  c[2] = hi0 ^ c[3] ^ k1;
  c[3] = lo0;
}

__device__ __forceinline__ float philox_uniform_at(long e, unsigned int seed,
                                                   unsigned int op_seed) {
  const unsigned long long b = (unsigned long long)e >> 2;
  unsigned int c[4] = {(unsigned int)b, (unsigned int)(b >> 32), 0u, 0u};
  unsigned int k0 = seed, k1 = op_seed;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const unsigned int bits = c[e & 3];
  return (float)(bits >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;   // (0, 1)
}

// Acklam-style rational approximation of the normal quantile, refined by one Newton step on
// erfc (|error| < 1e-7 in fp32 over the [-2, 2] range used here)
__device__ __forceinline__ float normal_quantile(float p) {
  const float a[6] = {-3.969683028665376e+01f, 2.209460984245205e+02f, -2.759285104469687e+02f,
                      1.383577518672690e+02f, -3.066479806614716e+01f, 2.506628277459239e+00f};
  const float b[5] = {-5.447609879822406e+01f, 1.615858368580409e+02f, -1.556989798598866e+02f,
                      6.680131188771972e+01f, -1.328068155288572e+01f};
  const float q = p - 0.5f, r = q * q;
  float x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
            (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0f);
  // Newton: f(x) = Phi(x) - p, f'(x) = phi(x)
  const float e = 0.5f * erfcf(-x * 0.70710678118654752f) - p;
  const float u = e * 2.50662827463100050f * expf(0.5f * x * x);
  x = x - u / (1.0f + 0.5f * x * u);
  return x;
}

__global__ __launch_bounds__(BLOCK) void truncated_normal_kernel(float* __restrict__ out, long n,
                                                                 unsigned int seed,
                                                                 unsigned int op_seed, float mean,
                                                                 float stddev) {
  const float lo = 0.02275013194817921f;           // Phi(-2)
  const float span = 1.0f - 2.0f * lo;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const float u = philox_uniform_at(i, seed, op_seed);
    float z = normal_quantile(lo + u * span);
    z = fminf(fmaxf(z, -2.0f), 2.0f);
    out[i] = mean + stddev * z;
  }
}

// ---- generic cast between the element types a graph can carry --------------------------------
enum CastKind { K_F32 = 0, K_BF16 = 1, K_I32 = 2, K_I64 = 3, K_BOOL = 4, K_F64 = 5, K_FP8 = 6 };

template <int KIND>
__device__ __forceinline__ double load_as_double(const void* p, long i) {
  if (KIND == K_F32) return (double)reinterpret_cast<const float*>(p)[i];
  if (KIND == K_BF16) return (double)dtf_bf16_to_f32(reinterpret_cast<const unsigned short*>(p)[i]);
  if (KIND == K_I32) return (double)reinterpret_cast<const int*>(p)[i];
  if (KIND == K_I64) return (double)reinterpret_cast<const long*>(p)[i];
  if (KIND == K_BOOL) return reinterpret_cast<const unsigned char*>(p)[i] ? 1.0 : 0.0;
  if (KIND == K_F64) return reinterpret_cast<const double*>(p)[i];
  return (double)dtf_e4m3_to_f32(reinterpret_cast<const unsigned char*>(p)[i]);
}

template <int KIND>
__device__ __forceinline__ void store_from_double(void* p, long i, double v, long iv) {
  if (KIND == K_F32) reinterpret_cast<float*>(p)[i] = (float)v;
  else if (KIND == K_BF16) reinterpret_cast<unsigned short*>(p)[i] = dtf_f32_to_bf16((float)v);
  else if (KIND == K_I32) reinterpret_cast<int*>(p)[i] = (int)iv;
  else if (KIND == K_I64) reinterpret_cast<long*>(p)[i] = iv;
  else if (KIND == K_BOOL) reinterpret_cast<unsigned char*>(p)[i] = (v != 0.0 || iv != 0) ? 1 : 0;
  else if (KIND == K_F64) reinterpret_cast<double*>(p)[i] = v;
  else reinterpret_cast<unsigned char*>(p)[i] = dtf_f32_to_e4m3((float)v);
}

// integer sources keep their exact value on the way to an integer destination (a 64-bit index does
// not survive a trip through double); floating sources truncate toward zero like a C cast
template <int SRC, int DST>
__global__ __launch_bounds__(BLOCK) void cast_generic_kernel(const void* __restrict__ src,
                                                             void* __restrict__ dst, long n) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    double v;
    long iv;
    if (SRC == K_I64) { iv = reinterpret_cast<const long*>(src)[i]; v = (double)iv; }
    else if (SRC == K_I32) { iv = reinterpret_cast<const int*>(src)[i]; v = (double)iv; }
    else { v = load_as_double<SRC>(src, i); iv = (long)v; }
    store_from_double<DST>(dst, i, v, iv);
  }
}

template <int SRC>
int cast_dispatch_dst(const void* src, void* dst, int dst_kind, long n, hipStream_t stream) {
  const dim3 grid(grid_for(n, 1)), block(BLOCK);
  switch (dst_kind) {
#define DTF_CAST_DST(K) \
    case K: hipLaunchKernelGGL((cast_generic_kernel<SRC, K>), grid, block, 0, stream, src, dst, n); break;
    DTF_CAST_DST(K_F32) DTF_CAST_DST(K_BF16) DTF_CAST_DST(K_I32) DTF_CAST_DST(K_I64)
    DTF_CAST_DST(K_BOOL) DTF_CAST_DST(K_F64) DTF_CAST_DST(K_FP8)
#undef DTF_CAST_DST
    default: return DTF_ERR_UNSUPPORTED;
  }
  return DTF_OK;
}

bool aligned(const void* p, unsigned a) { return ((uintptr_t)p & (a - 1)) == 0; }

}  // namespace

// dst[r, 0:dst_row_bytes) = src[r, 0:src_row_bytes) then zeros; everything a multiple of 16 bytes
// and 16-byte aligned (DTF_ERR_UNSUPPORTED otherwise).
DTF_API int dtf_pad_rows(const void* src, void* dst, long rows, long src_row_bytes,
                         long dst_row_bytes, long src_pitch, long dst_pitch, hipStream_t stream) {
  if (rows <= 0 || dst_row_bytes <= 0) return DTF_OK;
  if (!src || !dst || src_row_bytes < 0 || dst_row_bytes < src_row_bytes ||
      src_pitch < src_row_bytes || dst_pitch < dst_row_bytes)
    return DTF_ERR_BAD_ARG;
  if (!aligned(src, 16) || !aligned(dst, 16) ||
      (src_row_bytes | dst_row_bytes | src_pitch | dst_pitch) % 16 != 0)
    return DTF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(grid_for(rows * (dst_row_bytes / 16), 1)), dim3(BLOCK),
                     0, stream, (const unsigned char*)src, (unsigned char*)dst, rows,
                     src_row_bytes / 16, dst_row_bytes / 16, src_pitch, dst_pitch);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dst [dst_rows, dst_ld] (dense) = src [rows, cols] (row pitch src_ld) padded with zeros
DTF_API int dtf_pad_2d_f32(const float* src, float* dst, long rows, long cols, long src_ld,
                           long dst_rows, long dst_ld, hipStream_t stream) {
  if (dst_rows <= 0 || dst_ld <= 0) return DTF_OK;
  if (!src || !dst || rows < 0 || cols < 0 || rows > dst_rows || cols > dst_ld || src_ld < cols)
    return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(pad_2d_kernel<float>, dim3(grid_for(dst_rows * dst_ld, 1)), dim3(BLOCK), 0,
                     stream, src, dst, rows, cols, src_ld, dst_rows, dst_ld);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// the same for 2-byte elements (bf16; an all-zero bit pattern is 0.0 there too)
DTF_API int dtf_pad_2d_b16(const void* src, void* dst, long rows, long cols, long src_ld,
                           long dst_rows, long dst_ld, hipStream_t stream) {
  if (dst_rows <= 0 || dst_ld <= 0) return DTF_OK;
  if (!src || !dst || rows < 0 || cols < 0 || rows > dst_rows || cols > dst_ld || src_ld < cols)
    return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(pad_2d_kernel<unsigned short>, dim3(grid_for(dst_rows * dst_ld, 1)),
                     dim3(BLOCK), 0, stream, (const unsigned short*)src, (unsigned short*)dst, rows,
                     cols, src_ld, dst_rows, dst_ld);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// ... and for 1-byte elements (fp8 e4m3: 0x00 is 0.0)
DTF_API int dtf_pad_2d_b8(const void* src, void* dst, long rows, long cols, long src_ld,
                          long dst_rows, long dst_ld, hipStream_t stream) {
  if (dst_rows <= 0 || dst_ld <= 0) return DTF_OK;
  if (!src || !dst || rows < 0 || cols < 0 || rows > dst_rows || cols > dst_ld || src_ld < cols)
    return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(pad_2d_kernel<unsigned char>, dim3(grid_for(dst_rows * dst_ld, 1)),
                     dim3(BLOCK), 0, stream, (const unsigned char*)src, (unsigned char*)dst, rows,
                     cols, src_ld, dst_rows, dst_ld);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// Copy a [rows, row_bytes] block between buffers with row pitches src_pitch / dst_pitch (bytes).
DTF_API int dtf_copy_2d(const void* src, void* dst, long rows, long row_bytes, long src_pitch,
                        long dst_pitch, hipStream_t stream) {
  if (rows <= 0 || row_bytes <= 0) return DTF_OK;
  if (!src || !dst || src_pitch < row_bytes || dst_pitch < row_bytes) return DTF_ERR_BAD_ARG;
  const unsigned char* s = (const unsigned char*)src;
  unsigned char* d = (unsigned char*)dst;
  const long all = row_bytes | src_pitch | dst_pitch;
  if (aligned(s, 16) && aligned(d, 16) && all % 16 == 0) {
    const long units = row_bytes / 16;
    hipLaunchKernelGGL(copy_2d_kernel<u32x4>, dim3(grid_for(rows * units, 1)), dim3(BLOCK), 0,
                       stream, s, d, rows, units, src_pitch, dst_pitch);
  } else if (aligned(s, 4) && aligned(d, 4) && all % 4 == 0) {
    const long units = row_bytes / 4;
    hipLaunchKernelGGL(copy_2d_kernel<unsigned int>, dim3(grid_for(rows * units, 1)), dim3(BLOCK),
                       0, stream, s, d, rows, units, src_pitch, dst_pitch);
  } else if (aligned(s, 2) && aligned(d, 2) && all % 2 == 0) {
    const long units = row_bytes / 2;
    hipLaunchKernelGGL(copy_2d_kernel<unsigned short>, dim3(grid_for(rows * units, 1)),
                       dim3(BLOCK), 0, stream, s, d, rows, units, src_pitch, dst_pitch);
  } else {
    hipLaunchKernelGGL(copy_2d_kernel<unsigned char>, dim3(grid_for(rows * row_bytes, 1)),
                       dim3(BLOCK), 0, stream, s, d, rows, row_bytes, src_pitch, dst_pitch);
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// var[i] += sign * delta[i % dn] in place (AssignAdd: sign = 1, AssignSub: sign = -1)
DTF_API int dtf_assign_add_f32(float* var, const float* delta, long n, long dn, float sign,
                               hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!var || !delta || dn <= 0 || dn > n || n % dn) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(assign_add_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, var, delta,
                     n, dn, sign);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_one_hot_f32(const void* idx, int wide, float* out, long rows, long depth, float on,
                            float off, hipStream_t stream) {
  if (rows <= 0 || depth <= 0) return DTF_OK;
  if (!idx || !out) return DTF_ERR_BAD_ARG;
  const dim3 grid(grid_for(rows * depth, 1)), block(BLOCK);
  if (wide) hipLaunchKernelGGL(one_hot_kernel<true>, grid, block, 0, stream, idx, out, rows, depth, on, off);
  else hipLaunchKernelGGL(one_hot_kernel<false>, grid, block, 0, stream, idx, out, rows, depth, on, off);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_sign_f32(const float* x, float* out, long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(sign_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, x, out, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// out[outer, mid, inner] = g[outer, inner] * scale
DTF_API int dtf_broadcast_mid_f32(const float* g, float* out, long outer, long mid, long inner,
                                  float scale, hipStream_t stream) {
  if (outer <= 0 || mid <= 0 || inner <= 0) return DTF_OK;
  if (!g || !out) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(broadcast_mid_kernel, dim3(grid_for(outer * mid * inner, 1)), dim3(BLOCK), 0,
                     stream, g, out, outer, mid, inner, scale);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_log_softmax_grad_f32(const float* y, const float* dy, float* dx, long rows,
                                     long cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return DTF_OK;
  if (!y || !dy || !dx) return DTF_ERR_BAD_ARG;
  long blocks = (rows + BLOCK / 64 - 1) / (BLOCK / 64);
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  hipLaunchKernelGGL(log_softmax_grad_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, stream, y, dy,
                     dx, rows, cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API long dtf_absmax_workspace(long n) { return grid_for(n, 8); }

// out[0] = max|x|, out[1] = target / max|x| (1 if the tensor is all zero), out[2] = 1 / out[1];
// src_kind 0 = fp32, 1 = bf16; workspace: dtf_absmax_workspace(n) floats
DTF_API int dtf_absmax_scale(const void* x, int src_kind, long n, float target, float* workspace,
                             float* out, hipStream_t stream) {
  if (!out || !workspace || (n > 0 && !x) || src_kind < 0 || src_kind > 1) return DTF_ERR_BAD_ARG;
  const int blocks = n > 0 ? grid_for(n, 8) : 1;
  if (src_kind == 1)
    hipLaunchKernelGGL(absmax_partial_kernel<true>, dim3(blocks), dim3(BLOCK), 0, stream, x,
                       n > 0 ? n : 0, workspace);
  else
    hipLaunchKernelGGL(absmax_partial_kernel<false>, dim3(blocks), dim3(BLOCK), 0, stream, x,
                       n > 0 ? n : 0, workspace);
  hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(64), 0, stream, workspace, blocks, target,
                     out);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_cast_to_fp8_dev_scale(const void* x, int src_kind, void* out, long n,
                                      const float* scale, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || !scale || !aligned(x, 16) || !aligned(out, 4) || src_kind < 0 || src_kind > 1)
    return DTF_ERR_BAD_ARG;
  if (src_kind == 1)
    hipLaunchKernelGGL(cast_to_fp8_dev_scale_kernel<true>, dim3(grid_for(n, 4)), dim3(BLOCK), 0,
                       stream, x, (unsigned char*)out, n, scale);
  else
    hipLaunchKernelGGL(cast_to_fp8_dev_scale_kernel<false>, dim3(grid_for(n, 4)), dim3(BLOCK), 0,
                       stream, x, (unsigned char*)out, n, scale);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_truncated_normal_f32(float* out, long n, unsigned int seed, unsigned int op_seed,
                                     float mean, float stddev, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!out) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(truncated_normal_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, out, n,
                     seed, op_seed, mean, stddev);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dst[i] = (dst type) src[i]; kinds: 0 f32, 1 bf16, 2 i32, 3 i64, 4 bool, 5 f64, 6 fp8 e4m3
DTF_API int dtf_cast_generic(const void* src, int src_kind, void* dst, int dst_kind, long n,
                             hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!src || !dst) return DTF_ERR_BAD_ARG;
  int rc;
  switch (src_kind) {
    case K_F32: rc = cast_dispatch_dst<K_F32>(src, dst, dst_kind, n, stream); break;
    case K_BF16: rc = cast_dispatch_dst<K_BF16>(src, dst, dst_kind, n, stream); break;
    case K_I32: rc = cast_dispatch_dst<K_I32>(src, dst, dst_kind, n, stream); break;
    case K_I64: rc = cast_dispatch_dst<K_I64>(src, dst, dst_kind, n, stream); break;
    case K_BOOL: rc = cast_dispatch_dst<K_BOOL>(src, dst, dst_kind, n, stream); break;
    case K_F64: rc = cast_dispatch_dst<K_F64>(src, dst, dst_kind, n, stream); break;
    case K_FP8: rc = cast_dispatch_dst<K_FP8>(src, dst, dst_kind, n, stream); break;
    default: return DTF_ERR_UNSUPPORTED;
  }
  if (rc != DTF_OK) return rc;
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// ---- MX block-scaled fp8 quantisation ------------------------------------------------------------
// x[R, K] (fp32 or bf16, row pitch ld) -> q[R, K] OCP e4m3 (row pitch ldq) + one E8M0 scale per 32
// consecutive k of a row: scale = 2^e with e the smallest exponent for which the block's largest
// magnitude still fits e4m3 (|x| / 2^e <= 448), so nothing saturates; an all-zero block gets
// scale 1. Eight lanes share a block (four elements each, one 16-byte load per lane for fp32).
//
// Scales are written in the order the block-scaled GEMM (gemm_bf16.hip, MX) reads them. With r the
// row inside its 256-row tile, kb the index of the 32-block, kt = kb / 4 its K-tile of 128 k and
// lane = 16 (kb % 4) + r % 16:
//   role 0 (the GEMM's A operand): per PAIR of K-tiles and row tile 2 KiB;
//       byte offset 1024 ((r / 64) % 2) + 16 lane + 8 (kt % 2) + 4 (r / 128) + (r / 16) % 4
//   role 1 (its Bt operand): per FOUR K-tiles and row tile 4 KiB;
//       byte offset 1024 ((r / 32) % 4) + 16 lane + 4 (kt % 4) + 2 (r / 128) + (r / 16) % 2 `scales_rm` (optional)
// receives the same bytes in plain row-major [R, K / 32] order.
namespace {

template <bool BF16_IN>
__global__ __launch_bounds__(256) void quantize_mx_kernel(const void* __restrict__ x_, long R, long K,
                                                          long ld, unsigned char* __restrict__ q,
                                                          long ldq, unsigned char* __restrict__ sc,
                                                          unsigned char* __restrict__ sc_rm,
                                                          int role, long row_tiles) {
  const long kblocks = K >> 5;
  const long gb = (long)blockIdx.x * 32 + (threadIdx.x >> 3);     // (row, 32-block) of this octet
  const int l8 = threadIdx.x & 7;
  const bool in = gb < R * kblocks;
  const long row = in ? gb / kblocks : 0, kb = in ? gb % kblocks : 0;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (in) {
    if (BF16_IN) {
      const unsigned short* p = reinterpret_cast<const unsigned short*>(x_) + row * ld + kb * 32 + l8 * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = dtf_bf16_to_f32(p[j]);
    } else {
      const float* p = reinterpret_cast<const float*>(x_) + row * ld + kb * 32 + l8 * 4;
      if ((((uintptr_t)p) & 15) == 0) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = p[j];
      }
    }
  }
  float amax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
  // e8 - 127 = ceil(log2(amax / 448)): exponent field of amax / 448, plus one unless the
  // mantissa is zero; clamped to the E8M0 range that has an fp32 reciprocal
  int e8 = 127;
  if (amax > 0.f && amax < INFINITY) {
    const unsigned bits = __float_as_uint(amax * (1.0f / 448.0f));
    e8 = (int)((bits >> 23) & 255u) + ((bits & 0x7fffffu) ? 1 : 0);
    e8 = e8 < 1 ? 1 : (e8 > 253 ? 253 : e8);
  }
  const float inv = __uint_as_float((unsigned)(254 - e8) << 23);      // 2^-(e8 - 127)
  if (!in) return;
  unsigned packed = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) packed |= (unsigned)dtf_f32_to_e4m3(v[j] * inv) << (8 * j);
  *reinterpret_cast<unsigned*>(q + row * ldq + kb * 32 + l8 * 4) = packed;
  if (l8 == 0) {
    const long kt = kb >> 2, tile = row >> 8;
    const int r = (int)(row & 255), ln = 16 * (int)(kb & 3) + (r & 15);
    long off;
    if (role == 0)
      off = ((kt >> 1) * row_tiles + tile) * 2048 + 1024 * ((r >> 6) & 1) + 16 * ln +
            8 * (int)(kt & 1) + 4 * (r >> 7) + ((r >> 4) & 3);
    else
      off = ((kt >> 2) * row_tiles + tile) * 4096 + 1024 * ((r >> 5) & 3) + 16 * ln +
            4 * (int)(kt & 3) + 2 * (r >> 7) + ((r >> 4) & 1);
    sc[off] = (unsigned char)e8;
    if (sc_rm) sc_rm[row * kblocks + kb] = (unsigned char)e8;
  }
}

}  // namespace

// bytes of the GEMM-ordered scale buffer for an [R, K] operand (R rounded up to whole 256-row
// tiles, K % 128 == 0)
DTF_API long dtf_mx_scale_bytes(long R, long K) {
  if (R <= 0 || K <= 0 || K % 128) return 0;
  return ((K / 128 + 3) / 4 * 4) * ((R + 255) / 256) * 1024;    // whole groups of four K-tiles
}

// src_kind 0: fp32, 1: bf16. q: [R, K] bytes with row pitch ldq (ldq % 4 == 0). scales:
// dtf_mx_scale_bytes(R, K) bytes, 4-byte aligned (rows past R in the last tile keep what the
// buffer held: the caller zero-fills or ignores them). scales_rm may be null.
DTF_API int dtf_quantize_mx_fp8(const void* x, int src_kind, long R, long K, long ld, void* q,
                                long ldq, void* scales, void* scales_rm, int role,
                                hipStream_t stream) {
  if (!x || !q || !scales || R <= 0 || K <= 0 || ld < K || ldq < K || (role != 0 && role != 1) ||
      (src_kind != 0 && src_kind != 1))
    return DTF_ERR_BAD_ARG;
  if (K % 128 || ldq % 4 || ((uintptr_t)q & 3) || (src_kind == 1 && (ld % 2)))
    return DTF_ERR_UNSUPPORTED;
  const long blocks = (R * (K / 32) + 31) / 32;
  if (blocks > 0x7fffffffL) return DTF_ERR_UNSUPPORTED;
  const long row_tiles = (R + 255) / 256;
  if (src_kind == 1)
    hipLaunchKernelGGL(quantize_mx_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, x, R,
                       K, ld, (unsigned char*)q, ldq, (unsigned char*)scales,
                       (unsigned char*)scales_rm, role, row_tiles);
  else
    hipLaunchKernelGGL(quantize_mx_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, x,
                       R, K, ld, (unsigned char*)q, ldq, (unsigned char*)scales,
                       (unsigned char*)scales_rm, role, row_tiles);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// ---------------------------------------------------------------------------------------------
// Multi-tensor copy: up to 16 (source, destination, bytes) triples in ONE launch (grid.y picks the
// pair). The gradient buckets of the overlapped all-reduce are packed with it -- every member of a
// bucket into its span of the flat bucket buffer -- instead of one copy launch per gradient.
// 16 bytes per lane where both pointers and the length allow, bytes otherwise.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int MAX_COPY_PAIRS = 16;

struct CopyPairs {
  const unsigned char* src[MAX_COPY_PAIRS];
  unsigned char* dst[MAX_COPY_PAIRS];
  long bytes[MAX_COPY_PAIRS];
};

__global__ __launch_bounds__(256) void multi_copy_kernel(CopyPairs args) {
  const unsigned char* src = args.src[blockIdx.y];
  unsigned char* dst = args.dst[blockIdx.y];
  const long n = args.bytes[blockIdx.y];
  const long start = blockIdx.x * 256L + threadIdx.x, step = (long)gridDim.x * 256L;
  long done = 0;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const long n16 = n >> 4;
    const u32x4* s4 = reinterpret_cast<const u32x4*>(src);
    u32x4* d4 = reinterpret_cast<u32x4*>(dst);
    for (long i = start; i < n16; i += step) d4[i] = s4[i];
    done = n16 << 4;
  }
  for (long i = done + start; i < n; i += step) dst[i] = src[i];
}

}  // namespace

DTF_API int dtf_multi_copy(int count, const void* const* src, void* const* dst, const long* bytes,
                           hipStream_t stream) {
  if (count < 0 || count > MAX_COPY_PAIRS || (count > 0 && (!src || !dst || !bytes)))
    return DTF_ERR_BAD_ARG;
  if (count == 0) return DTF_OK;
  CopyPairs args = {};
  long max_bytes = 1;
  for (int i = 0; i < count; ++i) {
    if (bytes[i] < 0 || (bytes[i] > 0 && (!src[i] || !dst[i]))) return DTF_ERR_BAD_ARG;
    args.src[i] = static_cast<const unsigned char*>(src[i]);
    args.dst[i] = static_cast<unsigned char*>(dst[i]);
    args.bytes[i] = bytes[i];
    if (bytes[i] > max_bytes) max_bytes = bytes[i];
  }
  long blocks = (max_bytes + 256L * 16 * 4 - 1) / (256L * 16 * 4);     // 4 vectors per lane
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0,
                     stream, args);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
