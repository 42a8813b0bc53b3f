// Vectorised elementwise kernels (memory-bound: 16 B per lane, grid-stride, <= 2048 blocks).
//
// Covers the pointwise ops of the reference graphs -- add with bias broadcast, mul, neg, log,
// sigmoid, cast, equal (tf_distributed.py:62-81) -- their gradient pointwise pieces, and the
// add/scale primitive (axpby) used on sharded-GEMM partial sums. Broadcasting is expressed as
// "operand index = i % operand_numel", which covers same-shape, trailing-vector ([R,C] op [C])
// and scalar operands without materialising the broadcast.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_BLOCKS = 2048;

inline int grid_for(long n, int per_thread) {
  long blocks = (n + (long)BLOCK * per_thread - 1) / ((long)BLOCK * per_thread);
  if (blocks < 1) blocks = 1;
  return (int)(blocks > MAX_BLOCKS ? MAX_BLOCKS : blocks);
}

enum BinaryOp { B_ADD = 0, B_SUB, B_MUL, B_DIV, B_MAX, B_MIN, B_POW, B_SIGMOID_GRAD, B_TANH_GRAD,
                B_RELU_GRAD };
enum CompareOp { C_EQ = 0, C_NE, C_LT, C_GT, C_LE, C_GE };
enum UnaryOp { U_NEG = 0, U_LOG, U_EXP, U_SQRT, U_SQUARE, U_RECIP, U_ABS, U_SIGMOID, U_TANH,
               U_RELU, U_FLOOR, U_COS, U_SIN, U_SOFTPLUS };

template <int OP>
__device__ __forceinline__ float bin(float a, float b) {
  if (OP == B_ADD) return a + b;
  if (OP == B_SUB) return a - b;
  if (OP == B_MUL) return a * b;
  if (OP == B_DIV) return a / b;
  if (OP == B_MAX) return fmaxf(a, b);
  if (OP == B_MIN) return fminf(a, b);
  if (OP == B_POW) return powf(a, b);
  if (OP == B_SIGMOID_GRAD) return b * a * (1.f - a);   // (y, dy)
  if (OP == B_TANH_GRAD) return b * (1.f - a * a);      // (y, dy)
  return a > 0.f ? b : 0.f;                             // relu grad (y, dy)
}

template <int OP>
__device__ __forceinline__ float una(float x) {
  if (OP == U_NEG) return -x;
  if (OP == U_LOG) return logf(x);
  if (OP == U_EXP) return expf(x);
  if (OP == U_SQRT) return sqrtf(x);
  if (OP == U_SQUARE) return x * x;
  if (OP == U_RECIP) return 1.f / x;
  if (OP == U_ABS) return fabsf(x);
  if (OP == U_SIGMOID) return 1.f / (1.f + expf(-x));
  if (OP == U_TANH) return tanhf(x);
  if (OP == U_FLOOR) return floorf(x);
  if (OP == U_COS) return cosf(x);
  if (OP == U_SIN) return sinf(x);
  if (OP == U_SOFTPLUS) return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));   // no overflow
  return x > 0.f ? x : 0.f;
}

// same-shape fast path: 16 B per lane
template <int OP>
__global__ __launch_bounds__(BLOCK) void binary_vec_kernel(const f32x4* __restrict__ a,
                                                           const f32x4* __restrict__ b,
                                                           f32x4* __restrict__ out, long n4) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n4; i += (long)gridDim.x * BLOCK) {
    f32x4 x = a[i], y = b[i], r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = bin<OP>(x[j], y[j]);
    out[i] = r;
  }
}

template <int OP>
__global__ __launch_bounds__(BLOCK) void binary_bcast_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ b,
                                                             float* __restrict__ out, long n,
                                                             long an, long bn) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    out[i] = bin<OP>(a[an == n ? i : i % an], b[bn == n ? i : i % bn]);
}

template <int OP>
__global__ __launch_bounds__(BLOCK) void unary_kernel(const float* __restrict__ x,
                                                      float* __restrict__ out, long n) {
  const long n4 = n >> 2;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  for (long i = start; i < n4; i += step) {
    f32x4 v = x4[i], r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = una<OP>(v[j]);
    o4[i] = r;
  }
  for (long i = (n4 << 2) + start; i < n; i += step) out[i] = una<OP>(x[i]);
}

template <int OP>
__global__ __launch_bounds__(BLOCK) void compare_kernel(const float* __restrict__ a,
                                                        const float* __restrict__ b,
                                                        unsigned char* __restrict__ out, long n,
                                                        long an, long bn) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const float x = a[an == n ? i : i % an], y = b[bn == n ? i : i % bn];
    out[i] = OP == C_EQ ? x == y : OP == C_NE ? x != y : OP == C_LT ? x < y : OP == C_GT ? x > y
             : OP == C_LE ? x <= y : x >= y;
  }
}

// out[i] = cond[i] ? a[i] : b[i]; cond / a / b broadcast by period (a scalar operand has period 1)
__global__ __launch_bounds__(BLOCK) void select_kernel(const unsigned char* __restrict__ cond,
                                                       const float* __restrict__ a,
                                                       const float* __restrict__ b,
                                                       float* __restrict__ out, long n, long cn,
                                                       long an, long bn) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    out[i] = cond[cn == n ? i : i % cn] ? a[an == n ? i : i % an] : b[bn == n ? i : i % bn];
}

__global__ __launch_bounds__(BLOCK) void compare_i64_eq_kernel(const long* __restrict__ a,
                                                               const long* __restrict__ b,
                                                               unsigned char* __restrict__ out,
                                                               long n) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    out[i] = a[i] == b[i];
}

// out = alpha * x + beta * y (y may be null when beta == 0)
__global__ __launch_bounds__(BLOCK) void axpby_f32_kernel(float alpha, const float* __restrict__ x,
                                                          float beta, const float* __restrict__ y,
                                                          float* __restrict__ out, long n) {
  const long n4 = n >> 2;
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const f32x4* y4 = reinterpret_cast<const f32x4*>(y);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  for (long i = start; i < n4; i += step) {
    f32x4 r = x4[i] * alpha;
    if (y) r += y4[i] * beta;
    o4[i] = r;
  }
  for (long i = (n4 << 2) + start; i < n; i += step)
    out[i] = alpha * x[i] + (y ? beta * y[i] : 0.f);
}

// bf16 axpby: 8 elements (16 B) per lane, fp32 math, round-to-nearest-even on the way out
__global__ __launch_bounds__(BLOCK) void axpby_bf16_kernel(float alpha,
                                                           const unsigned short* __restrict__ x,
                                                           float beta,
                                                           const unsigned short* __restrict__ y,
                                                           unsigned short* __restrict__ out,
                                                           long n) {
  const long n8 = n >> 3;
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  const bf16x8* x8 = reinterpret_cast<const bf16x8*>(x);
  const bf16x8* y8 = reinterpret_cast<const bf16x8*>(y);
  bf16x8* o8 = reinterpret_cast<bf16x8*>(out);
  for (long i = start; i < n8; i += step) {
    bf16x8 xv = x8[i], yv = y ? y8[i] : xv, r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = alpha * dtf_bf16_to_f32((unsigned short)xv[j]);
      if (y) v += beta * dtf_bf16_to_f32((unsigned short)yv[j]);
      r[j] = (short)dtf_f32_to_bf16(v);
    }
    o8[i] = r;
  }
  for (long i = (n8 << 3) + start; i < n; i += step) {
    float v = alpha * dtf_bf16_to_f32(x[i]);
    if (y) v += beta * dtf_bf16_to_f32(y[i]);
    out[i] = dtf_f32_to_bf16(v);
  }
}

__global__ __launch_bounds__(BLOCK) void cast_f32_bf16_kernel(const float* __restrict__ x,
                                                              unsigned short* __restrict__ out,
                                                              long n) {
  const long n4 = n >> 2;
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  bf16x4* o4 = reinterpret_cast<bf16x4*>(out);
  for (long i = start; i < n4; i += step) {
    f32x4 v = x4[i];
    bf16x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = (short)dtf_f32_to_bf16(v[j]);
    o4[i] = r;
  }
  for (long i = (n4 << 2) + start; i < n; i += step) out[i] = dtf_f32_to_bf16(x[i]);
}

__global__ __launch_bounds__(BLOCK) void cast_bf16_f32_kernel(const unsigned short* __restrict__ x,
                                                              float* __restrict__ out, long n) {
  const long n4 = n >> 2;
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  const bf16x4* x4 = reinterpret_cast<const bf16x4*>(x);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  for (long i = start; i < n4; i += step) {
    bf16x4 v = x4[i];
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = dtf_bf16_to_f32((unsigned short)v[j]);
    o4[i] = r;
  }
  for (long i = (n4 << 2) + start; i < n; i += step) out[i] = dtf_bf16_to_f32(x[i]);
}

// quantise to OCP fp8 e4m3: out = e4m3(clamp(x * scale, +-448)); 4 elements -> one 32-bit store
// through v_cvt_pk_fp8_f32 (round-to-nearest-even). SRC_BF16 selects the source type.
template <bool SRC_BF16>
__global__ __launch_bounds__(BLOCK) void cast_to_fp8_kernel(const void* __restrict__ x_,
                                                            unsigned char* __restrict__ out,
                                                            long n, float scale) {
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
    out[i] = dtf_f32_to_e4m3(f * scale);
  }
}

// dequantise: out = f32/bf16(e4m3(x)) * scale
template <bool DST_BF16>
__global__ __launch_bounds__(BLOCK) void cast_from_fp8_kernel(const unsigned char* __restrict__ x,
                                                              void* __restrict__ out_, long n,
                                                              float scale) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const float f = dtf_e4m3_to_f32(x[i]) * scale;
    if (DST_BF16) reinterpret_cast<unsigned short*>(out_)[i] = dtf_f32_to_bf16(f);
    else reinterpret_cast<float*>(out_)[i] = f;
  }
}

__global__ __launch_bounds__(BLOCK) void cast_u8_f32_kernel(const unsigned char* __restrict__ x,
                                                            float* __restrict__ out, long n) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    out[i] = x[i] ? 1.f : 0.f;
}

__global__ __launch_bounds__(BLOCK) void fill_f32_kernel(float* __restrict__ out, float value,
                                                         long n) {
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    out[i] = value;
}

// out[r, c] = x[r, c] * s[r]
__global__ __launch_bounds__(BLOCK) void row_scale_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ s,
                                                          float* __restrict__ out, long rows,
                                                          long cols) {
  const long n = rows * cols;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    out[i] = x[i] * s[i / cols];
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

#define DTF_BIN_CASE(OP)                                                                        \
  case OP:                                                                                      \
    if (vec)                                                                                    \
      hipLaunchKernelGGL((binary_vec_kernel<OP>), dim3(grid_for(n / 4, 1)), dim3(BLOCK), 0,     \
                         stream, (const f32x4*)a, (const f32x4*)b, (f32x4*)out, n / 4);         \
    else                                                                                        \
      hipLaunchKernelGGL((binary_bcast_kernel<OP>), dim3(grid_for(n, 1)), dim3(BLOCK), 0,       \
                         stream, a, b, out, n, an, bn);                                         \
    break;

// out[i] = op(a[i % an], b[i % bn]) for i < n
DTF_API int dtf_binary_f32(int op, const float* a, const float* b, float* out, long n, long an,
                           long bn, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!a || !b || !out || an <= 0 || bn <= 0) return DTF_ERR_BAD_ARG;
  const bool vec = an == n && bn == n && (n % 4 == 0) && aligned16(a) && aligned16(b) &&
                   aligned16(out);
  switch (op) {
    DTF_BIN_CASE(B_ADD) DTF_BIN_CASE(B_SUB) DTF_BIN_CASE(B_MUL) DTF_BIN_CASE(B_DIV)
    DTF_BIN_CASE(B_MAX) DTF_BIN_CASE(B_MIN) DTF_BIN_CASE(B_POW) DTF_BIN_CASE(B_SIGMOID_GRAD)
    DTF_BIN_CASE(B_TANH_GRAD) DTF_BIN_CASE(B_RELU_GRAD)
    default: return DTF_ERR_UNSUPPORTED;
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

#define DTF_UN_CASE(OP)                                                                    \
  case OP:                                                                                 \
    hipLaunchKernelGGL((unary_kernel<OP>), dim3(grid_for(n, 4)), dim3(BLOCK), 0, stream, x, \
                       out, n);                                                            \
    break;

DTF_API int dtf_unary_f32(int op, const float* x, float* out, long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || !aligned16(x) || !aligned16(out)) return DTF_ERR_BAD_ARG;
  switch (op) {
    DTF_UN_CASE(U_NEG) DTF_UN_CASE(U_LOG) DTF_UN_CASE(U_EXP) DTF_UN_CASE(U_SQRT)
    DTF_UN_CASE(U_SQUARE) DTF_UN_CASE(U_RECIP) DTF_UN_CASE(U_ABS) DTF_UN_CASE(U_SIGMOID)
    DTF_UN_CASE(U_TANH) DTF_UN_CASE(U_RELU) DTF_UN_CASE(U_FLOOR)
    DTF_UN_CASE(U_COS) DTF_UN_CASE(U_SIN) DTF_UN_CASE(U_SOFTPLUS)
    default: return DTF_ERR_UNSUPPORTED;
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

#define DTF_CMP_CASE(OP)                                                                     \
  case OP:                                                                                   \
    hipLaunchKernelGGL((compare_kernel<OP>), dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, a, \
                       b, out, n, an, bn);                                                   \
    break;

DTF_API int dtf_compare_f32(int op, const float* a, const float* b, unsigned char* out, long n,
                            long an, long bn, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!a || !b || !out || an <= 0 || bn <= 0) return DTF_ERR_BAD_ARG;
  switch (op) {
    DTF_CMP_CASE(C_EQ) DTF_CMP_CASE(C_NE) DTF_CMP_CASE(C_LT) DTF_CMP_CASE(C_GT)
    DTF_CMP_CASE(C_LE) DTF_CMP_CASE(C_GE)
    default: return DTF_ERR_UNSUPPORTED;
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_select_f32(const unsigned char* cond, const float* a, const float* b, float* out,
                           long n, long cn, long an, long bn, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!cond || !a || !b || !out || cn <= 0 || an <= 0 || bn <= 0) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(select_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, cond, a, b, out,
                     n, cn, an, bn);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_equal_i64(const long* a, const long* b, unsigned char* out, long n,
                          hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!a || !b || !out) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(compare_i64_eq_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, a, b,
                     out, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_axpby_f32(float alpha, const float* x, float beta, const float* y, float* out,
                          long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || !aligned16(x) || !aligned16(out) || (y && !aligned16(y)))
    return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(axpby_f32_kernel, dim3(grid_for(n, 4)), dim3(BLOCK), 0, stream, alpha, x,
                     beta, y, out, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_axpby_bf16(float alpha, const void* x, float beta, const void* y, void* out,
                           long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || !aligned16(x) || !aligned16(out) || (y && !aligned16(y)))
    return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(axpby_bf16_kernel, dim3(grid_for(n, 8)), dim3(BLOCK), 0, stream, alpha,
                     (const unsigned short*)x, beta, (const unsigned short*)y,
                     (unsigned short*)out, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_cast_f32_to_bf16(const float* x, void* out, long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || !aligned16(x) || ((uintptr_t)out & 7)) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 4)), dim3(BLOCK), 0, stream, x,
                     (unsigned short*)out, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_cast_bf16_to_f32(const void* x, float* out, long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || ((uintptr_t)x & 7) || !aligned16(out)) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 4)), dim3(BLOCK), 0, stream,
                     (const unsigned short*)x, out, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// out[c, r] = x[r, c] for any 1/2/4-byte element type: 64x64 tiles through LDS (row pitch 65
// elements, so the transposed read walks consecutive banks), coalesced on both sides.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ x,
                                                        T* __restrict__ out, long rows, long cols,
                                                        long ldx, long ldo) {
  __shared__ T tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
  const long r0 = blockIdx.y * 64L, c0 = blockIdx.x * 64L;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long r = r0 + ty + 4 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 4 * i][tx] = x[r * ldx + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const long c = c0 + ty + 4 * i, r = r0 + tx;
    if (c < cols && r < rows) out[c * ldo + r] = tile[tx][ty + 4 * i];
  }
}

// elem_size in {1, 2, 4}; x is [rows, cols] with row pitch ldx, out is [cols, rows] with ldo
DTF_API int dtf_transpose(const void* x, void* out, long rows, long cols, long ldx, long ldo,
                          int elem_size, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return DTF_OK;
  if (!x || !out || ldx < cols || ldo < rows) return DTF_ERR_BAD_ARG;
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64)), block(256);
  if (grid.y > 65535) return DTF_ERR_UNSUPPORTED;
  switch (elem_size) {
    case 1: hipLaunchKernelGGL(transpose_kernel<unsigned char>, grid, block, 0, stream,
                               (const unsigned char*)x, (unsigned char*)out, rows, cols, ldx, ldo);
      break;
    case 2: hipLaunchKernelGGL(transpose_kernel<unsigned short>, grid, block, 0, stream,
                               (const unsigned short*)x, (unsigned short*)out, rows, cols, ldx, ldo);
      break;
    case 4: hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, stream, (const float*)x,
                               (float*)out, rows, cols, ldx, ldo);
      break;
    default: return DTF_ERR_UNSUPPORTED;
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// src_kind / dst_kind: 0 = fp32, 1 = bf16
DTF_API int dtf_cast_to_fp8(const void* x, int src_kind, void* out, long n, float scale,
                            hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || !aligned16(x) || ((uintptr_t)out & 3) || src_kind < 0 || src_kind > 1)
    return DTF_ERR_BAD_ARG;
  if (src_kind == 1)
    hipLaunchKernelGGL(cast_to_fp8_kernel<true>, dim3(grid_for(n, 4)), dim3(BLOCK), 0, stream, x,
                       (unsigned char*)out, n, scale);
  else
    hipLaunchKernelGGL(cast_to_fp8_kernel<false>, dim3(grid_for(n, 4)), dim3(BLOCK), 0, stream, x,
                       (unsigned char*)out, n, scale);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_cast_from_fp8(const void* x, void* out, int dst_kind, long n, float scale,
                              hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || dst_kind < 0 || dst_kind > 1) return DTF_ERR_BAD_ARG;
  if (dst_kind == 1)
    hipLaunchKernelGGL(cast_from_fp8_kernel<true>, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream,
                       (const unsigned char*)x, out, n, scale);
  else
    hipLaunchKernelGGL(cast_from_fp8_kernel<false>, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream,
                       (const unsigned char*)x, out, n, scale);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_cast_bool_to_f32(const unsigned char* x, float* out, long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(cast_u8_f32_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, x, out, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_fill_f32(float* out, float value, long n, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!out) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for(n, 1)), dim3(BLOCK), 0, stream, out, value, n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_row_scale_f32(const float* x, const float* s, float* out, long rows, long cols,
                              hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return DTF_OK;
  if (!x || !s || !out) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(row_scale_kernel, dim3(grid_for(rows * cols, 1)), dim3(BLOCK), 0, stream, x,
                     s, out, rows, cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
