// Reductions and the fused MLP-gradient column reductions.
//
// * row reductions ([R,C] -> [R]: sum / mean / max, argmax): one wave per row, butterfly
//   shuffles across the 64 lanes (tf.reduce_sum(axis=[1]), tf.argmax(., 1) --
//   tf_distributed.py:70,80);
// * full reductions ([n] -> scalar, tf.reduce_mean -- tf_distributed.py:70,81): blocks write
//   partials to a caller-provided workspace, the last kernel folds them in a fixed order, so
//   the result is bitwise reproducible run to run (no float atomics);
// * column reductions ([R,C] -> [C]) with an elementwise prologue: plain sum (bias gradient),
//   sigmoid backward (dz = da*a*(1-a), db = colsum dz) and softmax-cross-entropy backward
//   (dz = (y*sum(labels) - labels) * dloss/R, db = colsum dz). A workgroup owns 64 columns and
//   walks the rows with 4 row-groups of 64 lanes, so no cross-workgroup combine is needed.
#include "common.h"

namespace {

constexpr int BLOCK = 256;

enum RowOp { R_SUM = 0, R_MEAN = 1, R_MAX = 2 };

template <int OP>
__global__ __launch_bounds__(BLOCK) void reduce_rows_kernel(const float* __restrict__ x,
                                                            float* __restrict__ out, long rows,
                                                            long cols) {
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)BLOCK + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * BLOCK) >> 6;
  for (long r = wave; r < rows; r += nwaves) {
    const float* row = x + r * cols;
    float acc = OP == R_MAX ? -INFINITY : 0.f;
    for (long c = lane; c < cols; c += 64) acc = OP == R_MAX ? fmaxf(acc, row[c]) : acc + row[c];
    acc = OP == R_MAX ? dtf_wave_max(acc) : dtf_wave_sum(acc);
    if (lane == 0) out[r] = OP == R_MEAN ? acc / (float)cols : acc;
  }
}

// first index of the row maximum, as int64 (ties -> lowest index, like numpy/torch)
__global__ __launch_bounds__(BLOCK) void argmax_rows_kernel(const float* __restrict__ x,
                                                            long* __restrict__ out, long rows,
                                                            long cols) {
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)BLOCK + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * BLOCK) >> 6;
  for (long r = wave; r < rows; r += nwaves) {
    const float* row = x + r * cols;
    float best = -INFINITY;
    long idx = cols;
    for (long c = lane; c < cols; c += 64) {
      const float v = row[c];
      if (v > best || (v == best && c < idx)) { best = v; idx = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const long oi = __shfl_xor(idx, off, 64);
      if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if (lane == 0) out[r] = idx;
  }
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = dtf_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < BLOCK / 64; ++w) t += sh[w];
  return t;   // valid on thread 0
}

// stage 1: partial[b] = sum of this block's grid-stride slice. 16-byte loads, four of them in
// flight per thread into independent accumulators (a single scalar-load chain per thread ran
// the 64 MiB mean of a 4096^2 product at 130 GB/s); the fold order is fixed by the launch
// shape, so the result is deterministic.
__global__ __launch_bounds__(BLOCK) void reduce_all_stage1(const float* __restrict__ x,
                                                           float* __restrict__ partial, long n) {
  __shared__ float sh[BLOCK / 64];
  const long tid = blockIdx.x * (long)BLOCK + threadIdx.x;
  const long nthreads = (long)gridDim.x * BLOCK;
  float acc = 0.f;
  const long head = (4 - (((uintptr_t)x >> 2) & 3)) & 3;      // elements before 16-byte alignment
  const long lead = head < n ? head : n;
  if (tid < lead) acc += x[tid];
  const f32x4* xv = reinterpret_cast<const f32x4*>(x + lead);
  const long nv = (n - lead) >> 2;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  long i = tid;
  for (; i + 3 * nthreads < nv; i += 4 * nthreads) {
    const f32x4 v0 = xv[i], v1 = xv[i + nthreads], v2 = xv[i + 2 * nthreads],
                v3 = xv[i + 3 * nthreads];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; i < nv; i += nthreads) a0 += xv[i];
  a0 += a1; a2 += a3; a0 += a2;
  acc += (a0[0] + a0[1]) + (a0[2] + a0[3]);
  const long tail0 = lead + (nv << 2);
  if (tail0 + tid < n) acc += x[tail0 + tid];                 // at most 3 elements
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// stage 2: out = scale * sum(partial[0..m))
__global__ __launch_bounds__(BLOCK) void reduce_all_stage2(const float* __restrict__ partial,
                                                           float* __restrict__ out, int m,
                                                           float scale) {
  __shared__ float sh[BLOCK / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < m; i += BLOCK) acc += partial[i];
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) out[0] = t * scale;
}

// Mean squared error, fused: stage 1 sums (a - b)^2 over this block's grid-stride slice (the
// difference and the square never exist in memory), reduce_all_stage2 folds the partials and
// scales by 1 / n. The fold order is fixed by the launch shape.
__global__ __launch_bounds__(BLOCK) void sq_diff_stage1(const float* __restrict__ a,
                                                        const float* __restrict__ b,
                                                        float* __restrict__ partial, long n) {
  __shared__ float sh[BLOCK / 64];
  const long nthreads = (long)gridDim.x * BLOCK;
  float acc = 0.f;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += nthreads) {
    const float d = a[i] - b[i];
    acc = __builtin_fmaf(d, d, acc);
  }
  const float t = block_sum(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// gradient of mean((a - b)^2) with respect to a: da = (2 g / n) (a - b), g a device scalar
__global__ __launch_bounds__(BLOCK) void mse_grad_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ b,
                                                         const float* __restrict__ g,
                                                         float* __restrict__ da, long n) {
  const float s = 2.f * g[0] / (float)n;
  for (long i = blockIdx.x * (long)BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK)
    da[i] = s * (a[i] - b[i]);
}

enum ColMode { COL_SUM = 0, COL_SIGMOID_GRAD = 1, COL_XENT_GRAD = 2 };

// Workgroup b owns columns [64b, 64b+64); thread (g = tid>>6, l = tid&63) walks rows g, g+G, ...
// with G = blockDim/64 row groups (16 for short matrices such as a 100-row batch, so a thread's
// serial chain of dependent-latency loads is a handful of rows, four of them in flight at once).
// MODE 0: p0 = x                      -> colsum[c] = sum_r x[r,c]
// MODE 1: p0 = da, p1 = a             -> dz = da*a*(1-a) written to dz_out, colsum of dz
// MODE 2: p0 = y, p1 = labels, p2 = dloss (scalar) -> dz = (y*rowsum(labels) - labels)*dloss/R
//         (cols <= 64: one workgroup sees whole rows, so the label row-sum is a wave reduction)
// Sums are folded in a fixed order: deterministic for a given launch shape.
constexpr int COLS_BLOCK = 1024;

template <int MODE>
__global__ __launch_bounds__(COLS_BLOCK) void reduce_cols_kernel(const float* __restrict__ p0,
                                                                 const float* __restrict__ p1,
                                                                 const float* __restrict__ p2,
                                                                 float* __restrict__ dz_out,
                                                                 float* __restrict__ colsum,
                                                                 long rows, long cols) {
  __shared__ float sh[COLS_BLOCK];
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int groups = blockDim.x >> 6;
  const long c = blockIdx.x * 64L + l;
  const bool live = c < cols;
  float scale = 1.f;
  if (MODE == COL_XENT_GRAD) scale = p2[0] / (float)rows;
  float acc = 0.f;
  if (MODE == COL_XENT_GRAD) {
    for (long r = g; r < rows; r += groups) {
      const float lab = live ? p1[r * cols + c] : 0.f;
      const float lsum = dtf_wave_sum(lab);
      if (live) {
        const float v = (p0[r * cols + c] * lsum - lab) * scale;
        dz_out[r * cols + c] = v;
        acc += v;
      }
    }
  } else if (live) {
    long r = g;
    for (; r + 3L * groups < rows; r += 4L * groups) {      // four independent rows in flight
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long idx = (r + (long)u * groups) * cols + c;
        if (MODE == COL_SUM) {
          v[u] = p0[idx];
        } else {
          const float a = p1[idx];
          v[u] = p0[idx] * a * (1.f - a);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == COL_SIGMOID_GRAD) dz_out[(r + (long)u * groups) * cols + c] = v[u];
        acc += v[u];
      }
    }
    for (; r < rows; r += groups) {
      const long idx = r * cols + c;
      float v;
      if (MODE == COL_SUM) {
        v = p0[idx];
      } else {
        const float a = p1[idx];
        v = p0[idx] * a * (1.f - a);
        dz_out[idx] = v;
      }
      acc += v;
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (g == 0 && live && colsum) {
    float t = 0.f;
    for (int k = 0; k < groups; ++k) t += sh[k * 64 + l];
    colsum[c] = t;
  }
}

// mean over rows of (argmax(y[r,:]) == argmax(labels[r,:])). Lane per row, 256 rows per
// workgroup; each workgroup leaves its hit count in partial[block], and a one-workgroup kernel
// adds the counts in block order (integers: exact) and divides by the row count. (One
// workgroup walking all rows took 48 us for the 10 000-image evaluation batch.)
constexpr int ACC_ROWS = 256;

__global__ __launch_bounds__(ACC_ROWS) void argmax_accuracy_kernel(const float* __restrict__ y,
                                                                   const float* __restrict__ labels,
                                                                   float* __restrict__ partial,
                                                                   long rows, long cols) {
  __shared__ float sh[ACC_ROWS / 64];
  const long r = (long)blockIdx.x * ACC_ROWS + threadIdx.x;
  float hit = 0.f;
  if (r < rows) {
    const float* yr = y + r * cols;
    const float* lr = labels + r * cols;
    long iy = 0, il = 0;
    float by = yr[0], bl = lr[0];
    for (long c = 1; c < cols; ++c) {
      const float yv = yr[c], lv = lr[c];
      if (yv > by) { by = yv; iy = c; }
      if (lv > bl) { bl = lv; il = c; }
    }
    hit = iy == il ? 1.f : 0.f;
  }
  hit = dtf_wave_sum(hit);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = hit;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < ACC_ROWS / 64; ++w) t += sh[w];
    partial[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void argmax_accuracy_finish_kernel(
    const float* __restrict__ partial, float* __restrict__ out, int blocks, long rows) {
  __shared__ float sh[256];
  float t = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 256) t += partial[b];
  sh[threadIdx.x] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int i = 0; i < 256; ++i) total += sh[i];
    out[0] = total / (float)rows;
  }
}

// threads of a reduce_cols workgroup: one 64-lane row group per 4 rows, 4..16 groups
int cols_threads(long rows) {
  long groups = (rows + 3) / 4;
  groups = groups < 4 ? 4 : groups > 16 ? 16 : groups;
  return (int)groups * 64;
}

int rows_grid(long rows) {
  long blocks = (rows + 3) / 4;   // 4 waves per block, one row per wave
  return (int)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks);
}

}  // namespace

DTF_API int dtf_reduce_rows_f32(int op, const float* x, float* out, long rows, long cols,
                                hipStream_t stream) {
  if (rows <= 0) return DTF_OK;
  if (!x || !out || cols <= 0) return DTF_ERR_BAD_ARG;
  dim3 grid(rows_grid(rows)), block(BLOCK);
  switch (op) {
    case R_SUM: hipLaunchKernelGGL((reduce_rows_kernel<R_SUM>), grid, block, 0, stream, x, out, rows, cols); break;
    case R_MEAN: hipLaunchKernelGGL((reduce_rows_kernel<R_MEAN>), grid, block, 0, stream, x, out, rows, cols); break;
    case R_MAX: hipLaunchKernelGGL((reduce_rows_kernel<R_MAX>), grid, block, 0, stream, x, out, rows, cols); break;
    default: return DTF_ERR_UNSUPPORTED;
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_argmax_rows_f32(const float* x, long* out, long rows, long cols,
                                hipStream_t stream) {
  if (rows <= 0) return DTF_OK;
  if (!x || !out || cols <= 0) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows_grid(rows)), dim3(BLOCK), 0, stream, x, out,
                     rows, cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// number of fp32 words of workspace dtf_reduce_all_f32 needs for n elements
DTF_API int dtf_reduce_all_workspace(long n) {
  long blocks = (n + BLOCK * 8L - 1) / (BLOCK * 8L);
  return (int)(blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks);
}

// out[0] = scale * sum(x[0..n)); deterministic two-stage fold through `workspace`
DTF_API int dtf_reduce_all_f32(const float* x, float* out, long n, float scale, float* workspace,
                               hipStream_t stream) {
  if (!x || !out || !workspace || n <= 0) return DTF_ERR_BAD_ARG;
  const int blocks = dtf_reduce_all_workspace(n);
  hipLaunchKernelGGL(reduce_all_stage1, dim3(blocks), dim3(BLOCK), 0, stream, x, workspace, n);
  hipLaunchKernelGGL(reduce_all_stage2, dim3(1), dim3(BLOCK), 0, stream, workspace, out, blocks,
                     scale);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// out[0] = mean((a - b)^2) over n elements; workspace: dtf_reduce_all_workspace(n) floats
DTF_API int dtf_mse_f32(const float* a, const float* b, float* out, long n, float* workspace,
                        hipStream_t stream) {
  if (!a || !b || !out || !workspace || n <= 0) return DTF_ERR_BAD_ARG;
  const int blocks = dtf_reduce_all_workspace(n);
  hipLaunchKernelGGL(sq_diff_stage1, dim3(blocks), dim3(BLOCK), 0, stream, a, b, workspace, n);
  hipLaunchKernelGGL(reduce_all_stage2, dim3(1), dim3(BLOCK), 0, stream, workspace, out, blocks,
                     1.f / (float)n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// da = (2 g[0] / n) (a - b)
DTF_API int dtf_mse_grad_f32(const float* a, const float* b, const float* g, float* da, long n,
                             hipStream_t stream) {
  if (!a || !b || !g || !da || n <= 0) return DTF_ERR_BAD_ARG;
  long blocks = (n + BLOCK * 4L - 1) / (BLOCK * 4L);
  blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(mse_grad_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, stream, a, b, g, da,
                     n);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_reduce_cols_f32(const float* x, float* colsum, long rows, long cols,
                                hipStream_t stream) {
  if (!x || !colsum || rows <= 0 || cols <= 0) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL((reduce_cols_kernel<COL_SUM>), dim3((cols + 63) / 64),
                     dim3(cols_threads(rows)), 0, stream, x, nullptr, nullptr, nullptr, colsum, rows,
                     cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dz = da * a * (1 - a); dbias[c] = sum_r dz[r, c] (dbias may be null)
DTF_API int dtf_sigmoid_grad_bias_f32(const float* da, const float* a, float* dz, float* dbias,
                                      long rows, long cols, hipStream_t stream) {
  if (!da || !a || !dz || rows <= 0 || cols <= 0) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL((reduce_cols_kernel<COL_SIGMOID_GRAD>), dim3((cols + 63) / 64),
                     dim3(cols_threads(rows)), 0, stream, da, a, nullptr, dz, dbias, rows, cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dz = (y * rowsum(labels) - labels) * dloss / rows; dbias[c] = sum_r dz[r, c] (may be null).
// Requires cols <= 64 (one workgroup sees whole rows).
DTF_API int dtf_softmax_xent_grad_f32(const float* y, const float* labels, const float* dloss,
                                      float* dz, float* dbias, long rows, long cols,
                                      hipStream_t stream) {
  if (!y || !labels || !dloss || !dz || rows <= 0 || cols <= 0) return DTF_ERR_BAD_ARG;
  if (cols > 64) return DTF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((reduce_cols_kernel<COL_XENT_GRAD>), dim3(1), dim3(cols_threads(rows)), 0,
                     stream, y, labels, dloss, dz, dbias, rows, cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// floats of workspace dtf_argmax_accuracy_f32 needs for `rows` rows
DTF_API long dtf_argmax_accuracy_workspace(long rows) { return (rows + ACC_ROWS - 1) / ACC_ROWS; }

DTF_API int dtf_argmax_accuracy_f32(const float* y, const float* labels, float* out,
                                    float* workspace, long rows, long cols, hipStream_t stream) {
  if (!y || !labels || !out || !workspace || rows <= 0 || cols <= 0) return DTF_ERR_BAD_ARG;
  const int blocks = (int)((rows + ACC_ROWS - 1) / ACC_ROWS);
  hipLaunchKernelGGL(argmax_accuracy_kernel, dim3(blocks), dim3(ACC_ROWS), 0, stream, y, labels,
                     workspace, rows, cols);
  hipLaunchKernelGGL(argmax_accuracy_finish_kernel, dim3(1), dim3(256), 0, stream, workspace, out,
                     blocks, rows);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
