// Row softmax and the fused softmax + cross-entropy forward.
//
// The reference computes y = softmax(z3) (tf_distributed.py:65) and then
// mean(-sum(y_ * log(y), 1)) (tf_distributed.py:70) as five separate ops. Here one kernel
// reads the logits once, keeps a row in the 64 lanes of a wave (max-subtracted, so it agrees
// with the reference's value wherever that is finite and stays finite where log(softmax)
// underflows), writes the softmax for the accuracy branch and the backward pass, and reduces
// the per-row losses to the batch mean through a fixed-order two-stage fold.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;

// log_out != nullptr -> also write log-softmax
__global__ __launch_bounds__(BLOCK) void softmax_rows_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y,
                                                             float* __restrict__ log_out,
                                                             long rows, long cols) {
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)BLOCK + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * BLOCK) >> 6;
  for (long r = wave; r < rows; r += nwaves) {
    const float* row = x + r * cols;
    float m = -INFINITY;
    for (long c = lane; c < cols; c += 64) m = fmaxf(m, row[c]);
    m = dtf_wave_max(m);
    float s = 0.f;
    for (long c = lane; c < cols; c += 64) s += expf(row[c] - m);
    s = dtf_wave_sum(s);
    const float inv = 1.f / s, ls = logf(s);
    for (long c = lane; c < cols; c += 64) {
      const float d = row[c] - m;
      if (y) y[r * cols + c] = expf(d) * inv;
      if (log_out) log_out[r * cols + c] = d - ls;
    }
  }
}

// per-row loss = -sum(labels * log_softmax(logits)); y = softmax; partial[b] = block's loss sum
__global__ __launch_bounds__(BLOCK) void softmax_xent_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ labels,
                                                             float* __restrict__ y,
                                                             float* __restrict__ row_loss,
                                                             float* __restrict__ partial,
                                                             float* __restrict__ mean_direct,
                                                             long rows, long cols) {
  __shared__ float sh[WAVES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long wave = (blockIdx.x * (long)BLOCK + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * BLOCK) >> 6;
  float wave_loss = 0.f;
  for (long r = wave; r < rows; r += nwaves) {
    const float* row = logits + r * cols;
    const float* lab = labels + r * cols;
    float m = -INFINITY;
    for (long c = lane; c < cols; c += 64) m = fmaxf(m, row[c]);
    m = dtf_wave_max(m);
    float s = 0.f;
    for (long c = lane; c < cols; c += 64) s += expf(row[c] - m);
    s = dtf_wave_sum(s);
    const float inv = 1.f / s, ls = logf(s);
    float l = 0.f;
    for (long c = lane; c < cols; c += 64) {
      const float d = row[c] - m;
      if (y) y[r * cols + c] = expf(d) * inv;
      l -= lab[c] * (d - ls);
    }
    l = dtf_wave_sum(l);
    if (row_loss && lane == 0) row_loss[r] = l;
    wave_loss += l;
  }
  if (lane == 0) sh[w] = wave_loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) t += sh[k];
    if (partial) partial[blockIdx.x] = t;
    if (mean_direct) mean_direct[0] = t / (float)rows;   // single-workgroup launch only
  }
}

// Narrow rows (cols <= NARROW_MAX, e.g. the 10 MNIST classes): one LANE per row instead of one
// wave per row -- the row lives in registers, there are no cross-lane reductions per row, and a
// 100-row training batch is one short wave-parallel pass instead of 25 serial row visits per
// wave. The block's loss sum is folded in fixed order (lanes by shuffle, waves through LDS).
constexpr int NARROW_MAX = 32;

template <int COLS_MAX>
__global__ __launch_bounds__(BLOCK) void softmax_xent_narrow_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, float* __restrict__ y,
    float* __restrict__ row_loss, float* __restrict__ partial, float* __restrict__ mean_direct,
    long rows, int cols) {
  __shared__ float sh[WAVES];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float mine = 0.f;
  for (long r = blockIdx.x * (long)BLOCK + threadIdx.x; r < rows; r += (long)gridDim.x * BLOCK) {
    float v[COLS_MAX];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < COLS_MAX; ++c) {
      v[c] = c < cols ? logits[r * cols + c] : -INFINITY;
      m = fmaxf(m, v[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < COLS_MAX; ++c) {
      v[c] -= m;
      s += c < cols ? expf(v[c]) : 0.f;
    }
    const float inv = 1.f / s, ls = logf(s);
    float l = 0.f;
#pragma unroll
    for (int c = 0; c < COLS_MAX; ++c) {
      if (c < cols) {
        if (y) y[r * cols + c] = expf(v[c]) * inv;
        l -= labels[r * cols + c] * (v[c] - ls);
      }
    }
    if (row_loss) row_loss[r] = l;
    mine += l;
  }
  mine = dtf_wave_sum(mine);
  if (lane == 0) sh[w] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < WAVES; ++k) t += sh[k];
    if (partial) partial[blockIdx.x] = t;
    if (mean_direct) mean_direct[0] = t / (float)rows;
  }
}

// backward of mean(xent(softmax)) for narrow rows, lane per row:
// dz = (y * rowsum(labels) - labels) * dloss / rows, dbias[c] = sum_r dz[r, c]; one workgroup
template <int COLS_MAX>
__global__ __launch_bounds__(1024) void xent_grad_narrow_kernel(
    const float* __restrict__ y, const float* __restrict__ labels, const float* __restrict__ dloss,
    float* __restrict__ dz, float* __restrict__ dbias, long rows, int cols) {
  __shared__ float sh[16][COLS_MAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float scale = dloss[0] / (float)rows;
  float colacc[COLS_MAX];
#pragma unroll
  for (int c = 0; c < COLS_MAX; ++c) colacc[c] = 0.f;
  for (long r = threadIdx.x; r < rows; r += 1024) {
    float lab[COLS_MAX];
    float lsum = 0.f;
#pragma unroll
    for (int c = 0; c < COLS_MAX; ++c) {
      lab[c] = c < cols ? labels[r * cols + c] : 0.f;
      lsum += lab[c];
    }
#pragma unroll
    for (int c = 0; c < COLS_MAX; ++c) {
      if (c < cols) {
        const float v = (y[r * cols + c] * lsum - lab[c]) * scale;
        dz[r * cols + c] = v;
        colacc[c] += v;
      }
    }
  }
  if (!dbias) return;
#pragma unroll
  for (int c = 0; c < COLS_MAX; ++c) {
    const float t = dtf_wave_sum(colacc[c]);
    if (lane == 0) sh[w][c] = t;
  }
  __syncthreads();
  if (threadIdx.x < cols) {
    const int nwaves = blockDim.x >> 6;
    float t = 0.f;
    for (int k = 0; k < nwaves; ++k) t += sh[k][threadIdx.x];
    dbias[threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(64) void fold_partials_kernel(const float* __restrict__ partial,
                                                           float* __restrict__ out, int m,
                                                           float scale) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < m; i += 64) acc += partial[i];
  acc = dtf_wave_sum(acc);
  if (threadIdx.x == 0) out[0] = acc * scale;
}

// softmax backward: dx = (dy - sum(dy*y)) * y, wave per row
__global__ __launch_bounds__(BLOCK) void softmax_grad_kernel(const float* __restrict__ y,
                                                             const float* __restrict__ dy,
                                                             float* __restrict__ dx, long rows,
                                                             long cols) {
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)BLOCK + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * BLOCK) >> 6;
  for (long r = wave; r < rows; r += nwaves) {
    float dot = 0.f;
    for (long c = lane; c < cols; c += 64) dot += dy[r * cols + c] * y[r * cols + c];
    dot = dtf_wave_sum(dot);
    for (long c = lane; c < cols; c += 64)
      dx[r * cols + c] = (dy[r * cols + c] - dot) * y[r * cols + c];
  }
}

int rows_grid(long rows) {
  long blocks = (rows + WAVES - 1) / WAVES;
  return (int)(blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks);
}

}  // namespace

DTF_API int dtf_softmax_rows_f32(const float* x, float* y, float* log_y, long rows, long cols,
                                 hipStream_t stream) {
  if (rows <= 0) return DTF_OK;
  if (!x || (!y && !log_y) || cols <= 0) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows_grid(rows)), dim3(BLOCK), 0, stream, x, y,
                     log_y, rows, cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_softmax_xent_workspace(long rows) { return rows_grid(rows); }

// y[rows, cols] = softmax(logits) (may be null), row_loss[rows] (may be null),
// mean_loss[0] = mean over rows (may be null; needs workspace of dtf_softmax_xent_workspace words)
DTF_API int dtf_softmax_xent_f32(const float* logits, const float* labels, float* y,
                                 float* row_loss, float* mean_loss, float* workspace, long rows,
                                 long cols, hipStream_t stream) {
  if (!logits || !labels || rows <= 0 || cols <= 0) return DTF_ERR_BAD_ARG;
  if (mean_loss && !workspace) return DTF_ERR_BAD_ARG;
  if (cols <= NARROW_MAX) {
    // lane per row; a training batch (100 rows in the reference) is one workgroup, which then
    // writes the mean itself: one launch instead of two
    long nb = (rows + BLOCK - 1) / BLOCK;
    const int ws = rows_grid(rows);                  // what the caller's workspace holds
    const int blocks = (int)(nb > ws ? ws : nb);
    const bool single = blocks == 1;
    float* part = (mean_loss && !single) ? workspace : nullptr;
    float* direct = single ? mean_loss : nullptr;
    if (cols <= 16)
      hipLaunchKernelGGL(softmax_xent_narrow_kernel<16>, dim3(blocks), dim3(BLOCK), 0, stream,
                         logits, labels, y, row_loss, part, direct, rows, (int)cols);
    else
      hipLaunchKernelGGL(softmax_xent_narrow_kernel<NARROW_MAX>, dim3(blocks), dim3(BLOCK), 0,
                         stream, logits, labels, y, row_loss, part, direct, rows, (int)cols);
    if (mean_loss && !single)
      hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(64), 0, stream, workspace, mean_loss,
                         blocks, 1.0f / (float)rows);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  const bool single = rows <= 8;
  const int blocks = single ? 1 : rows_grid(rows);
  hipLaunchKernelGGL(softmax_xent_kernel, dim3(blocks), dim3(BLOCK), 0, stream, logits, labels, y,
                     row_loss, (mean_loss && !single) ? workspace : nullptr,
                     single ? mean_loss : nullptr, rows, cols);
  if (mean_loss && !single)
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1), dim3(64), 0, stream, workspace, mean_loss,
                       blocks, 1.0f / (float)rows);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_softmax_grad_f32(const float* y, const float* dy, float* dx, long rows, long cols,
                                 hipStream_t stream) {
  if (rows <= 0) return DTF_OK;
  if (!y || !dy || !dx || cols <= 0) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(softmax_grad_kernel, dim3(rows_grid(rows)), dim3(BLOCK), 0, stream, y, dy, dx,
                     rows, cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dz = (y * rowsum(labels) - labels) * dloss / rows and dbias = column sums of dz, for rows of
// at most 32 columns (lane per row, one workgroup). Returns DTF_ERR_UNSUPPORTED for wider rows.
DTF_API int dtf_xent_grad_narrow_f32(const float* y, const float* labels, const float* dloss,
                                     float* dz, float* dbias, long rows, long cols,
                                     hipStream_t stream) {
  if (!y || !labels || !dloss || !dz || rows <= 0 || cols <= 0) return DTF_ERR_BAD_ARG;
  if (cols > NARROW_MAX) return DTF_ERR_UNSUPPORTED;
  const int threads = rows >= 1024 ? 1024 : (int)((rows + 63) / 64 * 64);
  if (cols <= 16)
    hipLaunchKernelGGL(xent_grad_narrow_kernel<16>, dim3(1), dim3(threads), 0, stream, y, labels,
                       dloss, dz, dbias, rows, (int)cols);
  else
    hipLaunchKernelGGL(xent_grad_narrow_kernel<NARROW_MAX>, dim3(1), dim3(threads), 0, stream, y,
                       labels, dloss, dz, dbias, rows, (int)cols);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
