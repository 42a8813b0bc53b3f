// Mean-squared-error loss, its gradient and the bias gradient behind it in ONE launch.
//
// A regression step written against the facade (models/linear_regression.py:
// loss = mean((x W + b - y)^2), plain SGD) runs, as general kernels, five launches between the
// prediction and the weight-gradient product: partial sums of the squared differences, their
// fold, the fill that seeds d(loss) = 1, the element-wise gradient and the column sum that is the
// bias gradient. All five see at most a few thousand elements, so each is a launch and one memory
// round trip and nothing else. This kernel is those five in one workgroup -- and, in a captured
// step, also the gather of the step's small results into the step ring (step_ring.h), so the step
// is prediction, this launch, the weight-gradient product and the update: four.
//
// Every sum is taken in the order the separate kernels take it (reduce.hip: sq_diff_stage1's
// grid-stride fma chains and block folds, reduce_all_stage2's fold of the partials,
// reduce_cols_kernel's row groups), with the same launch shapes emulated inside the one
// workgroup, so the fused step and the chain agree bit for bit.
#include "common.h"
#include "step_ring.h"

namespace {

constexpr int MS_THREADS = 1024;
constexpr int MS_BLOCK = 256;              // BLOCK of reduce.hip: threads per emulated workgroup
constexpr int MS_MAX_PACK = 16;

struct MsePack {
  const float* src[MS_MAX_PACK];
  float add[MS_MAX_PACK];
  int offset[MS_MAX_PACK + 1];
  int npack;
};

struct MseArgs {
  const float* pred;       // [rows, cols]
  const float* labels;     // [rows, cols]
  const float* g;          // device scalar d(loss), or null: g_const
  float g_const;
  float* loss;             // scalar
  float* dpred;            // [rows, cols]
  float* dbias;            // [cols] or null
  long rows, cols;
  float inv_n;             // 1 / (rows * cols), rounded on the host like the fold's scale
  int vblocks;             // emulated workgroups of the partial-sum stage
  int col_groups;          // row groups of the emulated column sum
};

// a * b rounded to fp32 as a value of its own: the separate kernels store this product to memory
// before the column sum adds it, so it must not be contracted into an fma with the running sum
__device__ __forceinline__ float rounded_product(float a, float b) {
  float v = a * b;
  asm volatile("" : "+v"(v));
  return v;
}

__global__ __launch_bounds__(MS_THREADS) void mse_step_kernel(MseArgs p, MsePack pack,
                                                              DtfStepRing ring, int ring_first,
                                                              int labels_index) {
  __shared__ float wave_sums[MS_THREADS / 64];
  __shared__ float partial[1024];
  __shared__ float colsh[MS_THREADS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long n = p.rows * p.cols;
  unsigned seq = 0;
  if (ring.ctr) {
    seq = ring_first ? dtf_ring_begin(ring) : dtf_ring_current(ring);
    // the fed labels, where this replay's batch lies (no gather of the feeds in front of the step)
    if (labels_index >= 0)
      p.labels = (const float*)(ring_first ? dtf_ring_feed(ring, seq, labels_index)
                                           : dtf_ring_feed_later(ring, labels_index));
  }

  // ---- loss: partial[vb] as sq_diff_stage1 leaves it for a grid of vblocks x 256 threads ------
  const long nthreads = (long)p.vblocks * MS_BLOCK;
  for (int base = 0; base < p.vblocks; base += MS_THREADS / MS_BLOCK) {
    const int vb = base + (wave >> 2);                 // four waves emulate one workgroup
    const int vt = (wave & 3) * 64 + lane;             // its thread
    float acc = 0.f;
    if (vb < p.vblocks)
      for (long i = (long)vb * MS_BLOCK + vt; i < n; i += nthreads) {
        const float d = p.pred[i] - p.labels[i];
        acc = __builtin_fmaf(d, d, acc);
      }
    acc = dtf_wave_sum(acc);
    if (lane == 0) wave_sums[wave] = acc;
    __syncthreads();
    if (tid < MS_THREADS / MS_BLOCK && base + tid < p.vblocks) {
      float t = 0.f;
      for (int w = 0; w < MS_BLOCK / 64; ++w) t += wave_sums[tid * 4 + w];
      partial[base + tid] = t;
    }
    __syncthreads();
  }
  // reduce_all_stage2: 256 threads fold the partials, scaled by 1 / n
  {
    float acc = 0.f;
    if (tid < MS_BLOCK)
      for (int i = tid; i < p.vblocks; i += MS_BLOCK) acc += partial[i];
    acc = dtf_wave_sum(acc);
    if (lane == 0) wave_sums[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < MS_BLOCK / 64; ++w) t += wave_sums[w];
      p.loss[0] = t * p.inv_n;
    }
  }

  // ---- d(pred) = (2 g / n) (pred - labels), as mse_grad_kernel spells it ------------------------
  const float s = 2.f * (p.g ? p.g[0] : p.g_const) / (float)n;
  for (long i = tid; i < n; i += MS_THREADS) p.dpred[i] = s * (p.pred[i] - p.labels[i]);

  // ---- d(bias)[c] = sum_r d(pred)[r, c], in reduce_cols_kernel<COL_SUM>'s order ----------------
  if (p.dbias) {
    const int g = wave, groups = p.col_groups;
    const long c = lane;
    const bool live = c < p.cols && g < groups;
    float acc = 0.f;
    if (live) {
      long r = g;
      for (; r + 3L * groups < p.rows; r += 4L * groups) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long idx = (r + (long)u * groups) * p.cols + c;
          v[u] = rounded_product(s, p.pred[idx] - p.labels[idx]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
      }
      for (; r < p.rows; r += groups) {
        const long idx = r * p.cols + c;
        acc += rounded_product(s, p.pred[idx] - p.labels[idx]);
      }
    }
    colsh[tid] = acc;
    __syncthreads();
    if (g == 0 && c < p.cols) {
      float t = 0.f;
      for (int k = 0; k < groups; ++k) t += colsh[k * 64 + lane];
      p.dbias[c] = t;
    }
  }

  // ---- the step's small results, for the host ---------------------------------------------------
  if (!ring.ctr || pack.npack < 0) return;        // (a ring for the labels only: no gather)
  __syncthreads();                     // loss (and anything else of this launch) is written
  float* dst = dtf_ring_out(ring, seq);
  for (int i = 0; i < pack.npack; ++i) {
    const int m = pack.offset[i + 1] - pack.offset[i];
    for (int j = tid; j < m; j += MS_THREADS) dst[pack.offset[i] + j] = pack.src[i][j] + pack.add[i];
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) dtf_ring_finish(ring, seq);
}

}  // namespace

// 1 when the fused launch takes this shape (one workgroup: at most 65536 elements; the bias
// gradient needs whole rows in one wave: at most 64 columns)
DTF_API int dtf_mse_step_supported(long rows, long cols, int want_dbias) {
  return rows >= 1 && cols >= 1 && rows * cols <= 65536 && (!want_dbias || cols <= 64);
}

// loss = mean((pred - labels)^2); dpred = (2 g / n)(pred - labels) with g = *g_dev (or g_const
// when g_dev is null); dbias = column sums of dpred (may be null). npack >= 0 with a ring: this
// launch also gathers `npack` small fp32 values (+ adds[i]) into the replay's ring slot, stamps it
// and opens the next replay (later kernels of the step may still read their inputs through the
// ring's device-side address copies); npack < 0: no gather. labels_index >= 0 (with a ring): the
// labels are input `labels_index` of the replay's address table (`labels` may then be null).
DTF_API int dtf_mse_step_f32(const float* pred, const float* labels, const float* g_dev,
                             float g_const, float* loss, float* dpred, float* dbias, long rows,
                             long cols, int npack, const float* const* src, const int* sizes,
                             const float* adds, const void* ring_table, void* ring_ctr,
                             void* ring_out, int ring, int ring_first, int labels_index,
                             hipStream_t stream) {
  if (!pred || !loss || !dpred) return DTF_ERR_BAD_ARG;
  if (!labels && !(ring_ctr && labels_index >= 0)) return DTF_ERR_BAD_ARG;
  if (labels_index >= DTF_RING_FEEDS || (labels_index >= 0 && !ring_table)) return DTF_ERR_BAD_ARG;
  if (!dtf_mse_step_supported(rows, cols, dbias != nullptr)) return DTF_ERR_UNSUPPORTED;
  if (npack > MS_MAX_PACK) return DTF_ERR_BAD_ARG;
  const bool ringed = ring_ctr != nullptr;
  if (npack >= 0 && !ringed) return DTF_ERR_BAD_ARG;
  if (ringed && (!ring_out || ring < 1 || (ring & (ring - 1)))) return DTF_ERR_BAD_ARG;
  const long n = rows * cols;
  MseArgs a = {};
  a.pred = pred;
  a.labels = labels;
  a.g = g_dev;
  a.g_const = g_const;
  a.loss = loss;
  a.dpred = dpred;
  a.dbias = dbias;
  a.rows = rows;
  a.cols = cols;
  a.inv_n = 1.f / (float)n;
  long blocks = (n + MS_BLOCK * 8L - 1) / (MS_BLOCK * 8L);        // dtf_reduce_all_workspace
  a.vblocks = (int)(blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks);
  long groups = (rows + 3) / 4;                                     // cols_threads / 64
  a.col_groups = (int)(groups < 4 ? 4 : groups > 16 ? 16 : groups);
  MsePack pk = {};
  pk.npack = npack;
  int off = 0;
  for (int i = 0; i < npack; ++i) {
    if (!src || !sizes || !src[i] || sizes[i] < 0) return DTF_ERR_BAD_ARG;
    pk.src[i] = src[i];
    pk.add[i] = adds ? adds[i] : 0.f;
    pk.offset[i] = off;
    off += sizes[i];
  }
  if (npack >= 0) pk.offset[npack] = off;
  if (off > DTF_RING_WORDS - 1) return DTF_ERR_BAD_ARG;
  const DtfStepRing r{(const unsigned long long*)ring_table, (unsigned*)ring_ctr, (float*)ring_out,
                      ringed ? ring : 0, -1};
  hipLaunchKernelGGL(mse_step_kernel, dim3(1), dim3(MS_THREADS), 0, stream, a, pk, r, ring_first,
                     ringed ? labels_index : -1);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
