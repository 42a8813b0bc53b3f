// Optimizer update, random initialisation and partial-sum reduction kernels.
//
// * multi-tensor SGD: w -= lr * g for up to 16 tensors in ONE launch (the reference applies
//   four ApplyGradientDescent ops plus a global_step increment per step,
//   tf_distributed.py:75-76); scalar counters (global_step) are bumped by the same launch.
// * Philox4x32-10 normal / uniform fill (tf.random_normal at tf_distributed.py:51-52 and
//   tf_distributed_1000Matrix.py:43-44): counter-based, so element e of a stream depends only
//   on (seed, op_seed, e) and any rank can generate any shard of a tensor independently.
//   Bit-compatible (up to libm rounding) with ops/philox.py, which is the host oracle.
// * reduce_partials: out = sum of up to 8 partial buffers (the local reduction step of a
//   direct reduce-scatter over peer-mapped buffers, and of K-sharded GEMM partial sums).
#include "common.h"
#include "step_ring.h"

namespace {

constexpr int BLOCK = 256;

struct SgdTensor {
  float* w;
  const float* g;
  long n;
  float lr;
};
constexpr int MAX_SGD_TENSORS = 16;
constexpr int MAX_COUNTERS = 4;
struct SgdArgs {
  SgdTensor t[MAX_SGD_TENSORS];
  float* counter[MAX_COUNTERS];
  float delta[MAX_COUNTERS];
  int num_tensors;
  int num_counters;
};

// grid = (blocks_x, num_tensors): blockIdx.y selects the tensor
__global__ __launch_bounds__(BLOCK) void multi_sgd_kernel(SgdArgs args) {
  const SgdTensor t = args.t[blockIdx.y];
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  const bool vec = (((uintptr_t)t.w | (uintptr_t)t.g) & 15) == 0;
  if (vec) {
    const long n4 = t.n >> 2;
    f32x4* w4 = reinterpret_cast<f32x4*>(t.w);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(t.g);
    // w = fma(-lr, g, w), one rounding: the update is spelled the same wherever it is applied
    // (here, and in the epilogues of the kernels that apply a gradient they have just computed)
    for (long i = start; i < n4; i += step) {
      const f32x4 g = g4[i];
      f32x4 w = w4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = __builtin_fmaf(-t.lr, g[j], w[j]);
      w4[i] = w;
    }
    for (long i = (n4 << 2) + start; i < t.n; i += step)
      t.w[i] = __builtin_fmaf(-t.lr, t.g[i], t.w[i]);
  } else {
    for (long i = start; i < t.n; i += step) t.w[i] = __builtin_fmaf(-t.lr, t.g[i], t.w[i]);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < args.num_counters)
    args.counter[threadIdx.x][0] += args.delta[threadIdx.x];
}

// ---- Philox4x32-10 -------------------------------------------------------------------------
struct U4 { unsigned int x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(unsigned long block, unsigned int k0, unsigned int k1) {
  unsigned int c0 = (unsigned int)block, c1 = (unsigned int)(block >> 32), c2 = 0u, c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long p0 = (unsigned long)0xD2511F53u * c0;
    const unsigned long p1 = (unsigned long)0xCD9E8D57u * c2;
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
    const unsigned int n1 = (unsigned int)p1;
    const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
    const unsigned int n3 = (unsigned int)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(unsigned int bits) {
  return (float)(bits >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;  // 2^-24, 2^-25
}

template <bool NORMAL, bool BF16_OUT>
__global__ __launch_bounds__(BLOCK) void philox_fill_kernel(void* __restrict__ out_, long n,
                                                            unsigned int seed,
                                                            unsigned int op_seed, long offset,
                                                            float a, float b) {
  // NORMAL: value = a (mean) + b (stddev) * z;   uniform: value = a (min) + b (max - min) * u
  const long first_block = offset >> 2;
  const long nblocks = ((offset + n + 3) >> 2) - first_block;
  for (long q = blockIdx.x * (long)BLOCK + threadIdx.x; q < nblocks;
       q += (long)gridDim.x * BLOCK) {
    const U4 r = philox4x32_10((unsigned long)(first_block + q), seed, op_seed);
    float v[4];
    const float u0 = u01(r.x), u1 = u01(r.y), u2 = u01(r.z), u3 = u01(r.w);
    if (NORMAL) {
      const float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
      const float t0 = 6.283185307179586f * u1, t1 = 6.283185307179586f * u3;
      v[0] = r0 * cosf(t0); v[1] = r0 * sinf(t0);
      v[2] = r1 * cosf(t1); v[3] = r1 * sinf(t1);
    } else {
      v[0] = u0; v[1] = u1; v[2] = u2; v[3] = u3;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long e = ((first_block + q) << 2) + j - offset;   // index into out
      if (e >= 0 && e < n) {
        const float val = a + b * v[j];
        if (BF16_OUT) reinterpret_cast<unsigned short*>(out_)[e] = dtf_f32_to_bf16(val);
        else reinterpret_cast<float*>(out_)[e] = val;
      }
    }
  }
}

// Dropout: out = x * m with m = (u < keep) / keep, u the uniform of stream element
// round * 4 * ceil(n / 4) + e. `round` is a 64-bit INTEGER counter in device memory that the caller
// bumps after every run (dtf_bump_counter_u64), so a captured step draws a fresh mask at every
// replay without the host touching anything -- and keeps doing so for 2^64 runs (an fp32 counter
// advanced by +1 sticks at 2^24 and would then repeat one mask for ever); `keep_dev`
// (optional) overrides the constant keep probability (a fed keep_prob placeholder: 1.0 at
// evaluation time). The scaled mask is kept for the backward pass.
__global__ __launch_bounds__(BLOCK) void dropout_kernel(const float* __restrict__ x,
                                                        float* __restrict__ out,
                                                        float* __restrict__ mask, long n,
                                                        unsigned int seed, unsigned int op_seed,
                                                        const unsigned long long* __restrict__ round,
                                                        float keep,
                                                        const float* __restrict__ keep_dev) {
  if (keep_dev) keep = keep_dev[0];
  const float scale = 1.f / keep;
  const long nblocks = (n + 3) >> 2;
  const unsigned long base = (unsigned long)round[0] * (unsigned long)nblocks;   // (wraps mod 2^64)
  for (long q = blockIdx.x * (long)BLOCK + threadIdx.x; q < nblocks;
       q += (long)gridDim.x * BLOCK) {
    const U4 r = philox4x32_10(base + (unsigned long)q, seed, op_seed);
    const float u[4] = {u01(r.x), u01(r.y), u01(r.z), u01(r.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long e = (q << 2) + j;
      if (e < n) {
        const float m = u[j] < keep ? scale : 0.f;
        mask[e] = m;
        out[e] = x[e] * m;
      }
    }
  }
}

// 2-D window of a stream laid out as a [*, global_cols] matrix: out[i, j] is stream element
// (row0 + i) * global_cols + col0 + j. This is how a rank materialises its shard of a global
// random matrix (row block, column block or 2-D block) without generating the rest.
// OUT: 0 = fp32, 1 = bf16, 2 = fp8 e4m3 (saturating)
template <bool NORMAL, int OUT>
__global__ __launch_bounds__(BLOCK) void philox_fill_2d_kernel(void* __restrict__ out_, long rows,
                                                               long cols, long ld_out,
                                                               unsigned int seed,
                                                               unsigned int op_seed, long row0,
                                                               long col0, long global_cols,
                                                               float a, float b) {
  const long n = rows * cols;
  for (long t = blockIdx.x * (long)BLOCK + threadIdx.x; t < n; t += (long)gridDim.x * BLOCK) {
    const long i = t / cols, j = t % cols;
    const long e = (row0 + i) * global_cols + col0 + j;
    const U4 r = philox4x32_10((unsigned long)(e >> 2), seed, op_seed);
    const int lane = (int)(e & 3);
    float v;
    if (NORMAL) {
      const float ua = u01(lane < 2 ? r.x : r.z), ub = u01(lane < 2 ? r.y : r.w);
      const float rad = sqrtf(-2.f * logf(ua)), th = 6.283185307179586f * ub;
      v = (lane & 1) ? rad * sinf(th) : rad * cosf(th);
    } else {
      v = u01(lane == 0 ? r.x : lane == 1 ? r.y : lane == 2 ? r.z : r.w);
    }
    const float val = a + b * v;
    if (OUT == 2) reinterpret_cast<unsigned char*>(out_)[i * ld_out + j] = dtf_f32_to_e4m3(val);
    else if (OUT == 1)
      reinterpret_cast<unsigned short*>(out_)[i * ld_out + j] = dtf_f32_to_bf16(val);
    else reinterpret_cast<float*>(out_)[i * ld_out + j] = val;
  }
}

// ---- partial-sum reduction -------------------------------------------------------------------
constexpr int MAX_PARTIALS = 8;
struct PartialArgs {
  const void* src[MAX_PARTIALS];
  int count;
};

__global__ __launch_bounds__(BLOCK) void reduce_partials_f32_kernel(PartialArgs args,
                                                                    float* __restrict__ out,
                                                                    long n, float scale) {
  const long n4 = n >> 2;
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  for (long i = start; i < n4; i += step) {
    f32x4 acc = reinterpret_cast<const f32x4*>(args.src[0])[i];
    for (int p = 1; p < args.count; ++p) acc += reinterpret_cast<const f32x4*>(args.src[p])[i];
    reinterpret_cast<f32x4*>(out)[i] = acc * scale;
  }
  for (long i = (n4 << 2) + start; i < n; i += step) {
    float acc = 0.f;
    for (int p = 0; p < args.count; ++p) acc += reinterpret_cast<const float*>(args.src[p])[i];
    out[i] = acc * scale;
  }
}

// bf16 partials, fp32 accumulation, bf16 result (halves the bytes a K-sharded GEMM moves)
__global__ __launch_bounds__(BLOCK) void reduce_partials_bf16_kernel(PartialArgs args,
                                                                     unsigned short* __restrict__ out,
                                                                     long n, float scale) {
  const long n8 = n >> 3;
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  for (long i = start; i < n8; i += step) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < args.count; ++p) {
      const bf16x8 v = reinterpret_cast<const bf16x8*>(args.src[p])[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += dtf_bf16_to_f32((unsigned short)v[j]);
    }
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (short)dtf_f32_to_bf16(acc[j] * scale);
    reinterpret_cast<bf16x8*>(out)[i] = r;
  }
  for (long i = (n8 << 3) + start; i < n; i += step) {
    float acc = 0.f;
    for (int p = 0; p < args.count; ++p)
      acc += dtf_bf16_to_f32(reinterpret_cast<const unsigned short*>(args.src[p])[i]);
    out[i] = dtf_f32_to_bf16(acc * scale);
  }
}

// ---- gather of small fetched values ------------------------------------------------------------
constexpr int MAX_PACK = 16;
struct PackArgs {
  const float* src[MAX_PACK];
  int offset[MAX_PACK + 1];   // dst[offset[i] .. offset[i+1]) <- src[i] + add[i]
  float add[MAX_PACK];        // (a step counter fetched ahead of the launch that bumps it)
  int count;
};

// one workgroup: wave-strided copy of up to 16 tiny tensors into one contiguous staging buffer.
// With a signal pair the kernel finally copies the word at `seq_src` (the host's step number,
// read when the kernel runs) to `seq_dst` behind a system-scope release: a host that polls
// `seq_dst` in pinned memory sees the packed words complete as soon as it sees the number, without
// going through a stream synchronisation (whose wake-up costs several microseconds).
__global__ __launch_bounds__(BLOCK) void pack_f32_kernel(PackArgs args, float* __restrict__ dst,
                                                         const float* __restrict__ seq_src,
                                                         float* __restrict__ seq_dst,
                                                         DtfStepRing ring, int ring_first) {
  // in a ring step the words go to this replay's slot of the ring, the slot's last word is the
  // replay's stamp, and the launch is the LAST ring kernel of the step: it opens the next replay
  unsigned seq = 0;
  if (ring.ctr) {
    seq = ring_first ? ring.ctr[1] : dtf_ring_current(ring);
    dst = dtf_ring_out(ring, seq);
  }
  for (int i = 0; i < args.count; ++i) {
    const int n = args.offset[i + 1] - args.offset[i];
    const float add = args.add[i];
    for (int j = threadIdx.x; j < n; j += BLOCK) dst[args.offset[i] + j] = args.src[i][j] + add;
  }
  if (ring.ctr) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) dtf_ring_finish(ring, seq);
  } else if (seq_dst != nullptr) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
      const float seq = __builtin_nontemporal_load(seq_src);
      __hip_atomic_store(seq_dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- feeds of a captured step, gathered through a pointer table ---------------------------------
// A captured step reads its inputs from fixed buffers. When the inputs already live in GPU memory
// (batch views of a device-resident data set) their addresses change every step; instead of one
// runtime copy per input in front of the graph launch, the graph's first node is this kernel: it
// reads the CURRENT source addresses from a small table in pinned host memory, which the host
// rewrites with plain stores before each launch, and copies the inputs into the fixed buffers.
constexpr int MAX_FEEDS = 8;
struct FeedDst {
  char* dst[MAX_FEEDS];
  long bytes[MAX_FEEDS];
  long first_block[MAX_FEEDS + 1];   // feed i owns workgroups [first_block[i], first_block[i+1])
  int count;
};
constexpr int FEED_BYTES_PER_BLOCK = BLOCK * 16 * 4;

__global__ __launch_bounds__(BLOCK) void gather_feeds_kernel(
    const unsigned long long* __restrict__ table, FeedDst args, DtfStepRing ring) {
  int seg = 0;
  while (seg + 1 < args.count && (long)blockIdx.x >= args.first_block[seg + 1]) ++seg;
  // in a ring step this is the FIRST kernel of the replay and the table is the ring's
  const char* src = ring.ctr
      ? reinterpret_cast<const char*>(dtf_ring_feed(ring, dtf_ring_begin(ring), seg))
      : reinterpret_cast<const char*>(
            __builtin_nontemporal_load(table + seg));      // host memory: never from a stale line
  char* dst = args.dst[seg];
  const long n = args.bytes[seg];
  const long base = ((long)blockIdx.x - args.first_block[seg]) * FEED_BYTES_PER_BLOCK;
  if ((((uintptr_t)src) & 15) == 0) {                     // dst buffers are allocator-aligned
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long o = base + ((long)u * BLOCK + threadIdx.x) * 16;
      if (o + 16 <= n) {
        *reinterpret_cast<f32x4*>(dst + o) = *reinterpret_cast<const f32x4*>(src + o);
      } else {
        for (long b = o; b < n && b < o + 16; ++b) dst[b] = src[b];
      }
    }
  } else {
    const long end = base + FEED_BYTES_PER_BLOCK < n ? base + FEED_BYTES_PER_BLOCK : n;
    for (long b = base + threadIdx.x; b < end; b += BLOCK) dst[b] = src[b];
  }
}

int grid_for(long n, int per_thread) {
  long blocks = (n + (long)BLOCK * per_thread - 1) / ((long)BLOCK * per_thread);
  return (int)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks);
}

}  // namespace

// w[i] -= lr[i] * g[i] for `count` (<= 16) tensors, counters[j][0] += deltas[j] for
// `num_counters` (<= 4) scalars -- one launch.
DTF_API int dtf_multi_sgd_f32(int count, float* const* w, const float* const* g, const long* n,
                              const float* lr, int num_counters, float* const* counters,
                              const float* deltas, hipStream_t stream) {
  if (count < 0 || count > MAX_SGD_TENSORS || num_counters < 0 || num_counters > MAX_COUNTERS)
    return DTF_ERR_BAD_ARG;
  if (count == 0 && num_counters == 0) return DTF_OK;
  SgdArgs args = {};
  long max_n = 1;
  for (int i = 0; i < count; ++i) {
    if (!w[i] || !g[i] || n[i] < 0) return DTF_ERR_BAD_ARG;
    args.t[i] = SgdTensor{w[i], g[i], n[i], lr[i]};
    if (n[i] > max_n) max_n = n[i];
  }
  for (int j = 0; j < num_counters; ++j) {
    if (!counters[j]) return DTF_ERR_BAD_ARG;
    args.counter[j] = counters[j];
    args.delta[j] = deltas[j];
  }
  args.num_tensors = count;
  args.num_counters = num_counters;
  if (count == 0)   // counters only: give the kernel one empty tensor to walk
    args.t[0] = SgdTensor{args.counter[0], args.counter[0], 0, 0.f};
  dim3 grid(grid_for(max_n, 4), count > 0 ? count : 1), block(BLOCK);
  hipLaunchKernelGGL(multi_sgd_kernel, grid, block, 0, stream, args);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// out = x * mask, mask[e] = (u_e < keep) / keep drawn from round `round[0]` of the (seed, op_seed)
// stream (see dropout_kernel); keep_dev: optional device scalar replacing `keep`.
__global__ void bump_counter_kernel(unsigned long long* counter, unsigned long long by) {
  if (threadIdx.x == 0 && blockIdx.x == 0) counter[0] += by;
}

// counter[0] += by (a 64-bit integer in device memory; stream-ordered, capturable)
DTF_API int dtf_bump_counter_u64(void* counter, long by, hipStream_t stream) {
  if (!counter || ((uintptr_t)counter & 7) || by < 0) return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(64), 0, stream,
                     (unsigned long long*)counter, (unsigned long long)by);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_dropout_f32(const float* x, float* out, float* mask, long n, unsigned int seed,
                            unsigned int op_seed, const unsigned long long* round, float keep,
                            const float* keep_dev, hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!x || !out || !mask || !round || ((uintptr_t)round & 7) ||
      (!keep_dev && !(keep > 0.f && keep <= 1.f)))
    return DTF_ERR_BAD_ARG;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((n + 3) >> 2, 1)), dim3(BLOCK), 0, stream, x,
                     out, mask, n, seed, op_seed, round, keep, keep_dev);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dst[i][0..bytes[i]) = (*table[i])[0..bytes[i]) for count <= 8 inputs: `table` holds the source
// ADDRESSES (64-bit, in pinned host memory or device memory) and is read when the kernel runs,
// not when it is launched or captured.
static int gather_feeds_launch(const void* table, const DtfStepRing& ring, int count,
                               void* const* dst, const long* bytes, hipStream_t stream) {
  if (count < 1 || count > MAX_FEEDS) return DTF_ERR_BAD_ARG;
  FeedDst args = {};
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (!dst[i] || bytes[i] <= 0 || ((uintptr_t)dst[i] & 15)) return DTF_ERR_BAD_ARG;
    args.dst[i] = (char*)dst[i];
    args.bytes[i] = bytes[i];
    args.first_block[i] = blocks;
    blocks += (bytes[i] + FEED_BYTES_PER_BLOCK - 1) / FEED_BYTES_PER_BLOCK;
  }
  args.first_block[count] = blocks;
  args.count = count;
  if (blocks > 0x7fffffffL) return DTF_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gather_feeds_kernel, dim3((unsigned)blocks), dim3(BLOCK), 0, stream,
                     (const unsigned long long*)table, args, ring);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_gather_feeds(const void* table, int count, void* const* dst, const long* bytes,
                             hipStream_t stream) {
  if (!table) return DTF_ERR_BAD_ARG;
  return gather_feeds_launch(table, DtfStepRing{nullptr, nullptr, nullptr, 0, -1}, count, dst, bytes,
                             stream);
}

// The same gather as the FIRST kernel of a ring step (step_ring.h): the source addresses are
// those of the replay's slot of the ring's table.
DTF_API int dtf_gather_feeds_ring(const void* ring_table, void* ring_ctr, void* ring_out, int ring,
                                  int count, void* const* dst, const long* bytes,
                                  hipStream_t stream) {
  if (!ring_table || !ring_ctr || ring < 1 || (ring & (ring - 1))) return DTF_ERR_BAD_ARG;
  const DtfStepRing r{(const unsigned long long*)ring_table, (unsigned*)ring_ctr, (float*)ring_out,
                      ring, -1};
  return gather_feeds_launch(ring_table, r, count, dst, bytes, stream);
}

DTF_API int dtf_philox_normal(void* out, long n, unsigned int seed, unsigned int op_seed,
                              long offset, float mean, float stddev, int bf16_out,
                              hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!out || offset < 0) return DTF_ERR_BAD_ARG;
  dim3 grid(grid_for((n + 3) / 4 + 1, 1)), block(BLOCK);
  if (bf16_out)
    hipLaunchKernelGGL((philox_fill_kernel<true, true>), grid, block, 0, stream, out, n, seed,
                       op_seed, offset, mean, stddev);
  else
    hipLaunchKernelGGL((philox_fill_kernel<true, false>), grid, block, 0, stream, out, n, seed,
                       op_seed, offset, mean, stddev);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_philox_uniform(void* out, long n, unsigned int seed, unsigned int op_seed,
                               long offset, float minval, float maxval, int bf16_out,
                               hipStream_t stream) {
  if (n <= 0) return DTF_OK;
  if (!out || offset < 0) return DTF_ERR_BAD_ARG;
  dim3 grid(grid_for((n + 3) / 4 + 1, 1)), block(BLOCK);
  if (bf16_out)
    hipLaunchKernelGGL((philox_fill_kernel<false, true>), grid, block, 0, stream, out, n, seed,
                       op_seed, offset, minval, maxval - minval);
  else
    hipLaunchKernelGGL((philox_fill_kernel<false, false>), grid, block, 0, stream, out, n, seed,
                       op_seed, offset, minval, maxval - minval);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// out = scale * sum_{p < count} src[p]   (count <= 8; dtype: 0 = fp32, 1 = bf16)
DTF_API int dtf_reduce_partials(int count, const void* const* src, void* out, long n, float scale,
                                int dtype, hipStream_t stream) {
  if (count < 1 || count > MAX_PARTIALS || !out || n <= 0) return DTF_ERR_BAD_ARG;
  PartialArgs args = {};
  for (int p = 0; p < count; ++p) {
    if (!src[p] || ((uintptr_t)src[p] & 15)) return DTF_ERR_BAD_ARG;
    args.src[p] = src[p];
  }
  if ((uintptr_t)out & 15) return DTF_ERR_BAD_ARG;
  args.count = count;
  if (dtype == 0)
    hipLaunchKernelGGL(reduce_partials_f32_kernel, dim3(grid_for(n, 4)), dim3(BLOCK), 0, stream,
                       args, (float*)out, n, scale);
  else if (dtype == 1)
    hipLaunchKernelGGL(reduce_partials_bf16_kernel, dim3(grid_for(n, 8)), dim3(BLOCK), 0, stream,
                       args, (unsigned short*)out, n, scale);
  else
    return DTF_ERR_UNSUPPORTED;
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dst = concat(src[0][0..n0), src[1][0..n1), ...) for count <= 16 small fp32 buffers: the values
// a step returns to the host (loss, step counter, accuracy) gathered by one launch so that
// one device->host copy delivers all of them.
DTF_API int dtf_pack_f32_signal(int count, const float* const* src, const int* sizes,
                                const float* adds, float* dst, const float* seq_src,
                                float* seq_dst, hipStream_t stream) {
  if (count < 1 || count > MAX_PACK || !dst || ((seq_src == nullptr) != (seq_dst == nullptr)))
    return DTF_ERR_BAD_ARG;
  PackArgs args = {};
  int off = 0;
  for (int i = 0; i < count; ++i) {
    if (!src[i] || sizes[i] < 0) return DTF_ERR_BAD_ARG;
    args.src[i] = src[i];
    args.offset[i] = off;
    args.add[i] = adds ? adds[i] : 0.f;
    off += sizes[i];
  }
  args.offset[count] = off;
  args.count = count;
  hipLaunchKernelGGL(pack_f32_kernel, dim3(1), dim3(BLOCK), 0, stream, args, dst, seq_src,
                     seq_dst, DtfStepRing{nullptr, nullptr, nullptr, 0, -1}, 0);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// The same gather as the LAST ring kernel of a captured step (step_ring.h): the words go to the
// replay's slot of the ring's result buffer, which is then stamped, and the next replay is
// opened. count may be 0 (a step that returns nothing small still stamps its slot: the host
// learns from the stamp when the slot's table entries may be rewritten). ring_first: no earlier
// kernel of the step has published the replay number.
DTF_API int dtf_pack_f32_ring(int count, const float* const* src, const int* sizes,
                              const float* adds, const void* ring_table, void* ring_ctr,
                              void* ring_out, int ring, int ring_first, hipStream_t stream) {
  if (count < 0 || count > MAX_PACK || !ring_ctr || !ring_out || ring < 1 || (ring & (ring - 1)))
    return DTF_ERR_BAD_ARG;
  PackArgs args = {};
  int off = 0;
  for (int i = 0; i < count; ++i) {
    if (!src[i] || sizes[i] < 0) return DTF_ERR_BAD_ARG;
    args.src[i] = src[i];
    args.offset[i] = off;
    args.add[i] = adds ? adds[i] : 0.f;
    off += sizes[i];
  }
  if (off > DTF_RING_WORDS - 1) return DTF_ERR_BAD_ARG;
  args.offset[count] = off;
  args.count = count;
  const DtfStepRing r{(const unsigned long long*)ring_table, (unsigned*)ring_ctr, (float*)ring_out,
                      ring, -1};
  hipLaunchKernelGGL(pack_f32_kernel, dim3(1), dim3(BLOCK), 0, stream, args, nullptr, nullptr,
                     nullptr, r, ring_first);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_pack_f32(int count, const float* const* src, const int* sizes, float* dst,
                         hipStream_t stream) {
  return dtf_pack_f32_signal(count, src, sizes, nullptr, dst, nullptr, nullptr, stream);
}

// out[i*ld_out + j] = a + b * g(stream element (row0+i)*global_cols + col0 + j), g = N(0,1) when
// normal else U(0,1). out_kind: 0 = fp32, 1 = bf16, 2 = fp8 e4m3.
DTF_API int dtf_philox_fill_2d(void* out, long rows, long cols, long ld_out, unsigned int seed,
                               unsigned int op_seed, long row0, long col0, long global_cols,
                               float a, float b, int normal, int out_kind, hipStream_t stream) {
  if (out_kind < 0 || out_kind > 2) return DTF_ERR_BAD_ARG;
  if (rows <= 0 || cols <= 0) return DTF_OK;
  if (!out || ld_out < cols || row0 < 0 || col0 < 0 || global_cols < col0 + cols)
    return DTF_ERR_BAD_ARG;
  dim3 grid(grid_for(rows * cols, 4)), block(BLOCK);
#define DTF_FILL2D(NRM, BF)                                                                      \
  hipLaunchKernelGGL((philox_fill_2d_kernel<NRM, BF>), grid, block, 0, stream, out, rows, cols,  \
                     ld_out, seed, op_seed, row0, col0, global_cols, a, b)
  if (normal) {
    if (out_kind == 2) DTF_FILL2D(true, 2);
    else if (out_kind == 1) DTF_FILL2D(true, 1);
    else DTF_FILL2D(true, 0);
  } else {
    if (out_kind == 2) DTF_FILL2D(false, 2);
    else if (out_kind == 1) DTF_FILL2D(false, 1);
    else DTF_FILL2D(false, 0);
  }
#undef DTF_FILL2D
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
