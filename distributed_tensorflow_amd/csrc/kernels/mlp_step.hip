// A whole training step of a two-layer classifier in THREE launches.
//
// The reference's model (tf_distributed.py:58-76) is x -> sigmoid(x W1 + b1) -> softmax(a W2 + b2),
// cross-entropy, plain SGD on W1, b1, W2, b2 and a global_step increment: 48 MFLOP on a batch of
// 100, against which a launch (a few microseconds from the end of one kernel to the first result
// of the next) is expensive. As general kernels the step takes eight launches (feed gather,
// first-layer product, head, head finish, weight-gradient product, multi-tensor SGD, result
// gather, ...). The dependency chain allows three, and that is what this file runs:
//
//   1. layer 1 a = sigmoid(x W1 + b1) on the small-product kernel body (gemm_f32_body.h: one
//              16x16 tile per workgroup, K split over its waves, bias and sigmoid in the
//              epilogue), x read through the step ring's address table -- no feed gather;
//   2. head    the classifier head (mlp_head_body.h), one workgroup per 16 rows: logits, softmax,
//              loss, accuracy bit, dz2, its share of dW2, db2 and db1, and
//              dz1 = (dz2 W2^T) a (1 - a), which goes to memory for launch 3; labels through the
//              ring's table;
//   3. update  workgroups 1..T compute dW1 = x^T dz1, one 16x16 tile each on the same
//              small-product body, and apply it in the epilogue, W1 = fma(-lr, dW1, W1) -- dW1
//              never exists in memory; a few more workgroups meanwhile sum the head's per-block
//              shares in block order and update W2, b2 and b1 the same way, and workgroup 0
//              writes loss and accuracy, bumps the step counters, gathers the step's small
//              fetched values into the ring slot of this replay (pinned host memory) and stamps
//              it -- no optimizer or gather launch.
//
// Every sum is taken in the order the chain of general kernels takes it (the same device
// functions, or loops written to the same order), so the two paths agree bit for bit; the tests
// hold them to that over 100 steps.
//
// GRADIENT MODE (g_w1 .. g_b2 set): the step of a job whose update is not this file's -- several
// ranks that exchange gradients first, or an optimizer rule with state. The same three launches
// then END in the gradients instead of applying them: launch 3 stores dW1 = x^T dz1 to g_w1 (the
// product's plain epilogue) and its tail stores the summed dW2, db2, db1 to g_w2, g_b2, g_b1 --
// straight into the slots of the runtime's flat gradient arena when the session hands those out --
// and still gathers loss, accuracy and the other small fetched values for the host (a fetched
// step counter as old value + its coming increment, pack_add). What follows is the collective on
// the arena and ONE update launch (multi_sgd / multi_optim with the counters): four kernels and a
// collective per step instead of the seven of the general chain.
#include "common.h"
#include "gemm_f32_body.h"
#include "mlp_head_body.h"
#include "step_ring.h"

namespace {

using namespace dtf_gemm;
using namespace dtf_head;

constexpr int MAX_PACK = 16;
constexpr int MAX_COUNTERS = 4;

struct TailArgs {
  float* w2;                  // [H, C] updated in place
  float* b2;                  // [C]
  float* b1;                  // [H]
  float lr_w2, lr_b2, lr_b1;
  float* counter[MAX_COUNTERS];
  float delta[MAX_COUNTERS];
  int ncounters;
  int blocks;                 // row blocks of the head launch
  // small values gathered for the host once everything above is written
  const float* src[MAX_PACK];
  float add[MAX_PACK];        // added to every word of part i (a counter fetched ahead of its bump)
  int offset[MAX_PACK + 1];
  int npack;
  float* pack_dst;            // without a ring: where the gathered words go (may be null)
  const float* seq_src;       // without a ring: the host's step number ...
  float* seq_dst;             // ... and where to echo it behind the words (may be null)
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void mlp_step_layer1_kernel(SkinnyArgs g, DtfStepRing ring,
                                                                  int a_index, int first) {
  if (ring.ctr) {
    if (first) {
      const unsigned seq = dtf_ring_begin(ring);
      if (a_index >= 0) g.A = (const float*)dtf_ring_feed(ring, seq, a_index);
    } else if (a_index >= 0) {
      g.A = (const float*)dtf_ring_feed_later(ring, a_index);
    }
  }
  gemm_f32_skinny_block<true, false, 0, NW>(g, blockIdx.x);
}

__global__ __launch_bounds__(HT) void mlp_step_head_kernel(HeadArgs p, DtfStepRing ring,
                                                           int labels_index) {
  if (ring.ctr && labels_index >= 0)
    p.labels = (const float*)dtf_ring_feed_later(ring, labels_index);
  head_rows_block(p, blockIdx.x);
}

// The first workgroups of the update launch (256 threads each). Workgroup 0 finishes loss and
// accuracy, bumps the counters and hands the step's small values to the host; workgroups
// 1 .. tail_blocks - 1 sum the head's shares of dW2, db2 and db1, one element per thread, and
// update those parameters. (A gathered value must therefore not be one of W2, b2, b1: the
// launcher's caller keeps such fetches out of the list.)
__device__ __forceinline__ void step_tail(const HeadArgs& p, const TailArgs& t,
                                          const DtfStepRing& ring, const int tb) {
  const int H = p.H, C = p.C;
  const int S = head_slice(H, C);
  const int tid = threadIdx.x;
  // element e of the share layout [dW2 | db2 | db1 | loss | hits]: its shares over the row
  // blocks and the parameter it updates, all requested before anything is summed
  const int e = tb == 0 ? (S - 2 + tid) : ((tb - 1) * HT + tid);
  const bool on = tb == 0 ? tid < 2 : e < S - 2;
  float sh[HB / RB];
#pragma unroll
  for (int g = 0; g < HB / RB; ++g) sh[g] = (on && g < t.blocks) ? p.ws[(long)g * S + e] : 0.f;
  float* param = nullptr;
  float lr = 0.f;
  if (on && tb > 0 && t.w2) {
    if (e < H * C) { param = t.w2 + e; lr = t.lr_w2; }
    else if (e < H * C + C) { param = t.b2 + (e - H * C); lr = t.lr_b2; }
    else { param = t.b1 + (e - H * C - C); lr = t.lr_b1; }
  }
  const float old = param ? *param : 0.f;
  float v = 0.f;                                   // block order, from zero
#pragma unroll
  for (int g = 0; g < HB / RB; ++g) v += sh[g];
  if (tb > 0) {
    if (!on) return;
    if (e < H * C) { if (p.dw) p.dw[e] = v; }
    else if (e < H * C + C) { if (p.db) p.db[e - H * C] = v; }
    else if (p.db_prev) p.db_prev[e - H * C - C] = v;
    if (param) *param = __builtin_fmaf(-lr, v, old);
    return;
  }
  if (tid == 0) p.loss[0] = v / (float)p.B;
  else if (tid == 1 && p.accuracy) p.accuracy[0] = v / (float)p.B;
  if (tid < t.ncounters) t.counter[tid][0] += t.delta[tid];
  if (t.npack == 0 && !ring.ctr && !t.seq_dst) return;
  __syncthreads();                 // loss, accuracy and counters are written
  unsigned seq = 0;
  float* dst = t.pack_dst;
  if (ring.ctr) {
    seq = dtf_ring_current(ring);
    dst = dtf_ring_out(ring, seq);
  }
  if (dst) {
    for (int i = 0; i < t.npack; ++i) {
      const int n = t.offset[i + 1] - t.offset[i];
      for (int j = tid; j < n; j += HT) dst[t.offset[i] + j] = t.src[i][j] + t.add[i];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    if (ring.ctr) {
      dtf_ring_finish(ring, seq);
    } else if (t.seq_dst) {
      const float s = __builtin_nontemporal_load(t.seq_src);
      __hip_atomic_store(t.seq_dst, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// APPLY: W1 = fma(-lr, dW1, W1) in the product's epilogue (g.C = W1); otherwise dW1 is stored
// (g.C = its destination) and the tail stores the small gradients instead of applying them.
template <bool APPLY>
__global__ __launch_bounds__(NTHREADS) void mlp_step_update_kernel(SkinnyArgs g, HeadArgs p,
                                                                   TailArgs t, DtfStepRing ring,
                                                                   int a_index, int tail_blocks,
                                                                   int ring_last) {
  if ((int)blockIdx.x < tail_blocks) {
    step_tail(p, t, ring_last ? ring : DtfStepRing{nullptr, nullptr, nullptr, 0, -1}, blockIdx.x);
    return;
  }
  if (ring.ctr && a_index >= 0)
    g.A = (const float*)dtf_ring_feed_later(ring, a_index);
  // dW1 = x^T dz1: A is x stored [K = batch, M = inputs] (M-contiguous), B is dz1 [K, N]
  gemm_f32_skinny_block<false, false, APPLY ? 1 : 0>(g, (int)blockIdx.x - tail_blocks);
}

}  // namespace

struct DtfMlpStep {
  // model (all fp32, contiguous)
  const float* x;        // [B, D] (ignored with a ring and x_index >= 0)
  const float* labels;   // [B, C] (ignored with a ring and labels_index >= 0)
  float* w1;             // [D, H]
  float* b1;             // [H]
  float* w2;             // [H, C]
  float* b2;             // [C]
  long B, D, H, C;
  float lr_w1, lr_b1, lr_w2, lr_b2;
  float dloss;           // d(loss): 1 for a plain minimize()
  // results
  float* loss;           // scalar
  float* accuracy;       // scalar or null
  float* y;              // [B, C] softmax or null
  // scratch
  float* a;              // [B, H] hidden activations
  float* dz1;            // [B, H]
  float* head_ws;        // dtf_mlp_head_workspace(B, H, C) floats
  // step counters bumped by the step
  int ncounters;
  float* counter[4];
  float delta[4];
  // small values gathered for the host at the end of the step
  int npack;
  const float* pack_src[16];
  int pack_size[16];
  float* pack_dst;       // without a ring (may be null: no gather)
  const float* seq_src;  // without a ring: completion echo (may be null)
  float* seq_dst;
  // step ring (ring_ctr null: none)
  const void* ring_table;
  void* ring_ctr;
  void* ring_out;
  int ring;
  int x_index, labels_index;   // table entries of x and labels (-1: the pointers above)
  int ring_first;              // this step's first launch is the first kernel of the replay
  int ring_last;               // its last launch gathers into the ring, stamps the slot and opens
                               // the next replay (otherwise a later kernel of the step does)
  long host_seq;               // >= 0: the replay number, for a step issued directly (the device
                               // counters are then not used); -1: read it from the counters
  // gradient mode (all four set, or none): where the step leaves dW1 [D, H], db1 [H], dW2 [H, C],
  // db2 [C] instead of applying them; parameters and counters are then not written
  float* g_w1;
  float* g_b1;
  float* g_w2;
  float* g_b2;
  float pack_add[16];          // addend of every word of packed part i
};

DTF_API int dtf_mlp_step_supported(long B, long D, long H, long C) {
  return D >= 1 && B >= 1 && H >= 1 && C >= 1 && B <= HB && H <= HH && C <= HC &&
         D <= (1L << 20);
}

// The three launches described at the top of this file, on `stream`.
DTF_API int dtf_mlp_step_train(const DtfMlpStep* s, hipStream_t stream) {
  if (!s) return DTF_ERR_BAD_ARG;
  const bool ringed = s->ring_ctr != nullptr;
  if (!dtf_mlp_step_supported(s->B, s->D, s->H, s->C)) return DTF_ERR_UNSUPPORTED;
  if (!s->w1 || !s->b1 || !s->w2 || !s->b2 || !s->loss || !s->a || !s->dz1 || !s->head_ws)
    return DTF_ERR_BAD_ARG;
  const bool grads = s->g_w1 || s->g_b1 || s->g_w2 || s->g_b2;
  if (grads && !(s->g_w1 && s->g_b1 && s->g_w2 && s->g_b2)) return DTF_ERR_BAD_ARG;
  if (grads && s->ncounters != 0) return DTF_ERR_BAD_ARG;     // the update launch bumps them
  const bool x_direct = !(ringed && s->x_index >= 0);
  const bool l_direct = !(ringed && s->labels_index >= 0);
  if ((x_direct && !s->x) || (l_direct && !s->labels)) return DTF_ERR_BAD_ARG;
  if (ringed && (!s->ring_table || !s->ring_out || s->ring < 1 || (s->ring & (s->ring - 1)) ||
                 s->x_index >= DTF_RING_FEEDS || s->labels_index >= DTF_RING_FEEDS))
    return DTF_ERR_BAD_ARG;
  if (s->ncounters < 0 || s->ncounters > MAX_COUNTERS || s->npack < 0 || s->npack > MAX_PACK)
    return DTF_ERR_BAD_ARG;
  const int B = (int)s->B, D = (int)s->D, H = (int)s->H, C = (int)s->C;
  const DtfStepRing ring{(const unsigned long long*)s->ring_table, (unsigned*)s->ring_ctr,
                         (float*)s->ring_out, s->ring, s->host_seq};

  TailArgs tail = {};
  tail.w2 = grads ? nullptr : s->w2;
  tail.b2 = grads ? nullptr : s->b2;
  tail.b1 = grads ? nullptr : s->b1;
  tail.lr_w2 = s->lr_w2;
  tail.lr_b2 = s->lr_b2;
  tail.lr_b1 = s->lr_b1;
  tail.ncounters = s->ncounters;
  for (int j = 0; j < s->ncounters; ++j) {
    if (!s->counter[j]) return DTF_ERR_BAD_ARG;
    tail.counter[j] = s->counter[j];
    tail.delta[j] = s->delta[j];
  }
  const int blocks = (B + RB - 1) / RB;
  tail.blocks = blocks;
  int off = 0;
  for (int i = 0; i < s->npack; ++i) {
    if (!s->pack_src[i] || s->pack_size[i] < 0) return DTF_ERR_BAD_ARG;
    tail.src[i] = s->pack_src[i];
    tail.add[i] = s->pack_add[i];
    tail.offset[i] = off;
    off += s->pack_size[i];
  }
  tail.offset[s->npack] = off;
  if (off > DTF_RING_WORDS - 1) return DTF_ERR_BAD_ARG;
  tail.npack = s->npack;
  tail.pack_dst = s->pack_dst;
  tail.seq_src = s->seq_src;
  tail.seq_dst = s->seq_src ? s->seq_dst : nullptr;
  if (!ringed && s->npack > 0 && !s->pack_dst) return DTF_ERR_BAD_ARG;

  // 1. a = sigmoid(x W1 + b1)
  {
    const int tiles_m = (B + SK_T - 1) / SK_T, tiles_n = (H + SK_T - 1) / SK_T;
    const SkinnyArgs g{x_direct ? s->x : nullptr, s->w1, s->a, s->b1, B, H, D, (long)D, (long)H,
                       (long)H, DTF_ACT_SIGMOID, 0.f, tiles_n};
    if (gemm_f32_skinny_waves(D) == 16)
      hipLaunchKernelGGL(mlp_step_layer1_kernel<16>, dim3(tiles_m * tiles_n), dim3(1024), 0, stream,
                         g, ring, x_direct ? -1 : s->x_index, s->ring_first);
    else
      hipLaunchKernelGGL(mlp_step_layer1_kernel<4>, dim3(tiles_m * tiles_n), dim3(256), 0, stream,
                         g, ring, x_direct ? -1 : s->x_index, s->ring_first);
  }
  // 2. head
  const HeadArgs head{s->a, s->w2, s->b2, l_direct ? s->labels : nullptr, nullptr, s->loss, s->y,
                      nullptr, nullptr, s->dz1, nullptr, s->accuracy, s->head_ws, B, H, C, 1,
                      s->dloss, 1};
  hipLaunchKernelGGL(mlp_step_head_kernel, dim3(blocks), dim3(HT), 0, stream, head, ring,
                     l_direct ? -1 : s->labels_index);
  // 3. W1 -= lr x^T dz1, next to the small-parameter tail
  {
    const int tiles_m = (D + SK_T - 1) / SK_T, tiles_n = (H + SK_T - 1) / SK_T;
    const SkinnyArgs g{x_direct ? s->x : nullptr, s->dz1, grads ? s->g_w1 : s->w1, nullptr, D, H,
                       B, (long)D, (long)H, (long)H, DTF_ACT_NONE, s->lr_w1, tiles_n};
    const int tail_blocks = 1 + (H * C + C + H + HT - 1) / HT;
    if (ringed && !s->ring_last && s->npack > 0) return DTF_ERR_BAD_ARG;
    if (grads) {
      HeadArgs sums = head;           // the tail leaves the block-ordered sums here
      sums.dw = s->g_w2;
      sums.db = s->g_b2;
      sums.db_prev = s->g_b1;
      hipLaunchKernelGGL(mlp_step_update_kernel<false>, dim3(tail_blocks + tiles_m * tiles_n),
                         dim3(NTHREADS), 0, stream, g, sums, tail, ring,
                         x_direct ? -1 : s->x_index, tail_blocks, s->ring_last);
    } else {
      hipLaunchKernelGGL(mlp_step_update_kernel<true>, dim3(tail_blocks + tiles_m * tiles_n),
                         dim3(NTHREADS), 0, stream, g, head, tail, ring,
                         x_direct ? -1 : s->x_index, tail_blocks, s->ring_last);
    }
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
