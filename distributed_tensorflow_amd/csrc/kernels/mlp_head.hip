// Forward + backward of a small softmax classifier head in two short launches.
//
// The reference's second layer and loss (tf_distributed.py:64-70) and everything their
// gradients need -- z = a W + b, y = softmax(z), loss = mean(-sum(labels * log y)), dz, dW, db,
// and the gradient flowing into the previous (sigmoid) layer -- are 0.3 MFLOP on a 100-row
// batch. As separate kernels that is seven launches of a few microseconds each on an otherwise
// idle chip. Done by ONE workgroup the work is issue-bound on a single CU (about 3000 vector
// instructions per wave with 16 waves: 20 us measured), so the batch is cut into blocks of 16
// rows, one workgroup (4 waves) each:
//
//   rows kernel (grid = ceil(B / 16))
//     loads    its 16 rows of a, all of W, its labels and the bias in one batch of independent
//              loads (one memory round trip)
//     phase 1  16 lanes per row accumulate the row's 16 (padded) logits over H; a recursive-
//              halving exchange leaves lane c with logit c (15 shuffles instead of 64); softmax,
//              loss, dz and the argmax-accuracy bit are then one class per lane
//     phase 2  partial dW[h,c] = sum over the block's rows of a[r,h] dz[r,c]; partial db
//     phase 3  da[r,h] = sum_c dz[r,c] W[h,c]; with SIG_GRAD the previous layer's
//              dz_prev = da * a * (1 - a) replaces a[r,h] in LDS, and
//     phase 4  partial db_prev[h] = sum over the block's rows of dz_prev[r,h]
//     the partials (dW, db, db_prev, loss, hits) go to the block's slice of a workspace
//   finish kernel: sums the slices in block order and writes dW, db, db_prev, loss, accuracy
//
// Every sum runs in a fixed order, so results are bit-identical from launch to launch.
// Limits (checked by the launcher): B <= 128, H <= 128, C <= 16.
#include "common.h"
#include "mlp_head_body.h"
#include "step_ring.h"

namespace {

using namespace dtf_head;

__global__ __launch_bounds__(HT) void mlp_head_rows_kernel(HeadArgs p) {
  head_rows_block(p, blockIdx.x);
}

// inside a captured step whose labels are a fed tensor: read where they lie, through the step
// ring's address table (never the step's first ring kernel: the first layer's product is)
__global__ __launch_bounds__(HT) void mlp_head_rows_ring_kernel(HeadArgs p, DtfStepRing ring,
                                                                int labels_index) {
  p.labels = (const float*)dtf_ring_feed_later(ring, labels_index);
  head_rows_block(p, blockIdx.x);
}

// element e of the slice layout [dW (H*C) | db (C) | db_prev (H) | loss | hits], summed over the
// blocks in block order
__global__ __launch_bounds__(256) void mlp_head_finish_kernel(HeadArgs p, int blocks) {
  const int H = p.H, C = p.C;
  const int S = head_slice(H, C);
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= S) return;
  const bool is_prev = e >= H * C + C && e < H * C + C + H;
  if (is_prev && (!p.sig_grad || !p.db_prev)) return;     // those slots were never written
  float t[HB / RB];
#pragma unroll
  for (int g = 0; g < HB / RB; ++g) t[g] = g < blocks ? p.ws[(long)g * S + e] : 0.f;
  float v = 0.f;
#pragma unroll
  for (int g = 0; g < HB / RB; ++g) v += t[g];
  if (e < H * C) p.dw[e] = v;
  else if (e < H * C + C) p.db[e - H * C] = v;
  else if (is_prev) p.db_prev[e - H * C - C] = v;
  else if (e == S - 2) p.loss[0] = v / (float)p.B;
  else if (p.accuracy) p.accuracy[0] = v / (float)p.B;
}

}  // namespace

// 1 when the head kernels can take the shape
DTF_API int dtf_mlp_head_supported(long B, long H, long C) {
  return B >= 1 && H >= 1 && C >= 1 && B <= HB && H <= HH && C <= HC;
}

// floats of workspace dtf_mlp_head_train needs for the shape
DTF_API long dtf_mlp_head_workspace(long B, long H, long C) {
  return ((B + RB - 1) / RB) * (H * C + C + H + 2);
}

// ring_table != null: `labels` is ignored and read through the step ring instead (entry
// labels_index of the replay's table slot)
static int head_train_launch(const float* a, const float* w, const float* bias,
                             const float* labels, const float* dloss, float* loss, float* y,
                             float* dw, float* db, float* da, float* db_prev, float* accuracy,
                             float* workspace, long B, long H, long C, int sig_grad,
                             const DtfStepRing* ring, int labels_index, hipStream_t stream) {
  if (!a || !w || !bias || (!labels && !ring) || !dloss || !loss || !y || !dw || !db || !da ||
      !workspace)
    return DTF_ERR_BAD_ARG;
  if (!dtf_mlp_head_supported(B, H, C)) return DTF_ERR_UNSUPPORTED;
  HeadArgs args{a, w, bias, labels, dloss, loss, y, dw, db, da, db_prev, accuracy, workspace,
                (int)B, (int)H, (int)C, sig_grad, 1.f, db_prev != nullptr};
  const int blocks = (int)((B + RB - 1) / RB);
  if (ring)
    hipLaunchKernelGGL(mlp_head_rows_ring_kernel, dim3(blocks), dim3(HT), 0, stream, args, *ring,
                       labels_index);
  else
    hipLaunchKernelGGL(mlp_head_rows_kernel, dim3(blocks), dim3(HT), 0, stream, args);
  const int S = (int)(H * C + C + H + 2);
  hipLaunchKernelGGL(mlp_head_finish_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, args,
                     blocks);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_mlp_head_train(const float* a, const float* w, const float* bias,
                               const float* labels, const float* dloss, float* loss, float* y,
                               float* dw, float* db, float* da, float* db_prev, float* accuracy,
                               float* workspace, long B, long H, long C, int sig_grad,
                               hipStream_t stream) {
  return head_train_launch(a, w, bias, labels, dloss, loss, y, dw, db, da, db_prev, accuracy,
                           workspace, B, H, C, sig_grad, nullptr, -1, stream);
}

DTF_API int dtf_mlp_head_train_ring(const float* a, const float* w, const float* bias,
                                    const float* dloss, float* loss, float* y, float* dw,
                                    float* db, float* da, float* db_prev, float* accuracy,
                                    float* workspace, long B, long H, long C, int sig_grad,
                                    const void* ring_table, void* ring_ctr, void* ring_out,
                                    int ring, int labels_index, hipStream_t stream) {
  if (!ring_table || !ring_ctr || ring < 1 || (ring & (ring - 1)) || labels_index < 0 ||
      labels_index >= DTF_RING_FEEDS)
    return DTF_ERR_BAD_ARG;
  const DtfStepRing r{(const unsigned long long*)ring_table, (unsigned*)ring_ctr, (float*)ring_out,
                      ring, -1};
  return head_train_launch(a, w, bias, nullptr, dloss, loss, y, dw, db, da, db_prev, accuracy,
                           workspace, B, H, C, sig_grad, &r, labels_index, stream);
}
