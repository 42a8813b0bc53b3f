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

namespace {

constexpr int HB = 128, HH = 128, HC = 16;
constexpr int RB = 16;                         // rows per workgroup
constexpr int HT = 256;                        // threads per workgroup
constexpr int A_LD = HH + 1;                   // padded row of a in LDS

struct HeadArgs {
  const float* a;        // [B, H] activations of the previous layer
  const float* w;        // [H, C]
  const float* bias;     // [C]
  const float* labels;   // [B, C]
  const float* dloss;    // scalar d(loss)
  float* loss;           // scalar mean loss
  float* y;              // [B, C] softmax
  float* dw;             // [H, C]
  float* db;             // [C]
  float* da;             // [B, H]: da, or dz_prev when sig_grad
  float* db_prev;        // [H] (sig_grad only; may be null)
  float* accuracy;       // scalar mean(argmax y == argmax labels) (may be null)
  float* ws;             // [blocks][H*C + C + H + 2] partial sums
  int B, H, C, sig_grad;
};

__device__ __forceinline__ int head_slice(int H, int C) { return H * C + C + H + 2; }

// sum / max over the 16 lanes of a row group (lanes differ in bits 0-3 only)
__device__ __forceinline__ float row16_sum(float v) {
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 1, 64);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 8, 64));
  v = fmaxf(v, __shfl_xor(v, 4, 64));
  v = fmaxf(v, __shfl_xor(v, 2, 64));
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  return v;
}
// index of the first maximum among the 16 lanes (value v at index `part`)
__device__ __forceinline__ int row16_argmax(float v, int part) {
  int idx = part;
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(v, m, 64);
    const int oi = __shfl_xor(idx, m, 64);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  return idx;
}

__global__ __launch_bounds__(HT) void mlp_head_rows_kernel(HeadArgs p) {
  __shared__ __attribute__((aligned(16))) float a_s[RB * A_LD];
  __shared__ __attribute__((aligned(16))) float w_s[HH * HC];
  __shared__ __attribute__((aligned(16))) float dz_s[RB * HC];
  __shared__ float bias_s[HC];
  __shared__ float red[2 * RB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, H = p.H, C = p.C;
  const int row0 = blockIdx.x * RB;
  float* const slice = p.ws + (long)blockIdx.x * head_slice(H, C);

  // ---- loads: everything this block reads from memory, requested in one batch --------------
  float scale;                                   // d(loss) / B
  {
    float av[RB / 4][2];                         // wave w: local rows w, w+4, w+8, w+12
#pragma unroll
    for (int k = 0; k < RB / 4; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = row0 + wave + 4 * k, h = lane + 64 * q;
        av[k][q] = (r < B && h < H) ? p.a[(long)r * H + h] : 0.f;
      }
    float wv[HH / 16];                           // columns >= C and rows >= H are zero: the
    const int wc = tid & 15, wh = tid >> 4;      // unrolled products run over the padded tile
#pragma unroll
    for (int k = 0; k < HH / 16; ++k) {
      const int h = wh + 16 * k;
      wv[k] = (h < H && wc < C) ? p.w[h * C + wc] : 0.f;
    }
    const int lr = row0 + (tid >> 4);
    const float lv = (lr < B && wc < C) ? p.labels[(long)lr * C + wc] : 0.f;
    const float bv = tid < C ? p.bias[tid] : 0.f;
    scale = p.dloss[0] / (float)B;
#pragma unroll
    for (int k = 0; k < RB / 4; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) a_s[(wave + 4 * k) * A_LD + lane + 64 * q] = av[k][q];
#pragma unroll
    for (int k = 0; k < HH / 16; ++k) w_s[(wh + 16 * k) * HC + wc] = wv[k];
    dz_s[tid] = lv;                              // labels wait in dz's slots until phase 1
    if (tid < HC) bias_s[tid] = bv;
  }
  __syncthreads();

  // ---- phase 1: logits, softmax, loss, dz ---------------------------------------------------
  const int rl = tid >> 4, part = tid & 15;      // local row, lane within the row group
  const int r = row0 + rl;
  float z[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) z[c] = 0.f;
#pragma unroll 2
  for (int h = part; h < H; h += 16) {
    const float av = a_s[rl * A_LD + h];
    const f32x4* wrow = reinterpret_cast<const f32x4*>(w_s + h * HC);
#pragma unroll
    for (int q = 0; q < HC / 4; ++q) {
      const f32x4 wv = wrow[q];
      z[4 * q] += av * wv[0];
      z[4 * q + 1] += av * wv[1];
      z[4 * q + 2] += av * wv[2];
      z[4 * q + 3] += av * wv[3];
    }
  }
  // recursive halving: after the step with mask m a lane keeps the half of its columns whose
  // bit m matches its own, plus the partner's sums for them; lane `part` ends with column `part`
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    const bool up = (part & m) != 0;
#pragma unroll
    for (int i = 0; i < m; ++i) {
      const float keep = up ? z[m + i] : z[i];
      const float send = up ? z[i] : z[m + i];
      z[i] = keep + __shfl_xor(send, m, 64);
    }
  }
  const bool live = part < C;
  const float zc = live ? z[0] + bias_s[part] : -INFINITY;
  const float mx = row16_max(zc);
  const float ex = live ? expf(zc - mx) : 0.f;
  const float s = row16_sum(ex);
  const float inv = 1.f / s, ls = logf(s);
  const float lab = dz_s[rl * HC + part];        // 0 beyond C and beyond B
  const float lsum = row16_sum(lab);
  const float yv = ex * inv;
  const float row_loss = row16_sum(live ? -lab * (zc - mx - ls) : 0.f);
  const int iy = row16_argmax(live ? yv : -INFINITY, part);
  const int il = row16_argmax(live ? lab : -INFINITY, part);
  if (r < B && live) p.y[(long)r * C + part] = yv;
  dz_s[rl * HC + part] = (r < B && live) ? (yv * lsum - lab) * scale : 0.f;
  if (part == 0) {
    red[rl] = r < B ? row_loss : 0.f;
    red[RB + rl] = (r < B && iy == il) ? 1.f : 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    float tl = 0.f, th = 0.f;
#pragma unroll
    for (int k = 0; k < RB; ++k) { tl += red[k]; th += red[RB + k]; }
    slice[H * C + C + H] = tl;
    slice[H * C + C + H + 1] = th;
  }

  // ---- phase 2: partial dW, db over this block's rows -------------------------------------
  {
    const int c = tid & 15, h0 = tid >> 4;
    float dzc[RB];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) dzc[rr] = dz_s[rr * HC + c];
#pragma unroll 2
    for (int k = 0; k < HH / 16; ++k) {
      const int h = h0 + 16 * k;
      if (h >= H) break;
      float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
      for (int rr = 0; rr < RB; rr += 4) {
        acc0 += a_s[rr * A_LD + h] * dzc[rr];
        acc1 += a_s[(rr + 1) * A_LD + h] * dzc[rr + 1];
        acc2 += a_s[(rr + 2) * A_LD + h] * dzc[rr + 2];
        acc3 += a_s[(rr + 3) * A_LD + h] * dzc[rr + 3];
      }
      if (c < C) slice[h * C + c] = (acc0 + acc1) + (acc2 + acc3);
    }
    if (tid < C) {
      float acc = 0.f;
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) acc += dzc[rr];
      slice[H * C + tid] = acc;
    }
  }
  __syncthreads();                               // dW is done reading a_s before it is rewritten

  // ---- phase 3: da = dz W^T (optionally through the sigmoid of the previous layer) -----------
  // wave per row, lanes along H; a lane's W row stays in registers across its rows
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int h = lane + 64 * q;
    if (h >= H) break;
    float wr[HC];
#pragma unroll
    for (int c = 0; c < HC; ++c) wr[c] = w_s[h * HC + c];
#pragma unroll
    for (int k = 0; k < RB / 4; ++k) {
      const int rr = wave + 4 * k;
      const f32x4* dzr = reinterpret_cast<const f32x4*>(dz_s + rr * HC);   // broadcast reads
      float acc = 0.f;
#pragma unroll
      for (int c4 = 0; c4 < HC / 4; ++c4) {
        const f32x4 d = dzr[c4];
        acc += d[0] * wr[4 * c4] + d[1] * wr[4 * c4 + 1] + d[2] * wr[4 * c4 + 2] +
               d[3] * wr[4 * c4 + 3];
      }
      if (p.sig_grad) {
        const float av = a_s[rr * A_LD + h];
        acc = acc * av * (1.f - av);
        a_s[rr * A_LD + h] = acc;
      }
      if (row0 + rr < B) p.da[(long)(row0 + rr) * H + h] = acc;
    }
  }
  if (!p.sig_grad || !p.db_prev) return;
  __syncthreads();

  // ---- phase 4: partial db_prev[h] = sum over this block's rows of dz_prev[r, h] ------------
  if (tid < H) {
    float acc = 0.f;
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) acc += a_s[rr * A_LD + tid];   // rows >= B hold zeros
    slice[H * C + C + tid] = acc;
  }
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

DTF_API int dtf_mlp_head_train(const float* a, const float* w, const float* bias,
                               const float* labels, const float* dloss, float* loss, float* y,
                               float* dw, float* db, float* da, float* db_prev, float* accuracy,
                               float* workspace, long B, long H, long C, int sig_grad,
                               hipStream_t stream) {
  if (!a || !w || !bias || !labels || !dloss || !loss || !y || !dw || !db || !da || !workspace)
    return DTF_ERR_BAD_ARG;
  if (!dtf_mlp_head_supported(B, H, C)) return DTF_ERR_UNSUPPORTED;
  HeadArgs args{a, w, bias, labels, dloss, loss, y, dw, db, da, db_prev, accuracy, workspace,
                (int)B, (int)H, (int)C, sig_grad};
  const int blocks = (int)((B + RB - 1) / RB);
  hipLaunchKernelGGL(mlp_head_rows_kernel, dim3(blocks), dim3(HT), 0, stream, args);
  const int S = (int)(H * C + C + H + 2);
  hipLaunchKernelGGL(mlp_head_finish_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, args,
                     blocks);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
