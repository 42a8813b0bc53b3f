// One-launch forward + backward of a small softmax classifier head.
//
// The reference's second layer and loss (tf_distributed.py:64-70) and everything their
// gradients need -- z = a W + b, y = softmax(z), loss = mean(-sum(labels * log y)), dz, dW, db,
// and the gradient flowing into the previous (sigmoid) layer -- are 0.3 MFLOP on a 100-row
// batch. As separate kernels that is seven launches of a few microseconds each on an otherwise
// idle chip; here one workgroup keeps the activations (B x H), the weights (H x C) and dz
// (B x C) in LDS and does all of it in one pass:
//
//   phase 1  8 lanes per row accumulate the row's C logits over H, butterfly-reduce, and the
//            row's first lane finishes softmax / loss / dz (and the argmax-accuracy bit)
//   phase 2  dW[h,c] = sum_r a[r,h] dz[r,c] (thread per output), db[c] = sum_r dz[r,c]
//   phase 3  da[r,h] = sum_c dz[r,c] W[h,c]; with SIG_GRAD the previous layer's
//            dz_prev = da * a * (1 - a) replaces a[r,h] in LDS, and
//   phase 4  db_prev[h] = sum_r dz_prev[r,h]
//
// Every sum runs in a fixed order, so results are bit-identical from launch to launch.
// Limits (checked by the launcher): B <= 128, H <= 128, C <= 16.
#include "common.h"

namespace {

constexpr int HB = 128, HH = 128, HC = 16;
constexpr int HT = 1024;                       // threads
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
  int B, H, C, sig_grad;
};

__global__ __launch_bounds__(HT) void mlp_head_kernel(HeadArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* a_s = lds;                            // [HB][A_LD]
  float* w_s = a_s + HB * A_LD;                // [HH][HC]
  float* dz_s = w_s + HH * HC;                 // [HB][HC]
  float* red = dz_s + HB * HC;                 // [HT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int B = p.B, H = p.H, C = p.C;

  for (int rr = wave; rr < B; rr += HT / 64)      // wave per row, lanes along H: coalesced
    for (int h = lane; h < H; h += 64) a_s[rr * A_LD + h] = p.a[rr * H + h];
  for (int e = tid; e < H * HC; e += HT) {      // columns >= C are zero: the unrolled
    const int h = e / HC, c = e % HC;           // products below run over all HC columns
    w_s[e] = c < C ? p.w[h * C + c] : 0.f;
  }
  __syncthreads();

  // ---- phase 1: logits, softmax, loss, dz ---------------------------------------------------
  const int r = tid >> 3, part = tid & 7;
  float z[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) z[c] = 0.f;
  if (r < B) {
#pragma unroll 2
    for (int h = part; h < H; h += 8) {
      const float av = a_s[r * A_LD + h];
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
  }
#pragma unroll
  for (int c = 0; c < HC; ++c) {
    z[c] += __shfl_xor(z[c], 1, 64);
    z[c] += __shfl_xor(z[c], 2, 64);
    z[c] += __shfl_xor(z[c], 4, 64);
  }
  float row_loss = 0.f, row_hit = 0.f;
  if (r < B && part == 0) {
    const float scale = p.dloss[0] / (float)B;
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      z[c] = c < C ? z[c] + p.bias[c] : -INFINITY;
      m = fmaxf(m, z[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      z[c] -= m;
      s += c < C ? expf(z[c]) : 0.f;
    }
    const float inv = 1.f / s, ls = logf(s);
    float lab[HC], lsum = 0.f;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      lab[c] = c < C ? p.labels[r * C + c] : 0.f;
      lsum += lab[c];
    }
    int iy = 0, il = 0;
    float by = -INFINITY, bl = -INFINITY;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
      if (c < C) {
        const float yv = expf(z[c]) * inv;
        p.y[r * C + c] = yv;
        row_loss -= lab[c] * (z[c] - ls);
        dz_s[r * HC + c] = (yv * lsum - lab[c]) * scale;
        if (yv > by) { by = yv; iy = c; }
        if (lab[c] > bl) { bl = lab[c]; il = c; }
      } else {
        dz_s[r * HC + c] = 0.f;
      }
    }
    row_hit = iy == il ? 1.f : 0.f;
  }
  // block sums of loss and hits: lanes by shuffle, waves through LDS, fixed order
  const float wl = dtf_wave_sum(row_loss), wh = dtf_wave_sum(row_hit);
  if (lane == 0) { red[wave] = wl; red[16 + wave] = wh; }
  __syncthreads();
  if (tid == 0) {
    float tl = 0.f, th = 0.f;
#pragma unroll
    for (int k = 0; k < HT / 64; ++k) { tl += red[k]; th += red[16 + k]; }
    p.loss[0] = tl / (float)B;
    if (p.accuracy) p.accuracy[0] = th / (float)B;
  }

  // ---- phase 2: dW, db --------------------------------------------------------------------
  for (int e = tid; e < H * C; e += HT) {
    const int h = e / C, c = e % C;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;   // four loads in flight per operand
    int rr = 0;
    for (; rr + 3 < B; rr += 4) {
      acc0 += a_s[rr * A_LD + h] * dz_s[rr * HC + c];
      acc1 += a_s[(rr + 1) * A_LD + h] * dz_s[(rr + 1) * HC + c];
      acc2 += a_s[(rr + 2) * A_LD + h] * dz_s[(rr + 2) * HC + c];
      acc3 += a_s[(rr + 3) * A_LD + h] * dz_s[(rr + 3) * HC + c];
    }
    for (; rr < B; ++rr) acc0 += a_s[rr * A_LD + h] * dz_s[rr * HC + c];
    p.dw[e] = (acc0 + acc1) + (acc2 + acc3);
  }
  if (tid < C) {
    float acc = 0.f;
    for (int rr = 0; rr < B; ++rr) acc += dz_s[rr * HC + tid];
    p.db[tid] = acc;
  }
  __syncthreads();                              // dW is done reading a_s before it is rewritten

  // ---- phase 3: da = dz W^T (optionally through the sigmoid of the previous layer) -----------
  // wave per row, lanes along H; a lane's W row(s) stay in registers across all its rows
  for (int h = lane; h < H; h += 64) {
    float wr[HC];
#pragma unroll
    for (int c = 0; c < HC; ++c) wr[c] = w_s[h * HC + c];
#pragma unroll 2
    for (int rr = wave; rr < B; rr += HT / 64) {
      const f32x4* dzr = reinterpret_cast<const f32x4*>(dz_s + rr * HC);   // broadcast reads
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < HC / 4; ++q) {
        const f32x4 d = dzr[q];
        acc += d[0] * wr[4 * q] + d[1] * wr[4 * q + 1] + d[2] * wr[4 * q + 2] +
               d[3] * wr[4 * q + 3];
      }
      if (p.sig_grad) {
        const float av = a_s[rr * A_LD + h];
        acc = acc * av * (1.f - av);
        a_s[rr * A_LD + h] = acc;
      }
      p.da[rr * H + h] = acc;
    }
  }
  if (!p.sig_grad || !p.db_prev) return;
  __syncthreads();

  // ---- phase 4: db_prev[h] = sum_r dz_prev[r, h]: 8 row groups per column -------------------
  {
    const int h = tid & 127, g = tid >> 7;
    float acc = 0.f;
    if (h < H) {
#pragma unroll 4
      for (int rr = g; rr < B; rr += 8) acc += a_s[rr * A_LD + h];
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < H) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[k * 128 + tid];
      p.db_prev[tid] = t;
    }
  }
}

}  // namespace

// 1 when the one-workgroup head kernel can take the shape
DTF_API int dtf_mlp_head_supported(long B, long H, long C) {
  return B >= 1 && H >= 1 && C >= 1 && B <= HB && H <= HH && C <= HC;
}

DTF_API int dtf_mlp_head_train(const float* a, const float* w, const float* bias,
                               const float* labels, const float* dloss, float* loss, float* y,
                               float* dw, float* db, float* da, float* db_prev, float* accuracy,
                               long B, long H, long C, int sig_grad, hipStream_t stream) {
  if (!a || !w || !bias || !labels || !dloss || !loss || !y || !dw || !db || !da)
    return DTF_ERR_BAD_ARG;
  if (!dtf_mlp_head_supported(B, H, C)) return DTF_ERR_UNSUPPORTED;
  const size_t lds = sizeof(float) * (HB * A_LD + HH * HC + HB * HC + HT);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)mlp_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)lds);
    attr_set = true;
  }
  HeadArgs args{a, w, bias, labels, dloss, loss, y, dw, db, da, db_prev, accuracy,
                (int)B, (int)H, (int)C, sig_grad};
  hipLaunchKernelGGL(mlp_head_kernel, dim3(1), dim3(HT), lds, stream, args);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
