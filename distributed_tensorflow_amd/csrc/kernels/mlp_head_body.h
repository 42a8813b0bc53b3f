// Device-side body of the small classifier head (see mlp_head.hip for the design), shared by
// the stand-alone head kernels and the three-launch MLP training step in mlp_step.hip.
#pragma once

#include "common.h"

namespace dtf_head {


constexpr int HB = 128, HH = 128, HC = 16;
constexpr int RB = 16;                         // rows per workgroup
constexpr int HT = 256;                        // threads per workgroup
constexpr int A_LD = HH + 4;                   // padded rows in LDS: a, W and dz
constexpr int W_LD = HC + 1;
constexpr int DZ_LD = HC + 1;

struct HeadArgs {
  const float* a;        // [B, H] activations of the previous layer
  const float* w;        // [H, C]
  const float* bias;     // [C]
  const float* labels;   // [B, C]
  const float* dloss;    // scalar d(loss)
  float* loss;           // scalar mean loss
  float* y;              // [B, C] softmax (may be null)
  float* dw;             // [H, C]
  float* db;             // [C]
  float* da;             // [B, H]: da, or dz_prev when sig_grad
  float* db_prev;        // [H] (sig_grad only; may be null)
  float* accuracy;       // scalar mean(argmax y == argmax labels) (may be null)
  float* ws;             // [blocks][H*C + C + H + 2] partial sums
  int B, H, C, sig_grad;
  float dloss_const;     // d(loss) when dloss is null
  int need_db_prev;      // sig_grad: also leave the shares of db_prev in the workspace
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

// One workgroup (4 waves) = 16 rows of the batch. All four products of the head run on the
// exact-fp32 v_mfma_f32_16x16x4_f32 (A[m = l&15][k = l>>4], B[k = l>>4][n = l&15]; C/D col = l&15,
// row = 4 (l>>4) + i), operands read from LDS one word per lane:
//   logits[16, 16]   = a[16, 128] W[128, 16]       wave w sums h in [32w, 32w + 32), 8 MFMAs;
//                                                   the four partial tiles are added in wave order
//   dW share[128,16] = a^T[128, 16] dz[16, 16]     wave w: h-tiles 2w and 2w+1, 4 MFMAs each
//   da[16, 128]      = dz[16, 16] W^T[16, 128]     wave w: h-tiles 2w and 2w+1, 4 MFMAs each
// (rows >= B, hidden units >= H and classes >= C are zeros in LDS, so the padded tiles add
// nothing). Between the first and the second, softmax / loss / dz / the accuracy bit are one
// class per lane, 16 lanes per row.
__device__ __forceinline__ void head_rows_block(const HeadArgs& p, const int bid) {
  __shared__ float a_s[RB * A_LD];
  __shared__ float w_s[HH * W_LD];
  __shared__ float dz_s[RB * DZ_LD];
  __shared__ float zp[4][RB][HC + 1];
  __shared__ float red[2 * RB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, kq = lane >> 4;
  const int B = p.B, H = p.H, C = p.C;
  const int row0 = bid * RB;
  float* const slice = p.ws + (long)bid * head_slice(H, C);
  const int rl = tid >> 4, part = tid & 15;      // softmax phase: local row, class of this thread
  const int r = row0 + rl;

  // ---- loads: everything this block reads from memory, requested in one batch --------------
  float scale, lab, bv;
  {
    float av[RB / 4][2];                         // wave w: local rows w, w+4, w+8, w+12
#pragma unroll
    for (int k = 0; k < RB / 4; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int gr = row0 + wave + 4 * k, h = lane + 64 * q;
        av[k][q] = (gr < B && h < H) ? p.a[(long)gr * H + h] : 0.f;
      }
    float wv[HH / 16];
    const int wc = tid & 15, wh = tid >> 4;
#pragma unroll
    for (int k = 0; k < HH / 16; ++k) {
      const int h = wh + 16 * k;
      wv[k] = (h < H && wc < C) ? p.w[h * C + wc] : 0.f;
    }
    lab = (r < B && part < C) ? p.labels[(long)r * C + part] : 0.f;
    bv = part < C ? p.bias[part] : 0.f;
    scale = (p.dloss ? p.dloss[0] : p.dloss_const) / (float)B;
#pragma unroll
    for (int k = 0; k < RB / 4; ++k)
#pragma unroll
      for (int q = 0; q < 2; ++q) a_s[(wave + 4 * k) * A_LD + lane + 64 * q] = av[k][q];
#pragma unroll
    for (int k = 0; k < HH / 16; ++k) w_s[(wh + 16 * k) * W_LD + wc] = wv[k];
  }
  __syncthreads();

  // ---- logits: this wave's quarter of H ------------------------------------------------------
  {
    f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int k = 32 * wave + 4 * i + kq;
      z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s[lm * A_LD + k], w_s[k * W_LD + lm], z0, 0, 0, 0);
      z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s[lm * A_LD + k + 4], w_s[(k + 4) * W_LD + lm],
                                                z1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) zp[wave][4 * kq + i][lm] = z0[i] + z1[i];
  }
  __syncthreads();

  // ---- softmax, loss, dz: one class per lane, 16 lanes per row ----------------------------------
  const bool live = part < C;
  const float zsum = ((zp[0][rl][part] + zp[1][rl][part]) + zp[2][rl][part]) + zp[3][rl][part];
  const float zc = live ? zsum + bv : -INFINITY;
  const float mx = row16_max(zc);
  const float ex = live ? expf(zc - mx) : 0.f;
  const float s = row16_sum(ex);
  const float inv = 1.f / s, ls = logf(s);
  const float lsum = row16_sum(lab);
  const float yv = ex * inv;
  const float row_loss = row16_sum(live ? -lab * (zc - mx - ls) : 0.f);
  const int iy = row16_argmax(live ? yv : -INFINITY, part);
  const int il = row16_argmax(live ? lab : -INFINITY, part);
  if (p.y && r < B && live) p.y[(long)r * C + part] = yv;
  dz_s[rl * DZ_LD + part] = (r < B && live) ? (yv * lsum - lab) * scale : 0.f;
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
  if (tid < C) {                                  // db share: the block's rows in order
    float acc = 0.f;
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) acc += dz_s[rr * DZ_LD + tid];
    slice[H * C + tid] = acc;
  }

  // ---- dW share and da: this wave's two h-tiles ---------------------------------------------------
  const int h0 = 32 * wave + lm, h1 = h0 + 16;    // this lane's hidden unit in either tile
  f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};   // dW[h-tile rows 4kq+i][c = lm]
  f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};   // da[rows 4kq+i][h = h0 / h1]
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int k = 4 * st + kq;                    // a batch row for dW, a class for da
    const float dzk = dz_s[k * DZ_LD + lm];
    w0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s[k * A_LD + h0], dzk, w0, 0, 0, 0);
    w1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s[k * A_LD + h1], dzk, w1, 0, 0, 0);
    const float dzr = dz_s[lm * DZ_LD + k];
    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dzr, w_s[h0 * W_LD + k], d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dzr, w_s[h1 * W_LD + k], d1, 0, 0, 0);
  }
  if (lm < C) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ha = 32 * wave + 4 * kq + i, hb = ha + 16;
      if (ha < H) slice[ha * C + lm] = w0[i];
      if (hb < H) slice[hb * C + lm] = w1[i];
    }
  }
  float c0 = 0.f, c1 = 0.f;                       // this lane's rows of dz_prev, summed in order
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rr = 4 * kq + i;
    float v0 = d0[i], v1 = d1[i];
    if (p.sig_grad) {                             // through the sigmoid of the previous layer
      const float a0 = a_s[rr * A_LD + h0], a1 = a_s[rr * A_LD + h1];
      v0 = v0 * a0 * (1.f - a0);
      v1 = v1 * a1 * (1.f - a1);
    }
    c0 += v0;
    c1 += v1;
    if (row0 + rr < B) {
      if (h0 < H) p.da[(long)(row0 + rr) * H + h0] = v0;
      if (h1 < H) p.da[(long)(row0 + rr) * H + h1] = v1;
    }
  }
  if (!p.sig_grad || !p.need_db_prev) return;
  // db_prev share: the lanes holding rows 4..7, 8..11, 12..15 of the same hidden unit are 16, 32
  // and 48 lanes away
  c0 += __shfl_xor(c0, 16, 64);
  c1 += __shfl_xor(c1, 16, 64);
  c0 += __shfl_xor(c0, 32, 64);
  c1 += __shfl_xor(c1, 32, 64);
  if (kq == 0) {
    if (h0 < H) slice[H * C + C + h0] = c0;
    if (h1 < H) slice[H * C + C + h1] = c1;
  }
}

}  // namespace dtf_head
