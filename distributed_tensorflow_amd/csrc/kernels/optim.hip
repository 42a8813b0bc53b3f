// Multi-tensor optimizer rules with per-variable state (slots): momentum, Adagrad, RMSProp, Adam.
//
// The reference trains with plain gradient descent (tf_distributed.py:75-76; multi_sgd_kernel in
// sgd_philox.hip). The rules here are the other dense updates of the same optimizer family, in
// the same form: every variable of a training step is updated by ONE launch (grid.y selects the
// tensor), each element is read and written once, 16 bytes per lane, and the update is spelled
// with explicit fused multiply-adds so that it rounds the same wherever it runs.
//
//   momentum : a = mu a + g;                 w -= lr a            (Nesterov: w -= lr (g + mu a))
//   adagrad  : a += g g;                     w -= lr g / sqrt(a)
//   rmsprop  : s += (1 - rho)(g g - s);      m = mu m + lr g / sqrt(s + eps);   w -= m
//   adam     : m += (1 - b1)(g - m);         v += (1 - b2)(g g - v);
//              w -= lr sqrt(1 - b2^t) / (1 - b1^t) m / (sqrt(v) + eps)
//   ema      : w -= rate (w - g)             (tf.train.ExponentialMovingAverage: w is the shadow,
//              g the variable it follows, rate = 1 - decay; difference, product and subtraction
//              each rounded on their own, as the three element-wise ops this replaces round them)
//
// Adam's b1^t and b2^t live in two device scalars (as TensorFlow keeps them in two variables):
// the update launch only READS them, and a one-wave launch behind it (affine_scalars_kernel, which
// also bumps the step counters) multiplies them by b1 and b2, so a captured step replays without any host-side step count.
#include "common.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_OPT_TENSORS = 16;

enum Rule { MOMENTUM = 0, NESTEROV = 1, ADAGRAD = 2, RMSPROP = 3, ADAM = 4, SGD = 5, ADADELTA = 6,
            EMA = 7 };

struct OptTensor {
  float* w;
  const float* g;
  float* s0;
  float* s1;
  long n;
};
struct OptArgs {
  OptTensor t[MAX_OPT_TENSORS];
  float h[6];              // rule-specific constants, see update()
  const float* pw0;        // adam: b1^t, b2^t
  const float* pw1;
  const float* lr;         // when set: the learning rate, read from the device (a schedule)
  int num_tensors;
};

// One element. h: momentum {lr, mu}; adagrad {lr}; rmsprop {lr, rho, mu, eps, 1 - rho};
// adam {lr_t (bias-corrected), b1, b2, eps, 1 - b1, 1 - b2}; adadelta {lr, rho, eps, 1 - rho}. The
// complements arrive from the host,
// rounded once from double (1.f - 0.999f is off by 5e-5 of its value).
template <int RULE>
__device__ __forceinline__ void update(float& w, float g, float& s0, float& s1, const float* h) {
  if (RULE == SGD) {
    w = __builtin_fmaf(-h[0], g, w);
  } else if (RULE == EMA) {
    float m = (w - g) * h[0];
    asm volatile("" : "+v"(m));          // a rounded product of its own, not an fma into w
    w -= m;
  } else if (RULE == MOMENTUM || RULE == NESTEROV) {
    s0 = __builtin_fmaf(h[1], s0, g);
    if (RULE == NESTEROV) w = __builtin_fmaf(-h[0], __builtin_fmaf(h[1], s0, g), w);
    else w = __builtin_fmaf(-h[0], s0, w);
  } else if (RULE == ADAGRAD) {
    s0 = __builtin_fmaf(g, g, s0);
    w = __builtin_fmaf(-h[0], g / __builtin_sqrtf(s0), w);
  } else if (RULE == RMSPROP) {
    s0 = __builtin_fmaf(h[4], __builtin_fmaf(g, g, -s0), s0);
    s1 = __builtin_fmaf(h[2], s1, h[0] * g / __builtin_sqrtf(s0 + h[3]));
    w -= s1;
  } else if (RULE == ADADELTA) {
    s0 = __builtin_fmaf(h[3], __builtin_fmaf(g, g, -s0), s0);
    const float u = __builtin_sqrtf(s1 + h[2]) / __builtin_sqrtf(s0 + h[2]) * g;
    s1 = __builtin_fmaf(h[3], __builtin_fmaf(u, u, -s1), s1);
    w = __builtin_fmaf(-h[0], u, w);
  } else {
    s0 = __builtin_fmaf(h[4], g - s0, s0);
    s1 = __builtin_fmaf(h[5], __builtin_fmaf(g, g, -s1), s1);
    w = __builtin_fmaf(-h[0], s0 / (__builtin_sqrtf(s1) + h[3]), w);
  }
}

// grid = (blocks_x, num_tensors)
template <int RULE>
__global__ __launch_bounds__(BLOCK) void multi_optim_kernel(OptArgs args) {
  const OptTensor t = args.t[blockIdx.y];
  constexpr bool TWO = RULE == RMSPROP || RULE == ADAM || RULE == ADADELTA;
  constexpr bool NONE = RULE == SGD || RULE == EMA;
  float h[6] = {args.h[0], args.h[1], args.h[2], args.h[3], args.h[4], args.h[5]};
  if (args.lr) h[0] = args.lr[0];
  if (RULE == ADAM) {
    // bias correction from the running powers (read only: the launch behind this one advances them)
    const float p1 = args.pw0[0], p2 = args.pw1[0];
    h[0] = h[0] * __builtin_sqrtf(1.f - p2) / (1.f - p1);
  }
  const long start = blockIdx.x * (long)BLOCK + threadIdx.x, step = (long)gridDim.x * BLOCK;
  uintptr_t bits = (uintptr_t)t.w | (uintptr_t)t.g;
  if (!NONE) bits |= (uintptr_t)t.s0;
  if (TWO) bits |= (uintptr_t)t.s1;
  long done = 0;
  if ((bits & 15) == 0) {
    const long n4 = t.n >> 2;
    f32x4* w4 = reinterpret_cast<f32x4*>(t.w);
    const f32x4* g4 = reinterpret_cast<const f32x4*>(t.g);
    f32x4* a4 = reinterpret_cast<f32x4*>(t.s0);
    f32x4* b4 = reinterpret_cast<f32x4*>(t.s1);
    for (long i = start; i < n4; i += step) {
      const f32x4 g = g4[i];
      f32x4 w = w4[i], a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
      if (!NONE) a = a4[i];
      if (TWO) b = b4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float wj = w[j], aj = a[j], bj = b[j];
        update<RULE>(wj, g[j], aj, bj, h);
        w[j] = wj;
        a[j] = aj;
        b[j] = bj;
      }
      w4[i] = w;
      if (!NONE) a4[i] = a;
      if (TWO) b4[i] = b;
    }
    done = n4 << 2;
  }
  for (long i = done + start; i < t.n; i += step) {
    float w = t.w[i], a = NONE ? 0.f : t.s0[i], b = TWO ? t.s1[i] : 0.f;
    update<RULE>(w, t.g[i], a, b, h);
    t.w[i] = w;
    if (!NONE) t.s0[i] = a;
    if (TWO) t.s1[i] = b;
  }
}

struct AffineArgs {
  float* p[8];
  float f[8];
  float d[8];
  int n;
};

__global__ void affine_scalars_kernel(AffineArgs args) {
  if (threadIdx.x < (unsigned)args.n)
    args.p[threadIdx.x][0] =
        __builtin_fmaf(args.p[threadIdx.x][0], args.f[threadIdx.x], args.d[threadIdx.x]);
}

int grid_x(long n) {
  long blocks = (n + (long)BLOCK * 4 - 1) / ((long)BLOCK * 4);
  return (int)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks);
}

}  // namespace

// In-place update of `count` (<= 16) fp32 tensors by `rule` (0 momentum, 1 Nesterov momentum,
// 2 Adagrad, 3 RMSProp, 4 Adam, 5 plain gradient descent: no state, 6 Adadelta, 7 moving average:
// w follows g, no state). lr: optional device scalar
// that replaces hyper[0] when the kernel runs. s0 / s1: the rule's per-element state (s1 only for RMSProp,
// Adam and Adadelta); hyper: the rule's constants in the order documented at update() (complements
// included); pw0 / pw1: device
// scalars holding b1^t and b2^t (Adam only; not modified here).
DTF_API int dtf_multi_optim_f32(int rule, int count, float* const* w, const float* const* g,
                                float* const* s0, float* const* s1, const long* n,
                                const float* hyper, int num_hyper, const float* pw0,
                                const float* pw1, const float* lr, hipStream_t stream) {
  const bool none = rule == SGD || rule == EMA;
  if (rule < 0 || rule > EMA || count < 0 || count > MAX_OPT_TENSORS || num_hyper < 0 ||
      num_hyper > 6 || (count > 0 && (!w || !g || (!none && !s0) || !n)) ||
      (num_hyper > 0 && !hyper))
    return DTF_ERR_BAD_ARG;
  if (count == 0) return DTF_OK;
  const bool two = rule == RMSPROP || rule == ADAM || rule == ADADELTA;
  if (two && !s1) return DTF_ERR_BAD_ARG;
  if (rule == ADAM && (!pw0 || !pw1)) return DTF_ERR_BAD_ARG;
  OptArgs args = {};
  long max_n = 1;
  for (int i = 0; i < count; ++i) {
    if (!w[i] || !g[i] || (!none && !s0[i]) || (two && !s1[i]) || n[i] < 0)
      return DTF_ERR_BAD_ARG;
    args.t[i] = OptTensor{w[i], g[i], none ? nullptr : s0[i], two ? s1[i] : nullptr, n[i]};
    if (n[i] > max_n) max_n = n[i];
  }
  for (int j = 0; j < num_hyper; ++j) args.h[j] = hyper[j];
  args.pw0 = pw0;
  args.pw1 = pw1;
  args.lr = lr;
  args.num_tensors = count;
  const dim3 grid(grid_x(max_n), count), block(BLOCK);
  switch (rule) {
    case MOMENTUM: hipLaunchKernelGGL(multi_optim_kernel<MOMENTUM>, grid, block, 0, stream, args); break;
    case NESTEROV: hipLaunchKernelGGL(multi_optim_kernel<NESTEROV>, grid, block, 0, stream, args); break;
    case ADAGRAD: hipLaunchKernelGGL(multi_optim_kernel<ADAGRAD>, grid, block, 0, stream, args); break;
    case RMSPROP: hipLaunchKernelGGL(multi_optim_kernel<RMSPROP>, grid, block, 0, stream, args); break;
    case SGD: hipLaunchKernelGGL(multi_optim_kernel<SGD>, grid, block, 0, stream, args); break;
    case ADADELTA: hipLaunchKernelGGL(multi_optim_kernel<ADADELTA>, grid, block, 0, stream, args); break;
    case EMA: hipLaunchKernelGGL(multi_optim_kernel<EMA>, grid, block, 0, stream, args); break;
    default: hipLaunchKernelGGL(multi_optim_kernel<ADAM>, grid, block, 0, stream, args); break;
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// p[i][0] = p[i][0] * f[i] + d[i] for `count` (<= 8) device scalars in one one-wave launch: Adam's
// running powers (f = b1 / b2, d = 0) and step counters (f = 1, d = increment) of a training step.
DTF_API int dtf_affine_scalars_f32(int count, float* const* p, const float* f, const float* d,
                                   hipStream_t stream) {
  if (count < 0 || count > 8 || (count > 0 && (!p || !f || !d))) return DTF_ERR_BAD_ARG;
  if (count == 0) return DTF_OK;
  AffineArgs args = {};
  for (int i = 0; i < count; ++i) {
    if (!p[i]) return DTF_ERR_BAD_ARG;
    for (int j = 0; j < i; ++j)
      if (p[j] == p[i]) return DTF_ERR_BAD_ARG;      // one lane per scalar: no scalar twice
    args.p[i] = p[i];
    args.f[i] = f[i];
    args.d[i] = d[i];
  }
  args.n = count;
  hipLaunchKernelGGL(affine_scalars_kernel, dim3(1), dim3(64), 0, stream, args);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
