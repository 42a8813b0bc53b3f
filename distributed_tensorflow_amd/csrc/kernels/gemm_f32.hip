// fp32 GEMM on the gfx950 matrix cores: C = act(op(A) * op(B) + bias).
//
// gfx950 has no TF32/xf32 path, but it has an exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) that
// runs at the vector-FMA rate while leaving the VALU free, so the kernel is built on it.
// This is the kernel behind tf.matmul in the reference graphs: the 1000x1000x1000 product
// (tf_distributed_1000Matrix.py:48), the MLP layers x*W1 [B,784]x[784,100] and a2*W2
// [B,100]x[100,10] (tf_distributed.py:62,64) and their NT / TN gradient products.
//
// Structure: 256 threads = 4 waves in a 2x2 grid over a BMxBN block tile (128x128 for large
// problems, 64x64 when that is what it takes to put a workgroup on every CU), BK = 16.
// Both operand tiles are staged K-major in LDS (As[k][m], Bs[k][n]) so the MFMA fragment read
// -- lane l takes A[m = l&31][k = l>>5], B[k = l>>5][n = l&31] -- is 32 consecutive words per
// half-wave: conflict-free ds_read_b32. An operand that is K-contiguous in memory is
// transposed on its way into LDS (row stride BT+2 keeps those scattered writes
// conflict-free); one that is M/N-contiguous is copied with 16-byte stores. Global loads run
// DEPTH K-tiles ahead of the MFMAs through a ring of register sets (a K-tile is only 8 or 32
// MFMAs of 64 cycles per wave, far less than a global-memory round trip, and a workgroup is
// often alone on its CU): iteration k refills the set tile k just left with tile k+DEPTH,
// runs the MFMAs of tile k and meanwhile moves tile k+1 -- loaded DEPTH-1 iterations ago -- into
// the other LDS buffer: one barrier per K-tile, placed inside the MFMA stream. Shapes need no alignment: edge tiles are
// predicated (1000 = 7*128 + 104, K = 784 / 100 / 10 all occur in the reference).
//
// Split-K: the exact-fp32 MFMA retires 32x32x2 per 64 cycles per wave, so a product with few
// output tiles and a long K (the MLP's x*W1: 100x100 outputs, K = 784 -- four 64x64 tiles)
// leaves the chip idle and takes 392 serial MFMAs per wave. Such problems are cut along K into
// `splits` workgroups per tile, each writing its partial tile to a workspace plane; a second,
// tiny kernel sums the planes in split order (deterministic), applies bias + activation and
// writes C. (A single-kernel variant in which the last workgroup to finish does the reduction
// needs a device-scope release/acquire pair per workgroup; with one L2 per XCD that fence is a
// cache write-back and measured 64 us on the 100x100x784 product against 23 us unsplit.)
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int BK = 16;
constexpr int NTHREADS = 256;

template <int BT, bool KCONTIG>
struct Loader {
  static constexpr int LD = KCONTIG ? BT + 2 : BT;
  static constexpr int PER_THREAD = (BT * BK / 4) / NTHREADS;
  static_assert(PER_THREAD >= 1, "tile too small for 256 threads");

  // X(t, k) = X[t * ld + k] when KCONTIG, X[k * ld + t] otherwise
  __device__ static void fetch(f32x4 (&regs)[PER_THREAD], const float* __restrict__ X, long ld,
                               int t0, int k0, int T, int K, bool vec_ok, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int idx = tid + i * NTHREADS;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (KCONTIG) {
        const int t = idx >> 2, k4 = idx & 3;
        const int gt = t0 + t, gk = k0 + 4 * k4;
        if (gt < T && gk < K) {
          const float* p = X + (long)gt * ld + gk;
          if (vec_ok && gk + 3 < K) {
            v = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (gk + j < K) v[j] = p[j];
          }
        }
      } else {
        constexpr int V = BT / 4;
        const int k = idx / V, t4 = idx % V;
        const int gk = k0 + k, gt = t0 + 4 * t4;
        if (gk < K && gt < T) {
          const float* p = X + (long)gk * ld + gt;
          if (vec_ok && gt + 3 < T) {
            v = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (gt + j < T) v[j] = p[j];
          }
        }
      }
      regs[i] = v;
    }
  }

  // Branch-free fetch for 16-byte-aligned operands (ld % 4 == 0, aligned base, K % 4 == 0 when
  // KCONTIG, T % 4 == 0 otherwise): every lane loads one aligned f32x4 from a CLAMPED address --
  // rows / k-rows past the edge re-read the last valid one -- and elements with k >= k_end are
  // zeroed by selects. Out-of-range rows only feed outputs the epilogue drops. With no branch
  // around the loads the compiler can count them (s_waitcnt vmcnt(N)), which is what lets
  // loads stay in flight across several K-tiles.
  __device__ static void fetch_fast(f32x4 (&regs)[PER_THREAD], const float* __restrict__ X,
                                    long ld, int t0, int k0, int T, int k_end, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int idx = tid + i * NTHREADS;
      if (KCONTIG) {
        const int t = idx >> 2, k4 = idx & 3;
        int gt = t0 + t;
        gt = gt < T - 1 ? gt : T - 1;
        const int gk = k0 + 4 * k4;                      // multiple of 4; k_end is too
        const int ck = gk < k_end - 4 ? gk : k_end - 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(X + (long)gt * ld + ck);
        regs[i] = gk < k_end ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
        constexpr int V = BT / 4;
        const int k = idx / V, t4 = idx % V;
        int gt = t0 + 4 * t4;
        gt = gt < T - 4 ? gt : T - 4;
        const int gk = k0 + k;
        const int ck = gk < k_end - 1 ? gk : k_end - 1;
        f32x4 v = *reinterpret_cast<const f32x4*>(X + (long)ck * ld + gt);
        regs[i] = gk < k_end ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }

  // Interior K-tiles (k0 + BK <= k_end) of aligned operands need neither the k clamp nor the
  // zero select, and what is left of the address -- the lane's row (clamped at the edge) times
  // the leading dimension plus its 16-byte column -- does not change from tile to tile. The
  // lane part is computed once as a 32-bit byte offset and the tile part is a wave-uniform base
  // advanced on the scalar unit, so an interior fetch is a buffer load with no vector ALU work
  // (recomputing row * ld in 64 bits per tile cost more issue time than the tile's MFMAs leave).
  // Needs BT * ld * 4 (KCONTIG) or BK * ld * 4 bytes below 4 GiB.
  __device__ static void lane_offsets(unsigned (&voff)[PER_THREAD], long ld, int t0, int T,
                                      int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int idx = tid + i * NTHREADS;
      if (KCONTIG) {
        const int t = idx >> 2, k4 = idx & 3;
        int gt = t0 + t;
        gt = gt < T - 1 ? gt : T - 1;
        voff[i] = (unsigned)(gt - t0) * (unsigned)(ld * 4) + 16u * k4;
      } else {
        constexpr int V = BT / 4;
        const int k = idx / V, t4 = idx % V;
        const int tb = t0 < T - 4 ? t0 : T - 4;
        int gt = t0 + 4 * t4;
        gt = gt < T - 4 ? gt : T - 4;
        voff[i] = (unsigned)k * (unsigned)(ld * 4) + 4u * (gt - tb);
      }
    }
  }

  // wave-uniform address of element (row t0, k = k0) -- KCONTIG -- or (k-row k0, column
  // min(t0, T - 4)) of the tile
  __device__ static const float* tile_base(const float* __restrict__ X, long ld, int t0, int T,
                                           int k0) {
    if (KCONTIG) return X + (long)t0 * ld + k0;
    return X + (long)k0 * ld + (t0 < T - 4 ? t0 : T - 4);
  }

  __device__ static void fetch_interior(f32x4 (&regs)[PER_THREAD], const float* base,
                                        const unsigned (&voff)[PER_THREAD]) {
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], 0, 0);
      regs[i] = __builtin_bit_cast(f32x4, v);
    }
  }

  __device__ static void stash(const f32x4 (&regs)[PER_THREAD], float* __restrict__ Xs, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int idx = tid + i * NTHREADS;
      if (KCONTIG) {
        const int t = idx >> 2, k4 = idx & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) Xs[(4 * k4 + j) * LD + t] = regs[i][j];
      } else {
        constexpr int V = BT / 4;
        const int k = idx / V, t4 = idx % V;
        *reinterpret_cast<f32x4*>(&Xs[k * LD + 4 * t4]) = regs[i];
      }
    }
  }
};

// ABL (timing experiments, -DDTF_GEMM_EXPERIMENTS only; results are wrong when set): bit 0 = no
// global loads in the K loop, bit 1 = no LDS stash, bit 2 = no barrier, bit 3 = no fragment reads
template <int BM, int BN, bool A_KC, bool B_KC, int ABL = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(
    const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
    const float* __restrict__ bias, int M, int N, int K, long lda, long ldb, long ldc,
    int a_vec, int b_vec, int act, int tiles_n, int ntiles, int splits, int kchunk,
    float* __restrict__ partial) {
  using LA = Loader<BM, A_KC>;
  using LB = Loader<BN, B_KC>;
  constexpr int WM = BM / 2, WN = BN / 2;     // per-wave tile
  constexpr int TM = WM / 32, TN = WN / 32;   // 32x32 MFMA tiles per wave
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LA::LD + LB::LD)];
  constexpr int A_BUF = BK * LA::LD, B_BUF = BK * LB::LD;   // floats per buffer
  float* const As = smem;                                   // As[buf][k][m]
  float* const Bs = smem + 2 * A_BUF;                       // Bs[buf][k][n]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int split = blockIdx.x / ntiles;
  // XCD-aware remap (each XCD's L2 sees a contiguous chunk of the walk) + grouped walk: the
  // ~32-64 tiles resident on one XCD form a block GROUP tile-rows deep, so they share A and B
  // panels instead of streaming a whole tile-row of B each
  const int tile = dtf_xcd_remap(blockIdx.x % ntiles, ntiles);
  constexpr int GROUP = 8;
  const int tiles_m = ntiles / tiles_n;
  const int per_group = GROUP * tiles_n;
  const int first_m = (tile / per_group) * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int m0 = (first_m + (tile % per_group) % gsize) * BM;
  const int n0 = ((tile % per_group) / gsize) * BN;
  const int k_begin = split * kchunk;                 // kchunk is a multiple of BK
  const int k_end = (k_begin + kchunk < K) ? k_begin + kchunk : K;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A wave with a single 32x32 output tile would issue every MFMA on the accumulator the one
  // before it is still writing; two accumulators (even / odd k-pairs, summed before the epilogue)
  // keep consecutive MFMAs independent.
  f32x16 acc_odd[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc_odd[0][r] = 0.f;

  const int nkt = (k_end - k_begin + BK - 1) / BK;
  const int lrow = lane & 31, lk = lane >> 5;
  // aligned operands take the branch-free loads (block-uniform choice, made once)
  const bool fast = a_vec != 0 && b_vec != 0 && k_end - k_begin >= 4 &&
                    lda * 4 * (A_KC ? BM : BK) < (1L << 32) &&
                    ldb * 4 * (B_KC ? BN : BK) < (1L << 32) &&
                    (A_KC ? (K & 3) == 0 : ((M & 3) == 0 && M >= 4)) &&
                    (B_KC ? (K & 3) == 0 : ((N & 3) == 0 && N >= 4));

  // The K loop, instantiated twice (FAST: branch-free loads; otherwise fully predicated ones).
  auto k_loop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr int DEPTH = (BM == 128) ? 3 : 4;        // register sets = K-tiles in flight
    f32x4 ra[DEPTH][LA::PER_THREAD], rb[DEPTH][LB::PER_THREAD];
    unsigned voff_a[LA::PER_THREAD], voff_b[LB::PER_THREAD];
    if (FAST) {
      LA::lane_offsets(voff_a, lda, m0, M, tid);
      LB::lane_offsets(voff_b, ldb, n0, N, tid);
    }
    // any tile: clamped addresses, zeros beyond k_end (K tail, prefetches past the end)
    auto fetch_tile = [&](f32x4 (&qa)[LA::PER_THREAD], f32x4 (&qb)[LB::PER_THREAD], int kt) {
      const int k0 = k_begin + kt * BK;
      if (FAST) {
        LA::fetch_fast(qa, A, lda, m0, k0, M, k_end, tid);
        LB::fetch_fast(qb, B, ldb, n0, k0, N, k_end, tid);
      } else {
        LA::fetch(qa, A, lda, m0, k0, M, k_end, a_vec != 0, tid);
        LB::fetch(qb, B, ldb, n0, k0, N, k_end, b_vec != 0, tid);
      }
    };
    // a tile that lies wholly inside [k_begin, k_end) of aligned operands
    auto fetch_interior_tile = [&](f32x4 (&qa)[LA::PER_THREAD], f32x4 (&qb)[LB::PER_THREAD],
                                   int kt) {
      const int k0 = k_begin + kt * BK;
      LA::fetch_interior(qa, LA::tile_base(A, lda, m0, M, k0), voff_a);
      LB::fetch_interior(qb, LB::tile_base(B, ldb, n0, N, k0), voff_b);
    };
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) fetch_tile(ra[u], rb[u], u);   // tiles past the end: zeros
    LA::stash(ra[0], As, tid);
    LB::stash(rb[0], Bs, tid);
    __syncthreads();

    // MFMA fragments run through a register ring PD k-pairs ahead of the MFMAs and straight
    // across K-tile boundaries. With one 32x32 tile per wave (64x64 block tile) a k-pair is a
    // single 64-cycle MFMA, less than an LDS round trip with four waves reading: PD = 3. The
    // barrier of a K-tile sits PD k-pairs before its end -- the last point before the ring reads
    // the next tile's buffer -- so the waves meet in mid-stream with MFMAs still queued, not at
    // an empty pipe followed by a cold fragment read.
    constexpr int KP = BK / 2;                        // k-pairs per K-tile
    constexpr int PD = (BM == 128) ? 1 : 3;
    constexpr int RING = PD + 1;                      // divides KP: slots line up across tiles
    static_assert(KP % RING == 0, "fragment ring must divide the K-tile");
    float a[RING][TM], b[RING][TN];
    auto read_frag = [&](int slot, const float* as, const float* bs, int kp) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[slot][i] = as[(2 * kp + lk) * LA::LD + 32 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[slot][j] = bs[(2 * kp + lk) * LB::LD + 32 * j];
    };
#pragma unroll
    for (int p = 0; p < PD; ++p) read_frag(p, As + wr * WM + lrow, Bs + wc * WN + lrow, p);

    // DEPTH consecutive K-tiles (ring slots 0..DEPTH-1). INTERIOR: every tile of the group, and
    // the tile each one prefetches, lies wholly inside the K range -- the loads are then buffer
    // loads with loop-invariant lane offsets. The two kinds of fetch live in separate loops: a
    // join of paths that issue different loads makes the compiler drain vmcnt to zero before
    // the next LDS stash, which empties the prefetch ring once per group.
    auto group = [&](auto interior_tag, int kt0) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int kt = kt0 + u;
        if (!INTERIOR && kt >= nkt) break;
        const int cur = kt & 1;
        // set u held tile kt, already in LDS: refill it with tile kt + DEPTH
        if (!(ABL & 1)) {
          if (INTERIOR) fetch_interior_tile(ra[u], rb[u], kt + DEPTH);
          else fetch_tile(ra[u], rb[u], kt + DEPTH);
        }
        const float* as = As + cur * A_BUF + wr * WM + lrow;
        const float* bs = Bs + cur * B_BUF + wc * WN + lrow;
        const float* asn = As + (cur ^ 1) * A_BUF + wr * WM + lrow;
        const float* bsn = Bs + (cur ^ 1) * B_BUF + wc * WN + lrow;
#pragma unroll
        for (int kk = 0; kk < KP; ++kk) {
          if (kk == (KP - PD) / 2 && !(ABL & 2)) {
            // tile kt+1 (loaded DEPTH-1 tiles ago; zeros past the end) goes into the other LDS
            // buffer, whose last readers finished before the previous tile's barrier. Neither
            // this nor the ring's reads below are skipped on the last tile: a branch around LDS
            // operations would force the compiler to drain lgkmcnt to zero at the join instead
            // of counting, and every MFMA would then wait for the newest fragment read.
            LA::stash(ra[(u + 1) % DEPTH], As + (cur ^ 1) * A_BUF, tid);
            LB::stash(rb[(u + 1) % DEPTH], Bs + (cur ^ 1) * B_BUF, tid);
          }
          if (kk == KP - PD && !(ABL & 4)) __syncthreads();
          {
            const int pf = kk + PD;
            if (ABL & 8) {}
            else if (pf < KP) read_frag(pf % RING, as, bs, pf);
            else read_frag(pf % RING, asn, bsn, pf - KP);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (TM * TN == 1 && (kk & 1))
                acc_odd[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk % RING][i], b[kk % RING][j],
                                                                  acc_odd[0], 0, 0, 0);
              else
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk % RING][i], b[kk % RING][j],
                                                                 acc[i][j], 0, 0, 0);
            }
        }
      }
    };
    int kt0 = 0;
    if constexpr (FAST) {
      const int n_full = (k_end - k_begin) / BK;
      for (; kt0 + 2 * DEPTH <= n_full; kt0 += DEPTH) group(std::true_type{}, kt0);
    }
    for (; kt0 < nkt; kt0 += DEPTH) group(std::false_type{}, kt0);
  };
  if (fast) k_loop(std::true_type{});
  else k_loop(std::false_type{});

  if (TM * TN == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] += acc_odd[0][r];
  }
  // C/D fragment: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  if (splits > 1) {
    const long plane = (long)M * N;
    float* const mine = partial + split * plane;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * WN + 32 * j + lrow;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wr * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (row < M) mine[(long)row * N + col] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wc * WN + 32 * j + lrow;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * WM + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < M) C[(long)row * ldc + col] = dtf_apply_act(acc[i][j][r] + bv, act);
      }
    }
}

// C[r, c] = act(sum_s partial[s][r][c] + bias[c]); four columns per thread when N % 4 == 0
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial,
                                                            const float* __restrict__ bias,
                                                            float* __restrict__ C, int M, int N,
                                                            long ldc, int splits, int act) {
  const long plane = (long)M * N;
  if ((N & 3) == 0) {
    const long n4 = plane >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      // planes are summed in split order, eight independent loads at a time
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int sp0 = 0; sp0 < splits; sp0 += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int sp = sp0 + u < splits ? sp0 + u : splits - 1;
          t[u] = reinterpret_cast<const f32x4*>(partial + sp * plane)[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (sp0 + u < splits) v += t[u];
      }
      const long e = i << 2;
      const int row = (int)(e / N), col = (int)(e % N);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        C[(long)row * ldc + col + j] = dtf_apply_act(v[j] + (bias ? bias[col + j] : 0.f), act);
    }
  } else {
    for (long e = blockIdx.x * 256L + threadIdx.x; e < plane; e += (long)gridDim.x * 256) {
      float v = partial[e];
      for (int sp = 1; sp < splits; ++sp) v += partial[sp * plane + e];
      const int row = (int)(e / N), col = (int)(e % N);
      C[(long)row * ldc + col] = dtf_apply_act(v + (bias ? bias[col] : 0.f), act);
    }
  }
}

template <int BM, int BN>
int launch(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
           long lda, long ldb, long ldc, int trans_a, int trans_b, int act, int splits,
           int kchunk, float* partial, hipStream_t stream) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int a_vec = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
  const int b_vec = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
  dim3 grid(ntiles * splits), block(NTHREADS);
  // op(A)(m,k): stored [M,K] -> K-contiguous; stored [K,M] (trans_a) -> M-contiguous
#define DTF_GEMM_CASE(AKC, BKC)                                                               \
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AKC, BKC>), grid, block, 0, stream, A, B, C,    \
                     bias, M, N, K, lda, ldb, ldc, a_vec, b_vec, act, tiles_n, ntiles, splits,  \
                     kchunk, partial)
#ifdef DTF_GEMM_EXPERIMENTS
  if (const char* e = getenv("DTF_F32_ABL")) {
    const int abl = atoi(e);
#define DTF_ABL_CASE(V)                                                                       \
  if (abl == V) {                                                                             \
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false, V>), grid, block, 0, stream, A,  \
                       B, C, bias, M, N, K, lda, ldb, ldc, a_vec, b_vec, act, tiles_n,        \
                       ntiles, splits, kchunk, partial);                                      \
    return DTF_OK;                                                                            \
  }
    DTF_ABL_CASE(1) DTF_ABL_CASE(2) DTF_ABL_CASE(4) DTF_ABL_CASE(8) DTF_ABL_CASE(6)
    DTF_ABL_CASE(7) DTF_ABL_CASE(15) DTF_ABL_CASE(14)
#undef DTF_ABL_CASE
  }
#endif
  if (!trans_a && !trans_b) DTF_GEMM_CASE(true, false);
  else if (!trans_a && trans_b) DTF_GEMM_CASE(true, true);
  else if (trans_a && !trans_b) DTF_GEMM_CASE(false, false);
  else DTF_GEMM_CASE(false, true);
#undef DTF_GEMM_CASE
  if (splits > 1) {
    const long work = ((long)M * N + 3) / 4;
    const int blocks = (int)((work + 255) / 256 > 1024 ? 1024 : (work + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, bias, C,
                       M, N, ldc, splits, act);
  }
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

}  // namespace

// Block tile for an M x N x K product, from measurements on MI355X: the 128x128 tile (2x2 MFMA
// tiles per wave, half the LDS reads per flop) wins from 4096^2 outputs on (4096^3: 136 vs 131
// TF, 8192^3: 140 vs 139 TF); below that the 64x64 tile keeps more workgroups per CU in flight
// and is faster (3072^3: 120 vs 96 TF, 2048^3: 128 vs 113 TF).
DTF_API int dtf_gemm_f32_tile(int M, int N, int K) {
  const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
  return big_tiles >= 1024 ? 128 : 64;
}

// Split-K plan: number of K splits (1 = none). A 64x64-tile grid of up to ~1000 workgroups that
// leaves CUs idle in its last round gains from up to 4 splits; tiny grids (the MLP's 100x100 outputs with K = 784) take as many splits as give
// every CU a workgroup, each split keeping at least four K-tiles.
DTF_API int dtf_gemm_f32_splits(int M, int N, int K, int* kchunk_out) {
  int splits = 1, kchunk = K;
  if (dtf_gemm_f32_tile(M, N, K) == 64 && K >= 256) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    int want = 1;
    if (tiles <= 64) {
      want = (int)(256 / tiles);
      if (want > 32) want = 32;
    } else if (tiles < 1024) {
      // a grid that fills whole rounds of the 256 CUs (1000^3: 256 tiles) runs unsplit at the
      // vendor library's speed (23.5 us); a ragged one (1536^3: 576 tiles = 2.25 rounds) gains
      // from finer work items
      const long rounds = (tiles + 255) / 256;
      if (tiles * 10 < rounds * 256 * 9) {
        want = (int)((1024 + tiles - 1) / tiles);
        if (want > 4) want = 4;
      }
    }
    if (want > 1) {
      int chunk = ((K + want - 1) / want + 15) / 16 * 16;
      if (chunk < 64) chunk = 64;
      splits = (K + chunk - 1) / chunk;
      kchunk = chunk;
      if (splits <= 1) { splits = 1; kchunk = K; }
    }
  }
  if (kchunk_out) *kchunk_out = kchunk;
  return splits;
}

// C[M,N] = act(op(A) * op(B) + bias[N]).  A is stored [M,K] (lda >= K), or [K,M] (lda >= M) when
// trans_a; B is stored [K,N] (ldb >= N), or [N,K] (ldb >= K) when trans_b. bias may be null.
// partial: split-K workspace of `splits` * M * N floats, `splits` being what
// dtf_gemm_f32_splits() returned (or the caller's own choice); splits <= 1 or a null workspace
// runs the product unsplit.
DTF_API int dtf_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N,
                         int K, long lda, long ldb, long ldc, int trans_a, int trans_b, int act,
                         float* partial, int splits, int tile, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return DTF_ERR_BAD_ARG;
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N) return DTF_ERR_BAD_ARG;
  if (tile != 64 && tile != 128) tile = dtf_gemm_f32_tile(M, N, K);
  int kchunk = ((K + BK - 1) / BK) * BK;
  if (!partial || splits <= 1) {
    splits = 1;
  } else {
    kchunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (K + kchunk - 1) / kchunk;          // never more than the caller allotted
  }
  if (tile == 128)
    return launch<128, 128>(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, splits,
                            kchunk, partial, stream);
  return launch<64, 64>(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, splits,
                        kchunk, partial, stream);
}
