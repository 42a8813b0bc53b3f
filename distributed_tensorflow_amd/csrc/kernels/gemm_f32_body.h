// Body of the fp32 MFMA GEMM (see gemm_f32.hip for the design): one workgroup's share of
// C = act(op(A) * op(B) + bias) as a device function, so that kernels which do more than one
// thing per launch -- the MLP step kernels in mlp_step.hip run a weight-gradient product next to
// the small-parameter update -- use the same code, instruction for instruction, as the plain
// product.
#pragma once

#include <type_traits>

#include "common.h"

namespace dtf_gemm {

constexpr int BK = 16;
constexpr int NTHREADS = 256;

// staging buffers of the 256x256 tile (just over the 64 KiB a kernel may declare statically)
extern __shared__ __attribute__((aligned(16))) float gemm_f32_dyn_lds[];

struct GemmF32Args {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  long lda, ldb, ldc;
  int a_vec, b_vec, act, tiles_n, ntiles, splits, kchunk;
  float* partial;
  float lr;            // EPI == 1 only
  // how far the branch-free loads may read along the contiguous dimension of an operand that is
  // NOT K-contiguous: M (N) rounded up to a multiple of 4 when the rows are that long in memory,
  // else M (N). Columns past M / N feed only outputs the epilogue drops.
  int m_load, n_load;
};


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

  // register i of a tile's set into its place in the LDS tile
  __device__ static void stash_one(const f32x4& reg, int i, float* __restrict__ Xs, int tid) {
    const int idx = tid + i * NTHREADS;
    if (KCONTIG) {
      const int t = idx >> 2, k4 = idx & 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) Xs[(4 * k4 + j) * LD + t] = reg[j];
    } else {
      constexpr int V = BT / 4;
      const int k = idx / V, t4 = idx % V;
      *reinterpret_cast<f32x4*>(&Xs[k * LD + 4 * t4]) = reg;
    }
  }

  __device__ static void stash(const f32x4 (&regs)[PER_THREAD], float* __restrict__ Xs, int tid) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) stash_one(regs[i], i, Xs, tid);
  }
};

// Epilogue through LDS for a 16-byte aligned C. An accumulator register holds one column of C per
// lane (C/D fragment of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) +
// 4 * (lane >> 5)); stored as it is, every instruction writes 64 single words. Here each wave
// turns its 32x32 tiles through a private 32 x XLD corner of the (idle) staging buffers and
// stores 16 bytes per lane, eight lanes to a 128-byte row segment: 8192 x 8192 outputs leave in
// 110 us instead of 175 (what a short-K product spends most of its time on). Values and the order
// of the additions are those of the one-word path. The caller has made sure no wave still reads
// the staging buffers.
template <int FM, int FN, int XLD>
__device__ __forceinline__ void gemm_f32_store_wide(float* xs, const f32x16 (&acc)[FM][FN],
                                                    float* __restrict__ C, long ldc, int row0,
                                                    int col0, int M, int N,
                                                    const float* __restrict__ bias, int act,
                                                    int lane) {
  static_assert(XLD % 4 == 0 && XLD >= 32, "rows of the turning buffer are read 16 bytes at a time");
  const int lrow = lane & 31, lk = lane >> 5;
  const int rrow = lane >> 3, c4 = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        xs[((r & 3) + 8 * (r >> 2) + 4 * lk) * XLD + lrow] = acc[i][j][r];
      const int col = col0 + 32 * j + c4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = col + e < N ? bias[col + e] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = row0 + 32 * i + 8 * q + rrow;
        f32x4 v = *reinterpret_cast<const f32x4*>(xs + (8 * q + rrow) * XLD + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dtf_apply_act(v[e] + bv[e], act);
        if (row >= M) continue;
        float* const dst = C + (long)row * ldc + col;
        if (col + 3 < N) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < N) dst[e] = v[e];
        }
      }
    }
}

// ABL (timing experiments, -DDTF_GEMM_EXPERIMENTS only; results are wrong when set): bit 0 = no
// global loads in the K loop, bit 1 = no LDS stash, bit 2 = no barrier, bit 3 = no fragment reads,
// bit 4 = every wave sums the shader-clock cycles each k-pair of its interior K-tiles took and
// leaves the means in p.partial[(workgroup * 4 + wave) * 16 + k-pair] (slot 15: tiles counted)
// EPI: 0 = C = act(acc + bias) (or the split's partial plane), 1 = C = fma(-lr, acc, C): the
// product is a gradient and C the variable it updates in place (plain SGD in the epilogue).
template <int BM, int BN, bool A_KC, bool B_KC, int ABL = 0, int EPI = 0>
__device__ __forceinline__ void gemm_f32_block(const GemmF32Args& p, const int bid) {
  const float* __restrict__ A = p.A;
  const float* __restrict__ B = p.B;
  float* __restrict__ C = p.C;
  const float* __restrict__ bias = p.bias;
  const int M = p.M, N = p.N, K = p.K;
  const long lda = p.lda, ldb = p.ldb, ldc = p.ldc;
  const int a_vec = p.a_vec, b_vec = p.b_vec, act = p.act, tiles_n = p.tiles_n;
  const int ntiles = p.ntiles, splits = p.splits, kchunk = p.kchunk;
  float* __restrict__ partial = p.partial;
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
  const int split = bid / ntiles;
  // XCD-aware remap (each XCD's L2 sees a contiguous chunk of the walk) + grouped walk: the
  // ~32-64 tiles resident on one XCD form a block GROUP tile-rows deep, so they share A and B
  // panels instead of streaming a whole tile-row of B each
  const int tile = dtf_xcd_remap(bid % ntiles, ntiles);
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
  const int Ml = A_KC ? M : p.m_load, Nl = B_KC ? N : p.n_load;    // extents the loads clamp to
  const bool fast = a_vec != 0 && b_vec != 0 && k_end - k_begin >= 4 &&
                    lda * 4 * (A_KC ? BM : BK) < (1L << 32) &&
                    ldb * 4 * (B_KC ? BN : BK) < (1L << 32) &&
                    (A_KC ? (K & 3) == 0 : ((Ml & 3) == 0 && Ml >= 4)) &&
                    (B_KC ? (K & 3) == 0 : ((Nl & 3) == 0 && Nl >= 4));

  // The K loop, instantiated twice (FAST: branch-free loads; otherwise fully predicated ones).
  auto k_loop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    constexpr int DEPTH = (BM == 128) ? 3 : 4;        // register sets = K-tiles in flight
    f32x4 ra[DEPTH][LA::PER_THREAD], rb[DEPTH][LB::PER_THREAD];
    unsigned voff_a[LA::PER_THREAD], voff_b[LB::PER_THREAD];
    if (FAST) {
      LA::lane_offsets(voff_a, lda, m0, Ml, tid);
      LB::lane_offsets(voff_b, ldb, n0, Nl, tid);
    }
    // any tile: clamped addresses, zeros beyond k_end (K tail, prefetches past the end)
    auto fetch_tile = [&](f32x4 (&qa)[LA::PER_THREAD], f32x4 (&qb)[LB::PER_THREAD], int kt) {
      const int k0 = k_begin + kt * BK;
      if (FAST) {
        LA::fetch_fast(qa, A, lda, m0, k0, Ml, k_end, tid);
        LB::fetch_fast(qb, B, ldb, n0, k0, Nl, k_end, tid);
      } else {
        LA::fetch(qa, A, lda, m0, k0, M, k_end, a_vec != 0, tid);
        LB::fetch(qb, B, ldb, n0, k0, N, k_end, b_vec != 0, tid);
      }
    };
    // a tile that lies wholly inside [k_begin, k_end) of aligned operands
    auto fetch_interior_tile = [&](f32x4 (&qa)[LA::PER_THREAD], f32x4 (&qb)[LB::PER_THREAD],
                                   int kt) {
      const int k0 = k_begin + kt * BK;
      LA::fetch_interior(qa, LA::tile_base(A, lda, m0, Ml, k0), voff_a);
      LB::fetch_interior(qb, LB::tile_base(B, ldb, n0, Nl, k0), voff_b);
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
    unsigned long long t_prev = 0, t_sum[KP] = {};
    int t_count = 0;
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
          if ((ABL & 16) && INTERIOR) {
            // slot kk: from the start of k-pair kk-1 to the start of k-pair kk
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (t_prev != 0) t_sum[(kk + KP - 1) % KP] += now - t_prev;
            t_prev = now;
            if (kk == 0) ++t_count;
          }
          if (!(ABL & 2)) {
            // tile kt+1 (loaded DEPTH-1 tiles ago; zeros past the end) goes into the other LDS
            // buffer, whose last readers finished before the previous tile's barrier: one
            // register of each operand's set per k-pair, in the k-pairs before the barrier. (All
            // of a 256x256 tile's 20 LDS writes in one place, from four waves at once, held the
            // waves -- and the MFMAs queued behind the writes -- for 500 cycles per K-tile.)
            // Neither this nor the ring's reads below are skipped on the last tile: a branch
            // around LDS operations would force the compiler to drain lgkmcnt to zero at the
            // join instead of counting, and every MFMA would then wait for the newest fragment
            // read.
            constexpr int SLICES = LA::PER_THREAD > LB::PER_THREAD ? LA::PER_THREAD : LB::PER_THREAD;
            constexpr int S0 = (KP - PD - SLICES) / 2 > 0 ? (KP - PD - SLICES) / 2 : 0;
            static_assert(S0 + SLICES <= KP - PD, "stash must end before the barrier");
            const int si = kk - S0;
            if (si >= 0 && si < LA::PER_THREAD)
              LA::stash_one(ra[(u + 1) % DEPTH][si], si, As + (cur ^ 1) * A_BUF, tid);
            if (si >= 0 && si < LB::PER_THREAD)
              LB::stash_one(rb[(u + 1) % DEPTH][si], si, Bs + (cur ^ 1) * B_BUF, tid);
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
    if ((ABL & 16) && lane == 0 && partial) {
#pragma unroll
      for (int kk = 0; kk < KP; ++kk)
        partial[(bid * 4 + wave) * 16 + kk] = (float)t_sum[kk] / (float)(t_count > 0 ? t_count : 1);
      partial[(bid * 4 + wave) * 16 + 15] = (float)t_count;
    }
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
  if (EPI == 0 && ((uintptr_t)C % 16 == 0) && (ldc % 4 == 0)) {
    static_assert(4 * 32 * 32 <= 2 * BK * (LA::LD + LB::LD), "turning buffers fit the staging");
    __syncthreads();                                     // staging buffers are idle from here on
    gemm_f32_store_wide<TM, TN, 32>(smem + wave * (32 * 32), acc, C, ldc, m0 + wr * WM,
                                    n0 + wc * WN, M, N, bias, act, lane);
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
        if (row >= M) continue;
        float* const c = C + (long)row * ldc + col;
        if (EPI == 1) *c = __builtin_fmaf(-p.lr, acc[i][j][r], *c);
        else *c = dtf_apply_act(acc[i][j][r] + bv, act);
      }
    }
}


// ---- large products: the same tiling with every instruction dealt into an MFMA slot -----------
// The exact-fp32 MFMA needs so little operand traffic (a 32x32x2 step is 64 cycles on two
// registers) that a large product is limited by nothing but how often the matrix pipe is left
// without its next instruction. The general kernel above leaves that to the compiler, which packs
// each group of LDS writes, fragment reads and address arithmetic between two MFMAs; the matrix
// pipe is one MFMA deep and ran dry behind every such group (11 % of the time at 8192^3). Here
// whatever is not an MFMA is dealt out by hand, one instruction behind each MFMA, so that it
// issues in the shadow of the 64 cycles that MFMA runs. For the 256x256 tile (four waves of 4x4
// MFMA tiles, 256 accumulators each, a K-tile of 16 = 128 MFMAs = 8192 cycles per wave):
//
//   k-pair kk of a K-tile (16 MFMAs):  slots 0..7   the 8 fragment reads of k-pair kk+1
//                                      slots 8..15  kk == 0: the tile's 8 global loads (tile t+2)
//                                                   kk == 1..4: LDS writes of tile t+1 (register
//                                                   kk-1 of both operands' sets)
//   before k-pair 7: wait for the LDS writes, barrier -- the fragments k-pair 7 reads are those of
//   tile t+1, out of the other LDS buffer.
//
// A fence after every slot (sched_barrier) keeps the compiler from regrouping what was dealt out.
// Two register sets and two LDS buffers alternate, so the K loop is unrolled by two tiles and
// every buffer and register index in it is a constant: the loop has no vector ALU instruction at
// all. Measured per k-pair: 1030 cycles (ideal 1024), 1160 for the one with the barrier. The
// 128x128 tile (T = 128 below) is dealt the same way and, with two workgroups per CU covering
// each other's barriers, is the faster of the two (8192^3: 151 against 150 TFLOP/s) and the one
// dtf_gemm_f32_tile chooses; 256x256 stays selectable. Operands must qualify for the branch-free
// loads (16-byte aligned rows, K -- and M or N where they are the contiguous dimension --
// multiples of 4); edges in M and N are clamped on load and predicated on store, a K tail reads
// zeros. Sums run over k in the same order as in the general kernel: the results are bitwise the
// same.
// ABL (timing experiments, -DDTF_GEMM_EXPERIMENTS only): bit 4 = per-k-pair cycle sums, as in
// gemm_f32_block.
constexpr int SLOT_MAX_T = 256;

// LDS bytes of the staging buffers of a T x T tile
__host__ __device__ constexpr size_t gemm_f32_slot_lds_bytes(int T, bool a_kc, bool b_kc) {
  return sizeof(float) * 2 * BK * ((a_kc ? T + 2 : T) + (b_kc ? T + 2 : T));
}

// T: block tile edge, 256 (above) or 128. With 128 a wave owns 2x2 MFMA tiles, a k-pair is four
// MFMAs, and the slot behind each of them takes a fragment read AND a load or an LDS write (two
// workgroups share a CU, so the other wave's MFMAs cover what one slot cannot).
template <int T, bool A_KC, bool B_KC, int ABL = 0>
__device__ __forceinline__ void gemm_f32_slot_block(const GemmF32Args& p, const int bid) {
  static_assert(T == 128 || T == 256, "tile edge");
  const float* __restrict__ A = p.A;
  const float* __restrict__ B = p.B;
  float* __restrict__ C = p.C;
  const int M = p.M, N = p.N, K = p.K;
  const long lda = p.lda, ldb = p.ldb, ldc = p.ldc;
  using LA = Loader<T, A_KC>;
  using LB = Loader<T, B_KC>;
  constexpr int PT = LA::PER_THREAD;            // registers (of 16 bytes) per thread and K-tile
  static_assert(PT == T / 64 && LB::PER_THREAD == PT, "registers per operand");
  constexpr int WT = T / 2;                     // wave tile edge
  constexpr int F = WT / 32;                    // MFMA tiles per wave and dimension
  constexpr int MF = F * F;                     // MFMAs per k-pair
  constexpr int J = 2 * PT;                     // loads / LDS-write jobs dealt out per k-pair
  constexpr int J0 = MF - J;                    // ... behind MFMAs J0 .. MF-1
  static_assert(J0 >= 0 && 2 * F <= MF, "slot plan");
  constexpr int A_BUF = BK * LA::LD, B_BUF = BK * LB::LD;    // floats per buffer
  float* const As = gemm_f32_dyn_lds;                        // As[buf][k][m]
  float* const Bs = gemm_f32_dyn_lds + 2 * A_BUF;            // Bs[buf][k][n]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int ntiles = p.ntiles, tiles_n = p.tiles_n;
  const int tile = dtf_xcd_remap(bid, ntiles);
  constexpr int GROUP = 8;
  const int tiles_m = ntiles / tiles_n;
  const int per_group = GROUP * tiles_n;
  const int first_m = (tile / per_group) * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int m0 = (first_m + (tile % per_group) % gsize) * T;
  const int n0 = ((tile % per_group) / gsize) * T;
  const int nkt = (K + BK - 1) / BK;
  const int lrow = lane & 31, lk = lane >> 5;

  f32x16 acc[F][F];
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[2][PT], rb[2][PT];
  unsigned voff_a[PT], voff_b[PT];
  const int Ml = A_KC ? M : p.m_load, Nl = B_KC ? N : p.n_load;    // extents the loads clamp to
  LA::lane_offsets(voff_a, lda, m0, Ml, tid);
  LB::lane_offsets(voff_b, ldb, n0, Nl, tid);

  // prologue: tiles 0 and 1 requested, tile 0 in LDS buffer 0, fragments of its k-pair 0 read
  LA::fetch_fast(ra[0], A, lda, m0, 0, Ml, K, tid);
  LB::fetch_fast(rb[0], B, ldb, n0, 0, Nl, K, tid);
  LA::fetch_fast(ra[1], A, lda, m0, BK, Ml, K, tid);
  LB::fetch_fast(rb[1], B, ldb, n0, BK, Nl, K, tid);
  LA::stash(ra[0], As, tid);
  LB::stash(rb[0], Bs, tid);
  __syncthreads();

  // fragment of k-pair kp in buffer `buf`: A[m = 32 i + lrow][k = 2 kp + lk] at a_rd[...]
  const float* const a_rd = As + lk * LA::LD + wr * WT + lrow;
  const float* const b_rd = Bs + lk * LB::LD + wc * WT + lrow;
  float a[2][F], b[2][F];
#pragma unroll
  for (int i = 0; i < F; ++i) {
    a[0][i] = a_rd[32 * i];
    b[0][i] = b_rd[32 * i];
  }

  unsigned long long t_prev = 0, t_sum[8] = {};
  int t_count = 0;

  // K-tiles kt0 (LDS buffer / register set 0) and kt0 + 1 (buffer / set 1)
  auto pair_of_tiles = [&](auto interior_tag, int kt0) {
    constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kt = kt0 + s;
      if (!INTERIOR && kt >= nkt) return;
      const int k_next = (kt + 2) * BK;                 // the tile this one requests
      const float* const base_a = LA::tile_base(A, lda, m0, Ml, k_next);
      const float* const base_b = LB::tile_base(B, ldb, n0, Nl, k_next);
      const __amdgpu_buffer_rsrc_t rsrc_a =
          __builtin_amdgcn_make_buffer_rsrc((void*)base_a, 0, 0xffffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsrc_b =
          __builtin_amdgcn_make_buffer_rsrc((void*)base_b, 0, 0xffffffff, 0x00020000);
      if (!INTERIOR) {
        // clamped addresses and zeros beyond K, all at once (two tiles per product)
        LA::fetch_fast(ra[s], A, lda, m0, k_next, Ml, K, tid);
        LB::fetch_fast(rb[s], B, ldb, n0, k_next, Nl, K, tid);
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        constexpr int KP = 8;
        if ((ABL & 16) && INTERIOR) {
          const unsigned long long now = __builtin_amdgcn_s_memtime();
          if (t_prev != 0) t_sum[(kk + KP - 1) % KP] += now - t_prev;
          t_prev = now;
          if (kk == 0) ++t_count;
        }
        if (kk == KP - 1) {
          // every wave's writes of tile kt+1 are in LDS and every wave's reads of this tile's
          // buffer are done: the other buffer may be read, this one refilled
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
        }
        const int c = kk & 1;                            // fragment ring slot of this k-pair
        // where k-pair kk + 1 lives: this tile's buffer, or k-pair 0 of the other one
        const int nb = kk + 1 < KP ? s : s ^ 1, nkp = (kk + 1) % KP;
        const float* const an = a_rd + nb * A_BUF + 2 * nkp * LA::LD;
        const float* const bn = b_rd + nb * B_BUF + 2 * nkp * LB::LD;
#pragma unroll
        for (int n = 0; n < MF; ++n) {
          acc[n / F][n % F] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][n / F], b[c][n % F],
                                                                   acc[n / F][n % F], 0, 0, 0);
          if (n < F) a[c ^ 1][n] = an[32 * n];
          else if (n < 2 * F) b[c ^ 1][n - F] = bn[32 * (n - F)];
          if (n >= J0 && kk == 0) {
            if (INTERIOR) {
              const int q = (n - J0) >> 1;
              if ((n - J0) & 1)
                rb[s][q] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, voff_b[q], 0, 0));
              else
                ra[s][q] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, voff_a[q], 0, 0));
            }
          } else if (n >= J0 && kk <= 4) {
            // tile kt+1's register sets into the other buffer, 8 PT jobs over k-pairs 1..4:
            // job = (register q, operand, word w). A K-contiguous operand is transposed on the
            // way (four one-word writes per register), the other goes as it is (one write)
            const int job = (kk - 1) * J + (n - J0);
            const int q = job >> 3, w = job & 7;
            float* const ad = As + (s ^ 1) * A_BUF;
            float* const bd = Bs + (s ^ 1) * B_BUF;
            const int idx = tid + q * NTHREADS;
            if (w < 4) {
              if (A_KC) ad[(4 * (idx & 3) + w) * LA::LD + (idx >> 2)] = ra[s ^ 1][q][w];
              else if (w == 0) LA::stash_one(ra[s ^ 1][q], q, ad, tid);
            } else {
              if (B_KC) bd[(4 * (idx & 3) + (w - 4)) * LB::LD + (idx >> 2)] = rb[s ^ 1][q][w - 4];
              else if (w == 4) LB::stash_one(rb[s ^ 1][q], q, bd, tid);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  int kt0 = 0;
  for (; (kt0 + 4) * BK <= K; kt0 += 2) pair_of_tiles(std::true_type{}, kt0);
  for (; kt0 < nkt; kt0 += 2) pair_of_tiles(std::false_type{}, kt0);

  if ((ABL & 16) && lane == 0 && p.partial) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      p.partial[(bid * 4 + wave) * 16 + kk] = (float)t_sum[kk] / (float)(t_count > 0 ? t_count : 1);
    p.partial[(bid * 4 + wave) * 16 + 15] = (float)t_count;
  }

  // (not for the 256x256 tile: with 256 accumulators live the turn spills registers)
  const bool wide = T == 128 && ((uintptr_t)C % 16 == 0) && (ldc % 4 == 0);
  if (wide) {
    __syncthreads();                                     // staging buffers are idle from here on
    gemm_f32_store_wide<F, F, 36>(gemm_f32_dyn_lds + wave * (32 * 36), acc, C, ldc,
                                  m0 + wr * WT, n0 + wc * WT, M, N, p.bias, p.act, lane);
    return;
  }
#pragma unroll
  for (int i = 0; i < F; ++i)
#pragma unroll
    for (int j = 0; j < F; ++j) {
      const int col = n0 + wc * WT + 32 * j + lrow;
      if (col >= N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * WT + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row >= M) continue;
        C[(long)row * ldc + col] = dtf_apply_act(acc[i][j][r] + bv, p.act);
      }
    }
}

// whether the operands of C[M,N] = op(A) op(B) qualify for gemm_f32_slot_block
__host__ inline bool gemm_f32_slot_ok(const void* A, const void* B, int m_load, int n_load, int K,
                                     long lda, long ldb, bool a_kc, bool b_kc) {
  if (((uintptr_t)A | (uintptr_t)B) % 16 != 0 || lda % 4 != 0 || ldb % 4 != 0) return false;
  if (K < 4 * BK || m_load < 4 || n_load < 4) return false;
  if (a_kc ? K % 4 != 0 : m_load % 4 != 0) return false;
  if (b_kc ? K % 4 != 0 : n_load % 4 != 0) return false;
  // lane offsets within a tile are 32-bit byte offsets
  if (lda * 4 * (a_kc ? SLOT_MAX_T : BK) >= (1L << 32) ||
      ldb * 4 * (b_kc ? SLOT_MAX_T : BK) >= (1L << 32))
    return false;
  return true;
}


// ---- small products: one 16x16 output tile per workgroup, K split over its four waves ----------
// A product with few outputs and a long or short K -- a dense layer on a batch of 100, the weight
// gradient of such a layer -- does not fill the chip with 64x64 tiles, and cutting K into
// workgroups costs a second launch and a round trip of partial planes through memory. Here a
// workgroup owns ONE 16x16 tile of C (100x100 outputs: 49 workgroups; 784x100: 343) and its four
// waves each take a contiguous quarter of K, in units of 16, on the exact-fp32
// v_mfma_f32_16x16x4_f32 (A[m = l&15][k = l>>4], B[k = l>>4][n = l&15], one VGPR each). Operands
// go from memory straight into MFMA registers, four 16-wide K-blocks in flight ahead of the
// MFMAs; there is no LDS staging to wait for and no barrier in the loop. A K-contiguous A is read
// 16 bytes per lane: lane (m, q) takes A[m][k0 + 4q .. 4q+3] and MFMA j of the block multiplies
// element j of every lane, i.e. the k's {4q + j}; B is read to match, B[k0 + 4q + j][n]. An
// M-contiguous A (the transposed operand of a weight gradient) is read one word per lane,
// A[k0 + 4j + q][m]. The four partial tiles meet in LDS and are summed in wave order, then bias
// and activation (EPI 0) or the in-place SGD update C = fma(-lr, acc, C) (EPI 1) follow, C's old
// value having been requested before the K loop. Every sum has a fixed order: results are
// bit-identical from launch to launch.
constexpr int SK_T = 16;            // tile edge
constexpr int SK_U = 4;             // K-blocks of 16 in flight per wave

struct SkinnyFrag {
  float a[4], b[4];
};

// B_KC: B is stored [N, K] (the caller's transposed operand)
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void skinny_load(SkinnyFrag& f, const float* __restrict__ A,
                                            const float* __restrict__ B, long lda, long ldb,
                                            int am, int bn, int kq, int k0, int k_end, bool a_vec,
                                            bool b_vec) {
  if (A_KC) {
    // element j of the lane is k = k0 + 4 kq + j
    const int k = k0 + 4 * kq;
    const float* pa = A + (long)am * lda + k;
    if (a_vec && k + 3 < k_end) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pa);
      f.a[0] = v[0]; f.a[1] = v[1]; f.a[2] = v[2]; f.a[3] = v[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.a[j] = k + j < k_end ? pa[j] : 0.f;
    }
    if (B_KC) {
      const float* pb = B + (long)bn * ldb + k;
      if (b_vec && k + 3 < k_end) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pb);
        f.b[0] = v[0]; f.b[1] = v[1]; f.b[2] = v[2]; f.b[3] = v[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = k + j < k_end ? pb[j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.b[j] = k + j < k_end ? B[(long)(k + j) * ldb + bn] : 0.f;
    }
  } else {
    // element j of the lane is k = k0 + 4 j + kq
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 4 * j + kq;
      f.a[j] = k < k_end ? A[(long)k * lda + am] : 0.f;
      f.b[j] = k < k_end ? (B_KC ? B[(long)bn * ldb + k] : B[(long)k * ldb + bn]) : 0.f;
    }
  }
}

struct SkinnyArgs {
  const float* A;      // [M, K] (A_KC) or [K, M]
  const float* B;      // [K, N], or [N, K] (B_KC)
  float* C;            // [M, N]
  const float* bias;   // [N] or null (EPI 0)
  int M, N, K;
  long lda, ldb, ldc;
  int act;             // EPI 0
  float lr;            // EPI 1
  int tiles_n;
};

// NW: waves per workgroup (4 or 16). With a long K sixteen waves each take a sixteenth: every wave
// then has all of its operands in flight at once, one memory round trip instead of several.
template <bool A_KC, bool B_KC, int EPI, int NW = 4>
__device__ __forceinline__ void gemm_f32_skinny_block(const SkinnyArgs& p, const int bid) {
  __shared__ float red[NW][SK_T][SK_T + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = (bid / p.tiles_n) * SK_T, n0 = (bid % p.tiles_n) * SK_T;
  const int lm = lane & 15, kq = lane >> 4;
  // rows / columns past the edge re-read the last valid one; their results are dropped
  const int am = m0 + lm < p.M ? m0 + lm : p.M - 1;
  const int bn = n0 + lm < p.N ? n0 + lm : p.N - 1;
  const bool a_vec = A_KC && (p.lda % 4 == 0) && ((uintptr_t)p.A % 16 == 0);
  const bool b_vec = B_KC && (p.ldb % 4 == 0) && ((uintptr_t)p.B % 16 == 0);

  // the output element this thread finishes: (row, col) = (tid / 16, tid % 16) of the tile
  const int orow = m0 + ((tid >> 4) & 15), ocol = n0 + (tid & 15);
  const bool oin = tid < SK_T * SK_T && orow < p.M && ocol < p.N;
  float c_old = 0.f, bv = 0.f;
  if (EPI == 1) c_old = oin ? p.C[(long)orow * p.ldc + ocol] : 0.f;
  else bv = (p.bias && oin) ? p.bias[ocol] : 0.f;

  const int nkb = (p.K + 15) / 16;
  const int per = (nkb + NW - 1) / NW;
  const int kb0 = wave * per;
  const int kb1 = kb0 + per < nkb ? kb0 + per : nkb;

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  SkinnyFrag cur[SK_U], nxt[SK_U];
#pragma unroll
  for (int u = 0; u < SK_U; ++u)
    skinny_load<A_KC, B_KC>(cur[u], p.A, p.B, p.lda, p.ldb, am, bn, kq, (kb0 + u) * 16,
                            kb0 + u < kb1 ? p.K : 0, a_vec, b_vec);
  for (int kb = kb0; kb < kb1; kb += SK_U) {
#pragma unroll
    for (int u = 0; u < SK_U; ++u)
      skinny_load<A_KC, B_KC>(nxt[u], p.A, p.B, p.lda, p.ldb, am, bn, kq, (kb + SK_U + u) * 16,
                              kb + SK_U + u < kb1 ? p.K : 0, a_vec, b_vec);
#pragma unroll
    for (int u = 0; u < SK_U; ++u) {
      // (blocks past the wave's range hold zeros: fma(0, 0, acc) leaves acc as it is)
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].a[0], cur[u].b[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].a[1], cur[u].b[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].a[2], cur[u].b[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[u].a[3], cur[u].b[3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < SK_U; ++u) cur[u] = nxt[u];
  }
  // C/D fragment: col = lane & 15, row = 4 * (lane >> 4) + i
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][4 * kq + i][lm] = acc0[i] + acc1[i];
  __syncthreads();
  if (tid >= SK_T * SK_T || !oin) return;
  const int r = tid >> 4, c = tid & 15;
  float v = red[0][r][c];                          // wave order
#pragma unroll
  for (int w = 1; w < NW; ++w) v += red[w][r][c];
  float* const out = p.C + (long)orow * p.ldc + ocol;
  if (EPI == 1) *out = __builtin_fmaf(-p.lr, v, c_old);
  else *out = dtf_apply_act(v + bv, p.act);
}

// ---- small-product kernel on 32x32 tiles -------------------------------------------------------
// The same idea one size up, for products whose 64x64 tiling leaves most CUs idle but whose
// 16x16 tiling is thousands of workgroups (384^3 ... 768^3): one 32x32 output tile per workgroup,
// K dealt to NW waves, operands straight from memory into v_mfma_f32_32x32x2_f32 (lane l holds
// row / column l & 31 and the k of half l >> 5), partial tiles summed through LDS in wave order.
// A K-block is 8 k = four MFMAs: with a K-contiguous operand a lane loads 16 bytes (k0 + 4 half
// + j, j = 0..3) and MFMA j takes element j, so it multiplies k0 + j and k0 + 4 + j; otherwise
// element j is k0 + 2 j + half. Both operands always use the same assignment.
constexpr int SK32_T = 32;
constexpr int SK32_U = 4;           // K-blocks of 8 in flight per wave

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void skinny32_load(SkinnyFrag& f, const float* __restrict__ A,
                                              const float* __restrict__ B, long lda, long ldb,
                                              int am, int bn, int half, int k0, int k_end,
                                              bool a_vec, bool b_vec) {
  if (A_KC) {
    const int k = k0 + 4 * half;
    const float* pa = A + (long)am * lda + k;
    if (a_vec && k + 3 < k_end) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pa);
      f.a[0] = v[0]; f.a[1] = v[1]; f.a[2] = v[2]; f.a[3] = v[3];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.a[j] = k + j < k_end ? pa[j] : 0.f;
    }
    if (B_KC) {
      const float* pb = B + (long)bn * ldb + k;
      if (b_vec && k + 3 < k_end) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pb);
        f.b[0] = v[0]; f.b[1] = v[1]; f.b[2] = v[2]; f.b[3] = v[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = k + j < k_end ? pb[j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.b[j] = k + j < k_end ? B[(long)(k + j) * ldb + bn] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 2 * j + half;
      f.a[j] = k < k_end ? A[(long)k * lda + am] : 0.f;
      f.b[j] = k < k_end ? (B_KC ? B[(long)bn * ldb + k] : B[(long)k * ldb + bn]) : 0.f;
    }
  }
}

// NW: waves per workgroup (4 or 8)
template <bool A_KC, bool B_KC, int NW>
__device__ __forceinline__ void gemm_f32_skinny32_block(const SkinnyArgs& p, const int bid) {
  __shared__ float red[NW][SK32_T][SK32_T + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = (bid / p.tiles_n) * SK32_T, n0 = (bid % p.tiles_n) * SK32_T;
  const int lm = lane & 31, half = lane >> 5;
  // rows / columns past the edge re-read the last valid one; their results are dropped
  const int am = m0 + lm < p.M ? m0 + lm : p.M - 1;
  const int bn = n0 + lm < p.N ? n0 + lm : p.N - 1;
  const bool a_vec = A_KC && (p.lda % 4 == 0) && ((uintptr_t)p.A % 16 == 0);
  const bool b_vec = B_KC && (p.ldb % 4 == 0) && ((uintptr_t)p.B % 16 == 0);

  const int nkb = (p.K + 7) / 8;
  const int per = (nkb + NW - 1) / NW;
  const int kb0 = wave * per;
  const int kb1 = kb0 + per < nkb ? kb0 + per : nkb;

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  SkinnyFrag cur[SK32_U], nxt[SK32_U];
#pragma unroll
  for (int u = 0; u < SK32_U; ++u)
    skinny32_load<A_KC, B_KC>(cur[u], p.A, p.B, p.lda, p.ldb, am, bn, half, (kb0 + u) * 8,
                              kb0 + u < kb1 ? p.K : 0, a_vec, b_vec);
  for (int kb = kb0; kb < kb1; kb += SK32_U) {
#pragma unroll
    for (int u = 0; u < SK32_U; ++u)
      skinny32_load<A_KC, B_KC>(nxt[u], p.A, p.B, p.lda, p.ldb, am, bn, half,
                                (kb + SK32_U + u) * 8, kb + SK32_U + u < kb1 ? p.K : 0, a_vec,
                                b_vec);
#pragma unroll
    for (int u = 0; u < SK32_U; ++u) {
      // (blocks past the wave's range hold zeros: fma(0, 0, acc) leaves acc as it is)
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u].a[0], cur[u].b[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u].a[1], cur[u].b[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u].a[2], cur[u].b[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[u].a[3], cur[u].b[3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < SK32_U; ++u) cur[u] = nxt[u];
  }
  // C/D fragment of the 32x32 MFMA: col = lane & 31, row = 8 (i / 4) + 4 (lane >> 5) + i % 4
#pragma unroll
  for (int i = 0; i < 16; ++i)
    red[wave][8 * (i >> 2) + 4 * half + (i & 3)][lm] = acc0[i] + acc1[i];
  __syncthreads();
  for (int e = tid; e < SK32_T * SK32_T; e += 64 * NW) {
    const int r = e >> 5, c = e & 31;
    const int orow = m0 + r, ocol = n0 + c;
    if (orow >= p.M || ocol >= p.N) continue;
    float v = red[0][r][c];                          // wave order
#pragma unroll
    for (int w = 1; w < NW; ++w) v += red[w][r][c];
    p.C[(long)orow * p.ldc + ocol] = dtf_apply_act(v + (p.bias ? p.bias[ocol] : 0.f), p.act);
  }
}

__host__ __device__ inline int gemm_f32_skinny32_waves(long K) { return K > 256 ? 8 : 4; }

// waves per workgroup the small-product kernel runs with for a given K
__host__ __device__ inline int gemm_f32_skinny_waves(long K) { return K > 256 ? 16 : 4; }

// 1 when the small-product kernel is the better plan for C[M,N] = op(A)[M,K] * B[K,N]: the
// 64x64 tiling would leave most of the chip idle and K is short enough for four waves
__host__ __device__ inline bool gemm_f32_skinny_fits(long M, long N, long K) {
  const long tiles64 = ((M + 63) / 64) * ((N + 63) / 64);
  return tiles64 <= 32 && K <= 4096 && M >= 1 && N >= 1 && K >= 1;
}

}  // namespace dtf_gemm
