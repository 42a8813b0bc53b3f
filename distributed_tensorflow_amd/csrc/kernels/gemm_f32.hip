// fp32 GEMM on the gfx950 matrix cores: C = act(op(A) * op(B) + bias).
//
// gfx950 has no TF32/xf32 path, but it has an exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) that
// runs at the vector-FMA rate while leaving the VALU free, so the kernel is built on it.
// This is the kernel behind tf.matmul in the reference graphs: the 1000x1000x1000 product
// (tf_distributed_1000Matrix.py:48), the MLP layers x*W1 [B,784]x[784,100] and a2*W2
// [B,100]x[100,10] (tf_distributed.py:62,64) and their NT / TN gradient products.
//
// Structure: 256 threads = 4 waves in a 2x2 grid over a BMxBN block tile (128x128 for large
// problems, 64x64 when that is what it takes to put a workgroup on every CU), BK = 16. Large
// products with aligned operands run the same tiling with a hand-dealt instruction order
// (gemm_f32_slot_block in gemm_f32_body.h); what follows describes the general kernel.
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
#include "gemm_f32_body.h"
#include "step_ring.h"

#include <cstdlib>

namespace {

using namespace dtf_gemm;

// ABL (timing experiments, -DDTF_GEMM_EXPERIMENTS only; results are wrong when set): bit 0 = no
// global loads in the K loop, bit 1 = no LDS stash, bit 2 = no barrier, bit 3 = no fragment reads
template <int BM, int BN, bool A_KC, bool B_KC, int ABL = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GemmF32Args p) {
  gemm_f32_block<BM, BN, A_KC, B_KC, ABL, 0>(p, blockIdx.x);
}

// large products: 128x128 or 256x256 tiles, every instruction dealt into an MFMA slot
// (gemm_f32_body.h)
template <int T, bool A_KC, bool B_KC, int ABL = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_slot_kernel(GemmF32Args p) {
  gemm_f32_slot_block<T, A_KC, B_KC, ABL>(p, blockIdx.x);
}

// C = act(op(A) op(B) + bias) for small outputs: one 16x16 tile per workgroup (gemm_f32_body.h)
template <bool A_KC, bool B_KC, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_f32_skinny32_kernel(SkinnyArgs p) {
  gemm_f32_skinny32_block<A_KC, B_KC, NW>(p, blockIdx.x);
}

template <bool A_KC, bool B_KC, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_f32_skinny_kernel(SkinnyArgs p) {
  gemm_f32_skinny_block<A_KC, B_KC, 0, NW>(p, blockIdx.x);
}

// The same product inside a CAPTURED step whose A operand is a fed tensor: A is entry `a_index` of
// the replay's slot of the step ring's address table (step_ring.h), so the batch is read where it
// lies in the resident data set and no gather launch copies it first. `first`: this is the step's
// first ring kernel and publishes the replay number.
template <bool A_KC, bool B_KC, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_f32_skinny_ring_kernel(SkinnyArgs p,
                                                                       DtfStepRing ring,
                                                                       int a_index, int first) {
  p.A = first ? (const float*)dtf_ring_feed(ring, dtf_ring_begin(ring), a_index)
              : (const float*)dtf_ring_feed_later(ring, a_index);
  gemm_f32_skinny_block<A_KC, B_KC, 0, NW>(p, blockIdx.x);
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

template <int BM, int BN, bool AKC, bool BKC, int ABL>
void launch_kernel(dim3 grid, dim3 block, hipStream_t stream, const GemmF32Args& args) {
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AKC, BKC, ABL>), grid, block, 0, stream, args);
}

// T x T tiles dealt out in MFMA slots (gemm_f32_slot_block); the staging buffers are dynamic LDS
// (those of the 256x256 tile are just over the 64 KiB a kernel may use without asking)
template <int T, bool AKC, bool BKC, int ABL>
void launch_slot_kernel(dim3 grid, hipStream_t stream, const GemmF32Args& args, int occ = 0) {
  // occ: workgroups per CU wanted (0 = as many as fit); asking for a larger LDS allocation than
  // the buffers need is what caps it
  size_t lds = gemm_f32_slot_lds_bytes(T, AKC, BKC);
  if (occ > 0 && (size_t)(160 * 1024 / (occ + 1) + 1024) > lds) lds = 160 * 1024 / (occ + 1) + 1024;
  static DtfPerDeviceOnce once;
  dtf_once_per_device(once, [&] {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gemm_f32_slot_kernel<T, AKC, BKC, ABL>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((gemm_f32_slot_kernel<T, AKC, BKC, ABL>), grid, dim3(NTHREADS), lds, stream,
                     args);
}

template <int T>
int launch_slot(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
               long lda, long ldb, long ldc, int trans_a, int trans_b, int act,
               float* partial, int m_load, int n_load, hipStream_t stream, int occ = 0) {
  const int tiles_m = (M + T - 1) / T, tiles_n = (N + T - 1) / T;
  const int ntiles = tiles_m * tiles_n;
  const GemmF32Args args{A, B, C, bias, M, N, K, lda, ldb, ldc, 1, 1, act, tiles_n, ntiles,
                         1, K, partial, 0.f, m_load, n_load};
  dim3 grid(ntiles);
#ifdef DTF_GEMM_EXPERIMENTS
  if (const char* e = getenv("DTF_F32_ABL")) {
    if (atoi(e) == 16 && !trans_a && !trans_b) {
      launch_slot_kernel<T, true, false, 16>(grid, stream, args);
      return DTF_OK;
    }
  }
#endif
  if (!trans_a && !trans_b) launch_slot_kernel<T, true, false, 0>(grid, stream, args, occ);
  else if (!trans_a && trans_b) launch_slot_kernel<T, true, true, 0>(grid, stream, args, occ);
  else if (trans_a && !trans_b) launch_slot_kernel<T, false, false, 0>(grid, stream, args, occ);
  else launch_slot_kernel<T, false, true, 0>(grid, stream, args, occ);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

template <int BM, int BN>
int launch(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
           long lda, long ldb, long ldc, int trans_a, int trans_b, int act, int splits,
           int kchunk, float* partial, int m_load, int n_load, hipStream_t stream) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;
  const int a_vec = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
  const int b_vec = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
  dim3 grid(ntiles * splits), block(NTHREADS);
  const GemmF32Args args{A, B, C, bias, M, N, K, lda, ldb, ldc, a_vec, b_vec, act, tiles_n, ntiles,
                         splits, kchunk, partial, 0.f, m_load, n_load};
  // op(A)(m,k): stored [M,K] -> K-contiguous; stored [K,M] (trans_a) -> M-contiguous
#define DTF_GEMM_CASE(AKC, BKC) launch_kernel<BM, BN, AKC, BKC, 0>(grid, block, stream, args)
#ifdef DTF_GEMM_EXPERIMENTS
  if (const char* e = getenv("DTF_F32_ABL")) {
    const int abl = atoi(e);
#define DTF_ABL_CASE(V)                                                                       \
  if (abl == V) {                                                                             \
    launch_kernel<BM, BN, true, false, V>(grid, block, stream, args);                         \
    return DTF_OK;                                                                            \
  }
    DTF_ABL_CASE(1) DTF_ABL_CASE(2) DTF_ABL_CASE(4) DTF_ABL_CASE(8) DTF_ABL_CASE(6)
    DTF_ABL_CASE(7) DTF_ABL_CASE(15) DTF_ABL_CASE(14) DTF_ABL_CASE(16)
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

// Block tile for an M x N x K product, from measurements on MI355X (profiles/
// gemm_f32_tiles_vs_hipblaslt.jsonl). The 128x128 tile on the slot-dealt kernel
// (gemm_f32_slot_block) keeps the matrix pipe 97-98 % busy (8192^3: 151 TF, 6144^3: 152,
// 4096^3: 148; the general kernel: 140, 139, 136) but holds a quarter or a half of a CU per
// workgroup for a long time, so it is chosen when its grid fills the chip evenly: up to 256 tiles
// run one per CU (2048^3: 123 TF against 112 with 64x64 tiles); more run two per CU, in rounds
// of 512, and a last round that is at most half full costs half a round (the stragglers have
// their CUs to themselves), a fuller one a whole round. Everything else takes 64x64 tiles, three
// or four workgroups to a CU (3072^3: 130 TF, 5120^3: 140).
DTF_API int dtf_gemm_f32_tile(int M, int N, int K) {
  if (gemm_f32_skinny_fits(M, N, K)) return 16;       // one 16x16 tile per workgroup, no split
  // one 32x32 tile per workgroup while those are at most one per CU and K is short: the 64x64
  // tiling would be at most 64 workgroups plus a K split. GPU time by kernel trace, this kernel /
  // 64x64 tiles with their split / torch.matmul: 384^3 8.0 / 11.4 / 6.1 us, 512^3 8.6 / 13.3 / 7.7,
  // 256 x 512 x 1024 9.0 / 13.5 / 7.8, 1024 x 512 x 256 8.7 / 13.7 / 7.5, 512 x 1024 x 512 13.5 /
  // 16.1 / 10.2; it loses from two tiles per CU on (640^3 17.1 / 16.1) and from K = 1536 on
  // (480 x 1536 x 480 20.1 / 18.4) -- `scripts/f32_tile32_trace.sh`
  if ((long)((M + 31) / 32) * ((N + 31) / 32) <= 256 && K <= 1024) return 32;
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (K >= 4 * BK) {       // (the slot-dealt kernel's prologue holds two K-tiles)
    if (t128 >= 230 && t128 <= 256) return 128;
    if (t128 > 256) {
      const long full = t128 / 512, rest = t128 % 512;
      // cost in half rounds against 2 * t128 / 512 half rounds of work: take it from 94 % on
      const long cost2 = 2 * full + (rest == 0 ? 0 : rest <= 256 ? 1 : 2);
      if (t128 * 100 >= cost2 * 256 * 94) return 128;
    }
  }
  return 64;
}

// Workgroups per CU the slot-dealt 128x128 kernel is launched with (see dtf_gemm_f32_tile)
static int gemm_f32_slot_occupancy(int M, int N) {
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  return t128 <= 256 ? 1 : 2;
}

// Split-K plan: number of K splits (1 = none). A 64x64-tile grid of at most 160 workgroups with a
// long K gains from up to 4 splits; tiny grids (the MLP's 100x100 outputs with K = 784) take as many splits as give
// every CU a workgroup, each split keeping at least four K-tiles.
DTF_API int dtf_gemm_f32_splits(int M, int N, int K, int* kchunk_out) {
  int splits = 1, kchunk = K;
  if (dtf_gemm_f32_tile(M, N, K) == 64 && K >= 256) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
    int want = 1;
    if (tiles <= 64) {
      want = (int)(256 / tiles);
      if (want > 32) want = 32;
    } else if (tiles <= 160 && K >= 2048) {
      // fewer workgroups than two thirds of the CUs and a long K: finer work items (1024 x 4096 x
      // 512, 128 tiles: 42 us split four ways, 71 unsplit; 768 x 2048 x 768, 144 tiles: 36 / 42;
      // with 200 or 256 tiles the split loses, 1280 x 2048 x 640 44 / 41, 1024 x 2048 x 1024 48 / 44). Larger grids, or a short K, run unsplit: three or four 64 x 64
      // workgroups share a CU, so grids of up to ~1000 tiles are one resident round whatever
      // their count, and a split costs them a second launch and a pass over the planes (GPU time
      // from a kernel trace, unsplit / split by the earlier ragged-round rule: 768^3 18.9 / 24.5
      // us, 896^3 20.9 / 30.1, 1280^3 48.7 / 54.5, 1664^3 86.7 / 100.2, 1920^3 120.9 / 134.5;
      // 1536^3 82.8 / 81.2 and 1792^3 114.0 / 113.0 were the rule's best cases --
      // `scripts/f32_split_trace.sh`)
      want = (int)(512 / tiles);
      if (want > 4) want = 4;
      if (want > K / 512) want = K / 512;
    }
    else if (tiles < 1024 && K >= 4096) {
      // a long K pays for the second launch, and a grid that is not a whole number of workgroups
      // per CU then gains from finer work items (576 tiles, 1536 x 4096 x 1536: 161 us split four
      // ways, 184 unsplit; 1536 x 8192 x 1536 303 / 359; 784 tiles, 1792 x 4096 x 1792 220 / 241);
      // an even one does not (512 tiles, 2048 x 4096 x 1024: 141 / 132)
      const long rounds = (tiles + 255) / 256;
      if (tiles * 10 < rounds * 256 * 9) want = 4;
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
// m_load / n_load: how far the loads may read along M of a [K,M] A / along N of a [K,N] B: M (N)
// itself, or M (N) rounded up to a multiple of 4 when every row -- the last one included -- is
// that long in memory (the caller knows its allocation). With the rounded-up extent an operand
// whose M (N) is not a multiple of 4 still takes the branch-free 16-byte loads; the columns past
// M (N) feed only outputs that are never stored. Ignored for K-contiguous operands.
DTF_API int dtf_gemm_f32_ex(const float* A, const float* B, float* C, const float* bias, int M,
                            int N, int K, long lda, long ldb, long ldc, int trans_a, int trans_b,
                            int act, float* partial, int splits, int tile, int m_load, int n_load,
                            hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return DTF_ERR_BAD_ARG;
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N) return DTF_ERR_BAD_ARG;
  if (!trans_a) m_load = M;
  if (trans_b) n_load = N;
  if (m_load < M || m_load > M + 3 || (trans_a && m_load > lda)) return DTF_ERR_BAD_ARG;
  if (n_load < N || n_load > N + 3 || (!trans_b && n_load > ldb)) return DTF_ERR_BAD_ARG;
  if (tile != 16 && tile != 32 && tile != 64 && tile != 128 && tile != 129 && tile != 256)
    tile = dtf_gemm_f32_tile(M, N, K);
  if (tile == 32) {      // one 32x32 tile per workgroup, K over four or eight waves, no split
    const int tiles_m = (M + SK32_T - 1) / SK32_T, tiles_n = (N + SK32_T - 1) / SK32_T;
    const SkinnyArgs args{A, B, C, bias, M, N, K, lda, ldb, ldc, act, 0.f, tiles_n};
    dim3 grid(tiles_m * tiles_n);
#define DTF_SKINNY32_CASE(AKC, BKC)                                                              \
  do {                                                                                           \
    if (gemm_f32_skinny32_waves(K) == 8)                                                         \
      hipLaunchKernelGGL((gemm_f32_skinny32_kernel<AKC, BKC, 8>), grid, dim3(512), 0, stream,    \
                         args);                                                                  \
    else                                                                                         \
      hipLaunchKernelGGL((gemm_f32_skinny32_kernel<AKC, BKC, 4>), grid, dim3(256), 0, stream,    \
                         args);                                                                  \
  } while (0)
    if (!trans_a && !trans_b) DTF_SKINNY32_CASE(true, false);
    else if (!trans_a && trans_b) DTF_SKINNY32_CASE(true, true);
    else if (trans_a && !trans_b) DTF_SKINNY32_CASE(false, false);
    else DTF_SKINNY32_CASE(false, true);
#undef DTF_SKINNY32_CASE
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if (tile == 16) {
    const int tiles_m = (M + SK_T - 1) / SK_T, tiles_n = (N + SK_T - 1) / SK_T;
    const SkinnyArgs args{A, B, C, bias, M, N, K, lda, ldb, ldc, act, 0.f, tiles_n};
    dim3 grid(tiles_m * tiles_n);
#define DTF_SKINNY_CASE(AKC, BKC)                                                                \
  do {                                                                                           \
    if (gemm_f32_skinny_waves(K) == 16)                                                          \
      hipLaunchKernelGGL((gemm_f32_skinny_kernel<AKC, BKC, 16>), grid, dim3(1024), 0, stream,    \
                         args);                                                                  \
    else                                                                                         \
      hipLaunchKernelGGL((gemm_f32_skinny_kernel<AKC, BKC, 4>), grid, dim3(256), 0, stream,      \
                         args);                                                                  \
  } while (0)
    if (!trans_a && !trans_b) DTF_SKINNY_CASE(true, false);
    else if (!trans_a && trans_b) DTF_SKINNY_CASE(true, true);
    else if (trans_a && !trans_b) DTF_SKINNY_CASE(false, false);
    else DTF_SKINNY_CASE(false, true);
#undef DTF_SKINNY_CASE
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  int kchunk = ((K + BK - 1) / BK) * BK;
  if (!partial || splits <= 1) {
    splits = 1;
  } else {
    kchunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (K + kchunk - 1) / kchunk;          // never more than the caller allotted
  }
  // 128 and 256: the slot-dealt kernel where the operands allow its loads (aligned rows,
  // contiguous dimensions multiples of 4), the general one otherwise; 129: the general kernel
  // with 128x128 tiles whatever the operands (for measurements)
  const bool slot_ok =
      gemm_f32_slot_ok(A, B, m_load, n_load, K, lda, ldb, !trans_a, trans_b != 0);
  if (tile == 256 && slot_ok)
    return launch_slot<256>(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, partial,
                            m_load, n_load, stream);
  if ((tile == 128 || tile == 256) && slot_ok && (!partial || splits <= 1))
    return launch_slot<128>(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, partial,
                            m_load, n_load, stream, gemm_f32_slot_occupancy(M, N));
  if (tile != 64)
    return launch<128, 128>(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, splits,
                            kchunk, partial, m_load, n_load, stream);
  return launch<64, 64>(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, splits,
                        kchunk, partial, m_load, n_load, stream);
}

// dtf_gemm_f32 for a product the 16x16-tile kernel takes (dtf_gemm_f32_tile() == 16), with A read
// through the step ring: entry a_index of the replay's table slot holds A's address (same
// extents and leading dimension at every replay). Anything else: DTF_ERR_UNSUPPORTED.
DTF_API int dtf_gemm_f32_skinny_ring(const void* ring_table, void* ring_ctr, void* ring_out,
                                     int ring, int a_index, int first, const float* B, float* C,
                                     const float* bias, int M, int N, int K, long lda, long ldb,
                                     long ldc, int trans_a, int trans_b, int act,
                                     hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !B || !C || !ring_table || !ring_ctr || ring < 1 ||
      (ring & (ring - 1)) || a_index < 0 || a_index >= DTF_RING_FEEDS)
    return DTF_ERR_BAD_ARG;
  if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N) || ldc < N) return DTF_ERR_BAD_ARG;
  if (dtf_gemm_f32_tile(M, N, K) != 16) return DTF_ERR_UNSUPPORTED;
  const DtfStepRing r{(const unsigned long long*)ring_table, (unsigned*)ring_ctr, (float*)ring_out,
                      ring, -1};
  const int tiles_m = (M + SK_T - 1) / SK_T, tiles_n = (N + SK_T - 1) / SK_T;
  const SkinnyArgs args{nullptr, B, C, bias, M, N, K, lda, ldb, ldc, act, 0.f, tiles_n};
  dim3 grid(tiles_m * tiles_n);
#define DTF_SKINNY_RING_CASE(AKC, BKC)                                                           \
  do {                                                                                           \
    if (gemm_f32_skinny_waves(K) == 16)                                                          \
      hipLaunchKernelGGL((gemm_f32_skinny_ring_kernel<AKC, BKC, 16>), grid, dim3(1024), 0,       \
                         stream, args, r, a_index, first);                                       \
    else                                                                                         \
      hipLaunchKernelGGL((gemm_f32_skinny_ring_kernel<AKC, BKC, 4>), grid, dim3(256), 0, stream, \
                         args, r, a_index, first);                                               \
  } while (0)
  if (!trans_a && !trans_b) DTF_SKINNY_RING_CASE(true, false);
  else if (!trans_a && trans_b) DTF_SKINNY_RING_CASE(true, true);
  else if (trans_a && !trans_b) DTF_SKINNY_RING_CASE(false, false);
  else DTF_SKINNY_RING_CASE(false, true);
#undef DTF_SKINNY_RING_CASE
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

DTF_API int dtf_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N,
                         int K, long lda, long ldb, long ldc, int trans_a, int trans_b, int act,
                         float* partial, int splits, int tile, hipStream_t stream) {
  return dtf_gemm_f32_ex(A, B, C, bias, M, N, K, lda, ldb, ldc, trans_a, trans_b, act, partial,
                         splits, tile, M, N, stream);
}
