// bf16 GEMM on the gfx950 matrix cores (fp32 accumulate): C[M,N] = A[M,K] * Bt[N,K]^T.
//
// This is the kernel behind the sharded-GEMM workloads (16384^3 bf16 and its per-rank
// K-/M-shards). Both operands are K-contiguous ("NT"): A is [M,K] row-major and the right-hand
// operand is held as Bt = B^T, [N,K] row-major -- the storage layout this framework gives a
// bf16 matrix variable that is used as a matmul right-hand side.
//
// Fast path (dtf_gemm_bf16_nt, tile 256x256x64, 512 threads = 8 waves as 2(M) x 4(N)):
//  * global -> LDS by LDS-DMA (16 bytes per lane, global_load_lds_dwordx4 or its buffer form
//    with SGPR base + one offset VGPR in the ping-pong kernel): no VGPR staging, two LDS buffers
//    (2 x 64 KiB), the loads of K-tile t+1 are in flight while the MFMAs of tile t run; one
//    barrier per K-tile.
//  * LDS image: row-major tile rows of 128 B (64 k), with the eight 16-byte chunks of row r
//    XOR-permuted: chunk c sits at position c ^ ((r >> 1) & 7). LDS-DMA writes lane-linearly
//    (wave base + lane * 16), so the permutation is applied to each lane's global SOURCE
//    address: lane i of a 1 KiB piece fetches row i/8, chunk (i%8) ^ swizzle -- consecutive
//    lanes read consecutive-or-permuted chunks of one 128-byte line, so every quad of lanes
//    stays inside one 64-byte segment (an earlier image that put consecutive lanes on
//    consecutive ROWS was conflict-free too but cost ~4x the address-coalescer time per
//    instruction and capped the kernel at 1.15 PF). A fragment read -- lane l takes row l&15,
//    chunk 4s + (l>>4) -- then hits 16 distinct 16-byte slots of the 256-byte bank row in
//    every 16-lane ds_read_b128 group: conflict-free (SQ_LDS_BANK_CONFLICT = 0).
//  * MFMA v_mfma_f32_16x16x32_bf16 with the operands swapped (Bt fragment as the instruction's
//    A operand, A fragment as its B operand): the accumulator then holds, per lane, four
//    consecutive n of one row m, so the epilogue issues 8-byte (bf16) / 16-byte (fp32) row
//    stores instead of 2-byte column scatters.
//  * workgroup id -> tile: XCD-aware. Large square-ish grids use the super-tile walk of
//    pp_tile_of_block (the eight XCDs as a 2 x 4 arrangement of 8 x 4-tile blocks that advance
//    together in a snake, so concurrently resident tiles share A/B panels inside an XCD's L2
//    and across XCDs through the memory-side cache); other grids a contiguous chunk of the
//    tile grid per XCD, walked 8 tile-rows deep.
//  * M and N edges are handled by clamping source rows (out-of-range rows are computed and
//    dropped at the store); K must be a multiple of 64.
//
// Kernels in this file, all sharing that LDS image and MFMA layout:
//  * gemm_nt_pp_kernel -- the default for large grids (bf16 and, instantiated with the
//    block-scaled MFMA, fp8): quadrant-phased ping-pong of two wave groups, persistent over
//    tiles, optional split-K work items for grids of at most 32 tiles (described below);
//  * gemm_bf16_nt_t128ws_kernel -- 128x128 tiles, four multiplying + four request-issuing
//    waves, for grids with a CU for every such tile and the work items of their K splits;
//  * gemm_bf16_nt_t128_kernel -- 128x128 tiles, two workgroups per CU, for grids of up to 128
//    large tiles with more than one 128x128 tile per CU;
//  * gemm_bf16_nt_kernel (variant 1) -- the plain double-buffered schedule described above;
//  * gemm_bf16_nt_w4_kernel (variant 3) -- one wave per SIMD with 128x128 outputs per wave, kept
//    with its measured ablations.
//
// Generic path (dtf_gemm_bf16_generic): 64x64 tile, register-staged, fully predicated; any
// shape/stride, NN/NT/TN; used for ragged problems and as an in-library cross-check.
#include <cstdlib>
#include "common.h"

#include <atomic>
#include <mutex>
#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_void;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int NTHREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;          // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;        // A + B
constexpr int WM = 128, WN = 64;                 // per-wave output
constexpr int MT = WM / 16, NT = WN / 16;        // 8 x 4 MFMA tiles per wave

// byte offset inside an operand tile of the 16-byte chunk (row, kc): see the header comment
__device__ __forceinline__ int chunk_off(int row, int kc) {
  return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
}

// Issue this wave's share of one operand tile: 4 LDS-DMA instructions of 1 KiB each.
// Piece c (0..31) covers rows 8c..8c+7; lane i lands at byte 16*i of the piece, i.e. row
// 8c + i/8, position i%8, and therefore fetches chunk (i%8) ^ swizzle(row) of that row.
__device__ __forceinline__ void stage_tile(const unsigned short* __restrict__ X, long ld, int row0,
                                           int max_row, int k0, char* lds_tile, int wave,
                                           int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = wave * 4 + j;
    const int trow = piece * 8 + (lane >> 3);     // tile-relative row
    int row = row0 + trow;
    row = row < max_row ? row : max_row;          // clamp: edge tiles re-read the last row
    const int kc = (lane & 7) ^ ((trow >> 1) & 7);
    const unsigned short* src = X + (long)row * ld + k0 + kc * 8;
    __builtin_amdgcn_global_load_lds((global_void*)src, (lds_void*)(lds_tile + piece * 1024), 16,
                                     0, 0);
  }
}

template <bool OUT_F32>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_nt_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ Bt,
    void* __restrict__ C, int M, int N, int K, long lda, long ldb, long ldc, int tiles_m,
    int tiles_n, float alpha) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A tile | B tile)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware, grouped tile walk
  const int ntiles = tiles_m * tiles_n;
  const int id = dtf_xcd_remap(blockIdx.x, ntiles);
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int g = id / per_group;
  const int first_m = g * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (id % per_group) % gsize;
  const int tn = (id % per_group) / gsize;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = K / BK;
  stage_tile(A, lda, m0, M - 1, 0, smem, wave, lane);
  stage_tile(Bt, ldb, n0, N - 1, 0, smem + TILE_BYTES, wave, lane);

  const int frow = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nkt; ++kt) {
    // every wave's LDS-DMA for tile kt has landed (hipcc drains vmcnt before the barrier) and
    // every wave is done reading the buffer tile kt+1 is about to overwrite
    __syncthreads();
    char* cur = smem + (kt & 1) * BUF_BYTES;
    if (kt + 1 < nkt) {
      char* nxt = smem + ((kt + 1) & 1) * BUF_BYTES;
      stage_tile(A, lda, m0, M - 1, (kt + 1) * BK, nxt, wave, lane);
      stage_tile(Bt, ldb, n0, N - 1, (kt + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* as = cur;
    const char* bs = cur + TILE_BYTES;
#pragma unroll
    for (int s = 0; s < BK / 32; ++s) {
      bf16x8 af[MT], bf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bf[j] = *reinterpret_cast<const bf16x8*>(bs + chunk_off(wc * WN + j * 16 + frow, 4 * s + fq));
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(as + chunk_off(wr * WM + i * 16 + frow, 4 * s + fq));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  // swapped-operand accumulator: lane holds C[m = frow][n = 4*fq + r], r = 0..3
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = m0 + wr * WM + i * 16 + frow;
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wc * WN + j * 16 + 4 * fq;
      if (col >= N) continue;
      f32x4 v = acc[i][j] * alpha;
      if (OUT_F32) {
        float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
        if (col + 3 < N) *reinterpret_cast<f32x4*>(dst) = v;
        else for (int r = 0; r < 4 && col + r < N; ++r) dst[r] = v[r];
      } else {
        unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
        if (col + 3 < N) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (short)dtf_f32_to_bf16(v[r]);
          *reinterpret_cast<bf16x4*>(dst) = o;
        } else {
          for (int r = 0; r < 4 && col + r < N; ++r) dst[r] = dtf_f32_to_bf16(v[r]);
        }
      }
    }
  }
}

// ---- quadrant-phased, group-staggered kernel ("ping-pong") -----------------------------------
// Same tile, LDS image, MFMA and epilogue as the kernel above; what changes is the schedule.
//  * A K-tile is four 16 KiB half-tiles (A rows 0-127 / 128-255, Bt rows 0-127 / 128-255). A
//    wave's 128x64 output is interleaved so that all eight waves work on the SAME block
//    quadrant at the same time: rows {wr*64 + [0,64)} and {128 + wr*64 + [0,64)}, columns
//    {wc*32 + [0,32)} and {128 + wc*32 + [0,32)}. Quadrant (mh, nh) therefore reads only
//    half-tiles A_mh and B_nh.
//  * A K-tile is four phases, quadrants (0,0) (0,1) (1,1) (1,0): each phase keeps one
//    operand's fragments in registers from the phase before (both Bt halves stay resident for
//    the whole K-tile), so the phases read 12 / 4 / 8 / 0 fragments from LDS and each runs 16
//    MFMAs. A half-tile's LDS slot is dead as soon as its fragments are in registers, which
//    lets the LDS-DMA stream run six half-tiles (1.5 K-tiles) ahead in just two K-tile buffers:
//    phase g issues half-tile g+6 and waits with a COUNTED s_waitcnt vmcnt(8) (four half-tiles
//    stay in flight across every barrier; the loop never drains to zero).
//  * Waves 4-7 (wr = 1; the partner of waves 0-3 on each SIMD) run one barrier behind waves
//    0-3. Each phase is [LDS reads + DMA issue] barrier [16 MFMAs] barrier, so while one wave
//    of a SIMD is in its MFMA segment its partner is in its load segment: the matrix pipe
//    always has a feeder, and the loads of one group hide under the MFMAs of the other.
//  * Hazards: a staged half-tile is read one full phase (two barriers) after the vmcnt that
//    retired it. A slot is restaged by a DMA issued behind a barrier that every wave passes only
//    after its own fragment reads have returned (s_waitcnt lgkmcnt(0) in front of the phase's
//    first barrier), and the slots a phase restages are those the PREVIOUS phases read: no DMA
//    can land in a slot with a read of it still queued, however short the memory latency is.
// ESIZE = bytes per element (2: bf16, 1: fp8); a K-tile is always 128 bytes of k per row, and
// lda / ldb are in elements.
// The DMA is a buffer load: its address is (block tile's base, an SGPR resource) + (K-tile's byte
// offset, an SGPR) + (32-bit per-lane offset in one VGPR). The per-lane part (row inside the block tile, clamped at the matrix edge,
// times the leading dimension, plus the swizzled 16-byte chunk) is loop-invariant and computed
// once per kind and piece, so staging a half-tile costs no vector ALU work and the instruction
// carries one address VGPR instead of a 64-bit pair.
// A_HALF_ROWS: rows from the first A half's first row to the second's (128, or 96 in the kernel
// of 192-row block tiles, which still stages 128 rows per half)
template <int ESIZE, int A_HALF_ROWS = 128>
__device__ __forceinline__ void stage_offsets(unsigned (&voff)[4][2], long lda, long ldb, int m0,
                                              int n0, int M, int N, int wave, int lane) {
#pragma unroll
  for (int kind = 0; kind < 4; ++kind) {
    const bool is_a = (kind == 0 || kind == 3);
    const int half = (kind >= 2) ? 1 : 0;
    const unsigned ld = (unsigned)((is_a ? lda : ldb) * ESIZE);
    const int base_row = is_a ? m0 : n0;
    const int max_row = (is_a ? M : N) - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = wave * 2 + j;               // 16 pieces of 8 rows per half-tile
      const int trow = piece * 8 + (lane >> 3);     // row inside the half-tile (128 = 0 mod 16)
      int row = base_row + half * (is_a ? A_HALF_ROWS : 128) + trow;
      row = row < max_row ? row : max_row;
      const int kc = (lane & 7) ^ ((trow >> 1) & 7);
      voff[kind][j] = (unsigned)(row - base_row) * ld + (unsigned)(kc * 16);
    }
  }
}

// a_tile / b_tile: unbounded raw buffer resources whose base is the block tile's first row
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const char* base) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0xffffffff, 0x00020000);
}

// SHARED_HALVES: the second 128-row half of an operand tile is staged with the FIRST half's
// per-lane offsets plus a wave-uniform 128 * ld bytes in the scalar offset -- valid when no row of
// the tile is clamped at the matrix edge (M and N multiples of the tile) -- so that only
// voff[0] (A) and voff[1] (Bt) stay live across the K loop: four registers instead of eight.
template <bool SHARED_HALVES = false>
__device__ __forceinline__ void stage_half(__amdgpu_buffer_rsrc_t a_tile,
                                           __amdgpu_buffer_rsrc_t b_tile,
                                           const unsigned (&voff)[4][2], int h, int nkt, int kind,
                                           int buf, char* smem, int wave, int a_half_bytes = 0,
                                           int b_half_bytes = 0) {
  int t = h >> 2;
  t = t < nkt ? t : nkt - 1;        // past the end: restage a dead slot with valid memory
  const bool is_a = (kind == 0 || kind == 3);
  const int half = (kind >= 2) ? 1 : 0;
  int k0 = t * 128;                 // 128 bytes of k per K-tile
  if (SHARED_HALVES && half) k0 += is_a ? a_half_bytes : b_half_bytes;
  const int vk = SHARED_HALVES ? (is_a ? 0 : 1) : kind;
  char* dst = smem + buf * BUF_BYTES + (is_a ? 0 : TILE_BYTES) + half * (TILE_BYTES / 2);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int piece = wave * 2 + j;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(is_a ? a_tile : b_tile,
                                             (lds_void*)(dst + piece * 1024), 16, voff[vk][j],
                                             k0, 0, 0);
  }
}

// Workgroup id -> output tile of the ping-pong kernel.
// Workgroups go to the 8 XCDs round-robin (id % 8) and, inside an XCD, start in id order on its
// 32 CUs; an XCD's 32 resident tiles form an 8(M) x 4(N) block that shares 8 A panels and 4 Bt
// panels through its L2 (12 panel loads per 32 tile loads) in both walks below.
//  * Chunk walk: a contiguous chunk of the tile grid per XCD, walked 8 tile-rows deep. With 64 or
//    more tile rows every XCD then streams a private strip of A (8 strips of 8 x 256 rows x K
//    at once -- 512 MiB at K = 16384 bf16, more than the memory-side cache holds).
//  * Super-tile walk (grids of >= 32 tile rows whose sides divide by 16): the 256 tiles in flight
//    on the chip form ONE 16 x 16 super-tile, XCDs arranged 2(M) x 4(N), so that XCDs share A
//    and Bt panels through the memory-side cache; super-tiles are walked in a snake along N so
//    that consecutive ones reuse the same A strip. Measured against the chunk walk: 16384^3
//    bf16 1533-1539 vs 1509-1513 TF, 16384 x 16384 x 2048 1288 vs 1226 TF, 16384^3 fp8 3050 vs
//    2931 TF; 8192^3 bf16 (32 tile rows) 1582-1602 vs 1577-1592 TF; no gain (or 1-2 % loss) on
//    grids with fewer tile rows, which keep the chunk walk.
__device__ __forceinline__ void pp_tile_of_block(int bid, int tiles_m, int tiles_n, int& tm,
                                                 int& tn, bool chunk_walk_only,
                                                 int min_super_rows = 32) {
  if (!chunk_walk_only && tiles_m >= min_super_rows && tiles_m % 16 == 0 && tiles_n % 16 == 0) {
    constexpr int xm = 2, xn = 4;
    const int xcd = bid & 7, local = bid >> 3;
    const int sup = local >> 5, within = local & 31;
    const int sup_n = tiles_n / (xn * 4);
    const int sm = sup / sup_n;
    int sn = sup - sm * sup_n;
    if (sm & 1) sn = sup_n - 1 - sn;
    tm = (sm * xm + xcd / xn) * 8 + (within & 7);
    tn = (sn * xn + xcd % xn) * 4 + (within >> 3);
    return;
  }
  const int id = dtf_xcd_remap(bid, tiles_m * tiles_n);
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  tm = first_m + (id % per_group) % gsize;
  tn = (id % per_group) / gsize;
}

// FP8 = false: bf16 operands, K-tile 64, two v_mfma_f32_16x16x32_bf16 per output tile and K-tile.
// FP8 = true : OCP e4m3 operands, K-tile 128 (the same 128 bytes per row, so the LDS image, the
//   DMA stream and the fragment reads are byte-for-byte those of the bf16 kernel); the two
//   16-byte fragment reads of a lane (k bytes 16q..16q+15 and 64+16q..64+16q+15, q = lane>>4)
//   form the 32-byte operand of ONE v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales
//   (E8M0 0x7f). A and Bt lanes of the same q hold the same k set, which is all the
//   instruction's k-assignment requires of a full-K reduction. That MFMA takes twice the
//   cycles of the bf16 one at 4x the k, so a K-tile costs the same matrix-pipe time and the
//   same bytes as in bf16 while doing twice the flops.
// MERGE = true (the default schedule; measured 1511-1515 TF against 1499 TF at 16384^3 bf16):
//   two quadrants (32 MFMAs) per phase instead of one -- half as many barrier crossings per
//   K-tile. A phase then reads (B0, A0, B1) or (A1) and stages two half-tiles;
//   the half-tile stream still runs six ahead (1.5 K-tiles), with vmcnt(6) = three half-tiles
//   in flight across each barrier.
// PERSIST = FP8 ? false : true: the fp8 instantiation already uses 254 VGPRs (its MFMA takes
//   8-register operands) and spills with the few more the tile loop keeps live, so it runs one
//   tile per workgroup. (A persistent fp8 build with one set of staging offsets per operand
//   spilled six registers outside the K loop and measured the same as this one: 3133 against
//   3131 TF at 16384^3, 3052 against 3030 at 8192^3; built again on top of the short interior-tile
//   epilogue it filled all 256 VGPRs, spilled four and ran at 2617 against 3119.)
// MX = true (FP8 only): block-scaled operands in the OCP MX manner -- every 32 consecutive k of a
//   row share one E8M0 scale (a power of two), applied by the MFMA itself. The instruction takes
//   the scale of (row r, 32-block b of its 128 k) from lane 16 b + r of a scale register, one byte
//   selected by an immediate. Scales are stored ready for that (dtf_quantize_mx_fp8 writes them
//   so): per K-tile and 256-row block tile 256 dwords; a wave's lane l = 16 b + r finds in ONE
//   dword the four scales it needs for a half of the tile -- for A the rows (MH, wr, i = 0..3,
//   r), byte i; for Bt the rows (NH, wc, j, r), byte 2 NH + j. Per K-tile a wave moves three such
//   dwords per lane (A half 0, A half 1, Bt) into a private LDS strip with three 4-byte LDS-DMA
//   instructions, one K-tile ahead of their use, and reads them back with ds_read_b32 next to the
//   fragment reads. The three instructions enter the counted vmcnt waits of the phases that issue
//   them (they sit in the queue before that phase's tile DMA, so those phases wait for
//   vmcnt(9) instead of vmcnt(6)).
// MI = 16-row MFMA tiles per wave and A half (4: block tiles of 256 rows; 3: of 192 rows; 2: of
//   128 rows -- for grids whose 256-row tiling leaves much of a round of workgroups idle: 3072^2
//   outputs are 144 tiles of 256 x 256 on 256 CUs but 192 tiles of 192 x 256, 2048 x 4096 are 128
//   or 256). A half of a 192-row tile is 96 rows (64 of a 128-row tile); it is still STAGED as 128
//   rows (the LDS image, the DMA stream and
//   its counted waits stay those of the 256-row kernel; the 32 extra rows are the next half's or
//   the next tile's, clamped at the matrix edge, and are not read).
template <bool OUT_F32, int DBG, bool FP8, bool MERGE = false, bool MX = false, int MI = 4>
__global__ __launch_bounds__(NTHREADS) void gemm_nt_pp_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt,
    void* __restrict__ C, int M, int N, int K, long lda, long ldb, long ldc, int tiles_m,
    int tiles_n, float alpha, float beta, int ksplits, long plane,
    const unsigned* __restrict__ sa, const unsigned* __restrict__ sb, int sa_pair = 0,
    int sb_quad = 0, const unsigned long long* __restrict__ mirrors = nullptr, int nmirrors = 0) {
  // mirrors / nmirrors: addresses (a table in device memory) of further copies of C -- the same
  // rows of the result in the peers' memory, mapped into this process. Every vector store of the
  // epilogue is repeated into each of them, so a rank's row block reaches its peers from the
  // product's own epilogue, tile by tile, with no copy pass behind it (beta = 0, no K split).
  // sa_pair / sb_quad: bytes from one K-tile pair (A) / quad (Bt) of scales to the next -- the
  // row-tile count of the matrix the scales were made for, which may be taller than this call's
  static_assert(!MX || (FP8 && MERGE), "block scales: fp8 operands, merged schedule");
  static_assert(MI == 4 || ((MI == 3 || MI == 2) && !MX && MERGE),
                "192- and 128-row tiles: merged schedule, no block scales");
  constexpr int BM_ROWS = 64 * MI, HALF_ROWS = 32 * MI;    // block tile / A half, in rows
  // ksplits > 1 (few output tiles, long K): work item (split, tile) multiplies K-range `split`
  // of K / ksplits columns and writes its partial tile to C + split * plane (elements); a
  // second kernel sums the planes in split order.
  constexpr int ESIZE = FP8 ? 1 : 2;
  constexpr bool PERSIST = !FP8 && !(DBG & 32);     // DBG bit 5: one tile per workgroup
  // DBG (timing experiments only, built with -DDTF_GEMM_EXPERIMENTS; results are wrong when
  // set): bit 0 = issue no LDS-DMA in the K-loop, bit 1 = always stage K-tile 0 (every load an
  // L2 hit), bit 2 = skip the counted wait, bit 3 = per-phase cycle sums, bit 4 = chunk walk
  // schedule A/B bits (results stay correct): bit 8 = raise the wave's priority over its MFMA
  // segment (s_setprio 1 / 0), bit 11 = issue a phase's LDS reads before its DMA, bit 12 = the
  // super-tile walk only from 64 tile rows on. The merged schedule ships with the first two OFF
  // (the block-scaled instantiation, whose counted waits assume the scales' DMA is queued ahead
  // of the tile's, keeps them): measured on one box at 16384^3 / 8192^3 / 4096^3 bf16, median of
  // 7 interleaved rounds, 1572 / 1586 / 1504 TF against 1556 / 1569 / 1493 TF with both on
  // (`profiles/gemm_bf16_pp_tweaks.jsonl`); fp8 16384^3 3159 against 3131 TF.
  constexpr int dbg = DBG;
  constexpr bool SETPRIO = !MERGE || MX || (DBG & 256) != 0;
  // (DMA ahead of the reads only on 256-row tiles: with the shorter phases of the 192- and 128-row
  // tiles the reads belong first -- 3072^3 1191 against 1166 TF, 1024 x 8192 x 8192 1302 against
  // 1223 -- and the LDS wait in front of the barrier then costs nothing. Bit 6: no LDS wait,
  // A/B only)
  constexpr bool DMA_FIRST = MERGE && !MX && MI == 4 && (DBG & 2048) == 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // Persistent over tiles: workgroup b computes tiles b, b + gridDim, b + 2 gridDim, ... (the
  // launcher starts one workgroup per CU). When a tile's K loop ends, every LDS slot is dead,
  // so the first six half-tiles of the NEXT tile are requested before the epilogue of this one:
  // their memory latency hides behind the 128 KiB of result stores instead of opening each
  // tile with an idle matrix pipe.
  const int ntiles_xy = tiles_m * tiles_n;
  const int ntiles = ntiles_xy * ksplits;
  const int ksub = K / ksplits;                       // columns of K per split
  const int nkt = ksub / (128 / ESIZE);
  int tile = blockIdx.x;
  int m0, n0, split;
  {
    int tm, tn;
    split = tile / ntiles_xy;
    pp_tile_of_block(tile - split * ntiles_xy, tiles_m, tiles_n, tm, tn, (DBG & 16) != 0,
                     (DBG & 4096) ? 64 : 32);
    m0 = tm * BM_ROWS;
    n0 = tn * BN;
  }
  const int first_m0 = m0, first_n0 = n0;

  f32x4 acc[2][2][MI][2];
  i32x4 af[MI][2];        // [m-tile][16-byte k-chunk] of the current A half
  i32x4 bf[2][2][2];      // [n-half][n-tile][k-chunk]: both Bt halves stay resident

  const int frow = lane & 15, fq = lane >> 4;
  // per-lane part of chunk_off() for k-step 0 / 1 (the XOR makes them differ by +-64 B); the
  // wave-uniform and compile-time parts are added per read
  const int lane_off0 = chunk_off(frow, fq), lane_off1 = chunk_off(frow, 4 + fq);
  const int a_wave = (wr * MI) * 2048;                 // wave's first 16-row block in an A half
  const int b_wave = TILE_BYTES + (wc * 2) * 2048;     // ... in a Bt half

  // timing experiment (dbg & 8): per-wave, per-phase cycle sums of the five segments of a phase
  // [issue reads + DMA | counted wait | barrier | MFMAs | barrier] and the phase count
  unsigned seg[4][6] = {};
  unsigned voff[4][2];
  stage_offsets<ESIZE, HALF_ROWS>(voff, lda, ldb, m0, n0, M, N, wave, lane);
  // (MX: tiles never straddle the matrix edge; the halves share their per-lane offsets)
  const int a_half_bytes = (int)(128 * lda * ESIZE), b_half_bytes = (int)(128 * ldb * ESIZE);
  __amdgpu_buffer_rsrc_t a_tile = tile_rsrc(A + ((long)m0 * lda + (long)split * ksub) * ESIZE);
  __amdgpu_buffer_rsrc_t b_tile = tile_rsrc(Bt + ((long)n0 * ldb + (long)split * ksub) * ESIZE);

  // block scales: this wave's strip of LDS, 16 bytes per lane and buffer,
  //   A:  [pair parity][64 lanes][(K-tile & 1, A half, i)]    one 16-byte LDS-DMA per PAIR of K-tiles
  //   Bt: [quad parity][64 lanes][(K-tile & 3, Bt half, j)]   one 16-byte LDS-DMA per FOUR K-tiles
  // ONE per-lane register, sc_lane = strip offset + 16 * lane, serves as LDS address of the
  // lane's slot (plus constants) and as the DMA's per-lane offset (the buffer bases are set back
  // by the strip's offset to make up for it).
  constexpr int SC_WAVE = 4 * 1024;
  const int sc_strip = 2 * BUF_BYTES + wave * SC_WAVE;
  const unsigned sc_lane = (unsigned)sc_strip + (unsigned)lane * 16u;
  __amdgpu_buffer_rsrc_t sa_tile = a_tile, sb_tile = b_tile;
  // stage the A scales of K-tile pair pr / the Bt scales of K-tile quad qd (clamped at the end:
  // a dead slot is refilled)
  auto stage_scales_a = [&](int pr) {
    const int last = (nkt - 1) >> 1;
    const int pc = (pr < last ? pr : last) + split * ((nkt + 1) >> 1);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(sa_tile, (lds_void*)(smem + sc_strip + (pr & 1) * 1024),
                                             16, sc_lane, pc * sa_pair, 0, 0);
  };
  auto stage_scales_b = [&](int qd) {
    const int last = (nkt - 1) >> 2;
    const int qc = (qd < last ? qd : last) + split * ((nkt + 3) >> 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(sb_tile,
                                             (lds_void*)(smem + sc_strip + 2048 + (qd & 1) * 1024),
                                             16, sc_lane, qc * sb_quad, 0, 0);
  };
  if constexpr (MX) {
    sa_tile = tile_rsrc(reinterpret_cast<const char*>(sa) + (long)(m0 / BM) * 2048 + wr * 1024 -
                        sc_strip);
    sb_tile = tile_rsrc(reinterpret_cast<const char*>(sb) + (long)(n0 / BN) * 4096 + wc * 1024 -
                        sc_strip);
    stage_scales_a(0);             // in front of the prologue's DMA: landed with half-tile 0
    stage_scales_b(0);
  }
  unsigned sc_a = 0x7f7f7f7fu, sc_b = 0x7f7f7f7fu;     // the scales of the running phase

  // prologue of the first tile: half-tiles 0..5 in flight
#pragma unroll
  for (int h = 0; h < 6; ++h)
    stage_half<MX>(a_tile, b_tile, voff, h, nkt, h & 3, (h >> 2) & 1, smem, wave, a_half_bytes, b_half_bytes);
  bool first = true;

  for (;;) {
  if (first) {
    if constexpr (MERGE) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // half-tiles 0-2 landed
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                   // half-tiles 0-1 landed
  } else {
    // prologue requested before the previous epilogue: wait for it and for those stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  first = false;
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define DTF_READ_A(MH, BUFI)                                                                     \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) \
      af[i][s2] = *reinterpret_cast<const i32x4*>(smem + (BUFI) * BUF_BYTES + (MH) * 16384 +     \
                                                  a_wave + i * 2048 +                           \
                                                  (s2 ? lane_off1 : lane_off0));
#define DTF_READ_B(NH, BUFI)                                                                     \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) \
      bf[NH][j][s2] = *reinterpret_cast<const i32x4*>(smem + (BUFI) * BUF_BYTES + (NH) * 16384 + \
                                                      b_wave + j * 2048 +                       \
                                                      (s2 ? lane_off1 : lane_off0));
// (the scale-byte selectors of the block-scaled MFMA are immediates: one macro per tile)
#define DTF_MFMA_F8(MH, NH, I, J)                                                                \
  acc[MH][NH][I][J] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                          \
      __builtin_shufflevector(bf[NH][J][0], bf[NH][J][1], 0, 1, 2, 3, 4, 5, 6, 7),               \
      __builtin_shufflevector(af[I][0], af[I][1], 0, 1, 2, 3, 4, 5, 6, 7), acc[MH][NH][I][J], 0, \
      0, MX ? 2 * (NH) + (J) : 0, sc_b, MX ? (I) : 0, sc_a);
#define DTF_MFMA_QUAD(MH, NH)                                                                    \
  if constexpr (FP8) {                                                                           \
    DTF_MFMA_F8(MH, NH, 0, 0) DTF_MFMA_F8(MH, NH, 0, 1) DTF_MFMA_F8(MH, NH, 1, 0)                \
    DTF_MFMA_F8(MH, NH, 1, 1)                                                                    \
    if constexpr (MI > 2) { DTF_MFMA_F8(MH, NH, 2, 0) DTF_MFMA_F8(MH, NH, 2, 1) }                \
    if constexpr (MI > 3) { DTF_MFMA_F8(MH, NH, 3, 0) DTF_MFMA_F8(MH, NH, 3, 1) }                \
  } else {                                                                                       \
    _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) _Pragma("unroll") for (int i = 0; i < MI;   \
                                                                            ++i)                 \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[MH][NH][i][j] =                        \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16((bf16x8)bf[NH][j][s2], (bf16x8)af[i][s2],    \
                                                    acc[MH][NH][i][j], 0, 0, 0);                 \
  }
// phase P (0..7) of an iteration that starts at global phase g0: LDS reads, DMA issue of
// half-tile g0+P+6, counted wait, barrier, 16 MFMAs, barrier
#define DTF_PHASE(P, READS, MH, NH)                                                              \
  {                                                                                              \
    READS;                                                                                       \
    if (!(dbg & 1))                                                                              \
      stage_half<MX>(a_tile, b_tile, voff, (dbg & 2) ? 0 : g0 + (P) + 6, nkt, ((P) + 2) & 3,         \
                 (((P) + 6) >> 2) & 1, smem, wave, a_half_bytes, b_half_bytes);                                              \
    if (!(dbg & 4)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    /* (as in the merged phase below) */   \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_barrier();                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_setprio(1);                                                               \
    DTF_MFMA_QUAD(MH, NH)                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_barrier();                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
  }

// merged phase P (0..3) of an iteration that starts at global (single-quadrant) phase g0:
// stages half-tiles g0 + 2P + 6 and g0 + 2P + 7
#define DTF_PHASE2(P, READS, MH, NH0, NH1)                                                       \
  {                                                                                              \
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;                     \
    if constexpr ((dbg & 8) != 0) ts0 = __builtin_amdgcn_s_memtime();                            \
    if constexpr (DMA_FIRST && !(dbg & 1)) {                                                     \
      stage_half<MX>(a_tile, b_tile, voff, (dbg & 2) ? 0 : g0 + 2 * (P) + 6, nkt,                \
                     (2 * (P) + 2) & 3, ((2 * (P) + 6) >> 2) & 1, smem, wave, a_half_bytes,      \
                     b_half_bytes);                                                              \
      stage_half<MX>(a_tile, b_tile, voff, (dbg & 2) ? 0 : g0 + 2 * (P) + 7, nkt,                \
                     (2 * (P) + 3) & 3, ((2 * (P) + 7) >> 2) & 1, smem, wave, a_half_bytes,      \
                     b_half_bytes);                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
    READS;                                                                                       \
    if constexpr (MX) {                                                                          \
      /* K-tile of this phase: kt = (g0 >> 2) + (P >> 1), of pair g0 >> 3 and quad g0 >> 4;     \
         its A half is P & 1 */                                                                  \
      sc_a = *reinterpret_cast<const unsigned*>(smem + sc_lane + ((g0 >> 3) & 1) * 1024 +        \
                                                ((P) >> 1) * 8 + ((P) & 1) * 4);                 \
      sc_b = *reinterpret_cast<const unsigned*>(smem + sc_lane + 2048 + ((g0 >> 4) & 1) * 1024 + \
                                                ((g0 >> 3) & 1) * 8 + ((P) >> 1) * 4);           \
      if ((P) == 0) {                                                                            \
        stage_scales_a((g0 >> 3) + 1);                                                           \
        if ((g0 >> 3) & 1) stage_scales_b((g0 >> 4) + 1);                                        \
      }                                                                                          \
    }                                                                                            \
    if (!(dbg & 1) && !DMA_FIRST) {                                                              \
      stage_half<MX>(a_tile, b_tile, voff, (dbg & 2) ? 0 : g0 + 2 * (P) + 6, nkt,                    \
                 (2 * (P) + 2) & 3, ((2 * (P) + 6) >> 2) & 1, smem, wave, a_half_bytes, b_half_bytes);                       \
      stage_half<MX>(a_tile, b_tile, voff, (dbg & 2) ? 0 : g0 + 2 * (P) + 7, nkt,                    \
                 (2 * (P) + 3) & 3, ((2 * (P) + 7) >> 2) & 1, smem, wave, a_half_bytes, b_half_bytes);                       \
    }                                                                                            \
    if constexpr ((dbg & 8) != 0) ts1 = __builtin_amdgcn_s_memtime();                            \
    if (!(dbg & 4)) {                                                                            \
      if (MX && (P) == 0) {                                                                      \
        if ((g0 >> 3) & 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                      \
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                                    \
      } else {                                                                                   \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                         \
      }                                                                                          \
    }                                                                                            \
    if constexpr ((dbg & 8) != 0) ts2 = __builtin_amdgcn_s_memtime();                            \
    /* the fragment reads of slots that are restaged behind this barrier have RETURNED before   \
       this wave passes it (phases 0 / 2: the B1 half's four reads, restaged a phase later,     \
       may stay in flight): a DMA can only overwrite slots nobody is still reading */            \
    if constexpr ((dbg & 64) == 0) {                                                             \
      /* (block scales: the two scale reads behind the fragment reads are of the wave's own     \
         strip and may stay in flight as well) */                                                \
      if constexpr (MX) {                                                                        \
        if ((P) & 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                          \
        else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");                                  \
      } else {                                                                                   \
        if ((P) & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                  \
      }                                                                                          \
    }                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    __builtin_amdgcn_s_barrier();                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if constexpr ((dbg & 8) != 0) ts3 = __builtin_amdgcn_s_memtime();                            \
    if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(1);                                        \
    DTF_MFMA_QUAD(MH, NH0)                                                                       \
    DTF_MFMA_QUAD(MH, NH1)                                                                       \
    if constexpr (SETPRIO) __builtin_amdgcn_s_setprio(0);                                        \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if constexpr ((dbg & 8) != 0) ts4 = __builtin_amdgcn_s_memtime();                            \
    __builtin_amdgcn_s_barrier();                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    if constexpr ((dbg & 8) != 0) {                                                              \
      ts5 = __builtin_amdgcn_s_memtime();                                                        \
      seg[P][0] += (unsigned)(ts1 - ts0);                                                        \
      seg[P][1] += (unsigned)(ts2 - ts1);                                                        \
      seg[P][2] += (unsigned)(ts3 - ts2);                                                        \
      seg[P][3] += (unsigned)(ts4 - ts3);                                                        \
      seg[P][4] += (unsigned)(ts5 - ts4);                                                        \
      seg[P][5] += 1u;                                                                           \
    }                                                                                            \
  }

  int g0 = 0;
  if constexpr (MERGE) {
    for (int it = 0; it < nkt / 2; ++it, g0 += 8) {
      DTF_PHASE2(0, DTF_READ_B(0, 0) DTF_READ_A(0, 0) DTF_READ_B(1, 0), 0, 0, 1)
      DTF_PHASE2(1, DTF_READ_A(1, 0), 1, 1, 0)
      DTF_PHASE2(2, DTF_READ_B(0, 1) DTF_READ_A(0, 1) DTF_READ_B(1, 1), 0, 0, 1)
      DTF_PHASE2(3, DTF_READ_A(1, 1), 1, 1, 0)
    }
    if (nkt & 1) {
      // (the lane's LDS offsets are recomputed for this tail from the execution mask: kept
      // live across the main loop for it, they would not fit the register budget)
      int t_lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(t_lane));
      const int lane_off0 = chunk_off(t_lane & 15, t_lane >> 4);
      const int lane_off1 = chunk_off(t_lane & 15, 4 + (t_lane >> 4));
      const unsigned sc_lane = (unsigned)sc_strip + (unsigned)t_lane * 16u;
      DTF_PHASE2(0, DTF_READ_B(0, 0) DTF_READ_A(0, 0) DTF_READ_B(1, 0), 0, 0, 1)
      DTF_PHASE2(1, DTF_READ_A(1, 0), 1, 1, 0)
    }
  } else {
  for (int it = 0; it < nkt / 2; ++it, g0 += 8) {
    DTF_PHASE(0, DTF_READ_B(0, 0) DTF_READ_A(0, 0), 0, 0)
    DTF_PHASE(1, DTF_READ_B(1, 0), 0, 1)
    DTF_PHASE(2, DTF_READ_A(1, 0), 1, 1)
    DTF_PHASE(3, , 1, 0)
    DTF_PHASE(4, DTF_READ_B(0, 1) DTF_READ_A(0, 1), 0, 0)
    DTF_PHASE(5, DTF_READ_B(1, 1), 0, 1)
    DTF_PHASE(6, DTF_READ_A(1, 1), 1, 1)
    DTF_PHASE(7, , 1, 0)
  }
  if (nkt & 1) {   // odd K-tile count: the last K-tile has an even index -> buffer 0
    DTF_PHASE(0, DTF_READ_B(0, 0) DTF_READ_A(0, 0), 0, 0)
    DTF_PHASE(1, DTF_READ_B(1, 0), 0, 1)
    DTF_PHASE(2, DTF_READ_A(1, 0), 1, 1)
    DTF_PHASE(3, , 1, 0)
  }
  }
#undef DTF_PHASE2
#undef DTF_PHASE
#undef DTF_MFMA_QUAD
#undef DTF_MFMA_F8
#undef DTF_READ_A
#undef DTF_READ_B
  if (wr == 0) __builtin_amdgcn_s_barrier();          // every wave has passed the same count
  // every wave has finished reading LDS: request the next tile's first half-tiles now
  const int cm0 = m0, cn0 = n0;                        // this tile's origin, for the epilogue
  char* const c_plane = reinterpret_cast<char*>(C) + (long)split * plane * (OUT_F32 ? 4 : 2);
  tile += gridDim.x;
  const bool more = PERSIST && tile < ntiles;
  if (more) {
    int tm, tn;
    split = tile / ntiles_xy;
    pp_tile_of_block(tile - split * ntiles_xy, tiles_m, tiles_n, tm, tn, (DBG & 16) != 0,
                     (DBG & 4096) ? 64 : 32);
    m0 = tm * BM_ROWS;
    n0 = tn * BN;
    int o_lane = lane;                    // opaque copy: keeps the offset arithmetic out of the
    asm volatile("" : "+v"(o_lane));      // K loop's live registers (see the epilogue)
    stage_offsets<ESIZE, HALF_ROWS>(voff, lda, ldb, m0, n0, M, N, wave, o_lane);
    a_tile = tile_rsrc(A + ((long)m0 * lda + (long)split * ksub) * ESIZE);
    b_tile = tile_rsrc(Bt + ((long)n0 * ldb + (long)split * ksub) * ESIZE);
#pragma unroll
    for (int h = 0; h < 6; ++h)
      stage_half<MX>(a_tile, b_tile, voff, h, nkt, h & 3, (h >> 2) & 1, smem, wave, a_half_bytes, b_half_bytes);
  }
  // epilogue: lane holds C[row = frow][col = 4*fq + r] of each 16x16 tile. The lane's row and
  // column pass through an opaque copy so that the address arithmetic stays HERE: hoisted out
  // of the tile loop it would occupy ~20 VGPRs across the K loop (the fp8 instantiation then
  // spills, and scratch reloads inside the K loop would also break its counted vmcnt waits).
  // (the lane id is recomputed from the execution mask here rather than kept in a register
  // across the K loop)
  int e_lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(e_lane));
  const int e_frow = e_lane & 15, e_fq = e_lane >> 4;
  const bool wide_bf16 = !OUT_F32 && beta == 0.f && (ldc & 7) == 0 &&
                         (reinterpret_cast<uintptr_t>(c_plane) & 15) == 0;
  // Interior tile, bf16 out, no mirrors: no per-lane bounds, one row address per 16 rows,
  // hardware bf16 converts -- a fifth of the instructions of the general path below (whose
  // execution-mask branches alone outnumber them). Nothing hides a persistent workgroup's
  // epilogue, so its length is paid once per tile: measured per 256 x 256 tile on the four-wave
  // kernel, 48 thousand cycles with the general path against 28 thousand with this one.
  if (wide_bf16 && nmirrors == 0 && cm0 + BM_ROWS <= M && cn0 + BN <= N) {
    const int colx = (e_fq & 1) ? 16 + 4 * (e_fq - 1) : 4 * e_fq;
    char* const cbase = c_plane + ((long)(cm0 + wr * (16 * MI) + e_frow) * ldc + cn0 + wc * 32 +
                                   colx) * 2;
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        char* const crow = cbase + (long)(mh * HALF_ROWS + i * 16) * ldc * 2;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          const f32x4 v0 = acc[mh][nh][i][0] * alpha, v1 = acc[mh][nh][i][1] * alpha;
          const unsigned p0a = dtf_f32x2_to_bf16x2(v0[0], v0[1]);
          const unsigned p0b = dtf_f32x2_to_bf16x2(v0[2], v0[3]);
          const unsigned p1a = dtf_f32x2_to_bf16x2(v1[0], v1[1]);
          const unsigned p1b = dtf_f32x2_to_bf16x2(v1[2], v1[3]);
          const auto sa = __builtin_amdgcn_permlane16_swap(p0a, p1a, false, false);
          const auto sb = __builtin_amdgcn_permlane16_swap(p0b, p1b, false, false);
          u32x4 o = {sa[0], sb[0], sa[1], sb[1]};
          if constexpr ((DBG & 8192) == 0) *reinterpret_cast<u32x4*>(crow + nh * 256) = o;
          else if (alpha == 123.f) *reinterpret_cast<u32x4*>(crow + nh * 256) = o;
        }
      }
  } else
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = cm0 + mh * HALF_ROWS + wr * (16 * MI) + i * 16 + e_frow;
      if (row >= M) continue;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        if (!OUT_F32) {
          // bf16 out: a lane holds 4 consecutive columns of each of its two adjacent 16x16
          // tiles (8 bytes + 8 bytes, 32 apart). One v_permlane16_swap per register pair
          // trades the odd 16-lane rows' first tile against the even rows' second one, after
          // which every lane owns 8 consecutive columns: one 16-byte store per lane, 64
          // contiguous bytes per output row and instruction instead of two stores of 32.
          const int cb = cn0 + nh * 128 + wc * 32;
          if (wide_bf16 && cb + 32 <= N) {
            const f32x4 v0 = acc[mh][nh][i][0] * alpha, v1 = acc[mh][nh][i][1] * alpha;
            const unsigned p0a = dtf_f32_to_bf16(v0[0]) | ((unsigned)dtf_f32_to_bf16(v0[1]) << 16);
            const unsigned p0b = dtf_f32_to_bf16(v0[2]) | ((unsigned)dtf_f32_to_bf16(v0[3]) << 16);
            const unsigned p1a = dtf_f32_to_bf16(v1[0]) | ((unsigned)dtf_f32_to_bf16(v1[1]) << 16);
            const unsigned p1b = dtf_f32_to_bf16(v1[2]) | ((unsigned)dtf_f32_to_bf16(v1[3]) << 16);
            const auto sa = __builtin_amdgcn_permlane16_swap(p0a, p1a, false, false);
            const auto sb = __builtin_amdgcn_permlane16_swap(p0b, p1b, false, false);
            // even rows of lanes (fq = 0, 2): columns 4 fq .. 4 fq + 7 of the first tile;
            // odd rows (fq = 1, 3): columns 4 (fq - 1) .. 4 fq + 3 of the second
            const int col = cb + ((e_fq & 1) ? 16 + 4 * (e_fq - 1) : 4 * e_fq);
            u32x4 o = {sa[0], sb[0], sa[1], sb[1]};
            const long off = ((long)row * ldc + col) * 2;
            *reinterpret_cast<u32x4*>(c_plane + off) = o;
            for (int q = 0; q < nmirrors; ++q)
              *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(mirrors[q]) + off) = o;
            continue;
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = cn0 + nh * 128 + wc * 32 + j * 16 + 4 * e_fq;
          if (col >= N) continue;
          f32x4 v = acc[mh][nh][i][j] * alpha;
          const long off = ((long)row * ldc + col) * (OUT_F32 ? 4 : 2);
          if (OUT_F32) {
            float* dst = reinterpret_cast<float*>(c_plane + off);
            if (col + 3 < N) {
              if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(dst) * beta;
              *reinterpret_cast<f32x4*>(dst) = v;
              for (int q = 0; q < nmirrors; ++q)
                *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(mirrors[q]) + off) = v;
            } else {
              for (int r = 0; r < 4 && col + r < N; ++r) {
                dst[r] = beta != 0.f ? v[r] + beta * dst[r] : v[r];
                for (int q = 0; q < nmirrors; ++q)
                  reinterpret_cast<float*>(reinterpret_cast<char*>(mirrors[q]) + off)[r] = v[r];
              }
            }
          } else {
            unsigned short* dst = reinterpret_cast<unsigned short*>(c_plane + off);
            if (beta != 0.f) {
              for (int r = 0; r < 4 && col + r < N; ++r) v[r] += beta * dtf_bf16_to_f32(dst[r]);
            }
            if (col + 3 < N) {
              bf16x4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (short)dtf_f32_to_bf16(v[r]);
              *reinterpret_cast<bf16x4*>(dst) = o;
              for (int q = 0; q < nmirrors; ++q)
                *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(mirrors[q]) + off) = o;
            } else {
              for (int r = 0; r < 4 && col + r < N; ++r) {
                dst[r] = dtf_f32_to_bf16(v[r]);
                for (int q = 0; q < nmirrors; ++q)
                  reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(mirrors[q]) + off)[r] =
                      dst[r];
              }
            }
          }
        }
      }
    }
  if (!more) break;
  }   // tile loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no LDS-DMA may outlive the workgroup
  if constexpr ((dbg & 8) != 0) {   // overwrite the head of the first tile's rows 0-7 with the sums
    __syncthreads();
    if (lane == 0 && first_m0 + 8 <= M && first_n0 + 64 <= N) {
      unsigned* dst = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(C) +
                                                  ((long)(first_m0 + wave) * ldc + first_n0) *
                                                      (OUT_F32 ? 4 : 2));
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 6; ++q) dst[p * 6 + q] = seg[p][q];
    }
  }
}

// ---- four-wave kernel: one wave per SIMD, 128x128 outputs per wave ------------------------------
// (variant 3; NOT the default. Measured at 16384^3 on MI355X: 1245 TF against 1456-1513 TF for
// the 8-wave ping-pong kernel above; with the K-loop's LDS-DMA removed 1785 vs 1811 TF, with every
// load an L2 hit 1491 vs 1556 TF. With one wave per SIMD nothing covers a wave's barrier and
// DMA-latency stalls, and two 64 KiB K-tile buffers allow only one K-tile of prefetch distance.
// Kept as a correct, tested alternative schedule.)
// Same 256x256x64 block tile and LDS image as above, but 256 threads = 4 waves as 2(M) x 2(N),
// each owning a 128x128 quarter (8x8 MFMA tiles, all 256 accumulator registers). A wave then
// reads (128 + 128) rows of fragments per k-step instead of the (128 + 64) x 2 of two 128x64
// waves: a third less LDS read traffic per flop.
//  * Two K-tile buffers of 64 KiB. A K-tile is two k-steps (chunks 0-3 / 4-7 of each 128-byte
//    row); fragments are double-buffered in registers one k-step ahead: while the 64 MFMAs of
//    (t, 0) run the wave reads the fragments of (t, 1), while those of (t, 1) run it reads
//    (t+1, 0) and issues the 16 LDS-DMA instructions of K-tile t+2 into the buffer of tile t.
//  * One barrier per K-tile, between the two k-steps: behind it every wave has finished reading
//    buffer t (so it may be refilled) and every wave's DMA of tile t+1 has landed (so it may be
//    read). The DMA stream therefore runs one K-tile ahead.
//  * A k-step is 64 slots of [one MFMA + at most one memory instruction]: a wave issues in
//    order, so whatever sits between two MFMAs must fit in the 16-cycle shadow of the first or
//    the matrix pipe idles. The MFMAs are inline asm with the accumulator tied to an AGPR quad:
//    with all 256 AGPRs holding accumulators the register allocator otherwise parks some in
//    VGPRs and shuttles them through v_accvgpr moves around every MFMA. The hardware interlocks
//    back-to-back MFMAs on one accumulator; operand registers come from ds_reads the compiler
//    still tracks.
constexpr int W4_THREADS = 256;

// DBG (timing experiments, -DDTF_GEMM_EXPERIMENTS only; results are wrong when set), bits:
// 1 = issue no LDS-DMA in the K-loop, 2 = always stage K-tile 0 (every load an L2 hit),
// 4 = no barrier in the K-loop, 8 = no wait in front of it either
template <bool OUT_F32, int DBG = 0>
__global__ __launch_bounds__(W4_THREADS) void gemm_bf16_nt_w4_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int ntiles = tiles_m * tiles_n;
  const int id = dtf_xcd_remap(blockIdx.x, ntiles);
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int grp = id / per_group;
  const int first_m = grp * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (id % per_group) % gsize;
  const int tn = (id % per_group) / gsize;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = K / BK;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA sources: this wave fills pieces 8*wave .. 8*wave+7 (8 rows of 128 B each) of the A tile
  // and of the Bt tile; lane i lands at row i>>3, position i&7 of its piece and fetches chunk
  // (i&7) ^ swizzle(row) of that row (see chunk_off)
  const char* src_a[8];
  const char* src_b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int trow = (wave * 8 + j) * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((trow >> 1) & 7);
    int ra = m0 + trow, rb = n0 + trow;
    ra = ra < M - 1 ? ra : M - 1;
    rb = rb < N - 1 ? rb : N - 1;
    src_a[j] = A + ((long)ra * lda) * 2 + kc * 16;
    src_b[j] = Bt + ((long)rb * ldb) * 2 + kc * 16;
  }

  const int frow = lane & 15, fq = lane >> 4;
  const int lane_off0 = chunk_off(frow, fq), lane_off1 = chunk_off(frow, 4 + fq);
  const int a_base = wr * 16384;                    // wave's 8 row blocks of 2 KiB in the A tile
  const int b_base = TILE_BYTES + wc * 16384;       // ... in the Bt tile

  i32x4 fa[2][8], fb[2][8];
// one k-step. CUR / NXT: fragment register sets. RD: byte address of (buffer, chunk half) the
// NXT fragments come from. DMA: issue K-tile KT_DMA into buffer offset WR_BUF during the step.
#define DTF_W4_STEP(CUR, NXT, RD_BUF, RD_LANE_OFF, DMA, KT_DMA, WR_BUF)                          \
  {                                                                                              \
    const char* rda_ = smem + (RD_BUF) + a_base + (RD_LANE_OFF);                                 \
    const char* rdb_ = smem + (RD_BUF) + b_base + (RD_LANE_OFF);                                 \
    const long koff_ = (DBG & 2) ? 0 : (long)((KT_DMA) < nkt ? (KT_DMA) : nkt - 1) * 128;         \
    char* wr_ = smem + (WR_BUF) + wave * 8192;                                                   \
    _Pragma("unroll") for (int n_ = 0; n_ < 64; ++n_) {                                          \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                      \
                   : "+a"(acc[n_ >> 3][n_ & 7])                                                  \
                   : "v"(fb[CUR][n_ & 7]), "v"(fa[CUR][n_ >> 3]));                               \
      if (n_ < 16) {                                                                             \
        if (n_ & 1)                                                                              \
          fb[NXT][n_ >> 1] = *reinterpret_cast<const i32x4*>(rdb_ + (n_ >> 1) * 2048);           \
        else                                                                                     \
          fa[NXT][n_ >> 1] = *reinterpret_cast<const i32x4*>(rda_ + (n_ >> 1) * 2048);           \
      } else if ((DMA) && !(DBG & 1) && n_ < 48 && ((n_ - 16) & 1) == 0) {                         \
        const int d_ = (n_ - 16) >> 1, j_ = d_ >> 1;                                             \
        if (d_ & 1)                                                                              \
          __builtin_amdgcn_global_load_lds((global_void*)(src_b[j_] + koff_),                    \
                                           (lds_void*)(wr_ + TILE_BYTES + j_ * 1024), 16, 0, 0); \
        else                                                                                     \
          __builtin_amdgcn_global_load_lds((global_void*)(src_a[j_] + koff_),                    \
                                           (lds_void*)(wr_ + j_ * 1024), 16, 0, 0);              \
      }                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  }

  // prologue: K-tiles 0 and 1 in flight, tile 0 landed and visible, fragments of (0, 0) requested
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long koff = (long)(t < nkt ? t : nkt - 1) * 128;
    char* wr = smem + t * BUF_BYTES + wave * 8192;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __builtin_amdgcn_global_load_lds((global_void*)(src_a[j] + koff),
                                       (lds_void*)(wr + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((global_void*)(src_b[j] + koff),
                                       (lds_void*)(wr + TILE_BYTES + j * 1024), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[0][i] = *reinterpret_cast<const i32x4*>(smem + a_base + lane_off0 + i * 2048);
    fb[0][i] = *reinterpret_cast<const i32x4*>(smem + b_base + lane_off0 + i * 2048);
  }

  for (int t = 0; t < nkt; ++t) {
    const int cur_buf = (t & 1) * BUF_BYTES, nxt_buf = BUF_BYTES - cur_buf;
    // (t, 0): fragments of (t, 1) from this tile's buffer
    DTF_W4_STEP(0, 1, cur_buf, lane_off1, false, 0, 0)
    // own reads of buffer t done (lgkmcnt 0), own DMA of tile t+1 landed (vmcnt 0); as the
    // builtin (imm 0x0070 = vmcnt 0, expcnt 7, lgkmcnt 0) so the compiler's wait-count
    // bookkeeping sees it and adds no second wait in front of the MFMAs
    if (!(DBG & 8)) __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // (t, 1): fragments of (t+1, 0) from the other buffer; K-tile t+2 into this tile's buffer
    DTF_W4_STEP(1, 0, nxt_buf, lane_off0, true, t + 2, cur_buf)
  }
#undef DTF_W4_STEP
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing may outlive the workgroup
  // the last MFMAs were issued from inline asm: cover their result latency before the
  // epilogue reads the accumulators
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = m0 + wr * 128 + i * 16 + frow;
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = n0 + wc * 128 + j * 16 + 4 * fq;
      if (col >= N) continue;
      f32x4 v = acc[i][j] * alpha;
      if (OUT_F32) {
        float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
        if (col + 3 < N) {
          if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(dst) * beta;
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          for (int r = 0; r < 4 && col + r < N; ++r)
            dst[r] = beta != 0.f ? v[r] + beta * dst[r] : v[r];
        }
      } else {
        unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
        if (beta != 0.f) {
          for (int r = 0; r < 4 && col + r < N; ++r) v[r] += beta * dtf_bf16_to_f32(dst[r]);
        }
        if (col + 3 < N) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (short)dtf_f32_to_bf16(v[r]);
          *reinterpret_cast<bf16x4*>(dst) = o;
        } else {
          for (int r = 0; r < 4 && col + r < N; ++r) dst[r] = dtf_f32_to_bf16(v[r]);
        }
      }
    }
  }
}

// ---- four-wave kernel on a five-stage ring of 32-k stages (variant 7) ---------------------------
// The four-wave kernel above starves because two 64 KiB K-tile buffers let its DMA run less than
// one K-tile ahead. Here the 160 KiB of LDS are a RING OF FIVE STAGES of 32 k each (A 256 rows x
// 64 B + Bt 256 rows x 64 B = 32 KiB): a stage is exactly one k-step of the MFMA, its slot is
// dead as soon as its fragments are in registers (one k-step before they are multiplied), so the
// DMA of stages s + 4 and s + 5 is issued during steps s and s + 1 (s even; A pieces in the
// first, Bt pieces in the second) and read during steps s + 3 and s + 4: three k-steps = 1.5 K-tiles ahead, what the eight-wave kernel has. One barrier per
// k-step (64 MFMAs per wave) behind a counted s_waitcnt vmcnt(16 / 17): two stages stay in flight
// across it.
//  * LDS image of a stage: row r at r * 64, its four 16-byte chunks XOR-permuted: chunk c sits at
//    position c ^ ((-(r >> 2)) & 3). A fragment read (lane l: row l & 15, chunk l >> 4) then puts
//    the 16 lanes of every ds_read_b128 lane group on 16 distinct 16-byte slots of the 256-byte
//    bank row. The DMA writes lane-linearly (1 KiB = 16 rows per instruction, lane i -> row i / 4,
//    position i % 4) and applies the permutation to its SOURCE address; four consecutive lanes
//    fetch one row's 64 bytes.
//  * Fragments are double-buffered in registers one k-step ahead (2 x 16 x 4 VGPRs), the 8 x 8
//    accumulator tiles fill the 256 AGPRs; a k-step is 64 slots of [MFMA + at most one memory
//    instruction]: a fragment read every fourth slot, a DMA request every eighth.
//  * A stage holds HALF of every 128-byte line of its rows, so each line is requested twice;
//    the two stages of a K-tile are therefore requested piece by piece together (the same wave
//    asks for the second half 128 cycles after the first and finds the line on its way):
//    requested a k-step apart the same kernel ran at 1065 TF (16384^3), bursts of sixteen
//    requests in every other step at 1262, spread as now at 1385-1453.
// NOT the default. Measured at 16384^3 / 8192^3 / 4096^3 beside the eight-wave kernel on one box:
// 1385 / 1436 / 1315 against 1502 / 1526 / 1446 TF (92-94 %); with the K loop's DMA removed 1732-
// 1847 TF (the eight-wave kernel: 1811). It draws 1373 W at a sampled 1.97 GHz where the eight-
// wave kernel sits at the 1400 W cap at 1.73 GHz (`scripts/gemm_power_probe.py`): it is not
// power-bound, its matrix pipe waits -- neither for memory (dropping the counted waits changes
// nothing) nor much at the barrier (1408 without it), but at the issue of the DMA requests
// themselves: four lanes per 64-byte half line are sixteen lines per request where the 128-byte
// rows of the other kernels are eight, and with one wave per SIMD the wave that waits for the
// address unit is the wave that feeds the matrix pipe. Full lines need 64 KiB K-tile slots, of
// which 160 KiB hold two and a half; the vendor kernel gets its depth from registers instead.
constexpr int RS_STAGE_BYTES = 32768, RS_STAGES = 5;

// DBG (timing experiments, -DDTF_GEMM_EXPERIMENTS only; results are wrong when set): 1 = no
// LDS-DMA in the K loop, 4 = no barrier
template <bool OUT_F32, int DBG = 0>
__global__ __launch_bounds__(W4_THREADS) void gemm_bf16_nt_w4ring_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  pp_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn, false);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nks = K / 32;                       // stages (k-steps) of this product

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // DMA: this wave fills pieces 4 * wave .. 4 * wave + 3 (16 rows each) of the stage's A half and
  // of its Bt half; the per-lane byte offsets are loop-invariant, the stage's 64 bytes of k are
  // the instruction's scalar offset
  unsigned voff_a[4], voff_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int trow = (wave * 4 + j) * 16 + (lane >> 2);
    const int kc = (lane & 3) ^ ((-(trow >> 2)) & 3);
    int ra = m0 + trow, rb = n0 + trow;
    ra = ra < M - 1 ? ra : M - 1;
    rb = rb < N - 1 ? rb : N - 1;
    voff_a[j] = (unsigned)(ra - m0) * (unsigned)(lda * 2) + 16u * kc;
    voff_b[j] = (unsigned)(rb - n0) * (unsigned)(ldb * 2) + 16u * kc;
  }
  const __amdgpu_buffer_rsrc_t a_rsrc = tile_rsrc(A + (long)m0 * lda * 2);
  const __amdgpu_buffer_rsrc_t b_rsrc = tile_rsrc(Bt + (long)n0 * ldb * 2);

  const int frow = lane & 15, fq = lane >> 4;
  // lane's byte offset inside a 16-row block of a stage half, plus the wave's quarter
  const int lane_off = frow * 64 + ((fq ^ ((-(frow >> 2)) & 3)) << 4);
  const int a_off = lane_off + wr * 8192;                // A rows wr * 128 ...
  const int b_off = lane_off + 16384 + wc * 8192;        // Bt rows wc * 128 ...

  i32x4 fa[2][8], fb[2][8];
// one DMA instruction of stage ST (clamped) into ring slot SLOT: D = 0..3 A pieces, 4..7 Bt pieces
#define DTF_RS_DMA(ST, SLOT_BYTES, D)                                                            \
  {                                                                                              \
    const int st_ = (ST) < nks ? (ST) : nks - 1;                                                 \
    char* dst_ = smem + (SLOT_BYTES) + ((D) >= 4 ? 16384 : 0) + (wave * 4 + ((D) & 3)) * 1024;   \
    if ((D) >= 4)                                                                                \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_void*)dst_, 16, voff_b[(D) & 3],     \
                                               st_ * 64, 0, 0);                                  \
    else                                                                                         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)dst_, 16, voff_a[(D) & 3],     \
                                               st_ * 64, 0, 0);                                  \
  }
// one k-step: 64 MFMAs on fragment set CUR; the fragments of the next stage (ring slot at byte
// RD) go into set NXT; pieces P0 .. P0 + 3 (0-3 A, 4-7 Bt) of stages ST_DMA and ST_DMA + 1 are
// requested into the ring slots at bytes WR and WR2, piece by piece for both stages (the two
// stages of a K-tile are the two halves of the same 128-byte lines: requested back to back by
// the same wave, the second request finds the line already on its way)
#define DTF_RS_STEP(CUR, NXT, RD, P0, ST_DMA, WR, WR2)                                           \
  {                                                                                              \
    const char* rda_ = smem + (RD) + a_off;                                                      \
    const char* rdb_ = smem + (RD) + b_off;                                                      \
    _Pragma("unroll") for (int n_ = 0; n_ < 64; ++n_) {                                          \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                      \
                   : "+a"(acc[n_ >> 3][n_ & 7])                                                  \
                   : "v"(fb[CUR][n_ & 7]), "v"(fa[CUR][n_ >> 3]));                               \
      /* a fragment read every fourth slot (the Bt fragments, all needed by the next step's     \
         first eight MFMAs, first), a DMA request every eighth slot between them: pieces P0 ..   \
         P0 + 3 of both stages, the two halves of a line eight slots apart */                    \
      if ((n_ & 3) == 0) {                                                                       \
        const int f_ = (n_ >> 2) & 7;                                                            \
        if (n_ < 32)                                                                             \
          fb[NXT][f_] = *reinterpret_cast<const i32x4*>(rdb_ + f_ * 1024);                       \
        else                                                                                     \
          fa[NXT][f_] = *reinterpret_cast<const i32x4*>(rda_ + f_ * 1024);                       \
      } else if (!(DBG & 1) && (n_ & 7) == 2) {                                                  \
        const int piece_ = (P0) + (n_ >> 4);                                                     \
        if (n_ & 8) DTF_RS_DMA((ST_DMA) + 1, WR2, piece_)                                        \
        else DTF_RS_DMA(ST_DMA, WR, piece_)                                                      \
      }                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  }

  // prologue: stages 0..3 in flight (pairwise interleaved), stage 0 landed and visible, its
  // fragments requested
#pragma unroll
  for (int st = 0; st < 4; st += 2)
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      DTF_RS_DMA(st, st * RS_STAGE_BYTES, d)
      DTF_RS_DMA(st + 1, (st + 1) * RS_STAGE_BYTES, d)
    }
  asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    fa[0][i] = *reinterpret_cast<const i32x4*>(smem + a_off + i * 1024);
    fb[0][i] = *reinterpret_cast<const i32x4*>(smem + b_off + i * 1024);
  }

  // ring positions (bytes): rd = slot of stage s + 1, wr = slot of stage s + 4 (= that of stage
  // s - 1), wr2 = slot of stage s + 5 (= that of stage s, whose fragments are in registers)
  int rd = RS_STAGE_BYTES, wr_slot = 4 * RS_STAGE_BYTES, wr2_slot = 0;
  auto next = [](int v) { return v == 4 * RS_STAGE_BYTES ? 0 : v + RS_STAGE_BYTES; };
  auto advance = [&]() {
    rd = next(rd);
    wr_slot = next(wr_slot);
    wr2_slot = next(wr2_slot);
  };
  for (int s = 0; s < nks; s += 2) {
    // stage s + 1 has landed for this wave (stages s + 2, s + 3 may still be in flight) ...
    if (!(DBG & 1)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();      // ... and for every wave; all are done reading slot wr
    __builtin_amdgcn_sched_barrier(0);
    DTF_RS_STEP(0, 1, rd, 0, s + 4, wr_slot, wr2_slot)
    // stage s + 2 has landed: of the pair (s + 2, s + 3) at most the last piece of s + 3 and the
    // eight requests of this step are still in flight  (K % 64 == 0: an even stage count)
    if (!(DBG & 1)) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rd = next(rd);
    DTF_RS_STEP(1, 0, rd, 4, s + 4, wr_slot, wr2_slot)
    advance();
    wr_slot = next(wr_slot);
    wr2_slot = next(wr2_slot);
  }
#undef DTF_RS_STEP
#undef DTF_RS_DMA
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // nothing may outlive the workgroup
  // the last MFMAs were issued from inline asm: cover their result latency before the
  // epilogue reads the accumulators
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = m0 + wr * 128 + i * 16 + frow;
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = n0 + wc * 128 + j * 16 + 4 * fq;
      if (col >= N) continue;
      f32x4 v = acc[i][j] * alpha;
      if (OUT_F32) {
        float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
        if (col + 3 < N) {
          if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(dst) * beta;
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          for (int r = 0; r < 4 && col + r < N; ++r)
            dst[r] = beta != 0.f ? v[r] + beta * dst[r] : v[r];
        }
      } else {
        unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
        if (beta != 0.f) {
          for (int r = 0; r < 4 && col + r < N; ++r) v[r] += beta * dtf_bf16_to_f32(dst[r]);
        }
        if (col + 3 < N) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (short)dtf_f32_to_bf16(v[r]);
          *reinterpret_cast<bf16x4*>(dst) = o;
        } else {
          for (int r = 0; r < 4 && col + r < N; ++r) dst[r] = dtf_f32_to_bf16(v[r]);
        }
      }
    }
  }
}

// ---- four-wave kernel with a register-staged load ring (variant 8) ------------------------------
// Same block tile, LDS image (128-byte rows, full lines) and 2 x 2 waves of 128 x 128 outputs as the
// four-wave kernels above, but the prefetch depth comes from REGISTERS: a K-tile's 64 KiB travel
// global -> VGPR (16 buffer loads of 16 bytes per lane) -> LDS (16 ds_write_b128), two register
// sets alternate, and the loads of tile t + 3 are issued while tile t is multiplied -- they are
// written to LDS three to four k-steps later. Per K-tile and wave: 128 MFMAs, 32 fragment reads,
// 16 loads, 16 LDS writes, ONE barrier.
//  step (t, 0): MFMAs of (t, 0) | reads the fragments of (t, 1) from buffer t | writes the register
//               set of tile t + 1 into buffer t + 1 (whose last readers passed the previous
//               barrier) and, behind each write, requests that register's part of tile t + 3
//  barrier      (tile t + 1 visible; every wave done with buffer t's k-step-1 fragments next step)
//  step (t, 1): MFMAs of (t, 1) | reads the fragments of (t + 1, 0) from buffer t + 1
// NOT the default. Measured at 16384^3 / 8192^3 / 4096^3 beside the eight-wave kernel on one box:
// 1409 / 1416 / 1334 against 1550 / 1579 / 1472 TF. With the LDS writes removed (loads kept) it
// runs at 1856 / 1832 / 1636, with the loads removed (writes kept) at 1675 / 1601 / 1497: the
// loads into registers are free, the sixteen ds_write_b128 per K-tile and wave are not, and
// neither a longer prefetch distance (two against three k-steps) nor relaxed fragment reads
// changed the result. It sits at 1394 W and a sampled 1.88 GHz: register staging pays for the
// extra register-file and LDS-write traffic in energy as well -- the vendor kernel's LDS
// instruction count (reads only, `profiles/gemm_bf16_16384_pmc_r3.txt`) says it fills LDS by DMA
// too.
// Fragments: the eight Bt fragments of a k-step stay in registers, double-buffered across steps
// (2 x 32 VGPRs); the eight A fragments are single-buffered (32 VGPRs): m-tile i's register is
// refilled for the next step as soon as its eight MFMAs are issued -- 96 registers where the
// kernels above hold 128, which is what leaves room for the two load sets (128) next to the 256
// accumulator registers.
// Contract: M and N multiples of 256 (no row clamping: the loads' lane offsets are two registers
// per operand), K a multiple of 128 (two K-tiles per loop trip).
template <bool OUT_F32, int DBG = 0>
__global__ __launch_bounds__(W4_THREADS) void gemm_bf16_nt_w4reg_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  pp_tile_of_block(blockIdx.x, tiles_m, tiles_n, tm, tn, false);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nkt = K / BK;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // loads: this wave moves pieces 8 * wave + j (j = 0..7; a piece = 8 rows of 128 bytes) of the A
  // tile and of the Bt tile. Lane i of a piece handles row i / 8, LDS position i % 8 and fetches
  // chunk (i % 8) ^ swizzle(row); the swizzle of piece p's rows is (4 p + i / 16) & 7, which
  // depends on j only through its parity -- two lane offsets per operand, the rest is scalar
  unsigned voff_a[2], voff_b[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int kc = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);
    voff_a[par] = (unsigned)(wave * 64 + (lane >> 3)) * (unsigned)(lda * 2) + 16u * kc;
    voff_b[par] = (unsigned)(wave * 64 + (lane >> 3)) * (unsigned)(ldb * 2) + 16u * kc;
  }
  const int piece_a = (int)(lda * 16), piece_b = (int)(ldb * 16);   // bytes from piece to piece
  const __amdgpu_buffer_rsrc_t a_rsrc = tile_rsrc(A + (long)m0 * lda * 2);
  const __amdgpu_buffer_rsrc_t b_rsrc = tile_rsrc(Bt + (long)n0 * ldb * 2);

  const int frow = lane & 15, fq = lane >> 4;
  const int lane_off0 = chunk_off(frow, fq), lane_off1 = chunk_off(frow, 4 + fq);
  const int a_base = wr * 16384;                    // wave's 8 row blocks of 2 KiB in the A tile
  const int b_base = TILE_BYTES + wc * 16384;       // ... in the Bt tile
  char* const wr_lane = smem + wave * 8192 + lane * 16;   // this lane's slot in the wave's pieces

  i32x4 g[2][16];          // two load sets: [set][0..7 A pieces, 8..15 Bt pieces]
  i32x4 fa[8], fb[2][8];   // A fragments (refilled for the next step as m-tiles finish), Bt
                           // fragments of this step / the next

// load Q (0..15) of K-tile KT into set GS
#define DTF_WR_LOAD(GS, Q, KT)                                                                   \
  {                                                                                              \
    const int kt_ = (KT) < nkt ? (KT) : nkt - 1;                                                 \
    if ((Q) < 8)                                                                                 \
      g[GS][Q] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, voff_a[(Q) & 1],                  \
                                                       kt_ * 128 + ((Q) & 7) * piece_a, 0);      \
    else                                                                                         \
      g[GS][Q] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, voff_b[(Q) & 1],                  \
                                                       kt_ * 128 + ((Q) & 7) * piece_b, 0);      \
  }
// LDS write Q of set GS into K-tile buffer BUFI
#define DTF_WR_STORE(GS, Q, BUFI)                                                                \
  *reinterpret_cast<i32x4*>(wr_lane + (BUFI) * BUF_BYTES + ((Q) >= 8 ? TILE_BYTES : 0) +         \
                            ((Q) & 7) * 1024) = g[GS][Q];
// one k-step of 64 MFMAs, m-tile by m-tile. FS: fragment set of this step's Bt fragments (the
// next step's go into FS ^ 1). The next step's fragments come from buffer RB at lane offset RL;
// this step's remaining A fragments from buffer CB at lane offset CL. MEM(q): the step's q-th
// (0..15) load or LDS write.
#define DTF_WR_STEP(FS, CB, CL, RB, RL, MEM)                                                     \
  {                                                                                              \
    const char* cur_a_ = smem + (CB) * BUF_BYTES + a_base + (CL);                                \
    const char* nxt_a_ = smem + (RB) * BUF_BYTES + a_base + (RL);                                \
    const char* nxt_b_ = smem + (RB) * BUF_BYTES + b_base + (RL);                                \
    _Pragma("unroll") for (int n_ = 0; n_ < 64; ++n_) {                                          \
      const int i_ = n_ >> 3, j_ = n_ & 7;                                                       \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                      \
                   : "+a"(acc[i_][j_])                                                           \
                   : "v"(fb[FS][j_]), "v"(fa[i_]));                                              \
      if (j_ == 0) {                                                                             \
        /* m-tile 7's fragment of THIS step (its register was busy until the last step ended);  \
           then, behind every finished m-tile, its fragment of the next step */                  \
        if (i_ == 0) fa[7] = *reinterpret_cast<const i32x4*>(cur_a_ + 7 * 2048);                 \
        else fa[i_ - 1] = *reinterpret_cast<const i32x4*>(nxt_a_ + (i_ - 1) * 2048);             \
      } else if (j_ == 1) {                                                                      \
        fb[(FS) ^ 1][i_] = *reinterpret_cast<const i32x4*>(nxt_b_ + i_ * 2048);                  \
      } else if (j_ == 2) {                                                                      \
        MEM(2 * i_, 0)                                                                           \
      } else if (j_ == 3) {                                                                      \
        MEM(2 * i_, 1)                                                                           \
      } else if (j_ == 5) {                                                                      \
        MEM(2 * i_ + 1, 0)                                                                       \
      } else if (j_ == 6) {                                                                      \
        MEM(2 * i_ + 1, 1)                                                                       \
      }                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  }

  // prologue: tiles 0 and 1 requested, tile 0 written to buffer 0 and visible, tile 2 requested
  // into the set tile 0 has left, the fragments of (0, 0) read
#pragma unroll
  for (int q = 0; q < 16; ++q) DTF_WR_LOAD(0, q, 0)
#pragma unroll
  for (int q = 0; q < 16; ++q) DTF_WR_LOAD(1, q, 1)
#pragma unroll
  for (int q = 0; q < 16; ++q) DTF_WR_STORE(0, q, 0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int q = 0; q < 16; ++q) DTF_WR_LOAD(0, q, 2)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (j < 7) fa[j] = *reinterpret_cast<const i32x4*>(smem + a_base + lane_off0 + j * 2048);
    fb[0][j] = *reinterpret_cast<const i32x4*>(smem + b_base + lane_off0 + j * 2048);
  }

  for (int t = 0; t < nkt; t += 2) {
    // tile t (buffer 0): (t, 0) writes tile t + 1 (set 1) into buffer 1
    // ... and, behind each write, requests the same register's part of tile t + 3
#define DTF_MEM_A(Q, PH)                                                                         \
  if ((PH) == 0) { if (!(DBG & 2)) { DTF_WR_STORE(1, Q, 1) } }                                   \
  else if (!(DBG & 1)) { DTF_WR_LOAD(1, Q, t + 3) }
    DTF_WR_STEP(0, 0, lane_off0, 0, lane_off1, DTF_MEM_A)
#undef DTF_MEM_A
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#define DTF_MEM_B(Q, PH)
    DTF_WR_STEP(1, 0, lane_off1, 1, lane_off0, DTF_MEM_B)
    // tile t + 1 (buffer 1): (t + 1, 0) writes tile t + 2 (set 0) into buffer 0
#define DTF_MEM_C(Q, PH)                                                                         \
  if ((PH) == 0) { if (!(DBG & 2)) { DTF_WR_STORE(0, Q, 0) } }                                   \
  else if (!(DBG & 1)) { DTF_WR_LOAD(0, Q, t + 4) }
    DTF_WR_STEP(0, 1, lane_off0, 1, lane_off1, DTF_MEM_C)
#undef DTF_MEM_C
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(DBG & 4)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    DTF_WR_STEP(1, 1, lane_off1, 0, lane_off0, DTF_MEM_B)
#undef DTF_MEM_B
  }
#undef DTF_WR_STEP
#undef DTF_WR_STORE
#undef DTF_WR_LOAD
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the inline-asm MFMAs' result latency

#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = m0 + wr * 128 + i * 16 + frow;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int col = n0 + wc * 128 + j * 16 + 4 * fq;
      f32x4 v = acc[i][j] * alpha;
      if (OUT_F32) {
        float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
        if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(dst) * beta;
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
        unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
        if (beta != 0.f) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += beta * dtf_bf16_to_f32(dst[r]);
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (short)dtf_f32_to_bf16(v[r]);
        *reinterpret_cast<bf16x4*>(dst) = o;
      }
    }
  }
}

// ---- generic, fully predicated bf16 GEMM (any shape, NN / NT / TN / TT via strides) -------------
// C[m,n] = alpha * sum_k A(m,k) * B(k,n), A(m,k) = A[m*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs]
constexpr int GB = 64, GK = 32;
#include "gemm_bf16_w4p2.h"

template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_bf16_generic_kernel(
    const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
    void* __restrict__ C, int M, int N, int K, long a_rs, long a_cs, long b_rs, long b_cs,
    long ldc, float alpha, float beta) {
  // As[m][k], Bs[n][k] with 80-byte rows (64 B data + 16 B pad: conflict-free b128 reads)
  __shared__ __attribute__((aligned(16))) unsigned short As[GB][GK + 8];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[GB][GK + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;       // 2x2 waves, 32x32 each
  const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  for (int k0 = 0; k0 < K; k0 += GK) {
    // 64 x 32 elements per operand, 256 threads -> 8 elements each
    for (int e = tid; e < GB * GK; e += 256) {
      const int r = e / GK, k = e % GK;
      const int gm = m0 + r, gn = n0 + r, gk = k0 + k;
      As[r][k] = (gm < M && gk < K) ? A[gm * a_rs + gk * a_cs] : (unsigned short)0;
      Bs[r][k] = (gn < N && gk < K) ? B[gk * b_rs + gn * b_cs] : (unsigned short)0;
    }
    __syncthreads();
    bf16x8 af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(&As[wr * 32 + i * 16 + frow][fq * 8]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = *reinterpret_cast<const bf16x8*>(&Bs[wc * 32 + j * 16 + frow][fq * 8]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = m0 + wr * 32 + i * 16 + frow;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = n0 + wc * 32 + j * 16 + 4 * fq + r;
        if (row < M && col < N) {
          float v = acc[i][j][r] * alpha;
          if (OUT_F32) {
            float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
            if (beta != 0.f) v += beta * *dst;
            *dst = v;
          } else {
            unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
            if (beta != 0.f) v += beta * dtf_bf16_to_f32(*dst);
            *dst = dtf_f32_to_bf16(v);
          }
        }
      }
    }
}

// ---- generic, fully predicated fp8 (e4m3) GEMM: v_mfma_f32_16x16x32_fp8_fp8, any shape/stride --
template <bool OUT_F32>
__global__ __launch_bounds__(256) void gemm_fp8_generic_kernel(
    const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
    void* __restrict__ C, int M, int N, int K, long a_rs, long a_cs, long b_rs, long b_cs,
    long ldc, float alpha, float beta) {
  // As[m][k], Bs[n][k] with 40-byte rows (32 B data + 8 B pad, keeps the 8-byte reads aligned)
  __shared__ __attribute__((aligned(16))) unsigned char As[GB][GK + 8];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[GB][GK + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * GB, n0 = blockIdx.x * GB;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  for (int k0 = 0; k0 < K; k0 += GK) {
    for (int e = tid; e < GB * GK; e += 256) {
      const int r = e / GK, k = e % GK;
      const int gm = m0 + r, gn = n0 + r, gk = k0 + k;
      As[r][k] = (gm < M && gk < K) ? A[gm * a_rs + gk * a_cs] : (unsigned char)0;
      Bs[r][k] = (gn < N && gk < K) ? B[gk * b_rs + gn * b_cs] : (unsigned char)0;
    }
    __syncthreads();
    long af[2], bf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      af[i] = *reinterpret_cast<const long*>(&As[wr * 32 + i * 16 + frow][fq * 8]);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      bf[j] = *reinterpret_cast<const long*>(&Bs[wc * 32 + j * 16 + frow][fq * 8]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bf[j], af[i], acc[i][j], 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = m0 + wr * 32 + i * 16 + frow;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = n0 + wc * 32 + j * 16 + 4 * fq + r;
        if (row < M && col < N) {
          float v = acc[i][j][r] * alpha;
          if (OUT_F32) {
            float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
            if (beta != 0.f) v += beta * *dst;
            *dst = v;
          } else {
            unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
            if (beta != 0.f) v += beta * dtf_bf16_to_f32(*dst);
            *dst = dtf_f32_to_bf16(v);
          }
        }
      }
    }
}

}  // namespace

// ---- 128x128 tile kernels for grids that leave CUs idle at 256x256 ----------------------------
// Four multiplying waves as 2(M) x 2(N), 64x64 outputs per wave, K-tile 64, the same LDS image
// and MFMA as above. A product of 2048^2 outputs is 64 tiles of 256^2 (a quarter of the chip)
// but 256 tiles of 128^2. Two forms:
//  * gemm_bf16_nt_t128_kernel: 256 threads, two 32 KiB K-tile buffers, so two workgroups share a
//    CU and cover each other's barrier and request waits: grids of more than one tile per CU.
//  * gemm_bf16_nt_t128ws_kernel: 512 threads, waves 4-7 do nothing but issue the LDS-DMA
//    requests (two K-tiles ahead, three buffers, one barrier per K-tile): grids with a CU for
//    every tile, and the work items of a K split. A wave that issues a request waits 60-180
//    clocks for the address unit before its next instruction, eight times per K-tile, and with
//    one four-wave workgroup on the CU its SIMD runs nothing else meanwhile: a K-tile took the
//    four-wave form 0.6 us whatever the grid, with two buffers or a ring of three or four
//    (equal or slower), and slower still with the requests spread between the MFMAs of a
//    two-buffer loop (0.78 us: the newest request's latency is then exposed). With the requests
//    on waves of their own it takes 0.4 us. GPU time by kernel trace, four waves / with request
//    waves / torch.matmul: 1024^3 12.7 / 10.0 / 8.9 us, 1024 x 1024 x 4096 40.1 / 26.6 / 19.2 (the
//    default plan splits K there: 22.0), 1536^3 19.5 / 13.8 / 15.7, 2048^3 24.0 / 19.5 / 22.2,
//    2048 x 2048 x 4096 42.8 / 32.1 / 36.6, 1280^3 14.9 / 11.8 / 12.2, 2048 x 1024 x 1024 13.1 /
//    10.4 / 10.6; 2560^3 (400 tiles, two per CU) 38.3 / 40.7 / 36.7 stays on the four-wave form
//    (`scripts/t128_spread_trace.sh`, `profiles/gemm_bf16_t128_request_waves.txt`).
namespace {

constexpr int T128 = 128;
constexpr int T128_THREADS = 256;
constexpr int T128_TILE_BYTES = T128 * BK * 2;     // 16 KiB per operand tile
constexpr int T128_BUF_BYTES = 2 * T128_TILE_BYTES;

// Epilogue of the 128x128-, 64x64- and 128x256-tile kernels: a wave's NI x NJ grid of 16x16
// accumulator tiles (swapped-operand layout: lane holds C[m = frow][n = 4*fq + r], r = 0..3) to
// C at (row0, col0). Interior tile with a bf16 result and no accumulate: the short epilogue of
// the ping-pong kernel -- no per-lane bounds, hardware converts, pairs of 16x16 tiles traded
// between the odd and even 16-lane rows so that a lane stores 16 contiguous bytes.
template <bool OUT_F32, int NI, int NJ>
__device__ __forceinline__ void dtf_store_wave_tile(const f32x4 (&acc)[NI][NJ], void* C, long ldc,
                                                    int M, int N, int row0, int col0, int lane,
                                                    float alpha, float beta, bool interior) {
  const int frow = lane & 15, fq = lane >> 4;
  if (!OUT_F32 && beta == 0.f && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
      interior) {
    const int colx = (fq & 1) ? 16 + 4 * (fq - 1) : 4 * fq;
    char* crow = reinterpret_cast<char*>(C) + ((long)(row0 + frow) * ldc + col0 + colx) * 2;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int jp = 0; jp < NJ / 2; ++jp) {
        const f32x4 v0 = acc[i][2 * jp] * alpha, v1 = acc[i][2 * jp + 1] * alpha;
        const unsigned p0a = dtf_f32x2_to_bf16x2(v0[0], v0[1]);
        const unsigned p0b = dtf_f32x2_to_bf16x2(v0[2], v0[3]);
        const unsigned p1a = dtf_f32x2_to_bf16x2(v1[0], v1[1]);
        const unsigned p1b = dtf_f32x2_to_bf16x2(v1[2], v1[3]);
        const auto sa = __builtin_amdgcn_permlane16_swap(p0a, p1a, false, false);
        const auto sb = __builtin_amdgcn_permlane16_swap(p0b, p1b, false, false);
        u32x4 o = {sa[0], sb[0], sa[1], sb[1]};
        *reinterpret_cast<u32x4*>(crow + jp * 64) = o;
      }
      crow += ldc * 32;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = row0 + i * 16 + frow;
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int col = col0 + j * 16 + 4 * fq;
      if (col >= N) continue;
      f32x4 v = acc[i][j] * alpha;
      if (OUT_F32) {
        float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
        if (col + 3 < N) {
          if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(dst) * beta;
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          for (int r = 0; r < 4 && col + r < N; ++r)
            dst[r] = beta != 0.f ? v[r] + beta * dst[r] : v[r];
        }
      } else {
        unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
        if (beta != 0.f)
          for (int r = 0; r < 4 && col + r < N; ++r) v[r] += beta * dtf_bf16_to_f32(dst[r]);
        if (col + 3 < N) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (short)dtf_f32_to_bf16(v[r]);
          *reinterpret_cast<bf16x4*>(dst) = o;
        } else {
          for (int r = 0; r < 4 && col + r < N; ++r) dst[r] = dtf_f32_to_bf16(v[r]);
        }
      }
    }
  }
}

template <bool OUT_F32, int WS, bool FP8 = false>
__device__ __forceinline__ void gemm_bf16_nt_t128_body(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta,
    int ksplits, long plane) {
  // ksplits > 1: workgroup (split, tile) multiplies K-range `split` of K / ksplits columns and
  // writes its partial tile to C + split * plane (elements); planes_reduce_kernel sums them
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x (A tile | B tile)
  const int tid = threadIdx.x, lane = tid & 63;
  // WS: eight waves, 0-3 multiply (64x64 outputs each), 4-7 only issue the LDS-DMA requests
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3;
  const bool producer = WS && wave8 >= 4;
  const int wr = wave >> 1, wc = wave & 1;

  const int ntiles = tiles_m * tiles_n;
  const int split = blockIdx.x / ntiles;
  const int id = dtf_xcd_remap(blockIdx.x - split * ntiles, ntiles);
  K /= ksplits;
  // FP8: e4m3 operands, K-tile 128 -- the same 128 bytes per row, LDS image, requests and
  // fragment reads; a lane's two 16-byte reads form the operand of one block-scaled MFMA with
  // unit scales (see gemm_nt_pp_kernel)
  static_assert(!FP8 || WS, "fp8 operands: only the form with request-issuing waves");
  constexpr int ES = FP8 ? 1 : 2;
  A += (long)split * K * ES;
  Bt += (long)split * K * ES;
  C = reinterpret_cast<char*>(C) + (long)split * plane * (OUT_F32 ? 4 : 2);
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int g = id / per_group;
  const int first_m = g * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (id % per_group) % gsize;
  const int tn = (id % per_group) / gsize;
  const int m0 = tm * T128, n0 = tn * T128;

  // staging: 16 pieces of 8 rows per operand tile, four per wave; the lane offsets are
  // loop-invariant, the K-tile offset is the buffer load's scalar offset
  unsigned voff_a[4], voff_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = wave * 4 + j;
    const int trow = piece * 8 + (lane >> 3);
    const int kc = (lane & 7) ^ ((trow >> 1) & 7);
    int ra = m0 + trow, rb = n0 + trow;
    ra = ra < M - 1 ? ra : M - 1;
    rb = rb < N - 1 ? rb : N - 1;
    voff_a[j] = (unsigned)(ra - m0) * (unsigned)(lda * ES) + 16u * kc;
    voff_b[j] = (unsigned)(rb - n0) * (unsigned)(ldb * ES) + 16u * kc;
  }
  const __amdgpu_buffer_rsrc_t a_rsrc = tile_rsrc(A + (long)m0 * lda * ES);
  const __amdgpu_buffer_rsrc_t b_rsrc = tile_rsrc(Bt + (long)n0 * ldb * ES);
  auto stage = [&](int kt, char* buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(buf + piece * 1024), 16,
                                               voff_a[j], kt * 128, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          b_rsrc, (lds_void*)(buf + T128_TILE_BYTES + piece * 1024), 16, voff_b[j], kt * 128, 0,
          0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = K / (FP8 ? 2 * BK : BK);
  if (!WS || producer) stage(0, smem);
  const int frow = lane & 15, fq = lane >> 4;
  if (WS) {
    // Three K-tile buffers, the requests two K-tiles ahead, one barrier per K-tile. A wave that
    // issues an LDS-DMA request waits 60-180 clocks for the address unit before its next
    // instruction, eight times per K-tile; with one workgroup of four waves on the CU nothing
    // else runs on its SIMD in that time. So waves 4-7 (one per SIMD) issue all the requests
    // and waves 0-3 only read fragments and multiply. (A fourth buffer, with the multiplying
    // waves fetching the next tile's first fragments ahead of the barrier so that their MFMAs
    // start right behind it, changed nothing: 26.8 -> 27.3 us at 1024 x 1024 x 4096. What is
    // left is the request path itself: 32 KiB per K-tile and CU in 0.4 us. K-tiles travelling in
    // pairs as in the 64x64-tile kernel -- two buffers of a pair, one barrier per pair, the next
    // pair requested behind it -- measured slower, 2048^3 21.0 against 19.5 us, 1536^3 16.9 / 13.8:
    // three pair buffers would not fit in LDS, and with two the requests run only one pair ahead
    // and the wait in front of the barrier is a full one.)
    if (producer) {
      if (nkt > 1) stage(1, smem + T128_BUF_BYTES);
      unsigned nxt2 = 2 * T128_BUF_BYTES;            // byte offset of tile kt + 2's buffer
      for (int kt = 0; kt < nkt; ++kt) {
        // own pieces of tile kt have landed (the newest tile's eight requests stay in flight);
        // behind the barrier the multiplying waves are done with the buffer of tile kt - 1
        if (kt + 1 < nkt)
          asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nkt) stage(kt + 2, smem + nxt2);
        nxt2 = nxt2 == 2 * T128_BUF_BYTES ? 0 : nxt2 + T128_BUF_BYTES;
      }
      return;
    }
    // (fragment reads and their counted waits are written out: inside a loop the compiler
    // waits for all sixteen reads before the first MFMA)
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_void*)smem);
    unsigned a_rd[2], b_rd[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      a_rd[s2] = lds0 + chunk_off(wr * 64 + frow, 4 * s2 + fq);
      b_rd[s2] = lds0 + T128_TILE_BYTES + chunk_off(wc * 64 + frow, 4 * s2 + fq);
    }
#define DTF_T128_RD(DST, ADDR, OFF) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DTF_T128_WAIT(N, S2, I)                                                              \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                   \
               : "+v"(bf[S2][0]), "+v"(bf[S2][1]), "+v"(bf[S2][2]), "+v"(bf[S2][3]),         \
                 "+v"(af[S2][I]))
    unsigned cur = 0;                                 // byte offset of tile kt's buffer
    for (int kt = 0; kt < nkt; ++kt) {
      asm volatile("s_barrier" ::: "memory");
      bf16x8 af[2][4], bf[2][4];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned ba = b_rd[s2] + cur, aa = a_rd[s2] + cur;
        DTF_T128_RD(bf[s2][0], ba, 0);
        DTF_T128_RD(bf[s2][1], ba, 2048);
        DTF_T128_RD(bf[s2][2], ba, 4096);
        DTF_T128_RD(bf[s2][3], ba, 6144);
        DTF_T128_RD(af[s2][0], aa, 0);
        DTF_T128_RD(af[s2][1], aa, 2048);
        DTF_T128_RD(af[s2][2], aa, 4096);
        DTF_T128_RD(af[s2][3], aa, 6144);
      }
      if constexpr (FP8) {
        // (reads return in order: a row of MFMAs waits for its second A fragment, the 13th +
        // i-th read)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i == 0) DTF_T128_WAIT(3, 1, 0);
          if (i == 1) DTF_T128_WAIT(2, 1, 1);
          if (i == 2) DTF_T128_WAIT(1, 1, 2);
          if (i == 3) DTF_T128_WAIT(0, 1, 3);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                __builtin_shufflevector((i32x4)bf[0][j], (i32x4)bf[1][j], 0, 1, 2, 3, 4, 5, 6, 7),
                __builtin_shufflevector((i32x4)af[0][i], (i32x4)af[1][i], 0, 1, 2, 3, 4, 5, 6, 7),
                acc[i][j], 0, 0, 0, 0x7f7f7f7fu, 0, 0x7f7f7f7fu);
        }
      } else {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (s2 == 0 && i == 0) DTF_T128_WAIT(11, 0, 0);
          if (s2 == 0 && i == 1) DTF_T128_WAIT(10, 0, 1);
          if (s2 == 0 && i == 2) DTF_T128_WAIT(9, 0, 2);
          if (s2 == 0 && i == 3) DTF_T128_WAIT(8, 0, 3);
          if (s2 == 1 && i == 0) DTF_T128_WAIT(3, 1, 0);
          if (s2 == 1 && i == 1) DTF_T128_WAIT(2, 1, 1);
          if (s2 == 1 && i == 2) DTF_T128_WAIT(1, 1, 2);
          if (s2 == 1 && i == 3) DTF_T128_WAIT(0, 1, 3);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
        }
      }
      cur = cur == 2 * T128_BUF_BYTES ? 0 : cur + T128_BUF_BYTES;
    }
#undef DTF_T128_RD
#undef DTF_T128_WAIT
  } else
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt has landed for every wave, and every wave is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* cur = smem + (kt & 1) * T128_BUF_BYTES;
    if (kt + 1 < nkt) stage(kt + 1, smem + ((kt + 1) & 1) * T128_BUF_BYTES);
    const char* as = cur;
    const char* bs = cur + T128_TILE_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[j] = *reinterpret_cast<const bf16x8*>(
            bs + chunk_off(wc * 64 + j * 16 + frow, 4 * s2 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(
            as + chunk_off(wr * 64 + i * 16 + frow, 4 * s2 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

  dtf_store_wave_tile<OUT_F32, 4, 4>(acc, C, ldc, M, N, m0 + wr * 64, n0 + wc * 64, lane, alpha,
                                     beta, m0 + T128 <= M && n0 + T128 <= N);
}

template <bool OUT_F32>
__global__ __launch_bounds__(T128_THREADS) void gemm_bf16_nt_t128_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta,
    int ksplits, long plane) {
  gemm_bf16_nt_t128_body<OUT_F32, 0>(A, Bt, C, M, N, K, lda, ldb, ldc, tiles_m, tiles_n,
                                         alpha, beta, ksplits, plane);
}

// the same tile with four request-issuing waves next to the four multiplying ones, three K-tile
// buffers: one workgroup per CU
template <bool OUT_F32>
__global__ __launch_bounds__(2 * T128_THREADS) void gemm_bf16_nt_t128ws_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta,
    int ksplits, long plane) {
  gemm_bf16_nt_t128_body<OUT_F32, 1>(A, Bt, C, M, N, K, lda, ldb, ldc, tiles_m, tiles_n,
                                        alpha, beta, ksplits, plane);
}

// ---- 64x64 tiles with request-issuing waves: products of at most one such tile per CU ---------
// The kernel above is bound by its request path (32 KiB per K-tile and CU in 0.4 us), so a
// product of few 128x128 tiles runs faster on four times as many CUs with a quarter of the tile
// each: waves 0-3 multiply 32x32 outputs each, waves 4-7 issue the four requests a K-tile needs
// of each of them (8 KiB per operand tile), three buffers of 16 KiB, one barrier per K-tile.
constexpr int T64 = 64;
constexpr int T64_TILE_BYTES = T64 * BK * 2;       // 8 KiB per operand tile
constexpr int T64_BUF_BYTES = 2 * T64_TILE_BYTES;
constexpr int T64_PAIR_BYTES = 2 * T64_BUF_BYTES;  // two K-tiles

template <bool OUT_F32, bool FP8>
__global__ __launch_bounds__(512) void gemm_nt_t64ws_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta,
    int ksplits, long plane) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // three buffers of (A tile | B tile) sets
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3;
  const bool producer = wave8 >= 4;
  const int wr = wave >> 1, wc = wave & 1;
  constexpr int ES = FP8 ? 1 : 2;

  const int ntiles = tiles_m * tiles_n;
  const int split = blockIdx.x / ntiles;
  const int id = dtf_xcd_remap(blockIdx.x - split * ntiles, ntiles);
  K /= ksplits;
  A += (long)split * K * ES;
  Bt += (long)split * K * ES;
  C = reinterpret_cast<char*>(C) + (long)split * plane * (OUT_F32 ? 4 : 2);
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int g = id / per_group;
  const int first_m = g * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (id % per_group) % gsize;
  const int tn = (id % per_group) / gsize;
  const int m0 = tm * T64, n0 = tn * T64;
  const int nkt = K / (FP8 ? 2 * BK : BK);
  const int npairs = (nkt + 1) >> 1;

  if (producer) {
    // 8 pieces of 8 rows per operand tile, two per wave
    unsigned voff_a[2], voff_b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = wave * 2 + j;
      const int trow = piece * 8 + (lane >> 3);
      const int kc = (lane & 7) ^ ((trow >> 1) & 7);
      int ra = m0 + trow, rb = n0 + trow;
      ra = ra < M - 1 ? ra : M - 1;
      rb = rb < N - 1 ? rb : N - 1;
      voff_a[j] = (unsigned)(ra - m0) * (unsigned)(lda * ES) + 16u * kc;
      voff_b[j] = (unsigned)(rb - n0) * (unsigned)(ldb * ES) + 16u * kc;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = tile_rsrc(A + (long)m0 * lda * ES);
    const __amdgpu_buffer_rsrc_t b_rsrc = tile_rsrc(Bt + (long)n0 * ldb * ES);
    auto stage = [&](int kt, char* buf) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int piece = wave * 2 + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(buf + piece * 1024), 16,
                                                 voff_a[j], kt * 128, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            b_rsrc, (lds_void*)(buf + T64_TILE_BYTES + piece * 1024), 16, voff_b[j], kt * 128, 0,
            0);
      }
    };
    // K-tiles travel in pairs (one barrier per 128 k): a pair is eight requests per wave; behind
    // an odd last tile its own requests are issued once more, so that the counted wait is the
    // same for every pair (the multiplying waves do not read that half)
    auto stage_pair = [&](int p, char* buf) {
      stage(2 * p, buf);
      stage(2 * p + 1 < nkt ? 2 * p + 1 : nkt - 1, buf + T64_BUF_BYTES);
    };
    stage_pair(0, smem);
    if (npairs > 1) stage_pair(1, smem + T64_PAIR_BYTES);
    unsigned nxt2 = 2 * T64_PAIR_BYTES;
    for (int p = 0; p < npairs; ++p) {
      // own pieces of pair p have landed (the newest pair's eight requests stay in flight);
      // behind the barrier the multiplying waves are done with the buffer of pair p - 1
      if (p + 1 < npairs)
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (p + 2 < npairs) stage_pair(p + 2, smem + nxt2);
      nxt2 = nxt2 == 2 * T64_PAIR_BYTES ? 0 : nxt2 + T64_PAIR_BYTES;
    }
    return;
  }

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_void*)smem);
  unsigned a_rd[2], b_rd[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    a_rd[s2] = lds0 + chunk_off(wr * 32 + frow, 4 * s2 + fq);
    b_rd[s2] = lds0 + T64_TILE_BYTES + chunk_off(wc * 32 + frow, 4 * s2 + fq);
  }
#define DTF_T64_RD(DST, ADDR, OFF) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DTF_T64_WAIT(N, S2, I) \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(bf[S2][0]), "+v"(bf[S2][1]), "+v"(af[S2][I]))
  unsigned pair = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (!(kt & 1)) asm volatile("s_barrier" ::: "memory");      // one barrier per pair of K-tiles
    const unsigned cur = pair + (kt & 1) * T64_BUF_BYTES;
    bf16x8 af[2][2], bf[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const unsigned ba = b_rd[s2] + cur, aa = a_rd[s2] + cur;
      DTF_T64_RD(bf[s2][0], ba, 0);
      DTF_T64_RD(bf[s2][1], ba, 2048);
      DTF_T64_RD(af[s2][0], aa, 0);
      DTF_T64_RD(af[s2][1], aa, 2048);
    }
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i == 0) DTF_T64_WAIT(1, 1, 0);
        if (i == 1) DTF_T64_WAIT(0, 1, 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              __builtin_shufflevector((i32x4)bf[0][j], (i32x4)bf[1][j], 0, 1, 2, 3, 4, 5, 6, 7),
              __builtin_shufflevector((i32x4)af[0][i], (i32x4)af[1][i], 0, 1, 2, 3, 4, 5, 6, 7),
              acc[i][j], 0, 0, 0, 0x7f7f7f7fu, 0, 0x7f7f7f7fu);
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (s2 == 0 && i == 0) DTF_T64_WAIT(5, 0, 0);
          if (s2 == 0 && i == 1) DTF_T64_WAIT(4, 0, 1);
          if (s2 == 1 && i == 0) DTF_T64_WAIT(1, 1, 0);
          if (s2 == 1 && i == 1) DTF_T64_WAIT(0, 1, 1);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
        }
    }
    if (kt & 1) pair = pair == 2 * T64_PAIR_BYTES ? 0 : pair + T64_PAIR_BYTES;
  }
#undef DTF_T64_RD
#undef DTF_T64_WAIT

  dtf_store_wave_tile<OUT_F32, 2, 2>(acc, C, ldc, M, N, m0 + wr * 32, n0 + wc * 32, lane, alpha,
                                     beta, m0 + T64 <= M && n0 + T64 <= N);
}

// ---- 128x256 tiles with request-issuing waves: grids of at most one such tile per CU ---------
// For grids too large for a CU per 128x128 tile: the same structure with 64x128 outputs per
// multiplying wave -- 48 KiB of requests per K-tile for twice the flops of the 128x128 tile (0.75
// of its bytes per flop), twelve requests per K-tile and request wave, three buffers of 48 KiB.
constexpr int TW_M = 128, TW_N = 256;
constexpr int TW_A_BYTES = TW_M * BK * 2;          // 16 KiB
constexpr int TW_B_BYTES = TW_N * BK * 2;          // 32 KiB
constexpr int TW_BUF_BYTES = TW_A_BYTES + TW_B_BYTES;

template <bool OUT_F32, bool FP8>
__global__ __launch_bounds__(512) void gemm_nt_t128x256ws_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // three buffers of (A tile | B tile) sets
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave8 & 3;
  const bool producer = wave8 >= 4;
  const int wr = wave >> 1, wc = wave & 1;
  constexpr int ES = FP8 ? 1 : 2;

  const int ntiles = tiles_m * tiles_n;
  const int id = dtf_xcd_remap(blockIdx.x, ntiles);
  constexpr int GROUP = 8;
  const int per_group = GROUP * tiles_n;
  const int g = id / per_group;
  const int first_m = g * GROUP;
  const int gsize = (tiles_m - first_m) < GROUP ? (tiles_m - first_m) : GROUP;
  const int tm = first_m + (id % per_group) % gsize;
  const int tn = (id % per_group) / gsize;
  const int m0 = tm * TW_M, n0 = tn * TW_N;
  const int nkt = K / (FP8 ? 2 * BK : BK);

  if (producer) {
    // 16 pieces of 8 rows of the A tile and 32 of the B tile: four and eight per wave
    unsigned voff_a[4], voff_b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int piece_b = wave * 8 + j;
      const int trow_b = piece_b * 8 + (lane >> 3);
      int rb = n0 + trow_b;
      rb = rb < N - 1 ? rb : N - 1;
      voff_b[j] = (unsigned)(rb - n0) * (unsigned)(ldb * ES) +
                  16u * ((lane & 7) ^ ((trow_b >> 1) & 7));
      if (j < 4) {
        const int piece_a = wave * 4 + j;
        const int trow_a = piece_a * 8 + (lane >> 3);
        int ra = m0 + trow_a;
        ra = ra < M - 1 ? ra : M - 1;
        voff_a[j] = (unsigned)(ra - m0) * (unsigned)(lda * ES) +
                    16u * ((lane & 7) ^ ((trow_a >> 1) & 7));
      }
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = tile_rsrc(A + (long)m0 * lda * ES);
    const __amdgpu_buffer_rsrc_t b_rsrc = tile_rsrc(Bt + (long)n0 * ldb * ES);
    auto stage = [&](int kt, char* buf) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            b_rsrc, (lds_void*)(buf + TW_A_BYTES + (wave * 8 + j) * 1024), 16, voff_b[j],
            kt * 128, 0, 0);
        if (j < 4)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              a_rsrc, (lds_void*)(buf + (wave * 4 + j) * 1024), 16, voff_a[j], kt * 128, 0, 0);
      }
    };
    stage(0, smem);
    if (nkt > 1) stage(1, smem + TW_BUF_BYTES);
    unsigned nxt2 = 2 * TW_BUF_BYTES;
    for (int kt = 0; kt < nkt; ++kt) {
      // own pieces of tile kt have landed (the newest tile's twelve requests stay in flight);
      // behind the barrier the multiplying waves are done with the buffer of tile kt - 1
      if (kt + 1 < nkt)
        asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      if (kt + 2 < nkt) stage(kt + 2, smem + nxt2);
      nxt2 = nxt2 == 2 * TW_BUF_BYTES ? 0 : nxt2 + TW_BUF_BYTES;
    }
    return;
  }

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fq = lane >> 4;
  const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_void*)smem);
  unsigned a_rd[2], b_rd[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    a_rd[s2] = lds0 + chunk_off(wr * 64 + frow, 4 * s2 + fq);
    b_rd[s2] = lds0 + TW_A_BYTES + chunk_off(wc * 128 + frow, 4 * s2 + fq);
  }
#define DTF_TW_RD(DST, ADDR, OFF) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define DTF_TW_WAIT(N, S2, I)                                                                \
  asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                   \
               : "+v"(bf[S2][0]), "+v"(bf[S2][1]), "+v"(bf[S2][2]), "+v"(bf[S2][3]),         \
                 "+v"(bf[S2][4]), "+v"(bf[S2][5]), "+v"(bf[S2][6]), "+v"(bf[S2][7]),         \
                 "+v"(af[S2][I]))
  unsigned cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_barrier" ::: "memory");
    bf16x8 af[2][4], bf[2][8];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const unsigned ba = b_rd[s2] + cur, aa = a_rd[s2] + cur;
      DTF_TW_RD(bf[s2][0], ba, 0);
      DTF_TW_RD(bf[s2][1], ba, 2048);
      DTF_TW_RD(bf[s2][2], ba, 4096);
      DTF_TW_RD(bf[s2][3], ba, 6144);
      DTF_TW_RD(bf[s2][4], ba, 8192);
      DTF_TW_RD(bf[s2][5], ba, 10240);
      DTF_TW_RD(bf[s2][6], ba, 12288);
      DTF_TW_RD(bf[s2][7], ba, 14336);
      DTF_TW_RD(af[s2][0], aa, 0);
      DTF_TW_RD(af[s2][1], aa, 2048);
      DTF_TW_RD(af[s2][2], aa, 4096);
      DTF_TW_RD(af[s2][3], aa, 6144);
    }
    // 24 reads, returned in order: a row of MFMAs waits for the read of its A fragment
    if constexpr (FP8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i == 0) DTF_TW_WAIT(3, 1, 0);
        if (i == 1) DTF_TW_WAIT(2, 1, 1);
        if (i == 2) DTF_TW_WAIT(1, 1, 2);
        if (i == 3) DTF_TW_WAIT(0, 1, 3);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
              __builtin_shufflevector((i32x4)bf[0][j], (i32x4)bf[1][j], 0, 1, 2, 3, 4, 5, 6, 7),
              __builtin_shufflevector((i32x4)af[0][i], (i32x4)af[1][i], 0, 1, 2, 3, 4, 5, 6, 7),
              acc[i][j], 0, 0, 0, 0x7f7f7f7fu, 0, 0x7f7f7f7fu);
      }
    } else {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (s2 == 0 && i == 0) DTF_TW_WAIT(15, 0, 0);
          if (s2 == 0 && i == 1) DTF_TW_WAIT(14, 0, 1);
          if (s2 == 0 && i == 2) DTF_TW_WAIT(13, 0, 2);
          if (s2 == 0 && i == 3) DTF_TW_WAIT(12, 0, 3);
          if (s2 == 1 && i == 0) DTF_TW_WAIT(3, 1, 0);
          if (s2 == 1 && i == 1) DTF_TW_WAIT(2, 1, 1);
          if (s2 == 1 && i == 2) DTF_TW_WAIT(1, 1, 2);
          if (s2 == 1 && i == 3) DTF_TW_WAIT(0, 1, 3);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[i][j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s2][j], af[s2][i], acc[i][j], 0, 0, 0);
        }
    }
    cur = cur == 2 * TW_BUF_BYTES ? 0 : cur + TW_BUF_BYTES;
  }
#undef DTF_TW_RD
#undef DTF_TW_WAIT

  dtf_store_wave_tile<OUT_F32, 4, 8>(acc, C, ldc, M, N, m0 + wr * 64, n0 + wc * 128, lane, alpha,
                                     beta, m0 + TW_M <= M && n0 + TW_N <= N);
}

template <bool OUT_F32>
__global__ __launch_bounds__(2 * T128_THREADS) void gemm_fp8_nt_t128ws_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta,
    int ksplits, long plane) {
  gemm_bf16_nt_t128_body<OUT_F32, 1, true>(A, Bt, C, M, N, K, lda, ldb, ldc, tiles_m, tiles_n,
                                           alpha, beta, ksplits, plane);
}

}  // namespace

// C[M,N] (bf16, or fp32 when out_f32) = alpha * A[M,K] * Bt[N,K]^T. Requirements of the fast
// kernel: K % 64 == 0, lda/ldb % 8 == 0, 16-byte aligned A/Bt, ldc % 4 == 0 and 16-byte
// (fp32) / 8-byte (bf16) aligned C. Returns DTF_ERR_UNSUPPORTED otherwise (callers then use
// dtf_gemm_bf16_generic).
// variant & 15: 0 = default (4; 5, 6 or 9 for tile grids that fill the chip badly), 1 = two-phase
// double-buffer, 6 / 9 = ping-pong on 192 x 256 / 128 x 256 block tiles, 7 / 8 = the four-wave
// kernels on an LDS ring / with register-staged loads,
// 2 = ping-pong with one quadrant per phase, 3 = four-wave kernel (128x128 per wave),
// 4 = ping-pong with two quadrants per phase, 5 = 128x128 tiles with two workgroups per CU,
// 11 = 128x128 tiles with request-issuing waves (one workgroup per CU), 12 = 64x64 tiles with
// request-issuing waves;
// variant >> 4: timing-experiment bits of the one-quadrant ping-pong kernel (0 in production).
// The staging DMA addresses a block tile with 32-bit offsets: 256 rows of lda / ldb bytes must
// stay below 4 GiB (DTF_ERR_UNSUPPORTED otherwise).
// default kernel choice: grids of more than 32 and fewer than this many 256x256 tiles run on
// 128x128 tiles. Measured (128x128 against 256x256 tiles): 2048^3 (64 tiles) 617 vs 395 TF,
// 2048x4096x4096 (128 tiles) 1020 vs 853 TF; 3072^3 (144 tiles) 707 vs 870 TF, 4096^3 (256
// tiles) 1035 vs 1388 TF.
// (tiles of 128x256 per 256 CUs up to which the bf16 launch takes the 128x256-tile kernel at any K)
#ifndef DTF_TW_MAX_TILES
#define DTF_TW_MAX_TILES 208
#endif
#ifndef DTF_T128_BELOW_TILES
#define DTF_T128_BELOW_TILES 129
#endif

// workgroups of the persistent ping-pong kernel: one per CU (128 KiB of LDS each), fewer when
// there are fewer tiles
static int dtf_pp_grid(int ntiles) {
  static std::atomic<int> cus_of[64];
  const int dev = dtf_current_device();
  int cus = cus_of[dev].load(std::memory_order_relaxed);
  if (cus == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        n <= 0)
      n = 256;
    cus_of[dev].store(cus = n, std::memory_order_relaxed);
  }
  return ntiles < cus ? ntiles : cus;
}

// Rows of the ping-pong kernel's block tile (columns are always 256): 256, 192 or 128. The
// persistent kernel's time goes with (rounds of the chip) x (rows of a tile): 3072^2 outputs are
// one round of 256-row tiles with 112 CUs idle, or one round of 192-row tiles with 64 idle and
// three quarters of the work per CU; 2048 x 4096 outputs are 128 tiles of 256 rows (half the
// chip) or one full round of 128-row tiles. A smaller tile moves more operand bytes per flop and
// hands the matrix pipe over more often, so it must win by its penalty (8 % for 192 rows, 15 %
// for 128) and put a workgroup on more than half of the CUs (with 130-176 workgroups of 128 rows
// it still beats the 128 x 128-tile kernel by 4-7 %: 2304^2 x 4096 51.1 against 53.3 us, 2048 x
// 2560 x 4096 49.0 / 51.5, 1664 x 2560 x 4096 48.0 / 50.4; with exactly 128 it loses, 2048^2 x 4096
// 47.4 / 41.4 -- `scripts/bf16_rows_threshold_sweep.py`). Measured (smaller tile / 256-row or
// 128 x 128 tiles / torch.matmul, TF): 3072^3 1180 / 1068 / 1038, 6144^3 1448 / 1323 / 1398,
// 3000 x 3072 x 4096 1209 / 1091 / 1085 (192 rows); 2048 x 4096 x 4096 1236 / 1124 / 1257,
// 3072 x 2048 x 4096 1034 / 937 / 1158, 1024 x 8192 x 8192 1286 / 1175 / 1306 (128 rows);
// where the rule keeps the other kernels: 7168^3 1302 (192) against 1338, 2048^3 628 (128 rows,
// 128 workgroups) against 683 on 128 x 128 tiles, 2560 x 4096 x 4096 889 (128) against 1278 (192).
// Grids of at most 32 tiles of 256 x 256 are left to the split-K plan.
static int dtf_pp_rows(int M, int N) {
  const long cus = dtf_pp_grid(1 << 30);
  const long tn = (N + BN - 1) / BN;
  const long t256 = (long)((M + 255) / 256) * tn;
  if (t256 <= 32) return 256;
  int best = 256;
  long best_cost = ((t256 + cus - 1) / cus) * 256 * 100;
  const int rows[2] = {192, 128}, penalty[2] = {108, 115};
  for (int c = 0; c < 2; ++c) {
    const long t = (long)((M + rows[c] - 1) / rows[c]) * tn;
    if (t * 2 <= cus) continue;      // (at most half the CUs: 128 x 128 tiles fill the chip better)
    const long cost = ((t + cus - 1) / cus) * rows[c] * penalty[c];
    if (cost < best_cost) {
      best = rows[c];
      best_cost = cost;
    }
  }
  return best;
}

// Which kernel the last bf16 / fp8 launch of this process took (the `variant` numbers: 4 the
// 256x256 ping-pong kernel, 6 / 9 its 192- / 128-row tilings, 5 / 11 the two forms of the 128x128
// tile, 12 64x64 tiles, 13 128x256 tiles; 0 = another path). For tests of the launch plan.
static std::atomic<int> g_last_kernel{0};
DTF_API int dtf_gemm_last_kernel() { return g_last_kernel.load(std::memory_order_relaxed); }

template <typename KernelT>
static void dtf_allow_lds(KernelT kernel, size_t lds) {
  hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// launch of the 64x64-tile kernel (operand contract checked by the caller)
template <bool FP8>
static void dtf_launch_t64ws(const void* A, const void* Bt, void* C, int M, int N, int K, long lda,
                             long ldb, long ldc, float alpha, float beta, int out_f32, int ksplits,
                             long plane, hipStream_t stream) {
  g_last_kernel.store(12, std::memory_order_relaxed);
  const size_t lds3 = 3 * T64_PAIR_BYTES;
  { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_nt_t64ws_kernel<true, FP8>, lds3);
    dtf_allow_lds(gemm_nt_t64ws_kernel<false, FP8>, lds3);
  }); }
  const int tm = (M + T64 - 1) / T64, tn = (N + T64 - 1) / T64;
  if (out_f32)
    hipLaunchKernelGGL((gemm_nt_t64ws_kernel<true, FP8>), dim3(tm * tn * ksplits), dim3(512), lds3,
                       stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tm, tn,
                       alpha, beta, ksplits, plane);
  else
    hipLaunchKernelGGL((gemm_nt_t64ws_kernel<false, FP8>), dim3(tm * tn * ksplits), dim3(512),
                       lds3, stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                       tm, tn, alpha, beta, ksplits, plane);
}

// number of 64x64 tiles of an M x N result
static int dtf_t64_tiles(int M, int N) { return ((M + T64 - 1) / T64) * ((N + T64 - 1) / T64); }

// launch of the 128x256-tile kernel (operand contract checked by the caller)
template <bool FP8>
static void dtf_launch_t128x256ws(const void* A, const void* Bt, void* C, int M, int N, int K,
                                  long lda, long ldb, long ldc, float alpha, float beta,
                                  int out_f32, hipStream_t stream) {
  g_last_kernel.store(13, std::memory_order_relaxed);
  const size_t lds3 = 3 * TW_BUF_BYTES;
  { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_nt_t128x256ws_kernel<true, FP8>, lds3);
    dtf_allow_lds(gemm_nt_t128x256ws_kernel<false, FP8>, lds3);
  }); }
  const int tm = (M + TW_M - 1) / TW_M, tn = (N + TW_N - 1) / TW_N;
  if (out_f32)
    hipLaunchKernelGGL((gemm_nt_t128x256ws_kernel<true, FP8>), dim3(tm * tn), dim3(512), lds3,
                       stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tm, tn,
                       alpha, beta);
  else
    hipLaunchKernelGGL((gemm_nt_t128x256ws_kernel<false, FP8>), dim3(tm * tn), dim3(512), lds3,
                       stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tm, tn,
                       alpha, beta);
}

DTF_API int dtf_gemm_bf16_nt(const void* A, const void* Bt, void* C, int M, int N, int K, long lda,
                             long ldb, long ldc, float alpha, float beta, int out_f32,
                             int variant, hipStream_t stream) {
  if (!A || !Bt || !C || M <= 0 || N <= 0 || K <= 0) return DTF_ERR_BAD_ARG;
  if (lda < K || ldb < K || ldc < N) return DTF_ERR_BAD_ARG;
  if (K % BK || lda % 8 || ldb % 8 || ldc % 4 || ((uintptr_t)A & 15) || ((uintptr_t)Bt & 15) ||
      ((uintptr_t)C & (out_f32 ? 15 : 7)))
    return DTF_ERR_UNSUPPORTED;
  if (lda * 2 * BM >= (1L << 32) || ldb * 2 * BN >= (1L << 32)) return DTF_ERR_UNSUPPORTED;
  g_last_kernel.store(0, std::memory_order_relaxed);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = 2 * BUF_BYTES;   // 128 KiB
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_bf16_nt_kernel<true>, lds);
    dtf_allow_lds(gemm_bf16_nt_kernel<false>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 0, false>, lds);
#ifdef DTF_GEMM_EXPERIMENTS
    dtf_allow_lds(gemm_nt_pp_kernel<false, 1, false>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 2, false>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 4, false>, lds);
#endif
  }); }
  dim3 grid(tiles_m * tiles_n), block(NTHREADS);
  const dim3 pp_grid(dtf_pp_grid(tiles_m * tiles_n));
  if ((variant & 15) == 10) {      // four-wave kernel, LDS-DMA two K-tiles ahead
    const size_t p2_lds = 2 * BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_bf16_nt_w4p2_kernel<true>, p2_lds);
      dtf_allow_lds(gemm_bf16_nt_w4p2_kernel<false>, p2_lds);
    }); }
#ifdef DTF_GEMM_EXPERIMENTS
#define DTF_P2_DBG(D)                                                                          \
  if ((variant >> 4) == D) {                                                                   \
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {                        \
      dtf_allow_lds((gemm_bf16_nt_w4p2_kernel<false, D>), p2_lds); }); }                       \
    hipLaunchKernelGGL((gemm_bf16_nt_w4p2_kernel<false, D>), pp_grid, dim3(W4_THREADS),        \
                       p2_lds, stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb,  \
                       ldc, tiles_m, tiles_n, alpha, beta);                                    \
    DTF_LAUNCH_CHECK();                                                                        \
    return DTF_OK;                                                                             \
  }
    DTF_P2_DBG(1) DTF_P2_DBG(2) DTF_P2_DBG(4) DTF_P2_DBG(8) DTF_P2_DBG(5) DTF_P2_DBG(16)
    DTF_P2_DBG(32) DTF_P2_DBG(48) DTF_P2_DBG(65) DTF_P2_DBG(69) DTF_P2_DBG(144)
    DTF_P2_DBG(256) DTF_P2_DBG(260) DTF_P2_DBG(768) DTF_P2_DBG(1280) DTF_P2_DBG(1792)
    DTF_P2_DBG(2048) DTF_P2_DBG(2052) DTF_P2_DBG(4096) DTF_P2_DBG(8192) DTF_P2_DBG(16384)
    DTF_P2_DBG(32768) DTF_P2_DBG(65536) DTF_P2_DBG(12288) DTF_P2_DBG(49152)
#undef DTF_P2_DBG
#endif
    if (out_f32)
      hipLaunchKernelGGL(gemm_bf16_nt_w4p2_kernel<true>, pp_grid, dim3(W4_THREADS), p2_lds,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tiles_m, tiles_n, alpha, beta);
    else
      hipLaunchKernelGGL(gemm_bf16_nt_w4p2_kernel<false>, pp_grid, dim3(W4_THREADS), p2_lds,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tiles_m, tiles_n, alpha, beta);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if ((variant & 15) == 8) {       // four-wave kernel, register-staged loads
    if (M % BM || N % BN || K % (2 * BK)) return DTF_ERR_UNSUPPORTED;
    const size_t wr_lds = 2 * BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_bf16_nt_w4reg_kernel<true>, wr_lds);
      dtf_allow_lds(gemm_bf16_nt_w4reg_kernel<false>, wr_lds);
    }); }
#ifdef DTF_GEMM_EXPERIMENTS
#define DTF_WRG_DBG(D)                                                                         \
  if ((variant >> 4) == D) {                                                                   \
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {                        \
      dtf_allow_lds((gemm_bf16_nt_w4reg_kernel<false, D>), wr_lds); }); }                      \
    hipLaunchKernelGGL((gemm_bf16_nt_w4reg_kernel<false, D>), grid, dim3(W4_THREADS), wr_lds,  \
                       stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,     \
                       tiles_m, tiles_n, alpha, beta);                                         \
    DTF_LAUNCH_CHECK();                                                                        \
    return DTF_OK;                                                                             \
  }
    DTF_WRG_DBG(1) DTF_WRG_DBG(2) DTF_WRG_DBG(3) DTF_WRG_DBG(4) DTF_WRG_DBG(7)
#undef DTF_WRG_DBG
#endif
    if (out_f32)
      hipLaunchKernelGGL(gemm_bf16_nt_w4reg_kernel<true>, grid, dim3(W4_THREADS), wr_lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta);
    else
      hipLaunchKernelGGL(gemm_bf16_nt_w4reg_kernel<false>, grid, dim3(W4_THREADS), wr_lds,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tiles_m, tiles_n, alpha, beta);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if ((variant & 15) == 7) {       // four-wave kernel on the five-stage ring
    const size_t rs_lds = (size_t)RS_STAGES * RS_STAGE_BYTES;       // all 160 KiB of the CU
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_bf16_nt_w4ring_kernel<true>, rs_lds);
      dtf_allow_lds(gemm_bf16_nt_w4ring_kernel<false>, rs_lds);
    }); }
#ifdef DTF_GEMM_EXPERIMENTS
#define DTF_RS_DBG(D)                                                                          \
  if ((variant >> 4) == D) {                                                                   \
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {                        \
      dtf_allow_lds((gemm_bf16_nt_w4ring_kernel<false, D>), rs_lds); }); }                     \
    hipLaunchKernelGGL((gemm_bf16_nt_w4ring_kernel<false, D>), grid, dim3(W4_THREADS), rs_lds, \
                       stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,     \
                       tiles_m, tiles_n, alpha, beta);                                         \
    DTF_LAUNCH_CHECK();                                                                        \
    return DTF_OK;                                                                             \
  }
    DTF_RS_DBG(1) DTF_RS_DBG(4) DTF_RS_DBG(5)
#undef DTF_RS_DBG
#endif
    if (out_f32)
      hipLaunchKernelGGL(gemm_bf16_nt_w4ring_kernel<true>, grid, dim3(W4_THREADS), rs_lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta);
    else
      hipLaunchKernelGGL(gemm_bf16_nt_w4ring_kernel<false>, grid, dim3(W4_THREADS), rs_lds,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tiles_m, tiles_n, alpha, beta);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if ((variant & 15) == 3) {       // four-wave kernel
        const size_t w4_lds = 2 * BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<true>, w4_lds);
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false>, w4_lds);
#ifdef DTF_GEMM_EXPERIMENTS
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false, 1>, w4_lds);
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false, 2>, w4_lds);
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false, 4>, w4_lds);
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false, 6>, w4_lds);
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false, 14>, w4_lds);
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false, 5>, w4_lds);
      dtf_allow_lds(gemm_bf16_nt_w4_kernel<false, 13>, w4_lds);
#endif
    }); }
#ifdef DTF_GEMM_EXPERIMENTS
#define DTF_W4_DBG(D)                                                                          \
  if ((variant >> 4) == D) {                                                                   \
    hipLaunchKernelGGL((gemm_bf16_nt_w4_kernel<false, D>), grid, dim3(W4_THREADS), w4_lds,     \
                       stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,     \
                       tiles_m, tiles_n, alpha, beta);                                         \
    DTF_LAUNCH_CHECK();                                                                        \
    return DTF_OK;                                                                             \
  }
    DTF_W4_DBG(1) DTF_W4_DBG(2) DTF_W4_DBG(4) DTF_W4_DBG(6) DTF_W4_DBG(14) DTF_W4_DBG(5)
    DTF_W4_DBG(13)
#undef DTF_W4_DBG
#endif
    if (out_f32)
      hipLaunchKernelGGL(gemm_bf16_nt_w4_kernel<true>, grid, dim3(W4_THREADS), w4_lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta);
    else
      hipLaunchKernelGGL(gemm_bf16_nt_w4_kernel<false>, grid, dim3(W4_THREADS), w4_lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
#ifdef DTF_GEMM_EXPERIMENTS
  if ((variant & 15) == 4 && (variant >> 4) != 0) {   // timing experiments on the default schedule
        { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_nt_pp_kernel<false, 8, false, true>, lds + 1024);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 10, false, true>, lds + 1024);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 9, false, true>, lds + 1024);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 16, false, true>, lds);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 32, false, true>, lds);
    }); }
    const int dbg = variant >> 4;
#define DTF_PP_TWEAK(D)                                                                         \
  if (dbg == D) {                                                                               \
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {                         \
      dtf_allow_lds(gemm_nt_pp_kernel<false, D, false, true>, lds); }); }                       \
    hipLaunchKernelGGL((gemm_nt_pp_kernel<false, D, false, true>), pp_grid, block, lds, stream, \
                       (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,     \
                       tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);                          \
    DTF_LAUNCH_CHECK();                                                                         \
    return DTF_OK;                                                                              \
  }
#define DTF_PP_TWEAK_MI(D, MIV, ROWS)                                                           \
  if (dbg == D + 1 + (4 - MIV)) {                                                               \
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {                         \
      dtf_allow_lds((gemm_nt_pp_kernel<false, D, false, true, false, MIV>), lds); }); }         \
    const int tmr = (M + ROWS - 1) / ROWS;                                                      \
    hipLaunchKernelGGL((gemm_nt_pp_kernel<false, D, false, true, false, MIV>),                  \
                       dim3(dtf_pp_grid(tmr * tiles_n)), block, lds, stream, (const char*)A,    \
                       (const char*)Bt, C, M, N, K, lda, ldb, ldc, tmr, tiles_n, alpha, beta,   \
                       1, 0L, nullptr, nullptr);                                                \
    DTF_LAUNCH_CHECK();                                                                         \
    return DTF_OK;                                                                              \
  }
    // (experiment number + 1 / + 2: the same bits on 192- / 128-row tiles)
    DTF_PP_TWEAK(64) DTF_PP_TWEAK(2112) DTF_PP_TWEAK_MI(64, 3, 192) DTF_PP_TWEAK_MI(64, 2, 128)
    DTF_PP_TWEAK_MI(2048, 3, 192) DTF_PP_TWEAK_MI(2048, 2, 128) DTF_PP_TWEAK_MI(2112, 3, 192)
    DTF_PP_TWEAK_MI(2112, 2, 128)
    DTF_PP_TWEAK(256) DTF_PP_TWEAK(2048) DTF_PP_TWEAK(2304) DTF_PP_TWEAK(4096) DTF_PP_TWEAK(8192) DTF_PP_TWEAK(2) DTF_PP_TWEAK(1)
#undef DTF_PP_TWEAK
    if (dbg == 8)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 8, false, true>), pp_grid, block, lds + 1024,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tiles_m, tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else if (dbg == 10)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 10, false, true>), pp_grid, block, lds + 1024,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tiles_m, tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else if (dbg == 9)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 9, false, true>), pp_grid, block, lds + 1024,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tiles_m, tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else if (dbg == 16)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 16, false, true>), pp_grid, block, lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else if (dbg == 32)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 32, false, true>), grid, block, lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else return DTF_ERR_UNSUPPORTED;
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
#endif
  // 128x256 tiles with request-issuing waves (variant 13; by default for grids of more than one
  // 128x128 tile per CU whose 128x256 tiles leave a fifth of the CUs idle or more: with every CU
  // busy the product is bound by what the L2s deliver to all of them and the eight-wave tilings
  // are as fast or faster -- 2048 x 4096 x 4096 1010 against 1177 TFLOP/s -- with CUs to spare it
  // is not: 2560^3 941 against 791, torch.matmul 877; up to K = 2048 it is ahead with every CU
  // busy too -- 4096 x 2048 x 2048 1085 against 1021, 3584 x 2048 x 2048 979 / 907, 2560 x 3072 x
  // 2048 1050 / 994; `scripts/bf16_tw_trace.sh`)
  const int tw_tiles = ((M + TW_M - 1) / TW_M) * ((N + TW_N - 1) / TW_N);
  if ((variant & 15) == 13 ||
      (variant == 0 &&
       (tw_tiles * 256L <= (long)DTF_TW_MAX_TILES * dtf_pp_grid(1 << 20) ||
        (dtf_pp_grid(tw_tiles) == tw_tiles && K <= 2048)) &&
       dtf_pp_grid(((M + T128 - 1) / T128) * ((N + T128 - 1) / T128)) !=
           ((M + T128 - 1) / T128) * ((N + T128 - 1) / T128))) {
    dtf_launch_t128x256ws<false>(A, Bt, C, M, N, K, lda, ldb, ldc, alpha, beta, out_f32, stream);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  // (a grid with a CU for every 128x128 tile goes to the kernels with request-issuing waves
  // below, whatever the row-tile rule would make of a ragged N)
  const int t128_all = ((M + T128 - 1) / T128) * ((N + T128 - 1) / T128);
  const int pp_rows =
      variant == 0 && dtf_pp_grid(t128_all) != t128_all ? dtf_pp_rows(M, N) : 256;
  if ((variant & 15) == 9 || pp_rows == 128) {       // 128-row block tiles (MI = 2)
    g_last_kernel.store(9, std::memory_order_relaxed);
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false, true, false, 2>, lds);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 0, false, true, false, 2>, lds);
    }); }
    const int tm128r = (M + 127) / 128;
    const dim3 grid128r(dtf_pp_grid(tm128r * tiles_n));
    if (out_f32)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<true, 0, false, true, false, 2>), grid128r, block, lds,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tm128r, tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 0, false, true, false, 2>), grid128r, block,
                         lds, stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tm128r, tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  // 192-row block tiles (variant 6, or by default when they need fewer row-rounds of the chip)
  if ((variant & 15) == 6 || pp_rows == 192) {
    g_last_kernel.store(6, std::memory_order_relaxed);
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false, true, false, 3>, lds);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 0, false, true, false, 3>, lds);
    }); }
    const int tm192 = (M + 191) / 192;
    const dim3 grid192(dtf_pp_grid(tm192 * tiles_n));
    if (out_f32)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<true, 0, false, true, false, 3>), grid192, block, lds,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tm192, tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 0, false, true, false, 3>), grid192, block, lds,
                         stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc,
                         tm192, tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  // 64x64 tiles with request-issuing waves (variant 12, or by default when every such tile has
  // a CU of its own)
  if ((variant & 15) == 12 ||
      (variant == 0 && dtf_pp_grid(dtf_t64_tiles(M, N)) == dtf_t64_tiles(M, N))) {
    dtf_launch_t64ws<false>(A, Bt, C, M, N, K, lda, ldb, ldc, alpha, beta, out_f32, 1, 0L, stream);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  // 128x128 tiles (variants 5 and 11, or by default when 256x256 tiles would leave CUs idle):
  // with request-issuing waves when every tile has a CU of its own (variant 11), else the
  // four-wave form, two workgroups per CU (variant 5)
  if ((variant & 15) == 5 || (variant & 15) == 11 ||
      (variant == 0 && tiles_m * tiles_n < DTF_T128_BELOW_TILES)) {
    const size_t t_lds = 2 * T128_BUF_BYTES, t_lds3 = 3 * T128_BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_bf16_nt_t128_kernel<true>, t_lds);
      dtf_allow_lds(gemm_bf16_nt_t128_kernel<false>, t_lds);
      dtf_allow_lds(gemm_bf16_nt_t128ws_kernel<true>, t_lds3);
      dtf_allow_lds(gemm_bf16_nt_t128ws_kernel<false>, t_lds3);
    }); }
    const int tm128 = (M + T128 - 1) / T128, tn128 = (N + T128 - 1) / T128;
    const dim3 t_grid(tm128 * tn128), t_block(T128_THREADS);
    if ((variant & 15) == 11 || (variant == 0 && dtf_pp_grid(tm128 * tn128) == tm128 * tn128)) {
      g_last_kernel.store(11, std::memory_order_relaxed);
      if (out_f32)
        hipLaunchKernelGGL(gemm_bf16_nt_t128ws_kernel<true>, t_grid, dim3(2 * T128_THREADS),
                           t_lds3, stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb,
                           ldc, tm128, tn128, alpha, beta, 1, 0L);
      else
        hipLaunchKernelGGL(gemm_bf16_nt_t128ws_kernel<false>, t_grid, dim3(2 * T128_THREADS),
                           t_lds3, stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb,
                           ldc, tm128, tn128, alpha, beta, 1, 0L);
      DTF_LAUNCH_CHECK();
      return DTF_OK;
    }
    g_last_kernel.store(5, std::memory_order_relaxed);
    if (out_f32)
      hipLaunchKernelGGL(gemm_bf16_nt_t128_kernel<true>, t_grid, t_block, t_lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tm128,
                         tn128, alpha, beta, 1, 0L);
    else
      hipLaunchKernelGGL(gemm_bf16_nt_t128_kernel<false>, t_grid, t_block, t_lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tm128,
                         tn128, alpha, beta, 1, 0L);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if ((variant & 15) == 4 || variant == 0) {       // ping-pong with two-quadrant phases
    g_last_kernel.store(4, std::memory_order_relaxed);
        { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false, true>, lds);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 0, false, true>, lds);
    }); }
    if (out_f32)
      hipLaunchKernelGGL((gemm_nt_pp_kernel<true, 0, false, true>), pp_grid, block, lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    else
      hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 0, false, true>), pp_grid, block, lds, stream,
                         (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                         tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if ((variant & 15) == 1 && beta != 0.f) return DTF_ERR_UNSUPPORTED;
  if ((variant & 15) != 1) {
#define DTF_PP_LAUNCH(F32, DBGV)                                                                  \
  hipLaunchKernelGGL((gemm_nt_pp_kernel<F32, DBGV, false>), pp_grid, block, lds, stream,             \
                     (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,         \
                     tiles_n, alpha, beta, 1, 0L, nullptr, nullptr)
    const int dbg = variant >> 4;
    if (dbg == 0) {
      if (out_f32) DTF_PP_LAUNCH(true, 0); else DTF_PP_LAUNCH(false, 0);
    }
#ifdef DTF_GEMM_EXPERIMENTS
    else if (dbg == 1) DTF_PP_LAUNCH(false, 1);
    else if (dbg == 2) DTF_PP_LAUNCH(false, 2);
    else if (dbg == 4) DTF_PP_LAUNCH(false, 4);
#endif
    else return DTF_ERR_UNSUPPORTED;
#undef DTF_PP_LAUNCH
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if (out_f32)
    hipLaunchKernelGGL(gemm_bf16_nt_kernel<true>, grid, block, lds, stream,
                       (const unsigned short*)A, (const unsigned short*)Bt, C, M, N, K, lda, ldb,
                       ldc, tiles_m, tiles_n, alpha);
  else
    hipLaunchKernelGGL(gemm_bf16_nt_kernel<false>, grid, block, lds, stream,
                       (const unsigned short*)A, (const unsigned short*)Bt, C, M, N, K, lda, ldb,
                       ldc, tiles_m, tiles_n, alpha);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// dtf_gemm_bf16_nt whose epilogue also stores every result vector into `nmirrors` (1..7) further
// copies of C: `mirrors` is a table IN DEVICE MEMORY of their addresses (same ldc, same alignment
// as C; typically the same rows of the peers' result buffers, mapped through HIP IPC). Always the
// ping-pong kernel (256- or 192-row tiles by the same rule as the default launch), unsplit.
DTF_API int dtf_gemm_bf16_nt_mirrored(const void* A, const void* Bt, void* C,
                                      const unsigned long long* mirrors, int nmirrors, int M,
                                      int N, int K, long lda, long ldb, long ldc, float alpha,
                                      int out_f32, hipStream_t stream) {
  if (!A || !Bt || !C || !mirrors || nmirrors < 1 || nmirrors > 7 || M <= 0 || N <= 0 || K <= 0)
    return DTF_ERR_BAD_ARG;
  if (lda < K || ldb < K || ldc < N) return DTF_ERR_BAD_ARG;
  if (K % BK || lda % 8 || ldb % 8 || ldc % 8 || ((uintptr_t)A & 15) || ((uintptr_t)Bt & 15) ||
      ((uintptr_t)C & 15) || ((uintptr_t)mirrors & 7))
    return DTF_ERR_UNSUPPORTED;
  if (lda * 2 * BM >= (1L << 32) || ldb * 2 * BN >= (1L << 32)) return DTF_ERR_UNSUPPORTED;
  const int tiles_n = (N + BN - 1) / BN;
  const size_t lds = 2 * BUF_BYTES;
  { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false, true>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 0, false, true>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false, true, false, 3>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 0, false, true, false, 3>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false, true, false, 2>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 0, false, true, false, 2>, lds);
  }); }
  const dim3 block(NTHREADS);
#define DTF_MIRRORED(F32, MIV, TM)                                                               \
  hipLaunchKernelGGL((gemm_nt_pp_kernel<F32, 0, false, true, false, MIV>),                       \
                     dim3(dtf_pp_grid((TM) * tiles_n)), block, lds, stream, (const char*)A,      \
                     (const char*)Bt, C, M, N, K, lda, ldb, ldc, (TM), tiles_n, alpha, 0.f, 1,   \
                     0L, nullptr, nullptr, 0, 0, mirrors, nmirrors)
  const int pp_rows = dtf_pp_rows(M, N);
  if (pp_rows == 192) {
    const int tm = (M + 191) / 192;
    if (out_f32) DTF_MIRRORED(true, 3, tm); else DTF_MIRRORED(false, 3, tm);
  } else if (pp_rows == 128) {
    const int tm = (M + 127) / 128;
    if (out_f32) DTF_MIRRORED(true, 2, tm); else DTF_MIRRORED(false, 2, tm);
  } else {
    const int tm = (M + BM - 1) / BM;
    if (out_f32) DTF_MIRRORED(true, 4, tm); else DTF_MIRRORED(false, 4, tm);
  }
#undef DTF_MIRRORED
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// ---- split-K for small tile grids -----------------------------------------------------------
namespace {

// C[r, c] = alpha * sum_s partial[s][r][c] (+ beta * C[r, c]); planes are M x N fp32, summed in
// split order; four columns per thread (N % 4 == 0)
template <bool OUT_F32>
__global__ __launch_bounds__(256) void planes_reduce_kernel(const float* __restrict__ partial,
                                                            int planes, long plane, void* C,
                                                            long ldc, int M, int N, float alpha,
                                                            float beta) {
  const long n4 = plane >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < planes; s0 += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sp = s0 + u < planes ? s0 + u : planes - 1;
        t[u] = reinterpret_cast<const f32x4*>(partial + sp * plane)[i];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + u < planes) v += t[u];
    }
    v *= alpha;
    const long e = i << 2;
    const int row = (int)(e / N), col = (int)(e % N);
    if (OUT_F32) {
      float* dst = reinterpret_cast<float*>(C) + (long)row * ldc + col;
      if (beta != 0.f) v += *reinterpret_cast<const f32x4*>(dst) * beta;
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      unsigned short* dst = reinterpret_cast<unsigned short*>(C) + (long)row * ldc + col;
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = beta != 0.f ? v[r] + beta * dtf_bf16_to_f32(dst[r]) : v[r];
        o[r] = (short)dtf_f32_to_bf16(x);
      }
      *reinterpret_cast<bf16x4*>(dst) = o;
    }
  }
}

}  // namespace

// Number of K splits the fast bf16 kernel should use for an M x N x K product (1 = none): a
// grid of 256x256 tiles that leaves most CUs without a workgroup is cut along K until there
// is about one work item per CU, each split keeping at least 512 columns of K. Only for grids
// of at most 32 tiles: the planes are fp32, and from 64 tiles on (2048^2 outputs) reading them
// back costs what the extra CUs gain (measured: 2048^3 44.6 us split four ways, 44.2 us unsplit).
// From 33 to 128 tiles the product normally runs on 128x128 tiles, whose operand traffic (1 GiB
// from L2 for 2048 x 2048 x 8192) is what bounds it. With a long K the 256x256 tiles split so
// that tiles x splits is about one work item per CU move half of that and win in spite of the
// planes (2048 x 2048 x 8192: 74 us split four ways, 89 us on 128x128 tiles, torch.matmul 74;
// 2048 x 4096 x 8192: 120 / 140 / 137; 2048 x 2048 x 16384: 119 / 160 / 126). Below K = 6144
// (four ways) or 8192 (two ways) the planes cost more than the traffic saved
// (`profiles/gemm_bf16_splitk_scan.jsonl`).
// Whether a K split of this product runs its work items on 128x128 tiles (env DTF_BF16_SPLIT_T128
// = 0 / 1 overrides, for measurements)
// on: grids of at most 32 tiles of 256x256. GPU time by kernel trace, best split on 128x128
// tiles / best split on 256x256 tiles / torch.matmul: 1024 x 1024 x 4096 23.0 / 30.4 / 18.7 us,
// 512 x 512 x 4096 16.6 / 25.3 / 10.9, 256 x 256 x 8192 14.7 / 23.7 / 10.2, 1024 x 2048 x 4096 31.2 /
// 38.1, 2048 x 1024 x 4096 30.7 / 38.2, 1024 x 1024 x 8192 33.2 / 38.4
// (`scripts/bf16_split_t128_trace.py`).
static bool dtf_gemm_bf16_split_on_t128(int M, int N, int K, int ksplits) {
  if (const char* e = getenv("DTF_BF16_SPLIT_T128")) return atoi(e) != 0;
  (void)K; (void)ksplits;
  return (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) <= 32;
}

// K splits of a product of at most 256 tiles of 64 x 64 whose K is `nkt` K-tiles long: work
// items on the 64x64-tile kernel, at most one per CU, each of at least eight K-tiles -- and only
// if the K-tiles a workgroup is spared (0.23 us each) outweigh the second launch and the pass
// over the fp32 planes (about 5 us on 512^2 outputs, 6-7 above). GPU time by kernel trace,
// unsplit / best split / torch.matmul, bf16: 512 x 512 x 4096 17.5 / 12.5 (4) / 10.6, 256 x 256 x
// 8192 22.7 / 9.2 (8) / 10.2, 512 x 1024 x 2048 8.8 / 11.5 (2) / 10.0, 512 x 1024 x 4096 14.2 / 14.8
// (2) / 14.5 and 640 x 640 x 4096 13.8 / 14.3 (2) / 13.9 with paired K-tiles; 1024^2 outputs have no split
// with a CU per work item: 1024 x 1024 x 4096 18.5 (22.4 split four ways on 128x128 tiles) / 18.2,
// 1024 x 1024 x 8192 30.5 (32.4) / 29.5 (`profiles/gemm_bf16_ws_plan.txt`).
static int dtf_t64_splits(int t64, int nkt) {
  const int cus = dtf_pp_grid(1 << 20);
  int s = 1;
  while (s < 16 && t64 * s * 2 <= cus && nkt % (s * 2) == 0 && nkt / (s * 2) >= 8) s *= 2;
  if (s > 1 && nkt * (s - 1) / s <= (t64 <= 64 ? 24 : 32)) s = 1;
  return s;
}

DTF_API int dtf_gemm_bf16_splits(int M, int N, int K) {
  if (M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % BK == 0 &&
      dtf_pp_grid(dtf_t64_tiles(M, N)) == dtf_t64_tiles(M, N)) {
    const int s = dtf_t64_splits(dtf_t64_tiles(M, N), K / BK);
    if (s > 1 || K / BK < 192) return s;     // (very long K: the rule of the 128x128 tiles below)
  }
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles > 32 && tiles <= 128) {
    if (N % 4 != 0) return 1;
    if (dtf_pp_rows(M, N) != 256) return 1;      // a 192- or 128-row tiling fills the chip unsplit
    // a CU for every 128x128 tile: the kernel with request-issuing waves, unsplit (2048 x 2048 x
    // 8192 58 us against 70 split on 256x256 tiles, 1280 x 2048 x 8192 52 / 60, 1536 x 2048 x
    // 6144 43 / 49)
    const int t128 = ((M + T128 - 1) / T128) * ((N + T128 - 1) / T128);
    if (dtf_pp_grid(t128) == t128) return 1;
    // the most splits that still leave at most one work item per CU and divide K into whole
    // K-tiles; worth it from 144 work items on (1280 x 2048 x 8192, 40 tiles: 54 us split four
    // ways, 79 us unsplit)
    for (int s = (int)(256 / tiles); s >= 2; --s) {
      if (K % (s * BK) != 0) continue;
      if (tiles * s >= 144 && K >= (s >= 3 ? 6144 : 8192)) return s;
      break;
    }
    return 1;
  }
  // Small grids run on 128x128 tiles with request-issuing waves (0.4 us per K-tile where the
  // four-wave form took 0.6), unsplit while K is short: a split costs a second launch and a pass
  // over fp32 planes, 9-12 us on 1024^2 outputs and about 6 on 512^2. From K = 3072 on (2048 for
  // at most 16 tiles) they are split until the work items number about one per CU, each keeping
  // at least eight K-tiles. GPU time by kernel trace, unsplit / 2 / 4 / 8 / 16 splits:
  // 1024 x 1024 x 2048 15.6 / 16.9 / 17.6 / 21.7 us, 1024 x 1024 x 4096 26.3 / 22.5 / 22.0 / 26.6 /
  // 39.3, 512 x 512 x 4096 24.4 / 19.7 / 15.8 / 15.1 / 16.6, 512 x 512 x 2048 14.1 / 14.7 / 13.0 / 13.5,
  // 256 x 256 x 8192 45.0 / 30.1 / 19.8 / 15.1 / 13.3, 1024 x 2048 x 4096 28.4 / 26.6 / 31.3,
  // 1024 x 2048 x 2048 17.2 / 19.8 / 24.9 (`scripts/bf16_ws_plan_trace.sh`)
  if (tiles > 32) return 1;
  const long t128 = (long)((M + T128 - 1) / T128) * ((N + T128 - 1) / T128);
  if (K < (t128 <= 16 ? 2048 : 3072)) return 1;
  int splits = 1;
  while (splits < 16 && t128 * splits * 2 <= 256 && K % (splits * 2 * BK) == 0 &&
         K / (splits * 2) >= 512)
    splits *= 2;
  return splits;
}

// Split-K form of dtf_gemm_bf16_nt: `workspace` holds ksplits * M * N floats. Same operand
// contract; additionally N % 4 == 0 and K % (ksplits * 64) == 0.
DTF_API int dtf_gemm_bf16_nt_splitk(const void* A, const void* Bt, void* C, float* workspace,
                                    int M, int N, int K, long lda, long ldb, long ldc,
                                    float alpha, float beta, int out_f32, int ksplits,
                                    hipStream_t stream) {
  if (!A || !Bt || !C || !workspace || M <= 0 || N <= 0 || K <= 0 || ksplits < 2 || ksplits > 64)
    return DTF_ERR_BAD_ARG;
  if (lda < K || ldb < K || ldc < N) return DTF_ERR_BAD_ARG;
  if (K % (ksplits * BK) || N % 4 || lda % 8 || ldb % 8 || ldc % 4 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bt & 15) || ((uintptr_t)C & (out_f32 ? 15 : 7)) ||
      ((uintptr_t)workspace & 15))
    return DTF_ERR_UNSUPPORTED;
  if (lda * 2 * BM >= (1L << 32) || ldb * 2 * BN >= (1L << 32)) return DTF_ERR_UNSUPPORTED;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = 2 * BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_nt_pp_kernel<true, 0, false, true>, lds);
  }); }
  const long plane = (long)M * N;
  if (dtf_gemm_bf16_split_on_t128(M, N, K, ksplits)) {
    // the split work items on 128x128 tiles: four times as many, each a quarter of the size
    const size_t t_lds = 2 * T128_BUF_BYTES, t_lds3 = 3 * T128_BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_bf16_nt_t128_kernel<true>, t_lds);
      dtf_allow_lds(gemm_bf16_nt_t128ws_kernel<true>, t_lds3);
    }); }
    const int tm128 = (M + T128 - 1) / T128, tn128 = (N + T128 - 1) / T128;
    const int items = tm128 * tn128 * ksplits;
    const int items64 = dtf_t64_tiles(M, N) * ksplits;
    if (dtf_pp_grid(items64) == items64)    // a CU for every 64x64 work item
      dtf_launch_t64ws<false>(A, Bt, workspace, M, N, K, lda, ldb, (long)N, 1.0f, 0.0f, 1, ksplits,
                              plane, stream);
    else if (dtf_pp_grid(items) == items)   // a CU for every work item: request-issuing waves
      hipLaunchKernelGGL(gemm_bf16_nt_t128ws_kernel<true>, dim3(items), dim3(2 * T128_THREADS),
                         t_lds3, stream, (const char*)A, (const char*)Bt, (void*)workspace, M, N,
                         K, lda, ldb, (long)N, tm128, tn128, 1.0f, 0.0f, ksplits, plane);
    else
      hipLaunchKernelGGL(gemm_bf16_nt_t128_kernel<true>, dim3(items), dim3(T128_THREADS), t_lds,
                         stream, (const char*)A, (const char*)Bt, (void*)workspace, M, N, K, lda,
                         ldb, (long)N, tm128, tn128, 1.0f, 0.0f, ksplits, plane);
    const long work = plane / 4;
    const int blocks = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
    if (out_f32)
      hipLaunchKernelGGL(planes_reduce_kernel<true>, dim3(blocks), dim3(256), 0, stream,
                         workspace, ksplits, plane, C, ldc, M, N, alpha, beta);
    else
      hipLaunchKernelGGL(planes_reduce_kernel<false>, dim3(blocks), dim3(256), 0, stream,
                         workspace, ksplits, plane, C, ldc, M, N, alpha, beta);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  const dim3 grid(dtf_pp_grid(tiles_m * tiles_n * ksplits)), block(NTHREADS);
  hipLaunchKernelGGL((gemm_nt_pp_kernel<true, 0, false, true>), grid, block, lds, stream,
                     (const char*)A, (const char*)Bt, (void*)workspace, M, N, K, lda, ldb,
                     (long)N, tiles_m, tiles_n, 1.0f, 0.0f, ksplits, plane, nullptr, nullptr);
  const long work = plane / 4;
  const int blocks = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
  if (out_f32)
    hipLaunchKernelGGL(planes_reduce_kernel<true>, dim3(blocks), dim3(256), 0, stream, workspace,
                       ksplits, plane, C, ldc, M, N, alpha, beta);
  else
    hipLaunchKernelGGL(planes_reduce_kernel<false>, dim3(blocks), dim3(256), 0, stream,
                       workspace, ksplits, plane, C, ldc, M, N, alpha, beta);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// fp8 (OCP e4m3fn) operands: C[M,N] (bf16 or fp32) = alpha * A[M,K] * Bt[N,K]^T, fp32
// accumulate; alpha carries the product of the operands' per-tensor dequantisation scales.
// Fast kernel contract: K % 128 == 0, lda/ldb % 16 == 0, 16-byte aligned A/Bt, C as for bf16.
DTF_API int dtf_gemm_fp8_nt(const void* A, const void* Bt, void* C, int M, int N, int K, long lda,
                            long ldb, long ldc, float alpha, float beta, int out_f32,
                            hipStream_t stream) {
  if (!A || !Bt || !C || M <= 0 || N <= 0 || K <= 0) return DTF_ERR_BAD_ARG;
  if (lda < K || ldb < K || ldc < N) return DTF_ERR_BAD_ARG;
  if (K % 128 || lda % 16 || ldb % 16 || ldc % 4 || ((uintptr_t)A & 15) || ((uintptr_t)Bt & 15) ||
      ((uintptr_t)C & (out_f32 ? 15 : 7)))
    return DTF_ERR_UNSUPPORTED;
  if (lda * BM >= (1L << 32) || ldb * BN >= (1L << 32)) return DTF_ERR_UNSUPPORTED;
  g_last_kernel.store(4, std::memory_order_relaxed);     // (overwritten by the branches below)
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = 2 * BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_nt_pp_kernel<true, 0, true, true>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 0, true, true>, lds);
  }); }
  const dim3 grid(tiles_m * tiles_n), block(NTHREADS);     // one tile per workgroup (no tile loop)
  // a CU for every 128x128 tile: the kernel with request-issuing waves (2048^3 fp8: see below)
  const int tm128 = (M + T128 - 1) / T128, tn128 = (N + T128 - 1) / T128;
  if (dtf_pp_grid(dtf_t64_tiles(M, N)) == dtf_t64_tiles(M, N)) {
    dtf_launch_t64ws<true>(A, Bt, C, M, N, K, lda, ldb, ldc, alpha, beta, out_f32, 1, 0L, stream);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  if (dtf_pp_grid(tm128 * tn128) == tm128 * tn128) {
    g_last_kernel.store(11, std::memory_order_relaxed);
    const size_t t_lds3 = 3 * T128_BUF_BYTES;
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_fp8_nt_t128ws_kernel<true>, t_lds3);
      dtf_allow_lds(gemm_fp8_nt_t128ws_kernel<false>, t_lds3);
    }); }
    if (out_f32)
      hipLaunchKernelGGL(gemm_fp8_nt_t128ws_kernel<true>, dim3(tm128 * tn128),
                         dim3(2 * T128_THREADS), t_lds3, stream, (const char*)A, (const char*)Bt,
                         C, M, N, K, lda, ldb, ldc, tm128, tn128, alpha, beta, 1, 0L);
    else
      hipLaunchKernelGGL(gemm_fp8_nt_t128ws_kernel<false>, dim3(tm128 * tn128),
                         dim3(2 * T128_THREADS), t_lds3, stream, (const char*)A, (const char*)Bt,
                         C, M, N, K, lda, ldb, ldc, tm128, tn128, alpha, beta, 1, 0L);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  // 128x256 tiles with request-issuing waves
  // (2560^3 1550 against 1354 TFLOP/s on the eight-wave tilings, torch._scaled_mm 1423;
  // 2560 x 2560 x 8192 2080 / 1911 / 1887; 1280 x 4096 x 4096 1637 / 1419 / 1539)
  // In fp8 it is ahead with every CU busy as well (2048 x 4096 x 4096 2136 against 1926,
  // torch._scaled_mm 1628; 2048 x 4096 x 8192 2543 / 2268 / 2048): any grid of at most one such
  // tile per CU.
  const int tw_tiles = ((M + TW_M - 1) / TW_M) * ((N + TW_N - 1) / TW_N);
  bool tw_ok = dtf_pp_grid(tw_tiles) == tw_tiles;
#ifdef DTF_GEMM_EXPERIMENTS
  if (getenv("DTF_FP8_NO_TW")) tw_ok = false;
#endif
  if (tw_ok) {
    dtf_launch_t128x256ws<true>(A, Bt, C, M, N, K, lda, ldb, ldc, alpha, beta, out_f32, stream);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
  // 192- or 128-row block tiles where they fill the chip better (the rule of the bf16 launch)
  int pp_rows = dtf_pp_rows(M, N);
#ifdef DTF_GEMM_EXPERIMENTS
  if (getenv("DTF_FP8_FORCE_256")) pp_rows = 256;
#endif
  if (pp_rows != 256) {
    g_last_kernel.store(pp_rows == 192 ? 6 : 9, std::memory_order_relaxed);
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
      dtf_allow_lds(gemm_nt_pp_kernel<true, 0, true, true, false, 3>, lds);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 0, true, true, false, 3>, lds);
      dtf_allow_lds(gemm_nt_pp_kernel<true, 0, true, true, false, 2>, lds);
      dtf_allow_lds(gemm_nt_pp_kernel<false, 0, true, true, false, 2>, lds);
    }); }
    const int tmr = (M + pp_rows - 1) / pp_rows;
    const dim3 grid_r(tmr * tiles_n);
#define DTF_FP8_ROWS(F32, MIV)                                                                   \
  hipLaunchKernelGGL((gemm_nt_pp_kernel<F32, 0, true, true, false, MIV>), grid_r, block, lds,    \
                     stream, (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tmr,    \
                     tiles_n, alpha, beta, 1, 0L, nullptr, nullptr)
    if (pp_rows == 192) { if (out_f32) DTF_FP8_ROWS(true, 3); else DTF_FP8_ROWS(false, 3); }
    else { if (out_f32) DTF_FP8_ROWS(true, 2); else DTF_FP8_ROWS(false, 2); }
#undef DTF_FP8_ROWS
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
#ifdef DTF_GEMM_EXPERIMENTS
  if (const char* tw = getenv("DTF_FP8_TWEAK"); tw && !out_f32) {
    const int bits = atoi(tw);
#define DTF_FP8_TWEAK_CASE(D)                                                                   \
  if (bits == D) {                                                                              \
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {                         \
      dtf_allow_lds(gemm_nt_pp_kernel<false, D, true, true>, lds); }); }                        \
    hipLaunchKernelGGL((gemm_nt_pp_kernel<false, D, true, true>), grid, block, lds, stream,     \
                       (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,     \
                       tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);                          \
    DTF_LAUNCH_CHECK();                                                                         \
    return DTF_OK;                                                                              \
  }
    DTF_FP8_TWEAK_CASE(256) DTF_FP8_TWEAK_CASE(2048) DTF_FP8_TWEAK_CASE(2304)
#undef DTF_FP8_TWEAK_CASE
  }
  if (getenv("DTF_PP_CHUNK_WALK") && !out_f32) {
    { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] { dtf_allow_lds(gemm_nt_pp_kernel<false, 16, true, true>, lds); }); }
    hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 16, true, true>), grid, block, lds, stream,
                       (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                       tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
    DTF_LAUNCH_CHECK();
    return DTF_OK;
  }
#endif
  if (out_f32)
    hipLaunchKernelGGL((gemm_nt_pp_kernel<true, 0, true, true>), grid, block, lds, stream,
                       (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                       tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
  else
    hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 0, true, true>), grid, block, lds, stream,
                       (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                       tiles_n, alpha, beta, 1, 0L, nullptr, nullptr);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// K splits of an fp8 product (1 = unsplit): the rule of the small bf16 grids in K-tiles of 128
// elements -- at most 32 tiles of 256 x 256, at least 48 K-tiles (32 for at most 16 tiles of
// 128 x 128), work items of at least eight K-tiles, about one per CU.
DTF_API int dtf_gemm_fp8_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 128 || N % 4) return 1;
  if (dtf_pp_grid(dtf_t64_tiles(M, N)) == dtf_t64_tiles(M, N)) {
    const int s = dtf_t64_splits(dtf_t64_tiles(M, N), K / 128);
    if (s > 1 || K / 128 < 192) return s;
  }
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles > 32) return 1;
  const long t128 = (long)((M + T128 - 1) / T128) * ((N + T128 - 1) / T128);
  const int nkt = K / 128;
  if (nkt < (t128 <= 16 ? 32 : 48)) return 1;
  int splits = 1;
  while (splits < 16 && t128 * splits * 2 <= 256 && nkt % (splits * 2) == 0 &&
         nkt / (splits * 2) >= 8)
    splits *= 2;
  return splits;
}

// Split-K form of dtf_gemm_fp8_nt: `workspace` holds ksplits * M * N floats; (split, tile) work
// items on the 128x128-tile kernel with request-issuing waves write fp32 planes, a second kernel
// sums them in split order and applies alpha / beta. Same operand contract; additionally
// N % 4 == 0, K % (ksplits * 128) == 0 and a CU for every work item (DTF_ERR_UNSUPPORTED
// otherwise: the caller runs the unsplit launch).
DTF_API int dtf_gemm_fp8_nt_splitk(const void* A, const void* Bt, void* C, float* workspace, int M,
                                   int N, int K, long lda, long ldb, long ldc, float alpha,
                                   float beta, int out_f32, int ksplits, hipStream_t stream) {
  if (!A || !Bt || !C || !workspace || M <= 0 || N <= 0 || K <= 0 || ksplits < 2 || ksplits > 64)
    return DTF_ERR_BAD_ARG;
  if (lda < K || ldb < K || ldc < N) return DTF_ERR_BAD_ARG;
  if (K % (ksplits * 128) || N % 4 || lda % 16 || ldb % 16 || ldc % 4 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bt & 15) || ((uintptr_t)C & (out_f32 ? 15 : 7)) || ((uintptr_t)workspace & 15))
    return DTF_ERR_UNSUPPORTED;
  if (lda * BM >= (1L << 32) || ldb * BN >= (1L << 32)) return DTF_ERR_UNSUPPORTED;
  const int tm128 = (M + T128 - 1) / T128, tn128 = (N + T128 - 1) / T128;
  const int items = tm128 * tn128 * ksplits;
  if (dtf_pp_grid(items) != items) return DTF_ERR_UNSUPPORTED;
  const size_t t_lds3 = 3 * T128_BUF_BYTES;
  { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_fp8_nt_t128ws_kernel<true>, t_lds3);
  }); }
  const long plane = (long)M * N;
  const int items64 = dtf_t64_tiles(M, N) * ksplits;
  if (dtf_pp_grid(items64) == items64)
    dtf_launch_t64ws<true>(A, Bt, workspace, M, N, K, lda, ldb, (long)N, 1.0f, 0.0f, 1, ksplits,
                           plane, stream);
  else
  hipLaunchKernelGGL(gemm_fp8_nt_t128ws_kernel<true>, dim3(items), dim3(2 * T128_THREADS), t_lds3,
                     stream, (const char*)A, (const char*)Bt, (void*)workspace, M, N, K, lda, ldb,
                     (long)N, tm128, tn128, 1.0f, 0.0f, ksplits, plane);
  const long work = plane / 4;
  const int blocks = (int)((work + 255) / 256 > 2048 ? 2048 : (work + 255) / 256);
  if (out_f32)
    hipLaunchKernelGGL(planes_reduce_kernel<true>, dim3(blocks), dim3(256), 0, stream, workspace,
                       ksplits, plane, C, ldc, M, N, alpha, beta);
  else
    hipLaunchKernelGGL(planes_reduce_kernel<false>, dim3(blocks), dim3(256), 0, stream, workspace,
                       ksplits, plane, C, ldc, M, N, alpha, beta);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// Block-scaled fp8 (OCP MX: e4m3 elements, one E8M0 scale per 32 consecutive k of a row), both
// operands quantised by dtf_quantize_mx_fp8 (which also lays the scales out the way the kernel
// reads them: sa with role 0 for A, sb with role 1 for Bt):
// C[M,N] (bf16 or fp32) = alpha * (A o SA)[M,K] * (Bt o SB)[N,K]^T, fp32 accumulate.
// Contract: M % 256 == N % 256 == 0, K % 128 == 0, lda/ldb % 16 == 0, 16-byte aligned A/Bt.
DTF_API int dtf_gemm_fp8_mx_nt(const void* A, const void* Bt, void* C, const void* sa,
                               const void* sb, int M, int N, int K, long lda, long ldb, long ldc,
                               long sa_row_tiles, long sb_row_tiles, float alpha, float beta,
                               int out_f32, hipStream_t stream) {
  if (!A || !Bt || !C || !sa || !sb || M <= 0 || N <= 0 || K <= 0) return DTF_ERR_BAD_ARG;
  if (lda < K || ldb < K || ldc < N) return DTF_ERR_BAD_ARG;
  if (M % BM || N % BN || K % 128 || lda % 16 || ldb % 16 || ldc % 4 || ((uintptr_t)A & 15) ||
      ((uintptr_t)Bt & 15) || ((uintptr_t)sa & 15) || ((uintptr_t)sb & 3) ||
      ((uintptr_t)C & (out_f32 ? 15 : 7)))
    return DTF_ERR_UNSUPPORTED;
  if (lda * BM >= (1L << 32) || ldb * BN >= (1L << 32)) return DTF_ERR_UNSUPPORTED;
  const int tiles_m = M / BM, tiles_n = N / BN;
  // the scales may belong to a taller matrix of which A / Bt is a block of whole row tiles
  if (sa_row_tiles < tiles_m || sb_row_tiles < tiles_n) return DTF_ERR_BAD_ARG;
  if ((long)(K / 128 + 3) * (sa_row_tiles > sb_row_tiles ? sa_row_tiles : sb_row_tiles) * 1024 >=
      (1L << 31))
    return DTF_ERR_UNSUPPORTED;
  const size_t lds = 2 * BUF_BYTES + (NTHREADS / 64) * 4096;      // all 160 KiB of the CU
  { static DtfPerDeviceOnce once__; dtf_once_per_device(once__, [&] {
    dtf_allow_lds(gemm_nt_pp_kernel<true, 0, true, true, true>, lds);
    dtf_allow_lds(gemm_nt_pp_kernel<false, 0, true, true, true>, lds);
  }); }
  const dim3 grid(tiles_m * tiles_n), block(NTHREADS);
  const int sa_pair = (int)(sa_row_tiles * 2048), sb_quad = (int)(sb_row_tiles * 4096);
  if (out_f32)
    hipLaunchKernelGGL((gemm_nt_pp_kernel<true, 0, true, true, true>), grid, block, lds, stream,
                       (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                       tiles_n, alpha, beta, 1, 0L, (const unsigned*)sa, (const unsigned*)sb,
                       sa_pair, sb_quad);
  else
    hipLaunchKernelGGL((gemm_nt_pp_kernel<false, 0, true, true, true>), grid, block, lds, stream,
                       (const char*)A, (const char*)Bt, C, M, N, K, lda, ldb, ldc, tiles_m,
                       tiles_n, alpha, beta, 1, 0L, (const unsigned*)sa, (const unsigned*)sb,
                       sa_pair, sb_quad);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// Any-shape bf16 GEMM: A(m,k) at A[m*a_rs + k*a_cs], B(k,n) at B[k*b_rs + n*b_cs].
DTF_API int dtf_gemm_bf16_generic(const void* A, const void* B, void* C, int M, int N, int K,
                                  long a_rs, long a_cs, long b_rs, long b_cs, long ldc, float alpha,
                                  float beta, int out_f32, hipStream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || ldc < N) return DTF_ERR_BAD_ARG;
  dim3 grid((N + GB - 1) / GB, (M + GB - 1) / GB), block(256);
  if (out_f32)
    hipLaunchKernelGGL(gemm_bf16_generic_kernel<true>, grid, block, 0, stream,
                       (const unsigned short*)A, (const unsigned short*)B, C, M, N, K, a_rs, a_cs,
                       b_rs, b_cs, ldc, alpha, beta);
  else
    hipLaunchKernelGGL(gemm_bf16_generic_kernel<false>, grid, block, 0, stream,
                       (const unsigned short*)A, (const unsigned short*)B, C, M, N, K, a_rs, a_cs,
                       b_rs, b_cs, ldc, alpha, beta);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// Any-shape fp8 (e4m3) GEMM, same addressing convention as dtf_gemm_bf16_generic.
DTF_API int dtf_gemm_fp8_generic(const void* A, const void* B, void* C, int M, int N, int K,
                                 long a_rs, long a_cs, long b_rs, long b_cs, long ldc, float alpha,
                                 float beta, int out_f32, hipStream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || ldc < N) return DTF_ERR_BAD_ARG;
  dim3 grid((N + GB - 1) / GB, (M + GB - 1) / GB), block(256);
  if (out_f32)
    hipLaunchKernelGGL(gemm_fp8_generic_kernel<true>, grid, block, 0, stream,
                       (const unsigned char*)A, (const unsigned char*)B, C, M, N, K, a_rs, a_cs,
                       b_rs, b_cs, ldc, alpha, beta);
  else
    hipLaunchKernelGGL(gemm_fp8_generic_kernel<false>, grid, block, 0, stream,
                       (const unsigned char*)A, (const unsigned char*)B, C, M, N, K, a_rs, a_cs,
                       b_rs, b_cs, ldc, alpha, beta);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
