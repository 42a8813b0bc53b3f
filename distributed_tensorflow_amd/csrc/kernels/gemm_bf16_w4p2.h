// Four-wave bf16 kernel whose LDS-DMA runs two K-tiles ahead (variant 10 of dtf_gemm_bf16_nt).
// Included by gemm_bf16.hip inside its anonymous namespace (uses its tile constants, chunk_off,
// tile_rsrc and pp_tile_of_block).
//
// One wave per SIMD, 128x128 outputs per wave (8 x 8 MFMA tiles in the 256 AGPRs), the 256x256x64
// block tile and LDS image of the other kernels, two K-tile buffers of 64 KiB. What distinguishes
// it from the four-wave kernel of variant 3 is WHEN a buffer is given back to the DMA: not as a
// whole once per K-tile, but operand by operand, as soon as the last fragment of that operand has
// been read into registers.
//
// Fragments are double-buffered in registers by k-step (2 x 16 x 4 VGPRs). During K-tile t the
// wave issues 128 MFMAs (k-step 0: slots 0-63, k-step 1: slots 64-127), and between them, at
// most one memory instruction per slot:
//   slots   0-14   read Bt fragments of (t, k-step 1)                      -> barrier 1 (slot 20)
//   slots  22-64   DMA the 8 Bt pieces of K-tile t+2 into THIS tile's buffer (its Bt half is dead
//                  behind barrier 1), one every 6 slots; read A fragments of (t, 1)
//                                                                          -> barrier 2 (slot 51)
//   slots  70-126  DMA the 8 A pieces of K-tile t+2 (dead behind barrier 2), one every 8 slots
//   slot   67      s_waitcnt vmcnt(16): the Bt pieces of K-tile t+1 (the oldest 8 of the 24
//                  requests in flight) have landed                         -> barrier 3 (slot 68)
//   slots  69-83   read Bt fragments of (t+1, 0) from the other buffer
//   slot  104      s_waitcnt vmcnt(13): all of K-tile t+1 has landed       -> barrier 4 (slot 105)
//   slots 106-120  read A fragments of (t+1, 0)
// A request is therefore issued 110-170 MFMAs (1800-2700 cycles) before the barrier that needs
// it, against 80-110 for variant 3, and 8 to 16 younger requests stay in flight across every
// wait. The four barriers and two counted waits are those of the vendor library's 256x256x64
// kernel (read off its disassembly); its requests come in bursts (every 4, 3 and 2 slots), here
// they are spread evenly -- 1570 against 1560 TF at 16384^3, b = 2255 against 2303 in the cycle
// counts below, which were taken with the library's spacing (DBG bit 256). The LDS image (XOR-
// swizzled instead of padded: no bank conflicts where the library has 2-way ones), the tile
// walk and the epilogue are this file's.
//
// Persistent over tiles like the ping-pong kernel: the first two K-tiles of the next output tile
// are requested before the epilogue of this one.
//
// NOT the default. Measured beside the eight-wave ping-pong kernel and torch.matmul on one box
// (`profiles/gemm_bf16_w4p2_development.jsonl`, `profiles/gemm_bf16_w4p2_cycles.txt`):
//   16384^3 1566-1580 TF (ping-pong 1583-1600, library 1629-1642), 8192^3 1615-1635 (1617-1644,
//   1673-1684), 4096^3 1530-1536 (1532-1558, 1591-1602): level with the eight-wave kernel, which
//   keeps the default because it also has 192- and 128-row tiles, split-K and the fp8 forms.
//   Variant 3, the same four waves with one barrier per K-tile, runs at 1236.
// In cycles (rocprofv3 GRBM_GUI_ACTIVE, a tile = a + 256 b at K = 16384, from runs at two K):
//   per K-tile b = 2303 (ideal 2048; library 2182, ping-pong 2168), per tile a = 28 thousand
//   (library 15, ping-pong 19). It draws less power than the eight-wave kernel (a third fewer
//   LDS reads) and so runs at a higher clock under the 1400 W cap (1.80-1.82 GHz against
//   1.71-1.74), which is why it is level in TFLOP/s with 6 % more cycles.
//   Where the cycles go: without the K loop's DMA b = 2133, without its barriers 2135, with
//   neither about the same -- DMA and barriers cost nothing alone and 170 cycles per K-tile
//   together. 4-byte requests in place of the 16-byte ones halve that (2221). Dropping the
//   counted waits changes nothing (memory latency is covered). Schedules tried against it, all
//   correct, none faster (`profiles/gemm_bf16_w4p2_schedules.jsonl`): two barriers per K-tile
//   (one wait for the whole K-tile) 1448 TF at 16384^3 against 1515; each wave's requests
//   shifted by its wave id so that the four waves never ask the address unit in the same slot
//   (one copy of the loop per wave) 1473 against 1524 (b = 2430); unpermuted source addresses and
//   the transposed MFMA order no change; the chunk walk 1545 against 1573; non-temporal A
//   requests 1524; every request an L2 hit (K-tile 0 restaged) b = 2228; the three-barrier plan
//   of the library's hand-written kernel (DBG bit 2048) 1559 against 1560; each workgroup
//   starting its K walk at another K-tile and wrapping around (what the library calls
//   StaggerU) 1262-1471 against 1539 -- the CUs of an XCD then no longer share panels in L2.
//   Dropping ONE of the four barriers saves 0.5-0.9 % of the cycles, a pair of them 1.4 %, all four
//   6-7 %: what costs is not a barrier but the lock-step that any remaining barrier keeps the
//   four waves in -- they then ask the LDS (fragment reads, DMA writes) and the address unit in
//   the same cycles. The eight-wave kernel has the same lock-step but a second wave per SIMD
//   to run in the gaps.
// What did pay: issuing the request's M0 write one slot ahead of the load (see DTF_P2_M0: 1528
// -> 1566 TF), and the short epilogue for interior tiles (a from 48 to 28 thousand cycles; also
// taken over by the ping-pong kernel).
//
// DBG (timing experiments, -DDTF_GEMM_EXPERIMENTS only; results are wrong when set), bits:
// 1 = no LDS-DMA in the K loop, 2 = always stage K-tile 0 (every load an L2 hit), 4 = no barriers,
// 8 = no counted vmcnt waits, 16 = unpermuted source chunks, 32 = A fragment in the inner MFMA
// loop (results stay correct), 64 = no fragment reads in the K loop, 128 = 4-byte instead of
// 16-byte requests, 256 = the library's request spacing, 512 = chunk walk, 1024 = non-temporal
// A requests, 2048 = the three-barrier plan, 4096 / 8192 / 16384 / 32768 = without the first /
// second / third / fourth barrier of the K-tile, 65536 = without the two LDS waits
template <int... Is, typename F>
__device__ __forceinline__ void dtf_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}), in order
template <int N, typename F>
__device__ __forceinline__ void dtf_static_for(F&& f) {
  dtf_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// What a slot of the K-tile does next to its MFMA: one fragment read or one synchronisation
// (dtf_p2_action) and, independently, one LDS-DMA request (dtf_p2_dma).
enum : int { P2_NONE, P2_RD_B1, P2_RD_A1, P2_RD_B0N, P2_RD_A0N, P2_LGKM0, P2_BARRIER, P2_VMWAIT };
struct P2Act { int kind, arg; };
// plan 0: requests spread evenly; plan 1: the request spacing of the vendor library's generated
// kernel (four barriers); plan 2: that of its hand-written kernel -- THREE barriers: one wait
// (vmcnt(13), slot 91) and one barrier (slot 92) for the whole of K-tile t+1, behind which all 16
// fragments of (t+1, 0) are read
constexpr P2Act dtf_p2_action(int s, int plan) {
  if (s < 16 && (s & 1) == 0) return {P2_RD_B1, s >> 1};
  if (s == 19 || s == 50) return {P2_LGKM0, 0};
  if (s == 20 || s == 51) return {P2_BARRIER, 0};
  if (plan == 2) {
    if (s >= 24 && s <= 36 && (s - 24) % 3 == 0) return {P2_RD_A1, (s - 24) / 3};      // 0-4
    if (s == 38 || s == 40 || s == 42) return {P2_RD_A1, 5 + ((s - 38) >> 1)};
    if (s == 91) return {P2_VMWAIT, 13};
    if (s == 92) return {P2_BARRIER, 0};
    if (s >= 93 && s <= 107 && (s & 1) == 1) return {P2_RD_B0N, (s - 93) >> 1};
    if (s >= 109 && s <= 123 && (s & 1) == 1) return {P2_RD_A0N, (s - 109) >> 1};
    return {P2_NONE, 0};
  }
  if (s == 68 || s == 105) return {P2_BARRIER, 0};
  if (s >= 24 && s <= 40 && ((s - 24) & 3) == 0) return {P2_RD_A1, (s - 24) >> 2};
  if (s == 42 || s == 44 || s == 46) return {P2_RD_A1, 5 + ((s - 42) >> 1)};
  if (s == 67) return {P2_VMWAIT, plan == 0 ? 16 : 18};
  if (s >= 69 && s <= 83 && (s & 1) == 1) return {P2_RD_B0N, (s - 69) >> 1};
  if (s == 104) return {P2_VMWAIT, plan == 0 ? 13 : 15};
  if (s >= 106 && s <= 120 && (s & 1) == 0) return {P2_RD_A0N, (s - 106) >> 1};
  return {P2_NONE, 0};
}
// LDS-DMA piece (0-7: Bt, 8-15: A) requested in slot s, or -1
constexpr int dtf_p2_dma(int s, int plan) {
  if (plan == 0) {       // Bt at 22, 28, .., 64; A at 70, 78, .., 126
    if (s >= 22 && s <= 64 && (s - 22) % 6 == 0) return (s - 22) / 6;
    if (s >= 70 && s <= 126 && ((s - 70) & 7) == 0) return 8 + ((s - 70) >> 3);
    return -1;
  }
  if (plan == 2) {
    if (s >= 22 && s <= 34 && (s - 22) % 3 == 0) return (s - 22) / 3;         // Bt 0-4
    if (s == 52 || s == 55 || s == 58) return 5 + (s - 52) / 3;               // Bt 5-7
    if (s == 61 || s == 64) return 8 + (s - 61) / 3;                          // A 0-1
    if (s == 85 || s == 87 || s == 89) return 10 + ((s - 85) >> 1);           // A 2-4
    if (s == 96) return 13;
    if (s == 102) return 14;
    if (s == 124) return 15;
    return -1;
  }
  if (s >= 22 && s <= 38 && ((s - 22) & 3) == 0) return (s - 22) >> 2;        // Bt 0-4
  if (s == 52 || s == 55 || s == 58) return 5 + (s - 52) / 3;                 // Bt 5-7
  if (s == 61 || s == 64) return 8 + (s - 61) / 3;                            // A 0-1
  if (s == 85 || s == 87 || s == 89) return 10 + ((s - 85) >> 1);             // A 2-4
  if (s == 97 || s == 101) return 13 + ((s - 97) >> 2);                       // A 5-6
  if (s == 125) return 15;                                                    // A 7
  return -1;
}

template <bool OUT_F32, int DBG = 0>
__global__ __launch_bounds__(W4_THREADS) void gemm_bf16_nt_w4p2_kernel(
    const char* __restrict__ A, const char* __restrict__ Bt, void* __restrict__ C, int M, int N,
    int K, long lda, long ldb, long ldc, int tiles_m, int tiles_n, float alpha, float beta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntiles = tiles_m * tiles_n;
  const int nkt = K / BK;

  int tile = blockIdx.x;
  int m0, n0;
  {
    int tm, tn;
    pp_tile_of_block(tile, tiles_m, tiles_n, tm, tn, (DBG & 512) != 0);
    m0 = tm * BM;
    n0 = tn * BN;
  }

  f32x4 acc[8][8];
  i32x4 fa[2][8], fb[2][8];

  // DMA: this wave fills pieces 8 wave .. 8 wave + 7 (8 rows of 128 B each) of the A tile and of
  // the Bt tile. Address = (tile's first row, SGPR resource) + (K-tile's byte offset, SGPR) +
  // (per-lane: row inside the tile, clamped at the matrix edge, times the leading dimension,
  // plus the swizzled chunk -- loop-invariant, one VGPR per piece).
  unsigned voff_a[8], voff_b[8];
  auto piece_offsets = [&](int l, int tm0, int tn0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int trow = (wave * 8 + j) * 8 + (l >> 3);
      const int kc = (DBG & 16) ? (l & 7) : (l & 7) ^ ((trow >> 1) & 7);
      int ra = tm0 + trow, rb = tn0 + trow;
      ra = ra < M - 1 ? ra : M - 1;
      rb = rb < N - 1 ? rb : N - 1;
      voff_a[j] = (unsigned)(ra - tm0) * (unsigned)(lda * 2) + (unsigned)(kc * 16);
      voff_b[j] = (unsigned)(rb - tn0) * (unsigned)(ldb * 2) + (unsigned)(kc * 16);
    }
  };
  piece_offsets(lane, m0, n0);
  // (the DMA is issued from inline asm, see DTF_P2_DMA: its buffer resource is built by hand --
  // base, stride 0, unbounded, raw dword format)
  auto tile_words = [](const char* base) {
    const unsigned long long p = reinterpret_cast<unsigned long long>(base);
    i32x4 w;
    w[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    w[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((p >> 32) & 0xffffu));
    w[2] = -1;
    w[3] = 0x00020000;
    return w;
  };
  i32x4 a_tile = tile_words(A + (long)m0 * lda * 2);
  i32x4 b_tile = tile_words(Bt + (long)n0 * ldb * 2);

  const int frow = lane & 15, fq = lane >> 4;
  const int off0 = chunk_off(frow, fq), off1 = chunk_off(frow, 4 + fq);
  const int a_rd = wr * 16384;                     // wave's 8 row blocks of 2 KiB in the A tile
  const int b_rd = TILE_BYTES + wc * 16384;        // ... in the Bt tile
  // LDS address of the wave's first piece in the A tile of buffer 0
  const unsigned wr_base = (unsigned)reinterpret_cast<uintptr_t>((lds_void*)smem) + wave * 8192;

// One LDS-DMA request = two instructions: M0 <- the piece's LDS address, then the buffer load.
// Written next to each other the load needs a wait state behind the M0 write and the compiler
// puts an s_nop there; that slot's three instructions then no longer fit the 16 cycles of its
// MFMA and the load -- whose issue takes the wave a few cycles -- pushes the next MFMA back
// (measured: 135 cycles per K-tile, 6 %). So M0 is written one slot EARLIER, in front of the
// MFMA that precedes the load (nothing else in the K loop uses M0), from inline asm; the
// compiler's own wait-count bookkeeping does not see these loads, every wait on them is written
// out below.
// P = piece 0-7: Bt, 8-15: A
#define DTF_P2_M0(P, BUF)                                                                        \
  asm volatile("s_mov_b32 m0, %0" ::"s"(wr_base + (BUF) * BUF_BYTES + ((P) >> 3 ? 0 : TILE_BYTES) + \
                                        ((P) & 7) * 1024)                                        \
               : "memory")
#define DTF_P2_LOAD(P, K0)                                                                       \
  if constexpr ((DBG & 128) != 0)                                                                \
    asm volatile("buffer_load_dword %0, %1, %2 offen lds" ::"v"(voff_a[(P) & 7]), "s"(a_tile),   \
                 "s"(K0)                                                                         \
                 : "memory");                                                                    \
  else if (((P) >> 3) && (DBG & 1024))                                                           \
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen nt lds" ::"v"(voff_a[(P) & 7]),           \
                 "s"(a_tile), "s"(K0)                                                            \
                 : "memory");                                                                    \
  else if ((P) >> 3)                                                                             \
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff_a[(P) & 7]), "s"(a_tile), \
                 "s"(K0)                                                                         \
                 : "memory");                                                                    \
  else                                                                                           \
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff_b[(P) & 7]), "s"(b_tile), \
                 "s"(K0)                                                                         \
                 : "memory")
#define DTF_P2_RD(BUF, BASE, OFF, I)                                                             \
  ((DBG & 64) ? fa[0][0] : *reinterpret_cast<const i32x4*>(smem + (BUF) * BUF_BYTES + (BASE) +   \
                                                           (OFF) + (I) * 2048))

// K-tile t in buffer CUR (the next one in NXT); K2 = byte offset of the K-tile requested now. (The 128 slots are instantiated one by one: a `#pragma unroll` loop of this size
// is past the compiler's full-unroll budget and would leave the accumulators indexed at run
// time.)
#define DTF_P2_TILE(CUR, NXT, K2)                                                                \
  {                                                                                              \
    const int k2_ = (K2);                                                                        \
    dtf_static_for<128>([&](auto slot_) __attribute__((always_inline)) {                         \
      constexpr int s_ = decltype(slot_)::value;                                                 \
      if constexpr ((DBG & 32) != 0)                                                             \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                    \
                     : "+a"(acc[s_ & 7][(s_ >> 3) & 7])                                          \
                     : "v"(fb[s_ >> 6][(s_ >> 3) & 7]), "v"(fa[s_ >> 6][s_ & 7]));               \
      else                                                                                       \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"                                      \
                   : "+a"(acc[(s_ >> 3) & 7][s_ & 7])                                            \
                   : "v"(fb[s_ >> 6][s_ & 7]), "v"(fa[s_ >> 6][(s_ >> 3) & 7]));                 \
      constexpr int spread_ = (DBG & 2048) ? 2 : (DBG & 256) ? 1 : 0;                            \
      constexpr P2Act act_ = dtf_p2_action(s_, spread_);                                         \
      constexpr int dma_ = dtf_p2_dma(s_, spread_);                                              \
      constexpr int dma_next_ = s_ < 127 ? dtf_p2_dma(s_ + 1, spread_) : -1;                     \
      if constexpr (dma_ >= 0) {                                                                 \
        if (!(DBG & 1)) { DTF_P2_LOAD(dma_, k2_); }                                              \
      }                                                                                          \
      if constexpr (act_.kind == P2_RD_B1) {                                                     \
        fb[1][act_.arg] = DTF_P2_RD(CUR, b_rd, off1, act_.arg);                                  \
      } else if constexpr (act_.kind == P2_RD_A1) {                                              \
        fa[1][act_.arg] = DTF_P2_RD(CUR, a_rd, off1, act_.arg);                                  \
      } else if constexpr (act_.kind == P2_RD_B0N) {                                             \
        fb[0][act_.arg] = DTF_P2_RD(NXT, b_rd, off0, act_.arg);                                  \
      } else if constexpr (act_.kind == P2_RD_A0N) {                                             \
        fa[0][act_.arg] = DTF_P2_RD(NXT, a_rd, off0, act_.arg);                                  \
      } else if constexpr (act_.kind == P2_LGKM0) {                                              \
        if (!(DBG & 65536)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
      } else if constexpr (act_.kind == P2_BARRIER) {                                            \
        /* bits 12-15: drop the barrier of slot 20 / 51 / 68 / 105 alone */                      \
        constexpr int which_ = s_ == 20 ? 4096 : s_ == 51 ? 8192 : s_ == 68 ? 16384 : 32768;     \
        if (!(DBG & 4) && !(DBG & which_)) __builtin_amdgcn_s_barrier();                         \
      } else if constexpr (act_.kind == P2_VMWAIT) {                                             \
        if (!(DBG & 9)) {                                                                        \
          if constexpr (act_.arg == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");        \
          else if constexpr (act_.arg == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   \
          else if constexpr (act_.arg == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");   \
          else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");                                 \
        }                                                                                        \
      }                                                                                          \
      /* M0 of the request that the NEXT slot issues */                                          \
      if constexpr (dma_next_ >= 0) {                                                            \
        if (!(DBG & 1)) DTF_P2_M0(dma_next_, CUR);                                               \
      }                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    });                                                                                          \
  }

  // prologue of the first tile: K-tiles 0 and 1 requested, Bt pieces ahead of A pieces (the
  // counted waits of the K loop rely on that order)
#define DTF_P2_PROLOGUE()                                                                        \
  _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) {                                             \
    const int k0_ = (t_ < nkt ? t_ : nkt - 1) * 128;                                         \
    _Pragma("unroll") for (int p_ = 0; p_ < 16; ++p_) {                                          \
      DTF_P2_M0(p_, t_);                                                                         \
      asm volatile("s_nop 0");                                                                   \
      DTF_P2_LOAD(p_, k0_);                                                                      \
    }                                                                                            \
  }
  DTF_P2_PROLOGUE()
  bool first = true;

  for (;;) {
    if (first) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");    // K-tile 0 landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // ... and the previous stores
    first = false;
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fb[0][i] = DTF_P2_RD(0, b_rd, off0, i);
      fa[0][i] = DTF_P2_RD(0, a_rd, off0, i);
    }

    // K-tile t + 2 is requested during K-tile t; past the end the last K-tile is restaged into
    // a dead slot (valid memory, an L2 hit), which keeps the counted waits uniform
    {
      int t = 0;
      for (; t + 1 < nkt; t += 2) {
        const int ka = (DBG & 2) ? 0 : (t + 2 < nkt ? t + 2 : nkt - 1) * 128;
        DTF_P2_TILE(0, 1, ka)
        const int kb = (DBG & 2) ? 0 : (t + 3 < nkt ? t + 3 : nkt - 1) * 128;
        DTF_P2_TILE(1, 0, kb)
      }
      if (t < nkt) {
        const int ka = (DBG & 2) ? 0 : (nkt - 1) * 128;
        DTF_P2_TILE(0, 1, ka)
      }
    }
    // Every wave is past barrier 4 of the last K-tile: nothing meaningful is read from LDS any
    // more (the fragment reads behind that barrier fetched a K-tile that does not exist). A
    // slot is always written by the same wave and a wave's loads land in order, so the next
    // tile's first K-tiles may be requested on top of the restaged ones still in flight.
    const int cm0 = m0, cn0 = n0;
    tile += gridDim.x;
    const bool more = tile < ntiles;
    if (more) {
      int tm, tn;
      pp_tile_of_block(tile, tiles_m, tiles_n, tm, tn, (DBG & 512) != 0);
      m0 = tm * BM;
      n0 = tn * BN;
      int o_lane = lane;
      asm volatile("" : "+v"(o_lane));
      piece_offsets(o_lane, m0, n0);
      a_tile = tile_words(A + (long)m0 * lda * 2);
      b_tile = tile_words(Bt + (long)n0 * ldb * 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of the old image returned
      DTF_P2_PROLOGUE()
    }
    // the last MFMAs were issued from inline asm: cover their result latency before the
    // epilogue reads the accumulators
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

    // epilogue: lane holds C[row = frow][col = 4 fq + r] of each 16x16 tile
    int e_lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(e_lane));
    const int e_frow = e_lane & 15, e_fq = e_lane >> 4;
    const bool wide_bf16 = !OUT_F32 && beta == 0.f && (ldc & 7) == 0 &&
                           (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    // Interior tile, bf16 out: no per-lane bounds, one base address per 16 rows, hardware
    // bf16 converts -- about 20 instructions per pair of 16x16 tiles. (The general path below
    // costs several times that in execution-mask branches alone; with one wave per SIMD the
    // epilogue is not hidden behind anything.)
    if (wide_bf16 && cm0 + BM <= M && cn0 + BN <= N) {
      const int colx = (e_fq & 1) ? 16 + 4 * (e_fq - 1) : 4 * e_fq;
      char* crow = reinterpret_cast<char*>(C) +
                   ((long)(cm0 + wr * 128 + e_frow) * ldc + cn0 + wc * 128 + colx) * 2;
      const long rstep = ldc * 32;                   // 16 rows, in bytes
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
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
        crow += rstep;
      }
    } else
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = cm0 + wr * 128 + i * 16 + e_frow;
      if (row >= M) continue;
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        if (!OUT_F32) {
          // bf16 out: one v_permlane16_swap per register pair trades the odd 16-lane rows'
          // first tile against the even rows' second one, after which every lane owns 8
          // consecutive columns -- one 16-byte store per lane and tile pair
          const int cb = cn0 + wc * 128 + jp * 32;
          if (wide_bf16 && cb + 32 <= N) {
            const f32x4 v0 = acc[i][2 * jp] * alpha, v1 = acc[i][2 * jp + 1] * alpha;
            const unsigned p0a = dtf_f32_to_bf16(v0[0]) | ((unsigned)dtf_f32_to_bf16(v0[1]) << 16);
            const unsigned p0b = dtf_f32_to_bf16(v0[2]) | ((unsigned)dtf_f32_to_bf16(v0[3]) << 16);
            const unsigned p1a = dtf_f32_to_bf16(v1[0]) | ((unsigned)dtf_f32_to_bf16(v1[1]) << 16);
            const unsigned p1b = dtf_f32_to_bf16(v1[2]) | ((unsigned)dtf_f32_to_bf16(v1[3]) << 16);
            const auto sa = __builtin_amdgcn_permlane16_swap(p0a, p1a, false, false);
            const auto sb = __builtin_amdgcn_permlane16_swap(p0b, p1b, false, false);
            const int col = cb + ((e_fq & 1) ? 16 + 4 * (e_fq - 1) : 4 * e_fq);
            u32x4 o = {sa[0], sb[0], sa[1], sb[1]};
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(C) + ((long)row * ldc + col) * 2) =
                o;
            continue;
          }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * jp + jj;
          const int col = cn0 + wc * 128 + j * 16 + 4 * e_fq;
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
    if (!more) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no LDS-DMA may outlive the workgroup
#undef DTF_P2_PROLOGUE
#undef DTF_P2_TILE
#undef DTF_P2_RD
#undef DTF_P2_M0
#undef DTF_P2_LOAD
}
