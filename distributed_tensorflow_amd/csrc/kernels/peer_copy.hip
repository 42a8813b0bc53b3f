// Push of one contiguous block of device memory into up to seven peer-mapped buffers with ONE
// kernel: the alternative transport of the peer-memory all-gather (parallel/peer_mem.py).
//
// The default transport issues one runtime copy per peer (the copy engines move the data and
// no CU is involved); this kernel instead lets a few workgroups per peer read the local block
// and store it over xGMI themselves. Which of the two drives a link faster is a property of the
// machine, so the sharded-GEMM benchmark times both. The kernel is small on purpose -- 256
// threads, no LDS, a handful of registers -- so its workgroups fit next to the persistent GEMM
// workgroup on a CU and the product of the next panel keeps running while it pushes.
//
// Plain stores only; completion is the end of the kernel (the host synchronises the stream and
// the ranks meet in a barrier before anyone reads), so there is no flag protocol and no spin.
#include "common.h"

namespace {

constexpr int PUSH_THREADS = 256;
constexpr int MAX_PEERS = 8;

struct PushArgs {
  char* dst[MAX_PEERS];
  int count;
};

// grid = count * blocks_per_peer; workgroup b serves peer b / blocks_per_peer and walks the
// block in grid-stride steps of (blocks_per_peer * 256) 16-byte chunks, four loads in flight
__global__ __launch_bounds__(PUSH_THREADS) void peer_push_kernel(const char* __restrict__ src,
                                                                 PushArgs args, long bytes,
                                                                 int blocks_per_peer) {
  const int peer = blockIdx.x / blocks_per_peer;
  const int lane_block = blockIdx.x - peer * blocks_per_peer;
  char* __restrict__ dst = args.dst[peer];
  const long nvec = bytes >> 4;
  const long stride = (long)blocks_per_peer * PUSH_THREADS;
  const f32x4* s = reinterpret_cast<const f32x4*>(src);
  f32x4* d = reinterpret_cast<f32x4*>(dst);
  long i = (long)lane_block * PUSH_THREADS + threadIdx.x;
  for (; i + 3 * stride < nvec; i += 4 * stride) {
    const f32x4 v0 = s[i], v1 = s[i + stride], v2 = s[i + 2 * stride], v3 = s[i + 3 * stride];
    d[i] = v0;
    d[i + stride] = v1;
    d[i + 2 * stride] = v2;
    d[i + 3 * stride] = v3;
  }
  for (; i < nvec; i += stride) d[i] = s[i];
  // tail bytes (bytes is a multiple of the element size, not necessarily of 16)
  if (lane_block == 0) {
    for (long b = (nvec << 4) + threadIdx.x; b < bytes; b += PUSH_THREADS) dst[b] = src[b];
  }
}

}  // namespace

// dst[p][0..bytes) = src[0..bytes) for p < count (count <= 8). src and every dst must be
// 16-byte aligned. blocks_per_peer workgroups serve each destination.
DTF_API int dtf_peer_push(const void* src, void* const* dst, int count, long bytes,
                          int blocks_per_peer, hipStream_t stream) {
  if (!src || !dst || count < 1 || count > MAX_PEERS || bytes < 0 || blocks_per_peer < 1 ||
      blocks_per_peer > 256)
    return DTF_ERR_BAD_ARG;
  if ((uintptr_t)src & 15) return DTF_ERR_BAD_ARG;
  PushArgs args = {};
  for (int p = 0; p < count; ++p) {
    if (!dst[p] || ((uintptr_t)dst[p] & 15)) return DTF_ERR_BAD_ARG;
    args.dst[p] = (char*)dst[p];
  }
  args.count = count;
  if (bytes == 0) return DTF_OK;
  hipLaunchKernelGGL(peer_push_kernel, dim3(count * blocks_per_peer), dim3(PUSH_THREADS), 0,
                     stream, (const char*)src, args, bytes, blocks_per_peer);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
