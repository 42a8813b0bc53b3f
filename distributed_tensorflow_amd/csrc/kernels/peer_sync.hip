// Device-side completion for the peer-memory collectives (parallel/peer_mem.py).
//
// A rank that has pushed rows into a peer's buffer tells the peer so with a FLAG, written by a
// one-lane kernel that sits on the pushing stream behind the copy; the peer's consumer stream
// waits for the flags of all ranks with a one-wave kernel in front of whatever reads the rows.
// Neither side synchronises its host, so pipelined steps never drain the compute stream.
//
// Every rank owns one flag page in its own HBM, allocated FINE-GRAINED (hipExtMallocWithFlags,
// hipDeviceMallocFinegrained -- the kind of memory RCCL keeps its own peer flags in): accesses to
// it are coherent across the GPUs of the node, so a system-scope release store that a peer's
// kernel sends over xGMI is seen by a system-scope acquire load of the owner. The page is
// exported with hipIpcGetMemHandle and opened by every peer. A rank POLLS ONLY ITS OWN page
// (local memory, no fabric traffic while waiting) and WRITES only into peers' pages. One flag is
// one 32-bit sequence number in its own 64-byte line: flag (kind, src) of rank dst's page =
// "exchange number `value` of rank src towards dst has happened" (kinds: rows pushed / local
// partials readable in place / pushed rows consumed / partials consumed). Sequence numbers only
// grow, so a waiter compares with >= and a flag never needs resetting.
//
// Every wait is bounded: a lane that does not see its flag within `timeout_us` (wall clock,
// 100 MHz counter) stores its position + 1 into the status word and the kernel ends -- the
// grid always drains. The host reads the status word (pinned memory) at its next check and
// raises CommError; what the consumer kernels computed from the incomplete rows is discarded.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

constexpr int FLAG_STRIDE = 16;      // ints per flag: one 64-byte line each
constexpr int MAX_RANKS = 64;

struct Targets {
  int* flag[MAX_RANKS];     // signal: the flag to write (in a peer's page); wait: the flag to poll
  int count;
};

// *flag[i] = value for i < count, each a system-scope release store: everything this stream did
// before the kernel (copies into peer memory, kernels that wrote partials) is complete and
// visible before the number is.
__global__ __launch_bounds__(64) void peer_signal_kernel(Targets t, int value) {
  const int i = threadIdx.x;
  if (i < t.count) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(t.flag[i], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// lane i < count waits until *flag[i] >= value (wrap-safe signed difference) or the time is up;
// status: 0 = all seen, else (position of the first missing flag) + 1
__global__ __launch_bounds__(64) void peer_wait_kernel(Targets t, int value, long timeout_ticks,
                                                       int* status) {
  const int i = threadIdx.x;
  bool ok = true;
  if (i < t.count) {
    const int* f = t.flag[i];
    const long start = (long)wall_clock64();
    for (;;) {
      const int seen = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (seen - value >= 0) break;
      if ((long)wall_clock64() - start > timeout_ticks) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  const unsigned long long missing = __ballot(!ok);
  if (threadIdx.x == 0 && missing != 0) {
    const int lane = __builtin_ctzll(missing);
    __hip_atomic_store(status, lane + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

int fill_targets(Targets& t, void* const* flags, int count) {
  if (!flags || count < 1 || count > MAX_RANKS) return DTF_ERR_BAD_ARG;
  for (int i = 0; i < count; ++i) {
    if (!flags[i] || ((uintptr_t)flags[i] & 3)) return DTF_ERR_BAD_ARG;
    t.flag[i] = (int*)flags[i];
  }
  t.count = count;
  return DTF_OK;
}

}  // namespace

// Allocate this rank's flag page: `bytes` of zeroed fine-grained device memory on the current
// device, and its 64-byte IPC handle for the peers. Returns 0 or a hipError_t (the runtime's
// last-error state is cleared on failure: callers fall back to host synchronisation).
DTF_API int dtf_flags_alloc(long bytes, void** ptr, void* handle64) {
  if (bytes <= 0 || !ptr || !handle64) return DTF_ERR_BAD_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handles travel as 64 bytes");
  void* p = nullptr;
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t h;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) {
    if (getenv("DTF_PEER_DEBUG")) fprintf(stderr, "dtf_flags_alloc: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    if (p) (void)hipFree(p);
    return (int)e;
  }
  memcpy(handle64, &h, 64);
  *ptr = p;
  return DTF_OK;
}

// Map a peer's flag page (the 64-byte handle its dtf_flags_alloc produced) into this process.
DTF_API int dtf_flags_open(const void* handle64, void** ptr) {
  if (!handle64 || !ptr) return DTF_ERR_BAD_ARG;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    if (getenv("DTF_PEER_DEBUG")) fprintf(stderr, "dtf_flags_open: %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    return (int)e;
  }
  *ptr = p;
  return DTF_OK;
}

DTF_API int dtf_flags_close(void* ptr, int owned) {
  if (!ptr) return DTF_ERR_BAD_ARG;
  const hipError_t e = owned ? hipFree(ptr) : hipIpcCloseMemHandle(ptr);
  if (e != hipSuccess) (void)hipGetLastError();
  return (int)e;
}

// On `stream`: *flags[i] = value (system-scope release) for i < count.
DTF_API int dtf_peer_signal(void* const* flags, int count, int value, hipStream_t stream) {
  Targets t = {};
  const int rc = fill_targets(t, flags, count);
  if (rc != DTF_OK) return rc;
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, stream, t, value);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}

// On `stream`: wait until every *flags[i] >= value. timeout_us is clamped to [1 ms, 60 s]: a wait
// kernel always ends; a flag still behind leaves its position + 1 in *status_dev.
DTF_API int dtf_peer_wait(void* const* flags, int count, int value, long timeout_us,
                          void* status_dev, hipStream_t stream) {
  if (!status_dev) return DTF_ERR_BAD_ARG;
  Targets t = {};
  const int rc = fill_targets(t, flags, count);
  if (rc != DTF_OK) return rc;
  if (timeout_us < 1000) timeout_us = 1000;
  if (timeout_us > 60000000) timeout_us = 60000000;
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, stream, t, value, timeout_us * 100,
                     (int*)status_dev);
  DTF_LAUNCH_CHECK();
  return DTF_OK;
}
