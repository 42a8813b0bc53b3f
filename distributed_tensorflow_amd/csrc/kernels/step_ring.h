// Step ring: how the kernels of a CAPTURED training step find their per-replay inputs and outputs.
//
// A HIP graph replays fixed kernel arguments, but every step reads a different batch (views of a
// data set resident in GPU memory) and must hand its small results (loss, accuracy, step counter)
// to a host that may already be preparing later steps. Both go through a ring of `ring` slots
// (a power of two) in pinned host memory, indexed by the number of the replay:
//
//   table[slot][i]  address of input i for that replay, stored by the host before the launch;
//   out[slot][...]  the packed small results of that replay, stored by the step's last kernel,
//                   whose final word is the replay's stamp -- the host polls it.
//
// Behind the two counter words the counter buffer holds a device-memory copy of the replay's
// input addresses (DTF_RING_FEEDS 64-bit words): the step's first kernel stores it next to the
// replay number, so that later kernels of the step find an input's address with a load from
// device memory instead of another round trip to pinned host memory.
//
// The replay number lives in device memory as two words so that no kernel reads a word another
// workgroup of the SAME kernel writes: ctr[1] is the number of the next replay. The first kernel
// of the step reads it (every workgroup) and its workgroup 0 publishes it as ctr[0], which all
// later kernels of the step read; the last kernel stores ctr[1] = ctr[0] + 1 when it is done.
// The host keeps its own count of launches and so knows each replay's slot without reading
// anything back; it never rewrites a slot before the replay `ring` launches earlier has stamped
// it. Numbers are unsigned and wrap; slot and stamp are taken from the low bits.
#pragma once

#include "common.h"

constexpr int DTF_RING_FEEDS = 8;          // inputs per slot
constexpr int DTF_RING_WORDS = 256;        // packed result words per slot, the last is the stamp
constexpr int DTF_RING_CTR_WORDS = 2 + 2 * DTF_RING_FEEDS;   // 32-bit words of the counter buffer

struct DtfStepRing {
  const unsigned long long* table;
  unsigned* ctr;                     // null: not a ring step (arguments are direct pointers);
                                     // DTF_RING_CTR_WORDS words: two counters, address copies
  float* out;                        // pinned [ring][DTF_RING_WORDS]
  int ring;
  long host_seq;                     // >= 0: the replay number, given by the host at launch (a
                                     // step issued directly, not replayed from a graph): the
                                     // counters are neither read nor written (the address
                                     // copies are: steps of one stream run in order)
};

__device__ __forceinline__ unsigned dtf_ring_stamp(unsigned seq) { return (seq & 0x7fffffu) + 1u; }

// first kernel of the step: every workgroup learns the replay number, workgroup 0 publishes it
__device__ __forceinline__ unsigned dtf_ring_begin(const DtfStepRing& r) {
  const unsigned seq = r.host_seq >= 0 ? (unsigned)r.host_seq : r.ctr[1];
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    if (threadIdx.x == 0 && r.host_seq < 0) r.ctr[0] = seq;
    if (threadIdx.x < DTF_RING_FEEDS) {        // the replay's input addresses, for later kernels
      const unsigned slot = seq & (unsigned)(r.ring - 1);
      reinterpret_cast<unsigned long long*>(r.ctr + 2)[threadIdx.x] =
          __builtin_nontemporal_load(r.table + slot * DTF_RING_FEEDS + threadIdx.x);
    }
  }
  return seq;
}

// later kernels of the step
__device__ __forceinline__ unsigned dtf_ring_current(const DtfStepRing& r) {
  return r.host_seq >= 0 ? (unsigned)r.host_seq : r.ctr[0];
}

// address of input `index` of replay `seq` (pinned host memory: never from a stale cache line)
__device__ __forceinline__ const void* dtf_ring_feed(const DtfStepRing& r, unsigned seq, int index) {
  const unsigned slot = seq & (unsigned)(r.ring - 1);
  return reinterpret_cast<const void*>(
      __builtin_nontemporal_load(r.table + slot * DTF_RING_FEEDS + index));
}

// the same address for a LATER kernel of the step (not the one that calls dtf_ring_begin): from the
// copy the first kernel left in device memory
__device__ __forceinline__ const void* dtf_ring_feed_later(const DtfStepRing& r, int index) {
  return reinterpret_cast<const void*>(
      reinterpret_cast<const unsigned long long*>(r.ctr + 2)[index]);
}

__device__ __forceinline__ float* dtf_ring_out(const DtfStepRing& r, unsigned seq) {
  return r.out + (size_t)(seq & (unsigned)(r.ring - 1)) * DTF_RING_WORDS;
}

// Last kernel of the step, ONE thread, after the packed words of the slot have been stored by
// threads it has synchronised with: stamp the slot for the host (system-scope release) and open
// the next replay.
__device__ __forceinline__ void dtf_ring_finish(const DtfStepRing& r, unsigned seq) {
  __hip_atomic_store(dtf_ring_out(r, seq) + (DTF_RING_WORDS - 1), (float)dtf_ring_stamp(seq),
                     __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (r.host_seq < 0) r.ctr[1] = seq + 1u;
}
