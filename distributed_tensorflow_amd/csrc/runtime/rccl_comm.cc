// Native RCCL communicator: ncclUniqueId / ncclCommInitRank / collectives on raw device pointers
// and HIP streams, without torch.distributed in between (parallel/rccl_native.py is the Python
// side; rendezvous.cc carries the unique id to the peers).
//
// librccl is opened at run time (dlopen "librccl.so.1", then "librccl.so"): the host library has
// no link-time dependency on ROCm, builds on a machine without it, and shares the copy of RCCL
// the process already holds, if any. Only C types cross this interface: pointers, counts, the
// nccl enums as ints, and the stream handle as an opaque pointer.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace {

struct UniqueId { char internal[128]; };     // ncclUniqueId
typedef void* Comm;                          // ncclComm_t
typedef void* Stream;                        // hipStream_t

struct Api {
  void* lib = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommAbort)(Comm) = nullptr;
  int (*CommGetAsyncError)(Comm, int*) = nullptr;
  int (*CommSplit)(Comm, int, int, Comm*, void*) = nullptr;     // optional (RCCL >= 2.18)
  const char* (*GetErrorString)(int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, Stream) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, Comm, Stream) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, Stream) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, Stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  bool ok = false;
};

Api g_api;
std::once_flag g_once;

template <typename F>
bool sym(void* lib, const char* name, F& out, bool required = true) {
  out = reinterpret_cast<F>(dlsym(lib, name));
  return out != nullptr || !required;
}

void load() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_api.lib) break;
  }
  if (!g_api.lib) return;
  void* L = g_api.lib;
  bool ok = sym(L, "ncclGetUniqueId", g_api.GetUniqueId) &&
            sym(L, "ncclCommInitRank", g_api.CommInitRank) &&
            sym(L, "ncclCommDestroy", g_api.CommDestroy) &&
            sym(L, "ncclGetErrorString", g_api.GetErrorString) &&
            sym(L, "ncclAllReduce", g_api.AllReduce) &&
            sym(L, "ncclReduceScatter", g_api.ReduceScatter) &&
            sym(L, "ncclAllGather", g_api.AllGather) &&
            sym(L, "ncclBroadcast", g_api.Broadcast) &&
            sym(L, "ncclGroupStart", g_api.GroupStart) && sym(L, "ncclGroupEnd", g_api.GroupEnd);
  sym(L, "ncclGetVersion", g_api.GetVersion, false);
  sym(L, "ncclCommAbort", g_api.CommAbort, false);
  sym(L, "ncclCommGetAsyncError", g_api.CommGetAsyncError, false);
  sym(L, "ncclCommSplit", g_api.CommSplit, false);
  g_api.ok = ok;
}

const Api* api() {
  std::call_once(g_once, load);
  return g_api.ok ? &g_api : nullptr;
}

constexpr int ERR_NO_LIBRARY = -100, ERR_ARG = -1;

}  // namespace

// 1 when librccl could be opened and exports what this file needs
DTF_API int dtf_rccl_available() { return api() != nullptr; }

DTF_API int dtf_rccl_version() {
  const Api* a = api();
  int v = 0;
  if (a && a->GetVersion) a->GetVersion(&v);
  return v;
}

// nccl result codes are >= 0 (0 = success); negative codes are this file's own
DTF_API const char* dtf_rccl_error_string(int code) {
  if (code == ERR_NO_LIBRARY) return "librccl could not be loaded";
  if (code == ERR_ARG) return "bad argument";
  const Api* a = api();
  return (a && code >= 0) ? a->GetErrorString(code) : "unknown error";
}

DTF_API int dtf_rccl_unique_id(void* id128) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!id128) return ERR_ARG;
  UniqueId id;
  const int rc = a->GetUniqueId(&id);
  if (rc == 0) memcpy(id128, &id, sizeof(id));
  return rc;
}

// on the calling thread's current HIP device
DTF_API int dtf_rccl_init(void** comm, int nranks, const void* id128, int rank) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return ERR_ARG;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  return a->CommInitRank(comm, nranks, id, rank);
}

DTF_API int dtf_rccl_destroy(void* comm, int abort) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm) return ERR_ARG;
  if (abort && a->CommAbort) return a->CommAbort(comm);
  return a->CommDestroy(comm);
}

// *out = the communicator's asynchronous error (0 = none); returns a call status
DTF_API int dtf_rccl_async_error(void* comm, int* out) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm || !out) return ERR_ARG;
  *out = 0;
  return a->CommGetAsyncError ? a->CommGetAsyncError(comm, out) : 0;
}

// 1 when ncclCommSplit exists in the loaded library
DTF_API int dtf_rccl_has_split() {
  const Api* a = api();
  return a && a->CommSplit != nullptr;
}

// color < 0: this rank joins no group (*newcomm = null)
DTF_API int dtf_rccl_split(void* comm, int color, int key, void** newcomm) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm || !newcomm || !a->CommSplit) return ERR_ARG;
  return a->CommSplit(comm, color < 0 ? -1 : color, key, newcomm, nullptr);
}

DTF_API int dtf_rccl_all_reduce(void* comm, const void* send, void* recv, long count, int dtype,
                                int op, void* stream) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm || !send || !recv || count < 0) return ERR_ARG;
  return a->AllReduce(send, recv, (size_t)count, dtype, op, comm, stream);
}

// recv holds recvcount elements, send recvcount * nranks
DTF_API int dtf_rccl_reduce_scatter(void* comm, const void* send, void* recv, long recvcount,
                                    int dtype, int op, void* stream) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm || !send || !recv || recvcount < 0) return ERR_ARG;
  return a->ReduceScatter(send, recv, (size_t)recvcount, dtype, op, comm, stream);
}

// send holds sendcount elements, recv sendcount * nranks
DTF_API int dtf_rccl_all_gather(void* comm, const void* send, void* recv, long sendcount, int dtype,
                                void* stream) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm || !send || !recv || sendcount < 0) return ERR_ARG;
  return a->AllGather(send, recv, (size_t)sendcount, dtype, comm, stream);
}

DTF_API int dtf_rccl_broadcast(void* comm, void* buf, long count, int dtype, int root,
                               void* stream) {
  const Api* a = api();
  if (!a) return ERR_NO_LIBRARY;
  if (!comm || !buf || count < 0 || root < 0) return ERR_ARG;
  return a->Broadcast(buf, buf, (size_t)count, dtype, root, comm, stream);
}
