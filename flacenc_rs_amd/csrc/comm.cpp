// comm.cpp -- ParSink's ordered gather (src/par.rs:67-95) across processes, inside the C library: an RCCL
// communicator per handle and the all-gather of fixed-size per-frame records on the caller's stream, so that a Rust
// or C++ host with one process per GPU has the collective behind the same ABI as the kernels around it
// (flacenc_hip_stereo_frame_wire_async in front, flacenc_hip_stream_offsets_async / _place_frames_async behind).
//
// librccl is opened at run time (dlopen): the library has no link-time dependency on it, a single-GPU drop-in never
// loads it, and inside a process that already holds an RCCL (PyTorch's) the loader hands back that copy.
#include "comm.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace flacenc_hip {

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) get_error_string = nullptr;
  std::string why;  // why it could not be loaded
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {std::getenv("FLACENC_HIP_RCCL"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names) {
      if (!name || !*name) continue;
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
      r.why = dlerror();
    }
    if (!r.lib) return;
    auto sym = [&](const char* s) -> void* {
      void* p = dlsym(r.lib, s);
      if (!p) r.why = std::string("librccl: no symbol ") + s;
      return p;
    };
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(sym("ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(sym("ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
    r.all_gather = reinterpret_cast<decltype(r.all_gather)>(sym("ncclAllGather"));
    r.get_error_string = reinterpret_cast<decltype(r.get_error_string)>(sym("ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather || !r.get_error_string) {
      dlclose(r.lib);
      r.lib = nullptr;
    }
  });
  return r;
}

int rccl_fail(flacenc_hip_handle* h, const char* what, ncclResult_t rc) {
  if (h) handle_set_error(h, std::string(what) + ": " + rccl().get_error_string(rc));
  return FLACENC_HIP_ERR_DEVICE;
}

}  // namespace

struct CommState {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 0;
};

void comm_release(CommState* c) {
  if (!c) return;
  if (c->comm && rccl().lib) rccl().comm_destroy(c->comm);
  delete c;
}

}  // namespace flacenc_hip

using namespace flacenc_hip;

extern "C" {

int flacenc_hip_comm_unique_id(uint8_t id[FLACENC_HIP_COMM_ID_BYTES]) {
  static_assert(FLACENC_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId");
  if (!id) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  Rccl& r = rccl();
  if (!r.lib) return FLACENC_HIP_ERR_UNSUPPORTED;
  ncclUniqueId u;
  if (r.get_unique_id(&u) != ncclSuccess) return FLACENC_HIP_ERR_DEVICE;
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return FLACENC_HIP_OK;
}

int flacenc_hip_comm_create(flacenc_hip_handle* h, const uint8_t id[FLACENC_HIP_COMM_ID_BYTES], int rank, int world) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  Rccl& r = rccl();
  if (!r.lib) {
    handle_set_error(h, "librccl could not be loaded: " + r.why);
    return FLACENC_HIP_ERR_UNSUPPORTED;
  }
  CommState*& slot = handle_comm_slot(h);
  if (slot) {
    handle_set_error(h, "the handle already has a communicator (flacenc_hip_comm_destroy first)");
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  if (hipSetDevice(handle_device(h)) != hipSuccess) return FLACENC_HIP_ERR_DEVICE;
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = r.comm_init_rank(&comm, world, u, rank);
  if (rc != ncclSuccess) return rccl_fail(h, "ncclCommInitRank", rc);
  slot = new CommState;
  slot->comm = comm;
  slot->rank = rank;
  slot->world = world;
  return FLACENC_HIP_OK;
}

int flacenc_hip_comm_destroy(flacenc_hip_handle* h) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  CommState*& slot = handle_comm_slot(h);
  comm_release(slot);
  slot = nullptr;
  return FLACENC_HIP_OK;
}

int flacenc_hip_comm_info(flacenc_hip_handle* h, int* rank, int* world) {
  if (!h) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  const CommState* c = handle_comm_slot(h);
  if (rank) *rank = c ? c->rank : 0;
  if (world) *world = c ? c->world : 0;
  return FLACENC_HIP_OK;
}

int flacenc_hip_allgather_async(flacenc_hip_handle* h, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
  if (!h || !recv || (!send && bytes_per_rank)) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  CommState* c = handle_comm_slot(h);
  if (!c) {
    handle_set_error(h, "no communicator: flacenc_hip_comm_create first");
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  if (bytes_per_rank == 0) return FLACENC_HIP_OK;
  const ncclResult_t rc = rccl().all_gather(send, recv, bytes_per_rank, ncclUint8, c->comm, static_cast<hipStream_t>(stream));
  if (rc != ncclSuccess) return rccl_fail(h, "ncclAllGather", rc);
  return FLACENC_HIP_OK;
}

int flacenc_hip_allgather_records_async(flacenc_hip_handle* h, const void* local, size_t n_local, size_t n_total,
                                        size_t record_bytes, void* gathered, void* stream) {
  if (!h || !gathered || record_bytes == 0 || (!local && n_local)) return FLACENC_HIP_ERR_BAD_ARGUMENT;
  CommState* c = handle_comm_slot(h);
  if (!c) {
    handle_set_error(h, "no communicator: flacenc_hip_comm_create first");
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  const size_t world = (size_t)c->world, rank = (size_t)c->rank;
  const size_t per_rank = (n_total + world - 1) / world;
  // frame f belongs to rank f mod world: rank r owns ceil((n_total - r) / world) frames
  const size_t mine = n_total > rank ? (n_total - rank + world - 1) / world : 0;
  if (n_local != mine) {
    handle_set_error(h, "flacenc_hip_allgather_records_async: n_local is not this rank's share of n_total");
    return FLACENC_HIP_ERR_BAD_ARGUMENT;
  }
  if (per_rank == 0) return FLACENC_HIP_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // in place: this rank's slot of the output is its send buffer, zero-padded when the rank is one frame short
  unsigned char* slot = static_cast<unsigned char*>(gathered) + rank * per_rank * record_bytes;
  if (slot != local && n_local) {
    if (hipMemcpyAsync(slot, local, n_local * record_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return FLACENC_HIP_ERR_DEVICE;
  }
  if (n_local < per_rank) {
    if (hipMemsetAsync(slot + n_local * record_bytes, 0, (per_rank - n_local) * record_bytes, s) != hipSuccess)
      return FLACENC_HIP_ERR_DEVICE;
  }
  const ncclResult_t rc = rccl().all_gather(slot, gathered, per_rank * record_bytes, ncclUint8, c->comm, s);
  if (rc != ncclSuccess) return rccl_fail(h, "ncclAllGather", rc);
  return FLACENC_HIP_OK;
}

}  // extern "C"
