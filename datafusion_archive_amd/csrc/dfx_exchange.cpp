// dfx_exchange.cpp -- multi-GPU GROUP BY: the exchange of group partials as ONE library call over RCCL.
//
// The reference is single-process (README.md:20); SURVEY.md section 8(e) defines the MI355X addition: every rank (one
// process per GPU) aggregates its own row range, the GROUPS (never the rows) are bucketed by hash(key) % world, one
// all-to-all of counts and one of payload move the buckets over xGMI, every rank merges what it received and emits the
// groups it owns.  dfx_aggregate_partial_{build,export,import} expose the three device steps to a host that brings its
// own collective (datafusion_archive_amd/distributed.py: torch.distributed); dfx_aggregate_exchange below is the same
// protocol inside the library: count kernel -> grouped ncclSend/ncclRecv of the counts -> ONE host read-back (buffer
// sizes) -> scatter kernel -> grouped ncclSend/ncclRecv of the buckets -> merge kernels, all on the library's stream.
// Ungrouped aggregates (one row per rank) are combined with an all-gather of the 2 x n_aggregates state words.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): the library has no link-time dependency on it, a process that has
// already loaded an RCCL (PyTorch ships one under the same soname) shares that copy, and hosts without RCCL can still
// load the library -- dfx_comm_* then fail with ExecutionError.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "dfx_relation.hpp"

namespace dfx {
namespace {
struct Rccl {
  void* handle = nullptr;
  std::string why;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      const char* e = dlerror();
      r.why = std::string("RCCL is not available (dlopen librccl.so.1: ") + (e ? e : "?") + ")";
      return;
    }
#define DFX_SYM(field, name)                                                      \
  r.field = (decltype(r.field))dlsym(r.handle, name);                             \
  if (!r.field && r.why.empty()) r.why = std::string("RCCL lacks the symbol ") + name;
    DFX_SYM(GetUniqueId, "ncclGetUniqueId")
    DFX_SYM(CommInitRank, "ncclCommInitRank")
    DFX_SYM(CommDestroy, "ncclCommDestroy")
    DFX_SYM(GroupStart, "ncclGroupStart")
    DFX_SYM(GroupEnd, "ncclGroupEnd")
    DFX_SYM(Send, "ncclSend")
    DFX_SYM(Recv, "ncclRecv")
    DFX_SYM(AllGather, "ncclAllGather")
    DFX_SYM(AllReduce, "ncclAllReduce")
    DFX_SYM(GetErrorString, "ncclGetErrorString")
#undef DFX_SYM
  });
  return r;
}

Status nccl_status(ncclResult_t rc, const char* what) {
  if (rc == ncclSuccess) return Status::OK();
  Rccl& r = rccl();
  return Status::Err(DFX_EXECUTION_ERROR, strfmt("RCCL %s failed: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "?"));
}
#define DFX_NCCL(call, what)                         \
  do {                                               \
    Status st__ = nccl_status((call), what);         \
    if (!st__.ok()) return st__;                     \
  } while (0)
}  // namespace
}  // namespace dfx

struct dfx_comm {
  ncclComm_t comm = nullptr;
  int world = 1;
  int rank = 0;
};

namespace dfx {

// all-to-all of `count_of(peer)` 64-bit words per peer; peer == rank is a device-to-device copy
template <typename SendAt, typename RecvAt>
static Status all_to_all_words(dfx_comm* c, SendAt send_at, RecvAt recv_at, hipStream_t s) {
  Rccl& r = rccl();
  if (c->world > 1) DFX_NCCL(r.GroupStart(), "ncclGroupStart");
  Status st = Status::OK();
  for (int peer = 0; peer < c->world && st.ok(); ++peer) {
    const void* sp = nullptr;
    void* rp = nullptr;
    size_t sn = 0, rn = 0;
    send_at(peer, &sp, &sn);
    recv_at(peer, &rp, &rn);
    if (peer == c->rank) {
      if (sn != rn) st = Status::Err(DFX_INTERNAL_ERROR, "exchange: a rank's own bucket changed size");
      else if (sn) {
        hipError_t e = hipMemcpyAsync(rp, sp, sn * sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) st = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s in the exchange", hipGetErrorString(e)));
      }
      continue;
    }
    if (sn) st = nccl_status(r.Send(sp, sn, ncclUint64, peer, c->comm, s), "ncclSend");
    if (st.ok() && rn) st = nccl_status(r.Recv(rp, rn, ncclUint64, peer, c->comm, s), "ncclRecv");
  }
  if (c->world > 1) {
    Status ge = nccl_status(r.GroupEnd(), "ncclGroupEnd");
    if (st.ok()) st = ge;
  }
  return st;
}

Status AggregateRelation::exchange(dfx_comm* c, int64_t* stats) {
  if (!c) return Status::Err(DFX_GENERAL, "null communicator");
  const int world = c->world;
  hipStream_t s = ctx().stream;
  int64_t host_syncs = 0;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (is_ungrouped()) {  // one row per rank: all-gather the accumulator states, fold them with the aggregates' algebra
    DFX_RETURN_IF_ERROR(ungrouped_state_begin());
    const int nw = ungrouped_state_words();
    Status st;
    auto all = device_alloc(sizeof(uint64_t) * (size_t)nw * (size_t)world, &st);
    if (!all) return st;
    if (world > 1) {
      DFX_NCCL(rccl().AllGather(ungrouped_state_device(), all.get(), (size_t)nw, ncclUint64, c->comm, s), "ncclAllGather");
    } else {
      DFX_HIP(hipMemcpyAsync(all.get(), ungrouped_state_device(), sizeof(uint64_t) * (size_t)nw, hipMemcpyDeviceToDevice, s));
    }
    std::vector<uint64_t> host((size_t)nw * (size_t)world);
    DFX_HIP(hipMemcpyAsync(host.data(), all.get(), sizeof(uint64_t) * host.size(), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    ++host_syncs;
    DFX_RETURN_IF_ERROR(ungrouped_state_merge(host.data(), world, c->rank));
    if (stats) {
      stats[0] = stats[1] = 1;
      stats[2] = (int64_t)(sizeof(uint64_t) * (size_t)nw);
      stats[3] = host_syncs;
    }
    return Status::OK();
  }
  // ---- grouped ----
  int nw = 0;
  std::vector<int64_t> send_counts((size_t)world, 0);
  uint64_t* d_counts = nullptr;
  std::shared_ptr<void> counts_owner;
  DFX_RETURN_IF_ERROR(partial_count_device(world, &nw, &d_counts, &counts_owner));  // [0, world): send counts, [world, 2 world): room for the received ones
  DFX_RETURN_IF_ERROR(all_to_all_words(
      c, [&](int peer, const void** p, size_t* n) { *p = d_counts + peer; *n = 1; },
      [&](int peer, void** p, size_t* n) { *p = d_counts + world + peer; *n = 1; }, s));
  std::vector<uint64_t> hc((size_t)world * 2);
  DFX_HIP(hipMemcpyAsync(hc.data(), d_counts, sizeof(uint64_t) * hc.size(), hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));  // the ONE read-back of the exchange: buffer sizes
  ++host_syncs;
  std::vector<int64_t> recv_counts((size_t)world, 0);
  std::vector<uint64_t> sbase((size_t)world + 1, 0), rbase((size_t)world + 1, 0);
  for (int r = 0; r < world; ++r) {
    send_counts[r] = (int64_t)hc[r];
    recv_counts[r] = (int64_t)hc[(size_t)world + r];
    sbase[r + 1] = sbase[r] + hc[r];
    rbase[r + 1] = rbase[r] + hc[(size_t)world + r];
  }
  Status st;
  auto send = device_alloc(sizeof(uint64_t) * (size_t)std::max<uint64_t>(1, sbase[world] * (uint64_t)nw), &st);
  if (!send) return st;
  auto recv = device_alloc(sizeof(uint64_t) * (size_t)std::max<uint64_t>(1, rbase[world] * (uint64_t)nw), &st);
  if (!recv) return st;
  DFX_RETURN_IF_ERROR(partial_export_with(send_counts, send.get(), (int64_t)(sbase[world] * (uint64_t)nw), /*sync=*/false));
  uint64_t* sw = (uint64_t*)send.get();
  uint64_t* rw = (uint64_t*)recv.get();
  DFX_RETURN_IF_ERROR(all_to_all_words(
      c, [&](int peer, const void** p, size_t* n) { *p = sw + sbase[peer] * (uint64_t)nw; *n = (size_t)(hc[peer] * (uint64_t)nw); },
      [&](int peer, void** p, size_t* n) { *p = rw + rbase[peer] * (uint64_t)nw; *n = (size_t)(hc[(size_t)world + peer] * (uint64_t)nw); }, s));
  DFX_RETURN_IF_ERROR(partial_import(recv.get(), recv_counts.data(), world));  // merges on the same stream, synchronises once
  ++host_syncs;
  if (stats) {
    stats[0] = (int64_t)sbase[world];
    stats[1] = (int64_t)rbase[world];
    stats[2] = (int64_t)(sizeof(uint64_t) * sbase[world] * (uint64_t)nw);
    stats[3] = host_syncs;
  }
  return Status::OK();
}

}  // namespace dfx

using namespace dfx;

extern "C" {

int32_t dfx_comm_unique_id(uint8_t* id, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!id) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    Rccl& r = rccl();
    if (!r.why.empty()) return to_c(Status::Err(DFX_EXECUTION_ERROR, r.why), err, errlen);
    static_assert(sizeof(ncclUniqueId) == DFX_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    Status st = nccl_status(r.GetUniqueId(&u), "ncclGetUniqueId");
    if (st.ok()) memcpy(id, &u, sizeof(u));
    return to_c(st, err, errlen);
  });
}

int32_t dfx_comm_init(const uint8_t* id, int32_t world, int32_t rank, dfx_comm** out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!id || !out) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    if (world < 1 || world > 1024 || rank < 0 || rank >= world) return to_c(Status::Err(DFX_GENERAL, "bad world / rank"), err, errlen);
    Status st = ensure_init();  // the library's device is the communicator's device
    if (!st.ok()) return to_c(st, err, errlen);
    Rccl& r = rccl();
    if (!r.why.empty()) return to_c(Status::Err(DFX_EXECUTION_ERROR, r.why), err, errlen);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    std::unique_ptr<dfx_comm> c(new dfx_comm());
    c->world = world;
    c->rank = rank;
    {  // ncclCommInitRank binds the communicator to the calling thread's CURRENT device
      hipError_t e = hipSetDevice(ctx().device);
      if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, strfmt("hipSetDevice(%d): %s", ctx().device, hipGetErrorString(e))), err, errlen);
    }
    st = nccl_status(r.CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");
    if (!st.ok()) return to_c(st, err, errlen);
    *out = c.release();
    return DFX_OK;
  });
}

void dfx_comm_destroy(dfx_comm* c) {
  if (!c) return;
  if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
  delete c;
}

int32_t dfx_aggregate_exchange(struct ArrowArrayStream* agg, dfx_comm* comm, int64_t* stats, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    Relation* r = peek_exported(agg);
    if (!r || r->kind() != REL_AGGREGATE)
      return to_c(Status::Err(DFX_GENERAL, "not an aggregate stream of this library"), err, errlen);
    return to_c(static_cast<AggregateRelation*>(r)->exchange(comm, stats), err, errlen);
  });
}

}  // extern "C"
