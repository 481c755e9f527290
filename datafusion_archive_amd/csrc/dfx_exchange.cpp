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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "dfx_relation.hpp"

namespace dfx {
namespace {
struct Rccl {
  void* handle = nullptr;
  std::string why;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
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
    // DFX_RCCL_LIB: bind this library instead (a differently named RCCL build; the tests' host-staged stand-in that lets
    // several ranks share one GPU, tests/native/rccl_stub.cpp)
    const char* override_lib = getenv("DFX_RCCL_LIB");
    const char* names[] = {override_lib, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | (n == override_lib ? RTLD_LOCAL : RTLD_GLOBAL));
      if (r.handle || n == override_lib) break;  // an override that does not load is an error, not a reason to look elsewhere
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
    DFX_SYM(CommCount, "ncclCommCount")
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
  // Device words reserved when the communicator is created: everything the ranks use to tell each other how they are --
  // counts, failure marks, ready flags, the ungrouped state blocks -- lives here, so that no allocation can fail between a
  // rank's decision to take part in a collective and the collective itself.
  std::shared_ptr<void> slab;
  uint64_t* words = nullptr;
};

namespace dfx {

// slab layout (64-bit words; W = world)
static size_t slab_flags(int) { return 0; }                                   // [2 W]  send | received: counts, marks, flags
static size_t slab_dict_mine(int W) { return 2 * (size_t)W; }                  // [3]
static size_t slab_dict_all(int W) { return 2 * (size_t)W + 4; }               // [3 W]
static size_t slab_state_mine(int W) { return 5 * (size_t)W + 8; }             // [2 kMaxAggs]  (ungrouped)
static size_t slab_state_all(int W) { return 5 * (size_t)W + 8 + 2 * kMaxAggs; }  // [2 kMaxAggs x W]
static size_t slab_gather_mine(int W) { return slab_state_all(W) + (size_t)2 * kMaxAggs * (size_t)W + 8; }  // [W + 3]  (grouped, round 1)
static size_t slab_gather_all(int W) { return slab_gather_mine(W) + (size_t)W + 4; }                      // [W][W + 3]
static size_t slab_words(int W) { return slab_gather_all(W) + (size_t)W * ((size_t)W + 3) + 8; }

// all-to-all of `count_of(peer)` 64-bit words per peer; peer == rank is a device-to-device copy
// t_out / t_in (may be null): one TRAILER word more per peer -- t_out[peer] travels behind the bucket, lands in t_in[peer]
template <typename SendAt, typename RecvAt>
static Status all_to_all_words(dfx_comm* c, SendAt send_at, RecvAt recv_at, hipStream_t s, const uint64_t* t_out = nullptr, uint64_t* t_in = nullptr) {
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
      if (st.ok() && t_out) {
        hipError_t e = hipMemcpyAsync(t_in + peer, t_out + peer, sizeof(uint64_t), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) st = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s in the exchange", hipGetErrorString(e)));
      }
      continue;
    }
    if (sn) st = nccl_status(r.Send(sp, sn, ncclUint64, peer, c->comm, s), "ncclSend");
    if (st.ok() && t_out) st = nccl_status(r.Send(t_out + peer, 1, ncclUint64, peer, c->comm, s), "ncclSend");
    if (st.ok() && rn) st = nccl_status(r.Recv(rp, rn, ncclUint64, peer, c->comm, s), "ncclRecv");
    if (st.ok() && t_in) st = nccl_status(r.Recv(t_in + peer, 1, ncclUint64, peer, c->comm, s), "ncclRecv");
  }
  if (c->world > 1) {
    Status ge = nccl_status(r.GroupEnd(), "ncclGroupEnd");
    if (st.ok()) st = ge;
  }
  return st;
}

// every rank's variable-sized blob to every rank: `mine` (device, n_mine words) lands in all[r] on rank r's peers.  sizes[]
// (words per rank) is known to everybody beforehand.  Grouped sends / receives like the all-to-all.
static Status all_gather_v_words(dfx_comm* c, const uint64_t* mine, const std::vector<uint64_t>& sizes, uint64_t* all, hipStream_t s) {
  std::vector<uint64_t> base((size_t)c->world + 1, 0);
  for (int r = 0; r < c->world; ++r) base[(size_t)r + 1] = base[(size_t)r] + sizes[(size_t)r];
  return all_to_all_words(
      c, [&](int, const void** p, size_t* n) { *p = mine; *n = (size_t)sizes[(size_t)c->rank]; },
      [&](int peer, void** p, size_t* n) { *p = all + base[(size_t)peer]; *n = (size_t)sizes[(size_t)peer]; }, s);
}

constexpr uint64_t kPeerFailed = ~0ull;  // travels instead of a count / a flag: the sender hit an error, every rank gives up together

// test hook (tests/test_gpu_exchange_world2.py): the process-wide switch dfx_set_option("test.exchange_fail", rank << 8 | stage)
// makes that rank fail locally at that stage (stage numbers: kFailStages below; 0 = off, the default).  A switch the host
// sets explicitly, not an environment variable the production library would consult in every exchange.
static int64_t g_exchange_fail = 0;
void set_exchange_test_failure(int64_t v) { g_exchange_fail = v; }
static const char* const kFailStages[] = {"", "drain", "count", "payload_alloc", "export", "dict_local", "dict_blob_alloc", "merge"};
static bool inject_failure(const dfx_comm* c, const char* stage) {
  const int64_t v = g_exchange_fail;
  if (v <= 0) return false;
  const int st = (int)(v & 0xFF);
  return st > 0 && st < (int)(sizeof(kFailStages) / sizeof(kFailStages[0])) && (int)(v >> 8) == c->rank && !strcmp(kFailStages[st], stage);
}
static Status injected(const dfx_comm* c, const char* stage) {
  return inject_failure(c, stage) ? Status::Err(DFX_EXECUTION_ERROR, strfmt("injected failure at stage '%s' (test.exchange_fail)", stage)) : Status::OK();
}
static Status hip_local(hipError_t e) {
  return e == hipSuccess ? Status::OK() : Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s in the exchange", hipGetErrorString(e)));
}

// THE rule of this file: between its first and its last collective a rank never returns on a LOCAL failure.  It folds the
// failure into `local`, goes on taking part in every collective with well-formed (if meaningless) messages, and all ranks
// leave together at the next agree(): every rank tells every rank whether it is still well -- one word per peer over the
// reserved slab -- plus a word describing what it is about to exchange (`shape`: keys, chunks, dictionaries ...), so ranks
// whose queries differ (one of them carries a deferred set-up error, say) find out instead of waiting for each other.
// Returns the rank's own error if it has one, else "rank r failed <what>" / "ranks disagree", else OK -- the same verdict
// (ok or not) on every rank.  (A collective that itself fails inside RCCL is RCCL's to report: nothing to agree over.)
static Status agree(dfx_comm* c, const Status& local, uint64_t shape, const char* what, int64_t* host_syncs, hipStream_t s) {
  if (c->world <= 1) return local;
  const int W = c->world;
  const uint64_t word = local.ok() ? (1ull | (shape << 8)) : kPeerFailed;
  std::vector<uint64_t> hf((size_t)W * 2, 0);
  for (int r = 0; r < W; ++r) hf[(size_t)r] = word;
  uint64_t* d = c->words + slab_flags(W);
  // a local HIP failure here does not excuse this rank from the round: its peers are already waiting in it.  It takes part
  // with whatever the slab holds (a stale word at worst: the peers then disagree on the shape and leave as well) and
  // reports its own error afterwards.
  Status mine = local;
  {
    Status cp = hip_local(hipMemcpyAsync(d, hf.data(), sizeof(uint64_t) * hf.size(), hipMemcpyHostToDevice, s));
    if (!cp.ok()) (void)hipStreamSynchronize(s);
    if (mine.ok()) mine = cp;
  }
  Status coll = all_to_all_words(
      c, [&](int peer, const void** p, size_t* n) { *p = d + peer; *n = 1; },
      [&](int peer, void** p, size_t* n) { *p = d + W + peer; *n = 1; }, s);
  if (!coll.ok()) return mine.ok() ? coll : mine;  // (a collective that fails inside RCCL is RCCL's to report)
  {
    Status rb = hip_local(hipMemcpyAsync(hf.data(), d, sizeof(uint64_t) * hf.size(), hipMemcpyDeviceToHost, s));
    Status sy = hip_local(hipStreamSynchronize(s));
    if (mine.ok()) mine = rb.ok() ? sy : rb;
  }
  ++*host_syncs;
  if (!mine.ok()) return mine;
  for (int r = 0; r < W; ++r)
    if (hf[(size_t)W + r] == kPeerFailed) return Status::Err(DFX_EXECUTION_ERROR, strfmt("multi-GPU exchange: rank %d failed %s", r, what));
  for (int r = 0; r < W; ++r)
    if (hf[(size_t)W + r] != word)
      return Status::Err(DFX_EXECUTION_ERROR, strfmt("multi-GPU exchange: rank %d is exchanging a different query (keys / accumulator chunks / dictionaries)", r));
  return Status::OK();
}

// Utf8 GROUP BY keys: dictionary ids are rank-local.  Every rank learns every rank's strings (sizes first, then lengths +
// bytes), builds the SAME global dictionary on the host (rank 0's strings in id order, then rank 1's new ones, ...), rewrites
// its key plane to global ids and installs the global dictionary for the emit.  O(distinct strings of all ranks) on the host.
// Aligned on every rank: per dictionary a sizes round (carrying the failure mark), an agree() after the blob buffers were
// allocated, the blob round, an agree() after the host-side build.  Returns the same verdict on every rank.
static Status globalise_dictionaries(AggregateRelation* a, dfx_comm* c, int64_t* host_syncs) {
  const int world = c->world;
  hipStream_t s = ctx().stream;
  Status local = Status::OK();
  for (int d = 0; d < a->exchange_dicts(); ++d) {
    std::vector<uint32_t> lens;
    std::vector<uint8_t> pool;
    if (local.ok()) local = a->exchange_dict_local(d, &lens, &pool);
    if (local.ok()) local = injected(c, "dict_local");
    if (!local.ok()) { lens.clear(); pool.clear(); }
    // blob = lens (u32, padded to words) + bytes (padded to words)
    const uint64_t len_words = ((uint64_t)lens.size() + 1) / 2, pool_words = ((uint64_t)pool.size() + 7) / 8;
    uint64_t hmine[3] = {local.ok() ? (uint64_t)lens.size() : kPeerFailed, (uint64_t)pool.size(), 0};
    uint64_t* dmine = c->words + slab_dict_mine(world);
    uint64_t* dall = c->words + slab_dict_all(world);
    DFX_HIP(hipMemcpyAsync(dmine, hmine, sizeof(hmine), hipMemcpyHostToDevice, s));
    std::vector<uint64_t> three((size_t)world, 3);
    DFX_RETURN_IF_ERROR(all_gather_v_words(c, dmine, three, dall, s));
    std::vector<uint64_t> hall((size_t)world * 3, 0);
    DFX_HIP(hipMemcpyAsync(hall.data(), dall, sizeof(uint64_t) * hall.size(), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    ++*host_syncs;
    hall[(size_t)c->rank * 3] = hmine[0];
    hall[(size_t)c->rank * 3 + 1] = hmine[1];
    for (int r = 0; r < world; ++r)
      if (hall[(size_t)r * 3] == kPeerFailed)
        return local.ok() ? Status::Err(DFX_EXECUTION_ERROR, strfmt("multi-GPU exchange: rank %d failed before the exchange", r)) : local;
    std::vector<uint64_t> sizes((size_t)world), base((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) {
      sizes[(size_t)r] = (hall[(size_t)r * 3] + 1) / 2 + (hall[(size_t)r * 3 + 1] + 7) / 8;
      base[(size_t)r + 1] = base[(size_t)r] + sizes[(size_t)r];
    }
    std::vector<uint64_t> blob((size_t)std::max<uint64_t>(1, len_words + pool_words), 0);
    if (!lens.empty()) memcpy(blob.data(), lens.data(), sizeof(uint32_t) * lens.size());
    if (!pool.empty()) memcpy(blob.data() + len_words, pool.data(), pool.size());
    // the blob buffers: allocated before the ranks commit themselves to the blob round
    Status st;
    auto dblob = device_alloc(sizeof(uint64_t) * blob.size(), &st);
    std::shared_ptr<void> dblobs;
    if (dblob) dblobs = device_alloc(sizeof(uint64_t) * (size_t)std::max<uint64_t>(1, base[(size_t)world]), &st);
    if (!dblob || !dblobs) local = st;
    if (local.ok()) local = injected(c, "dict_blob_alloc");
    if (local.ok()) {
      hipError_t e = hipMemcpyAsync(dblob.get(), blob.data(), sizeof(uint64_t) * blob.size(), hipMemcpyHostToDevice, s);
      if (e != hipSuccess) local = Status::Err(DFX_EXECUTION_ERROR, strfmt("HIP error %s in the exchange", hipGetErrorString(e)));
    }
    DFX_RETURN_IF_ERROR(agree(c, local, (uint64_t)d, "while it prepared its dictionary", host_syncs, s));
    DFX_RETURN_IF_ERROR(all_gather_v_words(c, (const uint64_t*)dblob.get(), sizes, (uint64_t*)dblobs.get(), s));
    std::vector<uint64_t> blobs((size_t)std::max<uint64_t>(1, base[(size_t)world]), 0);
    DFX_HIP(hipMemcpyAsync(blobs.data(), dblobs.get(), sizeof(uint64_t) * blobs.size(), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    ++*host_syncs;
    // the same global dictionary on every rank
    std::unordered_map<std::string, uint64_t> ids;
    std::vector<uint32_t> glens;
    std::vector<uint8_t> gpool;
    std::vector<uint64_t> remap;
    for (int r = 0; r < world && local.ok(); ++r) {
      const uint64_t n_ids = hall[(size_t)r * 3], n_bytes = hall[(size_t)r * 3 + 1];
      const uint32_t* rl = r == c->rank ? lens.data() : (const uint32_t*)(blobs.data() + base[(size_t)r]);
      const uint8_t* rp = r == c->rank ? pool.data() : (const uint8_t*)(blobs.data() + base[(size_t)r] + (n_ids + 1) / 2);
      uint64_t at = 0;
      for (uint64_t i = 0; i < n_ids; ++i) {
        if (at + rl[i] > n_bytes) {
          local = Status::Err(DFX_INTERNAL_ERROR, "multi-GPU exchange: malformed dictionary blob");
          break;
        }
        std::string str((const char*)rp + at, (size_t)rl[i]);
        at += rl[i];
        auto ins = ids.emplace(std::move(str), (uint64_t)glens.size());
        if (ins.second) {
          glens.push_back(rl[i]);
          gpool.insert(gpool.end(), (const uint8_t*)ins.first->first.data(), (const uint8_t*)ins.first->first.data() + ins.first->first.size());
        }
        if (r == c->rank) remap.push_back(ins.first->second);
      }
    }
    if (local.ok()) local = a->exchange_dict_globalise(d, glens, gpool, remap);
    DFX_RETURN_IF_ERROR(agree(c, local, (uint64_t)d, "while it installed the global dictionary", host_syncs, s));
  }
  return Status::OK();
}

Status AggregateRelation::exchange(dfx_comm* c, int64_t* stats) {
  if (!c) return Status::Err(DFX_GENERAL, "null communicator");
  const int world = c->world;
  hipStream_t s = ctx().stream;
  int64_t host_syncs = 0;
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  // phase clocks of this rank (counters xchg_*): the local scan until the stream is idle, the first agreement (the wait for the
  // slowest rank), everything after
  ++counters().xchg_calls;
  const long long t_begin = ScopedUs::now();
  if (is_ungrouped()) {  // one row per rank: all-gather the accumulator states, fold them with the aggregates' algebra
    Status local = ungrouped_state_begin();  // (drains the input: DivideByZero, out of memory ...)
    if (local.ok()) local = hip_local(hipStreamSynchronize(s));
    const long long t_local = ScopedUs::now();
    counters().xchg_local_us += t_local - t_begin;
    if (local.ok()) local = injected(c, "drain");
    const int nw = ungrouped_state_words();
    const int n_chunks = exchange_chunks();
    {
      Status ag = agree(c, local, 0x100ull | ((uint64_t)n_chunks << 16) | ((uint64_t)nw << 24), "before the exchange", &host_syncs, s);
      counters().xchg_wait_peers_us += ScopedUs::now() - t_local;
      DFX_RETURN_IF_ERROR(ag);
    }
    ScopedUs t_rest(&counters().xchg_exchange_us);
    if (nw > 2 * kMaxAggs) return Status::Err(DFX_INTERNAL_ERROR, "exchange: ungrouped state wider than the communicator's slab");
    uint64_t* all = c->words + slab_state_all(world);
    for (int ch = 0; ch < n_chunks; ++ch) {  // more than kMaxAggs accumulators: one state block per chunk
      if (local.ok()) local = ungrouped_select_chunk(ch);
      if (local.ok() && ch == n_chunks - 1) local = injected(c, "merge");
      const void* mine = ungrouped_state_device();
      if (!local.ok() || !mine) mine = c->words + slab_state_mine(world);  // (something well-formed to send: every rank gives up at the agree() below)
      if (world > 1) {
        DFX_NCCL(rccl().AllGather(mine, all, (size_t)nw, ncclUint64, c->comm, s), "ncclAllGather");
      } else {
        Status cp = hip_local(hipMemcpyAsync(all, mine, sizeof(uint64_t) * (size_t)nw, hipMemcpyDeviceToDevice, s));
        if (local.ok()) local = cp;
      }
      // (a local HIP failure is folded into `local`: the peers are about to enter the next chunk's all-gather and the
      // agree() behind the loop -- this rank goes with them)
      std::vector<uint64_t> host((size_t)nw * (size_t)world);
      {
        Status rb = hip_local(hipMemcpyAsync(host.data(), all, sizeof(uint64_t) * host.size(), hipMemcpyDeviceToHost, s));
        Status sy = hip_local(hipStreamSynchronize(s));
        if (local.ok()) local = rb.ok() ? sy : rb;
      }
      ++host_syncs;
      if (local.ok()) local = ungrouped_state_merge(host.data(), world, c->rank);
    }
    if (local.ok()) local = ungrouped_select_chunk(0);
    DFX_RETURN_IF_ERROR(agree(c, local, 0x101ull, "while it merged the ranks' states", &host_syncs, s));
    if (stats) {
      stats[0] = stats[1] = 1;
      stats[2] = (int64_t)(sizeof(uint64_t) * (size_t)nw * (size_t)n_chunks);
      stats[3] = host_syncs;
    }
    return Status::OK();
  }
  // ---- grouped ----
  // Round 6: TWO collective rounds and TWO host synchronisations in the common case (rounds 4-5: agree -> counts -> read-back ->
  // agree -> payload -> agree, four read-backs).
  //   round 1  ONE all-gather of world + 3 words per rank: {how the rank is | what it is about to exchange, the groups it can
  //            receive and the groups it can send without allocating anything more, its group count per destination rank}.  Every rank then holds the whole
  //            count matrix and every rank's state -- the first agreement, the count round and the buffer agreement in one.
  //            Everything a rank needs for the payload round is allocated BEFORE this round (send buffer: its own group count,
  //            exact; receive buffer and import table: capacity for twice its own group count -- ranks that scanned row ranges
  //            of one table own about as many groups as each holds), so a rank that cannot allocate says so here.
  //   (extra)  only if some rank receives (or holds) more than the capacity it announced -- every rank computes that from the same matrix,
  //            so all of them take this branch or none does -- that rank allocates again and the ranks agree() on the outcome.
  //   round 2  the buckets, chunk by chunk as before; the LAST chunk's message to every peer carries one trailer word more: how
  //            the sender was when it sent.  A rank that failed locally has kept sending well-formed buckets; its mark ends the
  //            exchange on every rank after the one synchronisation of the payload rounds (no closing agree()).
  // What a rank does after its last collective (the merge of what it received, the control block's read-back) can only fail
  // that rank: nobody waits for it any more.
  const int W = world;
  constexpr size_t H = 3;                                       // header words of a rank's round-1 message: state, receive capacity, send capacity
  const size_t MW = (size_t)W + H;                              // ... + its counts
  uint64_t* d_mine = c->words + slab_gather_mine(W);            // [W + 3]
  uint64_t* d_all = c->words + slab_gather_all(W);              // [W][W + 3]
  Status local = exchange_drain();
  if (local.ok()) local = hip_local(hipStreamSynchronize(s));
  const long long t_local = ScopedUs::now();
  counters().xchg_local_us += t_local - t_begin;
  if (local.ok()) local = injected(c, "drain");
  const int n_chunks = exchange_chunks();
  int widest = 0;
  uint64_t shape = 0x200ull | ((uint64_t)n_chunks << 12) | ((uint64_t)exchange_dicts() << 20);
  for (int ch = 0; ch < n_chunks; ++ch) {
    widest = std::max(widest, exchange_chunk_words(ch));
    shape = shape * 31 + (uint64_t)exchange_chunk_words(ch);
  }
  shape &= 0xFFFFFFFFFFFFull;
  int64_t rounds = 0;
  if (exchange_dicts() > 0) {
    // Utf8 keys: the dictionaries are globalised first (rounds of their own: sizes, blobs, agreements); a rank that failed its
    // drain must not enter them, so this form keeps round 4's agreement in front
    Status ag = agree(c, local, shape, "before the exchange", &host_syncs, s);
    ++rounds;
    DFX_RETURN_IF_ERROR(ag);
    DFX_RETURN_IF_ERROR(globalise_dictionaries(this, c, &host_syncs));
  }
  // everything the payload round needs, before anybody commits to it
  const uint64_t own_groups = local.ok() ? exchange_group_bound() : 0;  // (what the host knows; the count kernel has the last word)
  uint64_t send_cap = own_groups + 64;
  uint64_t cap_groups = 2 * own_groups + 4096;
  Status st;
  std::shared_ptr<void> send, recv;
  if (local.ok()) {
    send = device_alloc(sizeof(uint64_t) * (size_t)(send_cap * (uint64_t)widest + (uint64_t)W), &st);   // (+ W trailer words)
    if (send) recv = device_alloc(sizeof(uint64_t) * (size_t)(cap_groups * (uint64_t)widest + (uint64_t)W), &st);
    if (!send || !recv) local = st;
  }
  if (local.ok()) local = injected(c, "payload_alloc");
  if (local.ok()) local = exchange_import_begin(cap_groups);
  // round 1
  {
    if (local.ok()) {
      local = exchange_count(W, d_mine + H);
      if (local.ok()) local = injected(c, "count");
    }
    if (!local.ok()) {  // a rank that is not well sends zeros behind its failure mark (its table may not even exist)
      Status z = hip_local(hipMemsetAsync(d_mine + H, 0, sizeof(uint64_t) * (size_t)W, s));
      (void)z;
    }
    const uint64_t word = local.ok() ? (1ull | (shape << 8)) : kPeerFailed;
    Status f0 = hip_local(launch_fill_u64(d_mine, word, 1, s));
    Status f1 = hip_local(launch_fill_u64(d_mine + 1, cap_groups, 1, s));
    Status f2 = hip_local(launch_fill_u64(d_mine + 2, send_cap, 1, s));
    if (local.ok()) local = f0.ok() ? (f1.ok() ? f2 : f1) : f0;
    if (W > 1) {
      DFX_NCCL(rccl().AllGather(d_mine, d_all, MW, ncclUint64, c->comm, s), "ncclAllGather");
    } else {
      Status cp = hip_local(hipMemcpyAsync(d_all, d_mine, sizeof(uint64_t) * MW, hipMemcpyDeviceToDevice, s));
      if (local.ok()) local = cp;
    }
    ++rounds;
  }
  std::vector<uint64_t> M((size_t)W * MW);
  {  // the ONE read-back before the payload: every rank's state, capacity and counts
    Status rb = hip_local(hipMemcpyAsync(M.data(), d_all, sizeof(uint64_t) * M.size(), hipMemcpyDeviceToHost, s));
    Status sy = hip_local(hipStreamSynchronize(s));
    if (local.ok()) local = rb.ok() ? sy : rb;
  }
  ++host_syncs;
  counters().xchg_wait_peers_us += ScopedUs::now() - t_local;
  ScopedUs t_rest(&counters().xchg_exchange_us);
  // (every rank has finished round 1 -- the last collective before this point -- so leaving here strands nobody)
  if (!local.ok()) return local;
  const uint64_t my_word = 1ull | (shape << 8);
  for (int r = 0; r < W; ++r)
    if (M[(size_t)r * MW] == kPeerFailed) return Status::Err(DFX_EXECUTION_ERROR, strfmt("multi-GPU exchange: rank %d failed before the exchange", r));
  for (int r = 0; r < W; ++r)
    if (M[(size_t)r * MW] != my_word)
      return Status::Err(DFX_EXECUTION_ERROR, strfmt("multi-GPU exchange: rank %d is exchanging a different query (keys / accumulator chunks / dictionaries)", r));
  std::vector<int64_t> send_counts((size_t)W, 0), recv_counts((size_t)W, 0);
  std::vector<uint64_t> sbase((size_t)W + 1, 0), rbase((size_t)W + 1, 0);
  bool need_more = false;  // (the same verdict on every rank: it is computed from the same matrix)
  for (int r = 0; r < W; ++r) {
    uint64_t into_r = 0, from_r = 0;
    for (int q = 0; q < W; ++q) {
      into_r += M[(size_t)q * MW + H + (size_t)r];
      from_r += M[(size_t)r * MW + H + (size_t)q];
    }
    if (into_r > M[(size_t)r * MW + 1] || from_r > M[(size_t)r * MW + 2]) need_more = true;
  }
  for (int r = 0; r < W; ++r) {
    send_counts[(size_t)r] = (int64_t)M[(size_t)c->rank * MW + H + (size_t)r];
    recv_counts[(size_t)r] = (int64_t)M[(size_t)r * MW + H + (size_t)c->rank];
    sbase[(size_t)r + 1] = sbase[(size_t)r] + (uint64_t)send_counts[(size_t)r];
    rbase[(size_t)r + 1] = rbase[(size_t)r] + (uint64_t)recv_counts[(size_t)r];
  }
  if (need_more) {  // some rank owns (or holds) more groups than it made room for: allocate again, agree on the outcome (round 5's form)
    local = Status::OK();
    if (sbase[(size_t)W] > send_cap) {
      send.reset();
      send_cap = sbase[(size_t)W];
      send = device_alloc(sizeof(uint64_t) * (size_t)(send_cap * (uint64_t)widest + (uint64_t)W), &st);
      if (!send) local = st;
    }
    if (local.ok() && rbase[(size_t)W] > cap_groups) {
      recv.reset();
      cap_groups = rbase[(size_t)W];
      recv = device_alloc(sizeof(uint64_t) * (size_t)(cap_groups * (uint64_t)widest + (uint64_t)W), &st);
      if (!recv) local = st;
      if (local.ok()) local = exchange_import_begin(cap_groups);
    }
    Status ag = agree(c, local, 0x201ull, "while it allocated its exchange buffers", &host_syncs, s);
    ++rounds;
    DFX_RETURN_IF_ERROR(ag);
    if (!local.ok()) return local;  // (world == 1: agree() is the identity)
  }
  // round 2: the buckets
  uint64_t* sw = (uint64_t*)send.get();
  uint64_t* rw = (uint64_t*)recv.get();
  uint64_t* t_out = sw + send_cap * (uint64_t)widest;            // [W] how this rank was when it sent its last bucket
  uint64_t* t_in = rw + cap_groups * (uint64_t)widest;           // [W] ... and its peers
  int64_t sent_words = 0;
  for (int ch = 0; ch < n_chunks; ++ch) {  // accumulators beyond kMaxAggs: one round per chunk of planes, the same keys every time
    const uint64_t nw = (uint64_t)exchange_chunk_words(ch);
    const bool last = ch == n_chunks - 1;
    // a rank whose scatter / merge kernels fail keeps sending and receiving what was agreed (its peers merge rows they will
    // throw away): the trailer of the last round ends the exchange on every rank
    if (local.ok()) local = exchange_export_chunk(ch, send_counts, send.get(), (int64_t)(sbase[(size_t)W] * nw));
    if (local.ok() && last) local = injected(c, "export");
    if (last) {
      Status f = hip_local(launch_fill_u64(t_out, local.ok() ? 1ull : kPeerFailed, W, s));
      if (local.ok()) local = f;
      Status z = hip_local(hipMemsetAsync(t_in, 0, sizeof(uint64_t) * (size_t)W, s));
      if (local.ok()) local = z;
    }
    DFX_RETURN_IF_ERROR(all_to_all_words(
        c, [&](int peer, const void** p, size_t* n) { *p = sw + sbase[(size_t)peer] * nw; *n = (size_t)((uint64_t)send_counts[(size_t)peer] * nw); },
        [&](int peer, void** p, size_t* n) { *p = rw + rbase[(size_t)peer] * nw; *n = (size_t)((uint64_t)recv_counts[(size_t)peer] * nw); }, s,
        last ? t_out : nullptr, last ? t_in : nullptr));
    ++rounds;
    if (local.ok()) local = exchange_import_chunk(ch, recv.get(), recv_counts.data(), W);  // merges on the same stream
    sent_words += (int64_t)(sbase[(size_t)W] * nw);
  }
  std::vector<uint64_t> trailers((size_t)W, 0);
  {
    Status rb = hip_local(hipMemcpyAsync(trailers.data(), t_in, sizeof(uint64_t) * (size_t)W, hipMemcpyDeviceToHost, s));
    if (local.ok()) local = rb;
  }
  {
    Status fin = exchange_import_finish();  // the one synchronisation of the payload rounds (also when this rank is not well: the trailers)
    if (local.ok()) local = fin;
  }
  ++host_syncs;
  counters().xchg_rounds += rounds;
  counters().xchg_host_syncs += host_syncs;
  if (!local.ok()) return local;
  for (int r = 0; r < W; ++r)
    if (trailers[(size_t)r] == kPeerFailed) return Status::Err(DFX_EXECUTION_ERROR, strfmt("multi-GPU exchange: rank %d failed during the payload rounds", r));
  if (stats) {
    stats[0] = (int64_t)sbase[(size_t)W];
    stats[1] = (int64_t)rbase[(size_t)W];
    stats[2] = (int64_t)sizeof(uint64_t) * sent_words;
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
    // the slab first: a rank that cannot have it must not enter the collective communicator set-up half-way
    c->slab = device_alloc(sizeof(uint64_t) * slab_words(world), &st);
    if (!c->slab) return to_c(st, err, errlen);
    c->words = (uint64_t*)c->slab.get();
    {
      hipError_t e = hipMemset(c->words, 0, sizeof(uint64_t) * slab_words(world));
      if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, strfmt("hipMemset: %s", hipGetErrorString(e))), err, errlen);
    }
    st = nccl_status(r.CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");
    if (!st.ok()) return to_c(st, err, errlen);
    *out = c.release();
    return DFX_OK;
  });
}

int32_t dfx_comm_ranks(const dfx_comm* c) {
  if (!c || !c->comm || !rccl().CommCount) return -1;
  int n = -1;
  return rccl().CommCount(c->comm, &n) == ncclSuccess ? (int32_t)n : -1;
}

void dfx_comm_destroy(dfx_comm* c) {
  if (!c) return;
  (void)hipStreamSynchronize(ctx().stream);  // nothing of this communicator is still queued on the library's stream
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
