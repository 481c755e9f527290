// dfx_k_dict.hip -- Utf8 GROUP BY keys (reference: GroupByScalar::Utf8(String), aggregate.rs:65-76, :838-846).
//
// The reference clones every row's string into a heap-allocated key.  Here a Utf8 key column is
// dictionary-encoded ON THE DEVICE, batch by batch, into stable 64-bit ids (insertion order); the ids are
// bound to the fused aggregation program as an ordinary UInt64 key column, so all GROUP BY strategies work
// unchanged.  At emit time the group ids are turned back into an Arrow Utf8 column.
//
// Dictionary = open-addressing slot table {state, hash, id} + per-id {pool offset, length} + byte pool.
// A slot is claimed with a 0 -> 1 CAS on its state word, filled, and published with state = 2; the winner
// publishes inside the loop iteration in which it won, so lanes of the same wave that wait on it cannot
// starve it.  Ids never change when the slot table is rebuilt (growth), hence ids already stored in the
// group table stay valid.
#include <algorithm>

#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {

DEV uint64_t hash_bytes(const uint8_t* p, uint32_t len) {  // FNV-1a 64 + finaliser (the reference feeds FNV too, aggregate.rs:793)
  uint64_t h = 0xCBF29CE484222325ull;
  for (uint32_t i = 0; i < len; ++i) h = (h ^ p[i]) * 0x100000001B3ull;
  return mix64(h ^ len);
}

DEV bool bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t len) {
  for (uint32_t i = 0; i < len; ++i)
    if (a[i] != b[i]) return false;
  return true;
}

// ids[i] = dictionary id of string i (value(i) as the reference reads it: offsets[i] .. offsets[i + 1], no null check)
__global__ __launch_bounds__(kBlock) void k_dict_encode(const int32_t* __restrict__ offsets, const uint8_t* __restrict__ data,
                                                        int64_t n, const DevDict D, uint64_t* __restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const int32_t o0 = offsets[i], o1 = offsets[i + 1];
  const uint32_t len = o1 > o0 ? (uint32_t)(o1 - o0) : 0u;
  const uint8_t* str = data + o0;
  const uint64_t h = hash_bytes(str, len);
  uint64_t slot = (h >> D.shift) & D.mask;
  uint64_t id = ~0ull;
  uint32_t spins = 0;
  for (uint64_t probes = 0; probes <= D.mask;) {
    uint32_t st = __hip_atomic_load(&D.state[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if (st == 0u) {
      const uint32_t old = atomicCAS(&D.state[slot], 0u, 1u);
      if (old == 0u) {  // ours: allocate an id and pool space, publish
        const uint64_t my = atomicAdd((unsigned long long*)&D.cursors[DICT_IDS], 1ull);
        const uint64_t at = atomicAdd((unsigned long long*)&D.cursors[DICT_POOL], (unsigned long long)len);
        if (my >= D.id_cap || at + len > D.pool_cap) {  // full: the host grows the dictionary and re-encodes the batch
          atomicExch((unsigned long long*)&D.cursors[DICT_OVERFLOW], 1ull);
          __hip_atomic_store(&D.state[slot], 3u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // tombstone until the rebuild
          break;
        }
        for (uint32_t b = 0; b < len; ++b) D.pool[at + b] = str[b];
        D.str_off[my] = at;
        D.str_len[my] = len;
        D.hash[slot] = h;
        D.sid[slot] = my;
        __hip_atomic_store(&D.state[slot], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        id = my;
        break;
      }
      st = old;
    }
    if (st == 1u) {  // another lane is filling this slot
      if (++spins > (1u << 22)) {
        atomicExch((unsigned long long*)&D.cursors[DICT_OVERFLOW], 2ull);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    if (st == 2u && D.hash[slot] == h) {
      const uint64_t cand = D.sid[slot];
      if (D.str_len[cand] == len && bytes_equal(D.pool + D.str_off[cand], str, len)) {
        id = cand;
        break;
      }
    }
    slot = (slot + 1) & D.mask;
    ++probes;
  }
  if (id == ~0ull) atomicExch((unsigned long long*)&D.cursors[DICT_OVERFLOW], 1ull);
  ids[i] = id;
}

// rebuild the slot table of a grown dictionary from the per-id arrays (ids, offsets and pool are kept)
__global__ __launch_bounds__(kBlock) void k_dict_rebuild(const DevDict D, uint64_t n_ids) {
  const uint64_t id = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (id >= n_ids) return;
  const uint32_t len = D.str_len[id];
  const uint64_t h = hash_bytes(D.pool + D.str_off[id], len);
  uint64_t slot = (h >> D.shift) & D.mask;
  for (uint64_t probes = 0; probes <= D.mask; ++probes) {
    if (atomicCAS(&D.state[slot], 0u, 2u) == 0u) {  // distinct strings: no lookups race with this kernel
      D.hash[slot] = h;
      D.sid[slot] = id;
      return;
    }
    slot = (slot + 1) & D.mask;
  }
}

// emit: lengths of the group keys, then their bytes
__global__ __launch_bounds__(kBlock) void k_dict_lengths(const uint64_t* __restrict__ ids, int64_t g, const DevDict D,
                                                         uint32_t* __restrict__ lens) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < g) lens[i] = D.str_len[ids[i]];
}
__global__ __launch_bounds__(kBlock) void k_dict_gather(const uint64_t* __restrict__ ids, int64_t g, const DevDict D,
                                                        const uint64_t* __restrict__ starts, int32_t* __restrict__ offsets,
                                                        uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i > g) return;
  offsets[i] = (int32_t)starts[i];  // starts has g + 1 entries (exclusive scan + total)
  if (i == g) return;
  const uint64_t id = ids[i];
  const uint8_t* src = D.pool + D.str_off[id];
  const uint32_t len = D.str_len[id];
  uint8_t* dst = out + starts[i];
  for (uint32_t b = 0; b < len; ++b) dst[b] = src[b];
}

hipError_t launch_dict_encode(const int32_t* offsets, const uint8_t* data, int64_t n, const DevDict& D, uint64_t* ids,
                              hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_GATHER_UTF8, s, 0);
  hipLaunchKernelGGL(k_dict_encode, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, offsets, data, n, D, ids);
  return hipGetLastError();
}
hipError_t launch_dict_rebuild(const DevDict& D, uint64_t n_ids, hipStream_t s) {
  if (n_ids == 0) return hipSuccess;
  hipLaunchKernelGGL(k_dict_rebuild, dim3((unsigned)((n_ids + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, D, n_ids);
  return hipGetLastError();
}
// multi-GPU exchange of Utf8 keys: the rank-local ids of a key plane become GLOBAL ids (remap[local id]); slots that hold no
// id (empty: kEmptyKey or stale words >= n_ids) are left alone
__global__ __launch_bounds__(kBlock) void k_dict_remap_plane(uint64_t* __restrict__ plane, uint64_t n_slots, const uint64_t* __restrict__ remap,
                                                             uint64_t n_ids) {
  for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * kBlock) {
    const uint64_t v = plane[i];
    if (v < n_ids) plane[i] = remap[v];
  }
}
hipError_t launch_dict_remap_plane(uint64_t* plane, uint64_t n_slots, const uint64_t* remap, uint64_t n_ids, hipStream_t s) {
  if (n_slots == 0 || n_ids == 0) return hipSuccess;
  hipLaunchKernelGGL(k_dict_remap_plane, dim3((unsigned)std::min<uint64_t>((n_slots + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0, s, plane, n_slots, remap, n_ids);
  return hipGetLastError();
}
hipError_t launch_dict_lengths(const uint64_t* ids, int64_t g, const DevDict& D, uint32_t* lens, hipStream_t s) {
  if (g <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_dict_lengths, dim3((unsigned)((g + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, ids, g, D, lens);
  return hipGetLastError();
}
hipError_t launch_dict_gather(const uint64_t* ids, int64_t g, const DevDict& D, const uint64_t* starts, int32_t* offsets,
                              uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_dict_gather, dim3((unsigned)((g + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, ids, g, D, starts, offsets, out);
  return hipGetLastError();
}

}  // namespace dfx
