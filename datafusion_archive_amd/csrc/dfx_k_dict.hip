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

// The hash of a string = a chain of mix64 over its 8-byte little-endian words (the last one zero-padded), seeded with the
// length.  Two ways to get the words, one value: dict_word() assembles word k byte by byte (any address: the rebuild, long
// strings); k_dict_encode assembles the first two from ALIGNED 8-byte loads (round 6: a byte-at-a-time FNV walk was eleven
// dependent byte loads per string).  (The reference hashes its key enum with FNV-1a, aggregate.rs:793; the hash only decides
// the iteration order of its map, i.e. nothing a result depends on.)
DEV uint64_t dict_word(const uint8_t* p, uint32_t len, uint32_t k) {
  uint64_t w = 0;
  const uint32_t lo = k * 8u, hi = lo + 8u < len ? lo + 8u : len;
  for (uint32_t i = lo; i < hi; ++i) w |= (uint64_t)p[i] << (8u * (i - lo));
  return w;
}
DEV uint64_t dict_hash_begin(uint32_t len) { return 0xCBF29CE484222325ull ^ ((uint64_t)len * 0x9E3779B97F4A7C15ull); }
DEV uint64_t dict_hash_step(uint64_t h, uint64_t w) { return mix64(h ^ w); }
DEV uint64_t hash_bytes(const uint8_t* p, uint32_t len) {
  uint64_t h = dict_hash_begin(len);
  for (uint32_t k = 0; k * 8u < len; ++k) h = dict_hash_step(h, dict_word(p, len, k));
  return h;
}

DEV bool bytes_equal(const uint8_t* a, const uint8_t* b, uint32_t len) {
  for (uint32_t i = 0; i < len; ++i)
    if (a[i] != b[i]) return false;
  return true;
}

// ids[i] = dictionary id of string i (value(i) as the reference reads it: offsets[i] .. offsets[i + 1], no null check).
//
// Round 6.  Rounds 1-5: one thread per string, a byte-wise hash and an open-addressing probe of the global dictionary with
// agent-scope ACQUIRE loads (each invalidates the CU's L1) -- 3.7 ms per GB of CSV text for ~10^3 distinct 11-byte strings,
// three times what all CSV kernels together take: with so few distinct strings every lane of the chip hammers the same few
// hundred slot lines in L2.  Now the workgroups are PERSISTENT (a few per CU, each walks n / grid strings) and keep an LDS
// FRONT CACHE of strings they have resolved: kDictCache direct-mapped, insert-once entries {state, hash, id, length, the
// string's first 16 bytes}; a string of at most 16 bytes whose entry is there is answered from LDS alone (the BYTES are
// compared, not just the hash: an id is a group key).  Everything else -- the first sight of a string in a workgroup, a
// cache slot taken by another string, strings longer than 16 bytes -- takes the global path unchanged.  An entry is claimed
// 0 -> 1 by CAS, filled, published with state = 2 and never written again, so a reader that saw 2 reads a complete entry.
constexpr int kDictCache = 4096;  // entries of 24 bytes: 96 KB of LDS
constexpr int kDictBlock = 1024;  // ONE workgroup per CU
constexpr int kDictProbes = 8;    // linear probing, at most this many slots per lookup
constexpr int kDictPreload = 3072;  // entries of the dictionary (ids 0 ..) a workgroup copies into its cache when it starts

DEV uint64_t dict_global_lookup(const DevDict& D, const uint8_t* str, uint32_t len, uint64_t h) {
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
  return id;
}

// the string's first two 8-byte words (zero-padded; len <= 16) out of the (at most) three ALIGNED words that hold them; a word
// that holds none of its bytes is not loaded (nothing is read beyond the aligned word of the string's last byte)
DEV void dict_words16(const uint8_t* str, uint32_t len, uint64_t& w0, uint64_t& w1) {
  const uintptr_t a = (uintptr_t)str;
  const uint64_t* q = (const uint64_t*)(a & ~(uintptr_t)7);
  const uint32_t sh = (uint32_t)(a & 7u) * 8u;
  const uint32_t span = (uint32_t)(a & 7u) + len;  // bytes from q[0]'s first byte to the string's end
  const uint64_t x0 = len ? q[0] : 0ull;
  const uint64_t x1 = span > 8u ? q[1] : 0ull;
  const uint64_t x2 = span > 16u ? q[2] : 0ull;
  w0 = sh ? (x0 >> sh) | (x1 << (64u - sh)) : x0;
  w1 = sh ? (x1 >> sh) | (x2 << (64u - sh)) : x1;
  if (len < 8u) {
    w0 &= len ? (~0ull >> (64u - 8u * len)) : 0ull;
    w1 = 0;
  } else if (len < 16u) {
    w1 &= len > 8u ? (~0ull >> (64u - 8u * (len - 8u))) : 0ull;
  }
}
DEV uint64_t dict_hash16(uint32_t len, uint64_t w0, uint64_t w1) {  // == hash_bytes for len <= 16
  uint64_t h = dict_hash_begin(len);
  if (len > 0u) h = dict_hash_step(h, w0);
  if (len > 8u) h = dict_hash_step(h, w1);
  return h;
}

// The LDS front cache: open addressing, linear probing, INSERT-ONCE entries {tag, id, w0, w1}.  tag = 0 empty, 1 being filled,
// else 2 | length << 2 | (hash's bits 8..31) << 8 -- never 0 or 1, and two strings with equal tags are still compared by
// their bytes (w0, w1 hold them all: only strings of at most 16 bytes are cached).  An entry is claimed 0 -> 1 by CAS, filled,
// published with its tag and never written again: a reader that sees the tag reads a complete entry.
struct DictCache {
  uint32_t* tag;
  uint32_t* id;
  uint64_t* w0;
  uint64_t* w1;
};
DEV uint32_t dict_tag(uint64_t h, uint32_t len) { return 2u | (len << 2) | ((uint32_t)h & 0xFFFFFF00u); }
DEV uint32_t dict_cache_slot(uint64_t h) { return (uint32_t)(h >> 32) & (uint32_t)(kDictCache - 1); }  // (the dictionary's own slot: the TOP bits)
// id of the cached string, or ~0: `free_at` = the first empty slot seen (kDictCache: none -- the neighbourhood is full)
DEV uint64_t dict_cache_find(const DictCache& c, uint64_t h, uint32_t len, uint64_t w0, uint64_t w1, uint32_t& free_at) {
  const uint32_t tg = dict_tag(h, len);
  uint32_t e = dict_cache_slot(h);
  free_at = (uint32_t)kDictCache;
#pragma unroll 1
  for (int k = 0; k < kDictProbes; ++k) {
    const uint32_t t = __hip_atomic_load(&c.tag[e], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (t == 0u) {
      free_at = e;
      break;
    }
    if (t == tg && c.w0[e] == w0 && c.w1[e] == w1) return (uint64_t)c.id[e];
    e = (e + 1u) & (uint32_t)(kDictCache - 1);
  }
  return ~0ull;
}
DEV void dict_cache_insert(const DictCache& c, uint32_t at, uint64_t h, uint32_t len, uint64_t w0, uint64_t w1, uint64_t id) {
  if (at >= (uint32_t)kDictCache || id >= 0xFFFFFFFFull) return;
  if (atomicCAS(&c.tag[at], 0u, 1u) != 0u) return;  // somebody else's by now: this string stays uncached (or is cached twice: harmless)
  c.id[at] = (uint32_t)id;
  c.w0[at] = w0;
  c.w1[at] = w1;
  __hip_atomic_store(&c.tag[at], dict_tag(h, len), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// n_known: ids the dictionary held when this launch was queued (complete and visible: an earlier launch wrote them) -- the
// first kDictPreload of them are copied into the cache before the first string is looked at.  Without that EVERY workgroup
// meets every string for the first time once per launch, and a wave is as slow as its slowest lane: with ~10^3 distinct
// strings and 16 K strings per workgroup 6 % of the lanes -- 98 % of the waves -- still took the global path.
__global__ __launch_bounds__(kDictBlock) void k_dict_encode(const int32_t* __restrict__ offsets, const uint8_t* __restrict__ data,
                                                            int64_t n, const DevDict D, uint64_t n_known, uint64_t* __restrict__ ids) {
  __shared__ uint32_t c_tag[kDictCache];
  __shared__ uint32_t c_id[kDictCache];
  __shared__ uint64_t c_w0[kDictCache];
  __shared__ uint64_t c_w1[kDictCache];
  DictCache c;
  c.tag = c_tag;
  c.id = c_id;
  c.w0 = c_w0;
  c.w1 = c_w1;
  for (int e = threadIdx.x; e < kDictCache; e += kDictBlock) c_tag[e] = 0u;
  __syncthreads();
  for (uint64_t id = threadIdx.x; id < n_known && id < (uint64_t)kDictPreload; id += kDictBlock) {
    const uint32_t len = D.str_len[id];
    if (len > 16u) continue;
    const uint8_t* str = D.pool + D.str_off[id];
    const uint64_t w0 = dict_word(str, len, 0), w1 = dict_word(str, len, 1);
    const uint64_t h = dict_hash16(len, w0, w1);
    // (distinct strings: nothing to find, only a free slot to claim -- the CAS lets one of two lanes in, the other moves on)
    uint32_t e = dict_cache_slot(h);
    for (int k = 0; k < kDictProbes; ++k) {
      if (atomicCAS(&c.tag[e], 0u, 1u) == 0u) {
        c.id[e] = (uint32_t)id;
        c.w0[e] = w0;
        c.w1[e] = w1;
        __hip_atomic_store(&c.tag[e], dict_tag(h, len), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        break;
      }
      e = (e + 1u) & (uint32_t)(kDictCache - 1);
    }
  }
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * kDictBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kDictBlock) {
    const int32_t o0 = offsets[i], o1 = offsets[i + 1];
    const uint32_t len = o1 > o0 ? (uint32_t)(o1 - o0) : 0u;
    const uint8_t* str = data + o0;
    uint64_t h, w0 = 0, w1 = 0, id = ~0ull;
    uint32_t free_at = (uint32_t)kDictCache;
    const bool small = len <= 16u;
    if (small) {
      dict_words16(str, len, w0, w1);
      h = dict_hash16(len, w0, w1);
      id = dict_cache_find(c, h, len, w0, w1, free_at);
    } else {
      h = hash_bytes(str, len);
    }
    if (id == ~0ull) {
      id = dict_global_lookup(D, str, len, h);
      if (small && id != ~0ull) dict_cache_insert(c, free_at, h, len, w0, w1, id);
    }
    ids[i] = id;
  }
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

hipError_t launch_dict_encode(const int32_t* offsets, const uint8_t* data, int64_t n, const DevDict& D, uint64_t n_known, uint64_t* ids,
                              hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_GATHER_UTF8, s, 0);
  // persistent workgroups, one per CU
  const int64_t blocks = (n + kDictBlock - 1) / kDictBlock;
  const int grid = (int)std::min<int64_t>(blocks, (int64_t)device_cu_count());
  hipLaunchKernelGGL(k_dict_encode, dim3((unsigned)grid), dim3(kDictBlock), 0, s, offsets, data, n, D, n_known, ids);
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
