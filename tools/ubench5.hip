// ubench5.hip -- what does the WRITE side of pass 1 cost the memory system, and which chunk geometry is cheapest?
// ubench4 showed: two nt read streams alone reach 7.2 TB/s, the same scan that also stores 192-byte chunks into 65536 append
// streams (256 workgroups x 256 partitions) falls to 3.8 TB/s of traffic when every row is routed -- below what the real
// dense pass 1 reaches with its whole LDS protocol (4.7).  So the dense pass 1 may be bound by its write PATTERN, not by its
// instruction count.  Here readers and writers are different waves of a persistent 1024-lane workgroup (as in k_partition_ws):
//   waves 0..7   read two columns (nt loads, 4 row groups per trip, one trip ahead);
//   waves 8..15  append chunks of CH rows x RB bytes to the workgroup's 256 regions: the region of a chunk is pseudo-random,
//                its position the region's cursor (one LDS atomic per chunk), so every stream is written sequentially.
// Prints, per geometry: kernel time for 2^26 rows scanned with `sel` of them routed, traffic GB/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench5 ubench5.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Geo {
  uint32_t chunk_lanes;   // lanes that write one chunk (rows per chunk)
  uint32_t row_bytes;     // 12 or 16
  uint32_t region_bytes;  // distance between the regions of one workgroup
  uint32_t chunks_per_wave;  // chunks each writer wave appends
  uint32_t nt_store;
  uint32_t read_groups;   // 64-row groups to scan (0: writers only)
  uint32_t parts;         // regions per workgroup
  uint32_t wrap;          // > 0: a region is a ring of this many chunks (the scratch stays small enough for the 256 MB Infinity Cache)
};

__global__ __launch_bounds__(1024) void k_rw(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, uint8_t* regions, uint64_t wg_bytes, Geo g, uint64_t* out) {
  __shared__ uint32_t cursor[1024];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < g.parts; i += 1024) cursor[i] = 0;
  __syncthreads();
  if (wave < 8) {
    if (g.read_groups == 0) return;
    constexpr int U = 4;
    const int64_t n_groups = g.read_groups;
    const int64_t wave_global = (int64_t)blockIdx.x * 8 + wave;
    const int64_t n_waves = (int64_t)gridDim.x * 8;
    uint64_t acc = 0;
    uint64_t na[U], nb[U];
    auto load = [&](int64_t w0, uint64_t (&xa)[U], uint64_t (&xb)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int64_t gg = w0 + u;
        if (gg >= n_groups) gg = n_groups - 1;
        xa[u] = __builtin_nontemporal_load(a + gg * 64 + lane);
        xb[u] = __builtin_nontemporal_load(b + gg * 64 + lane);
      }
    };
    load(wave_global * U, na, nb);
    for (int64_t w0 = wave_global * U; w0 < n_groups; w0 += n_waves * U) {
      uint64_t ca[U], cb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { ca[u] = na[u]; cb[u] = nb[u]; }
      load(w0 + n_waves * U, na, nb);
#pragma unroll
      for (int u = 0; u < U; ++u) acc += ca[u] ^ (cb[u] >> 3);
    }
    if (acc == 0x1234567ull) out[0] = acc;
  } else {
    uint8_t* my = regions + (uint64_t)blockIdx.x * wg_bytes;
    const uint32_t per_inst = 64 / g.chunk_lanes;          // chunks one store instruction writes
    const uint32_t sub = (uint32_t)lane / g.chunk_lanes;    // this lane's chunk within the instruction
    const uint32_t rr = (uint32_t)lane % g.chunk_lanes;
    const uint32_t chunk_bytes = g.chunk_lanes * g.row_bytes;
    uint32_t x = (uint32_t)(blockIdx.x * 977u + wave * 131u + 12345u);
    for (uint32_t c0 = 0; c0 < g.chunks_per_wave; c0 += per_inst) {
      // one pseudo-random region per chunk (the same value in the chunk's lanes)
      x = x * 1664525u + 1013904223u;
      const uint32_t part = ((x >> 8) + sub * 0x9E3779B1u) % g.parts;
      uint32_t pos = 0;
      if (rr == 0) pos = atomicAdd(&cursor[part], 1u);
      pos = __shfl(pos, (int)(sub * g.chunk_lanes), 64);
      if (g.wrap) pos %= g.wrap;
      uint8_t* o = my + (uint64_t)part * g.region_bytes + (uint64_t)pos * chunk_bytes + rr * g.row_bytes;
      if ((uint64_t)(pos + 1) * chunk_bytes > g.region_bytes) continue;  // (a region that is full: skew of the generator)
      if (g.row_bytes == 12) {
        uint32_t* o32 = (uint32_t*)o;
        if (g.nt_store) { __builtin_nontemporal_store(x, o32); __builtin_nontemporal_store(pos, o32 + 1); __builtin_nontemporal_store(part, o32 + 2); }
        else { o32[0] = x; o32[1] = pos; o32[2] = part; }
      } else {
        uint4 v = make_uint4(x, pos, part, rr);
        if (g.nt_store) __builtin_nontemporal_store(v.x, (uint32_t*)o), __builtin_nontemporal_store(v.y, (uint32_t*)o + 1), __builtin_nontemporal_store(v.z, (uint32_t*)o + 2), __builtin_nontemporal_store(v.w, (uint32_t*)o + 3);
        else *(uint4*)o = v;
      }
    }
  }
}

int main() {
  const int64_t rows = 1ll << 26;
  const int64_t col_rows = 1ll << 28;
  uint64_t *a, *b, *out;
  CK(hipMalloc((void**)&a, col_rows * 8)); CK(hipMalloc((void**)&b, col_rows * 8)); CK(hipMalloc((void**)&out, 64));
  CK(hipMemset(a, 1, col_rows * 8)); CK(hipMemset(b, 2, col_rows * 8));
  const uint64_t reg_total = 6ull << 30;
  uint8_t* regions; CK(hipMalloc((void**)&regions, reg_total));
  CK(hipMemset(regions, 0, reg_total));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { const char* name; uint32_t lanes, rb, pad, nt; double sel; int read; uint32_t parts; uint32_t wrap = 0; };
  std::vector<Case> cases = {
      {"read_only", 16, 12, 0, 0, 0.0, 1, 256},
      // full-line chunks: 8 lanes x 16 B = one 128-byte line; 16 x 16 = two lines; 4 lanes x 16 B = 64 bytes (half a line)
      {"write_only_dense_64B_rows16", 4, 16, 0, 0, 0.75, 0, 256},
      {"write_only_dense_128B_rows16", 8, 16, 0, 0, 0.75, 0, 256},
      {"write_only_dense_256B_rows16_same_bytes", 16, 16, 0, 0, 0.75, 0, 256},
      {"dense_128B_rows16_same_bytes", 8, 16, 0, 0, 0.75, 1, 256},
      {"dense_256B_rows16_same_bytes", 16, 16, 0, 0, 0.75, 1, 256},
      {"sel20_256B_rows16_same_bytes", 16, 16, 0, 0, 0.15, 1, 256},
      {"sel20_128B_rows16_same_bytes", 8, 16, 0, 0, 0.15, 1, 256},
      {"write_only_dense_192B", 16, 12, 0, 0, 1.0, 0, 256},
      {"write_only_dense_384B", 32, 12, 0, 0, 1.0, 0, 256},
      {"write_only_dense_768B", 64, 12, 0, 0, 1.0, 0, 256},
      {"write_only_dense_256B_rows16", 16, 16, 0, 0, 1.0, 0, 256},
      {"write_only_dense_1024B_rows16", 64, 16, 0, 0, 1.0, 0, 256},
      {"dense_192B", 16, 12, 0, 0, 1.0, 1, 256},
      {"dense_192B_pad64", 16, 12, 64, 0, 1.0, 1, 256},
      {"dense_192B_pad128", 16, 12, 128, 0, 1.0, 1, 256},
      {"dense_192B_pad4160", 16, 12, 4160, 0, 1.0, 1, 256},
      {"dense_192B_nt", 16, 12, 0, 1, 1.0, 1, 256},
      {"dense_384B", 32, 12, 0, 0, 1.0, 1, 256},
      {"dense_384B_nt", 32, 12, 0, 1, 1.0, 1, 256},
      {"dense_768B", 64, 12, 0, 0, 1.0, 1, 256},
      {"dense_768B_nt", 64, 12, 0, 1, 1.0, 1, 256},
      {"dense_256B_rows16", 16, 16, 0, 0, 1.0, 1, 256},
      {"dense_512B_rows16", 32, 16, 0, 0, 1.0, 1, 256},
      {"dense_192B_128parts", 16, 12, 0, 0, 1.0, 1, 128},
      {"dense_192B_64parts", 16, 12, 0, 0, 1.0, 1, 64},
      {"dense_384B_128parts", 32, 12, 0, 0, 1.0, 1, 128},
      {"sel50_192B", 16, 12, 0, 0, 0.5, 1, 256},
      {"sel50_384B", 32, 12, 0, 0, 0.5, 1, 256},
      {"sel20_192B", 16, 12, 0, 0, 0.2, 1, 256},
      {"sel20_384B", 32, 12, 0, 0, 0.2, 1, 256},
      {"sel20_768B", 64, 12, 0, 0, 0.2, 1, 256},
      // the scratch as a small ring per region: 8 chunks x 192 B x 65536 regions = 100 MB (fits the Infinity Cache), 4 chunks = 50 MB, 2 = 25 MB (fits L2)
      {"write_only_dense_192B_wrap8", 16, 12, 0, 0, 1.0, 0, 256, 8},
      {"write_only_dense_384B_wrap4", 32, 12, 0, 0, 1.0, 0, 256, 4},
      {"dense_192B_wrap8", 16, 12, 0, 0, 1.0, 1, 256, 8},
      {"dense_192B_wrap2", 16, 12, 0, 0, 1.0, 1, 256, 2},
      {"dense_384B_wrap4", 32, 12, 0, 0, 1.0, 1, 256, 4},
      {"dense_384B_wrap16", 32, 12, 0, 0, 1.0, 1, 256, 16},
      {"sel20_192B_wrap8", 16, 12, 0, 0, 0.2, 1, 256, 8},
      {"sel20_384B_wrap4", 32, 12, 0, 0, 0.2, 1, 256, 4},
  };
  for (const Case& c : cases) {
    Geo g;
    g.chunk_lanes = c.lanes; g.row_bytes = c.rb; g.nt_store = c.nt; g.parts = c.parts; g.wrap = c.wrap;
    const uint32_t chunk_bytes = c.lanes * c.rb;
    const double routed_rows_per_wg = c.sel * (double)rows / 256.0;
    const uint32_t chunks_per_wg = (uint32_t)(routed_rows_per_wg / c.lanes);
    g.chunks_per_wave = chunks_per_wg / 8;
    const uint32_t chunks_per_region = c.wrap ? c.wrap : (uint32_t)(1.3 * chunks_per_wg / c.parts) + 8;
    g.region_bytes = chunks_per_region * chunk_bytes + c.pad;
    g.read_groups = c.read ? (uint32_t)(rows / 64) : 0;
    const uint64_t wg_bytes = ((uint64_t)g.region_bytes * c.parts + 255) / 256 * 256;
    if (wg_bytes * 256 > reg_total) { printf("{\"case\":\"%s\",\"error\":\"scratch too small\"}\n", c.name); continue; }
    std::vector<float> ms;
    for (int r = 0; r < 9; ++r) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_rw, dim3(256), dim3(1024), 0, s, a + (int64_t)(r % 3) * rows, b + (int64_t)(r % 3) * rows, regions, wg_bytes, g, out);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    const double t = ms[ms.size() / 2];
    const double rd = c.read ? 16.0 * rows : 0.0;
    const double wr = (double)g.chunks_per_wave * 8 * 256 * chunk_bytes;
    printf("{\"case\":\"%s\",\"ms\":%.4f,\"read_GB\":%.3f,\"write_GB\":%.3f,\"traffic_GBps\":%.0f,\"region_bytes\":%u}\n", c.name, t, rd * 1e-9, wr * 1e-9,
           (rd + wr) / t * 1e-6, g.region_bytes);
    fflush(stdout);
  }
  return 0;
}
