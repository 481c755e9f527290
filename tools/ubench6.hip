// ubench6.hip -- what can pass 2's READ pattern reach?  One 1024-lane workgroup per partition; wave w walks the regions of
// producers w, w + 16, ... of its partition (rows_per_region rows of 12 bytes each, contiguous), one 768-byte trip per load
// instruction (dwordx3 per lane), PF trips in flight; no table, no LDS, no atomics: the loads are folded into a checksum.
// Against it: the same bytes read as plain sequential streams.  Prints GB/s per variant.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench6 ubench6.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Lay {
  uint64_t part_stride;  // bytes between the regions of consecutive partitions of one producer
  uint64_t prod_stride;  // bytes between the regions of consecutive producers of one partition
  uint32_t region_bytes; // bytes of rows in a region (a multiple of 768)
  uint32_t n_prod;
};

template <int PF, int NT, int WIDE>  // WIDE: 2 = two consecutive trips per load pair (1536 bytes per step)
__global__ __launch_bounds__(1024) void k_walk(const uint8_t* __restrict__ rows, Lay L, uint64_t* out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x;
  const uint8_t* part = rows + (uint64_t)p * L.part_stride;
  uint32_t acc = 0;
  const uint32_t trips_per_region = L.region_bytes / 768;
  const uint32_t n_mine = (L.n_prod + 15 - wave) / 16;
  const uint32_t total = n_mine * trips_per_region;
  struct R3 { uint32_t x, y, z; };
  R3 buf[PF];
  auto addr = [&](uint32_t t) -> const uint32_t* {
    const uint32_t reg = t / trips_per_region, tr = t % trips_per_region;
    return (const uint32_t*)(part + (uint64_t)(wave + 16 * reg) * L.prod_stride + (uint64_t)tr * 768 + lane * 12);
  };
  auto ld = [&](uint32_t t, R3& r) {
    const uint32_t* a = addr(t < total ? t : 0);
    if (NT) { r.x = __builtin_nontemporal_load(a); r.y = __builtin_nontemporal_load(a + 1); r.z = __builtin_nontemporal_load(a + 2); }
    else { r.x = a[0]; r.y = a[1]; r.z = a[2]; }
  };
#pragma unroll
  for (int d = 0; d < PF; ++d) ld(d, buf[d]);
  for (uint32_t t = 0; t < total; t += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const R3 r = buf[d];
      ld(t + PF + d, buf[d]);
      acc += r.x ^ (r.y >> 3) ^ r.z;
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}

// the same bytes as ONE sequential stream: block b reads its contiguous share, 16 bytes per lane
template <int NT>
__global__ __launch_bounds__(1024) void k_seq(const uint4* __restrict__ rows, uint64_t n16, uint64_t* out) {
  uint32_t acc = 0;
  const uint64_t per = n16 / gridDim.x;
  const uint4* p = rows + (uint64_t)blockIdx.x * per;
  for (uint64_t i = threadIdx.x; i + 3 * 1024 < per; i += 4 * 1024) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const v4* q = (const v4*)p;
    v4 a, b, c, d;
    if (NT) { a = __builtin_nontemporal_load(q + i); b = __builtin_nontemporal_load(q + i + 1024); c = __builtin_nontemporal_load(q + i + 2048); d = __builtin_nontemporal_load(q + i + 3072); }
    else { a = q[i]; b = q[i + 1024]; c = q[i + 2048]; d = q[i + 3072]; }
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345u) out[0] = acc;
}

int main() {
  const uint32_t n_parts = 256, n_prod = 256;
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint64_t* out; CK(hipMalloc((void**)&out, 64));
  const uint64_t scratch = 4ull << 30;
  uint8_t* rows; CK(hipMalloc((void**)&rows, scratch)); CK(hipMemset(rows, 3, scratch));
  auto time_it = [&](auto launch) {
    std::vector<float> ms;
    for (int r = 0; r < 9; ++r) {
      CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end()); return (double)ms[ms.size() / 2];
  };
  struct Case { const char* name; uint32_t region_rows; uint32_t cap_rows; int layout; };  // layout 1 producer-major, 0 partition-major
  const Case cases[] = {
      {"dense_1024rows_prodmajor", 1024, 1408, 1}, {"dense_1024rows_partmajor", 1024, 1408, 0}, {"dense_1024rows_packed", 1024, 1024, 0},
      {"sel20_448rows_prodmajor", 448, 2816, 1},   {"sel20_448rows_partmajor", 448, 2816, 0},   {"sel20_448rows_packed", 448, 448, 0},
  };
  for (const Case& c : cases) {
    Lay L;
    const uint64_t region_cap = (uint64_t)c.cap_rows * 12;
    L.region_bytes = c.region_rows * 12;
    L.n_prod = n_prod;
    if (c.layout == 1) { L.part_stride = region_cap; L.prod_stride = (uint64_t)n_parts * region_cap; }
    else { L.prod_stride = region_cap; L.part_stride = (uint64_t)n_prod * region_cap; }
    const double bytes = (double)n_parts * n_prod * L.region_bytes;
    if ((uint64_t)n_parts * n_prod * region_cap > scratch) { printf("{\"case\":\"%s\",\"error\":\"scratch\"}\n", c.name); continue; }
    double t;
    t = time_it([&] { hipLaunchKernelGGL((k_walk<8, 1, 1>), dim3(n_parts), dim3(1024), 0, s, rows, L, out); });
    printf("{\"case\":\"%s\",\"variant\":\"pf8_nt\",\"ms\":%.4f,\"GBps\":%.0f}\n", c.name, t, bytes / t * 1e-6);
    t = time_it([&] { hipLaunchKernelGGL((k_walk<8, 0, 1>), dim3(n_parts), dim3(1024), 0, s, rows, L, out); });
    printf("{\"case\":\"%s\",\"variant\":\"pf8_plain\",\"ms\":%.4f,\"GBps\":%.0f}\n", c.name, t, bytes / t * 1e-6);
    t = time_it([&] { hipLaunchKernelGGL((k_walk<4, 1, 1>), dim3(n_parts), dim3(1024), 0, s, rows, L, out); });
    printf("{\"case\":\"%s\",\"variant\":\"pf4_nt\",\"ms\":%.4f,\"GBps\":%.0f}\n", c.name, t, bytes / t * 1e-6);
    t = time_it([&] { hipLaunchKernelGGL((k_walk<16, 1, 1>), dim3(n_parts), dim3(1024), 0, s, rows, L, out); });
    printf("{\"case\":\"%s\",\"variant\":\"pf16_nt\",\"ms\":%.4f,\"GBps\":%.0f}\n", c.name, t, bytes / t * 1e-6);
    fflush(stdout);
  }
  {
    const double bytes = 256.0 * 256 * 1024 * 12;
    double t = time_it([&] { hipLaunchKernelGGL((k_seq<1>), dim3(256), dim3(1024), 0, s, (const uint4*)rows, (uint64_t)(bytes / 16), out); });
    printf("{\"case\":\"sequential_805MB\",\"variant\":\"nt\",\"ms\":%.4f,\"GBps\":%.0f}\n", t, bytes / t * 1e-6);
    t = time_it([&] { hipLaunchKernelGGL((k_seq<0>), dim3(256), dim3(1024), 0, s, (const uint4*)rows, (uint64_t)(bytes / 16), out); });
    printf("{\"case\":\"sequential_805MB\",\"variant\":\"plain\",\"ms\":%.4f,\"GBps\":%.0f}\n", t, bytes / t * 1e-6);
  }
  return 0;
}
