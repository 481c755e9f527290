// ubench7.hip -- ONE column read by a persistent grid (config 2's access pattern): what do load width, row groups per trip,
// waves per CU and the nt hint give?  k_reduce reaches 5.5 TB/s, two lock-step columns reach 7.2 (ubench4).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench7 ubench7.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t v4 __attribute__((ext_vector_type(4)));
typedef uint32_t v2 __attribute__((ext_vector_type(2)));

template <int U, int W16, int NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_read1(const uint8_t* __restrict__ a, int64_t n_bytes, uint64_t* out) {
  constexpr int LB = W16 ? 16 : 8;                 // bytes per lane per load
  const int64_t gtid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
  const int64_t wave = gtid >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = ((int64_t)gridDim.x * BLOCK) >> 6;
  const int64_t trip_bytes = (int64_t)U * 64 * LB;
  const int64_t n_trips = n_bytes / trip_bytes;
  uint32_t acc = 0;
  v4 nx[U];
  auto load = [&](int64_t t, v4 (&x)[U]) {
    if (t >= n_trips) t = n_trips - 1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint8_t* p = a + t * trip_bytes + (int64_t)u * 64 * LB + lane * LB;
      if (W16) x[u] = NT ? __builtin_nontemporal_load((const v4*)p) : *(const v4*)p;
      else { v2 y = NT ? __builtin_nontemporal_load((const v2*)p) : *(const v2*)p; x[u] = (v4){y.x, y.y, 0u, 0u}; }
    }
  };
  load(wave, nx);
  for (int64_t t = wave; t < n_trips; t += n_waves) {
    v4 c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) c[u] = nx[u];
    load(t + n_waves, nx);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += c[u].x ^ c[u].y ^ c[u].z ^ c[u].w;
  }
  if (acc == 0x12345u) out[0] = acc;
}

int main() {
  const int64_t n_bytes = 1ll << 30;  // 2^27 rows of 8 bytes
  uint8_t* a; uint64_t* out;
  CK(hipMalloc((void**)&a, 4 * n_bytes)); CK(hipMemset(a, 1, 4 * n_bytes)); CK(hipMalloc((void**)&out, 64));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](auto launch) {
    std::vector<float> ms;
    for (int r = 0; r < 11; ++r) {
      CK(hipEventRecord(e0, s)); launch(r); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1)); if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end()); return (double)ms[ms.size() / 2];
  };
#define RUN(U, W16, NT, BLOCK, BPC)                                                                                             \
  {                                                                                                                             \
    double t = time_it([&](int r) { hipLaunchKernelGGL((k_read1<U, W16, NT, BLOCK>), dim3(256 * BPC), dim3(BLOCK), 0, s, a + (int64_t)(r % 4) * n_bytes, n_bytes, out); }); \
    printf("{\"U\":%d,\"bytes_per_lane\":%d,\"nt\":%d,\"block\":%d,\"blocks_per_cu\":%d,\"waves_per_cu\":%d,\"ms\":%.4f,\"GBps\":%.0f}\n", U, W16 ? 16 : 8, NT, BLOCK, BPC, \
           BLOCK / 64 * BPC, t, n_bytes / t * 1e-6);                                                                           \
    fflush(stdout);                                                                                                             \
  }
  RUN(8, 0, 1, 256, 8) RUN(8, 0, 0, 256, 8) RUN(8, 0, 1, 256, 4) RUN(8, 0, 1, 256, 2) RUN(4, 0, 1, 256, 8) RUN(16, 0, 1, 256, 4)
  RUN(8, 0, 1, 512, 1) RUN(8, 0, 1, 1024, 1) RUN(16, 0, 1, 512, 1)
  RUN(4, 1, 1, 256, 8) RUN(4, 1, 0, 256, 8) RUN(4, 1, 1, 256, 4) RUN(8, 1, 1, 256, 4) RUN(8, 1, 1, 256, 2) RUN(4, 1, 1, 512, 1) RUN(8, 1, 1, 512, 1) RUN(2, 1, 1, 256, 8)
  return 0;
}
