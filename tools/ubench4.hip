// ubench4.hip -- does the RELATIVE PLACEMENT of the two columns a scan reads in lock step change its bandwidth?
// Round 5 found +-5 % between identical pass-1 launches "depending on where hipMalloc put the table" and did not explain it.
// This probe reads two streams A[i], B[i] with pass 1's access pattern (persistent 1024-lane workgroups, one per CU; NS
// scanning waves per workgroup, U consecutive 64-row groups per trip, one trip of loads ahead) out of ONE big allocation,
// with B = A + span + delta, and prints GB/s per (base shift, delta).  Build: hipcc --offload-arch=gfx950 -O3 -o ubench4 ubench4.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NS, int U, int NT>
__global__ __launch_bounds__(1024) void k_read2(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, int64_t n_groups, uint64_t* out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (wave >= NS) return;
  const int64_t wave_global = (int64_t)blockIdx.x * NS + wave;
  const int64_t n_waves = (int64_t)gridDim.x * NS;
  uint64_t acc = 0;
  uint64_t na[U], nb[U];
  auto load = [&](int64_t w0, uint64_t (&xa)[U], uint64_t (&xb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t g = w0 + u;
      if (g >= n_groups) g = n_groups - 1;
      if (NT) {
        xa[u] = __builtin_nontemporal_load(a + g * 64 + lane);
        xb[u] = __builtin_nontemporal_load(b + g * 64 + lane);
      } else {
        xa[u] = a[g * 64 + lane];
        xb[u] = b[g * 64 + lane];
      }
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
}

// the same scan that also WRITES one 12-byte row per passing row (selectivity 1/5) into per-(workgroup, partition) regions
// in 192-byte runs -- the shape of pass 1's write side without any of its LDS protocol: lane groups of 16 write a chunk.
template <int NS, int U, int DENSE>
__global__ __launch_bounds__(1024) void k_read2_write(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b, int64_t n_groups, uint32_t* regions,
                                                       uint64_t region_words, uint64_t* out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (wave >= NS) return;
  const int64_t wave_global = (int64_t)blockIdx.x * NS + wave;
  const int64_t n_waves = (int64_t)gridDim.x * NS;
  uint64_t acc = 0;
  uint64_t na[U], nb[U];
  auto load = [&](int64_t w0, uint64_t (&xa)[U], uint64_t (&xb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t g = w0 + u;
      if (g >= n_groups) g = n_groups - 1;
      xa[u] = a[g * 64 + lane];
      xb[u] = b[g * 64 + lane];
    }
  };
  load(wave_global * U, na, nb);
  uint32_t* my = regions + (uint64_t)blockIdx.x * 256 * region_words;  // 256 regions per workgroup
  uint32_t chunk = wave;                                                  // this wave's chunk counter (chunks of 16 rows x 12 B = 48 words)
  uint32_t trip = 0;
  for (int64_t w0 = wave_global * U; w0 < n_groups; w0 += n_waves * U) {
    uint64_t ca[U], cb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { ca[u] = na[u]; cb[u] = nb[u]; }
    load(w0 + n_waves * U, na, nb);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += ca[u] ^ (cb[u] >> 3);
    // U groups = 256 rows scanned -> 51 rows routed -> ~3.2 chunks of 16: write 4 chunks on 4 of 5 trips
    // DENSE: every row is routed: 16 chunks per trip (four store instructions of four chunks each)
#pragma unroll
    for (int rep = 0; rep < (DENSE ? 4 : 1); ++rep) {
      if (DENSE || trip % 5 != 4) {
        const uint32_t part = (uint32_t)((wave * 37u + trip * 101u + (lane >> 4) * 59u + rep * 83u) & 255u);
        const uint32_t c = (chunk >> 4) % (uint32_t)(region_words / 48);
        uint32_t* o = my + (uint64_t)part * region_words + (uint64_t)c * 48 + (lane & 15) * 3;
        o[0] = (uint32_t)acc; o[1] = (uint32_t)ca[rep % U]; o[2] = (uint32_t)cb[rep % U];
        chunk += 16;
      }
    }
    ++trip;
  }
  if (acc == 0x1234567ull) out[0] = acc;
}

static double time_launches(void (*launch)(hipStream_t, int), int reps, hipStream_t s) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, s));
    launch(s, r);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

static const uint64_t *gA, *gB;
static uint64_t* gOut;
static uint32_t* gReg;
static int64_t gGroups;
static int64_t gRowsPerLaunch;
static int gMode;
static uint64_t gRegionWords;

static void do_launch(hipStream_t s, int r) {
  const int64_t off = (int64_t)(r % 7) * gRowsPerLaunch;  // walk through the column like consecutive batches
  const uint64_t* a = gA + off;
  const uint64_t* b = gB + off;
  switch (gMode) {
    case 0: hipLaunchKernelGGL((k_read2<8, 4, 0>), dim3(256), dim3(1024), 0, s, a, b, gGroups, gOut); break;
    case 1: hipLaunchKernelGGL((k_read2<16, 4, 0>), dim3(256), dim3(1024), 0, s, a, b, gGroups, gOut); break;
    case 2: hipLaunchKernelGGL((k_read2<8, 4, 1>), dim3(256), dim3(1024), 0, s, a, b, gGroups, gOut); break;
    case 3: hipLaunchKernelGGL((k_read2<8, 8, 0>), dim3(256), dim3(1024), 0, s, a, b, gGroups, gOut); break;
    case 4: hipLaunchKernelGGL((k_read2_write<8, 4, 0>), dim3(256), dim3(1024), 0, s, a, b, gGroups, gReg, gRegionWords, gOut); break;
    case 5: hipLaunchKernelGGL((k_read2_write<8, 4, 1>), dim3(256), dim3(1024), 0, s, a, b, gGroups / 2, gReg, gRegionWords, gOut); break;  // 2^26 rows, all routed
    case 6: hipLaunchKernelGGL((k_read2_write<16, 4, 1>), dim3(256), dim3(1024), 0, s, a, b, gGroups / 2, gReg, gRegionWords, gOut); break;
  }
}

int main(int argc, char** argv) {
  const int64_t rows_per_launch = 1ll << 27;
  const int64_t col_rows = 1ll << 30;               // 8 GiB per column
  const uint64_t span = (uint64_t)col_rows * 8;
  const uint64_t slack = 2ull << 30;
  uint8_t* big;
  CK(hipMalloc((void**)&big, 2 * span + 2 * slack));
  CK(hipMemset(big, 1, 2 * span + 2 * slack));
  CK(hipMalloc((void**)&gOut, 64));
  gRegionWords = 48 * 88;  // 88 chunks per region (17 MB per workgroup: dense 2^26-row windows put ~1024 rows = 64 chunks into a region)
  CK(hipMalloc((void**)&gReg, (size_t)256 * 256 * gRegionWords * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  gGroups = rows_per_launch / 64; gRowsPerLaunch = rows_per_launch;
  printf("{\"bench\":\"alloc\",\"base_mod_2MB\":%llu,\"base_mod_1GB\":%llu}\n", (unsigned long long)((uintptr_t)big & ((2ull << 20) - 1)),
         (unsigned long long)((uintptr_t)big & ((1ull << 30) - 1)));
  const uint64_t deltas[] = {0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1ull << 20, 2ull << 20, 3ull << 20,
                             4ull << 20, 8ull << 20, 16ull << 20, 32ull << 20, 40ull << 20, 64ull << 20, 88ull << 20, 128ull << 20, 256ull << 20, 512ull << 20, 1ull << 30,
                             (1ull << 30) + 4096, (1ull << 30) + (1ull << 20)};
  const uint64_t shifts[] = {0, 40ull << 20, 128ull << 20, 216ull << 20};
  const char* names[] = {"read2_ns8_u4", "read2_ns16_u4", "read2_ns8_u4_nt", "read2_ns8_u8", "read2_write_ns8_u4", "read2_write_dense_2e26_ns8", "read2_write_dense_2e26_ns16"};
  const int modes = argc > 1 ? atoi(argv[1]) : 7;
  for (int mode = 0; mode < modes; ++mode) {
    gMode = mode;
    for (uint64_t sh : shifts) {
      if (mode != 0 && mode != 4 && sh != 0) continue;  // the full shift matrix for the basic pattern and the writing one
      for (uint64_t d : deltas) {
        if (mode != 0 && mode != 4 && !(d == 0 || d == 4096 || d == (1ull << 20) || d == (64ull << 20) || d == (1ull << 30) + 4096)) continue;
        gA = (const uint64_t*)(big + sh);
        gB = (const uint64_t*)(big + sh + span + d);
        time_launches(do_launch, 3, s);
        const double ms = time_launches(do_launch, 14, s);
        printf("{\"bench\":\"%s\",\"shift_mb\":%llu,\"delta\":%llu,\"ms\":%.4f,\"gbps\":%.1f}\n", names[mode], (unsigned long long)(sh >> 20), (unsigned long long)d, ms,
               16.0 * (mode >= 5 ? rows_per_launch / 2 : rows_per_launch) / ms * 1e-6);
        fflush(stdout);
      }
    }
  }
  return 0;
}
