// ubench2.hip -- store-pattern micro-benchmarks (what bounds the routing stores of k_partition?).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench2.hip -o tools/ubench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)
__device__ __forceinline__ uint32_t fmix32(uint32_t x) { x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; return x ^ (x >> 16); }

// every `team` consecutive lanes write one contiguous run of team*16 B at a random (run-aligned) place;
// only lanes with (lane % act_mod) == 0 teams are active (act_mod = 1: all)
template <int TEAM>
__global__ __launch_bounds__(1024) void k_scatter(ulonglong2* dst, uint64_t n_runs_mask, int64_t rows_per_thread, int act_mod) {
  const uint32_t tid = blockIdx.x * 1024u + threadIdx.x;
  const uint32_t team_id = tid / TEAM, in_team = tid % TEAM;
  const bool active = ((threadIdx.x & 63) / TEAM) % act_mod == 0;
  for (int64_t i = 0; i < rows_per_thread; ++i) {
    const uint64_t run = fmix32(team_id * 0x9E3779B9u + (uint32_t)i * 0x7F4A7C15u) & n_runs_mask;
    if (active) dst[run * TEAM + in_team] = make_ulonglong2(run, (uint64_t)i);
  }
}
template <typename F>
static double time_ms(F&& f, int reps = 3) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}
template <int TEAM>
static void run(ulonglong2* buf, int cus, int64_t foot_mb, int act_mod) {
  const int64_t rpt = 64;                       // row-stores per ACTIVE thread
  const int64_t threads = (int64_t)cus * 1024;
  const int64_t active_rows = threads * rpt / act_mod;
  const uint64_t n_runs = (uint64_t)foot_mb * (1 << 20) / (16 * TEAM);
  double ms = time_ms([&] { hipLaunchKernelGGL(k_scatter<TEAM>, dim3(cus), dim3(1024), 0, 0, buf, n_runs - 1, rpt, act_mod); });
  printf("{\"bench\":\"scatter16\",\"team_lanes\":%d,\"run_bytes\":%d,\"footprint_mb\":%lld,\"active_1_in\":%d,\"g_rows_per_s\":%.2f,\"gbps\":%.1f}\n",
         TEAM, TEAM * 16, (long long)foot_mb, act_mod, active_rows / ms * 1e-6, active_rows * 16.0 / ms * 1e-6);
}
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  ulonglong2* buf; CK(hipMalloc(&buf, 1ll << 30));
  CK(hipMemset(buf, 0, 1ll << 30));
  for (int64_t mb : {8, 64, 1024}) {
    run<1>(buf, cus, mb, 1); run<1>(buf, cus, mb, 5);
    run<2>(buf, cus, mb, 1); run<4>(buf, cus, mb, 1); run<8>(buf, cus, mb, 1); run<16>(buf, cus, mb, 1); run<64>(buf, cus, mb, 1);
  }
  return 0;
}
