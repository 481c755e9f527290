// ubench_overlap.hip -- can an LDS/VALU-bound kernel hide under an HBM-bound one on MI355X, and what does it take?
// WRITTEN AT THE END OF ROUND 1 WITHOUT GPU BUDGET LEFT: compiled (hipcc), NOT YET RUN.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_overlap.hip -o tools/ubench_overlap && tools/ubench_overlap
//
// Background: one step of the headline query is 15 x (pass 1: 0.217 ms, HBM-bound, VALU ~45 %,
// ~100 KB of LDS, 1024 lanes per CU) + 15 x (pass 2: 0.08 ms, VALU/LDS-bound, no HBM traffic to speak of, ~132 KB of
// LDS, 1024 lanes per CU).  The two are bound by different resources, so pass 2 of batch i could run under pass 1 of
// batch i + 1 -- but only if a CU can hold a workgroup of each at the same time: 160 KB of LDS, 32 wave slots.
//
// Model kernels:
//   A "stream"  persistent grid (one workgroup of 1024 lanes per CU), non-temporal 16-byte loads over `bytes` of HBM,
//               a few VALU ops per load, holds lds_a bytes of (otherwise unused) dynamic LDS
//   B "probe"   one workgroup per CU of `threads_b` lanes, `iters` rounds of hash + LDS read + LDS f64 atomic in a table
//               that fills lds_b bytes; no global traffic
// For a list of (lds_a, lds_b, threads_b) the program times A alone, B alone, and A + B launched back to back on two
// streams (B first, so that the persistent A does not occupy every CU before B's workgroups are placed -- and the other
// order too).  If A + B takes ~max(A, B) the pair shares CUs; ~A + B means they serialise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

typedef uint64_t u64x2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void k_stream(const u64x2_t* __restrict__ src, size_t n16, uint64_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_a[];
  if (threadIdx.x == 0) lds_a[0] = 1;  // keep the allocation
  uint64_t acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // 4 loads in flight per lane
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const u64x2_t a = __builtin_nontemporal_load(src + i);
    const u64x2_t b = __builtin_nontemporal_load(src + i + stride);
    const u64x2_t c = __builtin_nontemporal_load(src + i + 2 * stride);
    const u64x2_t d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc += (a.x ^ (a.y >> 7)) + (b.x ^ (b.y >> 7)) + (c.x ^ (c.y >> 7)) + (d.x ^ (d.y >> 7));
  }
  for (; i < n16; i += stride) {
    const u64x2_t a = __builtin_nontemporal_load(src + i);
    acc += a.x ^ (a.y >> 7);
  }
  if (acc == 0x1234567ull) out[0] = acc + lds_a[0];  // practically never: keeps the loads alive
}

__global__ void k_probe(int iters, uint32_t slots, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t raw[];
  uint64_t* keys = (uint64_t*)raw;                       // [slots]
  double* accs = (double*)(raw + (size_t)slots * 8);     // [slots]
  for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) {
    keys[i] = i;
    accs[i] = 0.0;
  }
  __syncthreads();
  uint32_t x = blockIdx.x * 9781u + threadIdx.x * 6271u + 1u;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    uint32_t h = (x ^ (x >> 15)) * 0x85EBCA6Bu;
    h ^= h >> 13;
    const uint32_t g = (h >> 8) & (slots / 4 - 1);
    const ulonglong2 ka = *(const ulonglong2*)&keys[g * 4];
    const ulonglong2 kb = *(const ulonglong2*)&keys[g * 4 + 2];
    const uint64_t want = g * 4 + (h & 3u);
    const uint32_t j = ka.x == want ? 0u : ka.y == want ? 1u : kb.x == want ? 2u : 3u;
    unsafeAtomicAdd(&accs[g * 4 + j], 1.0);
  }
  __syncthreads();
  double s = 0;
  for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) s += accs[i];
  if (s < 0) out[blockIdx.x] = s;  // never: keeps the loop alive
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
};

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 30);  // pass 1 reads 1.07 GB per launch
  const int iters = argc > 2 ? atoi(argv[2]) : 256;                            // pass 2: 256 rows per lane per launch
  void* src;
  uint64_t* out_a;
  double* out_b;
  CK(hipMalloc(&src, bytes));
  CK(hipMemset(src, 1, bytes));
  CK(hipMalloc(&out_a, 8));
  CK(hipMalloc(&out_b, sizeof(double) * cus));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  CK(hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1, ea, eb;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&ea));
  CK(hipEventCreate(&eb));

  struct Case { int lds_a_kb, lds_b_kb, threads_b; };
  const Case cases[] = {{100, 132, 1024},   // today's footprints: cannot share a CU
                        {64, 64, 1024},     // both halved: 128 KB, 32 waves -> fits
                        {64, 64, 512},      // and a smaller pass-2 workgroup
                        {24, 132, 1024},    // pass 1 with 8-row rings only, pass 2 as it is: 156 KB
                        {100, 32, 512}};    // pass 1 as it is, pass 2 on quarter blocks
  for (const Case& c : cases) {
    const size_t lds_a = (size_t)c.lds_a_kb * 1024, lds_b = (size_t)c.lds_b_kb * 1024;
    const uint32_t slots = (uint32_t)(lds_b / 16);  // keys + accs; a power of two for 32/64/128 KB, rounded down otherwise
    uint32_t p2 = 1;
    while (p2 * 2 <= slots) p2 *= 2;
    auto launch_a = [&](hipStream_t s) { hipLaunchKernelGGL(k_stream, dim3(cus), dim3(1024), lds_a, s, (const u64x2_t*)src, bytes / 16, out_a); };
    auto launch_b = [&](hipStream_t s) { hipLaunchKernelGGL(k_probe, dim3(cus), dim3(c.threads_b), lds_b, s, iters * (1024 / c.threads_b), p2, out_b); };
    float t_a = 0, t_b = 0, t_ab = 0, t_ba = 0;
    for (int rep = 0; rep < 3; ++rep) {  // the last repetition counts
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, sa)); launch_a(sa); CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_a, e0, e1));
      CK(hipEventRecord(e0, sb)); launch_b(sb); CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_b, e0, e1));
      // B first, then A, concurrently: wall time from the first launch to the later of the two completions
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, sb)); CK(hipStreamWaitEvent(sa, e0, 0));
      launch_b(sb); launch_a(sa);
      CK(hipEventRecord(ea, sa)); CK(hipEventRecord(eb, sb)); CK(hipEventSynchronize(ea)); CK(hipEventSynchronize(eb));
      float x = 0, y = 0;
      CK(hipEventElapsedTime(&x, e0, ea)); CK(hipEventElapsedTime(&y, e0, eb));
      t_ba = x > y ? x : y;
      // A first, then B
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, sa)); CK(hipStreamWaitEvent(sb, e0, 0));
      launch_a(sa); launch_b(sb);
      CK(hipEventRecord(ea, sa)); CK(hipEventRecord(eb, sb)); CK(hipEventSynchronize(ea)); CK(hipEventSynchronize(eb));
      CK(hipEventElapsedTime(&x, e0, ea)); CK(hipEventElapsedTime(&y, e0, eb));
      t_ab = x > y ? x : y;
    }
    printf("{\"bench\":\"overlap\",\"lds_a_kb\":%d,\"lds_b_kb\":%d,\"threads_b\":%d,\"stream_alone_ms\":%.4f,\"stream_GBps\":%.0f,"
           "\"probe_alone_ms\":%.4f,\"probe_then_stream_ms\":%.4f,\"stream_then_probe_ms\":%.4f,\"sum_ms\":%.4f}\n",
           c.lds_a_kb, c.lds_b_kb, c.threads_b, t_a, bytes / t_a * 1e-6, t_b, t_ba, t_ab, t_a + t_b);
  }
  return 0;
}
