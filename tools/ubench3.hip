// ubench3.hip -- round-5 micro-benchmarks (MI355X).  hipcc --offload-arch=gfx950 -O3 -o tools/ubench3 tools/ubench3.hip
//   A. does the Infinity Cache (256 MB, memory side) hold WRITTEN data?  write X MB, then read the same X MB back:
//      read bandwidth against X.  If reads of a freshly written 64-128 MB buffer run well above the ~6.3 TB/s HBM ceiling,
//      a routing window whose routed rows fit the cache would not pay the pass-2 re-read in HBM traffic.
//   B. the same for read-after-read (the cache's read-allocate behaviour, as a reference).
//   C. LDS atomic throughput per CU: returning 32-bit adds on 256 counters (the tile-sorted pass 1's rank atomics),
//      f64 adds on 8192 random slots (pass 2's accumulator atomics), 16-byte random reads (pass 2's tag groups).
// Output: one JSON line per measurement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__global__ __launch_bounds__(1024) void k_write(uint4* __restrict__ p, int64_t n16, uint32_t tag) {
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 1024) p[i] = make_uint4(tag, (uint32_t)i, tag, 0);
}
__global__ __launch_bounds__(1024) void k_read(const uint4* __restrict__ p, int64_t n16, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 1024) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(1024) void k_read_nt(const uint4* __restrict__ p, int64_t n16, uint32_t* __restrict__ out) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 1024) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p + i);
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(uint32_t iters, uint32_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint64_t mem[8192 + 4096];
  uint32_t* m32 = (uint32_t*)mem;
  for (uint32_t i = threadIdx.x; i < 2 * (8192 + 4096); i += 1024) m32[i] = 0;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      x = x * 1664525u + 1013904223u;
      const uint32_t r = x >> 8;
      if (MODE == 0) acc += atomicAdd(&m32[r & 255u], 1u);                          // returning add, 256 counters
      else if (MODE == 1) atomicAdd((double*)&mem[r & 8191u], 1.0);                 // f64 add, 8192 slots, no return
      else if (MODE == 2) { const uint4 t = *(const uint4*)&m32[(r & 8191u & ~3u)]; acc += t.x ^ t.y ^ t.z ^ t.w; }  // 16-byte random read
      else if (MODE == 3) { m32[r & 8191u] = r; mem[4096 + (r & 8191u)] = r; }      // random 4-byte + 8-byte writes (sorted-buffer scatter)
      else acc += m32[(it * 8 + u) * 1024 + threadIdx.x & 16383u];                  // linear read
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

static float time_kernel(hipEvent_t a, hipEvent_t b) {
  float ms = 0;
  CK(hipEventSynchronize(b));
  CK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&e2));
  uint32_t* out;
  CK(hipMalloc(&out, 64));
  const size_t maxb = (size_t)4096 << 20;
  uint4* buf;
  CK(hipMalloc(&buf, maxb));
  uint4* other;  // flushes the caches between experiments
  CK(hipMalloc(&other, (size_t)1024 << 20));
  const int sizes_mb[] = {16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 2048, 4096};
  for (int si = 0; si < (int)(sizeof(sizes_mb) / sizeof(int)); ++si) {
    const int64_t n16 = ((int64_t)sizes_mb[si] << 20) / 16;
    float w_ms = 1e9f, rw_ms = 1e9f, rr_ms = 1e9f, rnt_ms = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(k_write, dim3(cus), dim3(1024), 0, 0, other, ((int64_t)1024 << 20) / 16, 7u);  // evict
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_write, dim3(cus), dim3(1024), 0, 0, buf, n16, (uint32_t)rep);
      CK(hipEventRecord(e1));
      hipLaunchKernelGGL(k_read, dim3(cus), dim3(1024), 0, 0, buf, n16, out);  // read after write
      CK(hipEventRecord(e2));
      float a = time_kernel(e0, e1), b = time_kernel(e1, e2);
      if (a < w_ms) w_ms = a;
      if (b < rw_ms) rw_ms = b;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_read, dim3(cus), dim3(1024), 0, 0, buf, n16, out);  // read after read
      CK(hipEventRecord(e1));
      b = time_kernel(e0, e1);
      if (b < rr_ms) rr_ms = b;
      hipLaunchKernelGGL(k_write, dim3(cus), dim3(1024), 0, 0, other, ((int64_t)1024 << 20) / 16, 9u);  // evict
      hipLaunchKernelGGL(k_write, dim3(cus), dim3(1024), 0, 0, buf, n16, (uint32_t)rep);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_read_nt, dim3(cus), dim3(1024), 0, 0, buf, n16, out);  // non-temporal read after write
      CK(hipEventRecord(e1));
      b = time_kernel(e0, e1);
      if (b < rnt_ms) rnt_ms = b;
    }
    const double gb = (double)sizes_mb[si] / 1024.0;
    printf("{\"bench\": \"mall\", \"mb\": %d, \"write_tbs\": %.2f, \"read_after_write_tbs\": %.2f, \"read_after_read_tbs\": %.2f, \"nt_read_after_write_tbs\": %.2f}\n",
           sizes_mb[si], gb / w_ms, gb / rw_ms, gb / rr_ms, gb / rnt_ms);
    fflush(stdout);
  }
  // cold reads for reference: a buffer that was evicted by 1 GB of other writes
  {
    const int64_t n16 = ((int64_t)128 << 20) / 16;
    hipLaunchKernelGGL(k_write, dim3(cus), dim3(1024), 0, 0, buf, n16, 1u);
    hipLaunchKernelGGL(k_write, dim3(cus), dim3(1024), 0, 0, other, ((int64_t)1024 << 20) / 16, 7u);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_read, dim3(cus), dim3(1024), 0, 0, buf, n16, out);
    CK(hipEventRecord(e1));
    printf("{\"bench\": \"mall_cold_read_128mb\", \"tbs\": %.2f}\n", 0.125 / time_kernel(e0, e1));
  }
  const uint32_t iters = 4096;
  const char* names[] = {"lds_add_rtn_u32_256", "lds_add_f64_8192", "lds_read_b128_random", "lds_scatter_b32_b64", "lds_read_b32_linear"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_lds<0>, dim3(cus), dim3(1024), 0, 0, iters, out); break;
        case 1: hipLaunchKernelGGL(k_lds<1>, dim3(cus), dim3(1024), 0, 0, iters, out); break;
        case 2: hipLaunchKernelGGL(k_lds<2>, dim3(cus), dim3(1024), 0, 0, iters, out); break;
        case 3: hipLaunchKernelGGL(k_lds<3>, dim3(cus), dim3(1024), 0, 0, iters, out); break;
        default: hipLaunchKernelGGL(k_lds<4>, dim3(cus), dim3(1024), 0, 0, iters, out); break;
      }
      CK(hipEventRecord(e1));
      const float ms = time_kernel(e0, e1);
      if (ms < best) best = ms;
    }
    const double wave_ops = (double)iters * 8 * 16;  // wave-level instructions per CU
    printf("{\"bench\": \"%s\", \"ms\": %.3f, \"ns_per_wave_instruction_per_cu\": %.2f, \"lane_ops_per_us_per_cu\": %.0f}\n", names[mode], best,
           best * 1e6 / wave_ops, wave_ops * 64 / (best * 1e3));
  }
  CK(hipDeviceSynchronize());
  return 0;
}
