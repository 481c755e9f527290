// ubench.hip -- standalone micro-benchmarks that decide the hash-aggregate strategy on MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench.hip -o tools/ubench
// Prints one JSON line per measurement.  Not part of the product path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_fill(double* p, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) p[i] = (double)(mix64(i) >> 44) * 0x1.0p-10;
}

// streaming read, 8 B per lane per load
__global__ __launch_bounds__(256) void k_read8(const double* __restrict__ p, int64_t n, double* out) {
  double s = 0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) s += p[i];
  if (s == 1.2345e300) out[0] = s;
}
// streaming read, 16 B per lane per load, 4 loads in flight
__global__ __launch_bounds__(256) void k_read16(const double2* __restrict__ p, int64_t n2, double* out) {
  double s = 0;
  const int64_t stride = gridDim.x * 256ll;
  int64_t i = blockIdx.x * 256ll + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    double2 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
  }
  for (; i < n2; i += stride) { double2 a = p[i]; s += a.x + a.y; }
  if (s == 1.2345e300) out[0] = s;
}
// two 8-byte columns read together (k, v) -- the group-by input pattern
__global__ __launch_bounds__(256) void k_read2col(const int64_t* __restrict__ k, const double* __restrict__ v, int64_t n, double* out) {
  double s = 0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) s += v[i] + (double)(k[i] & 1);
  if (s == 1.2345e300) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write8(double* __restrict__ p, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) p[i] = (double)i;
}

// random atomics on a table of `mask+1` doubles
template <int SCOPE>  // 0 agent, 1 workgroup (executes in the local XCD's L2: NOT coherent across XCDs)
__global__ __launch_bounds__(256) void k_atomic_f64(double* tab, uint64_t mask, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const uint64_t h = mix64((uint64_t)i) & mask;
    if (SCOPE == 0) unsafeAtomicAdd(&tab[h], 1.0);
    else __hip_atomic_fetch_add(&tab[h], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
__global__ __launch_bounds__(256) void k_atomic_u64(unsigned long long* tab, uint64_t mask, int64_t n) {
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    atomicAdd(&tab[mix64((uint64_t)i) & mask], 1ull);
}
// probe (8-byte sc1 load) + atomic add on a second plane: the steady-state group-by update
__global__ __launch_bounds__(256) void k_probe_atomic(const uint64_t* keys, double* accs, uint64_t mask, int64_t n, uint64_t* sink) {
  uint64_t acc = 0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const uint64_t h = mix64((uint64_t)i) & mask;
    acc += __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsafeAtomicAdd(&accs[h], 1.0);
  }
  if (acc == 0x1234567) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_random_read(const uint64_t* tab, uint64_t mask, int64_t n, uint64_t* sink) {
  uint64_t acc = 0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    acc += tab[mix64((uint64_t)i) & mask];
  if (acc == 0x1234567) sink[0] = acc;
}
// LDS atomics: random slots in an 8192-entry (64 KB) table per workgroup
__global__ __launch_bounds__(256) void k_lds_atomic(int64_t n, double* sink, int slots) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < slots; i += 256) lds[i] = 0;
  __syncthreads();
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll)
    unsafeAtomicAdd(&lds[mix64((uint64_t)i) & (uint64_t)(slots - 1)], 1.0);
  __syncthreads();
  if (threadIdx.x == 0 && lds[0] == 1.2345e300) sink[0] = lds[0];
}
// LDS probe (cmpst) + add: the LDS-table update
__global__ __launch_bounds__(256) void k_lds_probe(int64_t n, double* sink, int slots) {
  extern __shared__ double lds[];
  unsigned long long* keys = (unsigned long long*)lds;
  double* accs = lds + slots;
  for (int i = threadIdx.x; i < slots; i += 256) { keys[i] = ~0ull; accs[i] = 0; }
  __syncthreads();
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
    const unsigned long long key = mix64((uint64_t)i) & (uint64_t)(slots / 2 - 1);
    int s = (int)(mix64(key) & (uint64_t)(slots - 1));
    for (int p = 0; p < 8; ++p) {
      const unsigned long long k = keys[s];
      if (k == key) break;
      if (k == ~0ull) { const unsigned long long old = atomicCAS(&keys[s], ~0ull, key); if (old == ~0ull || old == key) break; }
      s = (s + 1) & (slots - 1);
    }
    unsafeAtomicAdd(&accs[s], 1.0);
  }
  __syncthreads();
  if (threadIdx.x == 0 && accs[0] == 1.2345e300) sink[0] = accs[0];
}

template <typename F>
static double time_ms(F&& f, int reps = 3) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a, 0));
    f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"bench\":\"device\",\"name\":\"%s\",\"arch\":\"%s\",\"cus\":%d,\"clock_mhz\":%d,\"l2_bytes\":%d}\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.clockRate / 1000, prop.l2CacheSize);
  const int cus = prop.multiProcessorCount;
  const int64_t N = 1ll << 29;  // 4 GiB of doubles
  double *buf, *buf2, *sink;
  CK(hipMalloc(&buf, N * 8));
  CK(hipMalloc(&buf2, N * 8));
  CK(hipMalloc(&sink, 4096));
  hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, buf, N);
  hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(256), 0, 0, buf2, N);
  CK(hipDeviceSynchronize());
  for (int per_cu : {2, 4, 8, 16}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_read8, dim3(cus * per_cu), dim3(256), 0, 0, buf, N, sink); });
    printf("{\"bench\":\"read8\",\"blocks_per_cu\":%d,\"gbps\":%.1f}\n", per_cu, N * 8.0 / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read16, dim3(cus * per_cu), dim3(256), 0, 0, (const double2*)buf, N / 2, sink); });
    printf("{\"bench\":\"read16x4\",\"blocks_per_cu\":%d,\"gbps\":%.1f}\n", per_cu, N * 8.0 / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read2col, dim3(cus * per_cu), dim3(256), 0, 0, (const int64_t*)buf, buf2, N, sink); });
    printf("{\"bench\":\"read2col\",\"blocks_per_cu\":%d,\"gbps\":%.1f}\n", per_cu, N * 16.0 / ms * 1e-6);
  }
  {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_write8, dim3(cus * 8), dim3(256), 0, 0, buf2, N); });
    printf("{\"bench\":\"write8\",\"gbps\":%.1f}\n", N * 8.0 / ms * 1e-6);
  }
  // cache residency: re-read a small buffer (L2 / Infinity Cache) right after writing it
  for (int64_t mb : {4, 32, 64, 128, 192, 256, 512, 1024}) {
    const int64_t n = mb * (1 << 20) / 8;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_read8, dim3(cus * 8), dim3(256), 0, 0, buf, n, sink); }, 5);
    printf("{\"bench\":\"reread\",\"mb\":%lld,\"gbps\":%.1f}\n", (long long)mb, n * 8.0 / ms * 1e-6);
    ms = time_ms([&] {
      hipLaunchKernelGGL(k_write8, dim3(cus * 8), dim3(256), 0, 0, buf2, n);
      hipLaunchKernelGGL(k_read8, dim3(cus * 8), dim3(256), 0, 0, buf2, n, sink);
    }, 5);
    printf("{\"bench\":\"write_then_read\",\"mb\":%lld,\"gbps_each_way\":%.1f}\n", (long long)mb, 2 * n * 8.0 / ms * 1e-6);
  }
  const int64_t OPS = 1ll << 28;
  for (int lg : {10, 14, 17, 19, 21, 23, 25}) {
    const uint64_t slots = 1ull << lg;
    CK(hipMemset(buf2, 0, slots * 8));
    double ms = time_ms([&] { hipLaunchKernelGGL(k_atomic_f64<0>, dim3(cus * 8), dim3(256), 0, 0, buf2, slots - 1, OPS); });
    printf("{\"bench\":\"atomic_add_f64_agent\",\"slots_log2\":%d,\"table_mb\":%.3f,\"gops\":%.2f}\n", lg, slots * 8.0 / 1048576, OPS / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_atomic_f64<1>, dim3(cus * 8), dim3(256), 0, 0, buf2, slots - 1, OPS); });
    printf("{\"bench\":\"atomic_add_f64_workgroup_scope\",\"slots_log2\":%d,\"gops\":%.2f}\n", lg, OPS / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_atomic_u64, dim3(cus * 8), dim3(256), 0, 0, (unsigned long long*)buf2, slots - 1, OPS); });
    printf("{\"bench\":\"atomic_add_u64_agent\",\"slots_log2\":%d,\"gops\":%.2f}\n", lg, OPS / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_random_read, dim3(cus * 8), dim3(256), 0, 0, (const uint64_t*)buf2, slots - 1, OPS, (uint64_t*)sink); });
    printf("{\"bench\":\"random_read8\",\"slots_log2\":%d,\"gops\":%.2f}\n", lg, OPS / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_probe_atomic, dim3(cus * 8), dim3(256), 0, 0, (const uint64_t*)buf, buf2, slots - 1, OPS, (uint64_t*)sink); });
    printf("{\"bench\":\"probe_plus_atomic\",\"slots_log2\":%d,\"gops\":%.2f}\n", lg, OPS / ms * 1e-6);
  }
  for (int slots : {64, 1024, 8192}) {
    double ms = time_ms([&] { hipLaunchKernelGGL(k_lds_atomic, dim3(cus * 2), dim3(256), slots * 8, 0, OPS, sink, slots); });
    printf("{\"bench\":\"lds_atomic_add_f64\",\"slots\":%d,\"gops\":%.2f}\n", slots, OPS / ms * 1e-6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_lds_probe, dim3(cus * 2), dim3(256), slots * 16, 0, OPS, sink, slots); });
    printf("{\"bench\":\"lds_probe_cas_add\",\"slots\":%d,\"gops\":%.2f}\n", slots, OPS / ms * 1e-6);
  }
  return 0;
}
