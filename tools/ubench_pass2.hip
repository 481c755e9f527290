// ubench_pass2.hip -- what would make pass 2 (k_partition_agg: find-or-claim in a 128 KB LDS table block) cheaper?
// WRITTEN AT THE END OF ROUND 1 WITHOUT GPU BUDGET LEFT: compiled (hipcc), NOT YET RUN.  First thing to run in round 2:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/ubench_pass2.hip -o tools/ubench_pass2 && tools/ubench_pass2
//
// Setting = pass 2 of the partitioned GROUP BY at the headline configuration: one 1024-lane workgroup per CU, a table
// block of 8192 slots (keys + f64 accumulators = 128 KB of LDS) holding 3906 keys (load 0.48), rows = (key, value) pairs
// whose keys are all present (steady state: the keys were inserted by the first batch).  Rows are generated in registers
// (no HBM traffic), so the numbers isolate the LDS / VALU cost of the lookup + the LDS atomic, which DESIGN.md section 5
// identifies as the bound (VALU issue ~65-75 %, LDS array ~43 % busy).
//
// Variants (all use group-base homes and exact linear probing over aligned 4-slot groups):
//   0  baseline   two 16-byte LDS reads per step, 4 + 4 64-bit compares (what k_partition_agg does today)
//   1  tags       a 16-bit tag per slot (16 KB more LDS): ONE 8-byte LDS read decides a 4-slot group, the key is read
//                 (8 bytes) only for a tag match -> ~3x less LDS traffic, ~half the compares
//   2  stragglers variant 0, but a lane that is not done after its first step parks its row in a per-wave LDS queue; the
//                 wave goes on with the next rows and works the queue off 64 at a time (a wave no longer pays for its
//                 slowest lane on every row)
//   3  tags + stragglers
// Every variant must produce the same per-workgroup checksum (sum of all accumulators == sum of all values).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

constexpr int kBlockThreads = 1024;
constexpr uint32_t kSlots = 8192;                 // per table block
constexpr uint32_t kKeys = 3906;                  // keys per block (10^6 groups over 256 blocks)
constexpr uint64_t kEmpty = 0x8000000000000000ull;
constexpr int kQueue = 128;                       // straggler queue entries per wave

__device__ __forceinline__ uint32_t hash_word(uint64_t k, uint32_t seed) {  // == dfx::hash_word
  uint32_t x = ((uint32_t)k ^ seed) * 0xCC9E2D51u;
  x ^= x >> 15;
  x ^= (uint32_t)(k >> 32);
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  return x ^ (x >> 16);
}
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// key j of block b (never kEmpty: the top bit is cleared)
__device__ __forceinline__ uint64_t block_key(uint32_t b, uint32_t j) { return mix64(((uint64_t)b << 32) | j) & 0x7FFFFFFFFFFFFFFFull; }
__device__ __forceinline__ uint32_t home_group(uint32_t h) { return (h >> 19) & (kSlots / 4 - 1); }  // top 13 bits -> slot, / 4
__device__ __forceinline__ uint16_t tag_of(uint32_t h) { return (uint16_t)((h & 0xFFFFu) | 1u); }      // low bits, never 0

struct Lds {
  uint64_t* keys;   // [kSlots]
  double* accs;     // [kSlots]
  uint16_t* tags;   // [kSlots]   (variants 1, 3)
  uint32_t* qrow;   // [waves][kQueue]  (variants 2, 3) parked row ids (i << 10 | thread): the row is re-derived, as the real
                    //                  kernel would re-read it from the (L2-resident) partition buffer
};

// one probe step of the baseline: returns the slot (>= 0) if the group holds the key, -2 if the group has an empty slot
// (key absent: cannot happen in the steady state), -1 to go on with the next group
__device__ __forceinline__ int step_keys(const Lds& L, uint32_t g, uint64_t key) {
  const ulonglong2 ka = *(const ulonglong2*)&L.keys[g * 4];
  const ulonglong2 kb = *(const ulonglong2*)&L.keys[g * 4 + 2];
  if (ka.x == key) return (int)(g * 4 + 0);
  if (ka.y == key) return (int)(g * 4 + 1);
  if (kb.x == key) return (int)(g * 4 + 2);
  if (kb.y == key) return (int)(g * 4 + 3);
  if (ka.x == kEmpty || ka.y == kEmpty || kb.x == kEmpty || kb.y == kEmpty) return -2;
  return -1;
}
// one probe step with tags: four 16-bit tags in one 8-byte read; a key read only on a tag match
__device__ __forceinline__ int step_tags(const Lds& L, uint32_t g, uint64_t key, uint16_t tag) {
  const uint64_t t4 = *(const uint64_t*)&L.tags[g * 4];
  bool any_empty = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint16_t t = (uint16_t)(t4 >> (16 * j));
    if (t == tag && L.keys[g * 4 + j] == key) return (int)(g * 4 + j);
    any_empty = any_empty || t == 0;
  }
  return any_empty ? -2 : -1;
}

template <bool TAGS>
__device__ __forceinline__ int lookup_from(const Lds& L, uint32_t g, uint64_t key, uint16_t tag) {
  for (uint32_t it = 0; it < kSlots / 4; ++it) {
    const int r = TAGS ? step_tags(L, g, key, tag) : step_keys(L, g, key);
    if (r != -1) return r;
    g = (g + 1) & (kSlots / 4 - 1);
  }
  return -2;
}

// insert during the build phase (linear probing from the group base, claim by CAS on the key, then publish the tag)
__device__ __forceinline__ void insert_key(const Lds& L, uint64_t key, bool tags) {
  const uint32_t h = hash_word(key, 0x9E3779B9u);
  uint32_t slot = home_group(h) * 4;
  for (uint32_t it = 0; it < kSlots; ++it) {
    const uint64_t old = atomicCAS((unsigned long long*)&L.keys[slot], (unsigned long long)kEmpty, (unsigned long long)key);
    if (old == kEmpty || old == key) {
      if (tags) L.tags[slot] = tag_of(h);
      return;
    }
    slot = (slot + 1) & (kSlots - 1);
  }
}

template <int VARIANT>
__global__ __launch_bounds__(kBlockThreads) void k_pass2(int rows_per_thread, double* __restrict__ checksum,
                                                        long long* __restrict__ loop_ticks, unsigned int* __restrict__ miss) {
  constexpr bool TAGS = (VARIANT & 1) != 0;
  constexpr bool QUEUE = (VARIANT & 2) != 0;
  extern __shared__ __attribute__((aligned(16))) uint8_t raw[];
  Lds L;
  L.keys = (uint64_t*)raw;
  L.accs = (double*)(raw + (size_t)kSlots * 8);
  L.tags = (uint16_t*)(raw + (size_t)kSlots * 16);
  uint8_t* qbase = raw + (size_t)kSlots * 16 + (TAGS ? (size_t)kSlots * 2 : 0);
  L.qrow = (uint32_t*)qbase;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < kSlots; i += kBlockThreads) {
    L.keys[i] = kEmpty;
    L.accs[i] = 0.0;
    if (TAGS) L.tags[i] = 0;
  }
  __syncthreads();
  for (uint32_t j = threadIdx.x; j < kKeys; j += kBlockThreads) insert_key(L, block_key(blockIdx.x, j), TAGS);
  __syncthreads();

  unsigned int misses = 0;
  uint32_t* wq = L.qrow + (size_t)wave * kQueue;
  // a parked row, looked up to the end starting at the group after its home
  auto finish = [&](uint32_t packed) {
    const uint32_t r2 = hash_word(((uint64_t)blockIdx.x << 40) | ((uint64_t)(packed >> 10) << 12) | (packed & 1023u), 0x1234567u);
    const uint64_t k2 = block_key(blockIdx.x, r2 % kKeys);
    const uint32_t h2 = hash_word(k2, 0x9E3779B9u);
    const int s2 = lookup_from<TAGS>(L, (home_group(h2) + 1) & (kSlots / 4 - 1), k2, tag_of(h2));
    if (s2 >= 0) unsafeAtomicAdd(&L.accs[s2], (double)((r2 >> 20) & 15u));
    else ++misses;
  };
  uint32_t qn = 0;  // wave-uniform
  double local = 0.0;  // sum of the values this thread fed in (for the checksum)
  const long long t0 = wall_clock64();
  for (int i = 0; i < rows_per_thread; ++i) {
    // the row: a key of this block (uniform over its key set), value = small integer (exact sums)
    const uint32_t r = hash_word(((uint64_t)blockIdx.x << 40) | ((uint64_t)i << 12) | threadIdx.x, 0x1234567u);
    const uint64_t key = block_key(blockIdx.x, r % kKeys);
    const double val = (double)((r >> 20) & 15u);
    local += val;
    const uint32_t h = hash_word(key, 0x9E3779B9u);
    const uint32_t g = home_group(h);
    const uint16_t tag = tag_of(h);
    if (!QUEUE) {
      const int s = lookup_from<TAGS>(L, g, key, tag);
      if (s >= 0) unsafeAtomicAdd(&L.accs[s], val);
      else ++misses;
    } else {
      // first step for everybody; whoever is not done parks the row (compacted with ballot + mbcnt) and goes on
      const int s = TAGS ? step_tags(L, g, key, tag) : step_keys(L, g, key);
      if (s >= 0) unsafeAtomicAdd(&L.accs[s], val);
      const bool later = s == -1;
      if (s == -2) ++misses;
      const uint64_t m = __ballot(later);
      if (m != 0) {
        if (later) {
          const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          wq[at] = ((uint32_t)i << 10) | threadIdx.x;
        }
        qn += (uint32_t)__popcll(m);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        while (qn >= 64) {  // work 64 parked rows off at full lane utilisation
          qn -= 64;
          finish(wq[qn + lane]);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
    }
  }
  if (QUEUE && (uint32_t)lane < qn) finish(wq[lane]);  // the wave's last < 64 parked rows
  const long long t1 = wall_clock64();
  __syncthreads();
  // checksum: sum of the accumulators minus the sum of the values fed in must be 0 (exact: small integers)
  double part = -local;
  for (uint32_t i = threadIdx.x; i < kSlots; i += kBlockThreads) part += L.accs[i];
  for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
  if (lane == 0) atomicAdd(&checksum[blockIdx.x], part);
  if (misses) atomicAdd(miss, misses);
  if (threadIdx.x == 0) loop_ticks[blockIdx.x] = t1 - t0;
}

template <int VARIANT>
static void run(int cus, int rows_per_thread, double* d_sum, long long* d_ticks, unsigned int* d_miss) {
  const size_t lds = (size_t)kSlots * 16 + ((VARIANT & 1) ? (size_t)kSlots * 2 : 0) + ((VARIANT & 2) ? (size_t)(kBlockThreads / 64) * kQueue * 4 : 0);
  CK(hipMemset(d_sum, 0, sizeof(double) * cus));
  CK(hipMemset(d_miss, 0, sizeof(unsigned int)));
  CK(hipFuncSetAttribute((const void*)k_pass2<VARIANT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_pass2<VARIANT>, dim3(cus), dim3(kBlockThreads), lds, 0, rows_per_thread, d_sum, d_ticks, d_miss);  // warm-up
  CK(hipDeviceSynchronize());
  CK(hipMemset(d_sum, 0, sizeof(double) * cus));
  CK(hipMemset(d_miss, 0, sizeof(unsigned int)));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(k_pass2<VARIANT>, dim3(cus), dim3(kBlockThreads), lds, 0, rows_per_thread, d_sum, d_ticks, d_miss);
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  double* h_sum = (double*)malloc(sizeof(double) * cus);
  long long* h_ticks = (long long*)malloc(sizeof(long long) * cus);
  unsigned int h_miss = 0;
  CK(hipMemcpy(h_sum, d_sum, sizeof(double) * cus, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_ticks, d_ticks, sizeof(long long) * cus, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&h_miss, d_miss, sizeof(unsigned int), hipMemcpyDeviceToHost));
  double worst = 0;
  long long tmax = 0;
  for (int i = 0; i < cus; ++i) {
    if (h_sum[i] > worst || -h_sum[i] > worst) worst = h_sum[i] < 0 ? -h_sum[i] : h_sum[i];
    if (h_ticks[i] > tmax) tmax = h_ticks[i];
  }
  const double rows = (double)cus * kBlockThreads * rows_per_thread;
  printf("{\"bench\":\"pass2_find_or_claim\",\"variant\":%d,\"lds_bytes\":%zu,\"rows\":%.0f,\"kernel_ms\":%.3f,\"g_rows_per_s\":%.1f,"
         "\"probe_loop_us_max\":%.1f,\"checksum_error\":%g,\"misses\":%u}\n",
         VARIANT, lds, rows, ms, rows / ms * 1e-6, tmax / 100.0, worst, h_miss);
  free(h_sum);
  free(h_ticks);
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int rows_per_thread = argc > 1 ? atoi(argv[1]) : 256;  // < 2^22 (parked row ids)  // 262144 rows per workgroup: config 3's pass 2 per batch
  double* d_sum;
  long long* d_ticks;
  unsigned int* d_miss;
  CK(hipMalloc(&d_sum, sizeof(double) * cus));
  CK(hipMalloc(&d_ticks, sizeof(long long) * cus));
  CK(hipMalloc(&d_miss, sizeof(unsigned int)));
  run<0>(cus, rows_per_thread, d_sum, d_ticks, d_miss);
  run<1>(cus, rows_per_thread, d_sum, d_ticks, d_miss);
  run<2>(cus, rows_per_thread, d_sum, d_ticks, d_miss);
  run<3>(cus, rows_per_thread, d_sum, d_ticks, d_miss);
  return 0;
}
