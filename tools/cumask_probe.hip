// cumask_probe.hip -- does hipExtStreamCreateWithCUMask split the MI355X's 256 CUs the way the partitioned GROUP BY
// wants to (pass 1 of batch i+1 on one CU set, pass 2 of batch i on the complement), and do two kernels on
// complementary masks really run side by side?
//   hipcc --offload-arch=gfx950 -O3 tools/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
// Prints one JSON line per experiment:
//   placement: which (XCC, SE, CU) the workgroups of a masked stream ran on, for a prefix mask of K bits and its complement
//   overlap:   time of a "scan-like" kernel (A: 1024 lanes, 100 KB LDS, streams `bytes` from HBM) and a "probe-like" kernel
//              (B: 1024 lanes, 132 KB LDS, LDS atomics only) alone and together on the two masked streams
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void k_where(uint32_t* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    out[blockIdx.x] = ((xcc & 15u) << 16) | (((hwid >> 13) & 7u) << 8) | ((hwid >> 8) & 15u);  // xcc, se, cu
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(32);
}

typedef uint64_t u64x2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_scan(const u64x2_t* __restrict__ src, size_t n16, uint64_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  if (threadIdx.x == 0) lds[0] = 1;
  uint64_t acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const u64x2_t a = __builtin_nontemporal_load(src + i);
    acc += a.x ^ (a.y >> 7);
  }
  if (acc == 0x1234567ull) out[0] = acc + lds[0];
}
__global__ __launch_bounds__(1024) void k_probe(int iters, uint32_t slots, double* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t raw[];
  double* accs = (double*)raw;
  for (uint32_t i = threadIdx.x; i < slots; i += blockDim.x) accs[i] = 0.0;
  __syncthreads();
  uint32_t x = blockIdx.x * 9781u + threadIdx.x * 6271u + 1u;
  for (int it = 0; it < iters; ++it) {
    x = x * 1664525u + 1013904223u;
    uint32_t h = (x ^ (x >> 15)) * 0x85EBCA6Bu;
    h ^= h >> 13;
    unsafeAtomicAdd(&accs[h & (slots - 1)], 1.0);
  }
  __syncthreads();
  if (threadIdx.x == 0 && accs[0] == -1.0) out[0] = accs[1];
}

static void make_mask(std::vector<uint32_t>& m, int n_cu, int lo, int hi) {
  m.assign((size_t)(n_cu + 31) / 32, 0u);
  for (int i = lo; i < hi; ++i) m[(size_t)i / 32] |= 1u << (i % 32);
}

static void placement(hipStream_t s, const char* what, int n_wg) {
  uint32_t* d = nullptr;
  CK(hipMalloc(&d, sizeof(uint32_t) * (size_t)n_wg));
  hipLaunchKernelGGL(k_where, dim3(n_wg), dim3(64), 0, s, d, 200);
  CK(hipStreamSynchronize(s));
  std::vector<uint32_t> h((size_t)n_wg);
  CK(hipMemcpy(h.data(), d, sizeof(uint32_t) * (size_t)n_wg, hipMemcpyDeviceToHost));
  int per_xcc[16] = {0};
  bool seen[16][8][16];
  memset(seen, 0, sizeof(seen));
  int distinct = 0;
  for (uint32_t w : h) {
    const int x = (w >> 16) & 15, se = (w >> 8) & 7, cu = w & 15;
    if (!seen[x][se][cu]) {
      seen[x][se][cu] = true;
      ++distinct;
      ++per_xcc[x];
    }
  }
  printf("{\"exp\": \"placement\", \"stream\": \"%s\", \"workgroups\": %d, \"distinct_cus\": %d, \"per_xcc\": [", what, n_wg, distinct);
  for (int x = 0; x < 8; ++x) printf("%d%s", per_xcc[x], x < 7 ? ", " : "");
  printf("], \"xcc0_se_cu\": [");
  bool first = true;
  for (int se = 0; se < 8; ++se)
    for (int cu = 0; cu < 16; ++cu)
      if (seen[0][se][cu]) {
        printf("%s\"%d.%d\"", first ? "" : ", ", se, cu);
        first = false;
      }
  printf("]}\n");
  CK(hipFree(d));
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int n_cu = prop.multiProcessorCount;
  const int ka = argc > 1 ? atoi(argv[1]) : 160;  // CUs of stream A (prefix of the mask)
  std::vector<uint32_t> ma, mb;
  make_mask(ma, n_cu, 0, ka);
  make_mask(mb, n_cu, ka, n_cu);
  hipStream_t sa = nullptr, sb = nullptr, plain = nullptr;
  const hipError_t ea = hipExtStreamCreateWithCUMask(&sa, (uint32_t)ma.size(), ma.data());
  const hipError_t eb = hipExtStreamCreateWithCUMask(&sb, (uint32_t)mb.size(), mb.data());
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  printf("{\"exp\": \"create\", \"cus\": %d, \"mask_a_bits\": %d, \"rc_a\": \"%s\", \"rc_b\": \"%s\"}\n", n_cu, ka, hipGetErrorString(ea),
         hipGetErrorString(eb));
  if (ea != hipSuccess || eb != hipSuccess) return 1;
  placement(plain, "unmasked", 4096);
  placement(sa, "A(prefix)", 4096);
  placement(sb, "B(suffix)", 4096);

  // overlap
  const size_t bytes = (size_t)1 << 30;
  u64x2_t* src = nullptr;
  uint64_t* out = nullptr;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 1, bytes));
  hipEvent_t e0, e1, e2, e3;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  CK(hipFuncSetAttribute((const void*)k_scan, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int iters_b = argc > 2 ? atoi(argv[2]) : 600;
  auto run_a = [&](hipStream_t s, int grid) { hipLaunchKernelGGL(k_scan, dim3(grid), dim3(1024), 100 * 1024, s, src, bytes / 16, out); };
  auto run_b = [&](hipStream_t s) { hipLaunchKernelGGL(k_probe, dim3(256), dim3(1024), 132 * 1024, s, iters_b, 16384u, (double*)out); };
  auto time_one = [&](hipStream_t s, auto fn) {
    fn(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); for (int i = 0; i < 5; ++i) fn(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5;
  };
  const float a_full = time_one(plain, [&] { run_a(plain, n_cu); });
  const float a_masked = time_one(sa, [&] { run_a(sa, ka); });
  const float a_plain_ka = time_one(plain, [&] { run_a(plain, ka); });
  const float b_full = time_one(plain, [&] { run_b(plain); });
  const float b_masked = time_one(sb, [&] { run_b(sb); });
  // together: 5 A on stream A, 5 B on stream B, wall time by host clock around both
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, sa)); CK(hipEventRecord(e2, sb));
  for (int i = 0; i < 5; ++i) { run_a(sa, ka); run_b(sb); }
  CK(hipEventRecord(e1, sa)); CK(hipEventRecord(e3, sb));
  CK(hipDeviceSynchronize());
  float ta = 0, tb = 0;
  CK(hipEventElapsedTime(&ta, e0, e1)); CK(hipEventElapsedTime(&tb, e2, e3));
  // same pair on two UNMASKED streams
  hipStream_t p2 = nullptr;
  CK(hipStreamCreateWithFlags(&p2, hipStreamNonBlocking));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, plain)); CK(hipEventRecord(e2, p2));
  for (int i = 0; i < 5; ++i) { run_a(plain, ka); run_b(p2); }
  CK(hipEventRecord(e1, plain)); CK(hipEventRecord(e3, p2));
  CK(hipDeviceSynchronize());
  float ua = 0, ub = 0;
  CK(hipEventElapsedTime(&ua, e0, e1)); CK(hipEventElapsedTime(&ub, e2, e3));
  printf("{\"exp\": \"overlap\", \"a_cus\": %d, \"gb\": %.2f, \"a_alone_all_cus_ms\": %.4f, \"a_alone_masked_ms\": %.4f, \"a_alone_unmasked_%d_wgs_ms\": %.4f, "
         "\"b_alone_all_cus_ms\": %.4f, \"b_alone_masked_ms\": %.4f, \"together_masked_a_ms\": %.4f, \"together_masked_b_ms\": %.4f, "
         "\"together_unmasked_a_ms\": %.4f, \"together_unmasked_b_ms\": %.4f, \"a_GBps_all\": %.0f, \"a_GBps_masked\": %.0f, \"a_GBps_together\": %.0f}\n",
         ka, bytes / 1e9, a_full, a_masked, ka, a_plain_ka, b_full, b_masked, ta / 5, tb / 5, ua / 5, ub / 5, bytes / a_full * 1e-6,
         bytes / a_masked * 1e-6, bytes / (ta / 5) * 1e-6);
  return 0;
}
