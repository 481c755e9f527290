// does a 128 KB dynamic-LDS workgroup work on gfx950 (with / without hipFuncSetAttribute)?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(1024) void k(uint64_t* out, int words) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  for (int i = threadIdx.x; i < words; i += 1024) lds[i] = (uint64_t)i * 3 + blockIdx.x;
  __syncthreads();
  uint64_t s = 0;
  for (int i = threadIdx.x; i < words; i += 1024) s += lds[words - 1 - i];
  atomicAdd((unsigned long long*)&out[blockIdx.x], (unsigned long long)s);
}
int main() {
  uint64_t* d; hipMalloc(&d, 8 * 512); 
  for (int kb : {48, 64, 96, 128, 144, 160}) {
    for (int attr = 0; attr < 2; ++attr) {
      hipMemset(d, 0, 8 * 512);
      int words = kb * 1024 / 8;
      hipError_t ea = hipSuccess;
      if (attr) ea = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
      hipLaunchKernelGGL(k, dim3(512), dim3(1024), kb * 1024, 0, d, words);
      hipError_t el = hipGetLastError();
      hipError_t es = hipDeviceSynchronize();
      uint64_t h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      int bad = 0;
      for (int b = 0; b < 512; ++b) { uint64_t want = 0; for (int i = 0; i < words; ++i) want += (uint64_t)i * 3 + b; if (h[b] != want) ++bad; }
      printf("lds=%dKB attr=%d setattr=%s launch=%s sync=%s bad_blocks=%d\n", kb, attr, hipGetErrorString(ea), hipGetErrorString(el), hipGetErrorString(es), bad);
    }
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock=%zu maxSharedMemoryPerMultiProcessor=%zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
}
