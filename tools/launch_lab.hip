#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_empty(double* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }
__global__ void k_touch(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
// persistent kernel: `phases` grid barriers (sense counter), each phase touches memory like k_touch
__global__ void k_persist(double* p, int n, unsigned* bar, int phases) {
  unsigned target = 0;
  for (int ph = 0; ph < phases; ++ph) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0000001 + 1.0;
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(bar, 1u);
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __threadfence();
    }
    __syncthreads();
  }
}
int main() {
  double* p; unsigned* bar; const int n = 14084 * 4;
  CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); CK(hipMalloc(&bar, 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  for (int which = 0; which < 2; ++which) {
    for (int it = 0; it < 200; ++it) { if (which) hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, s, p, n); else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    const int reps = 2000;
    for (int it = 0; it < reps; ++it) { if (which) hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, s, p, n); else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s kernels back to back in one stream: %.2f us each\n", which ? "touch (56k doubles)" : "empty", ms * 1e3 / reps);
  }
  for (int blocks : {32, 64, 128, 256, 512}) {
    const int phases = 2000;
    CK(hipMemset(bar, 0, 4));
    hipLaunchKernelGGL(k_persist, dim3(blocks), dim3(256), 0, s, p, n, bar, 10); CK(hipStreamSynchronize(s));
    CK(hipMemset(bar, 0, 4));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_persist, dim3(blocks), dim3(256), 0, s, p, n, bar, phases);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("persistent, %3d blocks: %.2f us per phase (touch + grid barrier)\n", blocks, ms * 1e3 / phases);
  }
  return 0;
}
