// What a kernel boundary and a grid barrier cost on one MI355X (tools/launch_lab.hip; hipcc --offload-arch=gfx950 -O3).
//   * empty / small kernels back to back in one stream: the boundary between two dependent launches
//   * persistent kernel, barrier A: one counter that every block adds to AND polls (same-address atomics + loads)
//   * persistent kernel, barrier B: a flag per block (sc1 store to its own slot), block 0 gathers the slots with
//     coalesced loads and publishes ONE generation word that the others poll with plain sc1 loads (no atomics at all)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_empty(double* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0; }
__global__ void k_touch(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
__device__ __forceinline__ void touch(double* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0000001 + 1.0;
}
__global__ void k_persist_counter(double* p, int n, unsigned* bar, int phases, int work) {
  unsigned target = 0;
  for (int ph = 0; ph < phases; ++ph) {
    if (work) touch(p, n);
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
// flags[b]: generation block b has reached; flags[gridDim.x .. ] padded; go: generation everybody may pass
__global__ void k_persist_flags(double* p, int n, unsigned* flags, unsigned* go, int phases, int work) {
  for (int ph = 1; ph <= phases; ++ph) {
    if (work) touch(p, n);
    __syncthreads();  // (the block's stores are issued; sc1 stores below are ordered behind them by the fence)
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      if (lane == 0) {
        __threadfence();
        __hip_atomic_store(flags + blockIdx.x, static_cast<unsigned>(ph), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (blockIdx.x == 0) {
        // gather: lane l watches slots l, l + 64, ...
        bool all;
        do {
          all = true;
          for (unsigned b = lane; b < gridDim.x; b += 64)
            all = all && __hip_atomic_load(flags + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= static_cast<unsigned>(ph);
          all = __all(all);
          if (!all) __builtin_amdgcn_s_sleep(1);
        } while (!all);
        if (lane == 0) __hip_atomic_store(go, static_cast<unsigned>(ph), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (lane == 0) {
        while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < static_cast<unsigned>(ph)) __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) __threadfence();
    }
    __syncthreads();
  }
}
int main(int argc, char** argv) {
  double* p; unsigned *bar, *flags, *go; const int n = argc > 1 ? atoi(argv[1]) : 14084 * 4;
  CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); CK(hipMalloc(&bar, 4)); CK(hipMalloc(&flags, 4096 * 4)); CK(hipMalloc(&go, 4));
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
    printf("%s kernels back to back in one stream: %.2f us each\n", which ? "touch (n doubles)" : "empty", ms * 1e3 / reps);
  }
  {  // the same dependent kernels replayed from a hipGraph (20 launches captured once)
    for (int which = 0; which < 2; ++which) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int it = 0; it < 20; ++it) { if (which) hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, s, p, n); else hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int it = 0; it < 10; ++it) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      const int reps = 100;
      for (int it = 0; it < reps; ++it) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s kernels replayed from a graph of 20: %.2f us each\n", which ? "touch (n doubles)" : "empty", ms * 1e3 / (reps * 20));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  for (int work = 0; work < 2; ++work)
    for (int blocks : {16, 32, 64, 128, 256, 512, 1024}) {
      const int phases = 2000;
      float t[2];
      for (int kind = 0; kind < 2; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {  // (first: warm-up)
          CK(hipMemset(bar, 0, 4)); CK(hipMemset(flags, 0, 4096 * 4)); CK(hipMemset(go, 0, 4));
          CK(hipEventRecord(e0, s));
          if (kind == 0) hipLaunchKernelGGL(k_persist_counter, dim3(blocks), dim3(256), 0, s, p, n, bar, rep ? phases : 10, work);
          else hipLaunchKernelGGL(k_persist_flags, dim3(blocks), dim3(256), 0, s, p, n, flags, go, rep ? phases : 10, work);
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        t[kind] = ms * 1e3 / phases;
      }
      printf("persistent, %4d blocks, %s: counter barrier %.2f us per phase | flag-array barrier %.2f us per phase\n", blocks,
             work ? "touch + barrier" : "barrier alone ", t[0], t[1]);
    }
  return 0;
}
