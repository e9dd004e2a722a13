// Developer lab: streaming-read rate of a 46 MB array (MALL resident across launches) with 8-byte vs
// 16-byte loads per lane, one wavefront per block like k_spmm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int W>  // W doubles per lane per load (1 or 2), SEG doubles per wave
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_read(const double* __restrict__ a, long seg, double* out) {
  const double* p = a + (long)blockIdx.x * seg;
  double acc = 0;
  if (W == 1) {
#pragma unroll 12
    for (long k = threadIdx.x; k < seg; k += 64) acc += p[k];
  } else {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll 6
    for (long k = threadIdx.x; k < seg / 2; k += 64) { const double2 t = q[k]; acc += t.x + t.y; }
  }
  if (acc == 12345.678) out[0] = acc;
}

int main(int argc, char** argv) {
  const long mb = argc > 1 ? atol(argv[1]) : 46;
  const long seg = argc > 2 ? atol(argv[2]) : 64 * 33;  // doubles per wave (~ a pose slice: 11 slots x 3 values)
  const long n = mb * 1024 * 1024 / 8 / seg * seg;
  double *a, *out;
  CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&out, 8)); CK(hipMemset(a, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = (int)(n / seg);
  for (int w = 1; w <= 2; ++w) {
    for (int it = 0; it < 10; ++it) { if (w == 1) hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(64), 0, 0, a, seg, out); else hipLaunchKernelGGL(k_read<2>, dim3(grid), dim3(64), 0, 0, a, seg, out); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 200;
    for (int it = 0; it < reps; ++it) { if (w == 1) hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(64), 0, 0, a, seg, out); else hipLaunchKernelGGL(k_read<2>, dim3(grid), dim3(64), 0, 0, a, seg, out); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%ld MB, %ld doubles per wave, %d-byte loads: %.2f us  %.2f TB/s (grid %d)\n", mb, seg, 8 * w, us, n * 8.0 / us / 1e6, grid);
  }
  return 0;
}
