// Developer lab (not product code): times the product SpMM / Hvp kernel on a CSR
// dumped by tools/dump_csr.py for different format parameters.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Icora_amd/csrc -Iinclude tools/spmm_lab.hip \
//         cora_amd/csrc/format_build.cpp cora_amd/csrc/kernels.hip -o tools/spmm_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "cora_internal.h"
#include "kernels.h"

using namespace cora;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

// ---- experiment: lane = (unit, 16-byte piece); 4 waves per 64-unit slice (21/21/21/1)
template <int LD, int D>
__global__ __launch_bounds__(256) void k_piece(const SpmmArgs A) {
  constexpr int PP = LD / 2;
  constexpr int UPW = 64 / PP;  // units per wave
  const int lane = threadIdx.x & 63;
  const int gw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
  constexpr int WPS = (64 + UPW - 1) / UPW;  // waves per slice
  const int s = gw / WPS, sub = gw - s * WPS;
  if (s >= A.n_slices) return;
  const SliceDesc sd = A.slices[s];
  const int ul = lane / PP, pc = lane - ul * PP;
  const int u = sub * UPW + ul;  // unit within slice
  const bool ok = (ul < UPW) && (u < sd.nrows);
  const int uu = ok ? u : 0;
  const double2 *X2 = reinterpret_cast<const double2 *>(A.X);
  if (sd.type == kSliceStiefel) {
    double2 acc[D];
#pragma unroll
    for (int a = 0; a < D; ++a) acc[a] = make_double2(0.0, 0.0);
#pragma unroll 4
    for (int k = 0; k < sd.width; ++k) {
      const int c = A.scol[sd.coff + (size_t)k * 64 + uu];
      double v[D];
#pragma unroll
      for (int a = 0; a < D; ++a) v[a] = A.sval[sd.off + ((size_t)k * D + a) * 64 + uu];
      const double2 x = X2[(size_t)c * PP + pc];
#pragma unroll
      for (int a = 0; a < D; ++a) { acc[a].x = fma(v[a], x.x, acc[a].x); acc[a].y = fma(v[a], x.y, acc[a].y); }
    }
    if (ok) {
      double2 *O2 = reinterpret_cast<double2 *>(A.out);
#pragma unroll
      for (int a = 0; a < D; ++a) O2[((size_t)sd.row0 + (size_t)u * D + a) * PP + pc] = acc[a];
    }
  } else {
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll 4
    for (int k = 0; k < sd.width; ++k) {
      const int c = A.scol[sd.coff + (size_t)k * 64 + uu];
      const double v = A.sval[sd.off + (size_t)k * 64 + uu];
      const double2 x = X2[(size_t)c * PP + pc];
      acc.x = fma(v, x.x, acc.x);
      acc.y = fma(v, x.y, acc.y);
    }
    if (ok) {
      const size_t row = (sd.type == kSliceEuclidPerm) ? (size_t)A.perm[sd.row0 + u] : (size_t)sd.row0 + u;
      reinterpret_cast<double2 *>(A.out)[row * PP + pc] = acc;
    }
  }
}

// calibration: pure streaming read of n doubles (8 B / lane, coalesced) and n/2 ints
__global__ __launch_bounds__(256) void k_stream(const double *a, size_t n, const int *b, size_t m, double *sink) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) s += b[i];
  if (s == 123.456) *sink = s;
}

template <typename T>
T *dev(const std::vector<T> &v) {
  T *p;
  CK(hipMalloc((void **)&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char **argv) {
  if (argc < 2) { printf("usage: spmm_lab q.bin [sigma] [p] [chunkmode]\n"); return 1; }
  if (argc > 2) g_sigma = atoi(argv[2]);
  if (getenv("LAB_ODD")) g_pad_even = !atoi(getenv("LAB_ODD"));
  if (getenv("LAB_INTER")) g_interleave = atoi(getenv("LAB_INTER"));
  if (getenv("LAB_CHUNK")) g_long_chunk = atoi(getenv("LAB_CHUNK"));
  const int p = argc > 3 ? atoi(argv[3]) : 5;
  const bool nolong = argc > 4 && atoi(argv[4]) >= 0;
  if (getenv("LAB_ODD")) g_pad_even = !atoi(getenv("LAB_ODD"));
  const int LD = ld_for(p);
  FILE *f = fopen(argv[1], "rb");
  int64_t hdr[5];
  if (fread(hdr, 8, 5, f) != 5) return 1;
  const int d = hdr[0], n = hdr[1], r = hdr[2], nt = hdr[3];
  const int64_t nnz = hdr[4], N = (int64_t)d * n + r + nt;
  std::vector<int32_t> rowptr(N + 1), col(nnz);
  std::vector<double> val(nnz);
  if (fread(rowptr.data(), 4, N + 1, f) != (size_t)N + 1) return 1;
  if (fread(col.data(), 4, nnz, f) != (size_t)nnz) return 1;
  if (fread(val.data(), 8, nnz, f) != (size_t)nnz) return 1;
  fclose(f);
  HostFormat F;
  build_format(d, n, r, nt, rowptr.data(), col.data(), val.data(), 0, 1, F);
  if (nolong) {
    const int mode = atoi(argv[4]);  // 0: drop chunks; 1: empty single-chunk rows; 2: empty multi-chunk rows
    if (mode == 0) { F.chunks.clear(); F.chunk_order.clear(); }
    for (auto &ch : F.chunks) {
      ch.k1 = ch.k0;
      if (mode == 1) { ch.nchunks = 1; }
    }
  }
  printf("N=%lld nnz=%lld slices=%zu padded=%lld long=%lld chunks=%zu sigma=%d p=%d bytes: val %zu col %zu\n",
         (long long)N, (long long)nnz, F.slices.size(), (long long)F.padded_nnz, (long long)F.long_nnz,
         F.chunks.size(), g_sigma, p, F.sval.size() * 8, F.scol.size() * 4);

  std::vector<double> X((size_t)F.L.rows * LD, 0.0), Y(X.size(), 0.0), ref(X.size(), 0.0), got(X.size());
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1, 1);
  for (int64_t i = 0; i < F.L.rows; ++i)
    for (int j = 0; j < p; ++j) { X[i * LD + j] = U(rng); Y[i * LD + j] = U(rng); }
  format_spmm_host(F, X.data(), LD, ref.data());  // (chunks cleared above when nolong)
  std::vector<double> lam_st((size_t)n * d * d), lam_ob(std::max(r, 1));
  for (auto &v : lam_st) v = U(rng);
  for (auto &v : lam_ob) v = U(rng);

  SpmmArgs A;
  A.slices = dev(F.slices);
  A.n_slices = (int)F.slices.size();
  A.n_chunks = (int)F.chunks.size();
  A.sval = dev(F.sval);
  A.scol = dev(F.scol);
  A.perm = dev(F.perm);
  A.chunks = dev(F.chunks);
  A.chunk_order = dev(F.chunk_order);
  A.lval = dev(F.lval);
  A.lcol = dev(F.lcol);
  std::vector<double> part(std::max<size_t>(F.chunks.size(), 1) * kMaxLD, 0.0);
  std::vector<unsigned> tick(std::max(F.n_long_rows, 1), 0u);
  A.partials = dev(part);
  A.tickets = dev(tick);
  A.X = dev(X);
  double *dout;
  CK(hipMalloc((void **)&dout, ref.size() * 8));
  A.out = dout;
  A.Y = dev(Y);
  A.lam_st = dev(lam_st);
  A.lam_ob = dev(lam_ob);

  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double b_spmm = 12.0 * nnz + 4.0 * (N + 1) + 16.0 * N * p;
  const double b_hvp = b_spmm + 8.0 * ((double)d * n + r) * p + 8.0 * ((double)n * d * d + r);
  auto run = [&](const char *name, int epi, double bytes, bool check) {
    CK(hipMemsetAsync(dout, 0, ref.size() * 8, st));
    for (int i = 0; i < 10; ++i) CK(launch_spmm(A, LD, d, epi, st));
    CK(hipStreamSynchronize(st));
    const int reps = 100;
    float best = 1e30f, worst = 0.f, ms;
    for (int round = 0; round < 7; ++round) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) CK(launch_spmm(A, LD, d, epi, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
      worst = std::max(worst, ms);
    }
    ms = best;
    printf("[worst %.2f] ", worst * 1e3 / reps);
    double err = -1;
    if (check) {
      CK(hipMemcpy(got.data(), dout, ref.size() * 8, hipMemcpyDeviceToHost));
      err = 0;
      double mx = 0;
      for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, std::abs(got[i] - ref[i])); mx = std::max(mx, std::abs(ref[i])); }
      err /= mx;
    }
    const double us = ms * 1e3 / reps;
    printf("%-12s %8.2f us  %6.0f GB/s (%.1f%% of 8 TB/s)  relerr=%.2e\n", name, us, bytes / us / 1e3,
           bytes / us / 1e3 / 80, err);
  };
  run("spmm", EPI_NONE, b_spmm, !nolong);
  if (LD == 6 && nolong) {
    constexpr int WPS = 4;
    const int grid = (A.n_slices * WPS + 3) / 4;
    CK(hipMemsetAsync(dout, 0, ref.size() * 8, st));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_piece<6, 3>), dim3(grid), dim3(256), 0, st, A);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((k_piece<6, 3>), dim3(grid), dim3(256), 0, st, A);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(got.data(), dout, ref.size() * 8, hipMemcpyDeviceToHost));
    double err = 0, mx = 0;
    for (size_t i = 0; i < ref.size(); ++i) { err = std::max(err, std::abs(got[i] - ref[i])); mx = std::max(mx, std::abs(ref[i])); }
    printf("k_piece      %8.2f us  %6.0f GB/s (%.1f%%)  relerr=%.2e\n", ms * 10, b_spmm / (ms * 10) / 1e3,
           b_spmm / (ms * 10) / 1e3 / 80, err / mx);
  }
  {
    const size_t nv = F.sval.size(), nc = F.scol.size();
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 100; ++i)
      hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, st, A.sval, nv, A.scol, nc, dout);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream     %8.2f us  reads %.1f MB -> %.0f GB/s\n", ms * 10, (nv * 8 + nc * 4) / 1e6,
           (nv * 8 + nc * 4) / (ms * 10) / 1e3);
  }
  run("S-apply", EPI_S, b_hvp, false);
  run("hvp", EPI_HVP, b_hvp, false);
  return 0;
}
