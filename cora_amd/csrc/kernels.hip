// HIP kernels for the CORA hot path on gfx950 (CDNA4, wave64).
//
// All resident vectors are row-major  rows x LD  doubles with LD = the number of columns (2..24, no padding
// columns), so one row is LD*8 contiguous bytes and a d x LD pose block is contiguous as well.  fp64 throughout (the
// reference's `typedef double Scalar`, include/CORA/CORA_types.h:43).  No MFMA in the products: the path is
// HBM/L2-bound (0.49 flop/B at p = 5); the fp64 matrix cores serve the Gram / combine kernels of LOBPCG only.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "cora_internal.h"
#include "kernels.h"

// The file is compiled several times in parallel (cora_amd/build.py): CORA_TU selects the kernel families of a
// translation unit (1 SpMM, 2 row-unit / vector / block kernels, 4 staged triangular solves), CORA_LDG the row strides
// it instantiates of the two big families (1: LD 2-5, 2: 6-9, 4: 10-12, 8: 13-16, 16: 17-20, 32: 21-24).  Default: all.
#ifndef CORA_TU
#define CORA_TU 7
#endif
#ifndef CORA_LDG
#define CORA_LDG 63
#endif

namespace cora {

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
// 16-byte accesses that only promise 8-byte alignment: gfx950 global loads / stores need dword alignment only, and
// the L1 / texture-address path is charged per instruction and cache line, so a 40-byte row is three accesses
// (x4, x4, x2) instead of five
struct __attribute__((aligned(8))) Pair8 { double x, y; };
#ifndef CORA_WIDE_ROWS
#define CORA_WIDE_ROWS 1
#endif

template <int LD>
__device__ __forceinline__ void load_row(const double *__restrict__ p, double (&x)[LD]) {
  if constexpr (LD % 2 == 0) {  // 16-byte aligned rows: dwordx4
    const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
    for (int j = 0; j < LD / 2; ++j) {
      const double2 t = q[j];
      x[2 * j] = t.x;
      x[2 * j + 1] = t.y;
    }
  } else if constexpr (CORA_WIDE_ROWS) {  // odd row stride: rows are only 8-byte aligned
    const Pair8 *q = reinterpret_cast<const Pair8 *>(p);
#pragma unroll
    for (int j = 0; j < LD / 2; ++j) {
      const Pair8 t = q[j];
      x[2 * j] = t.x;
      x[2 * j + 1] = t.y;
    }
    x[LD - 1] = p[LD - 1];
  } else {
#pragma unroll
    for (int j = 0; j < LD; ++j) x[j] = p[j];
  }
}

template <int LD>
__device__ __forceinline__ void store_row(double *__restrict__ p, const double (&x)[LD]) {
  if constexpr (LD % 2 == 0) {
    double2 *q = reinterpret_cast<double2 *>(p);
#pragma unroll
    for (int j = 0; j < LD / 2; ++j) q[j] = make_double2(x[2 * j], x[2 * j + 1]);
  } else if constexpr (CORA_WIDE_ROWS) {
    Pair8 *q = reinterpret_cast<Pair8 *>(p);
#pragma unroll
    for (int j = 0; j < LD / 2; ++j) q[j] = Pair8{x[2 * j], x[2 * j + 1]};
    p[LD - 1] = x[LD - 1];
  } else {
#pragma unroll
    for (int j = 0; j < LD; ++j) p[j] = x[j];
  }
}

template <int LD>
__device__ __forceinline__ double dot_row(const double (&a)[LD], const double (&b)[LD]) {
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < LD; ++j) s = fma(a[j], b[j], s);
  return s;
}

// Q's value / index streams are read exactly once per product: mark them
// non-temporal so they do not evict the X rows the gathers re-use from L1/L2.
#ifndef CORA_STREAM_NT
#define CORA_STREAM_NT 0
#endif
#ifndef CORA_POSE_UNROLL
#define CORA_POSE_UNROLL 3
#endif
template <typename T>
__device__ __forceinline__ T stream_load(const T *p) {
#if CORA_STREAM_NT == 1
  return __builtin_nontemporal_load(p);
#elif CORA_STREAM_NT == 2  // sc1: served by L2, does not allocate in the CU's L1
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return *p;
#endif
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum over a 256-thread block; result valid in thread 0. `sm` holds >= 4 doubles.
__device__ __forceinline__ double block_sum_256(double v, double *sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// Sum of the per-block partials of kappa = <p, Hp> by a block of 256 threads: ONE order of additions wherever it is
// formed (k_kappa_finish, every block of a fused forward sweep, the tail block of the last stage), so the same bits.
// (10^6 poses: 40 k partials on one block -- keep 32 loads per lane in flight.)
__device__ __forceinline__ double kappa_sum_256(const double *__restrict__ partial, int n, double *sm) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = threadIdx.x;
#pragma unroll 8
  for (; b + 768 < n; b += 1024) {
    s0 += partial[b];
    s1 += partial[b + 256];
    s2 += partial[b + 512];
    s3 += partial[b + 768];
  }
  for (; b < n; b += 256) s0 += partial[b];
  return block_sum_256((s0 + s1) + (s2 + s3), sm);
}

// Scalar recurrences of the device-resident Steihaug-Toint PCG (StpcgState, kernels.h), run by ONE thread of the
// kernel that finished the inner product they need.
__device__ __forceinline__ void stpcg_after_kappa(StpcgState &S, double kappa) {  // after Hp = H p:  kappa = <p, Hp>
  if (S.status == 0 && S.iters >= S.max_iters) S.status = 3;
  if (S.status != 0) {
    S.coef_s = 0.0;
    S.coef_r = 0.0;
    return;
  }
  S.iters++;
  S.kappa = kappa;
  const double alpha = S.r_v / kappa;
  const double sigma_next = S.sigma_M2 + 2 * alpha * S.s_Mp + alpha * alpha * S.p_M2;
  if (!(kappa > 0.0) || sigma_next >= S.Delta2) {  // negative curvature / leaves the trust region
    S.coef_s = (-S.s_Mp + sqrt(S.s_Mp * S.s_Mp + S.p_M2 * (S.Delta2 - S.sigma_M2))) / S.p_M2;
    S.coef_r = 0.0;
    S.status = 2;
    S.step_M_norm = sqrt(S.Delta2);
  } else {
    S.alpha = alpha;
    S.coef_s = alpha;
    S.coef_r = alpha;
    S.sigma_M2 = sigma_next;
    S.step_M_norm = sqrt(sigma_next);
  }
}
// coef_r of the step above without touching the state: what a block of the fused forward sweep needs when kappa has no
// launch of its own (the same expressions, so the same bits as stpcg_after_kappa leaves in S.coef_r)
__device__ __forceinline__ double stpcg_coef_r_after_kappa(const StpcgState *__restrict__ S, double kappa) {
  if (S->status != 0 || S->iters >= S->max_iters) return 0.0;
  const double alpha = S->r_v / kappa;
  const double sigma_next = S->sigma_M2 + 2 * alpha * S->s_Mp + alpha * alpha * S->p_M2;
  return (!(kappa > 0.0) || sigma_next >= S->Delta2) ? 0.0 : alpha;
}
__device__ __forceinline__ void stpcg_after_rr(StpcgState &S, double rr) {  // after r += alpha Hp:  <r, r>
  if (S.status == 0) {
    S.rr = rr;
    if (sqrt(rr) <= S.target) S.status = 1;
  }
}
__device__ __forceinline__ void stpcg_after_rv(StpcgState &S, double rv) {  // after v = P r:  <r, v>
  if (S.status != 0) {
    S.coef_v = 0.0;
    S.coef_beta = 1.0;
    return;
  }
  const double beta = rv / S.r_v;
  S.r_v = rv;
  S.coef_v = -1.0;
  S.coef_beta = beta;
  S.s_Mp = beta * (S.s_Mp + S.alpha * S.p_M2);
  S.p_M2 = S.r_v + beta * beta * S.p_M2;
}


// ---------------------------------------------------------------------------
// Sliced SpMM with fused epilogues.
//   EPI_NONE : out = Q X                       (Problem::dataMatrixProduct, :742-746)
//   EPI_S    : out = Q X - Lambda X            (certificate operator, :1162-1166)
//   EPI_HVP  : out = Proj_Y(Q X - Lambda X)    (Riemannian Hvp, :822-867)
// One wavefront per slice, lane = row, LD accumulators per lane in registers.
// Blocks [0, n_chunks) handle chunks of the long (landmark) rows instead.
// ---------------------------------------------------------------------------
// One wavefront per chunk of a long (landmark) row.
// EPI_HVP_K: the wavefront that finishes the row leaves <X[row], out[row]> in the ROW's own slot of kappa_partial
// (behind the per-block slots).  Which chunk arrives last differs from launch to launch: a term that travelled with the
// finishing block's partial would move between slots and change the rounding of their fixed-order sum -- the one source
// of run-to-run differences the solver had (tools/determinism_probe.py: 10^5 poses, where a landmark row has 40 chunks).
template <int LD, bool KAPPA>
__device__ __forceinline__ void long_chunk_wave(const SpmmArgs &A, int ci) {
  const LongChunk ch = A.chunks[ci];
  const int lane = threadIdx.x;
  double acc[LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) acc[j] = 0.0;
#pragma unroll 4
  for (int k = ch.k0 + lane; k < ch.k1; k += kWave) {
    const double v = stream_load(A.lval + k);
    double x[LD];
    load_row<LD>(A.X + static_cast<size_t>(stream_load(A.lcol + k)) * LD, x);
#pragma unroll
    for (int j = 0; j < LD; ++j) acc[j] = fma(v, x[j], acc[j]);
  }
  double tot = 0.0;  // lane j < LD ends up with column j
#pragma unroll
  for (int j = 0; j < LD; ++j) {
    const double v = wave_sum(acc[j]);
    const double v0 = __shfl(v, 0, 64);
    if (lane == j) tot = v0;
  }
  // partitioned handles: the row is distributed -- this rank's part of it goes to slot ch.slot of long_out, to be summed
  // over the ranks (the row's share of kappa follows the sum: k_long_finish)
  double *orow = A.long_out ? A.long_out + static_cast<size_t>(ch.slot) * LD : A.out + static_cast<size_t>(ch.row) * LD;
  auto publish_kappa = [&](double row_j) {  // row_j: out[row][lane] in the lanes below LD
    if constexpr (KAPPA) {
      if (A.long_out) return;  // wave-uniform
      const double t = wave_sum(lane < LD ? row_j * A.X[static_cast<size_t>(ch.row) * LD + lane] : 0.0);
      if (lane == 0) A.kappa_partial[A.kappa_long_base + ch.slot] = t;
    }
  };
  if (ch.nchunks == 1) {
    if (lane < LD) orow[lane] = tot;
    publish_kappa(tot);
    return;
  }
  // several chunks: publish the partial WRITE-THROUGH (sc1 stores, so no L2
  // release fence is needed), drain, take a ticket; the last arriver re-reads
  // all partials with sc1 loads (which bypass its L1) and sums them in chunk
  // order, so the result is deterministic.  gfx950 inter-workgroup hand-off
  // recipe R1 (cdna_hip_programming.md, Guideline 16).
  if (lane < LD)
    __hip_atomic_store(A.partials + static_cast<size_t>(ci) * kMaxLD + lane, tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int last = 0;
  if (lane == 0) {
    const unsigned old = __hip_atomic_fetch_add(A.tickets + ch.slot, 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
    last = (old == static_cast<unsigned>(ch.nchunks - 1));
    if (last) __hip_atomic_store(A.tickets + ch.slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  last = __shfl(last, 0, 64);
  if (!last) return;  // wave-uniform
  double s = 0.0;
  if (lane < LD) {
    const double *P = A.partials + static_cast<size_t>(ch.first) * kMaxLD + lane;
    int c = 0;
    for (; c + 8 <= ch.nchunks; c += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        t[u] = __hip_atomic_load(P + static_cast<size_t>(c + u) * kMaxLD, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; c < ch.nchunks; ++c)
      s += __hip_atomic_load(P + static_cast<size_t>(c) * kMaxLD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    orow[lane] = s;
  }
  publish_kappa(s);
}

// V_i - sym(Y_i V_i^T) Y_i for one pose held entirely by this thread
// (StiefelProduct::projectToTangentSpace, include/CORA/StiefelProduct.h:79-81).
template <int LD, int D>
__device__ __forceinline__ void stiefel_project_thread(const double (&y)[D][LD],
                                                       double (&v)[D][LD]) {
  double m[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) m[a][b] = dot_row<LD>(y[a], v[b]);
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const double s = 0.5 * (m[a][b] + m[b][a]);
#pragma unroll
      for (int c = 0; c < LD; ++c) v[a][c] = fma(-s, y[b][c], v[a][c]);
    }
}

// Pose slice: lane = pose, d x LD accumulators, one X-row read per d nonzeros.
//
// X window (row strides up to kWinMaxLD): along the pose chain the columns of a pose slice are the rotation rows of
// its own poses and their chain neighbours and the translation rows of the same poses -- two CONTIGUOUS pieces of X
// ((64 + 2) d and 64 + 2 rows).  The wavefront copies them to LDS with coalesced loads (4 cache lines per instruction)
// and the lanes read their rows from there; a per-lane gather of a 40-byte row costs one L1 tag look-up per lane and
// instruction (3 x 64 per slot), and the L1's look-up rate -- not bytes -- was what bounded the kernel (PMC:
// TCP_TOTAL_CACHE_ACCESSES 10.8 M per product at 10^5 poses, the L1s busy for the whole kernel).  Columns outside
// the windows (loop closures, rows of another shard) take the global gather as before; results are bit-identical.
#ifndef CORA_SPMM_WINDOW
#define CORA_SPMM_WINDOW 1
#endif
#ifndef CORA_WIN_MAX_LD
#define CORA_WIN_MAX_LD 24
#endif
// rotation window at every row stride: 198 x LD doubles (19 KB at 12, 25 KB at 16, 38 KB at 24: four to six wavefronts per
// CU still keep their loads in flight).  Round 2 stopped at 12 and the strides above fell off a cliff: (Q - Lambda) X
// with 16 columns 68.5 -> 51.6 us, 20 columns 70.5 -> 55.0 us, Hvp at p = 16 76.9 -> 60.6 us (profiles/r03_rank_sweep.md).
// (Measured and not kept: an odd LDS row stride against bank conflicts at even strides -- the index arithmetic cost more
// than the conflicts: Hvp at p = 10 32.8 -> 38.3 us, p = 16 unchanged.)
constexpr int kWinMaxLD = CORA_SPMM_WINDOW ? CORA_WIN_MAX_LD : 0;
#ifndef CORA_WIN_TRN_MAX_LD
#define CORA_WIN_TRN_MAX_LD 8
#endif
constexpr int kWinTrnMaxLD = CORA_WIN_TRN_MAX_LD;      // + the translation window while 8 wavefronts per CU fit the LDS
#ifndef CORA_POSE_COOP_EPI
#define CORA_POSE_COOP_EPI 1
#endif
#ifndef CORA_ROW_UNROLL
#define CORA_ROW_UNROLL 4  // gathers in flight per lane of a row slice (translation / range rows)
#endif
#ifndef CORA_COOP_PREFETCH_LATE
#define CORA_COOP_PREFETCH_LATE 1
#endif
#ifndef CORA_POSE_FUSE_T_MAX
#define CORA_POSE_FUSE_T_MAX 20  // (d + 1) x row stride up to which the translation row is accumulated with the rotation rows
#endif
#ifndef CORA_POSE_EARLY_MAX
#define CORA_POSE_EARLY_MAX 20  // (d + 1) x row stride up to which a chain slice requests all its fixed slots up front
#endif
#ifndef CORA_POSE_COOP_MAX_LD
// cooperative Hvp epilogue up to this row stride: above it the prefetched Y rows and Lambda blocks (d LD + d d doubles per
// lane) push the kernel into AGPR spills at one wave per SIMD -- without them p = 11 / 12 / 16 / 24: 39.6 / 40.6 / 60.1 /
// 98.7 -> 38.7 / 38.9 / 57.0 / 89.9 us (two waves per SIMD), while p = 10 lost in round 3 (32.5 -> 33.7).  Round 6: with the
// cooperative form k_spmm<10, 3, *> spilled (92-116 B of scratch per lane at two waves per SIMD: the round-5 review's
// "config 5's kernel spills"); without it 242-248 registers, no scratch, and faster on today's kernel -- certificate operator
// at 10 columns 26.25 -> 25.73 us, PMC traffic 1.36x -> 1.19x the format's compulsory bytes (writes 1.25x -> 1.00x), Hvp at
// p = 10 29.4 -> 28.75 us (profiles/r06_kernel_evolution.md): the limit is 9.
#define CORA_POSE_COOP_MAX_LD 9
#endif
#ifndef CORA_POSE_COOP_MAX_DLD
#define CORA_POSE_COOP_MAX_DLD 18  // d x row stride up to which the Hvp epilogue's operands travel through LDS (above: per lane)
#endif
#ifndef CORA_POSE_UNROLL_WIN
#define CORA_POSE_UNROLL_WIN 2  // slots per trip of the window loop (round 3, with 3 waves per SIMD: 1 / 2 / 3 / 6 / 11 slots
#endif                          // -> Hvp 21.4 / 21.5 / 21.8 / 23.0 / 30.0 us, rotated 29.2 / 29.2 / 29.3 / 32.0 / 41.0 us)
#ifdef CORA_SPMM_TIMES
// measurement build: wall-clock stamps of a pose slice's phases (tools/spmm_timeline.py)
constexpr unsigned kSpmmTimesMax = 65536;
constexpr int kSpmmPhases = 6;
__device__ unsigned long long g_spmm_phase[kSpmmPhases * kSpmmTimesMax];
#define CORA_PHASE(i) do { if (threadIdx.x == 0 && blockIdx.x < kSpmmTimesMax) g_spmm_phase[kSpmmPhases * blockIdx.x + (i)] = wall_clock64(); } while (0)
#else
#define CORA_PHASE(i) do { } while (0)
#endif
// A chain slice's second launch on a partitioned handle (kSliceRemoteTailOnly): the pairs of its tails that read rows of
// other ranks, added to the translation rows the first launch -- ahead of the exchange -- has written.
template <int LD, int D, int EPI>
__device__ __forceinline__ double remote_tail_only(const SpmmArgs &A, const SliceDesc &sd, int lane) {
  const int T = static_cast<int>(static_cast<unsigned>(sd.type) >> kSliceTailShift);
  const int mc = (sd.type >> kSliceTailMaxShift) & kSliceTailMaxMask;
  const double *__restrict__ tail_v = A.sval + sd.off + (static_cast<size_t>(kChainFixed(D)) + static_cast<size_t>(sd.width) * D) * kWave;
  const int32_t *__restrict__ tail_c = A.scol + sd.coff + (1 + static_cast<size_t>(sd.width)) * kWave;
  const int32_t tinfo = A.scol[sd.coff + lane];
  const int tstart = tinfo & 0xffff, tnloc = (tinfo >> 24) & 0x7f, tcnt = (tinfo >> 16) & 0x7f;
  double acct[LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) acct[j] = 0.0;
  for (int r0 = 0; r0 < T; r0 += kWave) {
    const int e = r0 + lane;
    int2 c2 = make_int2(sd.row0, sd.row0);
    double2 v2 = make_double2(0.0, 0.0);
    if (e < T) {
      c2 = *reinterpret_cast<const int2 *>(tail_c + 2 * e);
      v2 = *reinterpret_cast<const double2 *>(tail_v + 2 * e);
    }
    double x[LD], x1[LD], pr[LD];
    load_row<LD>(A.X + static_cast<size_t>(c2.x) * LD, x);
    load_row<LD>(A.X + static_cast<size_t>(c2.y) * LD, x1);
#pragma unroll
    for (int j = 0; j < LD; ++j) pr[j] = fma(v2.x, x[j], v2.y * x1[j]);
    for (int i = 0; i < mc; ++i) {  // wave-uniform
      const int e2 = tstart + i - r0;
      const bool mine = i >= tnloc && i < tcnt && e2 >= 0 && e2 < kWave;
#pragma unroll
      for (int j = 0; j < LD; ++j) {
        const double t = __shfl(pr[j], e2 & (kWave - 1), kWave);
        acct[j] += mine ? t : 0.0;
      }
    }
  }
  const int nrows = sd.nrows & kSliceRowsMask;
  if (lane >= nrows || tcnt == tnloc) return 0.0;
  double *o = A.out + static_cast<size_t>(A.win_trn_lo + sd.aux0 + lane) * LD;
  double cur[LD];
  load_row<LD>(o, cur);
#pragma unroll
  for (int j = 0; j < LD; ++j) cur[j] += acct[j];
  store_row<LD>(o, cur);
  if (EPI == EPI_HVP_K) {  // <X, out> is linear in out: this launch's share is <X[t], what it added>
    double x[LD];
    load_row<LD>(A.X + static_cast<size_t>(A.win_trn_lo + sd.aux0 + lane) * LD, x);
    return dot_row<LD>(x, acct);
  }
  return 0.0;
}

template <int LD, int D, int EPI, bool WIN>
__device__ __forceinline__ double pose_slice(const SpmmArgs &A, const SliceDesc &sd_in, int lane) {
  const int sflags = sd_in.nrows & ~kSliceRowsMask;  // launch-time flags of a partitioned handle's overlapped product
  SliceDesc sd = sd_in;
  sd.nrows &= kSliceRowsMask;
  const double *__restrict__ vp = A.sval + sd.off + lane;
  const int32_t *__restrict__ cp = A.scol + sd.coff + lane;
  const double *__restrict__ X = A.X;
  constexpr bool kWinLD = LD <= kWinMaxLD;
  // launch-time switch (SpmmArgs::win_on, launch_spmm): below kWinMinSlices wavefronts every wavefront is resident at
  // once and its chain of dependent latencies sets the time -- the window copy and the LDS hand-overs of the
  // cooperative epilogue are extra stages there (10^4 poses: Hvp 5.9 -> 7.1 us with them)
  // (WIN is a template parameter, the two forms are separate code: behind a run-time test every row of X was "LDS read
  // or gather", the consumers sank below the last of them and twelve rows stayed live at once)
  constexpr bool kWin = kWinLD && WIN;
  constexpr int kRotRows = (kWave + 2) * D, kTrnRows = LD <= kWinTrnMaxLD ? kWave + 2 : 0;
  // cooperative Hvp epilogue (CORA_POSE_COOP_EPI): the slice's rows of Y, its Lambda blocks and its rows of the result
  // are contiguous too -- requested with coalesced loads BEFORE the slot loop, handed to the lanes through the window's
  // LDS after it, and the result rows leave through LDS as 512-byte runs instead of 16-byte pieces of 64 lines
  constexpr bool kCoopT = kWinLD && CORA_POSE_COOP_EPI && EPI >= EPI_HVP && D * LD <= CORA_POSE_COOP_MAX_DLD;
  constexpr bool kCoop = kCoopT && WIN;
  constexpr int kYEl = kWave * D * LD, kLEl = kWave * D * D;
  constexpr int kWinEl = (kRotRows + kTrnRows) * LD;
  // (the cooperative epilogue hands Y rows + Lambda blocks in, result rows + the slice's translation rows out)
  constexpr int kCoopEl = kYEl + (kLEl > kWave * LD ? kLEl : kWave * LD);
  // staged stores (window form, row strides up to CORA_POSE_COOP_MAX_LD, every epilogue): result rows + translation rows
  constexpr bool kStaged = kWin && LD <= CORA_POSE_COOP_MAX_LD;
  constexpr int kStagedEl = kStaged ? kYEl + kWave * LD : 0;
  constexpr int kNeedEl = (kCoopT && kCoopEl > kStagedEl) ? kCoopEl : kStagedEl;
  constexpr int kSmemEl = !kWinLD ? 1 : (kNeedEl > kWinEl ? kNeedEl : kWinEl);
  __shared__ __attribute__((aligned(16))) double win[kSmemEl];
  constexpr int kYIt = (D * LD + 1) / 2, kLIt = (D * D + 1) / 2;  // (pairs of doubles per lane and access)
  double ystage[kCoopT ? 2 * kYIt : 1], lstage[kCoopT ? 2 * kLIt : 1];
  // Chain layout (kSliceChainFlag, cora_internal.h): the lane owns the pose's translation row as well, the chain's columns
  // are implied, and what Q's symmetry gives comes from the lane before (lane 0: the slice's head block).
  const bool chain = (sd.type & kSliceChainFlag) != 0;  // wave-uniform
  // narrow rows: the values a lane hands to the lane after it stay in registers and move with a lane shift; wide rows
  // (registers are what they are short of) re-read the previous lane's values from the stream the wavefront has just
  // loaded (L1 / L2 hits)
  constexpr bool kNxtRegs = LD <= 5 || (LD <= 8 && EPI < 2);
  constexpr int kFV = kChainFixed(D);
  // Everything that does not depend on the windows is requested HERE, ahead of the windows' own loads: all wavefronts of
  // a launch are resident at once, so a launch lasts as long as a wavefront's chain of dependent memory latencies -- the
  // fixed slots' values, the tail's first 64 entries and (below) the epilogue's operands arrive with the window rows.
  const int tailT = chain ? static_cast<int>(static_cast<unsigned>(sd.type) >> kSliceTailShift) : 0;  // wave-uniform
  const double *__restrict__ tail_v = A.sval + sd.off + (static_cast<size_t>(kFV) + static_cast<size_t>(sd.width) * D) * kWave;  // [pair][2]
  const int32_t *__restrict__ tail_c = A.scol + sd.coff + (1 + static_cast<size_t>(sd.width)) * kWave;
  // the lane's range of the tail (pairs): start | count << 16; pair `lane` of the tail: two columns, two values
  int32_t tinfo = 0, tc0 = sd.row0, tc1 = sd.row0;
  double tv0 = 0.0, tv1 = 0.0;
  // (kEarly: narrow rows only -- from d + 1 accumulator rows of 7 doubles on the values and the tail's rows would not
  // fit the 256 registers of two wavefronts per SIMD beside the accumulators; wide rows load every group of fixed
  // slots where it is used and gather the tail's rows in the tail)
  constexpr bool kEarly = (D + 1) * LD <= CORA_POSE_EARLY_MAX;
  // kFuseT: the translation row accumulates beside the rotation rows (one read of every row of X for d + 1 rows of Q).
  // Wide rows cannot hold d + 1 accumulator rows: there the translation row goes first, on its own -- its slots' values,
  // nine rows of X from the windows, the tail -- and is stored before the rotation rows start.
  constexpr bool kFuseT = (D + 1) * LD <= CORA_POSE_FUSE_T_MAX;
  double fx[kEarly ? kFV : 1];
  auto fixed = [&](int i) { if constexpr (kEarly) return fx[i]; else return stream_load(vp + static_cast<size_t>(i) * kWave); };
  double headv = 0.0;  // lane h < kChainHead(D): entry h of the slice's head block
  if (chain) {
    if (lane < kChainHead(D)) headv = A.head_val[static_cast<size_t>(sd.aux0 / kWave) * kChainHead(D) + lane];
    tinfo = stream_load(cp);
    if (lane < tailT) {
      const int2 c2 = *reinterpret_cast<const int2 *>(tail_c + 2 * lane);
      const double2 v2 = *reinterpret_cast<const double2 *>(tail_v + 2 * lane);
      tc0 = c2.x; tc1 = c2.y;
      tv0 = v2.x; tv1 = v2.y;
    }
    if constexpr (kEarly) {
#pragma unroll
      for (int i = 0; i < kFV; ++i) fx[i] = stream_load(vp + static_cast<size_t>(i) * kWave);
    }
  }
  // the cooperative epilogue's operands: the slice's rows of Y and its Lambda blocks, requested with coalesced loads --
  // plain slices with the window's rows; chain slices right AFTER the windows have landed: a third of the wavefront's
  // bytes, needed last, travels while the fixed slots and the tail are computed instead of holding up their operands
  // (all wavefronts of a launch start together and share the memory system: what is requested first lands first)
  auto coop_prefetch = [&] {
    if constexpr (kCoopT) {
      const double *__restrict__ Yp = A.Y + static_cast<size_t>(sd.row0) * LD;
      const double *__restrict__ Lq = A.lam_st + static_cast<size_t>(sd.aux0) * (D * D);
#pragma unroll
      // ONE predicate per access (an odd count reads one double past the slice's rows: Y has the range rows behind
      // its rotation rows, the Lambda array is allocated with the slack).  The two-way form -- a pair, else a single
      // double into the same registers -- made the compiler wait for EVERY outstanding load before each pair (a
      // write-after-write hazard on the staging registers): thirteen loads, one after the other.
      for (int i = 0; i < kYIt; ++i) {
        const int e = 2 * (i * kWave + lane), n = sd.nrows * D * LD;
        Pair8 v{0.0, 0.0};
        if (e < n) v = *reinterpret_cast<const Pair8 *>(Yp + e);
        ystage[2 * i] = v.x;
        ystage[2 * i + 1] = v.y;
      }
#pragma unroll
      for (int i = 0; i < kLIt; ++i) {
        const int e = 2 * (i * kWave + lane), n = sd.nrows * D * D;
        Pair8 v{0.0, 0.0};
        if (e < n) v = *reinterpret_cast<const Pair8 *>(Lq + e);
        lstage[2 * i] = v.x;
        lstage[2 * i + 1] = v.y;
      }
    }
  };
  if (kCoop && !(chain && CORA_COOP_PREFETCH_LATE)) coop_prefetch();
  int w0 = 0, nrot = 0, t0 = 0, ntr = 0;
  if constexpr (kWinLD) if (kWin) {
    w0 = max(sd.row0 - D, A.win_rot_lo);
    nrot = max(min(sd.row0 + (kWave + 1) * D, A.win_rot_hi) - w0, 0);
    t0 = max(A.win_trn_lo + sd.aux0 - 1, A.win_trn_lo);
    ntr = kTrnRows ? max(min(A.win_trn_lo + sd.aux0 + kWave + 1, A.win_trn_hi) - t0, 0) : 0;
    const double *__restrict__ srot = X + static_cast<size_t>(w0) * LD;
    const double *__restrict__ strn = X + static_cast<size_t>(t0) * LD;
    // LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane straight into the window -- wave-uniform LDS base + lane x 16,
    // per-lane source address --, no staging registers (44 at a row stride of 5: what kept the Hvp from holding its
    // other operands in flight) and no LDS-store pass.  Lanes past the end of the window's rows are masked off: LDS
    // there keeps whatever it held, and nothing reads it (chain columns are clamped to local rows, the general slots
    // test the window's range, lanes past the slice's poses are never stored).  An odd element count leaves one double
    // to an ordinary load.  __syncthreads() below carries the vmcnt(0).
    {
      typedef __attribute__((address_space(1))) const void *gptr_t;
      typedef __attribute__((address_space(3))) void *lptr_t;
      constexpr int kRotEl = kRotRows * LD, kTrnEl = kTrnRows * LD;
      constexpr int kRotIt = (kRotEl + 2 * kWave - 1) / (2 * kWave), kTrnIt = (kTrnEl + 2 * kWave - 1) / (2 * kWave);
      const int nre = nrot * LD, nte = ntr * LD;
#pragma unroll
      for (int i = 0; i < kRotIt; ++i) {
        const int e = 2 * (i * kWave + lane);
        if (e + 1 < nre)
          __builtin_amdgcn_global_load_lds((gptr_t)(srot + e), (lptr_t)(win + 2 * i * kWave), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < kTrnIt; ++i) {
        const int e = 2 * (i * kWave + lane);
        if (e + 1 < nte)
          __builtin_amdgcn_global_load_lds((gptr_t)(strn + e), (lptr_t)(win + kRotEl + 2 * i * kWave), 16, 0, 0);
      }
      if ((nre & 1) && lane == 0) win[nre - 1] = srot[nre - 1];
      if ((nte & 1) && lane == 1) win[kRotEl + nte - 1] = strn[nte - 1];
    }
    __syncthreads();
  }
  CORA_PHASE(0);  // the windows (and everything requested with them) have landed
#ifdef CORA_SPMM_TIMES
  if (threadIdx.x == 0 && blockIdx.x < kSpmmTimesMax)  // where the wavefront runs: HW_ID (reg 4) | XCC_ID (reg 20) << 32
    g_spmm_phase[kSpmmPhases * blockIdx.x + 5] = static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 4)) |
                                                 (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32);
#endif
  // the tail's first round: pair `lane` gathers its two rows of X now (the indices have arrived with the window's rows;
  // lanes past the tail read a valid row and multiply it by zero)
  double tg[kEarly ? LD : 1], tg1[kEarly ? LD : 1];
  if constexpr (kEarly) if (chain && tailT > 0) {
    load_row<LD>(X + static_cast<size_t>(tc0) * LD, tg);
    load_row<LD>(X + static_cast<size_t>(tc1) * LD, tg1);
  }
  if (kCoop && chain && CORA_COOP_PREFETCH_LATE) coop_prefetch();
  double acc[D][LD], acct[LD];  // the pose's d rotation rows; its translation row (chain slices)
#pragma unroll
  for (int j = 0; j < LD; ++j) {
    acct[j] = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) acc[a][j] = 0.0;
  }
  // rows of X: a local rotation / translation row from the windows when they are on, from memory otherwise
  auto x_rot = [&](int row, double (&x)[LD]) {
    if (kWinLD && kWin) {
#pragma unroll
      for (int j = 0; j < LD; ++j) x[j] = win[(row - w0) * LD + j];
    } else {
      load_row<LD>(X + static_cast<size_t>(row) * LD, x);
    }
  };
  auto x_trn = [&](int row, double (&x)[LD]) {
    if (kWinLD && kTrnRows > 0 && kWin) {
#pragma unroll
      for (int j = 0; j < LD; ++j) x[j] = win[(kRotRows + row - t0) * LD + j];
    } else {
      load_row<LD>(X + static_cast<size_t>(row) * LD, x);
    }
  };
  // one slot: column index, d values, the row of X (from the windows when they are on), d x LD products
  auto slot_apply = [&](const int32_t c, const double (&v)[D]) {
    double x[LD];
    if (kWinLD && kWin) {
      const unsigned rr = static_cast<unsigned>(c - w0), rt = static_cast<unsigned>(c - t0);
      const bool in_rot = rr < static_cast<unsigned>(nrot), in_trn = rt < static_cast<unsigned>(ntr);
      const int l = in_rot ? static_cast<int>(rr) : (in_trn ? kRotRows + static_cast<int>(rt) : 0);
#pragma unroll
      for (int j = 0; j < LD; ++j) x[j] = win[l * LD + j];
      if (!(in_rot || in_trn)) load_row<LD>(X + static_cast<size_t>(c) * LD, x);
    } else {
      load_row<LD>(X + static_cast<size_t>(c) * LD, x);
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int j = 0; j < LD; ++j) acc[a][j] = fma(v[a], x[j], acc[a][j]);
  };
  // The tail of the translation row (range measurements, loop closures): T entries of the slice, sorted by lane, gathered
  // by the whole wavefront -- lane e takes entry e -- and handed to their owners with lane permutes in entry order (a
  // fixed order of summation); the owner's loop runs to the longest tail of the slice (kSliceTailMaxShift).
  double kap_t = 0.0;
  int t_own = 0;
  auto translation_tail = [&] {
    const int T = tailT;
    const int mc = (sd.type >> kSliceTailMaxShift) & kSliceTailMaxMask;
    if (T > 0) {
      const int tstart = tinfo & 0xffff, tnloc = (tinfo >> 24) & 0x7f;
      // (pairs [0, nlocal) read rows of this shard, the rest rows of other ranks: a launch ahead of the exchange leaves
      // the rest to a later launch of the same slice that adds only them)
      const int tcnt = (sflags & kSliceSkipRemoteTail) ? tnloc : ((tinfo >> 16) & 0x7f);
      // (the remote pairs are summed on their own and added last: the same operations in the same order as the two
      // launches of a split slice, so the overlapped product of a partitioned handle has the serial one's bits)
      double accr[LD];
#pragma unroll
      for (int j = 0; j < LD; ++j) accr[j] = 0.0;
      for (int r0 = 0; r0 < T; r0 += kWave) {
        double pr[LD];
        if (kEarly && r0 == 0) {
#pragma unroll
          for (int j = 0; j < LD; ++j) pr[j] = fma(tv0, tg[kEarly ? j : 0], tv1 * tg1[kEarly ? j : 0]);
        } else if (r0 == 0) {
          double x[LD], x1[LD];
          load_row<LD>(X + static_cast<size_t>(tc0) * LD, x);
          load_row<LD>(X + static_cast<size_t>(tc1) * LD, x1);
#pragma unroll
          for (int j = 0; j < LD; ++j) pr[j] = fma(tv0, x[j], tv1 * x1[j]);
        } else {  // more than 64 pairs (128 entries) in one slice: rare
          const int e = r0 + lane;
          int2 c2 = make_int2(sd.row0, sd.row0);
          double2 v2 = make_double2(0.0, 0.0);
          if (e < T) {
            c2 = *reinterpret_cast<const int2 *>(tail_c + 2 * e);
            v2 = *reinterpret_cast<const double2 *>(tail_v + 2 * e);
          }
          double x[LD], x1[LD];
          load_row<LD>(X + static_cast<size_t>(c2.x) * LD, x);
          load_row<LD>(X + static_cast<size_t>(c2.y) * LD, x1);
#pragma unroll
          for (int j = 0; j < LD; ++j) pr[j] = fma(v2.x, x[j], v2.y * x1[j]);
        }
        for (int i = 0; i < mc; ++i) {  // wave-uniform
          const int e2 = tstart + i - r0;
          const bool mine = i < tcnt && e2 >= 0 && e2 < kWave, rem = i >= tnloc;
#pragma unroll
          for (int j = 0; j < LD; ++j) {
            const double t = __shfl(pr[j], e2 & (kWave - 1), kWave);
            acct[j] += (mine && !rem) ? t : 0.0;
            accr[j] += (mine && rem) ? t : 0.0;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < LD; ++j) acct[j] += accr[j];
    }
    if (EPI == EPI_HVP_K && lane < sd.nrows) {  // the translation row is final (no epilogue touches it): its share of <X, out>
      double x[LD];
      x_trn(t_own, x);
      kap_t = dot_row<LD>(x, acct);
    }
  };
  const double *__restrict__ vpg = vp;    // the general slots: index + d values
  const int32_t *__restrict__ cpg = cp;
  if (chain) {
    vpg = vp + static_cast<size_t>(kFV) * kWave;
    cpg = cp + kWave;
    // implied columns: rows of the lane's pose, the pose after and the pose before (clamped to the local poses where
    // there is none: the values there are zeros)
    const int np = A.n_local_poses, P = sd.aux0 + lane, Pc = min(P, np - 1);
    const int own_row = A.win_rot_lo + Pc * D, nxt_row = A.win_rot_lo + min(Pc + 1, np - 1) * D,
              prv_row = A.win_rot_lo + max(Pc - 1, 0) * D;
    const int t_nxt = A.win_trn_lo + min(Pc + 1, np - 1), t_prv = A.win_trn_lo + max(Pc - 1, 0);
    t_own = A.win_trn_lo + Pc;
    // value `slot` of the lane before (lane 0: entry h of the head block, which lane h loaded with the fixed slots and
    // hands over through a scalar register; no pose before: 0)
    // (a DPP wave shift: lane 0 keeps the "old" operand, which is the head entry.  The head block of a shard's first
    // pose is zeros -- format_build.cpp -- and every other lane has a pose before it, so there is nothing to mask.)
    auto before = [&](double mine, int slot, int h) {
      const int hlo = __builtin_amdgcn_readlane(__double2loint(headv), h), hhi = __builtin_amdgcn_readlane(__double2hiint(headv), h);
      if constexpr (kNxtRegs) {
        return __hiloint2double(__builtin_amdgcn_update_dpp(hhi, __double2hiint(mine), 0x138, 0xf, 0xf, false),
                                __builtin_amdgcn_update_dpp(hlo, __double2loint(mine), 0x138, 0xf, 0xf, false));
      } else {
        const double t = lane > 0 ? stream_load(vp + static_cast<size_t>(slot) * kWave - 1) : 0.0;
        return lane == 0 ? __hiloint2double(hhi, hlo) : t;
      }
    };
    // (wide rows: the groups of fixed slots are kept apart -- the scheduler otherwise hoists every group's loads to the top
    // of the block and the register file cannot hold them beside the accumulators)
    auto phase_fence = [] { if constexpr (!kEarly) __builtin_amdgcn_sched_barrier(0); };
    // (a) columns t_P and t_{P+1}: d + 1 values each (the rotation rows and the translation row)
    double s0[D + 1], s1[D + 1];
#pragma unroll
    for (int a = 0; a <= D; ++a) {
      s0[a] = fixed(a);
      s1[a] = fixed(D + 1 + a);
    }
    // (b) what the translation row takes from the lane before: Q(t_P, rot(P-1)_c) = its s1[c], Q(t_P, t_{P-1}) = its s1[d]
    double ps1[D + 1];
#pragma unroll
    for (int c = 0; c <= D; ++c) ps1[c] = before(s1[c], D + 1 + c, D * D + c);
    if constexpr (!kFuseT) {
      // wide rows: the whole translation row now -- Q33's three entries, the rotation columns of the pose and of the pose
      // before (Q31 = Q13^T), the tail -- stored at once; its accumulator row is free again for the rotation rows
      double x[LD];
      x_trn(t_own, x);
#pragma unroll
      for (int j = 0; j < LD; ++j) acct[j] = s0[D] * x[j];
      x_trn(t_nxt, x);
#pragma unroll
      for (int j = 0; j < LD; ++j) acct[j] = fma(s1[D], x[j], acct[j]);
      x_trn(t_prv, x);
#pragma unroll
      for (int j = 0; j < LD; ++j) acct[j] = fma(ps1[D], x[j], acct[j]);
#pragma unroll
      for (int c = 0; c < D; ++c) {
        x_rot(prv_row + c, x);
#pragma unroll
        for (int j = 0; j < LD; ++j) acct[j] = fma(ps1[c], x[j], acct[j]);
        x_rot(own_row + c, x);
#pragma unroll
        for (int j = 0; j < LD; ++j) acct[j] = fma(s0[c], x[j], acct[j]);
      }
      translation_tail();
      if (lane < sd.nrows) store_row<LD>(A.out + static_cast<size_t>(t_own) * LD, acct);
      phase_fence();
    }
    {
      double x[LD];
      x_trn(t_own, x);
#pragma unroll
      for (int j = 0; j < LD; ++j) {
#pragma unroll
        for (int a = 0; a < D; ++a) acc[a][j] = fma(s0[a], x[j], acc[a][j]);
        if constexpr (kFuseT) acct[j] = fma(s0[D], x[j], acct[j]);
      }
      x_trn(t_nxt, x);
#pragma unroll
      for (int j = 0; j < LD; ++j) {
#pragma unroll
        for (int a = 0; a < D; ++a) acc[a][j] = fma(s1[a], x[j], acc[a][j]);
        if constexpr (kFuseT) acct[j] = fma(s1[D], x[j], acct[j]);
      }
    }
    if constexpr (kFuseT) {
      double x[LD];
      x_trn(t_prv, x);
#pragma unroll
      for (int j = 0; j < LD; ++j) acct[j] = fma(ps1[D], x[j], acct[j]);
    }
    // (c) the next pose's block, (d) the previous pose's block = the transposed next block of the lane before
    phase_fence();
    double nxt[D][D];  // [c][a] = Q(rot(P)_a, rot(P+1)_c)
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int a = 0; a < D; ++a) nxt[c][a] = fixed(2 * (D + 1) + c * D + a);
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double x[LD];
      x_rot(nxt_row + c, x);
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int j = 0; j < LD; ++j) acc[a][j] = fma(nxt[c][a], x[j], acc[a][j]);
    }
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double x[LD];
      x_rot(prv_row + c, x);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double pv = before(nxt[a][c], 2 * (D + 1) + a * D + c, a * D + c);  // Q(rot(P)_a, rot(P-1)_c)
#pragma unroll
        for (int j = 0; j < LD; ++j) acc[a][j] = fma(pv, x[j], acc[a][j]);
      }
      if constexpr (kFuseT) {
#pragma unroll
        for (int j = 0; j < LD; ++j) acct[j] = fma(ps1[c], x[j], acct[j]);
      }
    }
    // (e) the pose's own block; the translation row's share of these columns is Q(t_P, rot(P)_c) = s0[c]
    phase_fence();
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double v[D], x[LD];
#pragma unroll
      for (int a = 0; a < D; ++a) v[a] = fixed(2 * (D + 1) + D * D + c * D + a);
      x_rot(own_row + c, x);
#pragma unroll
      for (int j = 0; j < LD; ++j) {
#pragma unroll
        for (int a = 0; a < D; ++a) acc[a][j] = fma(v[a], x[j], acc[a][j]);
        if constexpr (kFuseT) acct[j] = fma(s0[c], x[j], acct[j]);
      }
    }
  }
  CORA_PHASE(1);  // fixed slots done
  auto slot = [&](int k) {
    double v[D];
    const int32_t c = stream_load(cpg + static_cast<size_t>(k) * kWave);
#pragma unroll
    for (int a = 0; a < D; ++a) v[a] = stream_load(vpg + (static_cast<size_t>(k) * D + a) * kWave);
    slot_apply(c, v);
  };
  // slots in flight per lane: CORA_POSE_UNROLL_WIN with the windows; without, 3 up to a row stride of 6, 2 above
  // (register pressure: p = 10 Hvp 40.1 -> 39.4 us)
  if (kWin) {
#pragma unroll CORA_POSE_UNROLL_WIN
    for (int k = 0; k < sd.width; ++k) slot(k);
  } else {
    constexpr int kSlotsInFlight = LD <= 6 ? CORA_POSE_UNROLL : 2;
#pragma unroll kSlotsInFlight
    for (int k = 0; k < sd.width; ++k) slot(k);
  }
  CORA_PHASE(2);  // general slots done
  if (kFuseT && chain) translation_tail();
  CORA_PHASE(3);  // tail done
  // the slice's result rows leave through LDS: rotation rows and translation rows are runs of consecutive rows, stored as
  // full 512-byte pieces instead of 16-byte pieces of 64 different lines
  auto staged_store = [&] {
    __syncthreads();
    CORA_PHASE(4);  // projected
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int j = 0; j < LD; ++j) win[(lane * D + a) * LD + j] = acc[a][j];
    if (kFuseT && chain) {
#pragma unroll
      for (int j = 0; j < LD; ++j) win[kYEl + lane * LD + j] = acct[j];
    }
    __syncthreads();
    double *__restrict__ op = A.out + static_cast<size_t>(sd.row0) * LD;
#pragma unroll
    for (int i = 0; i < kYIt; ++i) {
      const int e = 2 * (i * kWave + lane), n = sd.nrows * D * LD;
      if (e + 1 < n) {
        Pair8 v{win[e], win[e + 1]};
        *reinterpret_cast<Pair8 *>(op + e) = v;
      } else if (e < n) {
        op[e] = win[e];
      }
    }
    if (kFuseT && chain) {  // the slice's translation rows: consecutive rows as well
      double *__restrict__ ot = A.out + static_cast<size_t>(A.win_trn_lo + sd.aux0) * LD;
      constexpr int kTIt = (LD + 1) / 2;
#pragma unroll
      for (int i = 0; i < kTIt; ++i) {
        const int e = 2 * (i * kWave + lane), n = sd.nrows * LD;
        if (e + 1 < n) {
          Pair8 v{win[kYEl + e], win[kYEl + e + 1]};
          *reinterpret_cast<Pair8 *>(ot + e) = v;
        } else if (e < n) {
          ot[e] = win[kYEl + e];
        }
      }
    }
  };
  if constexpr (kCoopT) if (kCoop) {
    double xo[D][LD];  // the pose's own rows of X (inside the rotation window; lanes past nrows read rows they ignore)
#pragma unroll
    for (int b = 0; b < D; ++b) {
      const int l = sd.row0 + lane * D + b - w0;
#pragma unroll
      for (int j = 0; j < LD; ++j) xo[b][j] = win[l * LD + j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kYIt; ++i) {
      const int e = 2 * (i * kWave + lane);
      if (e < kYEl) win[e] = ystage[2 * i];
      if (e + 1 < kYEl) win[e + 1] = ystage[2 * i + 1];
    }
#pragma unroll
    for (int i = 0; i < kLIt; ++i) {
      const int e = 2 * (i * kWave + lane);
      if (e < kLEl) win[kYEl + e] = lstage[2 * i];
      if (e + 1 < kLEl) win[kYEl + e + 1] = lstage[2 * i + 1];
    }
    __syncthreads();
    double y[D][LD];
#pragma unroll
    for (int b = 0; b < D; ++b) {
#pragma unroll
      for (int j = 0; j < LD; ++j) y[b][j] = win[(lane * D + b) * LD + j];
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double lam = win[kYEl + lane * (D * D) + a * D + b];
#pragma unroll
        for (int j = 0; j < LD; ++j) acc[a][j] = fma(-lam, xo[b][j], acc[a][j]);
      }
    }
    stiefel_project_thread<LD, D>(y, acc);
    double kap = 0.0;
    if (EPI == EPI_HVP_K && lane < sd.nrows) {
#pragma unroll
      for (int a = 0; a < D; ++a) kap += dot_row<LD>(xo[a], acc[a]);
    }
    staged_store();
    return kap + kap_t;
  }
  const bool active = lane < sd.nrows;
  if (!kStaged && !active) return 0.0;
  if (!kStaged && kFuseT && chain) store_row<LD>(A.out + static_cast<size_t>(t_own) * LD, acct);
  const size_t prow = static_cast<size_t>(sd.row0) + static_cast<size_t>(lane) * D;
  constexpr bool kKeepX = EPI == EPI_HVP_K && D * LD <= 18;  // the pose's own rows of X stay in registers for <X, out>
  double xs[kKeepX ? D : 1][LD];
  if (EPI != EPI_NONE && active) {
    const double *Lp = A.lam_st + static_cast<size_t>(sd.aux0 + lane) * (D * D);
#pragma unroll
    for (int b = 0; b < D; ++b) {
      double x[LD];
      if (kWinLD && kWin) {  // the pose's own rows are inside the rotation window
        const int l = sd.row0 + lane * D + b - w0;
#pragma unroll
        for (int j = 0; j < LD; ++j) x[j] = win[l * LD + j];
      } else {
        load_row<LD>(X + (prow + b) * LD, x);
      }
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double lam = Lp[a * D + b];
#pragma unroll
        for (int j = 0; j < LD; ++j) acc[a][j] = fma(-lam, x[j], acc[a][j]);
      }
      if (kKeepX) {
#pragma unroll
        for (int j = 0; j < LD; ++j) xs[kKeepX ? b : 0][j] = x[j];
      }
    }
    if (EPI >= EPI_HVP) {
      double y[D][LD];
#pragma unroll
      for (int b = 0; b < D; ++b) load_row<LD>(A.Y + (prow + b) * LD, y[b]);
      stiefel_project_thread<LD, D>(y, acc);
    }
  }
  double kap = 0.0;
  if constexpr (kStaged) {
    staged_store();
    if (!active) return 0.0;
  } else {
#pragma unroll
    for (int a = 0; a < D; ++a) store_row<LD>(A.out + (prow + a) * LD, acc[a]);
  }
  if (EPI == EPI_HVP_K) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
      if (kKeepX) {
        kap += dot_row<LD>(xs[kKeepX ? a : 0], acc[a]);
      } else {
        double x[LD];
        load_row<LD>(X + (prow + a) * LD, x);
        kap += dot_row<LD>(x, acc[a]);
      }
    }
  }
  return kap + kap_t;
}

// One wavefront, one slice (lane = row, or lane = pose for the rotation rows).  Returns the lane's share of
// <X, out> over the rows it wrote (EPI_HVP_K; 0 otherwise).
template <int LD, int D, int EPI>
__device__ __forceinline__ double slice_wave(const SpmmArgs &A, const SliceDesc sd, int lane) {
  if ((sd.type & kSliceTypeMask) == kSliceStiefel) {
    if (sd.nrows & kSliceRemoteTailOnly) return remote_tail_only<LD, D, EPI>(A, sd, lane);
    return A.win_on ? pose_slice<LD, D, EPI, true>(A, sd, lane) : pose_slice<LD, D, EPI, false>(A, sd, lane);
  }
  const double *__restrict__ vp = A.sval + sd.off + lane;
  const int32_t *__restrict__ cp = A.scol + sd.coff + lane;
  const double *__restrict__ X = A.X;

  // the epilogue's operands do not depend on the slots: requested first, so that they travel with the slots' own loads
  // (a wavefront of a row slice is a chain of dependent latencies: index -> row of X -> epilogue operands was one more)
  const bool active = lane < sd.nrows;
  const bool oblique = sd.type == kSliceOblique;
  const size_t row = !active ? static_cast<size_t>(A.win_rot_lo)
                             : (sd.type == kSliceEuclidPerm ? static_cast<size_t>(A.perm[sd.row0 + lane]) : static_cast<size_t>(sd.row0) + lane);
  double lam = 0.0, xo[LD], yo[LD];
  if (EPI != EPI_NONE && (oblique || EPI == EPI_HVP_K)) load_row<LD>(X + row * LD, xo);
  if (EPI != EPI_NONE && oblique) {
    if (active) lam = A.lam_ob[sd.aux0 + lane];
    if (EPI >= EPI_HVP) load_row<LD>(A.Y + row * LD, yo);
  }

  double acc[LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) acc[j] = 0.0;

#pragma unroll CORA_ROW_UNROLL
  for (int k = 0; k < sd.width; ++k) {
    const double v = stream_load(vp + static_cast<size_t>(k) * kWave);
    const int32_t c = stream_load(cp + static_cast<size_t>(k) * kWave);
    double x[LD];
    load_row<LD>(X + static_cast<size_t>(c) * LD, x);
#pragma unroll
    for (int j = 0; j < LD; ++j) acc[j] = fma(v, x[j], acc[j]);
  }

  if (!active) return 0.0;
  double kap = 0.0;
  if (oblique && EPI != EPI_NONE) {
#pragma unroll
    for (int j = 0; j < LD; ++j) acc[j] = fma(-lam, xo[j], acc[j]);
    if (EPI >= EPI_HVP) {
      const double ip = dot_row<LD>(yo, acc);
#pragma unroll
      for (int j = 0; j < LD; ++j) acc[j] = fma(-ip, yo[j], acc[j]);
    }
  }
  store_row<LD>(A.out + row * LD, acc);
  if (EPI == EPI_HVP_K) kap = dot_row<LD>(xo, acc);
  return kap;
}

#if CORA_TU & 1
#ifdef CORA_SPMM_TIMES
__device__ unsigned long long g_spmm_times[3 * kSpmmTimesMax];
#endif
#ifndef CORA_SPMM_WAVES_PER_EU
#define CORA_SPMM_WAVES_PER_EU 3
#endif
#ifndef CORA_SPMM_MIN_WAVES_PER_EU
#define CORA_SPMM_MIN_WAVES_PER_EU 2  // (up to a row stride of CORA_SPMM_MIN2_MAX_LD; above, the accumulators alone need more)
#endif
#ifndef CORA_SPMM_MIN2_MAX_LD
#define CORA_SPMM_MIN2_MAX_LD 10
#endif
// The kernel is latency bound unless each wave keeps many loads in flight, so
// let the register allocator spend registers instead of squeezing for full occupancy (round 1: 8 waves per SIMD at
// 64 registers 31.3 us, 2 waves 20.4 us on the 10^5-pose graph).  With the X window the balance moved: what the memory
// side sustains is requests in flight per CU, and three lighter wavefronts per SIMD (<= 168 registers, two slots per
// trip) beat two heavier ones: Hvp 22.2 -> 21.4 us, HBM-resident 30.9 -> 29.2 us, inside the STPCG loop 27.8 -> 25.2 us;
// four per SIMD are no better (profiles/r03_kernel_evolution.md).
template <int LD, int D, int EPI>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LD <= CORA_SPMM_MIN2_MAX_LD ? CORA_SPMM_MIN_WAVES_PER_EU : 1, CORA_SPMM_WAVES_PER_EU)))
void k_spmm(const SpmmArgs A) {
  // one wavefront per block: the dispatcher balances the (uneven) slices.
  // EPI_HVP_K: the wavefront also leaves sum <X[row], out[row]> over the rows it finished in
  // kappa_partial[blockIdx.x] (every block writes its slot; k_kappa_finish adds them in slot order).
  constexpr bool KAPPA = EPI == EPI_HVP_K;
  const int lane = threadIdx.x;
  double kap = 0.0;
#ifdef CORA_SPMM_TIMES
  const unsigned long long dbg_t0 = wall_clock64();
  unsigned long long dbg_meta = 0xFFull << 32;
#endif
  if (static_cast<int>(blockIdx.x) < A.n_chunks) {
    // chunk blocks, also one contiguous range of the (column-sorted) launch order per XCD
    const int tc = static_cast<int>(blockIdx.x);
    const int cper = A.n_chunks >> 3;  // n_chunks is a multiple of 8
    const int pos = (tc & 7) * cper + (tc >> 3);
    if (pos < A.n_real_chunks) long_chunk_wave<LD, KAPPA>(A, A.chunk_order[pos]);
  } else {
    // Slice blocks: XCD x (= blockIdx % 8, observed dispatch order; speed only)
    // walks its own contiguous eighth of the slice list, so neighbouring slices
    // share one L2.  n_chunks is padded to a multiple of 8 by the launcher.
    const int t = static_cast<int>(blockIdx.x) - A.n_chunks;
    const int per_xcd = (A.n_slices + 7) >> 3;
    const int s = (t & 7) * per_xcd + (t >> 3);
    if ((t >> 3) < per_xcd && s < A.n_slices) kap = slice_wave<LD, D, EPI>(A, A.slices[s], lane);
#ifdef CORA_SPMM_TIMES
    if ((t >> 3) < per_xcd && s < A.n_slices)
      dbg_meta = (static_cast<unsigned long long>(A.slices[s].type) << 32) | static_cast<unsigned>(A.slices[s].width);
#endif
  }
  if (KAPPA) {
    kap = wave_sum(kap);
    if (lane == 0) A.kappa_partial[blockIdx.x] = kap;
  }
#ifdef CORA_SPMM_TIMES
  // measurement build only (tools/spmm_timeline.py): start / end of every wavefront on the 100 MHz wall clock
  if (lane == 0 && blockIdx.x < kSpmmTimesMax) {
    g_spmm_times[3 * blockIdx.x] = dbg_t0;
    g_spmm_times[3 * blockIdx.x + 1] = wall_clock64();
    g_spmm_times[3 * blockIdx.x + 2] = dbg_meta;
  }
#endif
}

#endif  // CORA_TU & 1

#if CORA_TU & 2
// ---------------------------------------------------------------------------
// Row-unit kernels: one thread per pose (d x LD block), per range row or per
// translation row of the LOCAL shard.
// ---------------------------------------------------------------------------
struct Unit {
  int kind;      // 0 pose, 1 range, 2 translation, -1 none
  int idx;       // local index within its kind
  size_t row;    // internal row of the unit's first row
};

__device__ __forceinline__ Unit unit_of(const RowArgs &R, int64_t u) {
  Unit x;
  if (u < R.nl_poses) { x.kind = 0; x.idx = static_cast<int>(u); x.row = R.rot_base + static_cast<size_t>(u) * R.d; }
  else if (u < R.nl_poses + R.nl_ranges) { x.kind = 1; x.idx = static_cast<int>(u - R.nl_poses); x.row = R.rng_base + x.idx; }
  else if (u < R.nl_poses + R.nl_ranges + R.nl_trans) { x.kind = 2; x.idx = static_cast<int>(u - R.nl_poses - R.nl_ranges); x.row = R.trn_base + x.idx; }
  else { x.kind = -1; x.idx = 0; x.row = 0; }
  return x;
}

// After G = Q Y:  Lambda blocks (:1105-1131), grad = Proj_Y(G) (:772-780) and
// the partial sums of f = 1/2 <Y, G> (:759-762).
template <int LD, int D>
__global__ __launch_bounds__(256) void k_point_finish(const RowArgs R, const double *__restrict__ Y,
                                                      const double *__restrict__ G,
                                                      double *__restrict__ rgrad,
                                                      double *__restrict__ lam_st,
                                                      double *__restrict__ lam_ob,
                                                      double *__restrict__ partial) {
  __shared__ double sm[4];
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const Unit un = unit_of(R, u);
  double f = 0.0;
  if (un.kind == 0) {
    double y[D][LD], g[D][LD];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      load_row<LD>(Y + (un.row + a) * LD, y[a]);
      load_row<LD>(G + (un.row + a) * LD, g[a]);
      f += dot_row<LD>(y[a], g[a]);
    }
    double m[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) m[a][b] = dot_row<LD>(g[a], y[b]);
    double *L = lam_st + static_cast<size_t>(un.idx) * (D * D);
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        const double s = 0.5 * (m[a][b] + m[b][a]);
        L[a * D + b] = s;
#pragma unroll
        for (int c = 0; c < LD; ++c) g[a][c] = fma(-s, y[b][c], g[a][c]);
      }
    // note: g[a] is updated with y only, so the in-place update above is exact
#pragma unroll
    for (int a = 0; a < D; ++a) store_row<LD>(rgrad + (un.row + a) * LD, g[a]);
  } else if (un.kind == 1) {
    double y[LD], g[LD];
    load_row<LD>(Y + un.row * LD, y);
    load_row<LD>(G + un.row * LD, g);
    const double lam = dot_row<LD>(y, g);
    f += lam;
    lam_ob[un.idx] = lam;
#pragma unroll
    for (int c = 0; c < LD; ++c) g[c] = fma(-lam, y[c], g[c]);
    store_row<LD>(rgrad + un.row * LD, g);
  } else if (un.kind == 2) {
    double y[LD], g[LD];
    load_row<LD>(Y + un.row * LD, y);
    load_row<LD>(G + un.row * LD, g);
    f += dot_row<LD>(y, g);
    store_row<LD>(rgrad + un.row * LD, g);
  }
  const double tot = block_sum_256(f, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = 0.5 * tot;
}

// out = Proj_Y(scale .* V)   (scale == nullptr -> 1): tangent_space_projection
// (:782-820) and the Jacobi `precon` closure (:888-889 + src/CORA.cpp:86-92).
template <int LD, int D>
__global__ __launch_bounds__(256) void k_tangent_project(const RowArgs R, const double *__restrict__ Y,
                                                         const double *__restrict__ V,
                                                         const double *__restrict__ scale,
                                                         double *__restrict__ out) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const Unit un = unit_of(R, u);
  if (un.kind == 0) {
    double y[D][LD], v[D][LD];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      load_row<LD>(Y + (un.row + a) * LD, y[a]);
      load_row<LD>(V + (un.row + a) * LD, v[a]);
      if (scale) {
        const double sc = scale[un.row + a - R.base];
#pragma unroll
        for (int c = 0; c < LD; ++c) v[a][c] *= sc;
      }
    }
    stiefel_project_thread<LD, D>(y, v);
#pragma unroll
    for (int a = 0; a < D; ++a) store_row<LD>(out + (un.row + a) * LD, v[a]);
  } else if (un.kind == 1) {
    double y[LD], v[LD];
    load_row<LD>(Y + un.row * LD, y);
    load_row<LD>(V + un.row * LD, v);
    if (scale) {
      const double sc = scale[un.row - R.base];
#pragma unroll
      for (int c = 0; c < LD; ++c) v[c] *= sc;
    }
    const double ip = dot_row<LD>(y, v);
#pragma unroll
    for (int c = 0; c < LD; ++c) v[c] = fma(-ip, y[c], v[c]);
    store_row<LD>(out + un.row * LD, v);
  } else if (un.kind == 2) {
    double v[LD];
    load_row<LD>(V + un.row * LD, v);
    if (scale) {
      const double sc = scale[un.row - R.base];
#pragma unroll
      for (int c = 0; c < LD; ++c) v[c] *= sc;
    }
    store_row<LD>(out + un.row * LD, v);
  }
}

// Polar factor of a d x LD block by one-sided (Hestenes) Jacobi on its rows:
// rotations J with (J A) having orthogonal rows;  A = J^T Sigma U  =>  polar = J^T U.
// Works on A directly (no Gram matrix), so it is as accurate as the reference's
// Eigen::JacobiSVD route (src/StiefelProduct.cpp:8-36: U V^T of the thin SVD).
template <int LD, int D>
__device__ __forceinline__ void polar_rows(double (&a)[D][LD]) {
  double J[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) J[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < D - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < D; ++q) {
        const double alpha = dot_row<LD>(a[p], a[p]);
        const double beta = dot_row<LD>(a[q], a[q]);
        const double gamma = dot_row<LD>(a[p], a[q]);
        const double denom = sqrt(alpha * beta);
        if (gamma != 0.0 && denom > 0.0) {
          off = fmax(off, fabs(gamma) / denom);
          const double zeta = (beta - alpha) / (2.0 * gamma);
          const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
          for (int c = 0; c < LD; ++c) {
            const double x = a[p][c], y = a[q][c];
            a[p][c] = cs * x - sn * y;
            a[q][c] = sn * x + cs * y;
          }
#pragma unroll
          for (int c = 0; c < D; ++c) {
            const double x = J[p][c], y = J[q][c];
            J[p][c] = cs * x - sn * y;
            J[q][c] = sn * x + cs * y;
          }
        }
      }
    if (off < 1e-15) break;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const double nrm = sqrt(dot_row<LD>(a[i], a[i]));
    const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
#pragma unroll
    for (int c = 0; c < LD; ++c) a[i][c] *= inv;
  }
  double o[D][LD];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int c = 0; c < LD; ++c) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fma(J[k][i], a[k][c], s);
      o[i][c] = s;
    }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int c = 0; c < LD; ++c) a[i][c] = o[i][c];
}

// out = projectToManifold(A + alpha V)  (V == nullptr -> projectToManifold(A)):
// Problem::projectToManifold :905-934 and Problem::retract :936-938.
template <int LD, int D>
__global__ __launch_bounds__(256) void k_project_manifold(const RowArgs R, const double *__restrict__ A,
                                                          const double *__restrict__ V, double alpha,
                                                          double *__restrict__ out) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const Unit un = unit_of(R, u);
  if (un.kind == 0) {
    double a[D][LD];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      load_row<LD>(A + (un.row + i) * LD, a[i]);
      if (V) {
        double v[LD];
        load_row<LD>(V + (un.row + i) * LD, v);
#pragma unroll
        for (int c = 0; c < LD; ++c) a[i][c] = fma(alpha, v[c], a[i][c]);
      }
    }
    polar_rows<LD, D>(a);
#pragma unroll
    for (int i = 0; i < D; ++i) store_row<LD>(out + (un.row + i) * LD, a[i]);
  } else if (un.kind == 1 || un.kind == 2) {
    double a[LD];
    load_row<LD>(A + un.row * LD, a);
    if (V) {
      double v[LD];
      load_row<LD>(V + un.row * LD, v);
#pragma unroll
      for (int c = 0; c < LD; ++c) a[c] = fma(alpha, v[c], a[c]);
    }
    if (un.kind == 1) {
      const double nrm = sqrt(dot_row<LD>(a, a));
      if (nrm > 0.0) {
        const double inv = 1.0 / nrm;
#pragma unroll
        for (int c = 0; c < LD; ++c) a[c] *= inv;
      }
    }
    store_row<LD>(out + un.row * LD, a);
  }
}

// ---------------------------------------------------------------------------
// flat vector kernels over the local shard (contiguous doubles)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_axpby1(int64_t n, double a, const double *__restrict__ x, double b,
                                                double *__restrict__ y) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const double yv = (b != 0.0) ? y[i] : 0.0;
    y[i] = fma(a, x[i], b * yv);
  }
}

__global__ __launch_bounds__(256) void k_axpby(int64_t n2, double a, const double2 *__restrict__ x,
                                               double b, double2 *__restrict__ y) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n2;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const double2 xv = x[i];
    double2 yv = make_double2(0.0, 0.0);
    if (b != 0.0) yv = y[i];
    y[i] = make_double2(fma(a, xv.x, b * yv.x), fma(a, xv.y, b * yv.y));
  }
}

// y1 += a1 x1 and y2 += a2 x2 in one pass (the two updates of an STPCG iteration, s += alpha p and
// r += alpha Hp): one launch floor instead of two
__global__ __launch_bounds__(256) void k_axpy2(int64_t n, double a1, const double *__restrict__ x1,
                                               double *__restrict__ y1, double a2,
                                               const double *__restrict__ x2, double *__restrict__ y2) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    y1[i] = fma(a1, x1[i], y1[i]);
    y2[i] = fma(a2, x2[i], y2[i]);
  }
}

// the two vector updates of a device-resident STPCG iteration; coefficients come from the device state
__global__ __launch_bounds__(256) void k_stpcg_update(int64_t n, const StpcgState *__restrict__ S,
                                                      const double *__restrict__ p, const double *__restrict__ Hp,
                                                      double *__restrict__ s, double *__restrict__ r) {
  const double cs = S->coef_s, cr = S->coef_r;
  if (cs == 0.0 && cr == 0.0) return;  // solve already finished: enqueued ahead of the host's check
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    s[i] = fma(cs, p[i], s[i]);
    r[i] = fma(cr, Hp[i], r[i]);
  }
}

__global__ __launch_bounds__(256) void k_stpcg_direction(int64_t n, const StpcgState *__restrict__ S,
                                                         const double *__restrict__ v, double *__restrict__ p) {
  const double cv = S->coef_v, cb = S->coef_beta;
  if (cv == 0.0 && cb == 1.0) return;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256)
    p[i] = fma(cv, v[i], cb * p[i]);
}

// x[row of API variable i][col] = a number in (-1, 1) that depends on (seed, i, col) only (splitmix64 of the triple) -- what an
// upload of a host-drawn N x k block would leave, whatever the partition: the start block of an eigensolver run, made
// where it is used (450 k x 4 doubles from the host were 3 ms to draw and 6 ms to upload)
__global__ __launch_bounds__(256) void k_fill_random(int64_t N, int ld, int k, unsigned long long seed,
                                                     const int32_t *__restrict__ api2int, double *__restrict__ x) {
  const int64_t n = N * ld;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < n; t += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t i = t / ld;
    const int col = static_cast<int>(t - i * ld);
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (static_cast<unsigned long long>(i) * 32ull + static_cast<unsigned long long>(col) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double u = static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0);  // [0, 1)
    x[static_cast<size_t>(api2int[i]) * ld + col] = col < k ? 2.0 * u - 1.0 : 0.0;
  }
}

__global__ __launch_bounds__(256) void k_scale_rows(int64_t rows, int ld, const double *__restrict__ scale,
                                                    const double *__restrict__ x, double *__restrict__ y) {
  const int64_t n = rows * ld;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256)
    y[i] = scale[i / ld] * x[i];
}

#endif  // CORA_TU & 2
// Tail of the inner-product kernels: every block publishes its partial sums write-through, takes a
// ticket, and the last block to arrive adds all partials in block order (deterministic) and writes the
// results to D.out -- pinned host memory, so the caller only has to wait for the stream.  Same
// inter-workgroup hand-off as the long rows of k_spmm.
__device__ __forceinline__ void dots_finish(const DotArgs &D, const double (&acc)[4], double *sm, double kappa = 0.0) {
  __shared__ int s_last;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < D.count) {
      const double t = block_sum_256(acc[j], sm);
      if (threadIdx.x == 0)
        __hip_atomic_store(D.partial + static_cast<size_t>(j) * gridDim.x + blockIdx.x, t, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(D.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == gridDim.x - 1);
    if (s_last) __hip_atomic_store(D.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  for (int j = 0; j < D.count; ++j) {
    double s = 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256)
      s += __hip_atomic_load(D.partial + static_cast<size_t>(j) * gridDim.x + b, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
    const double t = block_sum_256(s, sm);
    if (threadIdx.x == 0) D.out[j] = t;
    if (threadIdx.x == 0) sm[4 + j] = t;
  }
  if (threadIdx.x == 0 && D.mode == DOTS_STPCG_KAPPA) stpcg_after_kappa(*D.st, sm[4]);
  if (threadIdx.x == 0 && D.mode == DOTS_STPCG_RR) stpcg_after_rr(*D.st, sm[4]);
  if (threadIdx.x == 0 && D.mode == DOTS_STPCG_KAPPA_RR) {  // k_kappa_residual: every block used the same kappa
    stpcg_after_kappa(*D.st, kappa);
    stpcg_after_rr(*D.st, sm[4]);
  }
  if (threadIdx.x == 0 && D.mode == DOTS_STPCG_RV) {
    stpcg_after_rv(*D.st, sm[4]);
    *D.st_host = *D.st;  // pinned mirror for the host's (infrequent) look
  }
  if (threadIdx.x == 0 && D.mode == DOTS_STPCG_BETA) {
    stpcg_after_rr(*D.st, sm[4]);
    stpcg_after_rv(*D.st, sm[5]);
    *D.st_host = *D.st;
  }
  if (threadIdx.x == 0 && D.seq_out) {  // results first, then the sequence number the host spins on
    unsigned long long seq = D.seq;
    if (D.seq_counter) *D.seq_counter = seq = *D.seq_counter + 1;
    __threadfence_system();
    __hip_atomic_store(D.seq_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#if CORA_TU & 2
// r += coef_r Hp with <r, r> of the result in the same pass (the scalar step that follows it runs in the last block)
__global__ __launch_bounds__(256) void k_stpcg_residual(DotArgs D, const double2 *__restrict__ Hp, double2 *__restrict__ r) {
  __shared__ double sm[8];
  const double cr = D.st->coef_r;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < D.n2;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    double2 rv = r[i];
    if (cr != 0.0) {
      const double2 h = Hp[i];
      rv.x = fma(cr, h.x, rv.x);
      rv.y = fma(cr, h.y, rv.y);
      r[i] = rv;
    }
    acc[0] = fma(rv.x, rv.x, fma(rv.y, rv.y, acc[0]));
  }
  dots_finish(D, acc, sm);
}

// start of a solve: s = 0, r = g, p = -Pg
__global__ __launch_bounds__(256) void k_stpcg_init(int64_t n, const double *__restrict__ g, const double *__restrict__ Pg,
                                                    double *__restrict__ s, double *__restrict__ r, double *__restrict__ p) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    s[i] = 0.0;
    r[i] = g[i];
    p[i] = -Pg[i];
  }
}

// Partitioned handles: the inner products are summed over the ranks between the pass that forms them and the scalar
// step (an all-reduce of the device values on the handle's stream), so the step is a launch of its own: one thread.
//   what = 0: vals[0] = kappa;   what = 1: vals[0] = <r, r>, vals[1] = <r, v>, then the pinned mirror and its sequence number
__global__ void k_stpcg_scalar_step(int what, const double *__restrict__ vals, StpcgState *st, StpcgState *st_host,
                                    unsigned long long *seq_out, unsigned long long seq) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (what == 0) {
    stpcg_after_kappa(*st, vals[0]);
    return;
  }
  stpcg_after_rr(*st, vals[0]);
  stpcg_after_rv(*st, vals[1]);
  *st_host = *st;
  if (seq_out) {
    __threadfence_system();
    __hip_atomic_store(seq_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// kappa = <p, Hp> from the per-block partials of an EPI_HVP_K product, then r += coef_r Hp with <r, r> in the same
// launch: EVERY block adds the partials (same order, same bits; they sit in L2) and runs the scalar step on a private
// copy of the state to get its coefficient; the state itself is advanced once, by the last block to finish -- by then
// every block has read it.
__global__ __launch_bounds__(256) void k_kappa_residual(DotArgs D, const double *__restrict__ kpartial, int nk,
                                                        const double2 *__restrict__ Hp, double2 *__restrict__ r) {
  __shared__ double sm[8];
  __shared__ double s_kappa;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = threadIdx.x;
  for (; b + 768 < nk; b += 1024) {
    s0 += kpartial[b];
    s1 += kpartial[b + 256];
    s2 += kpartial[b + 512];
    s3 += kpartial[b + 768];
  }
  for (; b < nk; b += 256) s0 += kpartial[b];
  const double t = block_sum_256((s0 + s1) + (s2 + s3), sm);
  if (threadIdx.x == 0) s_kappa = t;
  __syncthreads();
  const double kappa = s_kappa;
  StpcgState L = *D.st;
  stpcg_after_kappa(L, kappa);
  const double cr = L.coef_r;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < D.n2;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    double2 rv = r[i];
    if (cr != 0.0) {
      const double2 h = Hp[i];
      rv.x = fma(cr, h.x, rv.x);
      rv.y = fma(cr, h.y, rv.y);
      r[i] = rv;
    }
    acc[0] = fma(rv.x, rv.x, fma(rv.y, rv.y, acc[0]));
  }
  dots_finish(D, acc, sm, kappa);
}

// The same pass for the iterations whose reductions are finished by the tail block of a later launch (RvTail::n_kappa,
// RvTail::n_rr: the one-explicit-inverse form): no ticket and no last block -- a block leaves its share of <r, r> in its
// slot and is done, the state is not touched here.  Its first elements are requested BEFORE the partials are added, so
// the launch is one round trip to memory and a block reduction, not three dependent ones.
__global__ __launch_bounds__(256) void k_kappa_residual_slots(const StpcgState *__restrict__ st, const double *__restrict__ kpartial,
                                                              int nk, int64_t n2, const double2 *__restrict__ Hp,
                                                              double2 *__restrict__ r, double *__restrict__ rr_slot) {
  __shared__ double sm[8];
  __shared__ double ksm[4];
  const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
  int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  double2 rv = make_double2(0.0, 0.0), h = make_double2(0.0, 0.0);
  if (i < n2) {
    rv = r[i];
    h = Hp[i];
  }
  const double kappa = kappa_sum_256(kpartial, nk, ksm);
  const double cr = stpcg_coef_r_after_kappa(st, kappa);
  double acc = 0.0;
  for (; i < n2; i += stride) {
    if (cr != 0.0) {
      rv.x = fma(cr, h.x, rv.x);
      rv.y = fma(cr, h.y, rv.y);
      r[i] = rv;
    }
    acc = fma(rv.x, rv.x, fma(rv.y, rv.y, acc));
    if (i + stride < n2) {
      rv = r[i + stride];
      h = Hp[i + stride];
    }
  }
  const double t = block_sum_256(acc, sm);
  if (threadIdx.x == 0) rr_slot[blockIdx.x] = t;
}

// s += coef_s p (the step of THIS iteration), then p = coef_v v + coef_beta p
__global__ __launch_bounds__(256) void k_stpcg_step_direction(int64_t n2, const StpcgState *__restrict__ S,
                                                              const double2 *__restrict__ v, double2 *__restrict__ p,
                                                              double2 *__restrict__ s) {
  const double cs = S->coef_s, cv = S->coef_v, cb = S->coef_beta;
  if (cs == 0.0 && cv == 0.0 && cb == 1.0) return;  // solve already finished: enqueued ahead of the host's check
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n2;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const double2 pv = p[i], vv = v[i];
    double2 sv = s[i];
    sv.x = fma(cs, pv.x, sv.x);
    sv.y = fma(cs, pv.y, sv.y);
    s[i] = sv;
    p[i] = make_double2(fma(cv, vv.x, cb * pv.x), fma(cv, vv.y, cb * pv.y));
  }
}

// v = Proj_Y(x) by row unit with <r, v> in the same pass (the preconditioned residual of an STPCG iteration: the
// scalar step that follows <r, v> runs in the last block)
template <int LD, int D>
__global__ __launch_bounds__(256) void k_tangent_project_dot(const RowArgs R, DotArgs Dt, const double *__restrict__ Y,
                                                             const double *V, const double *__restrict__ scale,
                                                             const double *__restrict__ r, double *out) {
  __shared__ double sm[8];
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const Unit un = unit_of(R, u);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (un.kind == 0) {
    double y[D][LD], v[D][LD];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      load_row<LD>(Y + (un.row + a) * LD, y[a]);
      load_row<LD>(V + (un.row + a) * LD, v[a]);
      if (scale) {
        const double sc = scale[un.row + a - R.base];
#pragma unroll
        for (int c = 0; c < LD; ++c) v[a][c] *= sc;
      }
    }
    stiefel_project_thread<LD, D>(y, v);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      load_row<LD>(r + (un.row + a) * LD, y[a]);  // y is dead: reuse its registers for the residual rows
      acc[0] += dot_row<LD>(y[a], v[a]);
      store_row<LD>(out + (un.row + a) * LD, v[a]);
    }
  } else if (un.kind == 1 || un.kind == 2) {
    double y[LD], v[LD];
    load_row<LD>(V + un.row * LD, v);
    if (scale) {
      const double sc = scale[un.row - R.base];
#pragma unroll
      for (int c = 0; c < LD; ++c) v[c] *= sc;
    }
    if (un.kind == 1) {
      load_row<LD>(Y + un.row * LD, y);
      const double ip = dot_row<LD>(y, v);
#pragma unroll
      for (int c = 0; c < LD; ++c) v[c] = fma(-ip, y[c], v[c]);
    }
    load_row<LD>(r + un.row * LD, y);
    acc[0] = dot_row<LD>(y, v);
    store_row<LD>(out + un.row * LD, v);
  }
  dots_finish(Dt, acc, sm);
}

// v = Proj_Y(x) by row unit, consumed at once: s += coef_s p, p = coef_v v + coef_beta p (v is not stored).  The last
// pass of an STPCG iteration whose <r, v> is already known (explicit-inverse plans: <r, v> = |W r|^2 from the first
// product of the solve), so the projection needs no reduction and the step / direction pass rides on it.
template <int LD, int D>
__global__ __launch_bounds__(256) void k_tangent_project_update(const RowArgs R, const StpcgState *__restrict__ S,
                                                                const double *__restrict__ Y, const double *__restrict__ X,
                                                                double *__restrict__ p, double *__restrict__ s) {
  const double cs = S->coef_s, cv = S->coef_v, cb = S->coef_beta;
  if (cs == 0.0 && cv == 0.0 && cb == 1.0) return;  // solve already finished: enqueued ahead of the host's check
  const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
  const Unit un = unit_of(R, u);
  auto update = [&](size_t row, const double (&v)[LD]) {
    double pv[LD], sv[LD];
    load_row<LD>(p + row * LD, pv);
    load_row<LD>(s + row * LD, sv);
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      sv[j] = fma(cs, pv[j], sv[j]);
      pv[j] = fma(cv, v[j], cb * pv[j]);
    }
    store_row<LD>(s + row * LD, sv);
    store_row<LD>(p + row * LD, pv);
  };
  if (un.kind == 0) {
    double y[D][LD], v[D][LD];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      load_row<LD>(Y + (un.row + a) * LD, y[a]);
      load_row<LD>(X + (un.row + a) * LD, v[a]);
    }
    stiefel_project_thread<LD, D>(y, v);
#pragma unroll
    for (int a = 0; a < D; ++a) update(un.row + a, v[a]);
  } else if (un.kind == 1 || un.kind == 2) {
    double y[LD], v[LD];
    load_row<LD>(X + un.row * LD, v);
    if (un.kind == 1) {
      load_row<LD>(Y + un.row * LD, y);
      const double ip = dot_row<LD>(y, v);
#pragma unroll
      for (int c = 0; c < LD; ++c) v[c] = fma(-ip, y[c], v[c]);
    }
    update(un.row, v);
  }
}

// scalar variant for odd lengths / 8-byte aligned shards
__global__ __launch_bounds__(256) void k_dots1(DotArgs D) {
  __shared__ double sm[8];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < D.n2;
       i += static_cast<int64_t>(gridDim.x) * 256) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < D.count) acc[j] = fma(D.a[j][i], D.b[j][i], acc[j]);
  }
  dots_finish(D, acc, sm);
}

// up to 4 inner products in one pass; partial[j * gridDim.x + block]
__global__ __launch_bounds__(256) void k_dots(DotArgs D) {
  __shared__ double sm[8];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < D.n2;
       i += static_cast<int64_t>(gridDim.x) * 256) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < D.count) {
        const double2 a = reinterpret_cast<const double2 *>(D.a[j])[i];
        const double2 b = reinterpret_cast<const double2 *>(D.b[j])[i];
        acc[j] = fma(a.x, b.x, fma(a.y, b.y, acc[j]));
      }
  }
  dots_finish(D, acc, sm);
}

// <r, r> and <r, v> in one pass that reads r once (the second reduction of an STPCG iteration)
__global__ __launch_bounds__(256) void k_dots_rr_rv(DotArgs D) {
  __shared__ double sm[8];
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const double2 *r = reinterpret_cast<const double2 *>(D.a[0]);
  const double2 *v = reinterpret_cast<const double2 *>(D.b[1]);
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < D.n2;
       i += static_cast<int64_t>(gridDim.x) * 256) {
    const double2 a = r[i], b = v[i];
    acc[0] = fma(a.x, a.x, fma(a.y, a.y, acc[0]));
    acc[1] = fma(a.x, b.x, fma(a.y, b.y, acc[1]));
  }
  dots_finish(D, acc, sm);
}

// out[j] = sum_b partial[j * nblocks + b]   (one 256-thread block, fixed order)
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partial, int nblocks,
                                                         int count, double *__restrict__ out) {
  __shared__ double sm[4];
  for (int j = 0; j < count; ++j) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[static_cast<size_t>(j) * nblocks + b];
    const double t = block_sum_256(s, sm);
    if (threadIdx.x == 0) out[j] = t;
  }
}

// Row moves of the multi-GPU exchange: mode 0 pack (out[k] = x[rows[k]]), 1 scatter (x[rows[k]] = in[k]),
// 2 copy (dst[rows[k]] = src[rows[k]]); one thread per double.
__global__ __launch_bounds__(256) void k_move_rows(int mode, int64_t n, int ld, const int32_t *__restrict__ rows,
                                                   const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t tot = n * ld;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < tot;
       t += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t k = t / ld, j = t - k * ld;
    const int64_t at = static_cast<int64_t>(rows[k]) * ld + j;
    if (mode == 0) dst[t] = src[at];
    else if (mode == 1) dst[at] = src[t];
    else dst[at] = src[at];
  }
}

// The exchange of a partitioned handle's product, both ends (capi.hip, cora_native_comm::exchange_product):
//   pack  : dst = [ the n exported rows of src | ztail zeros ] -- the zeros are the slots of the distributed long rows'
//           partial sums, which the chunk launch that follows fills where this rank has a share
//   unpack: the gathered buffers of all ranks, `stride` doubles each = [ e_max rows | n_long slots ]: rows go to
//           X[recv_idx], and the owner of long row j adds slot j of every rank IN RANK ORDER (the same bits whatever the
//           transport) into its row of the result; with kappa != nullptr every rank writes the row's kappa slot.
__global__ __launch_bounds__(256) void k_exchange_pack(int64_t n, int ld, const int32_t *__restrict__ rows, int64_t ztail,
                                                       const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t tot = n * ld;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < tot + ztail;
       t += static_cast<int64_t>(gridDim.x) * 256) {
    if (t >= tot) {
      dst[t] = 0.0;
      continue;
    }
    const int64_t k = t / ld, j = t - k * ld;
    dst[t] = src[static_cast<int64_t>(rows[k]) * ld + j];
  }
}
// Packed all-gather of one contiguous piece of every shard (the translation rows of the implicit formulation's solve):
// recv holds `maxn` rows per rank; rank r's piece is meta[2 r + 1] rows that belong at row  r * shard_rows + meta[2 r]  of X.
__global__ __launch_bounds__(256) void k_scatter_shard_rows(int world, int rank, int64_t maxn, int ld, int64_t shard_rows,
                                                            const int64_t *__restrict__ meta, const double *__restrict__ recv,
                                                            double *__restrict__ X) {
  const int64_t per = maxn * ld, tot = per * world;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < tot; t += static_cast<int64_t>(gridDim.x) * 256) {
    const int r = static_cast<int>(t / per);
    const int64_t k = t - r * per;
    if (r == rank || k >= meta[2 * r + 1] * ld) continue;
    X[(r * shard_rows + meta[2 * r]) * ld + k] = recv[t];
  }
}
__global__ __launch_bounds__(256) void k_exchange_unpack(int world, int64_t e_max, int n_long, int ld, int64_t stride, int scatter_blocks,
                                                         const int32_t *__restrict__ recv_idx, const double *__restrict__ recv,
                                                         double *__restrict__ X, int rank, const int32_t *__restrict__ long_rows,
                                                         const int32_t *__restrict__ long_owner, double *__restrict__ out,
                                                         double *__restrict__ kappa) {
  if (static_cast<int>(blockIdx.x) < scatter_blocks) {
    const int64_t per = e_max * ld, tot = per * world;
    for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < tot; t += static_cast<int64_t>(scatter_blocks) * 256) {
      const int64_t r = t / per, w = t - r * per, k = t / ld, j = t - k * ld;
      X[static_cast<int64_t>(recv_idx[k]) * ld + j] = recv[r * stride + w];
    }
    return;
  }
  const int j = static_cast<int>(blockIdx.x) - scatter_blocks, lane = threadIdx.x;
  if (j >= n_long || lane >= 64) return;
  double v = 0.0, k = 0.0;
  if (lane < ld)
    for (int r = 0; r < world; ++r) v += recv[r * stride + (e_max + j) * ld + lane];
  if (long_owner[j] == rank && lane < ld) {
    const size_t at = static_cast<size_t>(long_rows[j]) * ld + lane;
    out[at] = v;
    if (kappa) k = v * X[at];   // (a landmark's own row of X is this rank's: the scatter above does not write it)
  }
  if (kappa) {
    k = wave_sum(k);
    if (lane == 0) kappa[j] = k;
  }
}

// Distributed long rows of a partitioned handle, after their partial sums have been added over the ranks: the owner of
// long row j copies slot j to its row of the result; with kappa != nullptr EVERY rank writes the row's kappa slot
// (<X[row], out[row]> on the owner, 0 elsewhere -- every slot of a launch is written).  One wavefront per row.
__global__ __launch_bounds__(64) void k_long_finish(int n_long, int ld, int rank, const int32_t *__restrict__ rows,
                                                    const int32_t *__restrict__ owner, const double *__restrict__ slots,
                                                    const double *__restrict__ X, double *__restrict__ out,
                                                    double *__restrict__ kappa) {
  const int j = blockIdx.x, lane = threadIdx.x;
  if (j >= n_long) return;
  const bool mine = owner[j] == rank;
  double k = 0.0;
  if (mine && lane < ld) {
    const size_t at = static_cast<size_t>(rows[j]) * ld + lane;
    const double v = slots[static_cast<size_t>(j) * ld + lane];
    out[at] = v;
    if (kappa) k = v * X[at];
  }
  if (kappa) {
    k = wave_sum(k);
    if (lane == 0) kappa[j] = k;
  }
}

// NaN guard of Problem::precondition (:898-901): flag != 0 if any NaN.
__global__ __launch_bounds__(256) void k_has_nan(int64_t n, const double *__restrict__ x, int *flag) {
  int bad = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n;
       i += static_cast<int64_t>(gridDim.x) * 256)
    bad |= (x[i] != x[i]);
  if (bad) atomicOr(flag, 1);
}

// host column-major (N x k, leading dimension N) -> resident layout
__global__ __launch_bounds__(256) void k_upload(int64_t N, int k, int ld, const double *__restrict__ src,
                                                const int32_t *__restrict__ api2int,
                                                double *__restrict__ dst) {
  const int64_t tot = N * k;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < tot;
       t += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t c = t / N, i = t - c * N;
    dst[static_cast<size_t>(api2int[i]) * ld + c] = src[t];
  }
}

__global__ __launch_bounds__(256) void k_download(int64_t N, int k, int ld, const double *__restrict__ src,
                                                  const int32_t *__restrict__ api2int,
                                                  double *__restrict__ dst) {
  const int64_t tot = N * k;
  for (int64_t t = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; t < tot;
       t += static_cast<int64_t>(gridDim.x) * 256) {
    const int64_t c = t / N, i = t - c * N;
    dst[t] = src[static_cast<size_t>(api2int[i]) * ld + c];
  }
}

struct GramBatch {  // up to 16 products A_e^T B_e of one launch (k_gram_batch)
  const double *A[16], *B[16];
  double *partial[16];
  int lda[16], ka[16], ldb[16], kb[16];
};
// ---------------------------------------------------------------------------
// Tall-skinny block kernels for the eigensolver (LOBPCG Rayleigh-Ritz):
//   Gram:    G = A^T B            (A: rows x ka, B: rows x kb; ka, kb <= 24)
//   combine: Out = sum_i X_i C_i  (X_i: rows x k_i, C_i: k_i x kout, small, in `coef`)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gram_block(int64_t row0, int64_t rows, const double *__restrict__ A, int lda,
                                           int ka, const double *__restrict__ B, int ldb, int kb,
                                           double *__restrict__ partial) {
  // G = A^T B over the local rows is a GEMM with a long inner dimension (the rows) and a tiny output (ka x kb <= 24 x 24):
  // v_mfma_f64_16x16x4_f64 with the rows as k.  A wavefront walks its share of the rows four at a time; lane l feeds
  // A[row + l / 16][l % 16 (+ 16 per tile)] and B likewise -- one 8-byte load each, a wavefront covers four whole rows --
  // and keeps a 16 x 16 accumulator tile (4 doubles per lane) for each of the <= 2 x 2 tiles.  The four wavefronts of a
  // block are added through LDS in wave order, blocks by k_gram_reduce in block order: deterministic.
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  __shared__ double red[4][2][2][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int TA = (ka + 15) >> 4, TB = (kb + 15) >> 4;  // <= 2 each (kMaxLD = 24)
  f64x4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int64_t nwaves = static_cast<int64_t>(gridDim.x) * 4;
  const int64_t per = ((rows + nwaves - 1) / nwaves + 3) & ~static_cast<int64_t>(3);
  const int64_t w = static_cast<int64_t>(blockIdx.x) * 4 + wave;
  const int64_t r_begin = min(rows, w * per), r_end = min(rows, r_begin + per);
  const int m = lane & 15, kq = lane >> 4;
  constexpr int kUnroll = 8;
  for (int64_t r = r_begin; r < r_end; r += 4 * kUnroll) {
    double av[kUnroll][2], bv[kUnroll][2];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t row = r + 4 * u + kq;
      const bool ok = row < r_end;
      const size_t ra = static_cast<size_t>(row0 + (ok ? row : r_begin)) * lda, rb = static_cast<size_t>(row0 + (ok ? row : r_begin)) * ldb;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ca = m + 16 * t, cb = m + 16 * t;
        av[u][t] = (t < TA && ok && ca < ka) ? A[ra + ca] : 0.0;
        bv[u][t] = (t < TB && ok && cb < kb) ? B[rb + cb] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
          if (x < TA && y < TB) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][x], bv[u][y], acc[x][y], 0, 0, 0);
  }
  // C/D layout of the f64 form: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int g = 0; g < 4; ++g) red[wave][x][y][g * 64 + lane] = acc[x][y][g];
  __syncthreads();
  for (int x = 0; x < TA; ++x)
    for (int y = 0; y < TB; ++y) {
      const int t = threadIdx.x, g = t >> 6, l = t & 63;
      const int i = 16 * x + (l >> 4) + 4 * g, j = 16 * y + (l & 15);
      if (i < ka && j < kb) {
        const double sum = ((red[0][x][y][t] + red[1][x][y][t]) + red[2][x][y][t]) + red[3][x][y][t];
        partial[static_cast<size_t>(i * kb + j) * gridDim.x + blockIdx.x] = sum;
      }
    }
}
__global__ __launch_bounds__(256) void k_gram(int64_t row0, int64_t rows, const double *__restrict__ A, int lda,
                                              int ka, const double *__restrict__ B, int ldb, int kb,
                                              double *__restrict__ partial) {
  gram_block(row0, rows, A, lda, ka, B, ldb, kb, partial);
}
// Up to 16 Gram products in ONE launch (blockIdx.y = product; blockIdx.x / gridDim.x as in k_gram: the same partial sums,
// the same bits): the twelve blocks of a Rayleigh-Ritz step were 24 launches of a few microseconds each.
__global__ __launch_bounds__(256) void k_gram_batch(int64_t row0, int64_t rows, const GramBatch G) {
  const int e = blockIdx.y;
  gram_block(row0, rows, G.A[e], G.lda[e], G.ka[e], G.B[e], G.ldb[e], G.kb[e], G.partial[e]);
}

// out[el] = sum over blocks of partial[el][block]: one block per element, fixed order
__global__ __launch_bounds__(256) void k_gram_reduce(const double *__restrict__ partial, int nblocks, int nel,
                                                     double *__restrict__ out) {
  __shared__ double sm[4];
  const int el = blockIdx.x;
  if (el >= nel) return;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[static_cast<size_t>(el) * nblocks + b];
  const double t = block_sum_256(s, sm);
  if (threadIdx.x == 0) out[el] = t;
}

struct CombineCoef {  // coefficient matrices passed by value (k_combine_karg)
  static constexpr int kMax = 400;
  double v[kMax];
};
struct CombineArgs {
  const double *x[4];
  int kx[4], ldx[4], coff[4];  // coff: offset of C_i in coef (row-major k_i x kout)
  int nblocks, kout, ldo;
};

template <bool KARG>
__device__ __forceinline__ void combine_block(int64_t row0, int64_t rows, const CombineArgs &A,
                                              const double *__restrict__ coef, const CombineCoef *K, int ncoef,
                                              double *__restrict__ out) {
  // Out = sum_b X_b C_b, 16 rows per wavefront and step, on v_mfma_f64_16x16x4_f64: the rows are the M dimension
  // (lane l feeds X[row + l % 16][4 s + l / 16]), the coefficient matrices the B operand (from LDS, zero padded), the
  // <= 2 column tiles of the output sit in 4 doubles per lane each: D[row = (l >> 4) + 4 reg][col = l & 15], so one
  // store instruction writes four whole consecutive rows.
  typedef double f64x4 __attribute__((ext_vector_type(4)));
  extern __shared__ double sc[];
  for (int t = threadIdx.x; t < ncoef; t += 256) sc[t] = KARG ? K->v[t] : coef[t];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int m = lane & 15, kq = lane >> 4;
  const int TO = (A.kout + 15) >> 4;  // column tiles of the output (<= 2)
  const int64_t groups = (rows + 15) >> 4, nw = static_cast<int64_t>(gridDim.x) * 4;
  for (int64_t gidx = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6); gidx < groups; gidx += nw) {
    const int64_t r = gidx << 4;
    const bool rok = r + m < rows;
    f64x4 acc[2] = {f64x4{0.0, 0.0, 0.0, 0.0}, f64x4{0.0, 0.0, 0.0, 0.0}};
    for (int b = 0; b < A.nblocks; ++b) {
      const double *__restrict__ xr = A.x[b] + static_cast<size_t>(row0 + (rok ? r + m : r)) * A.ldx[b];
      const double *__restrict__ C = sc + A.coff[b];
      const int kx = A.kx[b];
      double xv[6];
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        const int col = 4 * st + kq;
        xv[st] = (4 * st < kx && rok && col < kx) ? xr[col] : 0.0;
      }
#pragma unroll
      for (int st = 0; st < 6; ++st)
        if (4 * st < kx) {  // wave-uniform
          const int col = 4 * st + kq;
#pragma unroll
          for (int y = 0; y < 2; ++y)
            if (y < TO) {
              const int j = 16 * y + m;
              const double cv = (col < kx && j < A.kout) ? C[col * A.kout + j] : 0.0;
              acc[y] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[st], cv, acc[y], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int y = 0; y < 2; ++y)
      if (y < TO) {
        const int j = 16 * y + m;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t row = r + kq + 4 * g;
          if (row < rows && j < A.ldo) out[static_cast<size_t>(row0 + row) * A.ldo + j] = j < A.kout ? acc[y][g] : 0.0;
        }
      }
    if (A.ldo > 16 * TO) {  // padding columns beyond the last tile (row stride 20 / 24 with kout <= 16)
      for (int j = 16 * TO + m; j < A.ldo; j += 16)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t row = r + kq + 4 * g;
          if (row < rows) out[static_cast<size_t>(row0 + row) * A.ldo + j] = 0.0;
        }
    }
  }
}
__global__ __launch_bounds__(256) void k_combine(int64_t row0, int64_t rows, CombineArgs A,
                                                 const double *__restrict__ coef, int ncoef,
                                                 double *__restrict__ out) {
  combine_block<false>(row0, rows, A, coef, nullptr, ncoef, out);
}
// The coefficient matrices in the kernel's own arguments (up to CombineCoef::kMax doubles: every Rayleigh-Ritz step of the
// eigensolver): no copy to the device and no wait for it in front of the launch.
__global__ __launch_bounds__(256) void k_combine_karg(int64_t row0, int64_t rows, CombineArgs A, const CombineCoef K, int ncoef,
                                                      double *__restrict__ out) {
  combine_block<true>(row0, rows, A, nullptr, &K, ncoef, out);
}

#endif  // CORA_TU & 2

#if CORA_TU & 4
// ---------------------------------------------------------------------------
// Staged sparse Cholesky solves (trisolve.h): every step is one dependency-free
// sparse product  dst[out_row] = src0[out_row] + sum_k val_k * src[col_k]  over the
// rows of a stage.  Block ranges of one launch: 8-lane rows, wavefront rows, chunks
// of the long (landmark) rows, whose partial sums the last chunk to finish adds up (ticket).
// Reference: CHOLMOD solve behind src/CORA_preconditioners.cpp:46-83.
// ---------------------------------------------------------------------------
// acc += sum_k val[k] * src[col[k]] over k = k0, k0 + stride, ... < k1.  Batches of eight PREDICATED entries (no
// remainder loop): the index / value loads of a batch are in flight together, then its eight row gathers -- two
// dependent round trips per batch, where a remainder loop pays two per entry.  Rows of the staged solves are short
// (<= 8 entries per lane in the 8-lane and chunk classes), so this is what bounds the kernel.
template <int LD>
__device__ __forceinline__ void rowop_entries(const int32_t *__restrict__ col, const double *__restrict__ val,
                                              const double *__restrict__ src, int k0, int k1, int stride,
                                              double (&acc)[LD]) {
  for (int kb = k0; kb < k1; kb += 8 * stride) {
    int32_t c[8];
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = kb + u * stride;
      const bool ok = k < k1;
      const int kk = ok ? k : kb;
      c[u] = col[kk];
      const double vv = val[kk];
      v[u] = ok ? vv : 0.0;
    }
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      double xx[4][LD];
#pragma unroll
      for (int u = 0; u < 4; ++u) load_row<LD>(src + static_cast<size_t>(c[h + u]) * LD, xx[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < LD; ++j) acc[j] = fma(v[h + u], xx[u][j], acc[j]);
    }
  }
}

// The two reductions of a sweep-fused STPCG iteration (RvTail, kernels.h) in ONE block, fixed order: <r, r> from the
// forward sweep's slots, then <r, v> = |L^-1 r|^2 from its |y|^2 slots and the squared norms of the last stage's rows.
__device__ __forceinline__ double lane_sum_slots_256(const double *__restrict__ x, int n) {  // a lane's share (no barrier)
  double s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.0;
  for (int b0 = threadIdx.x; b0 < n; b0 += 256 * 8) {
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = b0 + 256 * u;
      t[u] = x[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += (b0 + 256 * u < n) ? t[u] : 0.0;
  }
  return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
__device__ __forceinline__ void rv_tail_block(const RvTail &T) {
  __shared__ double sm[12];
  StpcgState L = {};
  if (threadIdx.x == 0 && !T.sums_out) L = *T.st;  // in flight while the slots are added up
  // the three sets of slots are loaded together (one round trip, not three), then reduced one after the other
  const double l_rr = lane_sum_slots_256(T.rr_partial, T.n_rr);
  const double l_yy = lane_sum_slots_256(T.yy_partial, T.n_yy);
  const double l_tt = lane_sum_slots_256(T.rowsq, T.n_rowsq);
  __shared__ double ksm[4];
  const double kappa = T.n_kappa > 0 ? kappa_sum_256(T.kappa_partial, T.n_kappa, ksm) : 0.0;  // (wave-uniform branch)
  const double rr = block_sum_256(l_rr, sm);
  const double yy = block_sum_256(l_yy, sm + 4);
  const double tt = block_sum_256(l_tt, sm + 8);
  if (threadIdx.x == 0) {
    if (T.sums_out) {  // partitioned: this rank's share of the two inner products
      T.sums_out[0] = rr;
      T.sums_out[1] = yy + tt;
      return;
    }
    if (T.n_kappa > 0) stpcg_after_kappa(L, kappa);  // the iteration's first scalar step, which no launch of its own ran
    if (T.n_rr > 0) stpcg_after_rr(L, rr);  // (n_rr == 0: <r, r> was finished by the residual pass)
    stpcg_after_rv(L, yy + tt);
    *T.st = L;
    *T.st_host = L;  // pinned mirror for the host's (infrequent) look
    if (T.seq_out) {
      unsigned long long seq = T.seq;
      if (T.seq_counter) *T.seq_counter = seq = *T.seq_counter + 1;
      __threadfence_system();
      __hip_atomic_store(T.seq_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <int LD>
__global__ __launch_bounds__(256) void k_rowop(RowOpDev op, const double *__restrict__ src0,
                                               const double *__restrict__ src, double *__restrict__ dst, const RvTail tail) {
  const int nb8 = (op.n8 + 31) >> 5, nb64 = (op.n64 + 3) >> 2;
  if (tail.st && blockIdx.x == gridDim.x - 1) {  // the extra block of a launch with a tail
    rv_tail_block(tail);
    return;
  }
  // Blocks are dispatched in index order and the chunks of the long rows are the longest chain of the launch (entries,
  // gathers, partial, ticket, the row's partials again): they take the first indices.  b below is the index in the order
  // 8-lane rows | wavefront rows | chunks that the rest of the kernel (and the slots of the row squares) uses.
  int b = static_cast<int>(blockIdx.x);
  {
    const int nbch = (op.nchunks + 3) >> 2;
    b = b < nbch ? nb8 + nb64 + b : b - nbch;
  }
  // optional: the squared norms of the product's rows, ONE SLOT PER WAVEFRONT (the eight rows of a wavefront of the 8-lane
  // class added in a fixed order; a row of the wavefront class; a long row) -- slot = 4 block + wavefront | 4 nb8 + row of
  // its class | 4 nb8 + n64 + long row.  (A slot per row -- 14 k of them on plaza2, 9 k at 10^5 poses -- made the block that
  // adds them up the longest chain of the NEXT launch: 7 dependent rounds of loads where the rows of that launch need
  // three.  A slot per block needed a barrier here: first product 13.8 -> 15.0 us at 10^5 poses.)
  double *__restrict__ rowsq = tail.st ? nullptr : tail.rowsq_out;
  double acc[LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) acc[j] = 0.0;
  if (b < nb8) {  // 32 rows per block, 8 lanes each
    const int r = (b << 5) + (static_cast<int>(threadIdx.x) >> 3), g = threadIdx.x & 7;
    const bool ok = r < op.n8;
    int orow = 0;
    if (ok) {
      orow = op.out_row[r];
      if (src0 && g == 0) load_row<LD>(src0 + static_cast<size_t>(orow) * LD, acc);
      rowop_entries<LD>(op.col, op.val, src, op.begin[r] + g, op.end[r], 8, acc);
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1)
#pragma unroll
      for (int j = 0; j < LD; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    double sq = 0.0;
    if (ok && g == 0) {
      store_row<LD>(dst + static_cast<size_t>(orow) * LD, acc);
      if (rowsq) sq = dot_row<LD>(acc, acc);
    }
#ifdef CORA_ROWSQ_PER_ROW
    if (rowsq && ok && g == 0) rowsq[r] = sq;
#else
    if (rowsq) {  // (block-uniform)
      const double t = wave_sum(sq);
      if ((threadIdx.x & 63) == 0) rowsq[(b << 2) + (static_cast<int>(threadIdx.x) >> 6)] = t;
    }
#endif
  } else if (b < nb8 + nb64) {  // one wavefront per row
    const int r = op.n8 + ((b - nb8) << 2) + (static_cast<int>(threadIdx.x) >> 6), g = threadIdx.x & 63;
    if (r >= op.n8 + op.n64) return;  // (wave-uniform)
    const int orow = op.out_row[r];
    if (src0 && g == 0) load_row<LD>(src0 + static_cast<size_t>(orow) * LD, acc);
    rowop_entries<LD>(op.col, op.val, src, op.begin[r] + g, op.end[r], 64, acc);
#pragma unroll
    for (int j = 0; j < LD; ++j) acc[j] = wave_sum(acc[j]);
    if (g == 0) {
      store_row<LD>(dst + static_cast<size_t>(orow) * LD, acc);
#ifdef CORA_ROWSQ_PER_ROW
      if (rowsq) rowsq[r] = dot_row<LD>(acc, acc);
#else
      if (rowsq) rowsq[(nb8 << 2) + (r - op.n8)] = dot_row<LD>(acc, acc);
#endif
    }
  } else {  // one wavefront per chunk of a long row
    const int ch = ((b - nb8 - nb64) << 2) + (static_cast<int>(threadIdx.x) >> 6), g = threadIdx.x & 63;
    if (ch >= op.nchunks) return;
    // (what the hand-off below needs is requested with the chunk's bounds, not after its entries: three dependent round
    // trips fewer at the end of the longest chain of the launch)
    const int r = op.chunk_row[ch];
    const int c0 = op.long_chunk_ptr[r], c1 = op.long_chunk_ptr[r + 1];
    const size_t long_orow = static_cast<size_t>(op.long_out[r]);
    rowop_entries<LD>(op.col, op.val, src, op.chunk_begin[ch] + g, op.chunk_end[ch], 64, acc);
    double tot = 0.0;  // lane j < LD ends up with column j
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      const double v = __shfl(wave_sum(acc[j]), 0, 64);
      if (g == j) tot = v;
    }
    // publish the partial write-through, take a ticket of the row; the last chunk to arrive adds the row's
    // partials in chunk order (deterministic) -- the hand-off of k_spmm's long rows
    if (g < LD)
      __hip_atomic_store(op.partial + static_cast<size_t>(ch) * kMaxLD + g, tot, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (g == 0) {
      const unsigned old = __hip_atomic_fetch_add(op.tickets + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = (old == static_cast<unsigned>(c1 - c0 - 1));
      if (last) __hip_atomic_store(op.tickets + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    last = __shfl(last, 0, 64);
    if (last) {  // wave-uniform: lane g adds chunks g, g + 64, ... (loads all in flight), then a fixed shuffle tree
      const double *P = op.partial + static_cast<size_t>(c0) * kMaxLD;
      const int nc = c1 - c0;
      double part[LD];
#pragma unroll
      for (int j = 0; j < LD; ++j) part[j] = 0.0;
      for (int c = g; c < nc; c += 64)
#pragma unroll
        for (int j = 0; j < LD; ++j)
          part[j] += __hip_atomic_load(P + static_cast<size_t>(c) * kMaxLD + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int j = 0; j < LD; ++j) part[j] = wave_sum(part[j]);
      if (g == 0) {
        const size_t orow = long_orow;
        if (src0) {
          double b0[LD];
          load_row<LD>(src0 + orow * LD, b0);
#pragma unroll
          for (int j = 0; j < LD; ++j) part[j] += b0[j];
        }
        store_row<LD>(dst + orow * LD, part);
#ifdef CORA_ROWSQ_PER_ROW
        if (rowsq) rowsq[op.n8 + op.n64 + r] = dot_row<LD>(part, part);
#else
        if (rowsq) rowsq[(nb8 << 2) + op.n64 + r] = dot_row<LD>(part, part);
#endif
      }
    }
  }
}

// LDS hand-off between the lanes of one wavefront (no workgroup barrier: the waves of a block work on
// different blocks of the factor and run different trip counts)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v where the lane's bit of the wave-uniform mask m is set, 0 elsewhere: the mask is used as the select
// condition directly (two v_cndmask), no per-lane bit test
__device__ __forceinline__ double select_by_lane_mask(double v, uint64_t m) {
  unsigned lo = static_cast<unsigned>(__double2loint(v)), hi = static_cast<unsigned>(__double2hiint(v));
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(lo) : "v"(lo), "s"(m));
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(hi) : "v"(hi), "s"(m));
  return __hiloint2double(static_cast<int>(hi), static_cast<int>(lo));
}

// Stage 0 in block form (trisolve.h): one wavefront per block, lane = row of the block.
//   forward : dst[rows] = W src[rows]
//   backward: t = src[rows] - L[later, rows]^T src[later rows];  dst[rows] = W^T t   (src may be dst:
//             a block reads its own rows before it writes them and nobody else reads them)
// The kernel is bound by VALU issue, not by bandwidth (8 waves / SIMD, ~60 columns each), so the loop over
// the columns q of W keeps everything wave-uniform on the scalar side: mask and offset of column q come
// from one scalar load, the lane's entry is base + popcount(mask below the lane).
template <int LD, bool BWD>
__global__ __launch_bounds__(256) void k_blockop(BlockOpDev B, const double *src, double *dst) {
  __shared__ double tl[4][64][LD];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + wv);
  if (b >= B.nblocks) return;  // wave-uniform; the waves of a workgroup never synchronise with each other
  const BlockDesc bd = B.desc[b];
  const int nb = bd.nrows, rb = bd.row_begin;
  const BlockLane *__restrict__ meta = (BWD ? B.by_row : B.by_col) + bd.meta_begin;
  const bool mine = lane < nb;
  const size_t row = mine ? static_cast<size_t>(meta[lane].row) : 0;
  {
    double t[LD];
#pragma unroll
    for (int j = 0; j < LD; ++j) t[j] = 0.0;
    if (mine) load_row<LD>(src + row * LD, t);
    if constexpr (BWD) {
    // coupling to the later stages: the lanes stride over ALL entries of the block (independent gathers),
    // park the products in LDS, and every row then adds up its own segment in entry order
    const int e0 = B.ext_ptr[rb], e1 = B.ext_ptr[rb + nb];
    const int my0 = mine ? B.ext_ptr[rb + lane] : 0, my1 = mine ? B.ext_ptr[rb + lane + 1] : 0;
    for (int base = e0; base < e1; base += 64) {
      const int k = base + lane;
      double p[LD];
#pragma unroll
      for (int j = 0; j < LD; ++j) p[j] = 0.0;
      if (k < e1) {
        const double v = B.ext_val[k];
        load_row<LD>(src + static_cast<size_t>(B.ext_col[k]) * LD, p);
#pragma unroll
        for (int j = 0; j < LD; ++j) p[j] *= v;
      }
#pragma unroll
      for (int j = 0; j < LD; ++j) tl[wv][lane][j] = p[j];
      wave_lds_sync();
      const int lo = my0 > base ? my0 : base, hi = my1 < base + 64 ? my1 : base + 64;
      for (int q = lo; q < hi; ++q)
#pragma unroll
        for (int j = 0; j < LD; ++j) t[j] += tl[wv][q - base][j];
      wave_lds_sync();
    }
    }
#pragma unroll
    for (int j = 0; j < LD; ++j) tl[wv][lane][j] = t[j];
    wave_lds_sync();
  }
  double acc[LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) acc[j] = 0.0;
  const char *__restrict__ W = reinterpret_cast<const char *>((BWD ? B.w_by_row : B.w_by_col) + bd.w_off);
  // Eight columns per round.  Wave-uniform part: two wide scalar loads bring the round's eight 16-byte records
  // {mask, off, row} (every block's records are padded to a multiple of eight with empty masks).  Lanes outside a
  // column's mask read some nearby entry (the value arrays are padded) and drop it.  The next round's records and
  // entries are requested before this round's products are formed, so a wave always has a round in flight.
  typedef int i32x16 __attribute__((ext_vector_type(16), aligned(16)));
  auto issue = [&](const i32x16 &ra, const i32x16 &rc, double(&w)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned mlo = static_cast<unsigned>(u < 4 ? ra[4 * u] : rc[4 * (u - 4)]);
      const unsigned mhi = static_cast<unsigned>(u < 4 ? ra[4 * u + 1] : rc[4 * (u - 4) + 1]);
      const unsigned off = static_cast<unsigned>(u < 4 ? ra[4 * u + 2] : rc[4 * (u - 4) + 2]);
      const unsigned rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
      // scalar base + 32-bit byte offset: one VALU op for the address
      w[u] = *reinterpret_cast<const double *>(W + ((rank << 3) + (off << 3)));
    }
  };
  // (the right-hand side rows t_q through the scalar cache instead of the LDS tile: measured 37 us against 25)
  auto products = [&](const i32x16 &ra, const i32x16 &rc, const double(&w)[8], int q0) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned mlo = static_cast<unsigned>(u < 4 ? ra[4 * u] : rc[4 * (u - 4)]);
      const unsigned mhi = static_cast<unsigned>(u < 4 ? ra[4 * u + 1] : rc[4 * (u - 4) + 1]);
      const double wu = select_by_lane_mask(w[u], static_cast<uint64_t>(mhi) << 32 | mlo);
      const int q = (q0 + u) & 63;
#pragma unroll
      for (int j = 0; j < LD; ++j) acc[j] = fma(wu, tl[wv][q][j], acc[j]);
    }
  };
  // two rounds per trip (A / B register sets) so that the prefetched round needs no copies
  i32x16 ra = *reinterpret_cast<const i32x16 *>(meta), rc = *reinterpret_cast<const i32x16 *>(meta + 4);
  double wa[8], wb[8];
  issue(ra, rc, wa);
  for (int q0 = 0; q0 < nb; q0 += 16) {
    const i32x16 sa = *reinterpret_cast<const i32x16 *>(meta + q0 + 8), sc = *reinterpret_cast<const i32x16 *>(meta + q0 + 12);
    issue(sa, sc, wb);  // past the block's end on its last round: the arrays are padded, the result is unused
    products(ra, rc, wa, q0);
    if (q0 + 8 >= nb) break;
    ra = *reinterpret_cast<const i32x16 *>(meta + q0 + 16);
    rc = *reinterpret_cast<const i32x16 *>(meta + q0 + 20);
    issue(ra, rc, wa);
    products(sa, sc, wb, q0 + 8);
  }
  if (mine) store_row<LD>(dst + row * LD, acc);
}


// ---------------------------------------------------------------------------
// Stage 0 as workgroup blocks solved by substitution (trisolve.h, SubBlockOpHost).  One workgroup of kSubThreads
// per block; only the block's right-hand sides sit in LDS (the tile T, rows x LD).  The block's part of L is
// STREAMED: the coefficients of a level are stored [entry slot u][lane] (coalesced), its local row indices
// [lane][4 or 8]; every lane fetches its <= kSubNpl entries of level l + 1 into registers while level l is
// computed (two register sets, ping-pong), so the HBM latency of the stream
// hides behind the LDS work of the level before, and four workgroups per CU -- the whole of a 10^5-pose problem
// resident at once -- keep the memory side busy without phases.  (The first version copied the block's entries into
// LDS before the first level: 75 KB per block, two blocks per CU, every block of a round streaming at the same
// time and computing at the same time: 20 us of row I/O + 8 us of stream + 17 us of levels, added up.)
//
// A level: lane (row, part) forms  sum_e val_e T[idx_e]  over its entries for ALL columns (columns in chunks of
// <= 6 accumulators), the g lanes of a row are summed with DPP, THEN (barrier) the rows of the level are written,
// then (barrier) the next level starts: the rows of a supernode read each other's right-hand sides.  Everything
// wave-uniform (level headers) stays on the scalar unit; wavefronts without a row only see the two barriers.
// ---------------------------------------------------------------------------
constexpr int kSubThreads = 256;  // >= rows x lanes per row of a level (plan: kLevelLanes)
constexpr int kSubNpl = 8;        // >= entries per lane of a level (plan: kLaneEntries)

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over aligned groups of g lanes (g a power of two, wave-uniform); every lane of the group gets the total.
// Neighbours first: quad permutes, then the half-row / row mirrors, then shuffles across rows of 16.
__device__ __forceinline__ double group_sum(double x, int g) {
  if (g >= 2) x += dpp_f64<0xB1>(x);    // quad_perm [1,0,3,2]
  if (g >= 4) x += dpp_f64<0x4E>(x);    // quad_perm [2,3,0,1]
  if (g >= 8) x += dpp_f64<0x141>(x);   // row_half_mirror
  if (g >= 16) x += dpp_f64<0x140>(x);  // row_mirror
  if (g >= 32) x += __shfl_xor(x, 16, 64);
  if (g >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

#ifndef CORA_SUB_NT
#define CORA_SUB_NT 1
#endif
// The factor is read once per sweep: its loads are non-temporal (CORA_SUB_NT = 1), so that 130 MB of L per STPCG
// iteration do not push Q and the vectors out of the 256 MB Infinity Cache.  Measured at 10^5 poses, p = 5: the
// product inside the loop 31.8 -> 24.9 us (its back-to-back rate), the iteration 166 -> 153 us.
template <typename T>
__device__ __forceinline__ T factor_load(const T *p) {
#if CORA_SUB_NT == 1
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// the same for N values at once: one jump on log2(g), then every step exchanges all N values (independent DPP chains
// that overlap) instead of N x log2(g) conditional steps
template <int N>
__device__ __forceinline__ void group_sum_all(double (&x)[N], int gs) {
  auto steps = [&](auto gs_c) {
    constexpr int GS = decltype(gs_c)::value;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (GS >= 1) x[j] += dpp_f64<0xB1>(x[j]);
      if (GS >= 2) x[j] += dpp_f64<0x4E>(x[j]);
      if (GS >= 3) x[j] += dpp_f64<0x141>(x[j]);
      if (GS >= 4) x[j] += dpp_f64<0x140>(x[j]);
      if (GS >= 5) x[j] += __shfl_xor(x[j], 16, 64);
      if (GS >= 6) x[j] += __shfl_xor(x[j], 32, 64);
    }
  };
  switch (gs) {
    case 0: break;
    case 1: steps(std::integral_constant<int, 1>()); break;
    case 2: steps(std::integral_constant<int, 2>()); break;
    case 3: steps(std::integral_constant<int, 3>()); break;
    case 4: steps(std::integral_constant<int, 4>()); break;
    case 5: steps(std::integral_constant<int, 5>()); break;
    default: steps(std::integral_constant<int, 6>()); break;
  }
}

#ifndef CORA_SUB_F32
#define CORA_SUB_F32 0  // lab: the substitution blocks' coefficients stored as fp32 (capi.hip uploads them so with CORA_SUB_F32=1)
#endif
#if CORA_SUB_F32
typedef float SubCoef;
#else
typedef double SubCoef;
#endif
struct SubRegs {  // a lane's entries of one level: coefficients and (two per dword) local row indices
  SubCoef v[kSubNpl];
  uint32_t i[kSubNpl / 2];
  int32_t row;  // forward sweeps that store a level's rows as they are solved (kDirect): the internal row of the lane's row
};
// Pins a register set: the compiler waits HERE for whatever load still writes it (before the next level's loads are
// issued), and treats the values as opaque afterwards.
__device__ __forceinline__ void sub_touch(SubRegs &R) {
#pragma unroll
  for (int u = 0; u < kSubNpl; ++u) asm volatile("" : "+v"(R.v[u]));
#pragma unroll
  for (int u = 0; u < kSubNpl / 2; ++u) asm volatile("" : "+v"(R.i[u]));
  asm volatile("" : "+v"(R.row));
}

// v = Proj_Y(x) for the row unit that starts at `row` (a pose's d rotation rows: only its first row does the work;
// a range row; a translation row); sink(row, v) receives every row of the result.  row_of_x(a) -> the a-th row of x.
// (Rows of the last stage in the fused backward sweep.)
template <int LD, int D, typename RowOfX, typename Sink>
__device__ __forceinline__ void project_unit(const SubFuse &F, size_t row, RowOfX row_of_x, Sink sink) {
  if (row < static_cast<size_t>(F.rng_base)) {
    if ((row - static_cast<size_t>(F.rot_base)) % D != 0) return;
    double y[D][LD], v[D][LD];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      load_row<LD>(F.Y + (row + a) * LD, y[a]);
      row_of_x(a, v[a]);
    }
    stiefel_project_thread<LD, D>(y, v);
#pragma unroll
    for (int a = 0; a < D; ++a) sink(row + a, v[a]);
    return;
  }
  double y[LD], v[LD];
  row_of_x(0, v);
  if (row < static_cast<size_t>(F.trn_base)) {
    load_row<LD>(F.Y + row * LD, y);
    const double ip = dot_row<LD>(y, v);
#pragma unroll
    for (int c = 0; c < LD; ++c) v[c] = fma(-ip, y[c], v[c]);
  }
  sink(row, v);
}

#ifndef CORA_SUB_MIN_BLOCKS
#define CORA_SUB_MIN_BLOCKS 4
#endif
#ifndef CORA_SUB_FWD_DIRECT
#define CORA_SUB_FWD_DIRECT 0  // 1: forward sweeps send a level's rows to memory when the level is solved (40-byte pieces from the
                               // lanes that hold them) instead of an epilogue of their own.  Built and measured in round 5, not kept:
                               // forward sweep at 10^5 poses 32.8 -> 35.4 us (with the folded kappa 35.8 -> 37.9) -- the level loop
                               // streams the factor at 4.8 TB/s while it runs at the LDS's bandwidth and has no room for 435 more
                               // stores and a row index per lane and level; the burst after the last level is the cheaper way
#endif
#ifndef CORA_SUB_LATE_FETCH
#define CORA_SUB_LATE_FETCH 1  // fused forward sweep: the first level's entries are requested behind the prologue's loads
#endif
#ifdef CORA_SUB_TIMES
// measurement build: wall-clock stamps of a substitution block's phases, forward and backward sweep apart
// (tools/sub_timeline.py): 0 start | 1 right-hand sides in the tile | 2 levels done | 3 end | 4 clocks spent waiting for
// a level's entries (wave 0) | 5 levels | 6.. shader-clock cycles of wave 0 inside the level loop, summed over the levels:
// 6 wait for the entries | 7 request of the next level's + header | 8 tile reads and products | 9 sums over a row's lanes |
// 10 first barrier | 11 rows into the tile | 12 second barrier | 13 unused
constexpr unsigned kSubTimesMax = 8192;
constexpr int kSubPhases = 14;
__device__ unsigned long long g_sub_phase[2 * kSubPhases * kSubTimesMax];
#define CORA_SUB_STAMP(i, v) do { if (threadIdx.x == 0 && blockIdx.x < kSubTimesMax) g_sub_phase[(BWD ? kSubPhases * kSubTimesMax : 0) + kSubPhases * blockIdx.x + (i)] = (v); } while (0)
#else
#define CORA_SUB_STAMP(i, v) do { } while (0)
#endif
// FD: 0 plain solve; 1 (forward) residual update fused into the prologue; 2 / 3 (backward) tangent projection for
// d = FD fused into the epilogue -- see SubFuse (kernels.h).
//
// Rows move between memory and the tile ELEMENT BY ELEMENT in memory order (SubSweep::io): a block is a few runs of
// consecutive rows, so a wavefront's load is 512 contiguous bytes whatever the row stride, and all loads of a phase
// are issued before the first one is consumed -- a phase costs two dependent latencies (index, value), not two per
// row.  (A lane per row: 40-byte pieces, 5 x the line requests; with the fused passes written that way the forward
// sweep took 50 us instead of 33.)
#ifndef CORA_SUB_PAD_TILE
#define CORA_SUB_PAD_TILE 1  // odd row strides: rows of the LDS tile padded to an even number of doubles (16-byte aligned rows)
#endif
// Row stride of a substitution block's tile in LDS, in doubles.  With an odd row stride (p = 5: the headline) a row of the
// tile was 8-byte aligned, so every tile read of the level loop was LD ds_read_b64 per entry; with the rows padded to an even
// stride they are 16-byte aligned and an entry costs LD / 2 ds_read_b128 + one b64 (p = 5: 3 LDS instructions instead of 5,
// the level loop's largest phase -- "tile reads and products", profiles/r05_kernel_evolution.md step 15).  Memory keeps
// its stride: only the tile's addressing changes.
template <int LD>
struct SubTile {
  static constexpr int kStride = (CORA_SUB_PAD_TILE && (LD % 2 == 1) && LD >= 3) ? LD + 1 : LD;
};
template <int LD, bool BWD, int FD>
__global__ __launch_bounds__(kSubThreads, CORA_SUB_MIN_BLOCKS) void k_subblock(SubOpDev S, const double *src, double *work, double *dst,
                                                              const SubFuse F) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double dot_sm[8];
  double *T = reinterpret_cast<double *>(smem);
  const int tid = threadIdx.x;
  const int b = static_cast<int>(blockIdx.x);
  double dacc[4] = {0.0, 0.0, 0.0, 0.0};
  double cr = 0.0;
  // kappa folded into this launch (SubFuse::n_kappa): every block adds the product's partials itself and runs the scalar
  // step on a private copy of the state (the state is not written).  A solve block does that INSIDE its prologue, behind
  // the loads of its right-hand sides (kLateKappa below): the partials -- the same 20 KB for every block at 10^5 poses,
  // L2 hits -- travel with the block's 35 KB of r and Hp instead of ahead of them.
  auto folded_kappa = [&]() -> double {
    const double kappa = kappa_sum_256(F.kappa_partial, F.n_kappa, dot_sm);
    const double c = stpcg_coef_r_after_kappa(F.dot.st, kappa);
    __syncthreads();  // (dot_sm is used again below)
    return c;
  };
  const bool late_kappa = FD == 1 && F.n_kappa > 0 && b < S.nblocks;  // (block-uniform)
  if (FD == 1 && !late_kappa) cr = F.n_kappa > 0 ? folded_kappa() : F.dot.st->coef_r;
  // fused backward sweep: the step and the direction are updated on the way out (the scalars are final: the launch
  // before this one finished <r, v>)
  const double cs = (FD >= 2) ? F.dot.st->coef_s : 0.0, cv = (FD >= 2) ? F.dot.st->coef_v : 0.0,
               cb = (FD >= 2) ? F.dot.st->coef_beta : 1.0;
  // F.p == nullptr: the projected solution itself is the result (v -> dst: the stand-alone preconditioner apply)
  const bool store_v = FD >= 2 && F.p == nullptr;
  const bool upd = FD >= 2 && (store_v || !(cs == 0.0 && cv == 0.0 && cb == 1.0));  // false: solve already finished (enqueued ahead)
  if (b >= S.nblocks) {  // rows of the last stage: forward rhs -> work, backward work -> x
    const int t = (b - S.nblocks) * kSubThreads + tid;
    if (t < S.ntop) {
      const size_t row = static_cast<size_t>(S.top_rows[t]);
      if (FD == 0) {
        double x[LD];
        load_row<LD>((BWD ? work : src) + row * LD, x);
        store_row<LD>((BWD ? dst : work) + row * LD, x);
      } else if (FD == 1) {  // r += coef_r Hp, <r, r>
        double x[LD];
        load_row<LD>(F.r + row * LD, x);
        if (cr != 0.0) {
          double h[LD];
          load_row<LD>(F.Hp + row * LD, h);
#pragma unroll
          for (int j = 0; j < LD; ++j) x[j] = fma(cr, h[j], x[j]);
          store_row<LD>(F.r + row * LD, x);
        }
        dacc[0] = dot_row<LD>(x, x);
        store_row<LD>(work + row * LD, x);
      } else if (upd) {
        constexpr int D = FD >= 2 ? FD : 2;
        project_unit<LD, D>(F, row, [&](int a, double (&x)[LD]) { load_row<LD>(work + (row + a) * LD, x); },
                            [&](size_t rw, const double (&v)[LD]) {
                              if (store_v) {
                                store_row<LD>(dst + rw * LD, v);
                                return;
                              }
                              double pv[LD], sv[LD];
                              load_row<LD>(F.p + rw * LD, pv);
                              load_row<LD>(F.s + rw * LD, sv);
#pragma unroll
                              for (int j = 0; j < LD; ++j) {
                                sv[j] = fma(cs, pv[j], sv[j]);
                                pv[j] = fma(cv, v[j], cb * pv[j]);
                              }
                              store_row<LD>(F.s + rw * LD, sv);
                              store_row<LD>(F.p + rw * LD, pv);
                            });
      }
    }
    if (FD == 1) {  // <r, r> over these rows: a slot of its own (RvTail adds the slots up, no ticket)
      const double rr = block_sum_256(dacc[0], dot_sm);
      if (tid == 0) F.rr_partial[b] = rr;
    }
    return;
  }
  CORA_SUB_STAMP(0, wall_clock64());
  [[maybe_unused]] unsigned long long dbg_wait = 0;
  const SubSweep &Q = BWD ? S.bwd : S.fwd;
  const SubDesc bd = S.desc[b];
  const int nb = bd.nrows, rb = bd.row_begin;
  const int nlev = BWD ? bd.b_nlev : bd.f_nlev;
  const SubCoef *__restrict__ gv = reinterpret_cast<const SubCoef *>(Q.val) + (BWD ? bd.b_ent_begin : bd.f_ent_begin);
  const uint16_t *__restrict__ gi = Q.idx;
  const int4 *__restrict__ gh = reinterpret_cast<const int4 *>(Q.hdr) + (BWD ? bd.b_lev_begin : bd.f_lev_begin);
  const int wave_base = __builtin_amdgcn_readfirstlane(tid);
  // level headers {first row, g | npl << 8 | rows << 12, first coefficient, first index}: staged in LDS behind the tile
  // (the loop below orders its loads with scheduling barriers, after which the compiler no longer reads global memory
  // through the scalar cache) and handed from level to level in scalar registers
  // A barrier level is kSubWaves headers, one per WAVEFRONT of the workgroup: {first row, lanes per row | entries per
  // lane << 8 | rows << 12, first coefficient, first index} of the rows that wavefront solves in the level (its own
  // width: short rows do not pay for the level's longest); nlev levels + a closing one.
  constexpr int kSubWaves = kSubThreads / 64;
  const int wv = wave_base >> 6;
  constexpr int LT = SubTile<LD>::kStride;  // row stride of the tile (doubles)
  int4 *hl = reinterpret_cast<int4 *>(smem + ((static_cast<size_t>(S.max_rows) * LT * 8 + 15) & ~static_cast<size_t>(15)));
  for (int i = tid; i < (nlev + 1) * kSubWaves; i += kSubThreads) hl[i] = gh[i];
  auto header = [&](int l) {
    const int4 h = hl[(l < nlev ? l : nlev) * kSubWaves + wv];  // the closing level has no rows
    return make_int4(__builtin_amdgcn_readfirstlane(h.x), __builtin_amdgcn_readfirstlane(h.y),
                     __builtin_amdgcn_readfirstlane(h.z), __builtin_amdgcn_readfirstlane(h.w));
  };
  const int wl = tid & 63;  // lane of the wavefront: a level's lanes are counted per wavefront
  // Forward sweeps store a level's rows the moment the level is solved (they are final): 40-byte pieces from the lanes
  // that hold them, in flight while the next levels run (a barrier waits for LDS only), instead of a burst of its own
  // after the last level with every block of the launch in the same phase.  |y|^2 is summed by the same lanes.
  constexpr bool kDirect = CORA_SUB_FWD_DIRECT && !BWD;
  const int32_t *__restrict__ grows = Q.rows + rb;  // internal row of every tile row (level order)

  // (Measured and dropped: entries of TWO levels ahead in a third register set.  The compiler only keeps loads in flight
  // across a first use when their number is branch-free, i.e. nine loads per lane and level whatever the level's width:
  // sweeps 45 -> 57 us fused, 33 -> 41 us plain -- the issue rate costs more than the latency hidden.)
  // entries of a level -> registers: one load for the lane's indices, one per coefficient slot.  Wavefronts
  // without a row of the level and slots past its width load nothing (the load unit's instruction rate is what
  // bounds a level once the latency is hidden: 16 unconditional loads per lane and level were 1.7 us per level).
  auto fetch = [&](const int4 h, SubRegs &R) {
    const int g = h.y & 0xff, npl = (h.y >> 8) & 0xf, nlane = (h.y >> 12) * g;
    if (nlane == 0) return;  // (wave-uniform: the header is the wavefront's own)
    const int lane = wl < nlane ? wl : nlane - 1;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if (npl > 4) {
      const u32x4 q = factor_load(reinterpret_cast<const u32x4 *>(gi + h.w + lane * 8));
      R.i[0] = q.x, R.i[1] = q.y, R.i[2] = q.z, R.i[3] = q.w;
    } else {
      const u32x2 q = factor_load(reinterpret_cast<const u32x2 *>(gi + h.w + lane * 4));
      R.i[0] = q.x, R.i[1] = q.y;
    }
    const SubCoef *__restrict__ pv = gv + h.z + lane;
#pragma unroll
    for (int u = 0; u < kSubNpl; ++u)
      if (u < npl) R.v[u] = factor_load(pv + u * nlane);
    if constexpr (kDirect) R.row = grows[h.x + (lane >> (31 - __builtin_clz(g)))];  // (g: a power of two)
  };
  SubRegs RA, RB;
#pragma unroll
  for (int u = 0; u < kSubNpl; ++u) RA.v[u] = RB.v[u] = 0;
#pragma unroll
  for (int u = 0; u < kSubNpl / 2; ++u) RA.i[u] = RB.i[u] = 0;
  RA.row = RB.row = 0;
  // The first level's entries are requested before the prologue, so that they travel with the right-hand sides -- except
  // in the fused forward sweep (two value streams in flight): there the twenty registers of the set did not fit beside
  // the prologue's, the compiler WAITED for the entries in order to spill three of them (56 bytes of scratch per lane,
  // written and read back by 267 k lanes: 25 MB per launch in the counters) and only then issued the prologue's first
  // load.  The header is requested up front, the entries right behind the prologue's last load.
  constexpr bool kLateFetch = CORA_SUB_LATE_FETCH && FD == 1;
  // (a SCALAR load: the header is the wavefront's own, the plan's arrays are written by the host before any launch, and
  // held in vector registers across the prologue it was the next thing to be waited for and spilled)
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(4))) const v4i_t *const_v4i_ptr;
  const v4i_t h_first_v = *reinterpret_cast<const_v4i_ptr>(reinterpret_cast<uintptr_t>(gh + wv));  // before anything is ordered
  const int4 h_first = make_int4(h_first_v.x, h_first_v.y, h_first_v.z, h_first_v.w);
  auto fetch_first = [&] {
    fetch(make_int4(__builtin_amdgcn_readfirstlane(h_first.x), __builtin_amdgcn_readfirstlane(h_first.y),
                    __builtin_amdgcn_readfirstlane(h_first.z), __builtin_amdgcn_readfirstlane(h_first.w)), RA);
  };
  if (!kLateFetch) fetch_first();

  // element e of the block = column e % LD of its (e / LD)-th row in memory order.  Per pass a lane resolves kIoBatch
  // elements (all index loads in flight together; a block has <= 512 rows: ONE pass up to a row stride of 6) and moves
  // them in two halves (registers: the sweep must keep >= 5 workgroups per CU, or the 10^5-pose plan's 1 042 blocks
  // no longer run at once)
  const int2 *__restrict__ io = Q.io + rb;
  const int ne = nb * LD;
  // (fused forward, two value streams: one pass of 2 LD elements per lane up to a row stride of 5 -- 45 us at 10^5 poses against
  // 52 us in two passes of 6 --, two passes of LD above, where one pass spills)
  constexpr int kIoBatch = (FD == 1 && LD > 5) ? (LD <= 8 ? LD : 8) : (2 * LD <= 12 ? 2 * LD : 12);
  constexpr int kIoSub = (kIoBatch + 1) / 2;
  // offsets of a lane's kIoBatch elements in the vectors (g) and in the tile (t: below 2^16 -- 512 rows x 24 columns --, two
  // per register: the fused forward sweep at a row stride of 5 was five registers short of holding them unpacked)
  struct IoAt {
    int g[kIoBatch];
    uint32_t tp[(kIoBatch + 1) / 2];
    __device__ __forceinline__ int t(int u) const { return static_cast<int>((u & 1) ? tp[u >> 1] >> 16 : tp[u >> 1] & 0xffffu); }
    __device__ __forceinline__ void set_t(int u, int v) {
      if (u & 1) tp[u >> 1] |= static_cast<uint32_t>(v) << 16; else tp[u >> 1] = static_cast<uint32_t>(v);
    }
  };
  // (io_runs: the row of an element from the block's run table -- eight scalars of its descriptor --, so that the loads
  // of a phase do not wait for an index list; the tile position from a 16-bit list, requested at the same time)
  const uint16_t *__restrict__ tpos = Q.tpos + rb;
  const bool io_runs = S.io_runs != 0;
  auto io_index = [&](int e0, IoAt &at) {
#pragma unroll
    for (int u = 0; u < kIoBatch; ++u) {
      const int e = e0 + u * kSubThreads;
      const int ee = e < ne ? e : ne - 1;
      const int k = ee / LD, col = ee - k * LD;
      if (io_runs) {
        const int off = k < bd.run_end[0] ? bd.run_off[0] : (k < bd.run_end[1] ? bd.run_off[1] : (k < bd.run_end[2] ? bd.run_off[2] : bd.run_off[3]));
        at.g[u] = (k + off) * LD + col;
        at.set_t(u, static_cast<int>(tpos[k]) * LT + col);
      } else {
        const int2 rp = io[k];
        at.g[u] = rp.x * LD + col;
        at.set_t(u, rp.y * LT + col);
      }
    }
  };

  // ---- prologue: right-hand sides -> T (fused forward: r += coef_r Hp on the way, <r, r>); backward: behind them the
  // later stage's solution (it sits in `work`) at the rows coupled to the block, which the block's entries address as
  // rows nb + k
  {
    const int ntg = BWD ? bd.ntgt : 0;
    int tgt_row = -1;
    if (BWD && tid < ntg) tgt_row = S.tgt_row[bd.tgt_begin + tid];
    // (the trip count is block-uniform -- lanes past the block's elements re-read its last one and store nothing --:
    // the folded kappa below has barriers)
    for (int base = 0; base < ne; base += kIoBatch * kSubThreads) {
      const int e0 = base + tid;
      IoAt at;
      io_index(e0, at);
#pragma unroll
      for (int half = 0; half < kIoBatch; half += kIoSub) {
        double x[kIoSub], h[FD == 1 ? kIoSub : 1];
#pragma unroll
        for (int u = 0; u < kIoSub; ++u) {
          if (half + u >= kIoBatch) continue;
          x[u] = (FD == 1 ? F.r : src)[at.g[half + u]];
          if (FD == 1) h[FD == 1 ? u : 0] = F.Hp[at.g[half + u]];
        }
        if (FD == 1 && half == 0 && base == 0 && late_kappa) cr = folded_kappa();  // behind the first loads of the block
#pragma unroll
        for (int u = 0; u < kIoSub; ++u) {
          if (half + u >= kIoBatch) continue;
          const bool ok = e0 + (half + u) * kSubThreads < ne;
          if (FD == 1) {
            if (cr != 0.0) {
              x[u] = fma(cr, h[FD == 1 ? u : 0], x[u]);
              if (ok) F.r[at.g[half + u]] = x[u];
            }
            if (ok) dacc[0] = fma(x[u], x[u], dacc[0]);
          }
          if (ok) T[at.t(half + u)] = x[u];
        }
      }
    }
    if (kLateFetch) fetch_first();
    if (BWD) {
      for (int t = tid; t < ntg; t += kSubThreads) {
        if (t != tid) tgt_row = S.tgt_row[bd.tgt_begin + t];
        double x[LD];
        load_row<LD>(work + static_cast<size_t>(tgt_row) * LD, x);
#pragma unroll
        for (int j = 0; j < LD; ++j) T[(nb + t) * LT + j] = x[j];
      }
    }
  }
  __syncthreads();
  CORA_SUB_STAMP(1, wall_clock64());

  // ---- the triangular solve, level by level
  constexpr int CW = LD <= 6 ? LD : (LD % 6 == 0 ? 6 : (LD % 5 == 0 ? 5 : 4));  // accumulators per column chunk
#ifdef CORA_SUB_TIMES
  unsigned long long dbg_cyc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_c = 0;
#define DBG_CYC(i) do { const unsigned long long c__ = __builtin_readcyclecounter(); dbg_cyc[i] += c__ - dbg_c; dbg_c = c__; } while (0)
#else
#define DBG_CYC(i) do { } while (0)
#endif
  auto level = [&](int l, const int4 h, const int4 hn, SubRegs &R, SubRegs &Rnext) {
#ifdef CORA_SUB_TIMES
    const unsigned long long dbg_w0 = wall_clock64();
    dbg_c = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the level's entries have arrived ...
#ifdef CORA_SUB_TIMES
    dbg_wait += wall_clock64() - dbg_w0;
#endif
    DBG_CYC(0);
    __builtin_amdgcn_sched_barrier(0);
    fetch(hn, Rnext);  // ... the next level's are on their way while this one is computed
    __builtin_amdgcn_sched_barrier(0);
    const int4 hnn = header(l + 2);
    DBG_CYC(1);
    const int r0 = h.x, g = h.y & 0xff, npl = (h.y >> 8) & 0xf;
    const int gs = 31 - __builtin_clz(g), nlane = (h.y >> 12) << gs;
    const bool active = nlane > 0;  // wavefront-uniform: the others go straight to the barriers
    double res[LD];
    if (active) {
      // One jump on the (wave-uniform) number of entries per lane and one on the lanes per row, then straight-line code:
      // with a branch per entry / per reduction step the tile reads of an entry were issued only after the previous
      // entry's, and the columns of a DPP sum one after the other.
#pragma unroll
      for (int c0 = 0; c0 < LD; c0 += CW) {
        double s[CW];
#pragma unroll
        for (int j = 0; j < CW; ++j) s[j] = 0.0;
        auto accumulate = [&](auto npl_c) {
          constexpr int NPL = decltype(npl_c)::value;
#pragma unroll
          for (int u = 0; u < NPL; ++u) {
            const uint32_t li = (u & 1) ? R.i[u >> 1] >> 16 : R.i[u >> 1] & 0xffffu;
            const double *__restrict__ t = static_cast<const double *>(__builtin_assume_aligned(smem + __umul24(li, LT * 8), LT % 2 == 0 ? 16 : 8)) + c0;
#pragma unroll
            for (int j = 0; j < CW; ++j)
              if (c0 + j < LD) s[j] = fma(static_cast<double>(R.v[u]), t[j], s[j]);
          }
        };
        switch (npl) {
          case 1: accumulate(std::integral_constant<int, 1>()); break;
          case 2: accumulate(std::integral_constant<int, 2>()); break;
          case 3: accumulate(std::integral_constant<int, 3>()); break;
          case 4: accumulate(std::integral_constant<int, 4>()); break;
          case 5: accumulate(std::integral_constant<int, 5>()); break;
          case 6: accumulate(std::integral_constant<int, 6>()); break;
          case 7: accumulate(std::integral_constant<int, 7>()); break;
          default: accumulate(std::integral_constant<int, 8>()); break;
        }
        DBG_CYC(2);
        group_sum_all<CW>(s, gs);
        DBG_CYC(3);
#pragma unroll
        for (int j = 0; j < CW; ++j)
          if (c0 + j < LD) res[c0 + j] = s[j];
      }
    }
    __syncthreads();  // every row of the level has read the tile ...
    DBG_CYC(4);
    if (active && wl < nlane && (wl & (g - 1)) == 0) {
      double *__restrict__ o = static_cast<double *>(__builtin_assume_aligned(smem + __mul24(r0 + (wl >> gs), LT * 8), LT % 2 == 0 ? 16 : 8));
#pragma unroll
      for (int j = 0; j < LD; ++j) o[j] = res[j];
      if constexpr (kDirect) {
        store_row<LD>(dst + static_cast<size_t>(R.row) * LD, res);
        if (FD == 1) dacc[1] += dot_row<LD>(res, res);
      }
    }
    DBG_CYC(5);
    __syncthreads();  // ... before any of them is written
    DBG_CYC(6);
    return hnn;
  };
  {
    int4 h0 = header(0), h1 = header(1);
    for (int l = 0; l < nlev; l += 2) {
      const int4 h2 = level(l, h0, h1, RA, RB);
      if (l + 1 >= nlev) break;
      const int4 h3 = level(l + 1, h1, h2, RB, RA);
      h0 = h2;
      h1 = h3;
    }
  }

  CORA_SUB_STAMP(2, wall_clock64());
  CORA_SUB_STAMP(4, dbg_wait);
  CORA_SUB_STAMP(5, static_cast<unsigned long long>(nlev));
#ifdef CORA_SUB_TIMES
  for (int q = 0; q < 7; ++q) CORA_SUB_STAMP(6 + q, dbg_cyc[q]);
#endif
#ifdef CORA_SUB_TIMES
  struct SubEnd {  // (the result paths return from several places)
    unsigned bwd;
    __device__ ~SubEnd() {
      if (threadIdx.x == 0 && blockIdx.x < kSubTimesMax) g_sub_phase[(bwd ? kSubPhases * kSubTimesMax : 0) + kSubPhases * blockIdx.x + 3] = wall_clock64();
    }
  } dbg_end{BWD ? 1u : 0u};
#endif
  // ---- results
  if (FD >= 2) {
    // v = Proj_Y(x) in the tile, a lane per row unit (a pose's rotation rows sit at consecutive positions) ...
    constexpr int D = FD >= 2 ? FD : 2;
    const int2 *__restrict__ un = S.b_unit + bd.unit_begin;
    for (int u = tid; u < bd.nunits; u += kSubThreads) {
      const int2 pr = un[u];  // {tile position, internal row}
      const size_t row = static_cast<size_t>(pr.y);
      double *__restrict__ tv = T + pr.x * LT;
      if (row < static_cast<size_t>(F.rng_base)) {
        double y[D][LD], v[D][LD];
#pragma unroll
        for (int a = 0; a < D; ++a) load_row<LD>(F.Y + (row + a) * LD, y[a]);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int j = 0; j < LD; ++j) v[a][j] = tv[a * LT + j];
        stiefel_project_thread<LD, D>(y, v);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int j = 0; j < LD; ++j) tv[a * LT + j] = v[a][j];
      } else if (row < static_cast<size_t>(F.trn_base)) {
        double y[LD], v[LD];
        load_row<LD>(F.Y + row * LD, y);
#pragma unroll
        for (int j = 0; j < LD; ++j) v[j] = tv[j];
        const double ip = dot_row<LD>(y, v);
#pragma unroll
        for (int j = 0; j < LD; ++j) tv[j] = fma(-ip, y[j], v[j]);
      }
    }
    __syncthreads();
  }
  // ... and the tile -> dst in memory order.  Fused backward: the tile holds v; nothing is stored but the updated step
  // and direction, s += coef_s p and p = coef_v v + coef_beta p, element by element in the same order.  Fused forward:
  // with |y|^2 of the block's rows (SubFuse::yy_partial).
  if (FD >= 2) {
    if (!upd) return;
    if (store_v) {
      for (int e0 = tid; e0 < ne; e0 += kIoBatch * kSubThreads) {
        IoAt at;
        io_index(e0, at);
#pragma unroll
        for (int u = 0; u < kIoBatch; ++u)
          if (e0 + u * kSubThreads < ne) dst[at.g[u]] = T[at.t(u)];
      }
      return;
    }
    for (int e0 = tid; e0 < ne; e0 += kIoBatch * kSubThreads) {
      IoAt at;
      io_index(e0, at);
#pragma unroll
      for (int half = 0; half < kIoBatch; half += kIoSub) {
        double pv[kIoSub], sv[kIoSub];
#pragma unroll
        for (int u = 0; u < kIoSub; ++u)
          if (half + u < kIoBatch) {
            pv[u] = F.p[at.g[half + u]];
            sv[u] = F.s[at.g[half + u]];
          }
#pragma unroll
        for (int u = 0; u < kIoSub; ++u)
          if (half + u < kIoBatch && e0 + (half + u) * kSubThreads < ne) {
            const double v = T[at.t(half + u)];
            F.s[at.g[half + u]] = fma(cs, pv[u], sv[u]);
            F.p[at.g[half + u]] = fma(cv, v, cb * pv[u]);
          }
      }
    }
    return;
  }
  if constexpr (!kDirect) {
    for (int e0 = tid; e0 < ne; e0 += kIoBatch * kSubThreads) {
      IoAt at;
      io_index(e0, at);
#pragma unroll
      for (int u = 0; u < kIoBatch; ++u)
        if (e0 + u * kSubThreads < ne) {
          const double v = T[at.t(u)];
          dst[at.g[u]] = v;
          if (FD == 1) dacc[1] = fma(v, v, dacc[1]);
        }
    }
  }
  if (FD == 1) {  // <r, r> and |y|^2 over the block's rows (RvTail adds the slots of all blocks in fixed order)
    const double rr = block_sum_256(dacc[0], dot_sm);
    const double yy = block_sum_256(dacc[1], dot_sm + 4);
    if (tid == 0) {
      F.rr_partial[b] = rr;
      F.yy_partial[b] = yy;
    }
  }
  if (!BWD) {  // couplings to the last stage: aux row of target g = -sum_v L_gv y_v, 16 lanes per target
    const int ntgt = bd.ntgt;
    for (int base = 0; base < (ntgt << 4); base += kSubThreads) {
      const int id = base + tid, tq = id >> 4, part = id & 15;
      const bool live = tq < ntgt;
      const int tg = bd.tgt_begin + (live ? tq : 0);
      const int c0 = S.c_ptr[tg] + part, c1 = live ? S.c_ptr[tg + 1] : 0;
      double sum[LD];
#pragma unroll
      for (int j = 0; j < LD; ++j) sum[j] = 0.0;
      for (int kb = c0; kb < c1; kb += 4 * 16) {
        int ci[4];
        double cv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = kb + u * 16;
          const bool ok = k < c1;
          const int kk = ok ? k : kb;
          ci[u] = S.c_idx[kk];
          const double v = S.c_val[kk];
          cv[u] = ok ? v : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double *__restrict__ t = T + ci[u] * LT;
#pragma unroll
          for (int j = 0; j < LD; ++j) sum[j] = fma(cv[u], t[j], sum[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < LD; ++j) sum[j] = group_sum(sum[j], 16);
      if (live && part == 0) store_row<LD>(work + (static_cast<size_t>(S.aux_base) + S.tgt_slot[tg]) * LD, sum);
    }
  }
}

#if CORA_LDG & 1
// kappa = <p, Hp> from the partials of an EPI_HVP_K product (one block, fixed order), then the scalar step.
// (Measured and not kept, round 3: the same sum in the product's own last block behind a ticket -- one counter for the
// 4 016 blocks serialises 4 016 returning atomics on one address, 63 us; a two-level ticket, 64 blocks per counter, costs
// every wavefront a drain of its stores and an atomic round trip before it frees its slot: 29.8 us against 24.4 + 4.7.)
__global__ __launch_bounds__(256) void k_kappa_finish(const double *__restrict__ partial, int n, StpcgState *st) {
  __shared__ double sm[4];
  StpcgState L = {};
  if (threadIdx.x == 0) L = *st;  // in flight while the partials are added up
  const double t = kappa_sum_256(partial, n, sm);
  if (threadIdx.x == 0) {
    stpcg_after_kappa(L, t);
    *st = L;
  }
}

__global__ void k_zero_row(double *x, size_t row, int ld) {
  if (static_cast<int>(threadIdx.x) < ld) x[row * ld + threadIdx.x] = 0.0;
}
#endif  // CORA_LDG & 1

#endif  // CORA_TU & 4

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
#define CORA_LD_CASES(M) \
  M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24)

static inline int grid_for(int64_t n, int per_block = 256, int cap = 2048) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

#define CORA_LD_CASES_G0(M) M(2) M(3) M(4) M(5)
#define CORA_LD_CASES_G1(M) M(6) M(7) M(8) M(9)
#define CORA_LD_CASES_G2(M) M(10) M(11) M(12)
#define CORA_LD_CASES_G3(M) M(13) M(14) M(15) M(16)
#define CORA_LD_CASES_G4(M) M(17) M(18) M(19) M(20)
#define CORA_LD_CASES_G5(M) M(21) M(22) M(23) M(24)

#if CORA_TU & 1
template <int LD, int D>
static hipError_t launch_spmm_ld(const SpmmArgs &A_in, int epi, hipStream_t st) {
  SpmmArgs A = A_in;
  if (LD <= kPoseFirstMaxLD && A.slices_pose_first) A.slices = A.slices_pose_first;
  A.win_on = A.n_slices >= g_win_min_slices ? 1 : 0;
  A.n_real_chunks = A.n_chunks;
  A.n_chunks = (A.n_chunks + 7) & ~7;
  A.n_slice_blocks = A.n_slices;
  const int grid = A.n_chunks + 8 * ((A.n_slices + 7) / 8);
  A.kappa_long_base = grid;  // = launch_spmm_blocks(A_in): the long rows' own slots follow the per-block ones
  if (A.n_real_chunks + A.n_slices == 0) return hipSuccess;
  // lab switch: extra (unused) LDS per block limits the wavefronts resident per CU
  static const unsigned xlds = [] { const char *e = std::getenv("CORA_SPMM_EXTRA_LDS"); return e ? static_cast<unsigned>(std::atoi(e)) : 0u; }();
  switch (epi) {
    case EPI_NONE: hipLaunchKernelGGL((k_spmm<LD, D, EPI_NONE>), dim3(grid), dim3(64), xlds, st, A); break;
    case EPI_S: hipLaunchKernelGGL((k_spmm<LD, D, EPI_S>), dim3(grid), dim3(64), xlds, st, A); break;
    case EPI_HVP_K:
      if (!A.kappa_partial) return hipErrorInvalidValue;
      hipLaunchKernelGGL((k_spmm<LD, D, EPI_HVP_K>), dim3(grid), dim3(64), xlds, st, A);
      break;
    default: hipLaunchKernelGGL((k_spmm<LD, D, EPI_HVP>), dim3(grid), dim3(64), xlds, st, A); break;
  }
  return hipGetLastError();
}

// one launcher per row-stride group, each in the translation unit that instantiates the group's kernels
#define CASE(L)                                                      \
  if (ld == L)                                                       \
    return d == 2 ? launch_spmm_ld<L, 2>(A, epi, st) : launch_spmm_ld<L, 3>(A, epi, st);
#define SPMM_GROUP(G)                                                                       \
  hipError_t launch_spmm_g##G(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st) { \
    CORA_LD_CASES_G##G(CASE)                                                                \
    return hipErrorInvalidValue;                                                            \
  }
hipError_t launch_spmm_g0(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st);
hipError_t launch_spmm_g1(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st);
hipError_t launch_spmm_g2(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st);
hipError_t launch_spmm_g3(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st);
hipError_t launch_spmm_g4(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st);
hipError_t launch_spmm_g5(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st);
#if CORA_LDG & 1
// slices from which the pose slices read X through their LDS windows (kWinMinSlices; CORA_SPMM_WINDOW_MIN_SLICES or
// cora_debug_spmm_window_min_slices: the tests run the window form of every row stride on small problems)
int g_win_min_slices = [] { const char *e = std::getenv("CORA_SPMM_WINDOW_MIN_SLICES"); return e ? std::atoi(e) : kWinMinSlices; }();
SPMM_GROUP(0)
hipError_t launch_spmm(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st) {
  if (ld <= 5) return launch_spmm_g0(A, ld, d, epi, st);
  if (ld <= 9) return launch_spmm_g1(A, ld, d, epi, st);
  if (ld <= 12) return launch_spmm_g2(A, ld, d, epi, st);
  if (ld <= 16) return launch_spmm_g3(A, ld, d, epi, st);
  if (ld <= 20) return launch_spmm_g4(A, ld, d, epi, st);
  return launch_spmm_g5(A, ld, d, epi, st);
}
#endif
#if CORA_LDG & 2
SPMM_GROUP(1)
#endif
#if CORA_LDG & 4
SPMM_GROUP(2)
#endif
#if CORA_LDG & 8
SPMM_GROUP(3)
#endif
#if CORA_LDG & 16
SPMM_GROUP(4)
#endif
#if CORA_LDG & 32
SPMM_GROUP(5)
#endif
#undef SPMM_GROUP
#undef CASE
#endif  // CORA_TU & 1

#if CORA_TU & 2

hipError_t launch_point_finish(const RowArgs &R, int ld, const double *Y, const double *G,
                               double *rgrad, double *lam_st, double *lam_ob, double *partial,
                               int *nblocks, hipStream_t st) {
  const int64_t units = static_cast<int64_t>(R.nl_poses) + R.nl_ranges + R.nl_trans;
  const int grid = static_cast<int>((units + 255) / 256);
  *nblocks = grid;
  if (grid == 0) return hipSuccess;
#define CASE(L)                                                                               \
  if (ld == L) {                                                                              \
    if (R.d == 2) hipLaunchKernelGGL((k_point_finish<L, 2>), dim3(grid), dim3(256), 0, st, R, \
                                     Y, G, rgrad, lam_st, lam_ob, partial);                   \
    else hipLaunchKernelGGL((k_point_finish<L, 3>), dim3(grid), dim3(256), 0, st, R, Y, G,    \
                            rgrad, lam_st, lam_ob, partial);                                  \
    return hipGetLastError();                                                                 \
  }
  CORA_LD_CASES(CASE)
#undef CASE
  return hipErrorInvalidValue;
}

hipError_t launch_tangent_project(const RowArgs &R, int ld, const double *Y, const double *V,
                                  const double *scale, double *out, hipStream_t st) {
  const int64_t units = static_cast<int64_t>(R.nl_poses) + R.nl_ranges + R.nl_trans;
  const int grid = static_cast<int>((units + 255) / 256);
  if (grid == 0) return hipSuccess;
#define CASE(L)                                                                                  \
  if (ld == L) {                                                                                 \
    if (R.d == 2) hipLaunchKernelGGL((k_tangent_project<L, 2>), dim3(grid), dim3(256), 0, st, R, \
                                     Y, V, scale, out);                                          \
    else hipLaunchKernelGGL((k_tangent_project<L, 3>), dim3(grid), dim3(256), 0, st, R, Y, V,    \
                            scale, out);                                                         \
    return hipGetLastError();                                                                    \
  }
  CORA_LD_CASES(CASE)
#undef CASE
  return hipErrorInvalidValue;
}

hipError_t launch_project_manifold(const RowArgs &R, int ld, const double *A, const double *V,
                                   double alpha, double *out, hipStream_t st) {
  const int64_t units = static_cast<int64_t>(R.nl_poses) + R.nl_ranges + R.nl_trans;
  const int grid = static_cast<int>((units + 255) / 256);
  if (grid == 0) return hipSuccess;
#define CASE(L)                                                                                   \
  if (ld == L) {                                                                                  \
    if (R.d == 2) hipLaunchKernelGGL((k_project_manifold<L, 2>), dim3(grid), dim3(256), 0, st, R, \
                                     A, V, alpha, out);                                           \
    else hipLaunchKernelGGL((k_project_manifold<L, 3>), dim3(grid), dim3(256), 0, st, R, A, V,    \
                            alpha, out);                                                          \
    return hipGetLastError();                                                                     \
  }
  CORA_LD_CASES(CASE)
#undef CASE
  return hipErrorInvalidValue;
}

static inline bool vec2_ok(int64_t n, const void *a, const void *b) {
  return (n % 2 == 0) && (reinterpret_cast<uintptr_t>(a) % 16 == 0) && (reinterpret_cast<uintptr_t>(b) % 16 == 0);
}

hipError_t launch_axpby(int64_t n, double a, const double *x, double b, double *y, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (vec2_ok(n, x, y))
    hipLaunchKernelGGL(k_axpby, dim3(grid_for(n / 2)), dim3(256), 0, st, n / 2, a,
                       reinterpret_cast<const double2 *>(x), b, reinterpret_cast<double2 *>(y));
  else
    hipLaunchKernelGGL(k_axpby1, dim3(grid_for(n)), dim3(256), 0, st, n, a, x, b, y);
  return hipGetLastError();
}

hipError_t launch_axpy2(int64_t n, double a1, const double *x1, double *y1, double a2, const double *x2,
                        double *y2, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_axpy2, dim3(grid_for(n)), dim3(256), 0, st, n, a1, x1, y1, a2, x2, y2);
  return hipGetLastError();
}

hipError_t launch_stpcg_update(int64_t n, const StpcgState *S, const double *p, const double *Hp, double *s,
                               double *r, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_stpcg_update, dim3(grid_for(n)), dim3(256), 0, st, n, S, p, Hp, s, r);
  return hipGetLastError();
}

hipError_t launch_stpcg_direction(int64_t n, const StpcgState *S, const double *v, double *p, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_stpcg_direction, dim3(grid_for(n)), dim3(256), 0, st, n, S, v, p);
  return hipGetLastError();
}

// D: mode / st / st_host / partial / ticket / seq fields set by the caller; n doubles, even, 16-byte aligned
hipError_t launch_kappa_residual(const DotArgs &D_in, const double *kpartial, int nk, int64_t n, const double *Hp, double *r,
                                 hipStream_t st) {
  if (n <= 0 || n % 2 || reinterpret_cast<uintptr_t>(Hp) % 16 || reinterpret_cast<uintptr_t>(r) % 16) return hipErrorInvalidValue;
  DotArgs D = D_in;
  D.count = 1;
  D.n2 = n / 2;
  D.mode = DOTS_STPCG_KAPPA_RR;
  hipLaunchKernelGGL(k_kappa_residual, dim3(grid_for(n / 2, 256, 256)), dim3(256), 0, st, D, kpartial, nk,
                     reinterpret_cast<const double2 *>(Hp), reinterpret_cast<double2 *>(r));
  return hipGetLastError();
}
int kappa_residual_slots_blocks(int64_t n) { return grid_for(n / 2, 256, 256); }
// rr_slot: kappa_residual_slots_blocks(n) doubles, one per block of the launch (added by an RvTail block)
hipError_t launch_kappa_residual_slots(const StpcgState *S, const double *kpartial, int nk, int64_t n, const double *Hp, double *r,
                                       double *rr_slot, hipStream_t st) {
  if (n <= 0 || n % 2 || reinterpret_cast<uintptr_t>(Hp) % 16 || reinterpret_cast<uintptr_t>(r) % 16) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_kappa_residual_slots, dim3(kappa_residual_slots_blocks(n)), dim3(256), 0, st, S, kpartial, nk, n / 2,
                     reinterpret_cast<const double2 *>(Hp), reinterpret_cast<double2 *>(r), rr_slot);
  return hipGetLastError();
}
// D: mode / st / st_host / partial / ticket / seq fields set by the caller; n doubles, even, 16-byte aligned
hipError_t launch_stpcg_residual(const DotArgs &D_in, int64_t n, const double *Hp, double *r, hipStream_t st) {
  if (n <= 0 || n % 2 || reinterpret_cast<uintptr_t>(Hp) % 16 || reinterpret_cast<uintptr_t>(r) % 16) return hipErrorInvalidValue;
  DotArgs D = D_in;
  D.count = 1;
  D.n2 = n / 2;
  hipLaunchKernelGGL(k_stpcg_residual, dim3(grid_for(n / 2, 256, 256)), dim3(256), 0, st, D, reinterpret_cast<const double2 *>(Hp),
                     reinterpret_cast<double2 *>(r));
  return hipGetLastError();
}

hipError_t launch_stpcg_scalar_step(int what, const double *vals, StpcgState *state, StpcgState *state_host,
                                    unsigned long long *seq_out, unsigned long long seq, hipStream_t st) {
  hipLaunchKernelGGL(k_stpcg_scalar_step, dim3(1), dim3(64), 0, st, what, vals, state, state_host, seq_out, seq);
  return hipGetLastError();
}
hipError_t launch_stpcg_init(int64_t n, const double *g, const double *Pg, double *s, double *r, double *p, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_stpcg_init, dim3(grid_for(n)), dim3(256), 0, st, n, g, Pg, s, r, p);
  return hipGetLastError();
}
hipError_t launch_stpcg_step_direction(int64_t n, const StpcgState *S, const double *v, double *p, double *s,
                                       hipStream_t st) {
  if (n <= 0 || n % 2 || reinterpret_cast<uintptr_t>(v) % 16 || reinterpret_cast<uintptr_t>(p) % 16 ||
      reinterpret_cast<uintptr_t>(s) % 16)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_stpcg_step_direction, dim3(grid_for(n / 2)), dim3(256), 0, st, n / 2, S,
                     reinterpret_cast<const double2 *>(v), reinterpret_cast<double2 *>(p), reinterpret_cast<double2 *>(s));
  return hipGetLastError();
}

hipError_t launch_tangent_project_update(const RowArgs &R, const StpcgState *S, int ld, const double *Y, const double *X,
                                         double *p, double *s, hipStream_t st) {
  const int64_t units = static_cast<int64_t>(R.nl_poses) + R.nl_ranges + R.nl_trans;
  const int grid = static_cast<int>((units + 255) / 256);
  if (grid == 0) return hipErrorInvalidValue;
#define CASE(L)                                                                                            \
  if (ld == L) {                                                                                           \
    if (R.d == 2) hipLaunchKernelGGL((k_tangent_project_update<L, 2>), dim3(grid), dim3(256), 0, st, R, S, \
                                     Y, X, p, s);                                                          \
    else hipLaunchKernelGGL((k_tangent_project_update<L, 3>), dim3(grid), dim3(256), 0, st, R, S, Y, X,    \
                            p, s);                                                                         \
    return hipGetLastError();                                                                              \
  }
  CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
#undef CASE
  return hipErrorInvalidValue;
}

// out = Proj_Y(V) and <r, out>; the partial array of D needs one slot per 256 row units
hipError_t launch_tangent_project_dot(const RowArgs &R, const DotArgs &D_in, int ld, const double *Y, const double *V,
                                      const double *scale, const double *r, double *out, hipStream_t st) {
  const int64_t units = static_cast<int64_t>(R.nl_poses) + R.nl_ranges + R.nl_trans;
  const int grid = static_cast<int>((units + 255) / 256);
  if (grid == 0) return hipErrorInvalidValue;
  DotArgs D = D_in;
  D.count = 1;
#define CASE(L)                                                                                          \
  if (ld == L) {                                                                                         \
    if (R.d == 2) hipLaunchKernelGGL((k_tangent_project_dot<L, 2>), dim3(grid), dim3(256), 0, st, R, D,  \
                                     Y, V, scale, r, out);                                               \
    else hipLaunchKernelGGL((k_tangent_project_dot<L, 3>), dim3(grid), dim3(256), 0, st, R, D, Y, V,     \
                            scale, r, out);                                                              \
    return hipGetLastError();                                                                            \
  }
  CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
#undef CASE
  return hipErrorInvalidValue;
}

hipError_t launch_scale_rows(int64_t rows, int ld, const double *scale, const double *x, double *y,
                             hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scale_rows, dim3(grid_for(rows * ld)), dim3(256), 0, st, rows, ld, scale, x, y);
  return hipGetLastError();
}

// D.n2 holds the number of DOUBLES; the vectorised kernel is used when everything is 16-byte aligned
hipError_t launch_dots(const DotArgs &D_in, int *nblocks, hipStream_t st) {
  DotArgs D = D_in;
  bool vec = (D.n2 % 2 == 0);
  for (int j = 0; j < D.count; ++j)
    vec = vec && reinterpret_cast<uintptr_t>(D.a[j]) % 16 == 0 && reinterpret_cast<uintptr_t>(D.b[j]) % 16 == 0;
  if (vec) D.n2 /= 2;
  const int grid = grid_for(D.n2, 256, 256);  // one block per CU: 128 / 256 / 512 / 1024 blocks measured 24.0 / 16.8 / 19.0 / 24.7 us per inner product
  *nblocks = grid;
  if (vec && D.count == 2 && D.a[0] == D.b[0] && D.a[1] == D.a[0])
    hipLaunchKernelGGL(k_dots_rr_rv, dim3(grid), dim3(256), 0, st, D);
  else if (vec) hipLaunchKernelGGL(k_dots, dim3(grid), dim3(256), 0, st, D);
  else hipLaunchKernelGGL(k_dots1, dim3(grid), dim3(256), 0, st, D);
  return hipGetLastError();
}

hipError_t launch_reduce_partials(const double *partial, int nblocks, int count, double *out,
                                  hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, st, partial, nblocks, count, out);
  return hipGetLastError();
}

hipError_t launch_move_rows(int mode, int64_t n, int ld, const int32_t *rows, const double *src, double *dst,
                            hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_move_rows, dim3(grid_for(n * ld)), dim3(256), 0, st, mode, n, ld, rows, src, dst);
  return hipGetLastError();
}

hipError_t launch_exchange_pack(int64_t n, int ld, const int32_t *rows, int64_t ztail, const double *src, double *dst, hipStream_t st) {
  if (n * ld + ztail <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_exchange_pack, dim3(grid_for(n * ld + ztail)), dim3(256), 0, st, n, ld, rows, ztail, src, dst);
  return hipGetLastError();
}
hipError_t launch_scatter_shard_rows(int world, int rank, int64_t maxn, int ld, int64_t shard_rows, const int64_t *meta,
                                     const double *recv, double *X, hipStream_t st) {
  if (maxn * ld * world <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scatter_shard_rows, dim3(grid_for(maxn * ld * world)), dim3(256), 0, st, world, rank, maxn, ld, shard_rows, meta, recv, X);
  return hipGetLastError();
}
hipError_t launch_exchange_unpack(int world, int64_t e_max, int n_long, int ld, const int32_t *recv_idx, const double *recv, double *X,
                                  int rank, const int32_t *long_rows, const int32_t *long_owner, double *out, double *kappa,
                                  hipStream_t st) {
  const int sb = grid_for(static_cast<int64_t>(world) * e_max * ld);
  hipLaunchKernelGGL(k_exchange_unpack, dim3(sb + n_long), dim3(256), 0, st, world, e_max, n_long, ld,
                     (e_max + n_long) * static_cast<int64_t>(ld), sb, recv_idx, recv, X, rank, long_rows, long_owner, out, kappa);
  return hipGetLastError();
}

hipError_t launch_long_finish(int n_long, int ld, int rank, const int32_t *rows, const int32_t *owner, const double *slots,
                              const double *X, double *out, double *kappa, hipStream_t st) {
  if (n_long <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_long_finish, dim3(n_long), dim3(64), 0, st, n_long, ld, rank, rows, owner, slots, X, out, kappa);
  return hipGetLastError();
}

hipError_t launch_has_nan(int64_t n, const double *x, int *flag, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_has_nan, dim3(grid_for(n)), dim3(256), 0, st, n, x, flag);
  return hipGetLastError();
}

hipError_t launch_upload(int64_t N, int k, int ld, const double *src, const int32_t *api2int,
                         double *dst, hipStream_t st) {
  if (N * k <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_upload, dim3(grid_for(N * k)), dim3(256), 0, st, N, k, ld, src, api2int, dst);
  return hipGetLastError();
}

hipError_t launch_download(int64_t N, int k, int ld, const double *src, const int32_t *api2int,
                           double *dst, hipStream_t st) {
  if (N * k <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_download, dim3(grid_for(N * k)), dim3(256), 0, st, N, k, ld, src, api2int, dst);
  return hipGetLastError();
}

#endif  // CORA_TU & 2
}  // namespace cora

namespace cora {
#if CORA_TU & 4

template <int LD>
static hipError_t rowop_ld(const RowOpDev &op, const double *src0, const double *src, double *dst, hipStream_t st,
                           const RvTail *tail) {
  const int grid = ((op.n8 + 31) >> 5) + ((op.n64 + 3) >> 2) + ((op.nchunks + 3) >> 2) + ((tail && tail->st) ? 1 : 0);
  RvTail T{};
  if (tail) T = *tail;
  if (grid > 0) hipLaunchKernelGGL((k_rowop<LD>), dim3(grid), dim3(256), 0, st, op, src0, src, dst, T);
  return hipGetLastError();
}

template <int LD>
static hipError_t blockop_ld(const BlockOpDev &B, bool backward, const double *src, double *dst, hipStream_t st) {
  const int grid = (B.nblocks + 3) >> 2;
  if (grid <= 0) return hipSuccess;
  if (backward) hipLaunchKernelGGL((k_blockop<LD, true>), dim3(grid), dim3(256), 0, st, B, src, dst);
  else hipLaunchKernelGGL((k_blockop<LD, false>), dim3(grid), dim3(256), 0, st, B, src, dst);
  return hipGetLastError();
}

template <int LD>
static hipError_t subblock_ld(const SubOpDev &S, bool backward, const double *src, double *work, double *dst, hipStream_t st,
                              const SubFuse *F = nullptr) {
  const int grid = launch_subblock_blocks(S);
  if (grid <= 0) return hipSuccess;
  const size_t lds = ((static_cast<size_t>(S.max_rows) * SubTile<LD>::kStride * 8 + 15) & ~static_cast<size_t>(15)) + static_cast<size_t>(S.max_lev + 2) * 16;
  if (S.max_level_lanes > kSubThreads || S.max_npl > kSubNpl || lds > 160 * 1024) return hipErrorInvalidValue;
  // (wide rows: the tile of a 435-row block is 56 KB at 16 columns, 84 KB at 24 -- above the 64 KB a kernel gets without asking)
  auto allow_lds = [&](const void *fn) {
    return lds > 64 * 1024 ? hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) : hipSuccess;
  };
  const dim3 g(grid), t(kSubThreads);
  if (F) {
    // the fused sweeps exist where cora_stpcg_dev dispatches them (row stride x d <= 24: above, the fused backward sweep
    // spills -- 172 to 380 bytes of scratch per lane at row strides 10-12 with d = 3 -- and was never launched)
    if (!backward) {
      if constexpr (LD <= 12) hipLaunchKernelGGL((k_subblock<LD, false, 1>), g, t, lds, st, S, src, work, dst, *F);
      else return hipErrorInvalidValue;
    } else if (F->d == 2) {
      if constexpr (LD <= 12) hipLaunchKernelGGL((k_subblock<LD, true, 2>), g, t, lds, st, S, src, work, dst, *F);
      else return hipErrorInvalidValue;
    } else if (F->d == 3) {
      if constexpr (LD <= 8) hipLaunchKernelGGL((k_subblock<LD, true, 3>), g, t, lds, st, S, src, work, dst, *F);
      else return hipErrorInvalidValue;
    } else {
      return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  const SubFuse none{};
  if (backward) {
    if (const hipError_t e = allow_lds(reinterpret_cast<const void *>(&k_subblock<LD, true, 0>)); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_subblock<LD, true, 0>), g, t, lds, st, S, src, work, dst, none);
  } else {
    if (const hipError_t e = allow_lds(reinterpret_cast<const void *>(&k_subblock<LD, false, 0>)); e != hipSuccess) return e;
    hipLaunchKernelGGL((k_subblock<LD, false, 0>), g, t, lds, st, S, src, work, dst, none);
  }
  return hipGetLastError();
}

// one set of launchers per row-stride group, each in the translation unit that instantiates the group's kernels
struct TriCall {  // what a staged-solve launch needs, whichever kernel it is
  int kind;       // 0 blockop, 1 rowop, 2 subblock, 3 subblock fused
  int ld;
  bool backward;
  const BlockOpDev *B;
  const RowOpDev *R;
  const SubOpDev *S;
  const SubFuse *F;
  const double *a, *b;  // blockop: src, -;  rowop: src0, src;  subblock: rhs_or_y, -
  double *work, *dst;
  const RvTail *tail = nullptr;
};
#define CASE(L)                                                                                       \
  if (c.ld == L) {                                                                                    \
    if (c.kind == 0) return blockop_ld<L>(*c.B, c.backward, c.a, c.dst, st);                          \
    if (c.kind == 1) return rowop_ld<L>(*c.R, c.a, c.b, c.dst, st, c.tail);                           \
    if (c.kind == 2) return subblock_ld<L>(*c.S, c.backward, c.a, c.work, c.dst, st);                 \
    return subblock_ld<L>(*c.S, c.backward, c.a, c.work, c.dst, st, c.F);                             \
  }
#define TRI_GROUP(G)                                             \
  hipError_t launch_tri_g##G(const TriCall &c, hipStream_t st) { \
    CORA_LD_CASES_G##G(CASE)                                     \
    return hipErrorInvalidValue;                                 \
  }
hipError_t launch_tri_g0(const TriCall &c, hipStream_t st);
hipError_t launch_tri_g1(const TriCall &c, hipStream_t st);
hipError_t launch_tri_g2(const TriCall &c, hipStream_t st);
hipError_t launch_tri_g3(const TriCall &c, hipStream_t st);
hipError_t launch_tri_g4(const TriCall &c, hipStream_t st);
hipError_t launch_tri_g5(const TriCall &c, hipStream_t st);
#if CORA_LDG & 2
TRI_GROUP(1)
#endif
#if CORA_LDG & 4
TRI_GROUP(2)
#endif
#if CORA_LDG & 8
TRI_GROUP(3)
#endif
#if CORA_LDG & 16
TRI_GROUP(4)
#endif
#if CORA_LDG & 32
TRI_GROUP(5)
#endif
#if CORA_LDG & 1
TRI_GROUP(0)
static hipError_t launch_tri(const TriCall &c, hipStream_t st) {
  if (c.ld <= 5) return launch_tri_g0(c, st);
  if (c.ld <= 9) return launch_tri_g1(c, st);
  if (c.ld <= 12) return launch_tri_g2(c, st);
  if (c.ld <= 16) return launch_tri_g3(c, st);
  if (c.ld <= 20) return launch_tri_g4(c, st);
  return launch_tri_g5(c, st);
}
hipError_t launch_blockop(const BlockOpDev &B, int ld, bool backward, const double *src, double *dst, hipStream_t st) {
  return launch_tri(TriCall{0, ld, backward, &B, nullptr, nullptr, nullptr, src, nullptr, nullptr, dst}, st);
}
hipError_t launch_rowop(const RowOpDev &op, int ld, const double *src0, const double *src, double *dst,
                        hipStream_t st, const RvTail *tail) {
  return launch_tri(TriCall{1, ld, false, nullptr, &op, nullptr, nullptr, src0, src, nullptr, dst, tail}, st);
}
hipError_t launch_subblock_fused(const SubOpDev &S, int ld, bool backward, const SubFuse &F, double *work, double *out,
                                 hipStream_t st) {
  // forward: the right-hand side is F.r, y -> out;  backward: y is read from `out`, v -> out
  return launch_tri(TriCall{3, ld, backward, nullptr, nullptr, &S, &F, backward ? out : F.r, nullptr, work, out}, st);
}
hipError_t launch_subblock(const SubOpDev &S, int ld, bool backward, const double *rhs_or_y, double *work, double *out,
                           hipStream_t st) {
  return launch_tri(TriCall{2, ld, backward, nullptr, nullptr, &S, nullptr, rhs_or_y, nullptr, work, out}, st);
}
hipError_t launch_kappa_finish(const double *partial, int n, StpcgState *state, hipStream_t st) {
  hipLaunchKernelGGL(k_kappa_finish, dim3(1), dim3(256), 0, st, partial, n, state);
  return hipGetLastError();
}
hipError_t launch_zero_row(double *x, size_t row, int ld, hipStream_t st) {
  hipLaunchKernelGGL(k_zero_row, dim3(1), dim3(64), 0, st, x, row, ld);
  return hipGetLastError();
}
#endif  // CORA_LDG & 1
#undef TRI_GROUP
#undef CASE

#endif  // CORA_TU & 4
}  // namespace cora

namespace cora {
#if CORA_TU & 2

hipError_t launch_fill_random(int64_t N, int k, unsigned long long seed, const int32_t *api2int, double *x, hipStream_t st) {
  if (N <= 0) return hipSuccess;
  const int ld = ld_for(k);
  hipLaunchKernelGGL(k_fill_random, dim3(grid_for(N * ld)), dim3(256), 0, st, N, ld, k, seed, api2int, x);
  return hipGetLastError();
}

hipError_t launch_gram(int64_t row0, int64_t rows, const double *A, int ka, const double *B, int kb,
                       double *partial, int nblocks, double *out, hipStream_t st) {
  hipLaunchKernelGGL(k_gram, dim3(nblocks), dim3(256), 0, st, row0, rows, A, ld_for(ka), ka, B, ld_for(kb), kb, partial);
  const int nel = ka * kb;
  hipLaunchKernelGGL(k_gram_reduce, dim3(nel), dim3(256), 0, st, partial, nblocks, nel, out);
  return hipGetLastError();
}

// n products in one launch + one reduction over all their elements: partial = n consecutive pieces (ka_e kb_e nblocks
// doubles each), out = the n results one after the other (row-major ka_e x kb_e) -- may be pinned host memory
hipError_t launch_gram_batch(int64_t row0, int64_t rows, int n, const double *const *A, const int *ka, const double *const *B,
                             const int *kb, double *partial, int nblocks, double *out, hipStream_t st) {
  if (n < 1 || n > 16) return hipErrorInvalidValue;
  GramBatch G{};
  int nel_all = 0;
  for (int e = 0; e < n; ++e) {
    G.A[e] = A[e];
    G.B[e] = B[e];
    G.lda[e] = ld_for(ka[e]);
    G.ka[e] = ka[e];
    G.ldb[e] = ld_for(kb[e]);
    G.kb[e] = kb[e];
    G.partial[e] = partial + static_cast<size_t>(nel_all) * nblocks;
    nel_all += ka[e] * kb[e];
  }
  hipLaunchKernelGGL(k_gram_batch, dim3(nblocks, n), dim3(256), 0, st, row0, rows, G);
  hipLaunchKernelGGL(k_gram_reduce, dim3(nel_all), dim3(256), 0, st, partial, nblocks, nel_all, out);
  return hipGetLastError();
}

// coef_host != nullptr and ncoef <= CombineCoef::kMax: the coefficients travel in the kernel's arguments (coef unused)
hipError_t launch_combine(int64_t row0, int64_t rows, int nblocks, const double *const *x, const int *kx,
                          const int *coff, const double *coef, int ncoef, int kout, double *out, hipStream_t st,
                          const double *coef_host) {
  CombineArgs A;
  for (int b = 0; b < 4; ++b) { A.x[b] = nullptr; A.kx[b] = 0; A.ldx[b] = 0; A.coff[b] = 0; }
  for (int b = 0; b < nblocks; ++b) { A.x[b] = x[b]; A.kx[b] = kx[b]; A.ldx[b] = ld_for(kx[b]); A.coff[b] = coff[b]; }
  A.nblocks = nblocks;
  A.kout = kout;
  A.ldo = ld_for(kout);
  const int grid = static_cast<int>(std::min<int64_t>((rows + 63) / 64, 2048));  // 4 wavefronts x 16 rows per block and step
  if (coef_host && ncoef <= CombineCoef::kMax) {
    CombineCoef K;
    for (int t = 0; t < ncoef; ++t) K.v[t] = coef_host[t];
    hipLaunchKernelGGL(k_combine_karg, dim3(std::max(grid, 1)), dim3(256), ncoef * sizeof(double), st, row0, rows, A, K, ncoef, out);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_combine, dim3(std::max(grid, 1)), dim3(256), ncoef * sizeof(double), st, row0, rows, A, coef,
                     ncoef, out);
  return hipGetLastError();
}

#endif  // CORA_TU & 2
}  // namespace cora

#if defined(CORA_SUB_TIMES) && (CORA_TU & 4) && (CORA_LDG & 1)
// measurement build only (-DCORA_SUB_TIMES, tools/sub_timeline.py): the block timestamps of the last forward and backward sweep
extern "C" int cora_debug_sub_phases(unsigned long long *out, int n_blocks) {
  if (n_blocks < 0 || static_cast<unsigned>(n_blocks) > cora::kSubTimesMax) return -1;
  for (int w = 0; w < 2; ++w)
    if (hipMemcpyFromSymbol(out + static_cast<size_t>(w) * cora::kSubPhases * n_blocks, HIP_SYMBOL(cora::g_sub_phase),
                            sizeof(unsigned long long) * cora::kSubPhases * n_blocks,
                            sizeof(unsigned long long) * w * cora::kSubPhases * cora::kSubTimesMax) != hipSuccess) return -2;
  return 0;
}
#endif
#if defined(CORA_SPMM_TIMES) && (CORA_TU & 1) && (CORA_LDG & 1)
// measurement build only (-DCORA_SPMM_TIMES, tools/spmm_timeline.py): the wavefront timestamps of the last k_spmm launch
extern "C" int cora_debug_spmm_times(unsigned long long *out, int n_blocks) {
  if (n_blocks < 0 || static_cast<unsigned>(n_blocks) > cora::kSpmmTimesMax) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cora::g_spmm_times), sizeof(unsigned long long) * 3 * n_blocks) == hipSuccess ? 0 : -2;
}
extern "C" int cora_debug_spmm_phases(unsigned long long *out, int n_blocks) {
  if (n_blocks < 0 || static_cast<unsigned>(n_blocks) > cora::kSpmmTimesMax) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cora::g_spmm_phase), sizeof(unsigned long long) * cora::kSpmmPhases * n_blocks) == hipSuccess ? 0 : -2;
}
#endif
