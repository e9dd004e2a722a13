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

// ONE source in six pieces (round 6: the file had grown to 3 700 lines); this file holds the switches of the translation units
// and, at the end, the measurement builds' read-back hooks.  Inside namespace cora, in this order:
//   kernels/common.inc   row loads / stores, wave reductions, the STPCG state's scalar steps
//   kernels/spmm.inc     the sliced SpMM with fused epilogues (chain slices, row slices, long-row chunks)        CORA_TU & 1
//   kernels/rows.inc     row-unit, vector, reduction, exchange and LOBPCG block kernels                           CORA_TU & 2
//   kernels/tri.inc      the staged Cholesky solve: k_rowop, k_blockop, k_subblock                                 CORA_TU & 4
//   kernels/launch.inc   host-side launch wrappers (kernels.h), per translation unit and row-stride group
namespace cora {
#include "kernels/common.inc"
#include "kernels/spmm.inc"
#include "kernels/rows.inc"
#include "kernels/tri.inc"
#include "kernels/launch.inc"
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
