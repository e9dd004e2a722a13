// Level-scheduled sparse triangular solves on the device for the
// RegularizedCholesky / BlockCholesky preconditioner (reference
// src/CORA_preconditioners.cpp:46-83: CHOLMOD `solve` with p right-hand sides).
// The host supplies L (CSC, diagonal first per column) of P A P^T; indices are
// remapped to the handle's internal row order so the solves run in place on
// resident vectors.
#pragma once

#include <cstdint>
#include <vector>

namespace cora {

struct TriLevel {
  int32_t begin, end;  // range of ordered rows
  int32_t lanes;       // lanes cooperating on one row: 1, 8 or 64
};

struct TriHost {          // one direction, rows ordered by level
  std::vector<int32_t> rowptr, cols, out_row;
  std::vector<double> vals, dinv;
  std::vector<TriLevel> levels;
};

struct BorderHost {       // the trailing dense rows (landmarks) of L
  int nb = 0;                           // number of border rows
  std::vector<int32_t> out_row;         // internal row of each border row (elimination order)
  std::vector<double> Lbb;              // nb x nb dense lower triangle (row-major), incl. diagonal
  // W = L[border, non-border] in chunked CSR for the forward sweep
  std::vector<int32_t> chunk_row, chunk_begin, chunk_end;  // per chunk
  std::vector<int32_t> wcols;
  std::vector<double> wvals;
};

struct TriPlan {
  int m = 0;                 // order of the factor
  int32_t zero_row = -1;     // internal row forced to zero when the factor has N-1 rows
  TriHost fwd, bwd;
  BorderHost border;
  int64_t nnzL = 0;
  int height = 0;
};

// row_of[i]: internal row of permuted variable i (= api2int[perm[i]]).
void build_tri_plan(int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                    const std::vector<int32_t> &row_of, TriPlan &plan);

}  // namespace cora
