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

constexpr int kTriSn = 6;  // max rows of a supernode (its partial sums live in registers)

struct TriLevel {
  int32_t begin, end;  // range of supernodes
  int32_t lanes;       // lanes cooperating on one supernode: 1, 8 or 64
};

// One direction.  Rows are grouped into small chain supernodes (consecutive rows on a path of
// the elimination tree, at most kTriSn of them: typically the d rotation rows, range rows and
// translation of one pose).  A supernode is one unit of the level schedule: its external
// dependencies (rows of earlier levels) are gathered with full memory parallelism, then the
// tiny internal triangular block is solved in registers -- ~3x fewer levels than row by row.
// Everything a lane group needs about its supernode in ONE record addressed by the supernode
// index alone (no pointer chasing: a level is a chain of dependent loads, so each removed
// indirection is ~1 us per level).
struct TriSn {
  int32_t ext_begin, ext_end;   // external entries in cols / vals
  int32_t nrows, pad;
  int32_t out_row[kTriSn];      // internal row of each of its rows, in processing order
  double dinv[kTriSn];          // 1 / L_ii
  double lint[kTriSn * (kTriSn - 1) / 2];  // internal coefficients, packed: (t, q<t) at t(t-1)/2 + q
};

struct TriHost {
  std::vector<TriSn> sn;                // supernodes in level order
  std::vector<int32_t> cols;            // EXTERNAL entries: internal row | (row position << 28)
  std::vector<double> vals;
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
