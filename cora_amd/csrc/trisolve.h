// Sparse Cholesky solves on the device for the RegularizedCholesky / BlockCholesky
// preconditioner and the translation-implicit formulation (reference
// src/CORA_preconditioners.cpp:46-83: CHOLMOD `solve` with p right-hand sides;
// src/CORA_problem.cpp:747-752: LtransCholRed_->solve).
//
// A level-scheduled triangular solve is a chain of ~60 dependent launches on these factors
// (elimination-tree height ~90), each bounded by launch + dependent-load latency, not by
// bandwidth.  Instead the rows are cut into a few STAGES along the elimination tree -- stage 0 is
// the union of the small subtrees at the bottom (a leaf or two of the nested dissection, <= 64
// rows), stage 1 the subtrees of what remains (<= 768 rows), ..., the last stage the top of the
// tree plus the dense landmark rows -- and the diagonal block of every stage (block diagonal: the
// subtrees of one stage do not touch each other) is inverted explicitly on the host, once.  A
// solve is then a fixed sequence of dependency-free sparse products
//
//   forward, stage k :  t_k = b_k - L[k, <k] y_{<k}       ("a": rows of L)
//                       y_k = W_k t_k                      ("b": W_k = L[k,k]^-1, sparse)
//   backward, stage k:  t_k = y_k - L[>k, k]^T x_{>k}      ("a": columns of L)
//                       x_k = W_k^T t_k                    ("b")
//
// i.e. at most 4K-2 launches for K stages (K = 3 at 10^5 poses) that each run at memory speed.  The host
// supplies L (CSC, diagonal first per column) of P A P^T; indices are remapped to the handle's
// internal row order so that the products read and write resident vectors directly.
#pragma once

#include <cstdint>
#include <vector>

namespace cora {

// One sparse product over a set of rows: dst[out_row] = src0[out_row] (if any) + sum_k val_k src[col_k].
// Rows are sorted by length class: [0, n8) short rows (8 lanes each), [n8, n8 + n64) one wavefront
// each, and `long` rows cut into chunks (one wavefront per chunk, partial sums reduced afterwards).
struct RowOpHost {
  std::vector<int32_t> out_row;         // internal row of each short / wavefront row
  std::vector<int32_t> begin, end;      // its entries in col / val
  int32_t n8 = 0, n64 = 0;
  std::vector<int32_t> long_out;        // internal row of each long row
  std::vector<int32_t> long_chunk_ptr;  // chunks of long row k: [ptr[k], ptr[k+1])
  std::vector<int32_t> chunk_begin, chunk_end;
  std::vector<int32_t> col;             // internal source row per entry
  std::vector<double> val;
  bool empty() const { return out_row.empty() && long_out.empty(); }
};

// Stage 0 in block form: its blocks are tiny (<= 64 rows, one wavefront each, lane = row).  The inverse
// W = L_bb^-1 of a block is non-zero exactly along the paths of the block's elimination tree, so it is
// stored without indices as a 64-bit lane mask per column plus the non-zeros in lane order -- packed
// twice so that both sweeps read it coalesced: by column for y = W t (mask_col[q]: lanes i >= q with
// W_iq != 0) and by row for x = W^T t (mask_row[q]: lanes j <= q with W_qj != 0).  A lane finds its
// entry at off + popcount(mask below the lane).  The backward coupling to the later stages (columns
// of L) is applied by the same kernel before the block product.
struct BlockOpHost {
  std::vector<int32_t> row_begin, nrows;  // per block: its rows in `rows`
  std::vector<int64_t> w_off;             // per block: start of its entries in w_by_col / w_by_row
  std::vector<int32_t> rows;              // internal row of every block row (elimination order)
  std::vector<uint64_t> mask_col, mask_row;  // per block row q
  std::vector<int32_t> off_col, off_row;     // per block row q: its first entry, relative to w_off
  std::vector<double> w_by_col, w_by_row;
  std::vector<int32_t> ext_ptr, ext_col;  // per block row: -L[later, row] entries (backward sweep)
  std::vector<double> ext_val;
};

// Stage 0 as WORKGROUP blocks solved by substitution (two-stage plans; the form used whenever what is left
// above the blocks is small enough for one explicit inverse).  A block is a subtree of the elimination tree
// of up to kSubRows rows whose part of L fits the LDS next to the block's right-hand sides; the workgroup
// copies both into LDS and runs the triangular solve level by level (rows of one level are independent;
// a level's tasks are (row, column of the right-hand side) pairs, long rows get several lanes per task).
// Reading L instead of an explicit inverse keeps the traffic at ~7 entries per row whatever the block size,
// so blocks can be large and the separator system above them tiny: two launches for all of it.
//   forward : T = rhs[rows];  level by level  T_i = sum_e val_e T[idx_e]  (a supernode -- a pose -- per level: the
//             inverse of its small dense diagonal block is folded into its rows, see trisolve_build.cpp);  y[rows] = T;
//             for every later-stage row g coupled to the block: aux[slot_g] = -sum_v L_gv y_v
//             (the later stage reads rhs_g + the aux rows of the blocks next to it: no gather product,
//             no split rows for the landmarks)
//   backward: T = y[rows] - L[later, rows]^T x[later];  level by level from the root, same form;  x[rows] = T
// Local row index = position in forward level order; the backward sweep has its own numbering (backward level order).
struct SubBlockOpHost {
  // per block
  std::vector<int32_t> row_begin, nrows;          // local rows of block b: [row_begin, row_begin + nrows)
  std::vector<int32_t> f_ent_begin, b_ent_begin;  // first in-block entry of the block (forward / backward arrays)
  std::vector<int32_t> f_nent, b_nent;            // ... and their number (padding included)
  std::vector<int32_t> f_lev_begin, b_lev_begin;  // first level header of the block in *_hdr; one extra entry
  std::vector<int32_t> tgt_begin;                 // first coupled later-stage row (target) of the block; one extra entry
  // per local row (global position row_begin + li)
  std::vector<int32_t> rows;                      // internal row
  // the backward sweep numbers the block's rows by backward level (position k, stored at row_begin + k)
  std::vector<int32_t> b_rows;                    // internal row of backward position k
  std::vector<int32_t> tgt_row;                   // per target: its row in the vectors (the backward sweep stages x[tgt_row] in tile row nrows + k)
  // A BARRIER LEVEL of the kernel is 4 headers, one per wavefront of the workgroup: {first row the wavefront solves in
  // the level (block-relative), lanes per row g (power of two <= 64) | entries per lane npl (<= 8) << 8 | rows << 12,
  // first coefficient (block-relative), first index}; the rows of a wavefront hold exactly g * npl entries each (null
  // padded) -- its own width, so short rows do not pay for the level's longest; a wavefront takes whole supernodes (their
  // rows read each other's right-hand sides: one barrier level), rows = 0 marks a wavefront without work.  A level of the
  // elimination tree that needs more than 4 wavefronts continues in the next barrier level.  nlev + 1 groups of 4
  // headers per block, the last one closes the row range (f_lev_begin / b_lev_begin count headers).
  std::vector<int32_t> f_hdr, b_hdr;
  // in-block entries
  std::vector<uint16_t> f_idx, b_idx;             // local column (forward) / backward position of the row (backward)
  std::vector<double> f_val, b_val;               // -L_ij
  // forward contributions
  std::vector<int32_t> tgt_slot;                  // aux row written by the target
  std::vector<int32_t> c_ptr;                     // entries of the target (one extra entry at the end)
  std::vector<uint16_t> c_idx;                    // local row v
  std::vector<double> c_val;                      // -L_gv
  int32_t n_aux = 0;                              // aux rows in total
  int32_t max_rows = 0, max_ent = 0, max_lev = 0; // largest block (LDS sizing); max_lev counts headers
  int32_t max_level_lanes = 0, max_npl = 0;       // widest level (rows x lanes per row), most entries per lane
};

struct TriStage {
  RowOpHost fwd_a, fwd_b, bwd_a, bwd_b;  // fwd_a is empty for the first stage, bwd_a for the last
  bool dense = false;                    // stage 0 only: `blocks_op` replaces fwd_b, bwd_a and bwd_b
  BlockOpHost blocks_op;
  bool sub = false;                      // stage 0 only: `sub_op` replaces them, and the next stage's fwd_a too
  SubBlockOpHost sub_op;
  int32_t rows = 0, blocks = 0;
};

struct TriPlan {
  int m = 0;               // order of the factor
  int32_t zero_row = -1;   // internal row forced to zero when the factor has N-1 rows
  std::vector<TriStage> stages;
  int64_t nnzL = 0, nnzW = 0;  // entries of L and of the explicit block inverses
  int height = 0;              // number of stages
  bool groups_whole = false;   // see build_tri_plan
  int32_t aux_base = 0;        // sub plans: aux row a of the forward sweep lives at row aux_base + a of the work vector
  std::vector<int32_t> top_rows;  // sub plans: internal rows of the last stage (+ the pinned row)
};

// group (optional, size m): variables with the same id >= 0 (the d rotation rows of a pose) are kept in
// one block; plan.groups_whole tells whether they also sit in adjacent lanes / consecutive positions.
// aux_base >= 0 allows the two-stage workgroup-block form (SubBlockOpHost): the caller's work vectors then
// have aux_base + plan.stages[0].sub_op.n_aux rows.
// row_of[i]: internal row of permuted variable i (= api2int[perm[i]]).  zero_row >= 0: an internal
// row outside the factor that every solve must set to zero (the pinned variable).
void build_tri_plan(int m, const int32_t *Lp, const int32_t *Li, const double *Lx,
                    const std::vector<int32_t> &row_of, int32_t zero_row, TriPlan &plan,
                    const std::vector<int32_t> *group = nullptr, int32_t aux_base = -1);


// Test hook: runs the plan's products on the host in launch order (rhs, out: `rows` doubles, one
// right-hand side, indexed by internal row).  Never used by a compute entry point.
void tri_plan_solve_host(const TriPlan &plan, int64_t rows, const double *rhs, double *out);

}  // namespace cora
