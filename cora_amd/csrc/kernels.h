// Kernel argument blocks and launcher declarations (host-visible).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "cora_internal.h"
#include "trisolve.h"

namespace cora {

// EPI_HVP_K: EPI_HVP that also leaves per-block partial sums of <X, out> in SpmmArgs::kappa_partial (the curvature
// kappa = <p, Hp> of an STPCG iteration without a pass of its own)
enum Epilogue : int { EPI_NONE = 0, EPI_S = 1, EPI_HVP = 2, EPI_HVP_K = 3 };

struct StpcgState;
struct SpmmArgs {
  const SliceDesc *slices;
  int n_slices;
  int n_chunks;        // set by the caller: number of long-row chunks
  int n_real_chunks;   // filled by launch_spmm
  int n_slice_blocks;  // filled by launch_spmm
  const double *sval;
  const double *head_val;  // [pose slice][kChainHead(d)]: what lane 0 of a chain slice takes from the pose before it
  const int32_t *scol;
  const int32_t *perm;
  const LongChunk *chunks;
  const int32_t *chunk_order;
  const double *lval;
  const int32_t *lcol;
  double *partials;    // [n_chunks][kMaxLD]
  unsigned *tickets;   // [n_long_rows], zero between launches
  const double *X;     // rows x LD
  double *out;         // rows x LD (only local rows are written)
  const double *Y;     // current point (epilogues)
  const double *lam_st;  // [local pose][d*d]
  const double *lam_ob;  // [local range]
  // EPI_HVP_K: [launch_spmm_kappa_slots()] one partial sum of <X, out> per block, then one per long row (written by
  // whichever chunk finishes the row -- a fixed slot whatever the arrival order)
  double *kappa_partial = nullptr;
  // partitioned handles: [n_long_rows][ld] partial sums of the DISTRIBUTED long rows (zeroed before the launch, summed
  // over the ranks after it: capi.hip, finish_long_rows); nullptr: long rows are whole and written to `out`
  double *long_out = nullptr;
  int n_long_rows = 0;      // set by the caller (HostFormat::n_long_rows)
  int kappa_long_base = 0;  // filled by launch_spmm

  // internal row ranges [lo, hi) of the LOCAL rotation and translation rows of X: the pose slices clip their LDS
  // windows of X to them (kernels.hip, pose_slice); empty ranges switch the windows off, never the result
  int32_t win_rot_lo = 0, win_rot_hi = 0, win_trn_lo = 0, win_trn_hi = 0;
  // the same slices with the pose slices first inside each XCD's eighth of the list (HostFormat::slices_pose_first);
  // launch_spmm takes this order up to a row stride of kPoseFirstMaxLD (measured: better below, worse above)
  const SliceDesc *slices_pose_first = nullptr;
  int32_t n_local_poses = 0;  // poses of the shard (chain slices clamp their implied columns to them)
  int32_t win_on = 0;  // set by launch_spmm: X window + cooperative epilogue of the pose slices (n_slices >= kWinMinSlices)
};
extern int g_win_min_slices;  // launch_spmm: window form from this many slices on (kernels.hip)
constexpr int kPoseFirstMaxLD = 6;
constexpr int kWinMinSlices = 2048;  // = the wavefronts resident at once (256 CUs x 8)
// number of blocks (= kappa partials) of a launch with these arguments
inline int launch_spmm_blocks(const SpmmArgs &A) { return ((A.n_chunks + 7) & ~7) + 8 * ((A.n_slices + 7) / 8); }
// number of kappa partials an EPI_HVP_K launch with these arguments writes (every slot, every launch)
inline int launch_spmm_kappa_slots(const SpmmArgs &A) { return launch_spmm_blocks(A) + A.n_long_rows; }

struct RowArgs {
  int d;
  int nl_poses, nl_ranges, nl_trans;
  size_t rot_base, rng_base, trn_base, base;
};

// Scalars of one Steihaug-Toint PCG solve, resident on the device so that an iteration needs no host
// round trip: the inner-product kernels update them in their last block, the vector kernels read their
// coefficients from here.  status: 0 running, 1 residual target met, 2 stopped on the trust-region
// boundary (or negative curvature), 3 iteration limit.  Once status != 0 all coefficients are neutral, so
// iterations the host has already enqueued do not change s, r or p.
struct StpcgState {
  double r_v, sigma_M2, s_Mp, p_M2;  // recurrences of the M-norms (Conn, Gould & Toint, Alg. 7.5.1)
  double Delta2, target;
  double alpha, coef_s, coef_r, coef_v, coef_beta;
  double kappa, rr, step_M_norm;
  int iters, status, max_iters, pad;
};
enum { DOTS_PLAIN = 0, DOTS_STPCG_KAPPA = 1, DOTS_STPCG_BETA = 2, DOTS_STPCG_RR = 3, DOTS_STPCG_RV = 4, DOTS_STPCG_KAPPA_RR = 5 };

struct DotArgs {
  const double *a[4];
  const double *b[4];
  int count;
  int64_t n2;        // number of doubles (launch_dots halves it for the double2 kernel)
  double *partial;   // [count][gridDim.x]
  unsigned *ticket;  // zero between launches; the last block to finish reduces the partials
  double *out;       // [count] results, written by that block (may be pinned host memory)
  unsigned long long *seq_out;  // optional (pinned): set to `seq` after the results are visible to the host
  unsigned long long seq;
  int mode;                     // DOTS_*: what the last block does with the results
  StpcgState *st, *st_host;     // device state and its pinned mirror (DOTS_STPCG_*)
  // non-null: the sequence number is ++(*seq_counter) (device memory) instead of `seq` -- launches replayed from a
  // hipGraph carry no per-launch argument; the host keeps the counter equal to its own count (capi.hip, stpcg_run)
  unsigned long long *seq_counter = nullptr;
};

struct RowOpDev {  // device copy of a RowOpHost (trisolve.h)
  const int32_t *out_row, *begin, *end;
  int n8, n64;
  const int32_t *long_out, *long_chunk_ptr, *chunk_begin, *chunk_end;
  int nlong, nchunks;
  const int32_t *col;
  const double *val;
  const int32_t *chunk_row;  // long-row ordinal of every chunk
  double *partial;     // [nchunks][kMaxLD]
  unsigned *tickets;   // [nlong], zero between launches
};
// Device copy of a BlockOpHost (trisolve.h), packed so that everything wave-uniform is one scalar load:
// a 16-byte descriptor per block and a 16-byte record per block row (its internal row for the lane that
// owns it; the lane mask and offset of column / row q of W for the wave's loop over q).
struct BlockDesc { int32_t row_begin, nrows, meta_begin, pad; int64_t w_off; int64_t pad2; };  // 32 bytes
struct BlockLane { uint64_t mask; int32_t off; int32_t row; };
struct BlockOpDev {
  const BlockDesc *desc;
  const BlockLane *by_col, *by_row;     // [desc.meta_begin + q]: column q of W (forward) / row q of W (backward)
  const double *w_by_col, *w_by_row;
  const int32_t *ext_ptr, *ext_col;
  const double *ext_val;
  int nblocks;
};
// Device copy of a SubBlockOpHost (trisolve.h): stage 0 as workgroup blocks solved by substitution in LDS.
struct SubDesc {  // 88 bytes per block
  int32_t row_begin, nrows;
  int32_t f_ent_begin, f_nent, b_ent_begin, b_nent;
  int32_t f_lev_begin, f_nlev, b_lev_begin, b_nlev;
  int32_t tgt_begin, ntgt;
  int32_t unit_begin, nunits;  // backward, fused projection: the block's row units in SubOpDev::b_unit
  // The block's rows in MEMORY order are a few runs of consecutive rows (a chain block: its poses' rotation rows, their
  // range rows, their translation rows): the k-th row in memory order is  k + run_off[r]  for the first r with
  // k < run_end[r] (unused runs end at INT32_MAX).  With SubOpDev::io_runs the sweeps form their global addresses from
  // these eight scalars -- the loads of a phase no longer wait for an index list -- and take the tile position of a row
  // from a 16-bit list (SubSweep::tpos).
  int32_t run_off[4], run_end[4];
};
constexpr int kSubMaxRuns = 4;
struct SubSweep {  // one direction of the solve; rows are numbered by level within the block
  const int32_t *rows;      // [row_begin + k]: internal row
  const int32_t *hdr;       // [4 * (lev_begin + l)]: {first row, lanes per row g | entries per lane npl << 8, first coefficient (block-relative), first index (absolute)} of level l
  const uint16_t *idx;      // local row a block entry multiplies: per level [lane = row * g + part][4 or 8]
  // val: per level [slot u < npl][lane]
  const double *val;        // coefficient
  // [row_begin + k]: {internal row, tile position} of the block's k-th row in MEMORY order: the tile is filled and
  // written back element by element in that order (a block is a few runs of consecutive rows: coalesced)
  const int2 *io;
  const uint16_t *tpos;  // [row_begin + k]: the tile position alone (SubOpDev::io_runs)
};
struct SubOpDev {
  const SubDesc *desc;
  SubSweep fwd, bwd;
  const int32_t *tgt_row;        // backward: rows of the last stage coupled to the block, staged behind the block's rows in the tile
  const int32_t *tgt_slot, *c_ptr;  // forward: aux rows written for the last stage
  const uint16_t *c_idx;
  const double *c_val;
  const int32_t *top_rows;  // rows of the last stage + the pinned row
  const int2 *b_unit;       // backward, fused projection: {tile position, internal row} of every row unit of a block (a pose's
                            // first rotation row -- the others follow it in the tile --, a range row, a translation row)
  int nblocks, ntop, max_rows, max_ent, max_lev, aux_base;
  int max_level_lanes, max_npl;  // widest level (rows x lanes per row) and most entries per lane of the plan
  int io_runs;  // 1: every block has at most kSubMaxRuns runs of consecutive rows (SubDesc::run_off / run_end are valid)
};
// STPCG passes fused into the two sweeps (cora_stpcg_dev; one shard, explicit formulation):
//   forward : the right-hand side IS the residual and is updated on the way in:  r += coef_r Hp; every row of the vector
//             is a block row or a row of the last stage; every block leaves <r, r> and |y|^2 over its rows in slots
//   backward: the solution is projected on the way out, v = Proj_Y(x), and the step and the direction are updated in the
//             same epilogue (s += coef_s p, p = coef_v v + coef_beta p); needs the d rotation rows of a pose at
//             consecutive tile positions (checked when the factor is installed)
// Between the sweeps, an extra block of the last stage's second product (RvTail) adds the slots and the squared norms of
// the rows of t_1 in fixed order and runs the scalar steps that follow <r, r> and <r, v>: no ticket, no atomics.
struct RvTail {
  const double *rr_partial;  // [n_rr]  <r, r> slots of the forward sweep (one per block of its launch)
  int n_rr;
  const double *yy_partial;  // [n_yy]  |y|^2 slots of the forward sweep's solve blocks
  int n_yy;
  const double *rowsq;       // [n_rowsq] squared norms of the rows of t_1, a slot per wavefront (the last stage's first product leaves them)
  int n_rowsq;
  StpcgState *st, *st_host;  // st == nullptr: no tail block in this launch ...
  double *rowsq_out;         // ... but, if set, the product leaves the squared norms of its rows in rowsq_out[rowop_rowsq_slots]
  unsigned long long *seq_out;  // pinned: set to seq after the mirror is visible to the host
  unsigned long long seq;
  unsigned long long *seq_counter;  // non-null: seq = ++(*seq_counter), see DotArgs
  // n_kappa > 0: the iteration's kappa step has not run yet (SubFuse::n_kappa): the block adds the partials and runs it
  // before the two steps of its own
  const double *kappa_partial;
  int n_kappa;
  // non-null (partitioned handle): the block leaves THIS RANK'S sums -- sums_out[0] = <r, r>, sums_out[1] = <r, v> -- and
  // touches neither the state nor the mirror: the sums are added over the ranks first (one all-reduce), then
  // k_stpcg_scalar_step runs the scalar step
  double *sums_out;
};
struct SubFuse {
  DotArgs dot;                 // only dot.st (the coefficients of the state) is used
  const double *Hp = nullptr;  // forward
  double *r = nullptr;         // forward: the right-hand side (updated in place)
  // forward: slot b of rr_partial receives <r, r> over the rows of block b of the launch (solve blocks, then the blocks
  // that carry the rows of the last stage); slot b of yy_partial |y|^2 over the rows of solve block b (y = L^-1 r).  With
  // |t_1|^2 of the last stage they make <r, v> = <r, Proj_Y(L^-T L^-1 r)> = |L^-1 r|^2 (r is a tangent vector and
  // Proj_Y is an orthogonal projector), so beta is known BEFORE the backward sweep (RvTail, k_rowop) and the direction
  // update rides on its epilogue.
  double *rr_partial = nullptr, *yy_partial = nullptr;
  const double *Y = nullptr;   // backward: the current point
  double *p = nullptr, *s = nullptr;  // backward: s += coef_s p, then p = coef_v v + coef_beta p  (v = Proj_Y(x), not stored)
  int d = 0;
  int64_t rot_base = 0, rng_base = 0, trn_base = 0;  // internal rows: rotations | ranges | translations
  // forward, n_kappa > 0: kappa = <p, Hp> has NOT been finished by a launch of its own -- every block adds the product's
  // n_kappa partial sums itself (same order, same bits: kappa_sum_256) and runs the scalar step on a private copy of the
  // state to get coef_r; the state itself is advanced later, by the tail block of the last stage (RvTail::n_kappa), which
  // adds the same partials the same way.  One launch per iteration less; worth it while blocks x partials is small.
  const double *kappa_partial = nullptr;
  int n_kappa = 0;
};
inline int launch_subblock_blocks(const SubOpDev &S) { return S.nblocks + (S.ntop + 255) / 256; }
hipError_t launch_subblock_fused(const SubOpDev &S, int ld, bool backward, const SubFuse &F, double *work, double *out,
                                 hipStream_t st);

// forward : y[block rows] = L_bb^-1 rhs[block rows] -> y;  work[aux rows] = couplings to the last stage;  work[top rows] = rhs[top rows]
// backward: x[block rows] = L_bb^-T (y[block rows] - L[top, rows]^T work[top rows]) -> x (may be y);  x[top rows] = work[top rows]
hipError_t launch_subblock(const SubOpDev &S, int ld, bool backward, const double *rhs_or_y, double *work, double *out,
                           hipStream_t st);

// forward: dst[rows] = W src[rows];  backward: dst[rows] = W^T (src[rows] - L[later, rows]^T src[later])
hipError_t launch_blockop(const BlockOpDev &B, int ld, bool backward, const double *src, double *dst, hipStream_t st);
// dst[out_row] = (src0 ? src0[out_row] : 0) + sum_k val_k * src[col_k] for every row of the product
// kappa = sum of the n partials of an EPI_HVP_K product (fixed order), then the scalar step that follows it
hipError_t launch_kappa_finish(const double *partial, int n, StpcgState *state, hipStream_t st);
// tail: optional RvTail (see SubFuse) -- per-row squared norms out, or one extra block that runs the reductions
// slots a product leaves its rows' squared norms in (RvTail::rowsq_out): one per wavefront of the 8-lane class, one per row of the other two
#ifdef CORA_ROWSQ_PER_ROW  // (lab: a slot per row, the form before)
inline int rowop_rowsq_slots(const RowOpDev &op) { return op.n8 + op.n64 + op.nlong; }
#else
inline int rowop_rowsq_slots(const RowOpDev &op) { return 4 * ((op.n8 + 31) >> 5) + op.n64 + op.nlong; }
#endif
hipError_t launch_rowop(const RowOpDev &op, int ld, const double *src0, const double *src, double *dst,
                        hipStream_t st, const RvTail *tail = nullptr);
hipError_t launch_gram(int64_t row0, int64_t rows, const double *A, int ka, const double *B, int kb,
                       double *partial, int nblocks, double *out, hipStream_t st);
hipError_t launch_fill_random(int64_t N, int k, unsigned long long seed, const int32_t *api2int, double *x, hipStream_t st);
hipError_t launch_gram_batch(int64_t row0, int64_t rows, int n, const double *const *A, const int *ka, const double *const *B,
                             const int *kb, double *partial, int nblocks, double *out, hipStream_t st);
constexpr int kCombineKargMax = 400;  // coefficients that travel in the kernel's arguments (CombineCoef::kMax)
hipError_t launch_combine(int64_t row0, int64_t rows, int nblocks, const double *const *x, const int *kx,
                          const int *coff, const double *coef, int ncoef, int kout, double *out, hipStream_t st,
                          const double *coef_host = nullptr);
hipError_t launch_zero_row(double *x, size_t row, int ld, hipStream_t st);

hipError_t launch_spmm(const SpmmArgs &A, int ld, int d, int epi, hipStream_t st);
hipError_t launch_point_finish(const RowArgs &R, int ld, const double *Y, const double *G,
                               double *rgrad, double *lam_st, double *lam_ob, double *partial,
                               int *nblocks, hipStream_t st);
hipError_t launch_tangent_project(const RowArgs &R, int ld, const double *Y, const double *V,
                                  const double *scale, double *out, hipStream_t st);
hipError_t launch_project_manifold(const RowArgs &R, int ld, const double *A, const double *V,
                                   double alpha, double *out, hipStream_t st);
hipError_t launch_axpby(int64_t n, double a, const double *x, double b, double *y, hipStream_t st);
hipError_t launch_axpy2(int64_t n, double a1, const double *x1, double *y1, double a2, const double *x2,
                        double *y2, hipStream_t st);
// s += coef_s p, r += coef_r Hp  /  p = coef_v v + coef_beta p  with the coefficients of the device state
hipError_t launch_stpcg_update(int64_t n, const StpcgState *S, const double *p, const double *Hp, double *s,
                               double *r, hipStream_t st);
hipError_t launch_stpcg_direction(int64_t n, const StpcgState *S, const double *v, double *p, hipStream_t st);
// the fused passes of the device-resident STPCG iteration (cora_stpcg_dev):
//   r += coef_r Hp with <r, r> (DOTS_STPCG_RR)  |  out = Proj_Y(V) with <r, out> (DOTS_STPCG_RV)  |
//   s += coef_s p, then p = coef_v v + coef_beta p
hipError_t launch_stpcg_residual(const DotArgs &D, int64_t n, const double *Hp, double *r, hipStream_t st);
// v = Proj_Y(X) consumed at once by  s += coef_s p, p = coef_v v + coef_beta p  (row strides up to 12)
hipError_t launch_tangent_project_update(const RowArgs &R, const StpcgState *S, int ld, const double *Y, const double *X,
                                         double *p, double *s, hipStream_t st);
// kappa from the nk partials of an EPI_HVP_K product, the scalar step, then r += coef_r Hp with <r, r> -- ONE launch:
// every block adds the partials itself (same order, same bits).  For the small plans, where a launch is what costs.
hipError_t launch_kappa_residual(const DotArgs &D, const double *kpartial, int nk, int64_t n, const double *Hp, double *r,
                                 hipStream_t st);
// The same pass without the scalar step: a block leaves its share of <r, r> in rr_slot[block] (kappa_residual_slots_blocks(n)
// slots) and the state is advanced by the tail block of a later launch (RvTail::n_kappa, RvTail::n_rr) -- no ticket here.
int kappa_residual_slots_blocks(int64_t n);
hipError_t launch_kappa_residual_slots(const StpcgState *S, const double *kpartial, int nk, int64_t n, const double *Hp, double *r,
                                       double *rr_slot, hipStream_t st);
hipError_t launch_tangent_project_dot(const RowArgs &R, const DotArgs &D, int ld, const double *Y, const double *V,
                                      const double *scale, const double *r, double *out, hipStream_t st);
// the scalar step of a partitioned handle's iteration, after the all-reduce of its inner products (k_stpcg_scalar_step)
hipError_t launch_stpcg_scalar_step(int what, const double *vals, StpcgState *state, StpcgState *state_host,
                                    unsigned long long *seq_out, unsigned long long seq, hipStream_t st);
// s = 0, r = g, p = -Pg (start of a solve whose preconditioned gradient is known)
hipError_t launch_stpcg_init(int64_t n, const double *g, const double *Pg, double *s, double *r, double *p, hipStream_t st);
hipError_t launch_stpcg_step_direction(int64_t n, const StpcgState *S, const double *v, double *p, double *s,
                                       hipStream_t st);
hipError_t launch_scale_rows(int64_t rows, int ld, const double *scale, const double *x, double *y,
                             hipStream_t st);
hipError_t launch_dots(const DotArgs &D, int *nblocks, hipStream_t st);
hipError_t launch_reduce_partials(const double *partial, int nblocks, int count, double *out,
                                  hipStream_t st);
// distributed long rows after the sum over the ranks: owner copies slot j -> out[rows[j]]; kappa != nullptr: kappa[j] = the
// row's share of <X, out> (0 on the other ranks)
hipError_t launch_long_finish(int n_long, int ld, int rank, const int32_t *rows, const int32_t *owner, const double *slots,
                              const double *X, double *out, double *kappa, hipStream_t st);
// the two ends of a partitioned product's exchange (kernels.hip, k_exchange_pack / k_exchange_unpack)
hipError_t launch_scatter_shard_rows(int world, int rank, int64_t maxn, int ld, int64_t shard_rows, const int64_t *meta,
                                     const double *recv, double *X, hipStream_t st);
hipError_t launch_exchange_pack(int64_t n, int ld, const int32_t *rows, int64_t ztail, const double *src, double *dst, hipStream_t st);
hipError_t launch_exchange_unpack(int world, int64_t e_max, int n_long, int ld, const int32_t *recv_idx, const double *recv, double *X,
                                  int rank, const int32_t *long_rows, const int32_t *long_owner, double *out, double *kappa,
                                  hipStream_t st);
hipError_t launch_has_nan(int64_t n, const double *x, int *flag, hipStream_t st);
// mode 0: dst[k] = src[rows[k]];  1: dst[rows[k]] = src[k];  2: dst[rows[k]] = src[rows[k]]  (rows of ld doubles)
hipError_t launch_move_rows(int mode, int64_t n, int ld, const int32_t *rows, const double *src, double *dst,
                            hipStream_t st);
hipError_t launch_upload(int64_t N, int k, int ld, const double *src, const int32_t *api2int,
                         double *dst, hipStream_t st);
hipError_t launch_download(int64_t N, int k, int ld, const double *src, const int32_t *api2int,
                           double *dst, hipStream_t st);

}  // namespace cora
