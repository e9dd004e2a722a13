/*
 * cora_host.h -- flat C view of the C++ host (cora_amd/csrc/host: CORA::Problem,
 * parsePyfgTextToProblem, solveCORA ...).  The C++ host mirrors the reference's
 * own C++ interface (include/CORA/CORA.h:19-37, include/CORA/CORA_problem.h:
 * 67-416, include/CORA/pyfg_text_parser.h:31); this header only exists so that
 * tests/ and bench.py can drive it through ctypes.  All matrices are
 * column-major double with an explicit leading dimension.  Every function
 * returns 0 on success; cora_host_last_error() holds the C++ exception text.
 */
#ifndef CORA_HOST_H_
#define CORA_HOST_H_

#include <stdint.h>

#include "cora_hip.h" /* cora_exchange_fn, cora_allreduce_fn, cora_allgather_fn */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cora_problem cora_problem;

const char *cora_host_last_error(void);

/* parsePyfgTextToProblem (src/pyfg_text_parser.cpp:112) */
int cora_problem_from_pyfg(const char *path, cora_problem **out);
/* Synthetic generator of SURVEY 8(d); pyfg_out may be NULL. precond: cora_precond_kind */
int cora_problem_synthetic(int dim, int n_poses, int n_landmarks, int n_ranges, int n_loops, uint64_t seed,
                           int precond, const char *pyfg_out, cora_problem **out);
/* Programmatic construction, one call per add* method of CORA::Problem
 * (include/CORA/CORA_problem.h:202-262; used like tests/test_construct_problem.cpp:21-99).  Symbols are the
 * reference's strings ("x1", "L3", ...).  Matrices are column-major: R d x d, t / pos d, cov (d + rotation
 * dim) square for poses (translation block first, include/CORA/Measurements.h:60-63) and d x d for landmarks. */
int cora_problem_new(int dim, int rank, int implicit, int precond, cora_problem **out);
int cora_problem_add_pose(cora_problem *p, const char *id);
int cora_problem_add_landmark(cora_problem *p, const char *id);
int cora_problem_add_range(cora_problem *p, const char *a, const char *b, double dist, double cov);
int cora_problem_add_rel_pose(cora_problem *p, const char *a, const char *b, const double *R, const double *t,
                              const double *cov);
int cora_problem_add_rel_pose_landmark(cora_problem *p, const char *a, const char *b, const double *t,
                                       const double *cov);
int cora_problem_add_pose_prior(cora_problem *p, const char *id, const double *R, const double *t, const double *cov);
int cora_problem_add_landmark_prior(cora_problem *p, const char *id, const double *pos, const double *cov);

/* Same generator with explicit noise levels sigmas = {sigma_t, sigma_R, sigma_range} (NULL: the
 * defaults 0.05, 0.01, 0.1) and, when x_gt != NULL, the ground truth as a column-major
 * ((dim+1) n + l + r) x dim point ([R_i^T]; range bearings; translations) -- zero cost when all
 * sigmas are zero, which gives the tests a known global optimum at any size. */
int cora_problem_synthetic_ex(int dim, int n_poses, int n_landmarks, int n_ranges, int n_loops, uint64_t seed,
                              int precond, const double *sigmas, const char *pyfg_out, double *x_gt,
                              cora_problem **out);
void cora_problem_destroy(cora_problem *p);

/* Problem::updateProblemData (src/CORA_problem.cpp:500-510): assembles Q on the host. */
int cora_problem_update(cora_problem *p);
/* dims: [0] d, [1] poses, [2] landmarks, [3] ranges, [4] N, [5] nnz(Q), [6] pose-pose
 * measurements, [7] relaxation rank */
int cora_problem_dims(const cora_problem *p, int64_t dims[8]);
/* Borrowed CSR pointers of a matrix by name: "DataMatrix", "Arange", "OmegaRange",
 * "RangeDistances", "Apose", "OmegaPose", "T", "RotConLaplacian". */
int cora_problem_matrix(cora_problem *p, const char *name, int64_t *rows, int64_t *cols, int64_t *nnz,
                        const int32_t **rowptr, const int32_t **colidx, const double **vals);
/* Problem::get_certificate_matrix(Y) (src/CORA_problem.cpp:1162-1166): S = Q - Lambda(Y) as CSR, Y column-major with the
 * current relaxation rank's columns; the arrays stay valid until the next call on this problem. */
int cora_problem_certificate_matrix(cora_problem *p, const double *Y, int ldy, int64_t *rows, int64_t *nnz,
                                    const int32_t **rowptr, const int32_t **colidx, const double **vals);

int cora_problem_set_rank(cora_problem *p, int rank);
int cora_problem_set_preconditioner(cora_problem *p, int kind);
int cora_problem_set_device(cora_problem *p, int device);
/* Problem::setPartition (this build's multi-GPU entry, cora_amd/csrc/host/CORA_problem.h): the process owns
 * partition `rank` of `world` and communicates through the callbacks of cora_set_comm (include/cora_hip.h). */
int cora_problem_set_partition(cora_problem *p, int rank, int world, cora_exchange_fn exchange,
                               cora_allreduce_fn allreduce, cora_allgather_fn allgather, void *user);
/* Problem::setFormulation (include/CORA/CORA_problem.h:338): implicit != 0 selects
 * Formulation::Implicit, where the variable is the leading d*n + r rows (rotations and ranges)
 * and the translations are eliminated analytically (src/CORA_problem.cpp:714-753). */
int cora_problem_set_formulation(cora_problem *p, int implicit);
/* Problem::getExpectedVariableSize (src/CORA_problem.cpp:944-952) */
int cora_problem_variable_size(cora_problem *p, int64_t *rows);

/* Operators of CORA::Problem, by name (all run on the GPU):
 *  "evaluateObjective"        A=Y                -> out[0] (scalar)
 *  "Euclidean_gradient"       A=Y                -> out N x p
 *  "Riemannian_gradient"      A=Y                -> out
 *  "tangent_space_projection" A=Y, B=Ydot        -> out
 *  "Riemannian_Hessian_vector_product" A=Y, B=nablaF_Y, C=Ydot -> out
 *  "precondition"             A=V                -> out
 *  "projectToManifold"        A                  -> out
 *  "retract"                  A=Y, B=V           -> out
 *  "getRandomInitialGuess"    (none)             -> out
 *  "getOdomInitialization"    (none)             -> out   (examples/paper_experiments.cpp:426-534)
 *  "getTranslationExplicitSolution" A=Y (implicit) -> out (d+1)n + r + l rows (src/CORA_problem.cpp:1168)
 *  "alignEstimateToOrigin"    A=Y (rank d)       -> out (d+1)n + r + l rows (src/CORA_problem.cpp:1234)
 * Inputs are N x cols with leading dimension N = cora_problem_variable_size() (cols is checked
 * against the relaxation rank by the C++ methods, like the reference's checkMatrixShape); the
 * output is N x rank unless stated otherwise. */
int cora_problem_op(cora_problem *p, const char *op, int cols, const double *A, const double *B,
                    const double *C, double *out);
/* compute_Lambda_blocks(Y): stiefel d x dn (ld d), oblique r */
int cora_problem_lambda_blocks(cora_problem *p, const double *Y, double *stiefel, double *oblique);

/* Riemannian TNT (the call of src/CORA.cpp:139-140 with the parameters of :95-109) from x0
 * (N x rank).  opts (may be NULL): [0] max_iterations, [1] max_TPCG_iterations, [2] gradient
 * tolerance, [3] preconditioned gradient tolerance, [4] max seconds, [5] verbose, [6] non-zero: drive
 * the inner STPCG from the host instead of cora_stpcg_dev (7 entries; 0 keeps a default).
 * stats out: [0] f, [1] |grad|, [2] |P grad|, [3] outer iterations, [4] Hessian-vector products,
 * [5] status (TNTStatus), [6] seconds. */
int cora_problem_tnt(cora_problem *p, const double *x0, const double *opts, double *x_out, double stats[7]);
/* ONE outer iteration of the same solver from x with trust-region radius Delta (test hook: the iteration is compared
 * with the CPU restatement step by step, so that rounding differences cannot accumulate over a chaotic trajectory).
 * out: [0] f at x_out, [1] the radius after the update, [2] inner (STPCG) iterations, [3] gain ratio rho,
 * [4] 1 if the step was accepted (x_out = the new point, else x), [5] |h|, [6] |h|_M, [7] status as in cora_problem_tnt. */
int cora_problem_tnt_step(cora_problem *p, const double *x, double Delta, int host_stpcg, double *x_out, double out[8]);

/* Problem::certify_solution(Y, eta, nx, bootstrap = Y) (src/CORA_problem.cpp:1030-1103).
 * out: [0] is_certified, [1] theta, [2] LOBPCG iterations; x: N (direction of negative curvature or 0). */
int cora_problem_certify(cora_problem *p, const double *Y, double eta, int nx, double out[3], double *x);

/* Two certifications in a row at Y, the second started from the first one's Ritz block -- handed over through the host
 * (resident = 0: Problem::certify_solution, all_eigvecs as the next bootstrap, the reference's flow src/CORA.cpp:158-170)
 * or left on the device (resident = 1: Problem::certify_solution_resident, what this host's solveCORA does).
 * out: [0..2] is_certified, theta, iterations of the first; [3..5] of the second; x: the second's direction. */
int cora_problem_certify_chain(cora_problem *p, const double *Y, double eta, int nx, int resident, double out[6], double *x);

/* Test switches of the eigensolver stage of certify_solution (src/CORA_utils.cpp:129-167): run step 3 without the seed
 * that the failed factorisation yields and / or without the incomplete-LDL^T preconditioner; and whether the last
 * cora_problem_certify got as far as step 3 (the preconditioned stage). */
int cora_problem_set_verification_lab(cora_problem *p, int seed_negative_direction, int use_ildl);
int cora_problem_certification_reached_step3(cora_problem *p, int *reached);

/* The PSD test of fast_verification alone (src/CORA_utils.cpp:36-51), on the host, no device needed: LL^T of S + shift I
 * (S: CSR, n x n, the certificate matrix of a problem with d, n_poses, n_ranges, n_trans) in the elimination order
 * certify_solution uses.  out: [0] 1 when the factorisation succeeded, [1] the permuted index of the first non-positive
 * pivot (-1), [2] nnz(L). */
int cora_host_cholesky_test(int d, int n_poses, int n_ranges, int n_trans, int n, const int32_t *rowptr, const int32_t *colidx,
                            const double *vals, double shift, int leaf_poses, int64_t out[3]);

/* fast_verification(S, eta, X0) (src/CORA_utils.cpp:17-186) for an arbitrary symmetric sparse S
 * (CSR, n x n).  X0: n x nx or NULL for a random block.  out: [0] is_certified, [1] theta,
 * [2] iterations; x: n. */
int cora_host_fast_verification(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                double eta, const double *X0, int nx, int max_iters, double out[3], double *x);
/* The same with the start block handed over as two pieces (columns [0, split) and [split, nx)) that are put side by side
 * on the device -- the form certify_solution uses (previous eigenvectors | cached random columns); same numbers. */
int cora_host_fast_verification_pieces(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                       double eta, const double *X0, int nx, int split, int max_iters, double out[3],
                                       double *x);

/* The same with the knobs of step 3 (src/CORA_utils.cpp:129-167) exposed for the tests: opts = {max_fill_factor,
 * drop_tol, seed the block with the failed factorisation's direction (0/1), use the ILDL preconditioner (0/1)};
 * out[3] = 1 when step 3 ran. */
int cora_host_fast_verification_lab(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                    double eta, const double *X0, int nx, int max_iters, const double opts[4],
                                    double out[4], double *x);

/* The certification call of solveCORA's loop (src/CORA.cpp:156-170): first != 0 -- the point itself is the eigensolver's
 * bootstrap (first level); first == 0 -- it starts from the Ritz block the previous call left on the device (what this host's
 * solveCORA does instead of handing all_eigvecs through the host).  out / x as cora_problem_certify. */
int cora_problem_certify_resident(cora_problem *p, const double *Y, double eta, int nx, int first, double out[3], double *x);

/* saddleEscape (src/CORA.cpp:245-350): Y is N x (rank - 1), the problem's relaxation rank has been incremented by the
 * caller (cora_problem_set_rank) as solveCORA does before the call; theta, v: the certificate's curvature and direction.
 * y_out: N x rank.  info: [0] f at [Y 0], [1] f at y_out, [2] 1 when y_out differs from [Y 0] (a trial point was taken). */
int cora_problem_saddle_escape(cora_problem *p, const double *Y, double theta, const double *v, double grad_tol,
                               double pgrad_tol, double *y_out, double info[3]);

/* projectSolution (src/CORA.cpp:352-441): Y is N x rank (the problem's current rank), y_out N x d. */
int cora_problem_project_solution(cora_problem *p, const double *Y, double *y_out);

/* solveCORA (src/CORA.cpp:26-243) from x0 (N x rank): Riemannian staircase up to max_rank, final
 * projection to rank d and refinement.  x_out: N x d.  opts[0..4] as in cora_problem_tnt (may be NULL).
 * stats: [0] f, [1] |grad|, [2] certified (the returned, rounded and refined solution), [3] eta, [4] theta, [5] final rank,
 * [6] staircase levels, [7] Hessian-vector products, [8] seconds, [9] 1 when the staircase stopped because its last level
 * was certified (0: it reached max_rank), [10] the rank of that level. */
int cora_problem_solve(cora_problem *p, const double *x0, int max_rank, int verbose, const double *opts,
                       double *x_out, double stats[11]);

/* After a preconditioned call: [0] regularisation lambda used, [1] nnz(L), [2] elimination-tree height. */
int cora_problem_precond_info(cora_problem *p, double info[3]);

/* Host sparse Cholesky of (Q + shift I)[0:m,0:m] in the CORA nested-dissection order (the
 * CHOLMOD stand-in, cora_amd/csrc/host/sparse_cholesky.h); m = N or N-1.  Solves for the k
 * right-hand sides in B (m x k, ld m) in place.  info: [0] ok (0/1), [1] nnz(L), [2] height of
 * the elimination tree. */
int cora_problem_cholesky_solve(cora_problem *p, int m, double shift, int leaf_poses, double *B, int k,
                                int64_t info[3]);
/* Test hook for the host factorisation itself: info = {ok, nnz(L), first failing column (permuted) or -1}, digest = two
 * order-dependent sums over the finished columns of L (bit-equal runs give bit-equal digests), negative_direction
 * (N doubles, optional) = the direction of non-positive curvature when the factorisation fails.  The subtree-parallel
 * factorisation (CORA_CHOL_THREADS) must reproduce the one-thread result exactly. */
int cora_problem_cholesky_probe(cora_problem *p, int m, double shift, int leaf_poses, int64_t info[3], double *digest,
                                double *negative_direction);
/* The same on a copy of Q whose diagonal entries bump_rows[b] (original numbering) have bump_vals[b] added: puts the
 * first non-positive pivot where a test wants it (e.g. inside the group of trailing landmark rows). */
int cora_problem_cholesky_probe_bumped(cora_problem *p, int m, double shift, int leaf_poses, int nbump,
                                       const int32_t *bump_rows, const double *bump_vals, int64_t info[3],
                                       double *digest, double *negative_direction);

/* Test / timing hook without a GPU: factor of (Q + shift I)[0:N-1] and the device solve plan built from it (the host part
 * of the preconditioner's set-up; CORA_TRI_TIMING=1 prints its phases).  info = {stages, nnz(L), substitution blocks,
 * rows of the top stage}. */
int cora_problem_plan_probe(cora_problem *p, double shift, int leaf_poses, int64_t info[4]);

/* getBlockCholeskyFactorization + blockCholeskySolve (include/CORA/CORA_preconditioners.h:40-44) on the host:
 * A symmetric CSR n x n, block sizes summing to n, B rhs_rows x k column-major with rhs_rows = n or n + 1
 * (then the last row of X is zero). */
int cora_host_block_cholesky_solve(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int nblocks,
                                   const int32_t *block_sizes, int rhs_rows, int k, const double *B, double *X);

/* The reference's manifold classes (include/CORA/StiefelProduct.h, ObliqueManifold.h, MatrixManifold.h) on the
 * GPU.  kind 0: StiefelProduct(k, p, n), points p x kn; kind 1: ObliqueManifold(p, n), points p x n
 * (column-major).  op: "projectToManifold" (A), "projectToTangentSpace" (A = Y, B = V), "retract" (A = Y, B = V),
 * "random_sample" (seed), "innerProduct" (A, B -> out[0]), "SymBlockDiagProduct" (A; B holds B then C back to
 * back). */
int cora_host_manifold_op(int kind, int k, int p, int n, const char *op, const double *A, const double *B,
                          uint64_t seed, double *out);

/* Problem::printProblem (src/CORA_problem.cpp:400-489): registry and measurements on stdout. */
int cora_problem_print(cora_problem *p);

/* saveSolnToTum / saveSolnToG20 (src/CORA_utils.cpp:235-346) for a rank-d solution of the explicit problem
 * (N x d, e.g. the output of "alignEstimateToOrigin"): one line per pose of robot `robot_chr` (the symbol
 * character, as getPoseSymbols(chr); 0 = every pose in index order), time stamp = position in that list. */
int cora_problem_save_trajectory(cora_problem *p, const double *X, int g2o, int robot_chr, const char *path);

/* The device handle (cora_ctx*, include/cora_hip.h) behind the problem's operators. */
void *cora_problem_context(cora_problem *p);

#ifdef __cplusplus
}
#endif
#endif /* CORA_HOST_H_ */
