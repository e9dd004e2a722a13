// CORA::Problem -- host-side mirror of the reference's class
// (include/CORA/CORA_problem.h:67-416).  Ingestion, the variable registry and
// the data-matrix assembly run on the host (they define Q once per problem);
// every operator of the optimisation hot path is a call through the C ABI of
// include/cora_hip.h into HIP kernels.  There is no CPU implementation of the
// operators in this class: without a usable gfx950 device they throw.
#pragma once

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <set>
#include <utility>
#include <vector>

#include "CORA_types.h"
#include "sparse_cholesky.h"
#include "Measurements.h"
#include "Symbol.h"

struct cora_ctx;
// the injected communication steps of a partitioned handle (include/cora_hip.h, cora_set_comm)
extern "C" {
typedef int (*cora_exchange_fn)(void *user, double *dX, int ld);
typedef int (*cora_allreduce_fn)(void *user, double *vals, int n);
typedef int (*cora_allgather_fn)(void *user, double *dX, int ld);
}

namespace CORA {

/** include/CORA/CORA_problem.h:31-46 */
struct CoraDataSubmatrices {
  SparseMatrix range_incidence_matrix;
  SparseMatrix range_precision_matrix;
  SparseMatrix range_dist_matrix;
  SparseMatrix rel_pose_incidence_matrix;
  SparseMatrix rel_pose_translation_data_matrix;
  SparseMatrix rotation_conn_laplacian;
  SparseMatrix rel_pose_translation_precision_matrix;
  SparseMatrix rel_pose_rotation_precision_matrix;
};

class Problem {
 private:
  int dim_;
  bool pin_last_translation_ = true;
  int relaxation_rank_;

  std::map<Symbol, int> pose_symbol_idxs_;
  std::map<Symbol, int> landmark_symbol_idxs_;
  std::vector<RangeMeasurement> range_measurements_;
  std::vector<RelativePoseMeasurement> rel_pose_pose_measurements_;
  std::vector<RelativePoseLandmarkMeasurement> rel_pose_landmark_measurements_;
  Symbol origin_symbol_;
  std::vector<PosePrior> pose_priors_;
  std::vector<LandmarkPrior> landmark_priors_;
  // The reference finds duplicates with std::find over the measurement vectors
  // (src/CORA_problem.cpp:42,56,69: O(m^2)); same semantics with a hash of the
  // unordered symbol pair in an ordered set, so that 10^5-edge graphs load in seconds.
  std::set<std::pair<Key, Key>> range_pairs_, rpm_pairs_, rplm_pairs_;
  std::set<Key> pose_prior_ids_, landmark_prior_ids_;

  Formulation formulation_;
  Preconditioner preconditioner_;
  CoraDataSubmatrices data_submatrices_;
  bool has_priors_ = false;
  bool problem_data_up_to_date_ = false;
  int device_ = 0;
  int part_rank_ = 0, part_world_ = 1;
  cora_exchange_fn comm_exchange_ = nullptr;
  cora_allreduce_fn comm_allreduce_ = nullptr;
  cora_allgather_fn comm_allgather_ = nullptr;
  void *comm_user_ = nullptr;
  mutable std::shared_ptr<cora_ctx> ctx_;
  // the Ritz block of the last certify_solution_resident, on the device (declared after ctx_: released before the handle)
  mutable std::shared_ptr<class LOBPCGSolver> cert_block_;
  CertResults certifyImpl(const Matrix &Y, Scalar eta, size_t nx, const Matrix &eigvec_bootstrap, size_t max_LOBPCG_iters,
                          bool resident) const;
  mutable bool precond_ready_ = false;
  mutable bool implicit_ready_ = false;  // chol(Q33[0:nt-1]) installed on the handle
  mutable Scalar precond_lambda_ = 0;   // regularisation actually used
  mutable long precond_nnz_ = 0;         // nnz(L)
  mutable int precond_levels_ = 0;       // height of the elimination tree
  // symbolic analyses of this problem's factorisations (preconditioner block, certificate matrix): kept with the
  // Problem and gone with it (shared between copies of one Problem: they factorise the same patterns)
  mutable std::shared_ptr<SymbolicCache> symbolic_cache_ = std::make_shared<SymbolicCache>();
  // cert_* below is written by prepareCertification(), which solveCORA runs on a thread of its own beside the first TNT
  // solve, and read by certify_solution / get_certificate_matrix: every one of them holds this lock for its whole body
  // (round-4 advice; shared between copies like the cache above so that Problem stays copyable)
  mutable std::shared_ptr<std::recursive_mutex> cert_mutex_ = std::make_shared<std::recursive_mutex>();
  mutable std::vector<int32_t> cert_perm_;  // elimination order of the full certificate matrix (pattern-only: kept per data matrix)
  // S = Q - Lambda(Y) kept between certifications: its pattern is Q's plus the Lambda blocks, only the entries under
  // Lambda change with Y.  cert_lambda_pos_: position in cert_S_.values of every Lambda entry (pose-block entries row
  // by row, then the range diagonal); cert_lambda_q_: the position of the same entry in Q's values, -1 when Q has none
  mutable SparseMatrix cert_S_;
  mutable std::vector<int32_t> cert_lambda_pos_, cert_lambda_q_;
  mutable Matrix cert_random_;  // the random start columns of the eigensolver (fixed seed: the same numbers every time)
  // test switches of certify_solution's eigensolver stage (FastVerificationLab, CORA_utils.h) and what the last call did
  bool cert_lab_on_ = false, cert_lab_seed_ = true, cert_lab_ildl_ = true;
  mutable bool cert_reached_step3_ = false;
  const SparseMatrix &certificateMatrixCached(const Matrix &Y) const;
  const SparseMatrix &certificateMatrixFrom(const std::pair<Matrix, Vector> &lambda_blocks) const;

  void checkUpToDate() const;
  void addOriginPose();
  void fillRangeSubmatrices();
  void fillRelPoseSubmatrices();
  void fillRotConnLaplacian();
  void fillDataMatrix();
  void updatePreconditioner();
  Matrix dataMatrixProduct(const Matrix &Y) const;
  void ensureContext() const;
  void ensurePreconditioner() const;
  void fillImplicitFormulationMatrices() const;
  [[noreturn]] void throwLast(int status, const char *where) const;

 public:
  Problem(int dim, int relaxation_rank, Formulation formulation = Formulation::Explicit,
          Preconditioner preconditioner = Preconditioner::RegularizedCholesky);

  void addPoseVariable(const Symbol &pose_id);
  void addPoseVariable(const std::string &pose_id) { addPoseVariable(Symbol(pose_id)); }
  void addLandmarkVariable(const Symbol &landmark_id);
  void addLandmarkVariable(const std::string &landmark_id) { addLandmarkVariable(Symbol(landmark_id)); }
  void addRangeMeasurement(const RangeMeasurement &range_measurement);
  void addRelativePoseMeasurement(const RelativePoseMeasurement &rel_pose_measure);
  void addRelativePoseLandmarkMeasurement(const RelativePoseLandmarkMeasurement &m);
  void addPosePrior(const PosePrior &pose_prior);
  void addLandmarkPrior(const LandmarkPrior &landmark_prior);

  Index getRotationIdx(const Symbol &pose_symbol) const;
  Index getRangeIdx(const SymbolPair &range_symbol_pair) const;
  Index getTranslationIdx(const Symbol &trans_symbol) const;

  const CoraDataSubmatrices &getDataSubmatrices() {
    if (!problem_data_up_to_date_) updateProblemData();
    return data_submatrices_;
  }
  Symbol getOriginSymbol() const { return origin_symbol_; }
  std::map<Symbol, int> getPoseSymbolMap() const { return pose_symbol_idxs_; }
  /** Poses of one robot (symbols with character chr) in symbol order, src/CORA_problem.cpp:954-962. */
  std::vector<Symbol> getPoseSymbols(unsigned char chr) const {
    std::vector<Symbol> s;
    for (const auto &kv : pose_symbol_idxs_)
      if (kv.first.chr() == chr) s.push_back(kv.first);
    return s;
  }
  std::map<Symbol, int> getLandmarkSymbolMap() const { return landmark_symbol_idxs_; }
  const std::vector<RangeMeasurement> &getRangeMeasurements() const { return range_measurements_; }
  const std::vector<RelativePoseMeasurement> &getRPMs() const { return rel_pose_pose_measurements_; }

  SparseMatrix data_matrix_;

  void updateProblemData();
  const SparseMatrix &getDataMatrix();
  int getDataMatrixSize() const;
  int getExpectedVariableSize() const;

  Formulation getFormulation() const { return formulation_; }
  Preconditioner getPreconditioner() const { return preconditioner_; }
  int dim() const { return dim_; }
  int numPoses() const { return static_cast<int>(pose_symbol_idxs_.size()); }
  int numPosePoseMeasurements() const { return static_cast<int>(rel_pose_pose_measurements_.size()); }
  int numPoseLandmarkMeasurements() const { return static_cast<int>(rel_pose_landmark_measurements_.size()); }
  int numPosePriors() const { return static_cast<int>(pose_priors_.size()); }
  int numLandmarkPriors() const { return static_cast<int>(landmark_priors_.size()); }
  int getNumPosePriors() const { return numPosePriors(); }          // include/CORA/CORA_problem.h:231-232
  int getNumLandmarkPriors() const { return numLandmarkPriors(); }
  /** Human-readable dump of the registry and of every measurement (src/CORA_problem.cpp:400-489). */
  void printProblem() const;
  int numLandmarks() const { return static_cast<int>(landmark_symbol_idxs_.size()); }
  int numRangeMeasurements() const { return static_cast<int>(range_measurements_.size()); }
  int numTranslationalStates() const { return numPoses() + numLandmarks(); }
  int numPosesDim() const { return dim() * numPoses(); }
  int rotAndRangeMatrixSize() const { return numPosesDim() + numRangeMeasurements(); }

  /*****  Riemannian optimization functions (all on the GPU)  *******/
  size_t getRelaxationRank() const { return static_cast<size_t>(relaxation_rank_); }
  Matrix getRandomInitialGuess(uint64_t seed = 7) const;
  void incrementRank() { setRank(relaxation_rank_ + 1); }
  void setRank(int r);
  void setPreconditioner(Preconditioner p) {
    preconditioner_ = p;
    precond_ready_ = false;
  }
  void setFormulation(Formulation f) { formulation_ = f; }
  void setDevice(int device) { device_ = device; }
  // Multi-GPU (one process per GPU): this process owns partition `rank` of `world` of the rows of Q (pose-aligned,
  // nnz-balanced, include/cora_hip.h cora_ctx_create_part) and reaches the others through the library's own
  // communication (cora_comm_create_rccl / cora_comm_create_local on context(), after this call and after every
  // rebuild of the handle -- until then a collective step fails with CORA_ERR_NOT_READY) or through three injected
  // steps (cora_set_comm semantics; pass nullptr to install communication on the handle afterwards).  Every operator of
  // this class, TNT, LOBPCG, certify_solution and solveCORA then run on the partition, all ranks calling the same
  // sequence: the Lambda blocks of the certificate are gathered, every rank runs the PSD test (same decision), LOBPCG
  // runs on the sharded operator.  The Cholesky preconditioners become block Jacobi over the ranks with exact
  // blocks (every rank factorises the diagonal block of ITS rows of Q + lambda I: weaker than the reference's global
  // factor, more inner iterations).  Not sharded: the implicit formulation (use explicit) and the ILDL branch of
  // fast_verification (skipped).  Call before the first operator.
  SymbolicCache *symbolicCache() const { return symbolic_cache_.get(); }
  void setPartition(int rank, int world, cora_exchange_fn exchange, cora_allreduce_fn allreduce,
                    cora_allgather_fn allgather, void *user) {
    part_rank_ = rank;
    part_world_ = world;
    comm_exchange_ = exchange;
    comm_allreduce_ = allreduce;
    comm_allgather_ = allgather;
    comm_user_ = user;
    cert_block_.reset();
    ctx_.reset();
    precond_ready_ = false;
  }
  int partitionWorld() const { return part_world_; }

  Scalar evaluateObjective(const Matrix &Y) const;
  Matrix Euclidean_gradient(const Matrix &Y) const;
  Matrix Riemannian_gradient(const Matrix &Y) const;
  Matrix Riemannian_gradient(const Matrix &Y, const Matrix &NablaF_Y) const;
  Matrix Riemannian_Hessian_vector_product(const Matrix &Y, const Matrix &NablaF_Y, const Matrix &Ydot) const;
  Matrix tangent_space_projection(const Matrix &Y, const Matrix &Ydot) const;
  Matrix precondition(const Matrix &V) const;
  Matrix projectToManifold(const Matrix &A) const;
  Matrix retract(const Matrix &Y, const Matrix &V) const;

  /********** Certification **************/
  using LambdaBlocks = std::pair<Matrix, Vector>;
  LambdaBlocks compute_Lambda_blocks(const Matrix &Y) const;
  SparseMatrix compute_Lambda_from_Lambda_blocks(const LambdaBlocks &Lambda_blocks, const int &Lambda_size) const;
  SparseMatrix get_certificate_matrix(const Matrix &Y) const;
  /** Everything a first certification needs that does not depend on the point: the elimination order of the certificate
   * matrix, its pattern (Q's plus the Lambda blocks), the symbolic analysis of its factorisation with the factor's
   * storage touched once, and the random start columns of the eigensolver.  solveCORA runs it on a thread of its own
   * WHILE the first TNT solve keeps the device busy (none of it is read by TNT): at 10^5 poses 0.07 s of the first
   * certification's 0.12 s, at 10^6 poses 0.8 of 1.4 s.  Calling it is optional; certify_solution does whatever is missing. */
  void prepareCertification(Index num_eigvecs = 12) const;
  /** Test switches of the eigensolver stage of certify_solution (see FastVerificationLab): without the seed of the failed
   * factorisation and / or without the incomplete-LDL^T preconditioner; and whether the last call got as far as step 3. */
  void setVerificationLab(bool seed_negative_direction, bool use_ildl) {
    cert_lab_on_ = true;
    cert_lab_seed_ = seed_negative_direction;
    cert_lab_ildl_ = use_ildl;
  }
  bool lastCertificationReachedStep3() const { return cert_reached_step3_; }
  CertResults certify_solution(const Matrix &Y, Scalar eta, size_t nx, const Matrix &eigvec_bootstrap,
                               size_t max_LOBPCG_iters = 500) const;
  /** The same for a caller that only needs the decision, theta and the direction x (solveCORA): the eigensolver's Ritz
   * block stays on the device (all_eigvecs comes back EMPTY) and, handed an empty bootstrap, the next call starts from it
   * where it is -- no download and no upload of the N x 12 block per certification (9 + 2 ms at 10^5 poses, 0.1 s at 10^6). */
  CertResults certify_solution_resident(const Matrix &Y, Scalar eta, size_t nx, const Matrix &eigvec_bootstrap,
                                        size_t max_LOBPCG_iters = 500) const;

  /************** Utilities **********************/
  void checkVariablesAreValid(const Matrix &Y) const;
  Matrix alignEstimateToOrigin(const Matrix &Y) const;
  Matrix getTranslationExplicitSolution(const Matrix &Y) const;
  /** Host variable <-> device row count: identity when explicit; [M; 0] / leading rows when implicit. */
  const Matrix &lifted(const Matrix &M, Matrix &tmp) const;
  Matrix lowered(Matrix &&M) const;

  /** Device handle behind the operators (for the resident TNT / LOBPCG loops). */
  cora_ctx *context() const {
    ensureContext();
    return ctx_.get();
  }
  void ensurePreconditionerReady() const { ensurePreconditioner(); }
  Scalar preconditionerLambda() const { return precond_lambda_; }
  long preconditionerNnz() const { return precond_nnz_; }
  int preconditionerLevels() const { return precond_levels_; }
};

}  // namespace CORA
