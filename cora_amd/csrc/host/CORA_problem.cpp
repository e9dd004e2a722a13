// Host side of CORA::Problem: registry + data-matrix assembly on the CPU (one
// pass per problem), operators on the GPU through include/cora_hip.h.
#include "CORA_problem.h"
#include "LOBPCG.h"

#include <chrono>
#include <thread>
#include <cstdio>

#include <cmath>
#include <iostream>

#include <cstdlib>

#include "../../../include/cora_hip.h"
#include "dense.h"
#include "sparse_cholesky.h"
#include "../parallel.h"

namespace CORA {

namespace {
std::pair<Key, Key> unordered_pair(const Symbol &a, const Symbol &b) {
  const Key x = a.key(), y = b.key();
  return x < y ? std::make_pair(x, y) : std::make_pair(y, x);
}
}  // namespace

Problem::Problem(int dim, int relaxation_rank, Formulation formulation, Preconditioner preconditioner)
    : dim_(dim),
      relaxation_rank_(relaxation_rank),
      origin_symbol_(Symbol("O0")),
      formulation_(formulation),
      preconditioner_(preconditioner) {
  if (relaxation_rank < dim) throw std::invalid_argument("relaxation rank must be >= dim");
}

// ---- registry: src/CORA_problem.cpp:24-113 ---------------------------------
void Problem::addPoseVariable(const Symbol &pose_id) {
  if (pose_symbol_idxs_.find(pose_id) != pose_symbol_idxs_.end())
    throw std::invalid_argument("Pose variable already exists");
  pose_symbol_idxs_.insert(std::make_pair(pose_id, static_cast<int>(pose_symbol_idxs_.size())));
  problem_data_up_to_date_ = false;
}

void Problem::addLandmarkVariable(const Symbol &landmark_id) {
  if (landmark_symbol_idxs_.find(landmark_id) != landmark_symbol_idxs_.end())
    throw std::invalid_argument("Landmark variable already exists");
  landmark_symbol_idxs_.insert(std::make_pair(landmark_id, static_cast<int>(landmark_symbol_idxs_.size())));
  problem_data_up_to_date_ = false;
}

void Problem::addRangeMeasurement(const RangeMeasurement &m) {
  if (!range_pairs_.insert(unordered_pair(m.first_id, m.second_id)).second) {
    std::cout << "Found duplicate measure: " << m.first_id.string() << " -> " << m.second_id.string()
              << std::endl;
    throw std::invalid_argument("Range measurement already exists");
  }
  range_measurements_.push_back(m);
  problem_data_up_to_date_ = false;
}

void Problem::addRelativePoseMeasurement(const RelativePoseMeasurement &m) {
  if (!rpm_pairs_.insert(unordered_pair(m.first_id, m.second_id)).second)
    throw std::invalid_argument("Relative pose measurement already exists: " + m.first_id.string() +
                                " -> " + m.second_id.string());
  rel_pose_pose_measurements_.push_back(m);
  problem_data_up_to_date_ = false;
}

void Problem::addRelativePoseLandmarkMeasurement(const RelativePoseLandmarkMeasurement &m) {
  if (!rplm_pairs_.insert(unordered_pair(m.first_id, m.second_id)).second)
    throw std::invalid_argument("Relative pose landmark measurement already exists");
  rel_pose_landmark_measurements_.push_back(m);
  problem_data_up_to_date_ = false;
}

void Problem::addOriginPose() {
  std::cout << "WARNING - using symbol " << origin_symbol_.string()
            << " to make an 'origin'. Could cause name collision." << std::endl;
  addPoseVariable(origin_symbol_);
}

void Problem::addPosePrior(const PosePrior &pose_prior) {
  if (!pose_prior_ids_.insert(pose_prior.id.key()).second)
    throw std::invalid_argument("Pose prior already exists");
  pose_priors_.push_back(pose_prior);
  problem_data_up_to_date_ = false;
  if (!has_priors_) {
    has_priors_ = true;
    addOriginPose();
  }
}

void Problem::addLandmarkPrior(const LandmarkPrior &landmark_prior) {
  if (!landmark_prior_ids_.insert(landmark_prior.id.key()).second)
    throw std::invalid_argument("Landmark prior already exists");
  landmark_priors_.push_back(landmark_prior);
  problem_data_up_to_date_ = false;
  if (!has_priors_) {
    has_priors_ = true;
    addOriginPose();
  }
}

// ---- index helpers: src/CORA_problem.cpp:964-1021 --------------------------
Index Problem::getRotationIdx(const Symbol &pose_symbol) const {
  auto it = pose_symbol_idxs_.find(pose_symbol);
  if (it != pose_symbol_idxs_.end()) return it->second;
  throw std::invalid_argument("Unknown pose symbol: " + pose_symbol.string());
}

Index Problem::getRangeIdx(const SymbolPair &p) const {
  for (size_t i = 0; i < range_measurements_.size(); ++i)
    if (range_measurements_[i].hasSymbolPair(p)) return static_cast<Index>(i) + numPosesDim();
  throw std::invalid_argument("Unknown range symbol");
}

Index Problem::getTranslationIdx(const Symbol &s) const {
  const Index off = rotAndRangeMatrixSize();
  auto pit = pose_symbol_idxs_.find(s);
  if (pit != pose_symbol_idxs_.end()) return pit->second + off;
  auto lit = landmark_symbol_idxs_.find(s);
  if (lit != landmark_symbol_idxs_.end()) return lit->second + off + numPoses();
  throw std::invalid_argument("Unknown translation symbol");
}

// ---- sub-matrices: src/CORA_problem.cpp:115-377 ----------------------------
static SparseMatrix diagonal(const std::vector<Scalar> &d) {
  SparseMatrix m(static_cast<Index>(d.size()), static_cast<Index>(d.size()));
  std::vector<Triplet> t;
  for (size_t i = 0; i < d.size(); ++i) t.push_back({static_cast<Index>(i), static_cast<Index>(i), d[i]});
  m.setFromTriplets(std::move(t));
  return m;
}

void Problem::fillRangeSubmatrices() {
  const Index off = rotAndRangeMatrixSize();
  const Index r = numRangeMeasurements(), nt = numTranslationalStates();
  std::vector<Scalar> dist(r), prec(r);
  std::vector<Triplet> inc;
  for (Index k = 0; k < r; ++k) {
    const RangeMeasurement &m = range_measurements_[k];
    dist[k] = m.r;
    prec[k] = m.getPrecision();
    inc.push_back({k, getTranslationIdx(m.first_id) - off, -1.0});
    inc.push_back({k, getTranslationIdx(m.second_id) - off, 1.0});
  }
  data_submatrices_.range_incidence_matrix = SparseMatrix(r, nt);
  data_submatrices_.range_incidence_matrix.setFromTriplets(std::move(inc));
  data_submatrices_.range_dist_matrix = diagonal(dist);
  data_submatrices_.range_precision_matrix = diagonal(prec);
}

void Problem::fillRelPoseSubmatrices() {
  fillRotConnLaplacian();
  const Index npp = numPosePoseMeasurements(), npl = numPoseLandmarkMeasurements();
  const Index nprior = numPosePriors(), nlp = numLandmarkPriors();
  const Index m = npp + nprior + npl + nlp;
  const Index nt = numTranslationalStates(), off = rotAndRangeMatrixSize();
  std::vector<Scalar> tprec(m), rprec(npp + nprior);
  std::vector<Triplet> inc, tdata;
  Index row = 0;
  auto add = [&](const Symbol &a, const Symbol &b, const Vector &t, Scalar tau) {
    tprec[row] = tau;
    const Index id1 = getTranslationIdx(a) - off, id2 = getTranslationIdx(b) - off;
    inc.push_back({row, id1, -1.0});
    inc.push_back({row, id2, 1.0});
    for (int k = 0; k < dim_; ++k) tdata.push_back({row, id1 * dim_ + k, -t(k)});
    ++row;
  };
  // row order: pose-pose, pose priors, pose-landmark, landmark priors (:190-294)
  for (const auto &rpm : rel_pose_pose_measurements_) {
    rprec[row] = rpm.getRotPrecision();
    add(rpm.first_id, rpm.second_id, rpm.t, rpm.getTransPrecision());
  }
  for (const auto &pp : pose_priors_) {
    rprec[row] = pp.getRotPrecision();
    add(origin_symbol_, pp.id, pp.t, pp.getTransPrecision());
  }
  for (const auto &pl : rel_pose_landmark_measurements_) add(pl.first_id, pl.second_id, pl.t, pl.getTransPrecision());
  for (const auto &lp : landmark_priors_) add(origin_symbol_, lp.id, lp.p, lp.getTransPrecision());

  data_submatrices_.rel_pose_incidence_matrix = SparseMatrix(m, nt);
  data_submatrices_.rel_pose_incidence_matrix.setFromTriplets(std::move(inc));
  data_submatrices_.rel_pose_translation_data_matrix = SparseMatrix(m, numPosesDim());
  data_submatrices_.rel_pose_translation_data_matrix.setFromTriplets(std::move(tdata));
  data_submatrices_.rel_pose_translation_precision_matrix = diagonal(tprec);
  data_submatrices_.rel_pose_rotation_precision_matrix = diagonal(rprec);
}

void Problem::fillRotConnLaplacian() {
  const Index d = dim_;
  std::vector<Triplet> t;
  t.reserve(static_cast<size_t>(2 * (d + d * d)) * (rel_pose_pose_measurements_.size() + pose_priors_.size()));
  auto add = [&](Index i, Index j, Scalar kappa, const Matrix &R) {
    for (Index k = 0; k < d; ++k) t.push_back({d * i + k, d * i + k, kappa});
    for (Index k = 0; k < d; ++k) t.push_back({d * j + k, d * j + k, kappa});
    for (Index r = 0; r < d; ++r)
      for (Index c = 0; c < d; ++c) t.push_back({i * d + r, j * d + c, -kappa * R(r, c)});
    for (Index r = 0; r < d; ++r)
      for (Index c = 0; c < d; ++c) t.push_back({j * d + r, i * d + c, -kappa * R(c, r)});
  };
  for (const auto &m : rel_pose_pose_measurements_)
    add(getRotationIdx(m.first_id), getRotationIdx(m.second_id), m.getRotPrecision(), m.R);
  for (const auto &p : pose_priors_)
    add(getRotationIdx(origin_symbol_), getRotationIdx(p.id), p.getRotPrecision(), p.R);
  data_submatrices_.rotation_conn_laplacian = SparseMatrix(numPosesDim(), numPosesDim());
  data_submatrices_.rotation_conn_laplacian.setFromTriplets(std::move(t));
}

// ---- data matrix: src/CORA_problem.cpp:625-712 ------------------------------
void Problem::fillDataMatrix() {
  const auto &S = data_submatrices_;
  std::vector<Scalar> wt(static_cast<size_t>(S.rel_pose_translation_precision_matrix.rows()));
  for (size_t i = 0; i < wt.size(); ++i) wt[i] = S.rel_pose_translation_precision_matrix.values[i];
  std::vector<Scalar> wr(static_cast<size_t>(S.range_precision_matrix.rows()));
  for (size_t i = 0; i < wr.size(); ++i) wr[i] = S.range_precision_matrix.values[i];

  const SparseMatrix Tt = S.rel_pose_translation_data_matrix.transpose();
  const SparseMatrix Att = S.rel_pose_incidence_matrix.transpose();
  const SparseMatrix Art = S.range_incidence_matrix.transpose();
  const SparseMatrix Q11 = S.rotation_conn_laplacian.plus(Tt.times(&wt, S.rel_pose_translation_data_matrix));
  const SparseMatrix Q13 = Tt.times(&wt, S.rel_pose_incidence_matrix);
  const SparseMatrix OmegaRD = S.range_precision_matrix.times(nullptr, S.range_dist_matrix);
  const SparseMatrix Q22 = OmegaRD.times(nullptr, S.range_dist_matrix);
  const SparseMatrix Q23 = OmegaRD.times(nullptr, S.range_incidence_matrix);
  const SparseMatrix Q33 =
      Att.times(&wt, S.rel_pose_incidence_matrix).plus(Art.times(&wr, S.range_incidence_matrix));

  const Index rot = numPosesDim(), rr = rotAndRangeMatrixSize();
  std::vector<Triplet> all;
  all.reserve(static_cast<size_t>(Q11.nonZeros() + 2 * Q13.nonZeros() + Q22.nonZeros() + 2 * Q23.nonZeros() + Q33.nonZeros()));
  auto append = [&all](const std::vector<Triplet> &t) { all.insert(all.end(), t.begin(), t.end()); };
  append(Q11.triplets(0, 0));
  append(Q13.triplets(0, rr));
  append(Q22.triplets(rot, rot));
  append(Q23.triplets(rot, rr));
  append(Q33.triplets(rr, rr));
  append(Q13.triplets(rr, 0, true));
  append(Q23.triplets(rr, rot, true));
  data_matrix_ = SparseMatrix(getDataMatrixSize(), getDataMatrixSize());
  data_matrix_.setFromTriplets(std::move(all));
}

void Problem::updateProblemData() {  // src/CORA_problem.cpp:500-510
  const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "  [update] %-26s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  fillRangeSubmatrices();
  tick("range submatrices");
  fillRelPoseSubmatrices();
  tick("relative-pose submatrices");
  fillDataMatrix();
  tick("data matrix");
  cert_block_.reset();  // (its device memory belongs to the handle)
  ctx_.reset();  // the device copy of Q is rebuilt lazily
  precond_ready_ = false;
  std::lock_guard<std::recursive_mutex> lock(*cert_mutex_);
  cert_perm_.clear();
  cert_S_ = SparseMatrix();
  cert_lambda_pos_.clear();
  cert_lambda_q_.clear();
  problem_data_up_to_date_ = true;
}

const SparseMatrix &Problem::getDataMatrix() {
  if (data_matrix_.nonZeros() == 0 || !problem_data_up_to_date_) updateProblemData();
  return data_matrix_;
}

int Problem::getDataMatrixSize() const { return numPoses() * (dim_ + 1) + numLandmarks() + numRangeMeasurements(); }
int Problem::getExpectedVariableSize() const {  // src/CORA_problem.cpp:944-952
  return formulation_ == Formulation::Implicit ? rotAndRangeMatrixSize() : getDataMatrixSize();
}

// Device vectors always have getDataMatrixSize() rows; in the implicit formulation the host-side
// variable is the leading rotAndRangeMatrixSize() rows and the translation rows are zero.
const Matrix &Problem::lifted(const Matrix &M, Matrix &tmp) const {
  if (formulation_ != Formulation::Implicit) return M;
  tmp = Matrix::Zero(getDataMatrixSize(), M.cols());
  tmp.setBlock(0, 0, M);
  return tmp;
}
Matrix Problem::lowered(Matrix &&M) const {
  if (formulation_ != Formulation::Implicit) return std::move(M);
  return M.block(0, 0, rotAndRangeMatrixSize(), M.cols());
}

void Problem::checkUpToDate() const {
  if (!problem_data_up_to_date_)
    throw std::runtime_error(
        "The data matrix must be constructed before the objective function can be evaluated. This error "
        "may be due to the fact that data has been modified since the last call to updateProblemData()");
}

// ---- device plumbing ---------------------------------------------------------
void Problem::throwLast(int status, const char *where) const {
  const std::string msg = std::string(where) + ": " + cora_last_error(ctx_.get());
  if (status == CORA_ERR_SHAPE) throw std::logic_error(msg);  // MatrixShapeException's base
  if (status == CORA_ERR_ARG) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}

void Problem::ensureContext() const {
  checkUpToDate();
  if (!ctx_) {
    cora_ctx *c = nullptr;
    const int rc = cora_ctx_create_part(device_, dim_, numPoses(), numRangeMeasurements(), numTranslationalStates(),
                                        data_matrix_.outerIndexPtr(), data_matrix_.innerIndexPtr(),
                                        data_matrix_.valuePtr(), part_rank_, part_world_, &c);
    if (rc != CORA_OK)
      throw std::runtime_error(std::string("CORA::Problem: cannot create the device problem: ") +
                               cora_last_error(nullptr));
    ctx_ = std::shared_ptr<cora_ctx>(c, [](cora_ctx *p) { cora_ctx_destroy(p); });
    if (part_world_ > 1) {
      cora_set_comm(c, comm_exchange_, comm_allreduce_, comm_allgather_, comm_user_);
      cora_require_comm(c, 1);  // no callbacks (they are installed on the handle afterwards): fail loudly until then
    }
    implicit_ready_ = false;
  }
  if (formulation_ == Formulation::Implicit && !implicit_ready_) fillImplicitFormulationMatrices();
  {
    const int rc = cora_set_formulation(ctx_.get(), formulation_ == Formulation::Implicit ? 1 : 0);
    if (rc != CORA_OK) throwLast(rc, "Problem::setFormulation");
  }
  if (cora_get_rank(ctx_.get()) != relaxation_rank_) {
    const int rc = cora_set_rank(ctx_.get(), relaxation_rank_);
    if (rc != CORA_OK) throwLast(rc, "Problem::setRank");
    // the point / preconditioner state of the handle is rank specific
  }
}

void Problem::updatePreconditioner() { ensurePreconditioner(); }

// src/CORA_problem.cpp:714-740.  The reference keeps Qmain, [Q13; Q23] without its last column and
// chol(Q33[0:nt-1, 0:nt-1]) on the host; here Q is already resident in full, so only the factor of
// the reduced translation block is computed (host sparse Cholesky) and handed to the device.
void Problem::fillImplicitFormulationMatrices() const {
  if (formulation_ != Formulation::Implicit)
    throw std::invalid_argument(
        "Implicit formulation matrices should only be filled when the problem is in implicit formulation mode");
  const Index tb = rotAndRangeMatrixSize(), nt = numTranslationalStates(), m = nt - 1;
  if (m < 1) throw std::invalid_argument("the implicit formulation needs at least two translational states");
  std::vector<Triplet> t;
  for (Index i = 0; i < m; ++i)
    for (int32_t q = data_matrix_.outer[tb + i]; q < data_matrix_.outer[tb + i + 1]; ++q) {
      const Index j = data_matrix_.inner[q] - tb;
      if (j >= 0 && j < m) t.push_back({i, j, data_matrix_.values[q]});
    }
  SparseMatrix M(m, m);
  M.setFromTriplets(std::move(t));
  // same nested-dissection order as the preconditioner, restricted to the translations
  int leaf = 8;
  if (const char *env = std::getenv("CORA_ND_LEAF")) leaf = std::max(1, std::atoi(env));
  SparseMatrix none(nt, nt);
  const auto perm = coraOrdering(0, numPoses(), 0, static_cast<int>(nt), none, static_cast<int>(m), leaf);
  const CholeskyFactor F = choleskyFactor(M, static_cast<int>(m), 0.0, perm, symbolic_cache_.get());
  if (!F.ok)
    throw std::runtime_error("Problem::fillImplicitFormulationMatrices: the reduced translation block of Q is "
                             "not positive definite (disconnected measurement graph?)");
  const int rc = cora_implicit_set_cholesky(ctx_.get(), static_cast<int>(m), F.Lp.data(), F.Li.data(), F.Lx.data(),
                                            F.perm.data());
  if (rc != CORA_OK) throwLast(rc, "Problem::fillImplicitFormulationMatrices");
  implicit_ready_ = true;
}

// ||Q||_2 = lambda_max(Q) = -lambda_min(-Q), estimated exactly as the reference does (src/CORA_problem.cpp:556-578):
// LOBPCG on the operator X -> -(Q X), block min(4, N), one wanted pair, at most 100 iterations, tolerance 1e-2.  The
// blocks stay on the device (LOBPCG.h), the operator is the device SpMM, so lambda_reg = ||Q||_2 / (kappa_max - 1) -- and
// with it every STPCG iteration count -- is what the reference's call yields to the eigensolver's tolerance.  (Rounds 1-4
// used a power iteration stopped at 1e-3: same converged results, a lambda_reg that differed in its third digit.)
// Collective on partitioned handles (the Gram matrices are all-reduced), same number on every rank.
static Scalar spectralNormEstimate(cora_ctx *c, Index N) {
  const Index block = std::min<Index>(4, N);
  DeviceOperator negQ = [c](const double *dX, int k, double *dOut) {
    if (cora_spmm_dev(c, dX, k, dOut) != CORA_OK || cora_axpby_cols_dev(c, k, 0.0, dOut, -1.0, dOut) != CORA_OK)
      throw std::runtime_error(std::string("spectral norm estimate: ") + cora_last_error(c));
  };
  // the random start block is drawn on the device (cora_fill_random_dev: the reference draws Matrix::Random on the host;
  // which numbers they are does not matter to an estimate with tolerance 1e-2, and 450 k x 4 of them cost 3 ms to draw and
  // 6 ms to upload).  Only the Ritz value is wanted: the block stays on the device and goes with the solver.
  struct Start {
    cora_ctx *c;
    double *d = nullptr;
    ~Start() { if (d) cora_dev_free(c, d); }
  } start{c};
  if (cora_dev_alloc(c, static_cast<int>(block), &start.d) != CORA_OK ||
      cora_fill_random_dev(c, static_cast<int>(block), 12345ull, start.d) != CORA_OK)
    throw std::runtime_error(std::string("spectral norm estimate: ") + cora_last_error(c));
  LOBPCGSolver solver(c, static_cast<int>(N));
  const LOBPCGResult r = solver.run(negQ, std::nullopt, {HostColumns{nullptr, static_cast<int>(block), start.d, static_cast<int>(block)}},
                                    /*nev=*/1, /*max_iters=*/100, /*tau=*/1e-2, std::nullopt, /*download=*/false);
  return -r.Theta(0);
}

void Problem::ensurePreconditioner() const {  // src/CORA_problem.cpp:512-623
  if (ctx_ && precond_ready_) {
    ensureContext();
    return;
  }
  checkUpToDate();
  int kind;
  switch (preconditioner_) {
    case Preconditioner::None: kind = CORA_PRECOND_NONE; break;
    case Preconditioner::Jacobi: kind = CORA_PRECOND_JACOBI; break;
    case Preconditioner::BlockCholesky: kind = CORA_PRECOND_BLOCK_CHOLESKY; break;
    default: kind = CORA_PRECOND_REGULARIZED_CHOLESKY; break;
  }
  if (kind == CORA_PRECOND_BLOCK_CHOLESKY || kind == CORA_PRECOND_REGULARIZED_CHOLESKY) {
    const Index N = getDataMatrixSize();
    const int m = static_cast<int>(pin_last_translation_ ? N - 1 : N);
    // poses per nested-dissection leaf: 8 / 4 / 2 / 1 give an STPCG iteration of 158 / 157 / 152 / 152 us at 10^5 poses
    // (fewer substitution levels per block; nnz(L) 4.99 / 4.82 / 4.82 / 4.8 M)
    int leaf = 2;
    if (const char *env = std::getenv("CORA_ND_LEAF")) leaf = std::max(1, std::atoi(env));
    const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
    auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
      const auto now = std::chrono::steady_clock::now();
      if (timing) std::fprintf(stderr, "  [precond] %-26s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
      t_prev = now;
    };
    // Partitioned handle: the exact solve is a sequential recurrence over the whole chain and does not shard, so the
    // preconditioner becomes BLOCK JACOBI OVER THE RANKS -- every rank factorises the diagonal block of its own rows of
    // (Q + lambda I) and applies it to its own rows with the same device solve plan.  The rows a rank owns,
    // in API order, are the rotation rows of its poses | its range rows | its translations: the block is the data
    // matrix of a smaller problem of the same shape, so the ordering, the factorisation and the plan are the ones above.
    const bool sharded = part_world_ > 1;
    if (sharded) ensureContext();  // (the row map below is the handle's)
    std::vector<int32_t> own;      // API rows of this rank, ascending (sharded)
    std::vector<int32_t> to_local; // API row -> index in `own`, -1 elsewhere
    int n_loc = numPoses(), r_loc = static_cast<int>(numRangeMeasurements()), nt_loc = static_cast<int>(numTranslationalStates());
    int m_fac = m;
    auto local_block = [&](const SparseMatrix &A) {  // A[own, own] in local numbering
      std::vector<Triplet> t;
      for (size_t k = 0; k < own.size(); ++k)
        for (int32_t q = A.outer[own[k]]; q < A.outer[own[k] + 1]; ++q) {
          const int32_t j = to_local[A.inner[q]];
          if (j >= 0) t.push_back({static_cast<Index>(k), static_cast<Index>(j), A.values[q]});
        }
      SparseMatrix B(static_cast<Index>(own.size()), static_cast<Index>(own.size()));
      B.setFromTriplets(std::move(t));
      return B;
    };
    if (sharded) {
      std::vector<int32_t> map(static_cast<size_t>(N));
      if (cora_row_map(ctx_.get(), map.data()) != CORA_OK) throwLast(CORA_ERR_ARG, "Problem::updatePreconditioner");
      const int64_t lo = cora_shard_begin(ctx_.get()), hi = lo + cora_shard_rows(ctx_.get());
      to_local.assign(static_cast<size_t>(N), -1);
      const Index b1 = numPosesDim(), b2 = rotAndRangeMatrixSize(), b3 = b2 + numPoses();
      n_loc = r_loc = nt_loc = 0;
      for (Index i = 0; i < N; ++i)
        if (map[i] >= lo && map[i] < hi) {
          to_local[i] = static_cast<int32_t>(own.size());
          own.push_back(static_cast<int32_t>(i));
          if (i < b1) n_loc += (i % dim_ == 0);
          else if (i < b2) ++r_loc;
          else ++nt_loc;
          (void)b3;
        }
      const bool owns_pin = pin_last_translation_ && to_local[N - 1] >= 0;  // the pinned variable is this rank's last row
      m_fac = static_cast<int>(own.size()) - (owns_pin ? 1 : 0);
    }
    // the elimination order (host) is worked out while the device handle is created (when this call is its first use: the
    // format of Q and its uploads) and while the device estimates ||Q||_2 below: order 10 ms + symbolic analysis 16 ms +
    // first touch of the factor's storage 10 ms at 10^5 poses, against 22 ms for the estimate alone
    std::vector<int32_t> perm;
    std::thread ordering;
    std::exception_ptr ordering_error;
    if (!sharded)
      ordering = std::thread([&] {
        try {
          const auto t_ord = std::chrono::steady_clock::now();
          // (the factor's storage is touched beside the order and the analysis, from a guess: nnz(L) is 1.0-1.9 x nnz(Q) on the
          // chain-ordered graphs of this problem class)
          std::thread reserve;
          if (kind == CORA_PRECOND_REGULARIZED_CHOLESKY)
            reserve = std::thread([n = static_cast<size_t>(data_matrix_.nonZeros())] { choleskyReserveStorage(2 * n); });
          struct JoinReserve {
            std::thread &t;
            ~JoinReserve() { if (t.joinable()) t.join(); }
          } join_reserve{reserve};
          perm = coraOrdering(dim_, numPoses(), numRangeMeasurements(), numTranslationalStates(), data_matrix_, m, leaf);
          if (timing) std::fprintf(stderr, "  [precond] elimination order (thread) %.4f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ord).count());
          // ... and so is everything of the factorisation that the regularisation does not decide: the symbolic analysis
          // (Q + lambda I has Q's pattern) and the first touch of the factor's storage (18 + 10 ms at 10^5 poses, behind
          // the 20-28 ms of the norm estimate)
          if (kind == CORA_PRECOND_REGULARIZED_CHOLESKY) choleskyAnalyze(data_matrix_, m, perm, symbolic_cache_.get());
          if (timing) std::fprintf(stderr, "  [precond] order + analysis + storage (thread) %.4f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ord).count());
        } catch (...) {
          ordering_error = std::current_exception();
        }
      });
    struct Joiner {  // (an exception below must not leave the thread joinable)
      std::thread &t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{ordering};
    auto wait_for_order = [&] {
      if (ordering.joinable()) ordering.join();
      if (ordering_error) std::rethrow_exception(ordering_error);
    };
    ensureContext();
    // factorisation of the whole matrix, or of this rank's diagonal block (F.perm then holds API rows again)
    auto factorise = [&](const SparseMatrix &A, double shift) {
      wait_for_order();
      if (!sharded) return choleskyFactor(A, m, shift, perm, symbolic_cache_.get());
      const SparseMatrix B = local_block(A);
      const auto lperm = coraOrdering(dim_, n_loc, r_loc, nt_loc, B, m_fac, leaf);
      CholeskyFactor Fl = choleskyFactor(B, m_fac, shift, lperm, symbolic_cache_.get());
      for (int32_t &q : Fl.perm) q = own[static_cast<size_t>(q)];
      return Fl;
    };
    CholeskyFactor F;
    if (kind == CORA_PRECOND_REGULARIZED_CHOLESKY) {
      // lambda_reg = ||Q||_2 / (kappa_max - 1), kappa_max = 1e6 or CORA_REG_CHOLESKY_MAX_COND (:581-591)
      // ||D||_2 of the full data matrix in either formulation (:556-578 works on data_matrix_)
      cora_set_formulation(ctx_.get(), 0);
      const Scalar Dnorm = spectralNormEstimate(ctx_.get(), N);
      cora_set_formulation(ctx_.get(), formulation_ == Formulation::Implicit ? 1 : 0);
      Scalar max_cond = 1e6;
      if (const char *env = std::getenv("CORA_REG_CHOLESKY_MAX_COND")) {
        max_cond = std::stod(env);
        std::cout << "Loaded CORA_REG_CHOLESKY_MAX_COND from environment variable: " << max_cond << std::endl;
      }
      precond_lambda_ = Dnorm / (max_cond - 1);
      tick("spectral norm (device)");
      F = factorise(data_matrix_, precond_lambda_);
      tick("factorisation");
    } else {
      // documented semantics (include/CORA/CORA_preconditioners.h:28-44): independent factors of the
      // diagonal blocks (rotations | ranges | translations) of Q + 1e-3 I (:513-543).  The reference's
      // own loop never advances block_start (src/CORA_preconditioners.cpp:30-41); not reproduced.
      const Index b1 = numPosesDim(), b2 = rotAndRangeMatrixSize();
      auto blk = [&](Index i) { return i < b1 ? 0 : (i < b2 ? 1 : 2); };
      std::vector<Triplet> t;
      for (Index i = 0; i < N; ++i)
        for (int32_t q = data_matrix_.outer[i]; q < data_matrix_.outer[i + 1]; ++q)
          if (blk(i) == blk(data_matrix_.inner[q])) t.push_back({i, data_matrix_.inner[q], data_matrix_.values[q]});
      SparseMatrix D(N, N);
      D.setFromTriplets(std::move(t));
      precond_lambda_ = 1e-3;
      F = factorise(D, 1e-3);
    }
    if (!F.ok) throw std::runtime_error("Problem::updatePreconditioner: regularised data matrix is not positive definite");
    precond_nnz_ = static_cast<long>(F.nnz());
    {
      std::vector<int> depth(static_cast<size_t>(F.n), 1);
      int h = 0;
      for (int i = 0; i < F.n; ++i) {
        if (F.parent[i] >= 0) depth[F.parent[i]] = std::max(depth[F.parent[i]], depth[i] + 1);
        h = std::max(h, depth[i]);
      }
      precond_levels_ = h;
    }
    tick("tree height");
    const int rc = cora_precond_set_cholesky(ctx_.get(), sharded ? m_fac : m, F.Lp.data(), F.Li.data(), F.Lx.data(), F.perm.data());
    if (rc != CORA_OK) throwLast(rc, "Problem::updatePreconditioner");
    tick("solve plan + upload");
  }
  ensureContext();
  const int rc = cora_precond_setup(ctx_.get(), kind);
  if (rc != CORA_OK) throwLast(rc, "Problem::updatePreconditioner");
  precond_ready_ = true;
}

void Problem::setRank(int r) {
  if (r < dim_) throw std::invalid_argument("relaxation rank must be >= dim");
  relaxation_rank_ = r;
}

#define CORA_CALL(expr, where)                 \
  do {                                         \
    const int rc__ = (expr);                   \
    if (rc__ != CORA_OK) throwLast(rc__, where); \
  } while (0)

// ---- operators (GPU): src/CORA_problem.cpp:742-938 --------------------------
// LIFT(M) is M itself in the explicit formulation and [M; 0] in the implicit one (see lifted()).
#define LIFT(M, name) \
  Matrix name##_tmp;  \
  const Matrix &name = lifted(M, name##_tmp)

Matrix Problem::dataMatrixProduct(const Matrix &Y) const {
  checkMatrixShape("Problem::dataMatrixProduct::Y", getExpectedVariableSize(), Y.cols(), Y.rows(), Y.cols());
  ensureContext();
  LIFT(Y, Yl);
  Matrix out(Yl.rows(), Yl.cols());
  CORA_CALL(cora_data_matrix_product(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), static_cast<int>(Yl.cols()),
                                     out.data(), static_cast<int>(out.rows())),
            "Problem::dataMatrixProduct");
  return lowered(std::move(out));
}

Scalar Problem::evaluateObjective(const Matrix &Y) const {
  checkUpToDate();
  checkMatrixShape("Problem::evaluateObjective::Y", getExpectedVariableSize(), relaxation_rank_, Y.rows(), Y.cols());
  ensureContext();
  LIFT(Y, Yl);
  Scalar f = 0;
  CORA_CALL(cora_evaluate_objective(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), &f), "Problem::evaluateObjective");
  return f;
}

Matrix Problem::Euclidean_gradient(const Matrix &Y) const {
  checkUpToDate();
  checkMatrixShape("Problem::Euclidean_gradient::Y", getExpectedVariableSize(), relaxation_rank_, Y.rows(), Y.cols());
  ensureContext();
  LIFT(Y, Yl);
  Matrix out(Yl.rows(), Yl.cols());
  CORA_CALL(cora_euclidean_gradient(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), out.data(),
                                    static_cast<int>(out.rows())),
            "Problem::Euclidean_gradient");
  return lowered(std::move(out));
}

Matrix Problem::Riemannian_gradient(const Matrix &Y) const {
  checkUpToDate();
  checkMatrixShape("Problem::Riemannian_gradient::Y", getExpectedVariableSize(), relaxation_rank_, Y.rows(), Y.cols());
  ensureContext();
  LIFT(Y, Yl);
  Matrix out(Yl.rows(), Yl.cols());
  CORA_CALL(cora_riemannian_gradient(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), out.data(),
                                     static_cast<int>(out.rows())),
            "Problem::Riemannian_gradient");
  return lowered(std::move(out));
}

Matrix Problem::Riemannian_gradient(const Matrix &Y, const Matrix &NablaF_Y) const {
  checkUpToDate();
  return tangent_space_projection(Y, NablaF_Y);
}

Matrix Problem::tangent_space_projection(const Matrix &Y, const Matrix &Ydot) const {
  checkMatrixShape("Problem::tangent_space_projection::Y", getExpectedVariableSize(), relaxation_rank_, Y.rows(), Y.cols());
  checkMatrixShape("Problem::tangent_space_projection::Ydot", getExpectedVariableSize(), relaxation_rank_,
                   Ydot.rows(), Ydot.cols());
  ensureContext();
  LIFT(Y, Yl);
  LIFT(Ydot, Vl);
  Matrix out(Yl.rows(), Yl.cols());
  CORA_CALL(cora_tangent_space_projection(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), Vl.data(),
                                          static_cast<int>(Vl.rows()), out.data(), static_cast<int>(out.rows())),
            "Problem::tangent_space_projection");
  return lowered(std::move(out));
}

Matrix Problem::Riemannian_Hessian_vector_product(const Matrix &Y, const Matrix &nablaF_Y, const Matrix &dotY) const {
  checkMatrixShape("Problem::Riemannian_Hessian_vector_product::Y", getExpectedVariableSize(), relaxation_rank_,
                   Y.rows(), Y.cols());
  checkMatrixShape("Problem::Riemannian_Hessian_vector_product::nablaF_Y", getExpectedVariableSize(),
                   relaxation_rank_, nablaF_Y.rows(), nablaF_Y.cols());
  checkMatrixShape("Problem::Riemannian_Hessian_vector_product::dotY", getExpectedVariableSize(), relaxation_rank_,
                   dotY.rows(), dotY.cols());
  ensureContext();
  LIFT(Y, Yl);
  LIFT(nablaF_Y, Gl);
  LIFT(dotY, Vl);
  Matrix out(Yl.rows(), Yl.cols());
  CORA_CALL(cora_riemannian_hessian_vector_product(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), Gl.data(),
                                                   static_cast<int>(Gl.rows()), Vl.data(),
                                                   static_cast<int>(Vl.rows()), out.data(),
                                                   static_cast<int>(out.rows())),
            "Problem::Riemannian_Hessian_vector_product");
  return lowered(std::move(out));
}

Matrix Problem::precondition(const Matrix &V) const {
  checkMatrixShape("Problem::precondition::input", getExpectedVariableSize(), relaxation_rank_, V.rows(), V.cols());
  ensurePreconditioner();
  LIFT(V, Vl);
  Matrix out(Vl.rows(), Vl.cols());
  CORA_CALL(cora_precondition(ctx_.get(), Vl.data(), static_cast<int>(Vl.rows()), out.data(), static_cast<int>(out.rows())),
            "Problem::precondition");
  return lowered(std::move(out));
}

Matrix Problem::projectToManifold(const Matrix &A) const {
  checkMatrixShape("Problem::projectToManifold", getExpectedVariableSize(), relaxation_rank_, A.rows(), A.cols());
  ensureContext();
  LIFT(A, Al);
  Matrix out(Al.rows(), Al.cols());
  CORA_CALL(cora_project_to_manifold(ctx_.get(), Al.data(), static_cast<int>(Al.rows()), out.data(),
                                     static_cast<int>(out.rows())),
            "Problem::projectToManifold");
  return lowered(std::move(out));
}

Matrix Problem::retract(const Matrix &Y, const Matrix &V) const {
  checkMatrixShape("Problem::retract::Y", getExpectedVariableSize(), relaxation_rank_, Y.rows(), Y.cols());
  checkMatrixShape("Problem::retract::V", getExpectedVariableSize(), relaxation_rank_, V.rows(), V.cols());
  ensureContext();
  LIFT(Y, Yl);
  LIFT(V, Vl);
  Matrix out(Yl.rows(), Yl.cols());
  CORA_CALL(cora_retract(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), Vl.data(), static_cast<int>(Vl.rows()),
                         out.data(), static_cast<int>(out.rows())),
            "Problem::retract");
  return lowered(std::move(out));
}

// src/CORA_problem.cpp:1168-1197: [Y; -chol(Q33red) \ (B^T Y); 0], computed on the device
Matrix Problem::getTranslationExplicitSolution(const Matrix &Y) const {
  checkMatrixShape("Problem::getTranslationExplicitSolution::Y", rotAndRangeMatrixSize(), Y.cols(), Y.rows(), Y.cols());
  if (formulation_ != Formulation::Implicit)
    throw std::invalid_argument("getTranslationExplicitSolution needs the implicit formulation");
  ensureContext();
  cora_ctx *c = ctx_.get();
  const int k = static_cast<int>(Y.cols()), N = getDataMatrixSize();
  Matrix Yl_tmp;
  const Matrix &Yl = lifted(Y, Yl_tmp);
  double *dY = nullptr, *dX = nullptr;
  CORA_CALL(cora_dev_alloc(c, k, &dY), "Problem::getTranslationExplicitSolution");
  CORA_CALL(cora_dev_alloc(c, k, &dX), "Problem::getTranslationExplicitSolution");
  Matrix Xfull(N, k);
  int rc = cora_upload(c, Yl.data(), N, k, dY);
  if (rc == CORA_OK) rc = cora_translation_explicit_dev(c, dY, k, dX);
  if (rc == CORA_OK) rc = cora_download(c, dX, k, Xfull.data(), N);
  cora_dev_free(c, dY);
  cora_dev_free(c, dX);
  if (rc != CORA_OK) throwLast(rc, "Problem::getTranslationExplicitSolution");
  checkVariablesAreValid(Xfull);
  return Xfull;
}

Matrix Problem::getRandomInitialGuess(uint64_t seed) const {  // src/CORA_problem.cpp:1023-1028
  checkUpToDate();
  Matrix Y = projectToManifold(Matrix::Random(getExpectedVariableSize(), relaxation_rank_, seed));
  // At rank d the polar factors are in O(d); a block of determinant -1 can neither be left by the solver nor
  // passes checkVariablesAreValid (:1212-1217).  Among the points the reference's sampler can return, take one
  // with every block in SO(d): flip the last row of the reflected blocks.
  if (relaxation_rank_ == dim_)
    for (Index i = 0; i < numPoses(); ++i) {
      Matrix R(dim_, dim_);
      for (Index a = 0; a < dim_; ++a)
        for (Index b = 0; b < dim_; ++b) R(a, b) = Y(i * dim_ + a, b);
      if (determinant(R) < 0)
        for (Index b = 0; b < dim_; ++b) Y(i * dim_ + dim_ - 1, b) = -Y(i * dim_ + dim_ - 1, b);
    }
  return Y;
}

// ---- certification pieces: src/CORA_problem.cpp:1105-1166 --------------------
Problem::LambdaBlocks Problem::compute_Lambda_blocks(const Matrix &Y) const {
  checkMatrixShape("Problem::compute_Lambda_blocks::Y", getExpectedVariableSize(), relaxation_rank_, Y.rows(), Y.cols());
  ensureContext();
  LIFT(Y, Yl);
  Matrix st(dim_, std::max(numPosesDim(), 1));
  Vector ob(std::max(numRangeMeasurements(), 1), 1);
  CORA_CALL(cora_compute_lambda_blocks(ctx_.get(), Yl.data(), static_cast<int>(Yl.rows()), st.data(), ob.data()),
            "Problem::compute_Lambda_blocks");
  return std::make_pair(st.block(0, 0, dim_, numPosesDim()), ob.block(0, 0, numRangeMeasurements(), 1));
}

SparseMatrix Problem::compute_Lambda_from_Lambda_blocks(const LambdaBlocks &L, const int &Lambda_size) const {
  std::vector<Triplet> t;
  for (Index i = 0; i < numPoses(); ++i)
    for (Index r = 0; r < dim_; ++r)
      for (Index c = 0; c < dim_; ++c) t.push_back({i * dim_ + r, i * dim_ + c, L.first(r, i * dim_ + c)});
  const Index rot = numPosesDim();
  for (Index i = 0; i < numRangeMeasurements(); ++i) t.push_back({rot + i, rot + i, L.second(i)});
  SparseMatrix Lambda(Lambda_size, Lambda_size);
  Lambda.setFromTriplets(std::move(t));
  return Lambda;
}

SparseMatrix Problem::get_certificate_matrix(const Matrix &Y) const {
  std::lock_guard<std::recursive_mutex> lock(*cert_mutex_);
  return certificateMatrixCached(Y);
}

const SparseMatrix &Problem::certificateMatrixCached(const Matrix &Y) const { return certificateMatrixFrom(compute_Lambda_blocks(Y)); }

void Problem::prepareCertification(Index num_eigvecs) const {
  std::lock_guard<std::recursive_mutex> lock(*cert_mutex_);
  const Index N = getDataMatrixSize();
  if (static_cast<Index>(cert_perm_.size()) != N)
    cert_perm_ = coraOrdering(dim_, numPoses(), numRangeMeasurements(), numTranslationalStates(), data_matrix_, static_cast<int>(N));
  const size_t n_lambda = static_cast<size_t>(numPosesDim()) * dim_ + static_cast<size_t>(numRangeMeasurements());
  if (cert_S_.rows() != N || cert_lambda_pos_.size() != n_lambda)  // the pattern, with Lambda = 0 for values
    certificateMatrixFrom(LambdaBlocks(Matrix(dim_, numPosesDim()), Vector(numRangeMeasurements(), 1)));
  choleskyAnalyze(cert_S_, static_cast<int>(N), cert_perm_, symbolic_cache_.get());
  num_eigvecs = std::min<Index>(std::min<Index>(num_eigvecs, N), 24);
  if (cert_random_.rows() != N || cert_random_.cols() < num_eigvecs) cert_random_ = Matrix::RandomColumns(N, num_eigvecs, 0xC0FFEEull);
}

const SparseMatrix &Problem::certificateMatrixFrom(const LambdaBlocks &L) const {
  // S = Q - Lambda.  Lambda lives on the d x d diagonal blocks of the rotation rows and on the diagonal of the range
  // rows.  The first call merges every row of Q (sorted by column) with its <= d entries of -Lambda in one pass
  // (entries of Lambda outside Q's pattern -- the off-diagonals of a pose without translation measurements -- are
  // inserted) and remembers where the Lambda entries landed; later calls, three per staircase on one data matrix,
  // rewrite those 9 n + r values and nothing else (14 ms -> 2 ms at 10^5 poses).
  const Index N = getDataMatrixSize(), rot = numPosesDim(), nr = numRangeMeasurements();
  const SparseMatrix &Q = data_matrix_;
  SparseMatrix &S = cert_S_;
  const size_t n_lambda = static_cast<size_t>(rot) * dim_ + static_cast<size_t>(nr);
  auto lambda_entry = [&](size_t e) -> Scalar {  // entry e of Lambda in the order of cert_lambda_pos_
    if (e < static_cast<size_t>(rot) * dim_) {
      const Index row = static_cast<Index>(e) / dim_, c = static_cast<Index>(e) % dim_;
      const Index i = row / dim_, r = row % dim_;
      return L.first(r, i * dim_ + c);
    }
    return L.second(static_cast<Index>(e - static_cast<size_t>(rot) * dim_));
  };
  if (S.rows() == N && cert_lambda_pos_.size() == n_lambda) {
    for (size_t e = 0; e < n_lambda; ++e) {
      const int32_t q = cert_lambda_q_[e];
      const Scalar lv = -lambda_entry(e);
      S.values[static_cast<size_t>(cert_lambda_pos_[e])] = q >= 0 ? Q.values[static_cast<size_t>(q)] + lv : lv;
    }
    return S;
  }
  S = SparseMatrix(N, N);
  cert_lambda_pos_.assign(n_lambda, -1);
  cert_lambda_q_.assign(n_lambda, -1);
  S.inner.reserve(Q.inner.size() + n_lambda);
  S.values.reserve(Q.inner.size() + n_lambda);
  for (Index row = 0; row < N; ++row) {
    int32_t lc[3] = {0, 0, 0};
    Scalar lv[3] = {0, 0, 0};
    size_t le[3] = {0, 0, 0};
    int nl = 0;
    if (row < rot) {
      const Index i = row / dim_;
      for (Index c = 0; c < dim_; ++c) {
        lc[nl] = static_cast<int32_t>(i * dim_ + c);
        le[nl] = static_cast<size_t>(row) * dim_ + static_cast<size_t>(c);
        lv[nl] = -lambda_entry(le[nl]);
        ++nl;
      }
    } else if (row < rot + nr) {
      lc[0] = static_cast<int32_t>(row);
      le[0] = static_cast<size_t>(rot) * dim_ + static_cast<size_t>(row - rot);
      lv[0] = -lambda_entry(le[0]);
      nl = 1;
    }
    int32_t q = Q.outer[static_cast<size_t>(row)];
    const int32_t qe = Q.outer[static_cast<size_t>(row) + 1];
    int k = 0;
    while (q < qe || k < nl) {
      if (k >= nl || (q < qe && Q.inner[q] < lc[k])) {
        S.inner.push_back(Q.inner[q]);
        S.values.push_back(Q.values[q]);
        ++q;
      } else if (q >= qe || lc[k] < Q.inner[q]) {
        cert_lambda_pos_[le[k]] = static_cast<int32_t>(S.inner.size());
        S.inner.push_back(lc[k]);
        S.values.push_back(lv[k]);
        ++k;
      } else {
        cert_lambda_pos_[le[k]] = static_cast<int32_t>(S.inner.size());
        cert_lambda_q_[le[k]] = q;
        S.inner.push_back(lc[k]);
        S.values.push_back(Q.values[q] + lv[k]);
        ++q;
        ++k;
      }
    }
    S.outer[static_cast<size_t>(row) + 1] = static_cast<int32_t>(S.inner.size());
  }
  return S;
}

// ---- printProblem: src/CORA_problem.cpp:400-489 -------------------------------
namespace {
void printMatrix(const char *label, const Matrix &M) {
  std::cout << label;
  for (Index i = 0; i < M.rows(); ++i) {
    std::cout << (M.rows() > 1 ? "\n  " : " ");
    for (Index j = 0; j < M.cols(); ++j) std::cout << M(i, j) << (j + 1 < M.cols() ? " " : "");
  }
  std::cout << std::endl;
}
}  // namespace

void Problem::printProblem() const {
  auto section = [](bool any, const char *title, const char *none) {
    std::cout << (any ? title : none) << std::endl;
    return any;
  };
  if (section(numPoses() > 0, "Pose variables:", "No pose variables"))
    for (const auto &kv : pose_symbol_idxs_) std::cout << kv.first.string() << " -> " << kv.second << std::endl;
  if (section(numLandmarks() > 0, "\nLandmark variables:", "No landmark variables"))
    for (const auto &kv : landmark_symbol_idxs_) std::cout << kv.first.string() << " -> " << kv.second << std::endl;
  if (section(numRangeMeasurements() > 0, "\nRange measurements:", "No range measurements"))
    for (const auto &m : range_measurements_)
      std::cout << m.first_id.string() << " -> " << m.second_id.string() << " " << m.r << " " << m.cov << std::endl;
  if (section(numPosePoseMeasurements() > 0, "\nRelative pose measurements:", "No relative pose measurements"))
    for (const auto &m : rel_pose_pose_measurements_) {
      std::cout << m.first_id.string() << " -> " << m.second_id.string() << std::endl;
      printMatrix("Rot:", m.R);
      printMatrix("Trans:", m.t.transpose());
      printMatrix("Cov:", m.cov);
    }
  if (section(numPoseLandmarkMeasurements() > 0, "\nRelative pose landmark measurements:",
              "No relative pose landmark measurements"))
    for (const auto &m : rel_pose_landmark_measurements_) {
      std::cout << m.first_id.string() << " -> " << m.second_id.string() << std::endl;
      printMatrix("Trans:", m.t.transpose());
      printMatrix("Cov:", m.cov);
    }
  if (section(!pose_priors_.empty(), "\nPose priors:", "No pose priors"))
    for (const auto &m : pose_priors_) {
      std::cout << m.id.string() << std::endl;
      printMatrix("Rot:", m.R);
      printMatrix("Trans:", m.t.transpose());
      printMatrix("Cov:", m.cov);
    }
  if (section(!landmark_priors_.empty(), "\nLandmark priors:", "No landmark priors"))
    for (const auto &m : landmark_priors_) {
      std::cout << m.id.string() << std::endl;
      printMatrix("Position:", m.p.transpose());
      printMatrix("Cov:", m.cov);
    }
}

// ---- utilities: src/CORA_problem.cpp:1199-1306 -------------------------------
void Problem::checkVariablesAreValid(const Matrix &Y) const {
  const Index p = Y.cols();
  for (Index i = 0; i < numPoses(); ++i) {
    for (Index a = 0; a < dim_; ++a)
      for (Index b = 0; b < dim_; ++b) {
        Scalar s = 0;
        for (Index c = 0; c < p; ++c) s += Y(i * dim_ + a, c) * Y(i * dim_ + b, c);
        if (std::abs(s - (a == b ? 1.0 : 0.0)) > 1e-6) {
          std::cout << "R^T R for pose " << i << " is not the identity" << std::endl;
          throw std::runtime_error("Pose is not a valid rotation matrix");
        }
      }
    if (p == dim_) {  // det(R) = +1 at rank d (src/CORA_problem.cpp:1212-1217)
      Matrix R(dim_, dim_);
      for (Index a = 0; a < dim_; ++a)
        for (Index b = 0; b < dim_; ++b) R(a, b) = Y(i * dim_ + a, b);
      const Scalar det = determinant(R);
      if (std::abs(det - 1) > 1e-6) {
        std::cout << "Pose " << i << " has determinant " << det << std::endl;
        throw std::runtime_error("Pose does not have determinant 1");
      }
    }
  }
  for (Index j = 0; j < numRangeMeasurements(); ++j) {
    Scalar s = 0;
    for (Index c = 0; c < p; ++c) s += Y(numPosesDim() + j, c) * Y(numPosesDim() + j, c);
    if (std::abs(std::sqrt(s) - 1.0) > 1e-6) {
      std::cout << "Range " << j << " has norm " << std::sqrt(s) << std::endl;
      throw std::runtime_error("Range is not a unit vector");
    }
  }
}

Matrix Problem::alignEstimateToOrigin(const Matrix &Y) const {
  checkVariablesAreValid(Y);
  Matrix Ya = Y;
  if (numPoses() > 0) {
    const Matrix first = Y.block(0, 0, dim_, Y.cols());  // d x p block of the first pose
    if (Y.cols() != dim_) throw std::runtime_error("alignEstimateToOrigin expects a rank-d solution");
    Ya = Y * first.transpose();
  }
  if (formulation_ == Formulation::Implicit) Ya = getTranslationExplicitSolution(Ya);  // :1250-1252
  checkVariablesAreValid(Ya);
  const Index off = rotAndRangeMatrixSize(), nt = numTranslationalStates();
  for (Index c = 0; c < Ya.cols(); ++c) {
    Scalar mean = 0;
    for (Index i = 0; i < nt; ++i) mean += Ya(off + i, c);
    mean /= static_cast<Scalar>(std::max<Index>(nt, 1));
    for (Index i = 0; i < nt; ++i) Ya(off + i, c) -= mean;
  }
  return Ya;
}

}  // namespace CORA

namespace CORA {
Matrix Matrix::RandomColumns(Index r, Index c, uint64_t seed) {
  Matrix m(r, c);
  const unsigned nth = r * c < 200000 ? 1u : static_cast<unsigned>(std::min<Index>(c, std::max(1u, std::thread::hardware_concurrency())));
  cora::parallel_parts(nth, [&](unsigned t) {
    for (Index j = c * t / nth; j < c * (t + 1) / nth; ++j) {
      std::mt19937_64 g(seed + static_cast<uint64_t>(j));
      std::uniform_real_distribution<Scalar> u(-1.0, 1.0);
      Scalar *col = m.data() + static_cast<size_t>(j) * static_cast<size_t>(r);
      for (Index i = 0; i < r; ++i) col[i] = u(g);
    }
  });
  return m;
}
}  // namespace CORA

// ---- certification: src/CORA_problem.cpp:1030-1103 ---------------------------
#include "CORA_utils.h"
#include "dense.h"

namespace CORA {

CertResults Problem::certify_solution(const Matrix &Y, Scalar eta, size_t nx, const Matrix &eigvec_bootstrap,
                                      size_t max_LOBPCG_iters) const {
  return certifyImpl(Y, eta, nx, eigvec_bootstrap, max_LOBPCG_iters, false);
}
CertResults Problem::certify_solution_resident(const Matrix &Y, Scalar eta, size_t nx, const Matrix &eigvec_bootstrap,
                                               size_t max_LOBPCG_iters) const {
  return certifyImpl(Y, eta, nx, eigvec_bootstrap, max_LOBPCG_iters, true);
}
CertResults Problem::certifyImpl(const Matrix &Y, Scalar eta, size_t nx, const Matrix &eigvec_bootstrap, size_t max_LOBPCG_iters,
                                 bool resident) const {
  std::lock_guard<std::recursive_mutex> lock(*cert_mutex_);
  checkMatrixShape("Problem::certify_solution::Y", getExpectedVariableSize(), relaxation_rank_, Y.rows(), Y.cols());
  const Index N = getDataMatrixSize(), p = Y.cols();
  const bool timing = std::getenv("CORA_TRI_TIMING") != nullptr;
  auto tick = [t_prev = std::chrono::steady_clock::now(), timing](const char *what) mutable {
    const auto now = std::chrono::steady_clock::now();
    if (timing) std::fprintf(stderr, "  [certify] %-26s %.3f s\n", what, std::chrono::duration<double>(now - t_prev).count());
    t_prev = now;
  };
  // ratio of the extreme singular values of Y from the p x p Gram matrix (:1039-1049)
  {
    Vector ev;
    Matrix V;
    // Y' Y over row ranges on threads, the partial matrices added in range order (only the ratio test below reads it)
    Matrix G(p, p);
    {
      const Index Nr = Y.rows();
      const unsigned nth = Nr < 50000 ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
      std::vector<Matrix> part(nth, Matrix(p, p));
      auto rows = [&](unsigned t) {
        Matrix &g = part[t];
        for (Index a = 0; a < p; ++a)
          for (Index b = a; b < p; ++b) {
            Scalar sum = 0;
            for (Index i = Nr * static_cast<Index>(t) / nth; i < Nr * static_cast<Index>(t + 1) / nth; ++i) sum += Y(i, a) * Y(i, b);
            g(a, b) = sum;
            g(b, a) = sum;
          }
      };
      cora::parallel_parts(nth, rows);
      for (unsigned t = 0; t < nth; ++t) G = G + part[t];
    }
    symmetricEigen(G, ev, V);
    const Scalar smax = std::sqrt(std::max(ev(p - 1), 0.0)), smin = std::sqrt(std::max(ev(0), 0.0));
    if (smax / smin > 1e6) {
      CertResults r;
      r.is_certified = true;
      r.theta = 0;
      r.x = Vector::Zero(N, 1);
      r.all_eigvecs = Matrix::Zero(N, static_cast<Index>(nx));
      r.num_iters = 0;
      return r;
    }
  }
  // S = Q - Lambda(Y): Lambda on the device (also leaves Y as the handle's current point, so the
  // certificate operator below uses the same Lambda), S assembled on the host for the Cholesky test
  tick("Gram of Y");
  const SparseMatrix &S = certificateMatrixCached(Y);
  tick("certificate matrix");
  cora_ctx *c = ctx_.get();
  // device vectors carry at most 24 columns (kMaxLD): the block is clamped there (the reference has no cap; p + 2 only
  // exceeds it at ranks solveCORA rejects up front)
  const Index num_eigvecs = std::min<Index>(std::min<Index>(std::max<Index>(static_cast<Index>(nx), p + 2), N), 24);
  if (N < p) throw std::invalid_argument("The number of rows of S must be greater than or equal to the number of columns of Y");
  // start block: the leading columns of the bootstrap block, the rest random (Matrix::RandomColumns(N, num_eigvecs) with a
  // fixed seed fills column after column, so a cached block with at least as many columns has the same numbers in
  // them); the two pieces go to the device as they are
  // (resident: an empty bootstrap means "the block the last certification left on the device")
  const std::shared_ptr<LOBPCGSolver> prev = resident && eigvec_bootstrap.rows() == 0 ? cert_block_ : nullptr;
  const bool from_device = prev && prev->deviceBlock() != nullptr;
  const Index from_bootstrap = from_device ? std::min<Index>(prev->width(), num_eigvecs)
                                           : (eigvec_bootstrap.rows() == N ? std::min(eigvec_bootstrap.cols(), num_eigvecs) : 0);
  std::vector<HostColumns> X0;
  if (from_device) {
    HostColumns h;
    h.cols = static_cast<int>(from_bootstrap);
    h.device = prev->deviceBlock();
    h.device_width = prev->width();
    X0.push_back(h);
  } else if (from_bootstrap > 0) {
    X0.push_back(HostColumns{eigvec_bootstrap.data(), static_cast<int>(from_bootstrap)});
  }
  if (from_bootstrap < num_eigvecs) {
    if (cert_random_.rows() != N || cert_random_.cols() < num_eigvecs) cert_random_ = Matrix::RandomColumns(N, num_eigvecs, 0xC0FFEEull);
    X0.push_back(HostColumns{cert_random_.data() + static_cast<size_t>(from_bootstrap) * static_cast<size_t>(N),
                             static_cast<int>(num_eigvecs - from_bootstrap)});
  }
  if (static_cast<Index>(cert_perm_.size()) != N)
    cert_perm_ = coraOrdering(dim_, numPoses(), numRangeMeasurements(), numTranslationalStates(), data_matrix_,
                              static_cast<int>(N));
  const std::vector<int32_t> &perm = cert_perm_;
  DeviceOperator Sop = [c](const double *dX, int k, double *dOut) {
    if (cora_certificate_product_dev(c, dX, k, dOut) != CORA_OK) throw std::runtime_error(cora_last_error(c));
  };
  tick("start block + ordering");
  FastVerificationLab lab;
  lab.seed_negative_direction = cert_lab_seed_;
  lab.use_ildl = cert_lab_ildl_;
  const FastVerificationLab *labp = cert_lab_on_ ? &lab : nullptr;
  std::shared_ptr<LOBPCGSolver> kept;
  std::shared_ptr<LOBPCGSolver> *keep = resident ? &kept : nullptr;
  CertResults results = fast_verification(S, eta, X0, max_LOBPCG_iters, perm, c, Sop, std::nullopt, 3, 1e-3, labp, symbolic_cache_.get(), keep);
  cert_reached_step3_ = lab.reached_step3;
  tick("fast_verification");
  while (std::isnan(results.theta)) {  // :1076-1083
    std::cout << "NaN in theta -- result not certified" << std::endl;
    eta *= 2;
    results = fast_verification(S, eta, X0, max_LOBPCG_iters, perm, c, Sop, std::nullopt, 3, 1e-3, labp, symbolic_cache_.get(), keep);
  }
  if (resident) cert_block_ = kept;  // (the previous block goes now: the new one has been built from it)
  if (!results.is_certified && formulation_ == Formulation::Implicit) {  // :1085-1100
    // leading (rotation + range) part of the direction, and its Rayleigh quotient with the
    // simplified certificate Q_impl - Lambda
    const Index m = rotAndRangeMatrixSize();
    Vector v = results.x.block(0, 0, m, 1);
    const Scalar nv = v.norm();
    if (nv > 0) v = v * (1.0 / nv);
    results.x = v;
    const LambdaBlocks Lb = compute_Lambda_blocks(Y);
    const Vector Sx = dataMatrixProduct(v) - compute_Lambda_from_Lambda_blocks(Lb, static_cast<int>(m)) * v;
    Scalar th = 0;
    for (Index i = 0; i < m; ++i) th += v(i) * Sx(i);
    results.theta = th;
    if (std::isnan(results.theta)) throw std::runtime_error("NaN in theta -- result not certified and implicit form");
  }
  return results;
}

}  // namespace CORA
