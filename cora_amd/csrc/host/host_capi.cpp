// Flat C view of the C++ host for ctypes (include/cora_host.h).
#include "../../../include/cora_host.h"

#include <chrono>
#include <cstring>
#include <memory>
#include <string>

#include "../../../include/cora_hip.h"
#include "CORA_preconditioners.h"
#include "CORA_problem.h"
#include "Manifolds.h"
#include "io.h"
#include "odometry_init.h"
#include "pyfg_text_parser.h"
#include "CORA.h"
#include "TNT.h"
#include "sparse_cholesky.h"
#include "../trisolve.h"
#include "synthetic.h"

using namespace CORA;

struct cora_problem {
  Problem problem;
  SparseMatrix last_certificate;  // cora_problem_certificate_matrix keeps its result alive here
  explicit cora_problem(Problem p) : problem(std::move(p)) {}
};

namespace {
thread_local std::string g_err;

template <typename F>
int guarded(F &&f) {
  try {
    f();
    return 0;
  } catch (const std::exception &e) {
    g_err = e.what();
    return 1;
  } catch (...) {
    g_err = "unknown exception";
    return 1;
  }
}

Preconditioner precondOf(int kind) {
  switch (kind) {
    case CORA_PRECOND_NONE: return Preconditioner::None;
    case CORA_PRECOND_JACOBI: return Preconditioner::Jacobi;
    case CORA_PRECOND_BLOCK_CHOLESKY: return Preconditioner::BlockCholesky;
    case CORA_PRECOND_REGULARIZED_CHOLESKY: return Preconditioner::RegularizedCholesky;
    default: throw std::invalid_argument("unknown preconditioner kind");
  }
}

Matrix wrap(const double *a, Index N, Index p) {
  if (!a) throw std::invalid_argument("missing input matrix");
  Matrix m(N, p);
  std::memcpy(m.data(), a, sizeof(double) * static_cast<size_t>(N * p));
  return m;
}
}  // namespace

extern "C" {

const char *cora_host_last_error(void) { return g_err.c_str(); }

int cora_problem_from_pyfg(const char *path, cora_problem **out) {
  return guarded([&] { *out = new cora_problem(parsePyfgTextToProblem(path)); });
}

int cora_problem_synthetic(int dim, int n_poses, int n_landmarks, int n_ranges, int n_loops, uint64_t seed,
                           int precond, const char *pyfg_out, cora_problem **out) {
  return guarded([&] {
    SyntheticSpec sp;
    sp.dim = dim;
    sp.num_poses = n_poses;
    sp.num_landmarks = n_landmarks;
    sp.num_ranges = n_ranges;
    sp.num_loop_closures = n_loops;
    sp.seed = seed;
    *out = new cora_problem(makeSyntheticProblem(sp, precondOf(precond), pyfg_out ? pyfg_out : ""));
  });
}

// ---- programmatic construction: the add* methods of CORA::Problem (include/CORA/CORA_problem.h:202-262)
int cora_problem_new(int dim, int rank, int implicit, int precond, cora_problem **out) {
  return guarded([&] {
    *out = new cora_problem(Problem(dim, rank, implicit ? Formulation::Implicit : Formulation::Explicit, precondOf(precond)));
  });
}
int cora_problem_add_pose(cora_problem *p, const char *id) {
  return guarded([&] { p->problem.addPoseVariable(Symbol(std::string(id))); });
}
int cora_problem_add_landmark(cora_problem *p, const char *id) {
  return guarded([&] { p->problem.addLandmarkVariable(Symbol(std::string(id))); });
}
int cora_problem_add_range(cora_problem *p, const char *a, const char *b, double dist, double cov) {
  return guarded([&] { p->problem.addRangeMeasurement(RangeMeasurement(Symbol(std::string(a)), Symbol(std::string(b)), dist, cov)); });
}
int cora_problem_add_rel_pose(cora_problem *p, const char *a, const char *b, const double *R, const double *t,
                              const double *cov) {
  return guarded([&] {
    const Index d = p->problem.dim(), c = d == 3 ? 6 : 3;
    p->problem.addRelativePoseMeasurement(
        RelativePoseMeasurement(Symbol(std::string(a)), Symbol(std::string(b)), wrap(R, d, d), wrap(t, d, 1), wrap(cov, c, c)));
  });
}
int cora_problem_add_rel_pose_landmark(cora_problem *p, const char *a, const char *b, const double *t,
                                       const double *cov) {
  return guarded([&] {
    const Index d = p->problem.dim();
    p->problem.addRelativePoseLandmarkMeasurement(
        RelativePoseLandmarkMeasurement(Symbol(std::string(a)), Symbol(std::string(b)), wrap(t, d, 1), wrap(cov, d, d)));
  });
}
int cora_problem_add_pose_prior(cora_problem *p, const char *id, const double *R, const double *t, const double *cov) {
  return guarded([&] {
    const Index d = p->problem.dim(), c = d == 3 ? 6 : 3;
    p->problem.addPosePrior(PosePrior(Symbol(std::string(id)), wrap(R, d, d), wrap(t, d, 1), wrap(cov, c, c)));
  });
}
int cora_problem_add_landmark_prior(cora_problem *p, const char *id, const double *pos, const double *cov) {
  return guarded([&] {
    const Index d = p->problem.dim();
    p->problem.addLandmarkPrior(LandmarkPrior(Symbol(std::string(id)), wrap(pos, d, 1), wrap(cov, d, d)));
  });
}

int cora_problem_synthetic_ex(int dim, int n_poses, int n_landmarks, int n_ranges, int n_loops, uint64_t seed,
                              int precond, const double *sigmas, const char *pyfg_out, double *x_gt,
                              cora_problem **out) {
  return guarded([&] {
    SyntheticSpec sp;
    sp.dim = dim;
    sp.num_poses = n_poses;
    sp.num_landmarks = n_landmarks;
    sp.num_ranges = n_ranges;
    sp.num_loop_closures = n_loops;
    sp.seed = seed;
    if (sigmas) {
      sp.sigma_t = sigmas[0];
      sp.sigma_R = sigmas[1];
      sp.sigma_range = sigmas[2];
    }
    Matrix gt;
    if (x_gt) sp.ground_truth = &gt;
    *out = new cora_problem(makeSyntheticProblem(sp, precondOf(precond), pyfg_out ? pyfg_out : ""));
    if (x_gt) std::memcpy(x_gt, gt.data(), sizeof(double) * static_cast<size_t>(gt.size()));
  });
}

void cora_problem_destroy(cora_problem *p) { delete p; }

int cora_problem_update(cora_problem *p) {
  return guarded([&] { p->problem.updateProblemData(); });
}

int cora_problem_dims(const cora_problem *p, int64_t dims[8]) {
  return guarded([&] {
    const Problem &q = p->problem;
    dims[0] = q.dim();
    dims[1] = q.numPoses();
    dims[2] = q.numLandmarks();
    dims[3] = q.numRangeMeasurements();
    dims[4] = q.getDataMatrixSize();
    dims[5] = q.data_matrix_.nonZeros();
    dims[6] = q.numPosePoseMeasurements();
    dims[7] = static_cast<int64_t>(q.getRelaxationRank());
  });
}

int cora_problem_matrix(cora_problem *p, const char *name, int64_t *rows, int64_t *cols, int64_t *nnz,
                        const int32_t **rowptr, const int32_t **colidx, const double **vals) {
  return guarded([&] {
    const std::string n(name);
    const SparseMatrix *m = nullptr;
    const CoraDataSubmatrices &s = p->problem.getDataSubmatrices();
    if (n == "DataMatrix") m = &p->problem.getDataMatrix();
    else if (n == "Arange") m = &s.range_incidence_matrix;
    else if (n == "OmegaRange") m = &s.range_precision_matrix;
    else if (n == "RangeDistances") m = &s.range_dist_matrix;
    else if (n == "Apose") m = &s.rel_pose_incidence_matrix;
    else if (n == "OmegaPose") m = &s.rel_pose_translation_precision_matrix;
    else if (n == "T") m = &s.rel_pose_translation_data_matrix;
    else if (n == "RotConLaplacian") m = &s.rotation_conn_laplacian;
    else throw std::invalid_argument("unknown matrix name " + n);
    *rows = m->rows();
    *cols = m->cols();
    *nnz = m->nonZeros();
    *rowptr = m->outerIndexPtr();
    *colidx = m->innerIndexPtr();
    *vals = m->valuePtr();
  });
}

int cora_problem_certificate_matrix(cora_problem *p, const double *Y, int ldy, int64_t *rows, int64_t *nnz,
                                    const int32_t **rowptr, const int32_t **colidx, const double **vals) {
  return guarded([&] {
    Problem &q = p->problem;
    const Matrix Ym = wrap(Y, static_cast<Index>(ldy), static_cast<Index>(q.getRelaxationRank()));
    p->last_certificate = q.get_certificate_matrix(Ym);  // S = Q - Lambda(Y), src/CORA_problem.cpp:1162-1166
    *rows = p->last_certificate.rows();
    *nnz = p->last_certificate.nonZeros();
    *rowptr = p->last_certificate.outerIndexPtr();
    *colidx = p->last_certificate.innerIndexPtr();
    *vals = p->last_certificate.valuePtr();
  });
}

int cora_problem_set_rank(cora_problem *p, int rank) {
  return guarded([&] { p->problem.setRank(rank); });
}
int cora_problem_set_preconditioner(cora_problem *p, int kind) {
  return guarded([&] { p->problem.setPreconditioner(precondOf(kind)); });
}
int cora_problem_set_formulation(cora_problem *p, int implicit) {
  return guarded([&] { p->problem.setFormulation(implicit ? Formulation::Implicit : Formulation::Explicit); });
}
int cora_problem_variable_size(cora_problem *p, int64_t *rows) {
  return guarded([&] { *rows = p->problem.getExpectedVariableSize(); });
}
int cora_problem_set_device(cora_problem *p, int device) {
  return guarded([&] { p->problem.setDevice(device); });
}
int cora_problem_set_partition(cora_problem *p, int rank, int world, cora_exchange_fn exchange,
                               cora_allreduce_fn allreduce, cora_allgather_fn allgather, void *user) {
  return guarded([&] { p->problem.setPartition(rank, world, exchange, allreduce, allgather, user); });
}

int cora_problem_op(cora_problem *p, const char *op, int cols, const double *A, const double *B,
                    const double *C, double *out) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = cols;
    const std::string o(op);
    Matrix res;
    if (o == "evaluateObjective") {
      out[0] = q.evaluateObjective(wrap(A, N, r));
      return;
    } else if (o == "Euclidean_gradient") res = q.Euclidean_gradient(wrap(A, N, r));
    else if (o == "Riemannian_gradient") res = q.Riemannian_gradient(wrap(A, N, r));
    else if (o == "tangent_space_projection") res = q.tangent_space_projection(wrap(A, N, r), wrap(B, N, r));
    else if (o == "Riemannian_Hessian_vector_product")
      res = q.Riemannian_Hessian_vector_product(wrap(A, N, r), wrap(B, N, r), wrap(C, N, r));
    else if (o == "precondition") res = q.precondition(wrap(A, N, r));
    else if (o == "projectToManifold") res = q.projectToManifold(wrap(A, N, r));
    else if (o == "retract") res = q.retract(wrap(A, N, r), wrap(B, N, r));
    else if (o == "getRandomInitialGuess") res = q.getRandomInitialGuess();
    else if (o == "getOdomInitialization") res = getOdomInitialization(q);
    else if (o == "getTranslationExplicitSolution") res = q.getTranslationExplicitSolution(wrap(A, N, r));
    else if (o == "alignEstimateToOrigin") res = q.alignEstimateToOrigin(wrap(A, N, r));
    else throw std::invalid_argument("unknown operator " + o);
    std::memcpy(out, res.data(), sizeof(double) * static_cast<size_t>(res.size()));
  });
}

int cora_problem_lambda_blocks(cora_problem *p, const double *Y, double *stiefel, double *oblique) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    auto L = q.compute_Lambda_blocks(wrap(Y, N, r));
    std::memcpy(stiefel, L.first.data(), sizeof(double) * static_cast<size_t>(L.first.size()));
    std::memcpy(oblique, L.second.data(), sizeof(double) * static_cast<size_t>(L.second.size()));
  });
}

int cora_problem_tnt(cora_problem *p, const double *x0, const double *opts, double *x_out, double stats[7]) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    TNTParams prm;
    if (opts) {
      if (opts[0] > 0) prm.max_iterations = static_cast<int>(opts[0]);
      if (opts[1] > 0) prm.max_TPCG_iterations = static_cast<int>(opts[1]);
      if (opts[2] > 0) prm.gradient_tolerance = opts[2];
      if (opts[3] > 0) prm.preconditioned_gradient_tolerance = opts[3];
      if (opts[4] > 0) prm.max_computation_time = opts[4];
      prm.verbose = opts[5] != 0;
      prm.device_stpcg = opts[6] == 0;
    }
    const TNTResult res = TNT(q, wrap(x0, N, r), prm);
    std::memcpy(x_out, res.x.data(), sizeof(double) * static_cast<size_t>(res.x.size()));
    stats[0] = res.f;
    stats[1] = res.gradfx_norm;
    stats[2] = res.preconditioned_gradfx_norm;
    stats[3] = static_cast<double>(res.inner_iterations.size());
    stats[4] = static_cast<double>(res.hessian_vector_products);
    stats[5] = static_cast<double>(static_cast<int>(res.status));
    stats[6] = res.elapsed_time;
  });
}

int cora_problem_tnt_step(cora_problem *p, const double *x, double Delta, int host_stpcg, double *x_out, double out[8]) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    TNTParams prm;
    prm.max_iterations = 1;
    prm.Delta0 = Delta;
    prm.device_stpcg = host_stpcg == 0;
    const TNTResult res = TNT(q, wrap(x, N, r), prm);
    std::memcpy(x_out, res.x.data(), sizeof(double) * static_cast<size_t>(res.x.size()));
    const bool ran = !res.inner_iterations.empty();
    out[0] = res.f;
    out[1] = res.final_trust_region_radius;
    out[2] = ran ? static_cast<double>(res.inner_iterations.back()) : 0.0;
    out[3] = ran ? res.gain_ratios.back() : 0.0;
    out[4] = static_cast<double>(res.accepted_steps);
    out[5] = ran ? res.update_step_norms.back() : 0.0;
    out[6] = ran ? res.update_step_M_norms.back() : 0.0;
    out[7] = static_cast<double>(static_cast<int>(res.status));
  });
}

int cora_problem_certify(cora_problem *p, const double *Y, double eta, int nx, double out[3], double *x) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    const Matrix Ym = wrap(Y, N, r);
    const Matrix boot = q.getFormulation() == Formulation::Implicit ? q.getTranslationExplicitSolution(Ym) : Ym;
    const CertResults c = q.certify_solution(Ym, eta, static_cast<size_t>(nx), boot);
    out[0] = c.is_certified ? 1.0 : 0.0;
    out[1] = c.theta;
    out[2] = static_cast<double>(c.num_iters);
    if (x) std::memcpy(x, c.x.data(), sizeof(double) * static_cast<size_t>(c.x.size()));
  });
}

int cora_problem_certify_resident(cora_problem *p, const double *Y, double eta, int nx, int first, double out[3], double *x) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    const Matrix Ym = wrap(Y, N, r);
    Matrix boot;  // empty: start from the Ritz block the previous certification left on the device (src/CORA.cpp:165-170)
    if (first) boot = q.getFormulation() == Formulation::Implicit ? q.getTranslationExplicitSolution(Ym) : Ym;  // :158-164
    const CertResults c = q.certify_solution_resident(Ym, eta, static_cast<size_t>(nx), boot);
    out[0] = c.is_certified ? 1.0 : 0.0;
    out[1] = c.theta;
    out[2] = static_cast<double>(c.num_iters);
    if (x) std::memcpy(x, c.x.data(), sizeof(double) * static_cast<size_t>(c.x.size()));
  });
}

int cora_problem_saddle_escape(cora_problem *p, const double *Y, double theta, const double *v, double grad_tol,
                               double pgrad_tol, double *y_out, double info[3]) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    if (r < 2) throw std::invalid_argument("saddle escape needs the incremented rank");
    const Matrix Ym = wrap(Y, N, r - 1);
    Vector vv(N, 1);
    for (Index i = 0; i < N; ++i) vv(i) = v[i];
    Matrix Yaug(N, r);
    Yaug.setBlock(0, 0, Ym);
    const Matrix out = saddleEscape(q, Ym, theta, vv, grad_tol, pgrad_tol);
    std::memcpy(y_out, out.data(), sizeof(double) * static_cast<size_t>(out.size()));
    info[0] = q.evaluateObjective(Yaug);
    info[1] = q.evaluateObjective(out);
    bool moved = false;
    for (Index k = 0; k < out.size() && !moved; ++k) moved = out.data()[k] != Yaug.data()[k];
    info[2] = moved ? 1.0 : 0.0;
  });
}

int cora_problem_project_solution(cora_problem *p, const double *Y, double *y_out) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    const Matrix out = projectSolution(q, wrap(Y, N, r), false);
    std::memcpy(y_out, out.data(), sizeof(double) * static_cast<size_t>(out.size()));
  });
}

int cora_problem_certify_chain(cora_problem *p, const double *Y, double eta, int nx, int resident, double out[6], double *x) {
  return guarded([&] {
    // two certifications in a row at the same point, the second one started from the first one's Ritz block: handed over
    // on the host (resident = 0: certify_solution with all_eigvecs as the bootstrap) or left on the device
    // (resident = 1: certify_solution_resident with an empty bootstrap) -- the two ways must give the same numbers
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    const Matrix Ym = wrap(Y, N, r);
    const Matrix boot = q.getFormulation() == Formulation::Implicit ? q.getTranslationExplicitSolution(Ym) : Ym;
    const CertResults a = resident ? q.certify_solution_resident(Ym, eta, static_cast<size_t>(nx), boot)
                                   : q.certify_solution(Ym, eta, static_cast<size_t>(nx), boot);
    const CertResults b = resident ? q.certify_solution_resident(Ym, eta, static_cast<size_t>(nx), Matrix())
                                   : q.certify_solution(Ym, eta, static_cast<size_t>(nx), a.all_eigvecs);
    out[0] = a.is_certified ? 1.0 : 0.0;
    out[1] = a.theta;
    out[2] = static_cast<double>(a.num_iters);
    out[3] = b.is_certified ? 1.0 : 0.0;
    out[4] = b.theta;
    out[5] = static_cast<double>(b.num_iters);
    if (x) std::memcpy(x, b.x.data(), sizeof(double) * static_cast<size_t>(b.x.size()));
  });
}

int cora_problem_set_verification_lab(cora_problem *p, int seed_negative_direction, int use_ildl) {
  return guarded([&] { p->problem.setVerificationLab(seed_negative_direction != 0, use_ildl != 0); });
}

int cora_problem_certification_reached_step3(cora_problem *p, int *reached) {
  return guarded([&] {
    if (!reached) throw std::invalid_argument("reached is NULL");
    *reached = p->problem.lastCertificationReachedStep3() ? 1 : 0;
  });
}

int cora_host_cholesky_test(int d, int n_poses, int n_ranges, int n_trans, int n, const int32_t *rowptr, const int32_t *colidx,
                            const double *vals, double shift, int leaf_poses, int64_t out[3]) {
  return guarded([&] {
    SparseMatrix S(n, n);
    S.outer.assign(rowptr, rowptr + n + 1);
    S.inner.assign(colidx, colidx + rowptr[n]);
    S.values.assign(vals, vals + rowptr[n]);
    const auto perm = coraOrdering(d, n_poses, n_ranges, n_trans, S, n, leaf_poses > 0 ? leaf_poses : 16);
    const CholeskyFactor F = choleskyFactor(S, n, shift, perm, nullptr);
    out[0] = F.ok ? 1 : 0;
    out[1] = F.failed_column;
    out[2] = F.nnz();
  });
}

int cora_host_fast_verification_lab(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                    double eta, const double *X0, int nx, int max_iters, const double opts[4],
                                    double out[4], double *x) {
  return guarded([&] {
    SparseMatrix S(n, n);
    S.outer.assign(rowptr, rowptr + n + 1);
    S.inner.assign(colidx, colidx + rowptr[n]);
    S.values.assign(vals, vals + rowptr[n]);
    const Matrix X = X0 ? wrap(X0, n, nx) : Matrix::Random(n, nx, 99);
    FastVerificationLab lab;
    lab.seed_negative_direction = opts[2] != 0.0;
    lab.use_ildl = opts[3] != 0.0;
    const CertResults c = fast_verification(S, eta, X, static_cast<size_t>(max_iters), {}, nullptr, std::nullopt,
                                            std::nullopt, opts[0], opts[1], &lab);
    out[0] = c.is_certified ? 1.0 : 0.0;
    out[1] = c.theta;
    out[2] = static_cast<double>(c.num_iters);
    out[3] = lab.reached_step3 ? 1.0 : 0.0;
    if (x) std::memcpy(x, c.x.data(), sizeof(double) * static_cast<size_t>(c.x.size()));
  });
}

int cora_host_fast_verification(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                double eta, const double *X0, int nx, int max_iters, double out[3], double *x) {
  return guarded([&] {
    SparseMatrix S(n, n);
    S.outer.assign(rowptr, rowptr + n + 1);
    S.inner.assign(colidx, colidx + rowptr[n]);
    S.values.assign(vals, vals + rowptr[n]);
    const Matrix X = X0 ? wrap(X0, n, nx) : Matrix::Random(n, nx, 99);
    const CertResults c = fast_verification(S, eta, X, static_cast<size_t>(max_iters));
    out[0] = c.is_certified ? 1.0 : 0.0;
    out[1] = c.theta;
    out[2] = static_cast<double>(c.num_iters);
    if (x) std::memcpy(x, c.x.data(), sizeof(double) * static_cast<size_t>(c.x.size()));
  });
}

int cora_host_fast_verification_pieces(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals,
                                       double eta, const double *X0, int nx, int split, int max_iters, double out[3],
                                       double *x) {
  return guarded([&] {
    if (!X0 || split <= 0 || split >= nx) throw std::invalid_argument("fast_verification_pieces: 0 < split < nx and a start block");
    SparseMatrix S(n, n);
    S.outer.assign(rowptr, rowptr + n + 1);
    S.inner.assign(colidx, colidx + rowptr[n]);
    S.values.assign(vals, vals + rowptr[n]);
    // the start block as two pieces of host memory (its first `split` columns, the rest): put side by side on the device
    const std::vector<HostColumns> pieces = {HostColumns{X0, split},
                                             HostColumns{X0 + static_cast<size_t>(split) * static_cast<size_t>(n), nx - split}};
    const CertResults c = fast_verification(S, eta, pieces, static_cast<size_t>(max_iters));
    out[0] = c.is_certified ? 1.0 : 0.0;
    out[1] = c.theta;
    out[2] = static_cast<double>(c.num_iters);
    if (x) std::memcpy(x, c.x.data(), sizeof(double) * static_cast<size_t>(c.x.size()));
  });
}

int cora_problem_solve(cora_problem *p, const double *x0, int max_rank, int verbose, const double *opts,
                       double *x_out, double stats[11]) {
  return guarded([&] {
    Problem &q = p->problem;
    const Index N = q.getExpectedVariableSize(), r = static_cast<Index>(q.getRelaxationRank());
    TNTParams prm;
    if (opts) {
      if (opts[0] > 0) prm.max_iterations = static_cast<int>(opts[0]);
      if (opts[1] > 0) prm.max_TPCG_iterations = static_cast<int>(opts[1]);
      if (opts[2] > 0) prm.gradient_tolerance = opts[2];
      if (opts[3] > 0) prm.preconditioned_gradient_tolerance = opts[3];
      if (opts[4] > 0) prm.max_computation_time = opts[4];
    }
    CoraSolveInfo info;
    const auto t0 = std::chrono::steady_clock::now();
    const CoraResult res = solveCORA(q, wrap(x0, N, r), max_rank, verbose != 0, false, false, &info, &prm);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::memcpy(x_out, res.first.x.data(), sizeof(double) * static_cast<size_t>(res.first.x.size()));
    stats[0] = res.first.f;
    stats[1] = res.first.gradfx_norm;
    stats[2] = info.certified ? 1.0 : 0.0;
    stats[3] = info.eta;
    stats[4] = info.theta;
    stats[5] = info.final_rank;
    stats[6] = info.staircase_levels;
    stats[7] = static_cast<double>(info.hessian_vector_products);
    stats[8] = secs;
    stats[9] = info.relaxation_certified ? 1.0 : 0.0;
    stats[10] = info.relaxation_rank;
  });
}

int cora_problem_precond_info(cora_problem *p, double info[3]) {
  return guarded([&] {
    p->problem.ensurePreconditionerReady();
    info[0] = p->problem.preconditionerLambda();
    info[1] = static_cast<double>(p->problem.preconditionerNnz());
    info[2] = p->problem.preconditionerLevels();
  });
}

int cora_problem_cholesky_solve(cora_problem *p, int m, double shift, int leaf_poses, double *B, int k,
                                int64_t info[3]) {
  return guarded([&] {
    Problem &q = p->problem;
    const SparseMatrix &Q = q.getDataMatrix();
    const auto perm = coraOrdering(q.dim(), q.numPoses(), q.numRangeMeasurements(), q.numTranslationalStates(),
                                   Q, m, leaf_poses > 0 ? leaf_poses : 16);
    const CholeskyFactor F = choleskyFactor(Q, m, shift, perm, q.symbolicCache());
    info[0] = F.ok ? 1 : 0;
    info[1] = F.nnz();
    int height = 0;
    if (F.ok) {
      std::vector<int> depth(static_cast<size_t>(F.n), 1);
      for (int i = 0; i < F.n; ++i)
        if (F.parent[i] >= 0) depth[F.parent[i]] = std::max(depth[F.parent[i]], depth[i] + 1);
      for (int v : depth) height = std::max(height, v);
      Matrix Bm(m, k);
      std::memcpy(Bm.data(), B, sizeof(double) * static_cast<size_t>(m) * k);
      F.solveInPlace(Bm);
      std::memcpy(B, Bm.data(), sizeof(double) * static_cast<size_t>(m) * k);
    }
    info[2] = height;
  });
}

int cora_problem_cholesky_probe_bumped(cora_problem *p, int m, double shift, int leaf_poses, int nbump,
                                       const int32_t *bump_rows, const double *bump_vals, int64_t info[3],
                                       double *digest, double *negative_direction) {
  return guarded([&] {
    Problem &q = p->problem;
    const SparseMatrix &Q0 = q.getDataMatrix();
    const auto perm = coraOrdering(q.dim(), q.numPoses(), q.numRangeMeasurements(), q.numTranslationalStates(),
                                   Q0, m, leaf_poses > 0 ? leaf_poses : 16);
    SparseMatrix bumped;
    if (nbump > 0) {  // a copy of Q with some diagonal entries moved (a pivot that turns negative where the test wants it)
      bumped = Q0;
      for (int b = 0; b < nbump; ++b) {
        const int32_t r = bump_rows[b];
        if (r < 0 || r >= bumped.rows()) throw std::invalid_argument("cholesky_probe: bump row out of range");
        bool found = false;
        for (int32_t e = bumped.outer[r]; e < bumped.outer[r + 1]; ++e)
          if (bumped.inner[e] == r) { bumped.values[e] += bump_vals[b]; found = true; }
        if (!found) throw std::invalid_argument("cholesky_probe: no diagonal entry in the bumped row");
      }
    }
    const SparseMatrix &Q = nbump > 0 ? bumped : Q0;
    const CholeskyFactor F = choleskyFactor(Q, m, shift, perm, q.symbolicCache());
    info[0] = F.ok ? 1 : 0;
    info[1] = F.nnz();
    info[2] = F.failed_column;
    double d0 = 0.0, d1 = 0.0;  // order-dependent sums over the columns that are complete (all of them when ok)
    const int upto = F.ok ? F.n : F.failed_column;
    for (int j = 0; j < upto; ++j)
      for (int32_t e = F.Lp[j]; e < F.Lp[j + 1]; ++e)
        if (F.ok || F.Li[e] < upto) {
          d0 += F.Lx[e] * static_cast<double>(1 + (e % 7));
          d1 += static_cast<double>(F.Li[e] % 1009) * F.Lx[e];
        }
    digest[0] = d0;
    digest[1] = d1;
    if (negative_direction && !F.ok)
      std::memcpy(negative_direction, F.negative_direction.data(), sizeof(double) * F.negative_direction.size());
  });
}

int cora_problem_cholesky_probe(cora_problem *p, int m, double shift, int leaf_poses, int64_t info[3], double *digest,
                                double *negative_direction) {
  return cora_problem_cholesky_probe_bumped(p, m, shift, leaf_poses, 0, nullptr, nullptr, info, digest, negative_direction);
}

int cora_problem_plan_probe(cora_problem *p, double shift, int leaf_poses, int64_t info[4]) {
  return guarded([&] {
    Problem &q = p->problem;
    const SparseMatrix &Q = q.getDataMatrix();
    const int N = static_cast<int>(Q.rows()), m = N - 1;
    const auto perm = coraOrdering(q.dim(), q.numPoses(), q.numRangeMeasurements(), q.numTranslationalStates(), Q, m,
                                   leaf_poses > 0 ? leaf_poses : 2);
    const CholeskyFactor F = choleskyFactor(Q, m, shift, perm, q.symbolicCache());
    if (!F.ok) throw std::runtime_error("plan probe: the factorisation failed");
    // rows in API order stand in for the handle's internal order (the probe is about the builder's time and the
    // plan's shape, not about a particular handle); pose groups as capi.hip forms them
    std::vector<int32_t> row_of(perm.begin(), perm.end()), group(static_cast<size_t>(m), -1);
    const int64_t dn = static_cast<int64_t>(q.dim()) * q.numPoses();
    for (int i = 0; i < m; ++i)
      if (perm[i] < dn) group[i] = perm[i] / q.dim();
    cora::TriPlan P;
    cora::build_tri_plan(m, F.Lp.data(), F.Li.data(), F.Lx.data(), row_of, N - 1, P, &group, N);
    info[0] = static_cast<int64_t>(P.stages.size());
    info[1] = P.nnzL;
    info[2] = P.stages.empty() ? 0 : (P.stages[0].sub ? static_cast<int64_t>(P.stages[0].sub_op.nrows.size()) : 0);
    info[3] = static_cast<int64_t>(P.top_rows.size());
  });
}

int cora_host_block_cholesky_solve(int n, const int32_t *rowptr, const int32_t *colidx, const double *vals, int nblocks,
                                   const int32_t *block_sizes, int rhs_rows, int k, const double *B, double *X) {
  return guarded([&] {
    SparseMatrix A(n, n);
    A.outer.assign(rowptr, rowptr + n + 1);
    A.inner.assign(colidx, colidx + rowptr[n]);
    A.values.assign(vals, vals + rowptr[n]);
    const CholFactorPtrVector F = getBlockCholeskyFactorization(A, std::vector<int>(block_sizes, block_sizes + nblocks));
    const Matrix x = blockCholeskySolve(F, wrap(B, rhs_rows, k));
    std::memcpy(X, x.data(), sizeof(double) * static_cast<size_t>(x.size()));
  });
}

int cora_host_manifold_op(int kind, int k, int p, int n, const char *op, const double *A, const double *B,
                          uint64_t seed, double *out) {
  return guarded([&] {
    const std::string o(op);
    std::unique_ptr<MatrixManifold> M;
    Index cols;
    StiefelProduct *st = nullptr;
    ObliqueManifold *ob = nullptr;
    if (kind == 0) { st = new StiefelProduct(k, p, n); M.reset(st); cols = static_cast<Index>(k) * n; }
    else { ob = new ObliqueManifold(p, n); M.reset(ob); cols = n; }
    Matrix res;
    if (o == "projectToManifold") res = M->projectToManifold(wrap(A, p, cols));
    else if (o == "projectToTangentSpace") res = M->projectToTangentSpace(wrap(A, p, cols), wrap(B, p, cols));
    else if (o == "retract") res = M->retract(wrap(A, p, cols), wrap(B, p, cols));
    else if (o == "random_sample") res = st ? st->random_sample(static_cast<unsigned>(seed)) : ob->random_sample(static_cast<unsigned>(seed));
    else if (o == "innerProduct") { out[0] = M->innerProduct(wrap(A, p, cols), wrap(B, p, cols)); return; }
    else if (o == "SymBlockDiagProduct" && st) res = st->SymBlockDiagProduct(wrap(A, p, cols), wrap(B, p, cols).transpose(), wrap(B + static_cast<size_t>(p) * cols, p, cols));
    else throw std::invalid_argument("unknown manifold operation " + o);
    std::memcpy(out, res.data(), sizeof(double) * static_cast<size_t>(res.size()));
  });
}

int cora_problem_print(cora_problem *p) {
  return guarded([&] { p->problem.printProblem(); });
}

int cora_problem_save_trajectory(cora_problem *p, const double *X, int g2o, int robot_chr, const char *path) {
  return guarded([&] {
    Problem &q = p->problem;
    const Matrix soln = wrap(X, q.getDataMatrixSize(), q.dim());
    if (robot_chr > 0) {
      const auto syms = q.getPoseSymbols(static_cast<unsigned char>(robot_chr));
      if (g2o) saveSolnToG20(syms, q, soln, path);
      else saveSolnToTum(syms, q, soln, path);
    } else {
      if (g2o) saveSolnToG20(q, soln, path);
      else saveSolnToTum(q, soln, path);
    }
  });
}

void *cora_problem_context(cora_problem *p) {
  void *c = nullptr;
  guarded([&] { c = p->problem.context(); });
  return c;
}

}  // extern "C"
