// Developer tool (CPU only): stage statistics of the device Cholesky-solve plan, and a host
// emulation of the staged products checked against CholeskyFactor::solveInPlace.
// hipcc -O2 -std=c++17 -Iinclude -Icora_amd/csrc -Icora_amd/csrc/host tools/tri_stats.cpp -Lcora_amd/lib -lcora_hip -Wl,-rpath,$PWD/cora_amd/lib -o tools/bin/tri_stats
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "sparse_cholesky.h"
#include "synthetic.h"
#include "trisolve.h"
using cora::RowOpHost;
static void stat(const char *name, const RowOpHost &op) {
  std::printf("    %-6s rows8 %7d rows64 %6d long %3zu chunks %5zu entries %9zu\n", name, op.n8, op.n64, op.long_out.size(),
              op.chunk_begin.size(), op.col.size());
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 100000, leaf = argc > 2 ? std::atoi(argv[2]) : 8;
  CORA::SyntheticSpec sp;
  sp.num_poses = n; sp.num_ranges = n / 2; sp.num_landmarks = 10; sp.num_loop_closures = argc > 3 ? std::atoi(argv[3]) : 0;
  CORA::Problem P = CORA::makeSyntheticProblem(sp);
  P.updateProblemData();
  const int N = P.getDataMatrixSize(), m = N - 1;
  const auto perm = CORA::coraOrdering(3, P.numPoses(), P.numRangeMeasurements(), P.numTranslationalStates(), P.data_matrix_, m, leaf);
  auto t0 = std::chrono::steady_clock::now();
  const CORA::CholeskyFactor F = CORA::choleskyFactor(P.data_matrix_, m, 9.4e3, perm);
  auto t1 = std::chrono::steady_clock::now();
  std::vector<int32_t> row_of(perm.begin(), perm.end());  // internal row = original row here
  cora::TriPlan plan;
  cora::build_tri_plan(m, F.Lp.data(), F.Li.data(), F.Lx.data(), row_of, N - 1, plan, nullptr, N);
  auto t2 = std::chrono::steady_clock::now();
  std::printf("factor %.3f s, plan %.3f s, nnz(L) %ld nnz(W) %ld stages %zu\n", std::chrono::duration<double>(t1 - t0).count(),
              std::chrono::duration<double>(t2 - t1).count(), (long)plan.nnzL, (long)plan.nnzW, plan.stages.size());
  for (size_t k = 0; k < plan.stages.size(); ++k) {
    const auto &S = plan.stages[k];
    std::printf("  stage %zu: rows %d blocks %d\n", k, S.rows, S.blocks);
    if (S.sub) {
      const auto &B = S.sub_op;
      std::printf("    sub: %zu blocks (max %d rows, %d entries, %d headers), fwd entries %zu, bwd entries %zu, ext %zu, targets %zu (aux rows %d), headers fwd %zu bwd %zu\n",
                  B.nrows.size(), B.max_rows, B.max_ent, B.max_lev, B.f_val.size(), B.b_val.size(), B.tgt_row.size(), B.tgt_slot.size(), B.n_aux,
                  B.f_hdr.size() / 4, B.b_hdr.size() / 4);
      for (size_t b : {size_t(0), B.nrows.size() / 2}) {
        for (int dir = 0; dir < 2; ++dir) {
          const auto &H = dir ? B.b_hdr : B.f_hdr;
          const auto &LB = dir ? B.b_lev_begin : B.f_lev_begin;
          std::printf("    block %zu %s levels (rows x g x npl):", b, dir ? "bwd" : "fwd");
          for (int l = LB[b]; l + 1 < LB[b + 1]; ++l) std::printf(" %dx%dx%d", H[4 * (l + 1)] - H[4 * l], H[4 * l + 1] & 0xff, (H[4 * l + 1] >> 8) & 0xf);
          std::printf("\n");
        }
      }
      continue;
    }
    if (S.dense) {
      std::printf("    dense: %zu blocks, %zu stored entries, %zu ext entries\n", S.blocks_op.nrows.size(), S.blocks_op.w_by_col.size(), S.blocks_op.ext_col.size());
      continue;
    }
    if (k > 0) stat("fwd_a", S.fwd_a);
    stat("fwd_b", S.fwd_b);
    if (k + 1 < plan.stages.size()) stat("bwd_a", S.bwd_a);
    stat("bwd_b", S.bwd_b);
  }
  // emulate
  std::vector<double> rhs(N, 0.0), out(N, 7.0);
  rhs[N - 1] = 3.0;
  for (int i = 0; i < m; ++i) rhs[i] = std::sin(0.37 * i) + 0.1;
  cora::tri_plan_solve_host(plan, N, rhs.data(), out.data());
  CORA::Matrix B(m, 1);
  for (int i = 0; i < m; ++i) B(i, 0) = rhs[i];
  F.solveInPlace(B);
  double err = 0, nrm = 0;
  for (int i = 0; i < m; ++i) { err = std::max(err, std::fabs(B(i, 0) - out[i])); nrm = std::max(nrm, std::fabs(B(i, 0))); }
  std::printf("staged vs direct solve: max err %.3e (max |x| %.3e), pinned row %.1e\n", err, nrm, out[N - 1]);
  return 0;
}
