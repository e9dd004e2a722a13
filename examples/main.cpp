// Command-line driver with the flow of the reference's examples/main.cpp:
//   parse PyFG -> updateProblemData -> random initial guess -> solveCORA(max_rank 10)
//   -> alignEstimateToOrigin, then print the result and optionally save a TUM trajectory.
// Build:  hipcc -O2 -std=c++17 -Iinclude -Icora_amd/csrc/host examples/main.cpp \
//               -Lcora_amd/lib -lcora_hip -Wl,-rpath,$PWD/cora_amd/lib -o cora_main
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include "CORA.h"
#include "io.h"
#include "odometry_init.h"

int main(int argc, char **argv) {
  if (argc < 2) {
    std::cout << "Usage: " << argv[0] << " [input .pyfg file] [--jacobi] [--implicit] [--odom-init] [--tum out.tum] [--save-dir dir] [--max-rank r]" << std::endl;
    return 1;
  }
  int max_rank = 10;
  std::string tum, save_dir;
  bool jacobi = false, implicit = false, odom = false;
  for (int i = 2; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--jacobi") jacobi = true;
    else if (a == "--implicit") implicit = true;   // "formulation": "Implicit" of examples/config.json
    else if (a == "--odom-init") odom = true;      // "init_type": "Odom"
    else if (a == "--tum" && i + 1 < argc) tum = argv[++i];
    else if (a == "--save-dir" && i + 1 < argc) save_dir = argv[++i];  // cora_<robot>.tum / .g2o per robot, like saveSolutions
    else if (a == "--max-rank" && i + 1 < argc) max_rank = std::atoi(argv[++i]);
  }
  using clk = std::chrono::steady_clock;
  auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  try {
    const auto t_start = clk::now();
    CORA::Problem problem = CORA::parsePyfgTextToProblem(argv[1]);
    const auto t_parsed = clk::now();
    if (jacobi) problem.setPreconditioner(CORA::Preconditioner::Jacobi);
    if (implicit) problem.setFormulation(CORA::Formulation::Implicit);
    problem.updateProblemData();
    const auto t_assembled = clk::now();
    problem.ensurePreconditionerReady();  // device copy of Q, Cholesky factor, solve plan
    const auto t_device = clk::now();
    std::printf("poses %d  landmarks %d  ranges %d  N %d  nnz(Q) %ld\n", problem.numPoses(), problem.numLandmarks(),
                problem.numRangeMeasurements(), problem.getDataMatrixSize(),
                static_cast<long>(problem.data_matrix_.nonZeros()));
    CORA::Matrix x0 = odom ? CORA::getOdomInitialization(problem) : problem.getRandomInitialGuess();
    if (odom && implicit)  // examples/paper_experiments.cpp:623-625
      x0 = x0.block(0, 0, problem.rotAndRangeMatrixSize(), x0.cols());
    CORA::CoraSolveInfo info;
    const CORA::CoraResult soln = CORA::solveCORA(problem, x0, max_rank, /*verbose=*/true, false, false, &info);
    const auto t_solved = clk::now();
    const CORA::Matrix aligned = problem.alignEstimateToOrigin(soln.first.x);
    std::printf("final cost %.9g  |grad| %.3e  certified %d  theta %.3e  staircase levels %d  Hvps %ld  %.3f s\n",
                soln.first.f, soln.first.gradfx_norm, static_cast<int>(info.certified), info.theta,
                info.staircase_levels, info.hessian_vector_products, soln.first.elapsed_time);
    std::printf("wall clock: parse %.3f s | assemble Q %.3f s | device setup + preconditioner %.3f s | solve %.3f s "
                "(TNT %.3f, certification %.3f, saddle escape %.3f)\n",
                secs(t_start, t_parsed), secs(t_parsed, t_assembled), secs(t_assembled, t_device),
                secs(t_device, t_solved), info.tnt_seconds, info.certify_seconds, info.escape_seconds);
    if (!tum.empty()) {
      CORA::saveSolnToTum(problem, aligned, tum);
      std::cout << "wrote " << tum << std::endl;
    }
    if (!save_dir.empty()) {  // examples/paper_experiments.cpp:536-592: one trajectory file pair per robot chain
      std::vector<unsigned char> robots;
      for (const auto &kv : problem.getPoseSymbolMap())
        if (!(kv.first == problem.getOriginSymbol()) &&
            std::find(robots.begin(), robots.end(), kv.first.chr()) == robots.end())
          robots.push_back(kv.first.chr());
      for (size_t r = 0; r < robots.size(); ++r) {
        const auto chain = problem.getPoseSymbols(robots[r]);
        const std::string base = save_dir + "/cora_" + std::to_string(r);
        CORA::saveSolnToTum(chain, problem, aligned, base + ".tum");
        CORA::saveSolnToG20(chain, problem, aligned, base + ".g2o");
        std::cout << "wrote " << base << ".tum / .g2o (" << chain.size() << " poses of robot '" << robots[r] << "')" << std::endl;
      }
    }
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 2;
  }
  return 0;
}
