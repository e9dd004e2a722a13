// Drop-in check (tests/test_dropin_build.py): the call sequence of the reference's examples/main.cpp:17-27 written
// against the reference's include layout (<CORA/...>), compiled and linked against this build.
#include <CORA/CORA.h>
#include <CORA/CORA_problem.h>
#include <CORA/CORA_types.h>
#include <CORA/pyfg_text_parser.h>

int main(int argc, char **argv) {
  if (argc != 2) {
    std::cout << "Usage: " << argv[0] << " [input .pyfg file]" << std::endl;
    return 1;
  }
  CORA::Problem problem = CORA::parsePyfgTextToProblem(argv[1]);
  problem.updateProblemData();
  CORA::Matrix x0 = problem.getRandomInitialGuess();
  int max_rank = 10;
  CORA::CoraResult soln = CORA::solveCORA(problem, x0, max_rank);
  CORA::Matrix aligned_soln = problem.alignEstimateToOrigin(soln.first.x);
  std::cout << "cost " << soln.first.f << " rows " << aligned_soln.rows() << std::endl;
  return 0;
}
