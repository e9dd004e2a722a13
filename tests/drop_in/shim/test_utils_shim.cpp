// Implementation of the test_utils.h stand-in (see the header).  Written for the Eigen-free types; the Matrix Market
// reader follows what the reference's helper does with Eigen::loadMarket: "symmetric" files hold the lower triangle and
// are mirrored (tests/test_utils.cpp:22-51).
#include <test_utils.h>

#include <filesystem>
#include <fstream>
#include <map>
#include <stdexcept>

namespace CORA {

SparseMatrix readMatrixMarketFile(const std::string &filename) {
  std::ifstream file(filename);
  if (!file.is_open()) throw std::runtime_error("Could not open file");
  std::string line;
  if (!std::getline(file, line)) throw std::runtime_error("Could not read first line of file");
  const bool symmetric = line.find("symmetric") != std::string::npos;
  const bool array = line.find("array") != std::string::npos;
  while (std::getline(file, line))
    if (!line.empty() && line[0] != '%') break;
  std::istringstream head(line);
  long rows = 0, cols = 0, nnz = 0;
  head >> rows >> cols;
  std::vector<Triplet> t;
  if (array) {  // dense, column by column
    for (long j = 0; j < cols; ++j)
      for (long i = 0; i < rows; ++i) {
        double v = 0;
        file >> v;
        if (v != 0.0) t.push_back({i, j, v});
      }
  } else {
    head >> nnz;
    for (long k = 0; k < nnz; ++k) {
      long i = 0, j = 0;
      double v = 0;
      file >> i >> j >> v;
      t.push_back({i - 1, j - 1, v});
      if (symmetric && i != j) t.push_back({j - 1, i - 1, v});
    }
  }
  SparseMatrix A(rows, cols);
  A.setFromTriplets(std::move(t));
  return A;
}

std::string getTestDataFpath(const std::string &data_subdir, const std::string &fname) {
  const std::string filepath = std::filesystem::current_path() / "./bin/data" / data_subdir / fname;
  if (!std::filesystem::exists(filepath))
    throw std::runtime_error("File does not exist: " + filepath +
                             "\nThis may be because you are running the tests from the wrong directory. We expect to be "
                             "running from <repo_root>/build");
  return filepath;
}

std::string checkSubmatricesAreCorrect(Problem prob, const std::string &data_subdir) {
  const CoraDataSubmatrices data_submatrices = prob.getDataSubmatrices();
  std::string error_msg;
  const std::map<std::string, SparseMatrix> submatrices = {
      {"Arange.mm", data_submatrices.range_incidence_matrix},
      {"OmegaRange.mm", data_submatrices.range_precision_matrix},
      {"RangeDistances.mm", data_submatrices.range_dist_matrix},
      {"Apose.mm", data_submatrices.rel_pose_incidence_matrix},
      {"OmegaPose.mm", data_submatrices.rel_pose_translation_precision_matrix},
      {"T.mm", data_submatrices.rel_pose_translation_data_matrix},
      {"RotConLaplacian.mm", data_submatrices.rotation_conn_laplacian},
      {"DataMatrix.mm", prob.getDataMatrix()}};
  for (const auto &submatrix : submatrices) {
    const SparseMatrix expected = readMatrixMarketFile(getTestDataFpath(data_subdir, submatrix.first));
    const SparseMatrix &actual = submatrix.second;
    if (expected.rows() == expected.cols() && expected.rows() == 0) {
      if (actual.rows() != 0 && actual.cols() != 0)
        error_msg += "Submatrix " + submatrix.first + " has " + std::to_string(actual.rows()) + " rows and " +
                     std::to_string(actual.cols()) + " cols but should have 0 rows or 0 cols\n";
      continue;
    }
    if (expected.rows() != actual.rows()) {
      error_msg += "Submatrix " + submatrix.first + " has " + std::to_string(actual.rows()) + " rows but should have " +
                   std::to_string(expected.rows()) + "\n";
    } else if (expected.cols() != actual.cols()) {
      error_msg += "Submatrix " + submatrix.first + " has " + std::to_string(actual.cols()) + " cols but should have " +
                   std::to_string(expected.cols()) + "\n";
    } else {
      // Eigen's isApprox: |a - b|_F^2 <= prec^2 min(|a|_F^2, |b|_F^2), prec = 1e-12
      SparseMatrix neg = actual;
      for (auto &v : neg.values) v = -v;
      const SparseMatrix diff = expected.plus(neg);
      double d2 = 0, a2 = 0, b2 = 0;
      for (double v : diff.values) d2 += v * v;
      for (double v : expected.values) a2 += v * v;
      for (double v : actual.values) b2 += v * v;
      if (!(d2 <= 1e-24 * std::min(a2, b2))) error_msg += "Submatrix " + submatrix.first + " is incorrect\n";
    }
  }
  return error_msg;
}

Problem getProblem(std::string data_subdir) { return parsePyfgTextToProblem(getTestDataFpath(data_subdir, "factor_graph.pyfg")); }
Matrix getRandInit(std::string data_subdir) { return readMatrixMarketFile(getTestDataFpath(data_subdir, "X_rand_dim2.mm")).toDense(); }
Matrix getGroundTruthState(std::string data_subdir) { return readMatrixMarketFile(getTestDataFpath(data_subdir, "X_gt.mm")).toDense(); }
Matrix getRandDX(std::string data_subdir) { return readMatrixMarketFile(getTestDataFpath(data_subdir, "rand_dX.mm")).toDense(); }
SparseMatrix getExpectedRandCertMatrix(std::string data_subdir) { return readMatrixMarketFile(getTestDataFpath(data_subdir, "S_rand.mm")); }

Scalar getExpectedCost(std::string data_subdir) {  // the reference's known answers, tests/test_utils.cpp:208-222
  if (data_subdir == "small_ra_slam_problem") return 1.063888372855624e+03;
  if (data_subdir == "single_rpm") return 0.809173848024762;
  if (data_subdir == "single_range") return 4.718031199983851;
  throw std::runtime_error("Do not have expected cost for: " + data_subdir);
}
Matrix getExpectedEgrad(std::string data_subdir) { return readMatrixMarketFile(getTestDataFpath(data_subdir, "expected_egrad.mm")).toDense(); }
Matrix getExpectedRgrad(std::string data_subdir) { return readMatrixMarketFile(getTestDataFpath(data_subdir, "expected_rgrad.mm")).toDense(); }
Matrix getExpectedHessProd(std::string data_subdir) { return readMatrixMarketFile(getTestDataFpath(data_subdir, "hessProd.mm")).toDense(); }

}  // namespace CORA
