// PyFG text ingestion (reference include/CORA/pyfg_text_parser.h:31,
// src/pyfg_text_parser.cpp:112-401).
#pragma once

#include <string>

#include "CORA_problem.h"

namespace CORA {

int getDimFromPyfgFirstLine(const std::string &filename);

/** Parses a PyFG text file into a Problem (defaults as in the reference:
 * Explicit formulation, RegularizedCholesky preconditioner, rank = dim). */
Problem parsePyfgTextToProblem(const std::string &filename);

}  // namespace CORA
