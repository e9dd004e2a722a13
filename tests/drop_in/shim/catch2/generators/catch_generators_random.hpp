// Stand-in header (test infrastructure, see catch_generators.hpp beside it).
#pragma once
#include "catch_generators.hpp"
