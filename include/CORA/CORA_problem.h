// <CORA/CORA_problem.h> of the reference (MarineRoboticsGroup/cora, include/CORA/CORA_problem.h): CORA::Problem.
// Forwarding header: code written against the reference's include layout compiles against this build with
// -I<repo>/include and links libcora_hip.so.
#pragma once
#include "../../cora_amd/csrc/host/CORA_problem.h"
