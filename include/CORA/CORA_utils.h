// <CORA/CORA_utils.h> of the reference (MarineRoboticsGroup/cora, include/CORA/CORA_utils.h): fast_verification, projectToSOd; the TUM / g2o writers live in io.h.
// Forwarding header: code written against the reference's include layout compiles against this build with
// -I<repo>/include and links libcora_hip.so.
#pragma once
#include "../../cora_amd/csrc/host/CORA_utils.h"
#include "../../cora_amd/csrc/host/io.h"
