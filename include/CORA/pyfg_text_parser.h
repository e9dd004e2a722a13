// <CORA/pyfg_text_parser.h> of the reference (MarineRoboticsGroup/cora, include/CORA/pyfg_text_parser.h): parsePyfgTextToProblem.
// Forwarding header: code written against the reference's include layout compiles against this build with
// -I<repo>/include and links libcora_hip.so.
#pragma once
#include "../../cora_amd/csrc/host/pyfg_text_parser.h"
