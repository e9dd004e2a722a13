#!/bin/bash
# Where the set-up seconds of the headline problem go: format builder, solve plan, numeric factorisations, certification
# (CORA_FORMAT_TIMING / CORA_TRI_TIMING print the phases on stderr).  bash tools/setup_split.sh > gpurun_out/setup_split.txt
export CORA_FORMAT_TIMING=1 CORA_TRI_TIMING=1
python bench.py --steps 20 --warmup 5 2>&1 | grep -v "^\s*$" | cut -c1-400
