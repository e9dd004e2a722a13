#!/bin/bash
# Knobs of the solve plan against the STPCG iteration at the headline size: one bench line per setting.
# bash tools/plan_sweep.sh > gpurun_out/plan_sweep.txt
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 2>/dev/null > /tmp/ps.json; python tools/benchsum.py /tmp/ps.json | cut -d'|' -f5-; }
run CORA_DUMMY=1
run CORA_TRI_SUB_ROWS=256
run CORA_TRI_SUB_ROWS=384
run CORA_TRI_LANE_ENTRIES=4
run CORA_TRI_LANE_ENTRIES=6
run CORA_TRI_LEVEL_LANES=128
run CORA_ND_LEAF=1
run CORA_ND_LEAF=4
run CORA_TRI_SN_CAP=8
run CORA_TRI_TOP_INV=1000000
run CORA_TRI_TOP_INV=5000000
