#!/bin/bash
# Row classes of the top stage's products against the STPCG iteration (10^5 poses).  bash tools/plan_sweep2.sh
run() { echo "== $*"; env "$@" python bench.py --steps 20 --warmup 5 --cpu-seconds 1 --pmc-traffic off 2>/dev/null > /tmp/ps.json; python tools/benchsum.py /tmp/ps.json | cut -d'|' -f5-; }
run CORA_DUMMY=1
run CORA_TRI_WAVE_ROW=512
run CORA_TRI_WAVE_ROW=512 CORA_TRI_CHUNK=256
run CORA_TRI_CHUNK=1024
run CORA_TRI_CHUNK=256
run CORA_TRI_SHORT_ROW=32
run CORA_TRI_SHORT_ROW=128
run CORA_TRI_WAVE_ROW=2048 CORA_TRI_CHUNK=1024
