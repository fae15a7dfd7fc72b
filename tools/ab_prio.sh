#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
DB=/tmp/shipped_copy.db; cp packnet-sfm_amd/csrc/tuned_gfx950.db $DB
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
BARGS="--no-cpu-baseline --no-extra --gpu-baseline off --no-prof"
S='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for i in 1 2; do
  for v in "" -1; do
    PNSFM_MAIN_PRIORITY=$v PNSFM_TUNE_DB=$DB timeout 300 python bench.py --steps 20 --warmup 4 $BARGS > $O/knob.log 2>&1
    echo "PNSFM_MAIN_PRIORITY=$v (compute stream; side streams at 0): $(tail -1 $O/knob.log | python -c "$S" 2>&1 | tail -1)"
  done
done | tee $O/r05_ab_main_priority.txt
