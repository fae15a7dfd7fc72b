#!/bin/bash
# same-box sweep of two runtime knobs on the final build (bench.py --steps 20 --warmup 4, shipped tuning database), two alternating rounds
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
DB=/tmp/shipped_copy.db; cp packnet-sfm_amd/csrc/tuned_gfx950.db $DB
BARGS="--no-cpu-baseline --no-extra --gpu-baseline off"
S='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
for i in 1 2; do
  for v in 1 130000 40000; do
    PNSFM_WGRAD_STREAM=$v PNSFM_TUNE_DB=$DB timeout 300 python bench.py --steps 20 --warmup 4 $BARGS --no-prof > $O/knob.log 2>&1
    echo "PNSFM_WGRAD_STREAM=$v: $(tail -1 $O/knob.log | python -c "$S")"
  done
  for v in 1 0.5 2; do
    PNSFM_BLOCK_MAP_WEIGHTS=$v PNSFM_TUNE_DB=$DB timeout 300 python bench.py --steps 20 --warmup 4 $BARGS --no-prof > $O/knob.log 2>&1
    echo "PNSFM_BLOCK_MAP_WEIGHTS=$v: $(tail -1 $O/knob.log | python -c "$S")"
  done
done | tee $O/r05_ab_knobs.txt
