#!/bin/bash
# Round-2 GPU session: parity tests, bench variants (384x1280, forced-DDP, 2-rank rehearsal), the autotuner's full candidate
# landscape (PNSFM_TUNE_LOG), rocprofv3 kernel traces and PMC passes.  PNSFM_TUNE_DB is set to gpurun_out/tune_<tag>.db: a fresh tag
# tunes from scratch (that is how csrc/tuned_gfx950.db is regenerated: `gpu_lab.sh <tag> bench bench_c3`, then copy the file).  usage: tools/gpu_lab.sh <tag> [what...]
TAG=${1:-r02a}; shift
WHAT=${@:-tests bench bench_eager bench_c3 bench_2rank prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
export PNSFM_TUNE_DB=$O/tune_$TAG.db
for w in $WHAT; do case $w in
tests)
  echo "== pytest (selected)"
  timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 -x \
    -k "${PYTEST_K:-conv2d_vs_cpu_oracle or tap_major or accumulation or trainer_fit or 384x1280 or training_step_golden or groupnorm}" \
    > $O/pytest_$TAG.log 2>&1
  tail -15 $O/pytest_$TAG.log ;;
alltests)
  echo "== pytest -m gpu (all)"
  timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_all_$TAG.log 2>&1
  tail -15 $O/pytest_all_$TAG.log ;;
bench_ddp1)
  echo "== bench, 1 rank but the DDP path forced (reducer over the FlatAdam arenas, RCCL world of 1)"
  PNSFM_FORCE_DDP=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof > $O/bench_ddp1_$TAG.log 2>&1
  tail -1 $O/bench_ddp1_$TAG.log | cut -c1-700 ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; tail -3 $O/smoke_$TAG.log ;;
bench)
  echo "== bench (defaults: eager launches, FlatAdam), tune log on"
  PNSFM_TUNE_LOG=$O/tunelog_$TAG.txt timeout 900 python bench.py --steps 10 --warmup 3 --layer-table $O/layers_$TAG.csv > $O/bench_$TAG.log 2>&1
  tail -2 $O/bench_$TAG.log | cut -c1-1800 ;;
bench_eager)
  echo "== bench"
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_eager_$TAG.log 2>&1
  tail -1 $O/bench_eager_$TAG.log | cut -c1-600 ;;
bench_c3)
  echo "== bench 384x1280 batch 2 (configs[2] shape)"
  PNSFM_TUNE_LOG=$O/tunelog_c3_$TAG.txt timeout 900 python bench.py --height 384 --width 1280 --batch 2 --steps 6 --warmup 2 --no-cpu-baseline \
     --layer-table $O/layers_c3_$TAG.csv > $O/bench_c3_$TAG.log 2>&1
  tail -1 $O/bench_c3_$TAG.log | cut -c1-1500 ;;
bench_2rank)
  echo "== bench --gpus 2 on this box (ranks share the device: functional rehearsal)"
  timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/bench_2rank_$TAG.log 2>&1
  tail -3 $O/bench_2rank_$TAG.log | cut -c1-900 ;;
prof)
  echo "== rocprofv3 kernel trace of bench.py (tuning database primed: no autotune candidates in the trace)"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- \
      python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/rocprof_$TAG.log 2>&1)
  tail -1 $O/rocprof_$TAG.log | cut -c1-400
  f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 $f | cut -c1-160 ;;
prof_iso)
  echo "== rocprofv3 kernel trace, weight-gradient side stream OFF (kernels run alone: the 'isolated' roofline)"
  (cd /tmp && PNSFM_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_iso_$TAG -o bench -- \
      python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/rocprof_iso_$TAG.log 2>&1)
  tail -1 $O/rocprof_iso_$TAG.log | cut -c1-400 ;;
prof_eager)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_eager_$TAG -o bench -- \
      python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/rocprof_eager_$TAG.log 2>&1)
  tail -1 $O/rocprof_eager_$TAG.log | cut -c1-400 ;;
pmc)
  echo "== rocprofv3 PMC passes (counters only, separate runs)"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-24)
    (cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${TAG}_$n -o bench -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $O/pmc_${TAG}_$n.log 2>&1)
    tail -1 $O/pmc_${TAG}_$n.log | cut -c1-300
  done ;;
esac; done
