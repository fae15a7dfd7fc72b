#!/bin/bash
# One GPU session on the MI355X box: parity tests, smoke, bench, rocprofv3 kernel trace (+ optional PMC passes).
# usage: tools/gpu_session.sh [tag] [what...]   what in: tests smoke bench prof pmc   (default: all but pmc)
TAG=${1:-r01}; shift
WHAT=${@:-tests smoke bench prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd $R
for w in $WHAT; do case $w in
tests)
  echo "== pytest -m gpu"
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > gpurun_out/pytest_gpu_$TAG.log 2>&1
  tail -25 gpurun_out/pytest_gpu_$TAG.log ;;
smoke)
  echo "== smoke"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; tail -4 gpurun_out/smoke_$TAG.log ;;
bench)
  echo "== bench"
  timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1; tail -4 gpurun_out/bench_$TAG.log ;;
prof)
  echo "== rocprofv3 kernel trace (tuning database primed first, so the trace holds no autotune candidates)"
  export PNSFM_TUNE_DB=$R/gpurun_out/tune_$TAG.db
  [ -s $PNSFM_TUNE_DB ] || timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
  wc -l $PNSFM_TUNE_DB
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o bench -- \
      python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-prof > $R/gpurun_out/rocprof_$TAG.log 2>&1)
  tail -2 gpurun_out/rocprof_$TAG.log; find gpurun_out/prof_$TAG -name "*stats*" | head
  f=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 $f ;;
pmc)
  echo "== rocprofv3 PMC passes (counters only, separate runs)"
  export PNSFM_TUNE_DB=$R/gpurun_out/tune_$TAG.db
  [ -s $PNSFM_TUNE_DB ] || timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > /dev/null 2>&1
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-24)
    (cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_$n -o bench -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1)
    tail -1 gpurun_out/pmc_${TAG}_$n.log
  done ;;
esac; done
