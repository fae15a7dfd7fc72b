#!/bin/bash
# Round-3 GPU sessions (one gpurun call each): usage tools/gpu_lab3.sh <tag> <what...>
#   r3tests   tests/test_gpu_round3.py (parity holes + two-rank rehearsal) with -s output (the error tables)
#   quick     the round-2 GPU tests that touch the kernels changed this round
#   alltests  pytest -m gpu (everything)
#   bench     bench.py, fresh tuning database gpurun_out/tune_<tag>.db (+ tune log, layer table); no CPU baseline
#   benchfull bench.py exactly as the driver runs it (shipped database, CPU baseline, extra block)
#   ab_1x1    bench.py with PNSFM_BX3_1X1=0 (1x1 layers back on the f32 kernels), same database
#   prof      rocprofv3 --kernel-trace --stats of bench.py (database primed by `bench`)
#   pmc       FETCH_SIZE / WRITE_SIZE / MFMA-busy passes (separate runs, counters only)
#   smoke     __graft_entry__.smoke()
TAG=${1:-r03a}; shift
WHAT=${@:-r3tests quick bench prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
DB=$O/tune_$TAG.db
for w in $WHAT; do
t0=$(date +%s)
case $w in
r3tests)
  echo "== tests/test_gpu_round3.py"
  timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider --timeout 900 -s -k "${PYTEST_K:-test}" > $O/pytest_r3_$TAG.log 2>&1
  grep -E "max\|err\||inputs scaled|passed|failed|Error|error|assert|\{\"ok\"" $O/pytest_r3_$TAG.log | cut -c1-300 | tail -40 ;;
quick)
  echo "== selected round-2 GPU tests"
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_packnet_san.py -m gpu -q -p no:cacheprovider --timeout 900 \
    -k "${PYTEST_K:-groupnorm or residual or conv2d_block or conv2d_vs_cpu_oracle or wgrad_split or training_step_golden or packnet01_golden or flat_adam or dropout or san}" > $O/pytest_quick_$TAG.log 2>&1
  tail -6 $O/pytest_quick_$TAG.log | cut -c1-300 ;;
alltests)
  echo "== pytest -m gpu (all)"
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $O/pytest_all_$TAG.log 2>&1
  tail -12 $O/pytest_all_$TAG.log | cut -c1-300 ;;
sparse)
  echo "== sparse branch: tests + next-rows bench"
  timeout 900 python -m pytest tests/test_sparse.py tests/test_packnet_san.py -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest_sparse_$TAG.log 2>&1
  tail -5 $O/pytest_sparse_$TAG.log | cut -c1-300
  timeout 600 python tools/next_rows_bench.py > $O/next_rows_$TAG.json 2> $O/next_rows_$TAG.err; tail -1 $O/next_rows_$TAG.json | cut -c1-1500 ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1; tail -3 $O/smoke_$TAG.log ;;
bench)
  echo "== bench, fresh tuning database $DB"
  PNSFM_TUNE_DB=$DB PNSFM_TUNE_LOG=$O/tunelog_$TAG.txt timeout 1500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline \
     --layer-table $O/layers_$TAG.csv > $O/bench_$TAG.log 2>&1
  tail -1 $O/bench_$TAG.log | cut -c1-2500; wc -l $DB ;;
benchfull)
  echo "== bench as the driver runs it"
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/benchfull_$TAG.log 2>&1
  tail -1 $O/benchfull_$TAG.log | cut -c1-3500 ;;
ab_1x1)
  echo "== bench with 1x1 layers on the f32 kernels"
  PNSFM_BX3_1X1=0 PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-prof > $O/bench_ab1x1_$TAG.log 2>&1
  tail -1 $O/bench_ab1x1_$TAG.log | cut -c1-400 ;;
ab_cat)
  echo "== A/B: concatenations folded into the conv K loop vs torch.cat (same database, A B A B)"
  for i in 1 2; do
    PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra --no-prof > $O/bench_catA_$TAG.log 2>&1
    echo "fold   : $(tail -1 $O/bench_catA_$TAG.log | cut -c1-150)"
    PNSFM_CAT_FOLD=0 PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra --no-prof > $O/bench_catB_$TAG.log 2>&1
    echo "no fold: $(tail -1 $O/bench_catB_$TAG.log | cut -c1-150)"
  done ;;
ab_pack)
  echo "== A/B: all conv weights re-packed in one launch after the optimizer step vs lazily per layer (same database, A B A B)"
  for i in 1 2; do
    PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra --no-prof > $O/bench_packA_$TAG.log 2>&1
    echo "batched: $(tail -1 $O/bench_packA_$TAG.log | cut -c1-150)"
    PNSFM_PACK_BATCH=0 PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra --no-prof > $O/bench_packB_$TAG.log 2>&1
    echo "lazy   : $(tail -1 $O/bench_packB_$TAG.log | cut -c1-150)"
  done ;;
ab_map)
  echo "== A/B: block order of the conv kernels (0 round robin as round 2, 1 operand-sharing order, 2 + contiguous range per XCD), same database"
  for i in 1 2; do for m in 2 0 1; do
    PNSFM_BLOCK_MAP=$m PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra > $O/bench_map${m}_$TAG.log 2>&1
    echo "map $m: $(tail -1 $O/bench_map${m}_$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'img/s  fwd+dgrad', r['achieved'], 'TF  wgrad', r['wgrad_kernel']['achieved'], 'TF')")"
  done; done ;;
ab_layout)
  echo "== A/B: LDS layout of the conv patch (1 half planes: conflict-free B fragments, 0 round 2: [pixel][half]), same database"
  for i in 1 2; do for m in 1 0; do
    PNSFM_PATCH_LAYOUT=$m PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra > $O/bench_layout${m}_$TAG.log 2>&1
    echo "layout $m: $(tail -1 $O/bench_layout${m}_$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], 'img/s  fwd+dgrad', r['achieved'], 'TF  wgrad', r['wgrad_kernel']['achieved'], 'TF')")"
  done; done ;;
c3)
  echo "== 384x1280 batch 2 (BASELINE configs[2] shape): bench + layer table, rocprofv3 kernel summary, PMC traffic passes"
  C3="--height 384 --width 1280 --batch 2"
  PNSFM_TUNE_DB=$DB timeout 900 python bench.py $C3 --steps 8 --warmup 2 --no-cpu-baseline --no-extra --layer-table $O/layers_c3_$TAG.csv > $O/bench_c3_$TAG.log 2>&1
  tail -1 $O/bench_c3_$TAG.log | cut -c1-300
  (cd /tmp && PNSFM_TUNE_DB=$DB timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_$TAG -o bench -- \
      python $R/bench.py $C3 --steps 4 --warmup 2 --no-cpu-baseline --no-extra > $O/rocprof_c3_$TAG.log 2>&1)
  for c in "FETCH_SIZE" "WRITE_SIZE"; do
    (cd /tmp && PNSFM_TUNE_DB=$DB timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_c3_${TAG}_$c -o bench -- \
      python $R/bench.py $C3 --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-extra > $O/pmc_c3_${TAG}_$c.log 2>&1)
  done
  python tools/pmc_traffic.py $(find $O/pmc_c3_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_c3_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/traffic_c3_$TAG.json $O/layers_c3_$TAG.csv 384,1280,2 | head -6 ;;
bench2)
  echo "== bench again on the primed database (run-to-run spread)"
  PNSFM_TUNE_DB=$DB timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-prof > $O/bench2_$TAG.log 2>&1
  tail -1 $O/bench2_$TAG.log | cut -c1-400 ;;
prof)
  echo "== rocprofv3 kernel trace (database primed)"
  (cd /tmp && PNSFM_TUNE_DB=$DB timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o bench -- \
      python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/rocprof_$TAG.log 2>&1)
  tail -1 $O/rocprof_$TAG.log | cut -c1-300
  f=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 $f | cut -c1-150
  t=$(find $O/prof_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_breakdown.py $t 80 > $O/step_breakdown_$TAG.txt 2>&1 && head -60 $O/step_breakdown_$TAG.txt ;;
pmc)
  echo "== rocprofv3 PMC passes (counters only, separate runs)"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-24)
    (cd /tmp && PNSFM_TUNE_DB=$DB timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${TAG}_$n -o bench -- \
      python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-extra > $O/pmc_${TAG}_$n.log 2>&1)
    tail -1 $O/pmc_${TAG}_$n.log | cut -c1-200
  done
  python tools/pmc_traffic.py $(find $O/pmc_${TAG}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/traffic_$TAG.json $O/layers_$TAG.csv | head -8
  python tools/pmc_mfma_busy.py $(find $O/pmc_${TAG}_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1) $O/mfma_busy_$TAG.json ;;
esac
echo "   [$w: $(( $(date +%s) - t0 )) s]"
done
