#!/bin/bash
# Round-6 GPU sessions (one gpurun call each): usage tools/gpu_r5.sh <what...>
#   tests      pytest -m gpu (everything) -> r06_pytest_gpu.log ; smoke
#   quick      the GPU tests selected by $PYTEST_K
#   bench      bench.py exactly as the driver runs it -> r06_bench_default_run.json
#   ab_streams same-box A/B of the side streams (wgrad + pose branch): off / on / off / on
#   lab        tools/conv_lab5.py (per-layer A/B of conv kernel variants; $LAB_ARGS)
#   prof       rocprofv3 --kernel-trace --stats of bench.py + step breakdown + layer tables
#   pmc        FETCH_SIZE / WRITE_SIZE / MFMA-busy / SQ-wait passes (separate runs, counters only)
#   tune       fresh tuning database -> gpurun_out/r06_tuned.db
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
DB=/tmp/shipped_copy.db
cp packnet-sfm_amd/csrc/tuned_gfx950.db $DB
BARGS="--no-cpu-baseline --no-extra --gpu-baseline off"
SUMMARY='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"] or {}; print(d["value"], d["ms_per_step"], "host_issue", d.get("host_issue_ms_per_step"), "frac", r.get("frac"), "as_run", (r.get("as_run") or {}).get("frac"), "wgrad", (r.get("wgrad_kernel") or {}).get("frac"))'
for w in "$@"; do
t0=$(date +%s)
case $w in
tests)
  timeout 1700 python -m pytest tests -m gpu -x -q > $O/r06_pytest_gpu.log 2>&1; tail -4 $O/r06_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; tail -2 $O/r06_smoke.log ;;
quick)
  timeout 1500 python -m pytest tests -m gpu -x -q -k "${PYTEST_K:-round4}" > $O/r06_pytest_quick.log 2>&1; tail -6 $O/r06_pytest_quick.log ;;
bench)
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_bench_default_run.log 2>&1
  tail -1 $O/r06_bench_default_run.log > $O/r06_bench_default_run.json; cut -c1-600 $O/r06_bench_default_run.json ;;
benchq)
  PNSFM_TUNE_DB=$DB timeout 600 python bench.py --steps 20 --warmup 4 $BARGS > $O/r06_benchq.log 2>&1
  echo "bench (shipped db copy): $(tail -1 $O/r06_benchq.log | python -c "$SUMMARY")" ;;
ab_streams)
  for i in 1 2; do for v in 0 1; do
    PNSFM_WGRAD_STREAM=$v PNSFM_BRANCH_STREAM=$v PNSFM_TUNE_DB=$DB timeout 300 python bench.py --steps 20 --warmup 4 $BARGS > $O/r06_st_$v.log 2>&1
    echo "side streams $v: $(tail -1 $O/r06_st_$v.log | python -c "$SUMMARY")"
  done; done | tee $O/r06_ab_streams.txt ;;
ab_env)
  # generic same-box A/B of one environment switch: AB_VAR=NAME (values 0 / 1), alternating, two rounds
  for i in 1 2; do for v in 0 1; do
    env $AB_VAR=$v PNSFM_TUNE_DB=$DB timeout 300 python bench.py --steps 20 --warmup 4 $BARGS > $O/r06_abenv_$v.log 2>&1
    echo "$AB_VAR=$v: $(tail -1 $O/r06_abenv_$v.log | python -c "$SUMMARY")"
  done; done | tee $O/r06_ab_$AB_VAR.txt ;;
ab_ddp)
  # plain step vs the N > 1 code path on one GPU (1-rank RCCL group), alternating
  for i in 1 2; do for v in 0 1; do
    PNSFM_FORCE_DDP=$v PNSFM_TUNE_DB=$DB timeout 300 python bench.py --steps 20 --warmup 4 $BARGS > $O/r06_ddp_$v.log 2>&1
    echo "forced 1-rank collectives $v: $(tail -1 $O/r06_ddp_$v.log | python -c "$SUMMARY") $(tail -1 $O/r06_ddp_$v.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps((d["config"].get("allreduce") or {})))' | cut -c1-400)"
  done; done | tee $O/r06_ab_forced_ddp.txt ;;
ab_adam)
  for i in 1 2; do for v in 0 1; do
    PNSFM_ADAM_OVERLAP=$v PNSFM_TUNE_DB=$DB timeout 300 python bench.py --steps 20 --warmup 4 $BARGS > $O/r06_ao_$v.log 2>&1
    echo "adam underneath backward $v: $(tail -1 $O/r06_ao_$v.log | python -c "$SUMMARY")"
  done; done | tee $O/r06_ab_adam_overlap.txt ;;
lab)
  timeout 1500 python tools/conv_lab5.py $LAB_ARGS > $O/r06_lab.txt 2> $O/r06_lab.err; tail -70 $O/r06_lab.txt; tail -5 $O/r06_lab.err ;;
tune)
  rm -f /tmp/new.db
  PNSFM_TUNE_DB=/tmp/new.db PNSFM_TUNE_LOG=$O/r06_tunelog_192x640.txt timeout 1200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gpu-baseline off > $O/r06_tune_run.log 2>&1
  tail -1 $O/r06_tune_run.log | cut -c1-160
  PNSFM_TUNE_DB=/tmp/new.db timeout 600 python bench.py --depth-net PackNetSlim01 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --gpu-baseline off > $O/r06_tune_slim.log 2>&1
  cp /tmp/new.db $O/r06_tuned.db; wc -l $O/r06_tuned.db ;;
prof)
  timeout 600 python bench.py --steps 10 --warmup 3 $BARGS --layer-table $O/r06_conv_layer_table.csv > $O/r06_bench.log 2>&1
  tail -1 $O/r06_bench.log > $O/r06_bench.json; cut -c1-300 $O/r06_bench.json
  timeout 600 python bench.py --height 384 --width 1280 --batch 2 --steps 8 --warmup 2 $BARGS --layer-table $O/r06_conv_layer_table_384x1280.csv > $O/r06_bench_384.log 2>&1
  tail -1 $O/r06_bench_384.log > $O/r06_bench_384x1280.json; cut -c1-200 $O/r06_bench_384x1280.json
  # (side streams OFF under the profiler: per-kernel durations of kernels that run alone, like the PMC passes)
  (cd /tmp && PNSFM_WGRAD_STREAM=0 PNSFM_BRANCH_STREAM=0 PNSFM_SHORTCUT_STREAM=0 PNSFM_TUNE_DB=$DB timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06 -o bench -- python $R/bench.py --steps 6 --warmup 2 $BARGS --no-calibration > $O/r06_rocprof.log 2>&1)
  f=$(find $O/prof_r06 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_bench_kernel_stats.csv && head -12 $f | cut -c1-140
  t=$(find $O/prof_r06 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_breakdown.py $t 90 > $O/r06_step_breakdown.txt 2>&1 && head -8 $O/r06_step_breakdown.txt
  rm -rf $O/prof_r06
  (cd /tmp && PNSFM_WGRAD_STREAM=0 PNSFM_BRANCH_STREAM=0 PNSFM_SHORTCUT_STREAM=0 PNSFM_TUNE_DB=$DB timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06b -o bench -- python $R/bench.py --height 384 --width 1280 --batch 2 --steps 6 --warmup 2 $BARGS --no-calibration > $O/r06_rocprof_384.log 2>&1)
  f=$(find $O/prof_r06b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_bench_kernel_stats_384x1280.csv
  t=$(find $O/prof_r06b -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_breakdown.py $t 90 > $O/r06_step_breakdown_384x1280.txt 2>&1 && head -4 $O/r06_step_breakdown_384x1280.txt
  rm -rf $O/prof_r06b ;;
proflite)
  # rocprofv3 kernel trace of a short run (side streams off: every kernel alone) -> step breakdown only
  (cd /tmp && PNSFM_WGRAD_STREAM=0 PNSFM_BRANCH_STREAM=0 PNSFM_TUNE_DB=$DB timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06l -o bench -- python $R/bench.py --steps 4 --warmup 2 $BARGS --no-prof --no-calibration > $O/r06_rocprof_lite.log 2>&1)
  t=$(find $O/prof_r06l -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_breakdown.py $t 90 5 "${LITE_RE:-^$}" > $O/r06_step_breakdown_lite.txt 2>&1 && head -${LITE_LINES:-40} $O/r06_step_breakdown_lite.txt
  rm -rf $O/prof_r06l ;;
pmc)
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-24)
    (cd /tmp && PNSFM_WGRAD_STREAM=0 PNSFM_BRANCH_STREAM=0 PNSFM_SHORTCUT_STREAM=0 PNSFM_TUNE_DB=$DB timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_r06_$n -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-prof --no-calibration $BARGS > $O/r06_pmc_$n.log 2>&1)
    echo "pass $n: $(tail -1 $O/r06_pmc_$n.log | cut -c1-100)"
  done
  python tools/pmc_traffic.py $(find $O/pmc_r06_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_r06_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/r06_traffic.json $O/r06_conv_layer_table.csv 192,640,4 | head -8
  python tools/pmc_mfma_busy.py $(find $O/pmc_r06_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1) $O/r06_mfma_busy.json | head -8
  python tools/pmc_sq_waits.py $(find $O/pmc_r06_SQ_WAVE_CYCLES_SQ_WAIT_A -name "*counter_collection.csv" | head -1) $O/r06_sq_waits.json | head -8
  rm -rf $O/pmc_r06_* ;;
esac
echo "-- $w took $(( $(date +%s) - t0 )) s"
done
