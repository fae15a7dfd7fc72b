#!/bin/bash
# Round-4 final measurement sessions (one gpurun call each): usage tools/gpu_final4.sh <what...>
#   tune      bench.py (default shape + the 384x1280 extra) on a FRESH tuning database -> gpurun_out/r04_tuned.db (copy to csrc/tuned_gfx950.db)
#   tests     pytest -m gpu (everything) -> r04_pytest_gpu.log ; smoke
#   bench     bench.py exactly as the driver runs it (shipped database, CPU baseline, extra block, gpu baseline auto) -> r04_bench_default_run.json
#   prof      rocprofv3 --kernel-trace --stats of bench.py + step breakdown + layer tables
#   pmc       FETCH_SIZE / WRITE_SIZE / MFMA-busy / SQ-wait passes (separate runs, counters only) + traffic json
#   ddp       1-rank RCCL rehearsal (PNSFM_FORCE_DDP=1), 2-rank gloo rehearsal, wgrad side stream A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd $R
DB=/tmp/shipped_copy.db
cp packnet-sfm_amd/csrc/tuned_gfx950.db $DB
for w in "$@"; do
t0=$(date +%s)
case $w in
tune)
  rm -f /tmp/new.db
  PNSFM_TUNE_DB=/tmp/new.db timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gpu-baseline off > $O/r04_tune_run.log 2>&1
  tail -1 $O/r04_tune_run.log | cut -c1-120
  PNSFM_TUNE_DB=/tmp/new.db timeout 600 python bench.py --depth-net PackNetSlim01 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --gpu-baseline off > $O/r04_tune_slim.log 2>&1
  cp /tmp/new.db $O/r04_tuned.db; wc -l $O/r04_tuned.db ;;
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/r04_pytest_gpu.log 2>&1; tail -4 $O/r04_pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke.log 2>&1; tail -2 $O/r04_smoke.log ;;
bench)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_default_run.log 2>&1
  tail -1 $O/r04_bench_default_run.log > $O/r04_bench_default_run.json; cut -c1-400 $O/r04_bench_default_run.json ;;
prof)
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --gpu-baseline off --layer-table $O/r04_conv_layer_table.csv > $O/r04_bench.log 2>&1
  tail -1 $O/r04_bench.log > $O/r04_bench.json; cut -c1-200 $O/r04_bench.json
  timeout 600 python bench.py --height 384 --width 1280 --batch 2 --steps 8 --warmup 2 --no-cpu-baseline --no-extra --gpu-baseline off --layer-table $O/r04_conv_layer_table_384x1280.csv > $O/r04_bench_384.log 2>&1
  tail -1 $O/r04_bench_384.log > $O/r04_bench_384x1280.json; cut -c1-200 $O/r04_bench_384x1280.json
  (cd /tmp && PNSFM_TUNE_DB=$DB timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r04 -o bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --gpu-baseline off > $O/r04_rocprof.log 2>&1)
  f=$(find $O/prof_r04 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r04_bench_kernel_stats.csv && head -12 $f | cut -c1-140
  t=$(find $O/prof_r04 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/step_breakdown.py $t 90 > $O/r04_step_breakdown.txt 2>&1 && head -8 $O/r04_step_breakdown.txt
  rm -rf $O/prof_r04 ;;
pmc)
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-24)
    (cd /tmp && PNSFM_TUNE_DB=$DB timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_r04_$n -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-extra --gpu-baseline off > $O/r04_pmc_$n.log 2>&1)
    echo "pass $n: $(tail -1 $O/r04_pmc_$n.log | cut -c1-100)"
  done
  python tools/pmc_traffic.py $(find $O/pmc_r04_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/pmc_r04_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/r04_traffic.json $O/r04_conv_layer_table.csv 192,640,4 | head -8
  python tools/pmc_mfma_busy.py $(find $O/pmc_r04_SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv" | head -1) $O/r04_mfma_busy.json | head -8
  python tools/pmc_sq_waits.py $(find $O/pmc_r04_SQ_WAVE_CYCLES_SQ_WAIT_A -name "*counter_collection.csv" | head -1) $O/r04_sq_waits.json | head -8
  rm -rf $O/pmc_r04_* ;;
ddp)
  PNSFM_FORCE_DDP=1 timeout 600 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra --gpu-baseline off > $O/r04_bench_forced_ddp.log 2>&1
  tail -1 $O/r04_bench_forced_ddp.log > $O/r04_bench_forced_ddp_1rank.json; cut -c1-160 $O/r04_bench_forced_ddp_1rank.json
  timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --gpu-baseline off > $O/r04_bench_2rank.log 2>&1
  tail -1 $O/r04_bench_2rank.log > $O/r04_bench_2rank_gloo_rehearsal.json; cut -c1-160 $O/r04_bench_2rank_gloo_rehearsal.json
  for ws in 0 1 0 1; do PNSFM_WGRAD_STREAM=$ws timeout 300 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extra --gpu-baseline off > $O/r04_ws_$ws.log 2>&1; echo "wgrad side stream $ws: $(tail -1 $O/r04_ws_$ws.log | cut -c1-110)"; done ;;
esac
echo "-- $w took $(( $(date +%s) - t0 )) s"
done
