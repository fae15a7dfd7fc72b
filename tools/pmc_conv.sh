#!/bin/bash
# PMC pass (SQ counters only, no tracing) over tools/conv_micro.py for a few shapes.  usage: tools/pmc_conv.sh tag "B Cin Cout H W ks what" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
i=0
for spec in "$@"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES \
     --output-format csv -d $R/gpurun_out/pmcconv_${TAG}_$i -o p -- python $R/tools/conv_micro.py $spec > $R/gpurun_out/pmcconv_${TAG}_$i.log 2>&1)
  tail -2 $R/gpurun_out/pmcconv_${TAG}_$i.log
done
