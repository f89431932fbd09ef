#!/bin/bash
# Kernel trace + SQ / LDS / HBM counters of any python tool on the GPU box (one rocprofv3 run per counter pass).
# usage: tools/profile_cmd.sh TAG KERNEL_NAME_PATTERN SCRIPT [ARGS...]      (SCRIPT runs as `python3 SCRIPT ARGS`)
# env PASSES="trace sq1 sq2 hbm" selects the passes (default all)
set -e
TAG=$1; KPAT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof
PASSES=${PASSES:-"trace sq1 sq2 hbm"}
mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
if [[ " $PASSES " == *" trace "* ]]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t --output-format csv -- python3 $SCRIPT "$@" > $OUT/${TAG}_run.json 2> $OUT/${TAG}_trace.err
  find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
fi
run_pmc() {   # name, counters
  timeout -s KILL 150 rocprofv3 --pmc $2 -d $OUT/${TAG}_pmc_$1 -o p --output-format csv -- python3 $SCRIPT "${@:3}" > /dev/null 2> $OUT/${TAG}_pmc_$1.err
}
if [[ " $PASSES " == *" sq1 "* ]]; then run_pmc sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "$@"; fi
if [[ " $PASSES " == *" sq2 "* ]]; then run_pmc sq2 "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "$@"; fi
if [[ " $PASSES " == *" hbm "* ]]; then
  run_pmc fetch "FETCH_SIZE" "$@"
  run_pmc write "WRITE_SIZE" "$@"
  run_pmc tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@"
  run_pmc grbm "GRBM_GUI_ACTIVE" "$@"
fi
cd $ROOT
python3 tools/pmc_summarize.py $OUT/${TAG}_pmc_summary.json "$KPAT" $OUT/${TAG}_pmc_*/ > /dev/null
[ -f $OUT/${TAG}_kernel_stats.csv ] && head -5 $OUT/${TAG}_kernel_stats.csv | cut -c1-70,150-400
python3 -c "
import json; s=json.load(open('$OUT/${TAG}_pmc_summary.json'))
for k,v in s.items():
    if isinstance(v,dict): print(k, '%.5g'%v['mean_per_dispatch'], v['dispatches'])
"
