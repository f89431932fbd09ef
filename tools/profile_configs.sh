#!/bin/bash
# Kernel trace + SQ/LDS counters of the secondary configs (tools/bench_configs.py c3|c4|c5) on the GPU box.
# usage: tools/profile_configs.sh TAG CONFIG KERNEL_NAME_PATTERN
set -e
TAG=$1; CFG=$2; KPAT=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t --output-format csv -- python3 $ROOT/tools/bench_configs.py $CFG > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/${TAG}_pmc_$name -o p --output-format csv -- python3 $ROOT/tools/bench_configs.py $CFG > /dev/null 2> $OUT/${TAG}_pmc_$name.err
done
cd $ROOT
python3 tools/pmc_summarize.py $OUT/${TAG}_pmc_summary.json "$KPAT" $OUT/${TAG}_pmc_*/ > /dev/null
find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
head -4 $OUT/${TAG}_kernel_stats.csv | cut -c1-60,150-400
python3 -c "
import json; s=json.load(open('$OUT/${TAG}_pmc_summary.json'))
for k,v in s.items():
    if isinstance(v,dict): print(k, '%.4g'%v['mean_per_dispatch'], v['dispatches'])
"
