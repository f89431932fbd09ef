#!/bin/bash
# Round profile of the headline bench (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of `bench.py`           -> gpurun_out/prof/TAG_trace/
#   2. PMC passes (each its own run, counters only): FETCH_SIZE | WRITE_SIZE | SQ set | GRBM
# usage: tools/profile_round.sh TAG [bench args]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -o t --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 90 rocprofv3 --pmc $pass -d $OUT/${TAG}_pmc_$name -o p --output-format csv -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_$name.err
done
cd $ROOT
python3 tools/pmc_summarize.py $OUT/${TAG}_pmc_summary.json "${KPAT:-rfft4096_wave_kernel<1,}" $OUT/${TAG}_pmc_*/
find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
# steady state: the 5 warm-up dispatches are left out, so the average is the one bench.py's timed region sees
find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 tools/steady_stats.py {} "${KPAT:-rfft4096_wave_kernel<1,}" 5 $OUT/${TAG}_steady.json
head -5 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
tail -1 $OUT/${TAG}_bench.json | cut -c1-400
