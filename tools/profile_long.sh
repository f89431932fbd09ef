#!/bin/bash
# HBM traffic of the long-frame kernels (rfft_large.hip) at n = 2^16 fp32, one lane / 128 MiB chunks so that every dispatch
# moves the same bytes: FETCH_SIZE and WRITE_SIZE passes (counters only), summarised per kernel.
# usage: tools/profile_long.sh TAG        (run on the GPU box through gpurun)
set -e
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SA_RL_LANES=1 SA_RL_CHUNK_MIB=128
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $pass -d $OUT/${TAG}_pmc_$pass -o p --output-format csv -- python3 $ROOT/tools/bench_long.py f32 16 > $OUT/${TAG}_$pass.out 2> $OUT/${TAG}_$pass.err
done
cd $ROOT
for k in rl_first_pass rl_final_pass; do
  python3 tools/pmc_summarize.py $OUT/${TAG}_${k}.json "$k" $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE > /dev/null
  python3 -c "
import json; s=json.load(open('$OUT/${TAG}_${k}.json'))
print('$k', {k:(round(v['mean_per_dispatch']),v['dispatches']) for k,v in s.items() if isinstance(v,dict)})"
done
