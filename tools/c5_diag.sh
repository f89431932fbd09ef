#!/bin/bash
# Temporary diagnostic: per-pass kernel durations of the 2^30 c2c with the first pass' reads / writes made contiguous.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c5diag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  export SA_C2C_DIAG=$d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/d$d -o t --output-format csv -- python3 $ROOT/tools/prof_c5.py 30 > $OUT/d$d.json 2> $OUT/d$d.err || exit 1
  f=$(find $OUT/d$d -name "*kernel_stats.csv" | head -1)
  echo "== diag $d"; grep c2c $f | awk -F, '{print substr($1,1,60), $2, $4}'
done
