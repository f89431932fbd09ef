#!/bin/bash
# Per-pass kernel times of the 2^30 c2c transform (kernel trace), nontemporal passes on (default) and off.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export C5_LOG2=30
for mib in 1024 0; do
  export SA_C2C_NT_MIB=$mib
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/c5_nt${mib}_trace -o t --output-format csv -- python3 $ROOT/tools/bench_configs.py c5 > $OUT/c5_nt${mib}.json 2> $OUT/c5_nt${mib}.err || exit 1
  f=$(find $OUT/c5_nt${mib}_trace -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/c5_nt${mib}_kernel_stats.csv
  echo "SA_C2C_NT_MIB=$mib"; grep c2c_ $f | cut -d, -f1-4 | sed -E 's/\(sa::C2cArgs\)//; s/void sa:://' | cut -c1-120
done
