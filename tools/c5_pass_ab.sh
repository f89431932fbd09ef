#!/bin/bash
# per-pass kernel durations of the 2^30 c2c with the FAR variant of the first pass off (0) and on (1).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/c5diag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in "$@"; do
  export SA_C2C_FAR=$d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/d$d -o t --output-format csv -- python3 $ROOT/tools/prof_c5.py 30 > $OUT/d$d.json 2> $OUT/d$d.err || exit 1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/d$d/t_kernel_trace.csv")) if "c2c_strided" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in rows]
print("diag $d pass1", [round(x,2) for x in d[2:8:2]], "pass2", [round(x,2) for x in d[3:8:2]])
PY
done
