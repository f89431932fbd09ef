#!/bin/bash
# A/B of the workgroup shape of the 4096-point wave kernel: SA_R4K_WH = 0: one looping 16-wave workgroup per CU, full
# window table; 3 (default) / 6: one-shot 3- / 4-wave workgroups on the half table (four / three per CU).  (1 / 5 / 7 — 8-wave and looping 4-wave
# forms — were measured with this script, profiles/r01_wg_shape_ab.txt, and are no longer instantiated.)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,2), 'M/s', round(d['roofline']['achieved']), 'GB/s')"; }
for lb in ${LBS:-20 18}; do
for wh in ${WHS:-0 3 6}; do
for fpw in ${FPWS:-10}; do
  SA_R4K_WH=$wh SA_R4K_FPW10=$fpw timeout -k 10 100 python $ROOT/bench.py --no-cpu --steps 20 --log2-batch $lb 2>/dev/null | line "modeA wh=$wh fpw10=$fpw 2^$lb" || exit 1
done
done
done
