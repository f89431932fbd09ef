#!/bin/bash
# A/B of the wave kernel's cache policy (SA_R4K_NT=1 nontemporal loads + stores, the default; 0 = default policy)
# against frames per wave, for mode A, mode B (peak only) and the fused read-out (analyze_frames).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']/1e6,2), 'M/s', round(d['roofline']['achieved']), 'GB/s')"; }
for nt in ${NTS:-1 0}; do
  for fpw in ${FPWS:-10 12 15 19}; do
    SA_R4K_NT=$nt SA_R4K_FPW10=$fpw timeout -k 10 100 python $ROOT/bench.py --no-cpu --steps 20 2>/dev/null | line "modeA nt=$nt fpw10=$fpw" || exit 1
  done
  for fpw in ${FPWS_B:-0 10 40}; do
    SA_R4K_NT=$nt SA_R4K_FPW10=$fpw timeout -k 10 100 python $ROOT/bench.py --no-cpu --steps 20 --peak-only 2>/dev/null | line "modeB nt=$nt fpw10=$fpw" || exit 1
  done
  for fpw in ${FPWS_F:-10 20 60}; do
    SA_R4K_NT=$nt SA_R4K_FPW10=$fpw timeout -k 10 200 python $ROOT/tools/bench_configs.py analyze 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('analyze nt=$nt fpw10=$fpw', {k:round(v,3) for k,v in r.items() if k.endswith('ms')})" || exit 1
  done
done
