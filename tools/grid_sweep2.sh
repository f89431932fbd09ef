#!/bin/bash
# Finer sweep of one family's workgroup-count factor: grid_sweep2.sh FAMILY CONFIG pct...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FAM=$1; CFG=$2; shift 2
for pct in "$@"; do
  export SA_GRIDX_$FAM=$pct
  if [ "$CFG" = generic ]; then
    SA_FORCE_GENERIC=1 timeout -k 10 100 python $ROOT/bench.py --no-cpu --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$FAM $pct generic 4096', round(d['value']/1e6,2), 'M/s')"
  else
    timeout -k 10 300 python $ROOT/tools/bench_configs.py $CFG 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$FAM $pct', r['config'][:34], {k:round(v,3) for k,v in r.items() if k.endswith('ms')})"
  fi
done
