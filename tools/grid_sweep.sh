#!/bin/bash
# Sweep of the workgroup-count tuning factors (SA_GRIDX_<family>, percent) over the secondary configs.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for pct in "$@"; do
  export SA_GRIDX_GENERIC=$pct SA_GRIDX_STFT=$pct SA_GRIDX_PIPE=$pct SA_GRIDX_C2C=$pct SA_GRIDX_METRICS=$pct
  echo "== $pct %"
  timeout -k 10 300 python $ROOT/tools/bench_configs.py c3 c4 c5 metrics 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('  ', r['config'][:34], {k:round(v,3) for k,v in r.items() if k.endswith('ms')})"
  SA_FORCE_GENERIC=1 timeout -k 10 100 python $ROOT/bench.py --no-cpu --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   generic 4096', round(d['value']/1e6,2), 'M/s')"
done
