#!/bin/bash
# A/B of the nontemporal c2c passes (SA_C2C_NT_MIB = smallest data set in MiB that uses them, 0 = never) over sizes.
# (Nontemporal spectrum STORES in the STFT / fused-generator / generic rfft kernels were A/B-ed the same way with a
# -DSA_STORE_AUX build: C3 10.20 vs 10.15 ms, C4 4.06 vs 4.06 ms, generic 4096 181 vs 187 M/s - not adopted.)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
show() { grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('$1', r['config'][:40], {k:round(v,3) for k,v in r.items() if k.endswith('ms')})"; }
for mib in ${MIBS:-0 128 1024}; do
  SA_C2C_NT_MIB=$mib C5_LOG2=${C5_LOG2:-22,23,24,25,26,27} timeout -k 10 300 python $ROOT/tools/bench_configs.py c5 2>/dev/null | show "c2c_nt_mib=$mib" || exit 1
done
