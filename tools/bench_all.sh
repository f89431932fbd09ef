#!/bin/bash
# usage: tools/bench_all.sh TAG  -- bench line for C2 (tuned + generic) and C3/C4/C5 into gpurun_out/TAG_*.log
set -e
T=$1
timeout -k 10 120 python bench.py --no-cpu > gpurun_out/${T}_c2.json 2>/dev/null
SA_FORCE_GENERIC=1 timeout -k 10 120 python bench.py --no-cpu > gpurun_out/${T}_c2gen.json 2>/dev/null
timeout -k 10 120 python bench.py --no-cpu --peak-only > gpurun_out/${T}_c2b.json 2>/dev/null
timeout -k 10 300 python tools/bench_configs.py c3 c4 c5 > gpurun_out/${T}_cfg.log 2>/dev/null
python - <<PY
import json
for f in ("c2","c2gen","c2b"):
    j=json.loads(open("gpurun_out/${T}_%s.json"%f).read().strip().splitlines()[-1])
    print("${T}",f,"%.1f M/s"%(j["value"]/1e6),"kernel_ms=%.3f"%j["roofline"]["kernel_ms"],"parity",j["parity_sample_ok"])
for l in open("gpurun_out/${T}_cfg.log"):
    if l.startswith("{"):
        j=json.loads(l); print("${T}",j["config"][:12],"ms=%.2f"%j["ms"])
PY
