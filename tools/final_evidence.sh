#!/bin/bash
# Evidence of a tree, in one gpurun call: headline kernel trace + PMC passes, HBM traffic file, every secondary
# config, and the default bench line.  usage: tools/final_evidence.sh TAG     (results under gpurun_out/)
set -e
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
bash tools/profile_round.sh $TAG --no-extra > gpurun_out/${TAG}_profile_round.txt 2>&1
python3 tools/make_hbm_traffic.py gpurun_out/prof/${TAG}_pmc_summary.json $TAG > /dev/null
timeout -k 10 900 python3 tools/bench_configs.py sweep long metrics analyze c3 c4 c5 > gpurun_out/${TAG}_configs.log 2>&1
cp gpurun_out/bench_configs.json gpurun_out/${TAG}_configs_final.json
timeout -k 10 900 python3 bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -1 gpurun_out/${TAG}_bench_default.json | cut -c1-600
