#!/bin/bash
# Everything profiles/rN_* is made of, in one gpurun call:  gpurun --timeout 1200 -- bash tools/round_profiles.sh r3
#   bench.py default run (fp32 line + bf16 leg + cpu baseline)          -> gpurun_out/TAG_bench.json
#   rocprofv3 --kernel-trace --stats of the fp32 and the bf16 bench      -> gpurun_out/prof_TAG/, prof_TAG_bf16/
#   the two --pmc passes (FETCH_SIZE, WRITE_SIZE) for fp32 and bf16      -> gpurun_out/pmc_traffic_TAG.json, pmc_traffic_TAG_bf16.json
TAG=${1:-r3}
cd "$(dirname "$0")/.."
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
cut -c1-240 gpurun_out/${TAG}_bench.json
bash tools/profile_bench.sh ${TAG} | cut -c1-200 || exit 1
BENCH_ARGS="--dtype bf16" bash tools/profile_bench.sh ${TAG}_bf16 | cut -c1-200 || exit 1
bash tools/pmc_traffic.sh ${TAG} || exit 1
BENCH_ARGS="--dtype bf16" bash tools/pmc_traffic.sh ${TAG}_bf16 || exit 1
