#!/bin/bash
# Everything profiles/rN_* is made of, in one gpurun call:  gpurun --timeout 1200 -- bash tools/round_profiles.sh r3
#   the two --pmc passes (FETCH_SIZE, WRITE_SIZE) for fp32 and bf16      -> gpurun_out/pmc_traffic_TAG.json, pmc_traffic_TAG_bf16.json
#     (copied to profiles/TAG_pmc_traffic.json / TAG_bf16_pmc_traffic.json on the box FIRST, so that the bench line below carries traffic
#      measured on its own sources: traffic_stale false)
#   bench.py default run (fp32 line + bf16 leg + cpu baseline)          -> gpurun_out/TAG_bench.json
#   rocprofv3 --kernel-trace --stats of the fp32 and the bf16 bench      -> gpurun_out/prof_TAG/, prof_TAG_bf16/
# afterwards, here: cp gpurun_out/pmc_traffic_TAG.json profiles/TAG_pmc_traffic.json (same for _bf16), TAG_bench.json, prof_*/kernel_stats.csv
TAG=${1:-r3}
cd "$(dirname "$0")/.."
bash tools/pmc_traffic.sh ${TAG} || exit 1
BENCH_ARGS="--dtype bf16" bash tools/pmc_traffic.sh ${TAG}_bf16 || exit 1
cp gpurun_out/pmc_traffic_${TAG}.json profiles/${TAG}_pmc_traffic.json
cp gpurun_out/pmc_traffic_${TAG}_bf16.json profiles/${TAG}_bf16_pmc_traffic.json
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
cut -c1-240 gpurun_out/${TAG}_bench.json
bash tools/profile_bench.sh ${TAG} | cut -c1-200 || exit 1
BENCH_ARGS="--dtype bf16" bash tools/profile_bench.sh ${TAG}_bf16 | cut -c1-200 || exit 1
