#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench.py run (run on the GPU box: gpurun -- bash tools/profile_bench.sh TAG)
# writes gpurun_out/prof_TAG/kernel_stats.csv (copy into profiles/ to have it judged)
TAG=${1:-r1}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o run -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extra --no-bf16 --steps 10 --warmup 3 $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/rocprof.err"
rc=$?
f=$(find "$OUT/raw" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
tail -1 "$OUT/bench.json"
exit $rc
