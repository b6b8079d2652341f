#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py in the SERIAL schedule (RESNET_MI_OVERLAP=0: weight gradients on the compute stream,
# nothing runs beside anything): every kernel's own duration, and the sum of kernel time against the step time = launch gaps.
#   gpurun -- bash tools/profile_serial.sh TAG [bench args]   -> gpurun_out/prof_TAG/{kernel_stats.csv,bench.json}
TAG=${1:-serial}; shift
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RESNET_MI_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o run -- python3 "$REPO/bench.py" --no-cpu-baseline --no-extra --no-bf16 --steps 10 --warmup 3 "$@" > "$OUT/bench.json" 2> "$OUT/rocprof.err"
rc=$?
f=$(find "$OUT/raw" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/raw"
tail -1 "$OUT/bench.json" | cut -c1-300
exit $rc
