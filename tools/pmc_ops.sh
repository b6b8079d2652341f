#!/bin/bash
# SQ counter pass over one bench_ops selection (GPU box):  bash tools/pmc_ops.sh TAG "<bench_ops args>" "<counters>"
TAG=$1; ARGS=$2; CTRS=${3:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/raw" -o run -- python3 "$REPO/tools/bench_ops.py" $ARGS > "$OUT/ops.log" 2> "$OUT/rocprof.err"
f=$(find "$OUT/raw" -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); cnt[k] += 1
for k, d in agg.items():
    if "fill_uniform" in k: continue
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-60s n=%d" % (k, cnt[k]))
    for c, v in sorted(d.items()): print("    %-28s %14.0f  %6.1f%% of wave cycles" % (c, v, 100 * v / wc))
PY
rm -rf "$OUT/raw"
