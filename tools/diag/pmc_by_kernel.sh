#!/bin/bash
# diagnostic: FETCH_SIZE / WRITE_SIZE per kernel NAME over 2 bench steps (two separate --pmc passes), to find kernels that over-fetch
#   gpurun -- bash tools/diag/pmc_by_kernel.sh [--dtype bf16]
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/pmc_by_kernel
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-extra --no-bf16 "$@" > "$OUT/$c.log" 2> "$OUT/$c.err" || exit 1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "ns": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60] + " g=%s" % r["Grid_Size"]
        agg[k][c] += float(r["Counter_Value"]) * 1024.0
        if c == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); agg[k]["n"] += 1
    t = glob.glob(out + "/" + c + "/**/*kernel_trace.csv", recursive=True)
    if c == "FETCH_SIZE" and t:
        for r in csv.DictReader(open(t[0])):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60] + " g=%d" % (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
            agg[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = sorted(agg.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"]))
print("%-78s %5s %10s %10s %9s %8s" % ("kernel", "n", "fetch2x MB", "write MB", "us/launch", "TB/s"))
for k, d in rows[:60]:
    n = max(d["n"], 1)
    by = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) / n
    us = d["ns"] / n / 1e3
    print("%-78s %5d %10.1f %10.1f %9.1f %8.2f" % (k, d["n"], 2 * d["FETCH_SIZE"] / n / 1e6, d["WRITE_SIZE"] / n / 1e6, us, by / max(us, 1e-9) / 1e6))
PY
