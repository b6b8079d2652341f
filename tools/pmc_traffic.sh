#!/bin/bash
# HBM traffic per kernel family of one bench step, from two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; with
# --kernel-trace only), as MI355X_MICROARCH.md's HBM section prescribes.  Run on the GPU box:
#   gpurun -- bash tools/pmc_traffic.sh TAG      ->  gpurun_out/pmc_traffic_TAG.json   (copy to profiles/)
TAG=${1:-r1}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_traffic_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o run -- python3 "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-extra --no-bf16 $BENCH_ARGS > "$OUT/$c.log" 2> "$OUT/$c.err" || exit 1
done
python3 - "$OUT" "$REPO/gpurun_out/pmc_traffic_$TAG.json" <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
def fam(name):
    if name.startswith("void igemm_kernel<") or name.startswith("void bgemm_kernel<"):
        ks = name.split("<")[1].split(",")[1].strip()
        return "igemm3x3" if ks == "3" else "gemm"
    # channel-last 3x3 (bf16) and the re-layout passes bench.py brackets with that family (mid_cl_relayout*, bg_s2d)
    if "cl_conv_kernel" in name or "cl_dgrad2_kernel" in name or "cl_wgrad" in name or "cl_relayout" in name or "bg_s2d" in name: return "igemm3x3"
    if "pw_wgrad_kernel" in name: return "gemm"
    if "bg_wt_" in name: return "igemm3x3_aux"
    if "igemm_tail_reduce" in name or "igemm_wt_kernel<9>" in name or "igemm_wgrad_reduce_kernel<9>" in name: return "igemm3x3_aux"
    if "gemm_mfma_kernel" in name or "igemm_w" in name: return "gemm"
    if "st_wgrad" in name or "st32_wgrad" in name: return "wgradC"
    if "dconv_kernel" in name or "wt_relayout" in name or name.startswith("st_") or name.startswith("st32_") or "void st_" in name or "void st32_" in name: return "dconv"
    if "wgradC_kernel" in name or "wgrad_kernel" in name or "split_reduce" in name: return "wgradC"
    if name.startswith("bn_") or "void bn_" in name: return "bn"
    return "other"
STEPS = 2.0
agg = collections.defaultdict(lambda: {"launches": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True)[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = fam(r["Kernel_Name"])
        agg[k][c] += float(r["Counter_Value"]) * 1024.0  # counter unit: KB
        if c == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); agg[k]["launches"] += 1
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(dst)))
import bench
res = {"source_sha16": bench.source_sha16(),
       "source": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (with --kernel-trace only) over "
                 "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof --no-extra --no-bf16 %s` (2 training steps each pass)" % __import__("os").environ.get("BENCH_ARGS", ""),
       "unit_note": "counter values are KB; bytes = value*1024. MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide "
                    "(16 B/lane) coalesced stream; other access widths are uncalibrated (raw value reported, doubled value = upper bound). "
                    "Infinity-Cache hits are counted, not excluded.",
       "families": {}}
for k, d in sorted(agg.items()):
    n = max(d["launches"], 1)
    res["families"][k] = {"launches_in_2_steps": d["launches"], "fetch_bytes_per_launch_raw": d["FETCH_SIZE"] / n,
                          "write_bytes_per_launch": d["WRITE_SIZE"] / n, "hbm_bytes_per_launch_raw": (d["FETCH_SIZE"] + d["WRITE_SIZE"]) / n,
                          "hbm_bytes_per_launch_fetch_doubled": (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) / n,
                          "hbm_GB_per_step_raw": (d["FETCH_SIZE"] + d["WRITE_SIZE"]) / STEPS / 1e9,
                          "hbm_GB_per_step_fetch_doubled": (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) / STEPS / 1e9}
json.dump(res, open(dst, "w"), indent=1)
for k, v in res["families"].items(): print("%-14s launches/2 steps %5d   HBM GB/step raw %8.2f  (fetch doubled %8.2f)" % (k, v["launches_in_2_steps"], v["hbm_GB_per_step_raw"], v["hbm_GB_per_step_fetch_doubled"]))
PY
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
