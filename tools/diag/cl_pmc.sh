#!/bin/bash
# diagnostic: SQ / TCC counters of the channel-last kernels on chosen layers (two --pmc passes, no tracing domains)
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
LAYERS=${1:-b3_proj,b7_proj,s2_3x3}
OUT=gpurun_out/cl_pmc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
    -d $OUT/p1 -o p1 --output-format csv -- python3 tools/diag/cl_bench.py 256 $LAYERS > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum \
    -d $OUT/p2 -o p2 --output-format csv -- python3 tools/diag/cl_bench.py 256 $LAYERS > $OUT/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob("gpurun_out/cl_pmc/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f: print(p, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    disp = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:60] + " grid=" + r["Grid_Size"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in disp: disp.add((k, r["Dispatch_Id"])); cnt[k] += 1
    for k in acc:
        if "bgemm" in k or "cl_" in k:
            print(k, "n=%d" % cnt[k], " ".join("%s=%.4g" % (c, v / cnt[k]) for c, v in sorted(acc[k].items())))
PY
