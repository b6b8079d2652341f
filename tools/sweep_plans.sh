for bm in 0 64 128; do for tl in 1 0; do
  RESNET_MI_IGEMM_BM=$bm RESNET_MI_IGEMM_TAIL=$tl python tools/bench_ops.py --only fwd,dgrad --reps 5 > gpurun_out/sw_${bm}_${tl}.log 2>&1
done; done
python3 - <<'PY'
import collections
res = collections.defaultdict(dict)
for bm in (0, 64, 128):
    for tl in (1, 0):
        for l in open("gpurun_out/sw_%d_%d.log" % (bm, tl)):
            p = l.split()
            if len(p) == 4 and p[1] in ("fwd", "dgrad"):
                res[(p[0], p[1])][(bm, tl)] = float(p[2])
tot_h = tot_b = 0
for k, d in res.items():
    h = d[(0, 1)]; best = min(d.values()); arg = min(d, key=d.get)
    tot_h += h; tot_b += best
    flag = "" if h <= best * 1.03 else "  <-- heuristic %.0f%% slower than %s" % (100 * (h / best - 1), arg)
    print("%-18s %-6s heur %.3f  bm64/t1 %.3f bm64/t0 %.3f bm128/t1 %.3f bm128/t0 %.3f%s" % (k[0], k[1], h, d[(64,1)], d[(64,0)], d[(128,1)], d[(128,0)], flag))
print("sum heuristic %.2f ms, sum best %.2f ms (unweighted by layer count)" % (tot_h, tot_b))
PY
