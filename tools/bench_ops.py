#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv kernel families through the C-ABI operator layer, on the reference
network's own layer shapes.  Operands are device-generated uniform random data (zero
operands would inflate clocks); timing is the library's HIP-event bracket around each kernel launch.
  python tools/bench_ops.py [--batch 256] [--reps 3] [--only fwd,dgrad,wgrad] [--filter 3x3]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resnet_amd import binding as B  # noqa: E402

# (name, C, H, K, k, stride, count per step)
LAYERS = [
    ("stem7x7", 3, 224, 64, 7, 2, 1),
    ("s0_3x3", 64, 56, 64, 3, 1, 3),
    ("b3_3x3s2", 128, 56, 128, 3, 2, 1),
    ("b3_proj", 256, 56, 512, 3, 2, 1),
    ("s1_3x3", 128, 28, 128, 3, 1, 3),
    ("b7_3x3s2", 256, 28, 256, 3, 2, 1),
    ("b7_proj", 512, 28, 1024, 3, 2, 1),
    ("s2_3x3", 256, 14, 256, 3, 1, 5),
    ("b13_3x3s2", 512, 14, 512, 3, 2, 1),
    ("b13_proj", 1024, 14, 2048, 3, 2, 1),
    ("s3_3x3", 512, 7, 512, 3, 1, 2),
    ("1x1_64_256@56", 64, 56, 256, 1, 1, 4),
    ("1x1_256_64@56", 256, 56, 64, 1, 1, 2),
    ("1x1_512_128@28", 512, 28, 128, 1, 1, 3),
    ("1x1_128_512@28", 128, 28, 512, 1, 1, 4),
    ("1x1_1024_256@14", 1024, 14, 256, 1, 1, 5),
    ("1x1_256_1024@14", 256, 14, 1024, 1, 1, 6),
    ("1x1_2048_512@7", 2048, 7, 512, 1, 1, 2),
    ("1x1_512_2048@7", 512, 7, 2048, 1, 1, 3),
]


# BASELINE.json configs[1]: "ResNet-18 fp32 batch 64, 224x224 synthetic, 1 MI355X (3x3 conv + BN kernels only, no MFMA)".  The reference
# has no BasicBlock (resnet.h:43-74), so the configuration is measured as what it names: the 3x3 convolutions (+ the stem) and the batch
# norms of a ResNet-18 at batch 64, operator by operator, through the C-ABI.  (name, C, H, K, k, stride, count)
R18_LAYERS = [
    ("stem7x7", 3, 224, 64, 7, 2, 1),
    ("l1_3x3", 64, 56, 64, 3, 1, 4),
    ("l2_3x3s2", 64, 56, 128, 3, 2, 1), ("l2_3x3", 128, 28, 128, 3, 1, 3),
    ("l3_3x3s2", 128, 28, 256, 3, 2, 1), ("l3_3x3", 256, 14, 256, 3, 1, 3),
    ("l4_3x3s2", 256, 14, 512, 3, 2, 1), ("l4_3x3", 512, 7, 512, 3, 1, 3),
]


def bench_bn(L, N, Cc, H, reps):
    """batch norm forward (+ReLU) and backward at one shape: ms from the library's HIP-event bracket (family 3)"""
    n = N * Cc * H * H
    x, y, dy, dx = (L.mi_malloc(4 * n) for _ in range(4))
    g, b, m, v, dg, db = (L.mi_malloc(4 * Cc) for _ in range(6))
    L.mi_op_fill_uniform(x, n, 5, -1.0, 1.0); L.mi_op_fill_uniform(dy, n, 6, -1.0, 1.0)
    L.mi_op_fill_uniform(g, Cc, 7, 0.9, 1.1); L.mi_op_fill_uniform(b, Cc, 8, -0.1, 0.1)
    out = []
    for which in ("fwd", "bwd"):
        L.mi_prof_enable(1)
        for rep in range(reps + 1):
            if rep == 1:
                L.mi_prof_reset()
            rc = L.mi_op_bn_fwd(x, g, b, m, v, y, N, Cc, H, 1e-7, 1) if which == "fwd" else L.mi_op_bn_bwd(x, g, b, m, v, dy, None, dx, dg, db, N, Cc, H, 1e-7, 1)
            if rc != 0:
                raise SystemExit("bn %s failed: %s" % (which, L.mi_last_error().decode()))
        n_, ms_, fl_, by_ = C.c_long(0), C.c_double(0), C.c_double(0), C.c_double(0)
        L.mi_prof_get(3, C.byref(n_), C.byref(ms_), C.byref(fl_), C.byref(by_))
        L.mi_prof_enable(0)
        out.append(ms_.value / reps)
    for p_ in (x, y, dy, dx, g, b, m, v, dg, db):
        L.mi_free(p_)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--r18", action="store_true", help="BASELINE configs[1]: the 3x3 convolutions + batch norms of a ResNet-18 at --batch (default 64); one JSON line at the end")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--filter", default="")
    ap.add_argument("--bf16", action="store_true", help="the bf16-activation kernels (x, y, dx as bf16 tensors; weights / dw fp32)")
    args = ap.parse_args()
    L = B.load()
    if L.mi_device_count() < 1:
        raise SystemExit("needs a HIP device")
    global LAYERS
    if args.r18:
        LAYERS = R18_LAYERS
        if args.batch == 256:
            args.batch = 64
    N = args.batch
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    stem_fwd_ms = 0.0
    print("%-18s %-6s %9s %9s" % ("layer", "op", "ms", "TFLOP/s"))
    for name, Cc, H, K, k, s, cnt in LAYERS:
        if args.filter and args.filter not in name:
            continue
        Ho = H // s
        nx, nw, ny = N * Cc * H * H, K * Cc * k * k, N * K * Ho * Ho
        x, w, y, dx, dw = (L.mi_malloc(4 * n) for n in (nx, nw, ny, nx, nw))
        L.mi_op_fill_uniform(x, nx, 1, -1.0, 1.0)
        L.mi_op_fill_uniform(w, nw, 2, -0.1, 0.1)
        L.mi_op_fill_uniform(y, ny, 3, -1.0, 1.0)
        flops = 2.0 * k * k * N * Ho * Ho * Cc * K
        stem16 = Cc == 3 and os.environ.get("RESNET_MI_IGEMM", "2") != "0"  # the matrix-core stem (kernels_stem_bf16.hip): fp32 tensors; operands rounded to bf16 (--bf16) or exact fp32
        if args.bf16:
            xb, yb, dxb = (L.mi_malloc(2 * n) for n in (nx, ny, nx))
            L.mi_op_convert(x, 0, xb, 1, nx)
            L.mi_op_convert(y, 0, yb, 1, ny)
        for op in args.only.split(","):
            if op == "dgrad" and Cc == 3:
                continue
            L.mi_prof_enable(1)
            for rep in range(args.reps + 1):
                if rep == 1:
                    L.mi_prof_reset()
                if stem16 and args.bf16:
                    rc = L.mi_op_stem_fwd_bf16(x, w, y, N, H) if op == "fwd" else L.mi_op_stem_wgrad_bf16(x, w, y, dw, N, H)
                elif stem16:
                    rc = L.mi_op_stem_fwd_f32(x, w, y, N, H) if op == "fwd" else L.mi_op_stem_wgrad_f32(x, w, y, dw, N, H)
                elif args.bf16:
                    if op == "fwd":
                        rc = L.mi_op_conv_fwd_bf16(xb, w, yb, N, Cc, H, K, k, s)
                    elif op == "dgrad":
                        rc = L.mi_op_conv_dgrad_bf16(w, yb, dxb, N, Cc, H, K, k, s, 0)
                    else:
                        rc = L.mi_op_conv_wgrad_bf16(xb, yb, dw, N, Cc, H, K, k, s)
                elif op == "fwd":
                    rc = L.mi_op_conv_fwd(x, w, y, N, Cc, H, K, k, s)
                elif op == "dgrad":
                    rc = L.mi_op_conv_dgrad(w, y, dx, N, Cc, H, K, k, s, 0)
                else:
                    rc = L.mi_op_conv_wgrad(x, y, dw, N, Cc, H, K, k, s)
                if rc != 0:
                    raise SystemExit("%s %s failed: %s" % (name, op, L.mi_last_error().decode()))
            ms = 0.0
            for fam in (0, 1, 2, 5):
                n_, ms_, fl_, by_ = C.c_long(0), C.c_double(0), C.c_double(0), C.c_double(0)
                L.mi_prof_get(fam, C.byref(n_), C.byref(ms_), C.byref(fl_), C.byref(by_))
                ms += ms_.value
            L.mi_prof_enable(0)
            ms /= args.reps
            if stem16 and op == "wgrad":  # (the operator runs the forward first to build the padded planes)
                ms -= stem_fwd_ms
            if stem16 and op == "fwd":
                stem_fwd_ms = ms
            tot[op] += ms * cnt
            print("%-18s %-6s %9.3f %9.1f" % (name, op, ms, flops / ms / 1e9))
        for p in (x, w, y, dx, dw) + ((xb, yb, dxb) if args.bf16 else ()):
            L.mi_free(p)
    print("per-step totals (ms, weighted by layer count):", {k_: round(v, 1) for k_, v in tot.items()})
    if args.r18:
        import json
        bn_f = bn_b = 0.0
        for name, Cc, H, K, k, s, cnt in LAYERS:  # one batch norm behind every convolution, at its output shape
            f, b_ = bench_bn(L, N, K, H // s, args.reps)
            bn_f += f * cnt; bn_b += b_ * cnt
        conv = tot["fwd"] + tot["dgrad"] + tot["wgrad"]
        step = conv + bn_f + bn_b
        print(json.dumps({"config": "BASELINE configs[1]: ResNet-18 fp32 batch %d 224x224, 3x3 conv (+ 7x7 stem) and BN kernels only" % N,
                          "route": "RESNET_MI_IGEMM=%s (0 = direct VALU kernels, no MFMA; default 2 = MFMA implicit GEMM)" % os.environ.get("RESNET_MI_IGEMM", "2"),
                          "conv_ms": {k_: round(v, 3) for k_, v in tot.items()}, "bn_fwd_ms": round(bn_f, 3), "bn_bwd_ms": round(bn_b, 3),
                          "ms_per_step_these_kernels": round(step, 3), "images_per_sec_these_kernels": round(N / step * 1e3, 1),
                          "note": "isolated launches through mi_op_*, HIP-event time per kernel; the 1x1 stride-2 shortcuts, pools, FC and the optimizer "
                                  "are not part of what configs[1] names"}))


if __name__ == "__main__":
    main()
