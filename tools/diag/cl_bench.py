"""diagnostic: the channel-last 3x3/s2 forward kernel against the NCHW kernel on the stride-2 layers of the benchmark network (N = 256)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from resnet_amd import binding as B
L = B.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
LAY = [("b3_3x3s2", 128, 56, 128, 2), ("b3_proj", 256, 56, 512, 2), ("b7_3x3s2", 256, 28, 256, 2), ("b7_proj", 512, 28, 1024, 2),
       ("b13_3x3s2", 512, 14, 512, 2), ("b13_proj", 1024, 14, 2048, 2),
       ("s0_3x3", 64, 56, 64, 1), ("s1_3x3", 128, 28, 128, 1), ("s2_3x3", 256, 14, 256, 1), ("s3_3x3", 512, 7, 512, 1)]
def fam_ms():
    ms = 0.0
    for fam in (0, 1, 2, 5):
        n_, ms_, fl_, by_ = C.c_long(0), C.c_double(0), C.c_double(0), C.c_double(0)
        L.mi_prof_get(fam, C.byref(n_), C.byref(ms_), C.byref(fl_), C.byref(by_)); ms += ms_.value
    return ms
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, Cc, H, K, S in LAY:
    if ONLY and name not in ONLY: continue
    Ho = H // S
    nx, nw, ny = N * Cc * H * H, K * Cc * 9, N * K * Ho * Ho
    x, w, y = (L.mi_malloc(4 * n) for n in (nx, nw, ny))
    L.mi_op_fill_uniform(x, nx, 1, -1.0, 1.0); L.mi_op_fill_uniform(w, nw, 2, -0.1, 0.1)
    xb, yb = L.mi_malloc(2 * nx), L.mi_malloc(2 * ny)
    L.mi_op_convert(x, 0, xb, 1, nx)
    L.mi_op_fill_uniform(y, ny, 3, -1.0, 1.0); L.mi_op_convert(y, 0, yb, 1, ny)
    flops = 2.0 * 9 * N * Ho * Ho * Cc * K
    out = []
    ops_ = [("fwd nchw", lambda: L.mi_op_conv_fwd_bf16(xb, w, yb, N, Cc, H, K, 3, S)), ("fwd cl", lambda: L.mi_op_conv_fwd_bf16_cl(xb, w, yb, N, Cc, H, K, S))]
    if S == 1:
        ops_ += [("dgrad nchw", lambda: L.mi_op_conv_dgrad_bf16(w, yb, xb, N, Cc, H, K, 3, 1, 0)), ("dgrad cl", lambda: L.mi_op_conv_dgrad_bf16_cl(w, yb, xb, N, Cc, H, K, 1, 0))]
    else:
        ops_ += [("dgrad nchw", lambda: L.mi_op_conv_dgrad_bf16(w, yb, xb, N, Cc, H, K, 3, 2, 0)), ("dgrad cl", lambda: L.mi_op_conv_dgrad_bf16_cl(w, yb, xb, N, Cc, H, K, 2, 0))]
    dw = L.mi_malloc(4 * nw)
    ops_ += [("wgrad nchw", lambda: L.mi_op_conv_wgrad_bf16(xb, yb, dw, N, Cc, H, K, 3, S))]
    if Cc % 128 == 0 and K % 128 == 0 and (Ho * Ho) % 4 == 0:
        ops_ += [("wgrad cl", lambda: L.mi_op_conv_wgrad_bf16_cl(xb, yb, dw, N, Cc, H, K, S))]
    if Cc % 128 == 0 and K % 128 == 0:
        ops_ += [("wgrad cl2", lambda: L.mi_op_conv_wgrad_bf16_cl2(xb, yb, dw, N, Cc, H, K, S))]
    for which, fn in ops_:
        L.mi_prof_enable(1)
        for rep in range(4):
            if rep == 1: L.mi_prof_reset()
            rc = fn()
            assert rc == 0, (name, which, rc, L.mi_last_error())
        ms = fam_ms() / 3; L.mi_prof_enable(0)
        out.append("%s %.3f ms %.0f TF/s" % (which, ms, flops / ms / 1e9))
    print("%-10s" % name, " | ".join(out), flush=True)
    L.mi_free(dw)
    for p in (x, w, y, xb, yb): L.mi_free(p)
