"""diagnostic: 1x1 forward on a dense channel-last input (one tap of cl_conv_kernel; the re-layout is NOT in the timed bracket: the question is
what the layer costs once activations are channel-last) against the NCHW kernel, benchmark shapes, N = 256"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from resnet_amd import binding as B
L = B.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
LAY = [("64_256@56", 64, 56, 256), ("256_64@56", 256, 56, 64), ("512_128@28", 512, 28, 128), ("128_512@28", 128, 28, 512), ("1024_256@14", 1024, 14, 256),
       ("256_1024@14", 256, 14, 1024), ("2048_512@7", 2048, 7, 512), ("512_2048@7", 512, 7, 2048)]
def fam_ms(fams):
    ms = 0.0
    for fam in fams:
        n_, ms_, fl_, by_ = C.c_long(0), C.c_double(0), C.c_double(0), C.c_double(0)
        L.mi_prof_get(fam, C.byref(n_), C.byref(ms_), C.byref(fl_), C.byref(by_)); ms += ms_.value
    return ms
for name, Cc, H, K in LAY:
    nx, nw, ny = N * Cc * H * H, K * Cc, N * K * H * H
    x, w = L.mi_malloc(4 * nx), L.mi_malloc(4 * nw)
    L.mi_op_fill_uniform(x, nx, 1, -1.0, 1.0); L.mi_op_fill_uniform(w, nw, 2, -0.1, 0.1)
    xb, yb = L.mi_malloc(2 * nx), L.mi_malloc(2 * ny)
    L.mi_op_convert(x, 0, xb, 1, nx)
    flops = 2.0 * N * H * H * Cc * K
    out = []
    for which, fn, fams in (("nchw", lambda: L.mi_op_conv_fwd_bf16(xb, w, yb, N, Cc, H, K, 1, 1), (2,)), ("channel-last", lambda: L.mi_op_conv1x1_fwd_bf16_cl(xb, w, yb, N, Cc, H, K), (5,))):
        L.mi_prof_enable(1)
        for rep in range(4):
            if rep == 1: L.mi_prof_reset()
            rc = fn()
            assert rc == 0, (name, which, rc, L.mi_last_error())
        ms = fam_ms(fams) / 3; L.mi_prof_enable(0)
        out.append("%s %.3f ms %.0f TF/s" % (which, ms, flops / ms / 1e9))
    print("%-12s" % name, " | ".join(out), flush=True)
    for p in (x, w, xb, yb): L.mi_free(p)
