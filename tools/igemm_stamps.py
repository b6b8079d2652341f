#!/usr/bin/env python3
"""Phase timing of igemm_kernel from in-kernel s_memtime stamps (diagnostic build: tools/variant.sh stamp "-DIG_STAMP" kernels_igemm.hip).
   RESNET_MI_LIB=variants/libresnet_mi_stamp.so python tools/igemm_stamps.py C H K k stride fwd|dgrad   (values: shader-clock cycles)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resnet_amd import binding as B
L = B.load()
raw = C.CDLL(os.environ["RESNET_MI_LIB"])
Cc, H, K, k, s, op = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
N = 256; Ho = H // s
nx, nw, ny = N * Cc * H * H, K * Cc * k * k, N * K * Ho * Ho
x, w, y, dx = (L.mi_malloc(4 * n) for n in (nx, nw, ny, nx))
L.mi_op_fill_uniform(x, nx, 1, -1.0, 1.0); L.mi_op_fill_uniform(w, nw, 2, -0.1, 0.1); L.mi_op_fill_uniform(y, ny, 3, -1.0, 1.0)
for _ in range(3):
    rc = L.mi_op_conv_fwd(x, w, y, N, Cc, H, K, k, s) if op == "fwd" else L.mi_op_conv_dgrad(w, y, dx, N, Cc, H, K, k, s, 0)
    assert rc == 0
nb = 2048
buf = (C.c_ulonglong * (8 * nb))()
raw.mi_debug_igemm_stamps.argtypes = [C.c_void_p, C.c_int]
assert raw.mi_debug_igemm_stamps(buf, nb) == 0
t = np.array(buf, dtype=np.uint64).reshape(nb, 8).astype(np.float64)
t = t[t[:, 4] > 0]
d = np.diff(t[:, :5], axis=1)  # s_memtime ticks = shader-clock cycles on gfx950
names = ["setup+prologue", "main loop", "epilogue stores", "bn stats"]
print("workgroups with stamps: %d (stamps of earlier launches persist: only per-workgroup differences are meaningful)" % len(t))
for i, n in enumerate(names):
    print("%-16s median %8.0f cycles   p10 %8.0f   p90 %8.0f" % (n, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
print("workgroup total  median %8.0f cycles" % np.median(t[:, 4] - t[:, 0]))
