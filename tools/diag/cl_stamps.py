#!/usr/bin/env python3
"""Where wave 0 of a cl_conv_kernel workgroup spends its cycles, from in-kernel s_memtime stamps (diagnostic build:
   tools/variant.sh stamp "-DCL_STAMP" kernels_cl_bf16.hip;  RESNET_MI_LIB=variants/libresnet_mi_stamp.so python tools/diag/cl_stamps.py C H K stride)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from resnet_amd import binding as B
L = B.load()
raw = C.CDLL(os.environ["RESNET_MI_LIB"])
Cc, H, K, S = (int(a) for a in sys.argv[1:5])
N = 256; Ho = H // S
nx, nw, ny = N * Cc * H * H, K * Cc * 9, N * K * Ho * Ho
x, w = L.mi_malloc(4 * nx), L.mi_malloc(4 * nw)
L.mi_op_fill_uniform(x, nx, 1, -1.0, 1.0); L.mi_op_fill_uniform(w, nw, 2, -0.1, 0.1)
xb, yb = L.mi_malloc(2 * nx), L.mi_malloc(2 * ny)
L.mi_op_convert(x, 0, xb, 1, nx)
for _ in range(3):
    assert L.mi_op_conv_fwd_bf16_cl(xb, w, yb, N, Cc, H, K, S) == 0
nb = 4096
buf = (C.c_ulonglong * (8 * nb))()
raw.mi_debug_cl_stamps.argtypes = [C.c_void_p, C.c_int]
assert raw.mi_debug_cl_stamps(buf, nb) == 0
t = np.array(buf, dtype=np.uint64).reshape(nb, 8).astype(np.float64)
t = t[t[:, 6] > 0]
tot = t[:, 1] - t[:, 0]
print("workgroups with stamps: %d, k-steps per workgroup %d" % (len(t), int(t[0, 6])))
print("main loop (first issue .. last MFMA issued)  median %8.0f cycles  = %.0f per k-step (64 MFMAs of 32 cycles on 4 SIMDs = 512 at full rate)" % (np.median(tot), np.median(tot / t[:, 6])))
for i, n in ((2, "s_waitcnt vmcnt(0)"), (3, "barrier"), (4, "issue 8 DMA + addresses"), (5, "16 ds_read + 16 MFMA issue")):
    print("%-28s median %8.0f cycles  %5.1f %% of the loop   per k-step %6.0f" % (n, np.median(t[:, i]), 100 * np.median(t[:, i] / tot), np.median(t[:, i] / t[:, 6])))
