"""Generates tests/golden/c1_golden.npz from the CPU oracle (double-accumulation build) on BASELINE.json
configs[0]: 1 bottleneck block, batch 4, 32x32 seeded synthetic input.  The reference itself cannot run
here (CUDA-only), so these vectors pin the ORACLE's behaviour over time, not the reference's
("parity unpinned", oracle/oracle.h).  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth  # noqa: E402
from oracle.oracle_py import Oracle, OracleNet  # noqa: E402

o = Oracle("f64")
dims, batch = synth.C1_DIMS, synth.C1_BATCH
params = synth.make_params(dims)
net = OracleNet(o, dims, batch)
for i, p in enumerate(params):
    net.param(i)[:] = p
im, lab = synth.make_batch(dims, batch)
net.set_batch(im, lab)
net.forward()
loss, wrong = net.loss()
logits = net.tensor("fc_output").reshape(batch, -1).copy()
net.backward()
gsum = np.array([np.abs(net.grad(i)).sum() for i in range(net.n_locations)], np.float64)
g0 = net.grad(0).copy()
gfc = net.grad(net.n_locations - 1).reshape(256, 1000)[:, :16].copy()
net.update()
np.savez_compressed(os.path.join(HERE, "c1_golden.npz"), images_probe=im[0, :2, :2].ravel(), labels=lab, loss=np.float64(loss),
                    n_wrong=np.int32(wrong), logits=logits, grad_abs_sum=gsum, grad_init_conv=g0, grad_fc_probe=gfc,
                    param3_after=net.param(3).copy())
print("loss", loss, "wrong", wrong)
