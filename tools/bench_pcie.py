#!/usr/bin/env python3
"""PCIe-inclusive step rate: the same ResNet-50 training step as bench.py, but every batch comes from a shard file
through host RAM (pinned staging -> H2D), blocking like the reference (resnet.cu:1315) and with the copy of
batch t+1 overlapped with step t (mi_batch_set_prefetch).  Not the headline metric (bench.py keeps inputs
resident in HBM); recorded in DESIGN.md."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resnet_amd import Trainer, resnet_dims  # noqa: E402
from resnet_amd import binding as B  # noqa: E402

batch, per_shard, steps = 256, 1024, 6
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
rng = np.random.default_rng(0)
img = rng.uniform(-124, 152, size=(per_shard, 3 * 224 * 224)).astype(np.float32)
img.tofile(os.path.join(d, "000.images"))
rng.integers(0, 1000, size=per_shard).astype(np.int32).tofile(os.path.join(d, "000.labels"))
del img
for prefetch in (False, True):
    tr = Trainer(resnet_dims(), batch, shard_n_images=per_shard, seed=1236)
    tr.source_shards(d, B.MI_LAYOUT_NCHW, prefetch=prefetch)
    tr.step()  # loads the shard, warms up
    tr.L.mi_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        tr.step()
    tr.L.mi_device_synchronize()
    dt = (time.perf_counter() - t0) / 3
    tr.check()
    print("shard source, prefetch=%s: %.1f ms/step, %.0f images/sec" % (prefetch, dt * 1e3, batch / dt))
    tr.close()
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
