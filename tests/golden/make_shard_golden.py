"""Writes tests/golden/shard_ref_golden.npz from the REFERENCE's own shard builder (oracle/_ref/build_training_shards,
compiled unmodified from /root/reference/build_training_shards.c by `make -C oracle ref`) run on the synthetic class files
of tests/test_shards.py.  Run in the build container (the reference does not travel):  python tests/golden/make_shard_golden.py"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_shards as T  # noqa: E402

subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
with tempfile.TemporaryDirectory() as root:
    T.write_inputs(root)
    env = dict(os.environ, LD_PRELOAD=T.REF_SHIM, MI_REF_ROOT=root)
    subprocess.run([T.REF_BIN], env=env, stdout=subprocess.DEVNULL, check=True)
    d = os.path.join(root, "data/vision/imagenet/2012/train_data_shards/nchw")
    img = np.fromfile(os.path.join(d, "000.images"), np.float32)
    lab = np.fromfile(os.path.join(d, "000.labels"), np.int32)
np.savez(os.path.join(HERE, "shard_ref_golden.npz"), labels=lab, n_floats=img.size, head=img[:64], tail=img[-64:],
         sha256=hashlib.sha256(img.tobytes()).hexdigest())
print("wrote shard_ref_golden.npz: %d images, %d floats" % (lab.size, img.size))
