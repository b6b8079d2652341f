"""the shard pipeline.  The shard builder is the one piece of the reference's data path that is plain C
(build_training_shards.c, gcc target BuildShards), so here parity is PINNED BY THE REFERENCE ITSELF: oracle/Makefile
compiles it unmodified from /root/reference into oracle/_ref/ (when the reference is present) and this test runs it, under
a preloaded shim that relocates its literal /mnt/storage paths, on synthetic class byte files -- the product's
mi_build_shard (resnet_amd/csrc/shards.c) must write the same %03d.images / %03d.labels BIT FOR BIT.  Where the reference
binary is absent (GPU box) the committed golden fixture it produced (tests/golden/shard_ref_*.npy, made by
tests/golden/make_shard_golden.py) and a numpy restatement of build_training_shards.c:88-144 carry the check.
CPU only: byte / integer / exact-float work."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "build_training_shards")
REF_SHIM = os.path.join(ROOT, "oracle", "_ref", "libmntredirect.so")
GOLD = os.path.join(ROOT, "tests", "golden")

DIM_IN, DIM_OUT = 256, 224
# (class, image number, row offset, col offset): corners, maximum offsets (32 = 256 - 224), repeats
ROWS = [(0, 0, 0, 0), (2, 3, 32, 32), (1, 1, 7, 19), (0, 2, 32, 0), (2, 0, 0, 32), (1, 0, 16, 16), (0, 0, 5, 31)]
N_CLASSES, IMGS_PER_CLASS = 3, 4


def class_bytes(c):
    """deterministic uint8 image bytes of class file c: IMGS_PER_CLASS images of 256x256x3 (B,G,R interleaved)"""
    rng = np.random.RandomState(1000 + c)
    return rng.randint(0, 256, size=(IMGS_PER_CLASS, DIM_IN, DIM_IN, 3), dtype=np.uint8)


def write_inputs(root):
    part = os.path.join(root, "data/vision/imagenet/2012/train_data_partioning")
    data = os.path.join(root, "data/vision/imagenet/2012/train_data")
    os.makedirs(part)
    os.makedirs(data)
    with open(os.path.join(part, "000_images.csv"), "w") as f:
        for c, n, r, s in ROWS:
            f.write("%03d,%04d,%02d,%02d\n" % (c, n, r, s))
    for c in range(N_CLASSES):
        class_bytes(c).tofile(os.path.join(data, "%08d.buffer" % c))
    return part, data


def numpy_restatement(layout_nchw=True):
    """build_training_shards.c:88-144 in numpy: crop, B,G,R -> R,G,B minus (103.94, 116.78, 123.68), NHWC -> NCHW"""
    out = []
    for c, n, r, s in ROWS:
        crop = class_bytes(c)[n, r:r + DIM_OUT, s:s + DIM_OUT, :].astype(np.float32)  # (h, w, BGR)
        rgb = np.empty_like(crop)
        # the subtraction happens in double and is rounded once (`((float) byte) - 123.68`)
        rgb[..., 2] = (crop[..., 0].astype(np.float64) - 123.68).astype(np.float32)
        rgb[..., 1] = (crop[..., 1].astype(np.float64) - 116.78).astype(np.float32)
        rgb[..., 0] = (crop[..., 2].astype(np.float64) - 103.94).astype(np.float32)
        out.append(np.transpose(rgb, (2, 0, 1)) if layout_nchw else rgb)
    return np.stack(out), np.array([c for c, _, _, _ in ROWS], np.int32)


def run_product(tmp, layout):
    from resnet_amd import binding as B
    lib = B.load()
    part, data = write_inputs(os.path.join(tmp, "in"))
    outdir = os.path.join(tmp, "out")
    os.makedirs(outdir)
    rc = lib.mi_build_shard(os.path.join(part, "000_images.csv").encode(), data.encode(), outdir.encode(), 0, DIM_IN, DIM_OUT, layout)
    assert rc == len(ROWS)
    return (np.fromfile(os.path.join(outdir, "000.images"), np.float32), np.fromfile(os.path.join(outdir, "000.labels"), np.int32))


def test_shard_builder_matches_numpy_restatement(tmp_path):
    from resnet_amd import binding as B
    img, lab = run_product(str(tmp_path), B.MI_LAYOUT_NCHW)
    ref_img, ref_lab = numpy_restatement(True)
    assert np.array_equal(lab, ref_lab)
    assert np.array_equal(img.view(np.uint32), ref_img.ravel().view(np.uint32))  # bit-exact
    img2, _ = run_product(str(tmp_path / "nhwc"), B.MI_LAYOUT_NHWC)
    assert np.array_equal(img2.view(np.uint32), numpy_restatement(False)[0].ravel().view(np.uint32))
    assert img.min() >= -124.0 and img.max() <= 152.0  # the range the synthetic source draws from


def test_shard_builder_matches_the_reference_golden_fixture(tmp_path):
    """tests/golden/shard_ref_*.npy were written by the REFERENCE binary (tests/golden/make_shard_golden.py): a hash of the
    whole images file plus the first / last 64 floats and all labels"""
    import hashlib
    from resnet_amd import binding as B
    img, lab = run_product(str(tmp_path), B.MI_LAYOUT_NCHW)
    gold = np.load(os.path.join(GOLD, "shard_ref_golden.npz"))
    assert np.array_equal(lab, gold["labels"])
    assert img.size == int(gold["n_floats"])
    assert np.array_equal(img[:64].view(np.uint32), gold["head"].view(np.uint32))
    assert np.array_equal(img[-64:].view(np.uint32), gold["tail"].view(np.uint32))
    assert hashlib.sha256(img.tobytes()).hexdigest() == str(gold["sha256"])


@pytest.mark.skipif(not (os.path.exists(REF_BIN) and os.path.exists(REF_SHIM)), reason="oracle/_ref not built (make -C oracle ref needs /root/reference)")
def test_shard_builder_matches_the_reference_binary(tmp_path):
    from resnet_amd import binding as B
    img, lab = run_product(str(tmp_path / "prod"), B.MI_LAYOUT_NCHW)
    root = str(tmp_path / "mnt")
    write_inputs(root)
    env = dict(os.environ, LD_PRELOAD=REF_SHIM, MI_REF_ROOT=root)
    subprocess.run([REF_BIN], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=300)
    shards = os.path.join(root, "data/vision/imagenet/2012/train_data_shards/nchw")
    rimg = np.fromfile(os.path.join(shards, "000.images"), np.float32)
    rlab = np.fromfile(os.path.join(shards, "000.labels"), np.int32)
    assert np.array_equal(lab, rlab)
    assert np.array_equal(img.view(np.uint32), rimg.view(np.uint32))
    assert os.path.getsize(os.path.join(shards, "001.images")) == 0  # shards without a partition file come out empty


def test_error_paths(tmp_path):
    from resnet_amd import binding as B
    lib = B.load()
    assert lib.mi_build_shard(b"/nonexistent.csv", b"/tmp", str(tmp_path).encode(), 0, 256, 224, 1) == -1
    csv = tmp_path / "000_images.csv"
    csv.write_text("005,0000,00,00\n")
    assert lib.mi_build_shard(str(csv).encode(), str(tmp_path).encode(), str(tmp_path).encode(), 0, 256, 224, 1) == -2  # no class file
