"""The callers / data formats either side of the hot path: the reference's batch and shard
files (resnet.cu:1275-1316, build_training_shards.c:150-160, Appendix B) and its dump / resume format
(resnet.cu:2250-2875)."""
import os

import numpy as np
import pytest

import synth
from util import nchw

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _trainer(batch, **kw):
    from resnet_amd import Trainer
    tr = Trainer(synth.C1_DIMS, batch, seed=1236, **kw)
    if tr.L.mi_device_count() < 1:
        pytest.fail("needs the MI355X box")
    return tr


def test_images_buffer_labels_buffer(tmp_path):
    """one dumped batch as the reference writes it (commented hook resnet.cu:1301-1311; inspect_input.ipynb reads it as
    (N, H, W, 3) fp32 + N int32).  labels.buffer is the reference's own 32-label fixture."""
    from resnet_amd import binding as B
    N = 32
    lab = np.fromfile(os.path.join(GOLD, "labels.buffer"), dtype=np.int32)
    im, _ = synth.make_batch(synth.C1_DIMS, N)
    ip, lp = tmp_path / "images.buffer", tmp_path / "labels.buffer"
    im.tofile(ip)
    lab.tofile(lp)
    tr = _trainer(N)
    tr.source_buffer(str(ip), str(lp), B.MI_LAYOUT_NHWC)
    tr.load_new_batch()
    assert tr.L.mi_batch_last_status(tr.c_batch) == 0
    assert np.array_equal(tr.activation("input"), nchw(im))
    assert np.array_equal(tr.labels(), lab) and tr.labels()[0] == 457 and tr.labels()[31] == 546
    tr.forward()
    loss, _ = tr.loss()
    assert np.isfinite(loss)
    tr.close()


@pytest.mark.parametrize("prefetch", [False, True])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_shard_rotation(tmp_path, layout, prefetch):
    """%03d.images / %03d.labels shards (build_training_shards.c:150-160): whole shard in host RAM, batches in order,
    next shard when the current one is exhausted (resnet.cu:1266-1295); a missing shard is reported, not dereferenced"""
    from resnet_amd import binding as B
    batch, per_shard = 4, 8
    shards = []
    for sid in range(2):
        im, lab = synth.make_batch(synth.C1_DIMS, per_shard, seed_img=100 + sid, seed_lab=200 + sid)
        (nchw(im) if layout == "nchw" else im).tofile(tmp_path / ("%03d.images" % sid))
        lab.tofile(tmp_path / ("%03d.labels" % sid))
        shards.append((im, lab))
    tr = _trainer(batch, shard_n_images=per_shard)
    tr.source_shards(str(tmp_path), B.MI_LAYOUT_NCHW if layout == "nchw" else B.MI_LAYOUT_NHWC, prefetch=prefetch)
    for step in range(4):
        tr.load_new_batch()
        assert tr.L.mi_batch_last_status(tr.c_batch) == 0
        sid, b = divmod(step, per_shard // batch)
        im, lab = shards[sid]
        assert np.array_equal(tr.activation("input"), nchw(im[b * batch:(b + 1) * batch])), "step %d" % step
        assert np.array_equal(tr.labels(), lab[b * batch:(b + 1) * batch])
        assert tr.c_batch.contents.cur_shard_id == sid
        assert tr.t.contents.cur_dump_id == step  # ++cur_dump_id per load (resnet.cu:1322)
        if prefetch:  # a full step between loads, as in training: the next batch is copied underneath it
            tr.forward(); tr.backward(); tr.update()
    tr.load_new_batch()  # shard 002 does not exist
    assert tr.L.mi_batch_last_status(tr.c_batch) == -1
    tr.close()


def test_dump_and_resume(tmp_path):
    """dump_trainer writes the reference's tree; overwrite_* restores a second trainer that then continues identically"""
    from resnet_amd import Trainer
    dims, batch = synth.C1_DIMS, 4
    root = str(tmp_path)

    def fresh():
        t = Trainer(dims, batch, seed=1236, dump_dir="run")
        t.source_synthetic(1234, 1235, pool_batches=4)
        t.L.mi_trainer_set_dump_root(t.t, root.encode())
        return t

    a = fresh()
    for _ in range(2):
        a.step()
    a.load_new_batch(); a.forward(); a.backward()  # state in the middle of step 3, gradients populated
    a.L.dump_trainer(2, a.t, b"run")
    d = os.path.join(root, "run", "%08d" % 2)
    n_loc = a.n_locations
    for sub in ("model_params", "gradients", "means", "vars"):
        for i in range(n_loc):
            f = os.path.join(d, sub, "%03d.buffer" % i)
            assert os.path.getsize(f) == 4 * a.sizes[i], f
    assert np.array_equal(np.fromfile(os.path.join(d, "model_params", "%03d.buffer" % (n_loc - 1)), np.float32), a.get("params", n_loc - 1))
    assert np.any(np.fromfile(os.path.join(d, "gradients", "000.buffer"), np.float32))
    # activation tree, image tensors NHWC like the reference (Appendix B of SURVEY.md)
    x = np.fromfile(os.path.join(d, "activations", "conv_blocks", "00", "output_activated.buffer"), np.float32).reshape(batch, 8, 8, 256)
    assert np.array_equal(nchw(x), a.activation("conv_blocks/00/output_activated"))
    assert os.path.getsize(os.path.join(d, "activations", "max_inds.buffer")) == 4 * batch * 64 * 8 * 8
    assert os.path.getsize(os.path.join(d, "activations", "batch_norms", "init", "means.buffer")) == 4 * 64
    ck = open(os.path.join(d, "trainer_checkpoint.txt")).read().split()
    assert len(ck) == 6 and int(ck[4]) == a.t.contents.cur_dump_id
    assert len(open(os.path.join(d, "trainer_metadata.txt")).read().splitlines()) == 16
    a.update()
    ref_losses = [a.step()[0] for _ in range(2)]
    ref_param = a.get("params", 3)
    a.close()

    b = fresh()
    b.L.overwrite_trainer_hyperparams(b.t, 2, b"run")
    b.L.overwrite_model_params(b.t, 2, b"run")
    assert b.t.contents.init_loaded == 1 and b.t.contents.cur_dump_id == 2
    # the synthetic pool has no shard position: replay the two consumed batches, then redo the interrupted step
    b.t.contents.cur_dump_id = -1
    for _ in range(2):
        b.load_new_batch()
    b.load_new_batch(); b.forward(); b.backward(); b.update()
    got = [b.step()[0] for _ in range(2)]
    # the reference's checkpoint stores cur_mean_decay / cur_var_decay with "%f" (6 decimals, resnet.cu:2747-2748), so
    # the resumed bias correction differs in the 7th digit: same trajectory to ~1e-6, not bit-identical
    assert got[0] == ref_losses[0]
    assert np.allclose(got, ref_losses, rtol=1e-5), (got, ref_losses)
    assert np.allclose(b.get("params", 3), ref_param, rtol=1e-4, atol=1e-6)
    b.close()


def test_reference_style_c_driver(tmp_path):
    """examples/resnet_main.c -- the reference's main() rebuilt on the C-ABI: plain C, gcc, linked against the library;
    prints the reference's per-iteration line (resnet.cu:3386) and writes avg_loss_log.txt (resnet.cu:3388)"""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ResNetMI")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "resnet_main.c"),
                           "-L", os.path.join(root, "resnet_amd"), "-lresnet_mi", "-lm",
                           "-Wl,-rpath," + os.path.join(root, "resnet_amd"), "-o", exe])
    log = str(tmp_path / "avg_loss_log.txt")
    out = subprocess.check_output([exe, "--input", "32", "--blocks", "1", "--batch", "8", "--iters", "6", "--loss-log", log],
                                  timeout=300).decode()
    lines = re.findall(r"Epoch: 0, Batch: (\d+) ----- Avg\. Loss: ([0-9.]+), Accuracy: ([0-9.]+)%", out)
    assert [int(a) for a, _, _ in lines] == list(range(6))
    losses = [float(x) for x in open(log).read().split()]
    assert len(losses) == 6 and all(np.isfinite(losses)) and [float(b) for _, b, _ in lines] == losses
    assert 6.0 < losses[0] < 8.0  # ln(1000) = 6.9 at random init


def test_c_driver_takes_iterations_per_epoch_from_the_class_counts(tmp_path):
    """resnet.cu:3236-3242, 3309: total_images = sum of id_to_img_count_mapping.txt, iterations_per_epoch = ceil(total / BATCH_SIZE).
    Three class files (labels / synsets / counts, one line per class), 1000 classes of which 21 hold one image, batch 8 ->
    3 iterations per epoch; two epochs -> 6 iteration lines, the second epoch numbered from 0 again."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ResNetMI")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "resnet_main.c"),
                           "-L", os.path.join(root, "resnet_amd"), "-lresnet_mi", "-lm",
                           "-Wl,-rpath," + os.path.join(root, "resnet_amd"), "-o", exe])
    n_classes = 1000
    (tmp_path / "labels.txt").write_text("".join("label %d\n" % i for i in range(n_classes)))
    (tmp_path / "synsets.txt").write_text("".join("n%08d\n" % i for i in range(n_classes)))
    (tmp_path / "counts.txt").write_text("1\n" * 21 + "0\n" * (n_classes - 21))
    out = subprocess.check_output([exe, "--input", "32", "--blocks", "1", "--batch", "8", "--classes", str(n_classes), "--epochs", "2",
                                   "--labels-file", str(tmp_path / "labels.txt"), "--synsets-file", str(tmp_path / "synsets.txt"),
                                   "--counts-file", str(tmp_path / "counts.txt"), "--loss-log", str(tmp_path / "log.txt")], timeout=300).decode()
    assert "class metadata: 1000 classes, 21 images" in out and "iterations per epoch: 3" in out
    lines = re.findall(r"Epoch: (\d+), Batch: (\d+) -----", out)
    assert [(int(a), int(b)) for a, b in lines] == [(0, 0), (0, 1), (0, 2), (1, 0), (1, 1), (1, 2)]
