"""Trainer state and the data path either side of a step, through the C-ABI on the GPU:

* Adam pinned on its own: the ORACLE's gradients injected into the gradient arena, parameters / first / second moments after
  update_parameters against the oracle's at 1e-6 (resnet.cu:605-662, 2910-2987) -- the whole-step tests compare gradients; what
  Adam does with them is measured here, where no gradient rounding is in play.
* Per-trainer state: two trainers in one process, a host write into one's parameters between ITS forward and backward while the
  other runs a forward pass in between; the first one's backward must use the NEW weights (oracle run the same way).
* check_errors (resnet.cu:2879-2907): the offending locations[] index is reported, and the 99999999 dump holds the failing
  step (the flag is read when load_new_batch is entered, before the batch is replaced).
* Shard loader: prefetch with many batches per shard under full training steps (inputs, labels and losses equal to the
  blocking loader's), and per-rank slices of a shard for data parallel runs (resnet.cu:1266-1299 with a rank offset).
"""
import os

import numpy as np
import pytest

import synth
from util import nchw, rel_l2

pytestmark = pytest.mark.gpu
HYPER = dict(lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7)


def _pair(dims, batch, oracle, wd=0.0):
    from oracle.oracle_py import OracleNet
    from resnet_amd import Trainer
    from resnet_amd import binding as B
    hyper = dict(HYPER, wd=wd)
    params = synth.make_params(dims, perturb_bn=True)
    net = OracleNet(oracle, dims, batch)
    net.set_hyper(hyper["lr"], hyper["wd"], hyper["b1"], hyper["b2"], hyper["eps"])
    tr = Trainer(dims, batch, **hyper)
    if tr.L.mi_device_count() < 1:
        pytest.fail("no HIP device: this test must run on the MI355X box")
    for i, p in enumerate(params):
        net.param(i)[:] = p
    tr.set_params(params)
    tr.source_host(B.MI_LAYOUT_NHWC)
    return net, tr


@pytest.mark.parametrize("cfg,wd", [("C1", 0.0), ("C1S", 0.0), ("C1S", 1e-3)])
def test_adam_pinned_with_the_oracles_gradients(oracle, cfg, wd):
    """Two updates.  Before each, the gradient arena is overwritten with the oracle's gradients, so both sides run Adam on the
    SAME numbers from the SAME state: parameters, means and vars must agree to 1e-6 rel-L2 and 2 ulp-level max error, decays
    advance before use (step t divides by 1 - beta^t, resnet.cu:2920-2921), the arena is cleared afterwards."""
    dims, batch = (synth.C1_DIMS, 4) if cfg == "C1" else (synth.C1S_DIMS, 4)
    net, tr = _pair(dims, batch, oracle, wd=wd)
    try:
        for step in range(2):
            im, lab = synth.make_batch(dims, batch, step=step)
            net.set_batch(im, lab); net.forward(); net.backward()
            tr.fill_host_batch(im, lab); tr.load_new_batch(); tr.forward(); tr.backward(); tr.check()
            for i in range(net.n_locations):
                tr.set("grads", i, net.grad(i))
            net.update(); tr.update()
            assert tr.check_errors() == 0
            for i in range(net.n_locations):
                for what, got, ref in (("param", tr.get("params", i), net.param(i)), ("mean", tr.get("means", i), net.mean(i)),
                                       ("var", tr.get("vars", i), net.var(i))):
                    assert rel_l2(got, ref) <= 1e-6, "%s %d step %d: %.3e" % (what, i, step, rel_l2(got, ref))
                assert not np.any(tr.get("grads", i))
            c = tr.t.contents
            assert abs(c.cur_mean_decay - 0.9 ** (step + 1)) < 1e-6 and abs(c.cur_var_decay - 0.999 ** (step + 1)) < 1e-6
            # the next step starts from the oracle's state on both sides (the parameters differ in the last bit otherwise)
            for i in range(net.n_locations):
                tr.set("params", i, net.param(i)); tr.set("means", i, net.mean(i)); tr.set("vars", i, net.var(i))
    finally:
        tr.close(); net.close()


def test_two_trainers_interleaved_with_a_host_write_between_forward_and_backward(oracle):
    """A.forward -> host write into A's convolution weights -> B.forward -> A.backward.  The re-laid weight copies A's dgrad kernels
    read were made for the OLD weights; the staleness flag is per trainer, so B's forward pass cannot clear it.  Oracle run the
    same way: forward with the old weights, weights replaced, backward."""
    dims, batch = synth.C1S_DIMS, 4
    netA, A = _pair(dims, batch, oracle)
    netB, B_ = _pair(dims, batch, oracle)
    try:
        imA, labA = synth.make_batch(dims, batch, step=0)
        imB, labB = synth.make_batch(dims, batch, step=1)
        A.fill_host_batch(imA, labA); A.load_new_batch(); A.forward()
        netA.set_batch(imA, labA); netA.forward()
        table = synth.location_table(dims)
        changed = [i for i, (_, kind, _) in enumerate(table) if kind == "w" and i > 0]
        for i in changed:
            new = (netA.param(i) * np.float32(1.5)).astype(np.float32)
            netA.param(i)[:] = new
            A.set("params", i, new)
        B_.fill_host_batch(imB, labB); B_.load_new_batch(); B_.forward()  # (clears a process-global flag, if there were one)
        netB.set_batch(imB, labB); netB.forward()
        A.backward(); A.check(); netA.backward()
        B_.backward(); B_.check(); netB.backward()
        for i in range(netA.n_locations):
            assert rel_l2(A.get("grads", i), netA.grad(i)) <= 1e-4, "trainer A, gradient %d: %.3e" % (i, rel_l2(A.get("grads", i), netA.grad(i)))
            assert rel_l2(B_.get("grads", i), netB.grad(i)) <= 1e-4, "trainer B, gradient %d" % i
    finally:
        A.close(); B_.close(); netA.close(); netB.close()


def test_nan_report_names_the_location_and_dumps_the_failing_step(tmp_path):
    """check_errors (resnet.cu:2879-2907) prints `location: %d` for the first offending tensor of its walk from the last location
    to the first (:2952), dumps to id 99999999 and exits.  Here (exit turned off for the test): NaN in the gradients of locations
    3 and 11 -> 11 is named; the dump's checkpoint carries the FAILING step's dump id and its activations."""
    from resnet_amd import Trainer
    dims, batch = synth.C1S_DIMS, 4
    tr = Trainer(dims, batch, seed=1236, dump_dir="run")
    tr.source_synthetic(1234, 1235, pool_batches=4)
    tr.L.mi_trainer_set_dump_root(tr.t, str(tmp_path).encode())
    tr.L.mi_trainer_set_nan_exit(tr.t, 0)
    try:
        tr.step()
        assert tr.L.mi_trainer_nan_location(tr.t) == -1
        tr.load_new_batch(); tr.forward(); tr.backward()
        failing_id = tr.t.contents.cur_dump_id
        act = tr.activation("conv_blocks/00/output_activated")
        for loc in (3, 11):
            g = tr.get("grads", loc)
            g[g.size // 2] = np.nan if loc == 3 else np.inf
            tr.set("grads", loc, g)
        tr.update()
        tr.load_new_batch()  # reads the flag before it touches the batch
        assert tr.L.mi_trainer_nan_location(tr.t) == 11
        d = os.path.join(str(tmp_path), "run", "%08d" % 99999999)
        ck = open(os.path.join(d, "trainer_checkpoint.txt")).read().split()
        assert int(ck[4]) == failing_id
        x = np.fromfile(os.path.join(d, "activations", "conv_blocks", "00", "output_activated.buffer"), np.float32)
        assert np.array_equal(nchw(x.reshape(batch, 8, 8, 256)), act)
        # the offending gradients are still in the dump (Adam clears only finite ones)
        g3 = np.fromfile(os.path.join(d, "gradients", "003.buffer"), np.float32)
        assert np.isnan(g3[g3.size // 2]) and np.count_nonzero(g3) == 1
        # the run goes on from a clean flag
        tr.forward(); tr.backward(); tr.set("grads", 3, np.zeros(tr.sizes[3], np.float32)); tr.set("grads", 11, np.zeros(tr.sizes[11], np.float32))
        tr.update()
        assert tr.check_errors() == 0
    finally:
        tr.close()


def _write_shards(tmp_path, n_shards, per_shard, layout="nchw"):
    shards = []
    for sid in range(n_shards):
        im, lab = synth.make_batch(synth.C1_DIMS, per_shard, seed_img=300 + sid, seed_lab=400 + sid)
        (nchw(im) if layout == "nchw" else im).tofile(tmp_path / ("%03d.images" % sid))
        lab.tofile(tmp_path / ("%03d.labels" % sid))
        shards.append((im, lab))
    return shards


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_prefetch_under_full_steps_equals_the_blocking_loader(tmp_path, layout):
    """6 batches per shard, 2 shards, every step a full forward / backward / update with input_reset on (update_parameters clears
    the batch buffers, resnet.cu:2981-2982, without synchronising the host): a prefetch hit is followed by the next enqueue into
    the buffer the step before was still using.  Inputs and labels of every step, and every loss, equal the blocking loader's."""
    from resnet_amd import Trainer
    from resnet_amd import binding as B
    batch, per_shard = 4, 24
    shards = _write_shards(tmp_path, 2, per_shard, layout)
    runs = []
    for prefetch in (False, True):
        tr = Trainer(synth.C1_DIMS, batch, seed=1236, shard_n_images=per_shard)
        tr.source_shards(str(tmp_path), B.MI_LAYOUT_NCHW if layout == "nchw" else B.MI_LAYOUT_NHWC, prefetch=prefetch)
        losses = []
        try:
            for step in range(12):
                tr.load_new_batch()
                assert tr.L.mi_batch_last_status(tr.c_batch) == 0
                sid, b = divmod(step, per_shard // batch)
                im, lab = shards[sid]
                tr.forward()
                losses.append(tr.loss()[0])
                # read the batch AFTER the forward pass was queued: what the kernels saw
                assert np.array_equal(tr.activation("input"), nchw(im[b * batch:(b + 1) * batch])), "step %d prefetch %d" % (step, prefetch)
                assert np.array_equal(tr.labels(), lab[b * batch:(b + 1) * batch])
                dev_lab = tr._to_host(tr.c_batch.contents.correct_classes, batch, np.int32)
                assert np.array_equal(dev_lab, lab[b * batch:(b + 1) * batch]), "device labels step %d" % step
                tr.backward(); tr.update()
            tr.check()
        finally:
            tr.close()
        runs.append(losses)
    assert runs[0] == runs[1] and all(np.isfinite(runs[0]))


@pytest.mark.parametrize("prefetch", [False, True])
def test_rank_slices_of_a_shard(tmp_path, prefetch):
    """mi_batch_set_rank_slice: global batch g of a shard = images [g*W*N, (g+1)*W*N), rank r reads [.. + r*N, .. + (r+1)*N); every
    rank rolls to the next shard at the same step, a ragged tail (here 4 of 20 images) is skipped like
    resnet_cudnn_lowmem.cu:1293-1297 skips it.  Two trainers = ranks 0 and 1 of a world of 2 over the same two shards."""
    from resnet_amd import Trainer
    from resnet_amd import binding as B
    N, W, per_shard = 4, 2, 20
    shards = _write_shards(tmp_path, 2, per_shard)
    trs = []
    try:
        for r in range(W):
            tr = Trainer(synth.C1_DIMS, N, seed=1236, shard_n_images=per_shard)
            tr.source_shards(str(tmp_path), B.MI_LAYOUT_NCHW, prefetch=prefetch)
            tr.L.mi_batch_set_rank_slice(tr.c_batch, r, W)
            trs.append(tr)
        per = per_shard // (W * N)  # whole global batches per shard: 2
        for step in range(2 * per):
            sid, g = divmod(step, per)
            im, lab = shards[sid]
            for r, tr in enumerate(trs):
                tr.load_new_batch()
                assert tr.L.mi_batch_last_status(tr.c_batch) == 0
                lo = g * W * N + r * N
                assert np.array_equal(tr.activation("input"), nchw(im[lo:lo + N])), "rank %d step %d" % (r, step)
                assert np.array_equal(tr.labels(), lab[lo:lo + N])
                assert tr.c_batch.contents.cur_shard_id == sid
                tr.forward(); tr.backward(); tr.update()
        for tr in trs:
            tr.load_new_batch()  # shard 002 does not exist: both ranks find out at the same step
            assert tr.L.mi_batch_last_status(tr.c_batch) == -1
    finally:
        for tr in trs:
            tr.close()
