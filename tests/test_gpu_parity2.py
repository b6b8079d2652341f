"""Round-2 hardening of whole-network parity (VERDICT r1, items 3 and 5 of "missing"):
* the dump tree of dump_trainer (resnet.cu:2321-2680) against the oracle's tensor table, file by file: activations/**,
  activation_derivs/** (FULL store policy: every derivative tensor has a buffer of its own), gradients/%03d, batch-norm
  statistics, arg-max indices, labels;
* the benchmark architecture itself (16 blocks) at batch 8 against BOTH oracles, every block's output and every BN's
  mean / variance, with absolute tolerances (DESIGN.md section 2 tabulates what is measured);
* shards built by the product's shard writer feed a training step; the example driver's per-epoch bookkeeping."""
import os
import re
import subprocess

import numpy as np
import pytest

import synth
from util import nchw, nhwc, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HYPER = dict(lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7)


def _pair(dims, batch, oracle, full=False):
    from oracle.oracle_py import OracleNet
    from resnet_amd import Trainer
    from resnet_amd import binding as B
    params = synth.make_params(dims, perturb_bn=True)
    net = OracleNet(oracle, dims, batch)
    net.set_hyper(HYPER["lr"], HYPER["wd"], HYPER["b1"], HYPER["b2"], HYPER["eps"])
    tr = Trainer(dims, batch, dump_dir="run", **HYPER)
    if tr.L.mi_device_count() < 1:
        pytest.fail("no HIP device: this test must run on the MI355X box")
    if full:
        tr.set_store_policy(B.MI_STORE_FULL)
    for i, p in enumerate(params):
        net.param(i)[:] = p
    tr.set_params(params)
    tr.source_host(B.MI_LAYOUT_NHWC)
    return net, tr


@pytest.mark.parametrize("cfg", ["C1", "C1S"])
def test_dump_tree_equals_the_oracle_tensor_table(oracle, tmp_path, cfg):
    dims, batch = (synth.C1_DIMS, 4) if cfg == "C1" else (synth.C1S_DIMS, 4)
    net, tr = _pair(dims, batch, oracle, full=True)
    try:
        tr.L.mi_trainer_set_dump_root(tr.t, str(tmp_path).encode())
        im, lab = synth.make_batch(dims, batch, step=0)
        net.set_batch(im, lab); net.forward(); net.backward()
        tr.fill_host_batch(im, lab); tr.load_new_batch(); tr.forward(); tr.backward(); tr.check()
        tr.L.dump_trainer(7, tr.t, b"run")
        d = os.path.join(str(tmp_path), "run", "%08d" % 7)
        seen = {"activations": 0, "activation_derivs": 0}
        worst = {"activations": 0.0, "activation_derivs": 0.0}
        for tree, prefix in (("activations", ""), ("activation_derivs", "d:")):
            base = os.path.join(d, tree)
            for dirpath, _, files in os.walk(base):
                for f in files:
                    rel = os.path.relpath(os.path.join(dirpath, f), base)[:-len(".buffer")]
                    name = prefix + rel
                    assert name in net.names, "dump file %s/%s has no tensor of that name in the oracle's table" % (tree, rel)
                    ref = net.tensor(name)
                    if rel in ("max_inds", "correct_classes"):
                        got = np.fromfile(os.path.join(dirpath, f), np.int32)
                        if rel == "correct_classes":
                            assert np.array_equal(got, ref.ravel())
                        else:  # flat NCHW index in the file, flat NHWC in the oracle: compare the (ih, iw) they point at
                            N, Hp, _, Cc = ref.shape
                            Hs = dims["input"] // dims["init_conv_stride"]
                            g = got.reshape(N, Cc, Hp, Hp)
                            n_, c_ = np.arange(N)[:, None, None, None], np.arange(Cc)[None, :, None, None]
                            assert np.array_equal(g - (n_ * Cc + c_) * Hs * Hs, (nchw(ref) - n_ * Hs * Hs * Cc - c_) // Cc)
                    else:
                        got = np.fromfile(os.path.join(dirpath, f), np.float32)
                        assert got.size == ref.size, name
                        r = rel_l2(got, ref.ravel())  # image tensors are written NHWC like the reference's, = the oracle's layout
                        worst[tree] = max(worst[tree], r)
                        assert r <= (1e-5 if tree == "activations" else 1e-4), "%s: rel-L2 %.3e" % (name, r)
                    seen[tree] += 1
        nb = dims["n_conv_blocks"]
        # every tensor the reference's dump names (Appendix B of SURVEY.md) and the product keeps
        assert seen["activations"] >= 9 + 2 + nb * (9 + 6) and seen["activation_derivs"] >= 5 + nb * 7, seen
        for i in range(net.n_locations):
            g = np.fromfile(os.path.join(d, "gradients", "%03d.buffer" % i), np.float32)
            assert rel_l2(g, net.grad(i)) <= 1e-4, "gradients/%03d" % i
            p = np.fromfile(os.path.join(d, "model_params", "%03d.buffer" % i), np.float32)
            assert np.array_equal(p, net.param(i).ravel())
        print("dump tree %s: %d + %d files, worst rel-L2 %.2e (activations) %.2e (derivatives)"
              % (cfg, seen["activations"], seen["activation_derivs"], worst["activations"], worst["activation_derivs"]))
    finally:
        tr.close()
        net.close()


# Absolute fp32 tolerances for the 16-block benchmark network at batch 8 (tabulated with the measured values in DESIGN.md
# section 2): block outputs and BN statistics against the sequential-fp32 oracle AND against its double-accumulation twin.
R50_ACT_REL_F32 = 1e-3      # vs the sequential-fp32 oracle, whose own distance to its f64 twin is 4.9e-4 at block 15 (measured)
R50_ACT_REL_F64 = 1.5e-4    # vs the double-accumulation oracle (measured 7.7e-5 at block 15, 2.7e-6 at block 3)
R50_STAT_REL = 2e-4
R50_LOSS_ABS = 2e-3
R50_GRAD_REL = 1e-3         # gradients, when every ReLU gate agrees with the f64 oracle's
R50_GRAD_REL_FLIPS = 5e-2   # when some gates sit on a rounding error (|pre-activation| <= 1e-3 checked, activations O(1)): each flipped gate moves
                            # every gradient upstream of it


def test_reference_resnet50_batch8_every_block(oracle, oracle64):
    from oracle.oracle_py import OracleNet
    dims, batch = synth.R50_DIMS, 8
    net, tr = _pair(dims, batch, oracle)
    ref = OracleNet(oracle64, dims, batch)
    try:
        for i in range(net.n_locations):
            ref.param(i)[:] = net.param(i)
        im, lab = synth.make_batch(dims, batch, step=0)
        for n_ in (net, ref):
            n_.set_batch(im, lab); n_.forward()
        tr.fill_host_batch(im, lab); tr.load_new_batch(); tr.forward(); tr.check()
        worst32 = worst64 = wstat = 0.0
        rows = []
        for b in range(dims["n_conv_blocks"]):
            nm = "conv_blocks/%02d/output_activated" % b
            g = nhwc(tr.activation(nm))
            e32, e64, o = rel_l2(g, net.tensor(nm)), rel_l2(g, ref.tensor(nm)), rel_l2(net.tensor(nm), ref.tensor(nm))
            rows.append((b, e32, e64, o))
            worst32, worst64 = max(worst32, e32), max(worst64, e64)
            for bn in ("reduced", "spatial", "expanded") + (("projected",) if b in (0, 3, 7, 13) else ()):
                for leaf in ("means", "vars"):
                    s = "batch_norms/%02d/%s/%s" % (b, bn, leaf)
                    wstat = max(wstat, rel_l2(tr.activation(s), ref.tensor(s)))
        for leaf in ("means", "vars"):
            wstat = max(wstat, rel_l2(tr.activation("batch_norms/init/" + leaf), ref.tensor("batch_norms/init/" + leaf)))
        for b, e32, e64, o in rows:
            print("  block %2d output: HIP vs f32 oracle %.2e, HIP vs f64 oracle %.2e, f32 oracle vs f64 oracle %.2e" % (b, e32, e64, o))
        (gl, gw), (ol, ow), (rl, rw) = tr.loss(), net.loss(), ref.loss()
        print("  loss HIP %.6f  f32 oracle %.6f  f64 oracle %.6f; BN statistics worst rel-L2 vs f64 %.2e" % (gl, ol, rl, wstat))
        assert worst32 <= R50_ACT_REL_F32 and worst64 <= R50_ACT_REL_F64, (worst32, worst64)
        assert wstat <= R50_STAT_REL, wstat
        assert abs(gl - rl) <= R50_LOSS_ABS and gw == rw, (gl, rl)
        net.backward(); ref.backward(); tr.backward(); tr.check()
        # ReLU gates on which the HIP forward and the f64 oracle disagree (pre-activations within fp32 rounding of 0)
        flips, mag = 0, 0.0
        gate_names = ["init_conv_activated"] + ["conv_blocks/%02d/%s" % (b, l) for b in range(dims["n_conv_blocks"])
                                                for l in ("reduction_activated", "spatial_activated", "output_activated")]
        for nm in gate_names:
            g, r = nhwc(tr.activation(nm)), ref.tensor(nm)
            dd = (g > 0) != (r > 0)
            if dd.any():
                flips += int(dd.sum())
                mag = max(mag, float(np.abs(g[dd]).max()), float(np.abs(r[dd]).max()))
        errs = [rel_l2(tr.get("grads", i), ref.grad(i)) for i in range(net.n_locations)]
        oerrs = [rel_l2(net.grad(i), ref.grad(i)) for i in range(net.n_locations)]
        wg, og = max(errs), max(oerrs)
        print("  gradients: worst rel-L2 HIP vs f64 oracle %.2e at location %d, median %.2e (f32 oracle vs f64 oracle: worst %.2e, median %.2e); "
              "%d ReLU gates differ from the f64 oracle's, largest magnitude involved %.1e"
              % (wg, int(np.argmax(errs)), float(np.median(errs)), og, float(np.median(oerrs)), flips, mag))
        assert mag <= 1e-3, "a ReLU gate differs on an element that is not at rounding level: %.2e" % mag
        assert wg <= (R50_GRAD_REL_FLIPS if flips else R50_GRAD_REL), (wg, flips)
    finally:
        tr.close()
        net.close()
        ref.close()


def test_built_shard_trains_a_step_and_epoch_bookkeeping(tmp_path):
    """f2 end to end: uint8 class files + partition CSV -> mi_build_shard -> %03d.images / .labels -> the loader -> one
    training step, compared with the same images handed over directly; then the example driver's epoch loop fills
    loss_per_epoch / accuracy_per_epoch like resnet.cu:3410-3412 (read back from trainer_metadata.txt)"""
    from resnet_amd import Trainer
    from resnet_amd import binding as B
    lib = B.load()
    dim_in, dim_out, n = 40, 32, 8
    rng = np.random.RandomState(5)
    data, part, out = tmp_path / "train_data", tmp_path / "part", tmp_path / "shards"
    for p in (data, part, out):
        p.mkdir()
    for c in range(3):
        rng.randint(0, 256, size=(4, dim_in, dim_in, 3), dtype=np.uint8).tofile(str(data / ("%08d.buffer" % c)))
    rows = [(i % 3, i % 4, (3 * i) % 9, (5 * i) % 9) for i in range(n)]
    (part / "000_images.csv").write_text("".join("%03d,%04d,%02d,%02d\n" % r for r in rows))
    assert lib.mi_build_shard(str(part / "000_images.csv").encode(), str(data).encode(), str(out).encode(), 0, dim_in, dim_out, B.MI_LAYOUT_NCHW) == n
    imgs = np.fromfile(str(out / "000.images"), np.float32).reshape(n, 3, dim_out, dim_out)
    labs = np.fromfile(str(out / "000.labels"), np.int32)
    dims = synth.C1_DIMS
    losses = []
    for src in ("shard", "host"):
        tr = Trainer(dims, 4, shard_n_images=n, **HYPER)
        tr.set_params(synth.make_params(dims, perturb_bn=True))
        if src == "shard":
            tr.source_shards(str(out), B.MI_LAYOUT_NCHW)
        else:
            tr.source_host(B.MI_LAYOUT_NCHW)
        run = []
        for step in range(2):
            if src == "host":
                tr.fill_host_batch(imgs[4 * step:4 * step + 4], labs[4 * step:4 * step + 4])
            run.append(tr.step()[0])
        assert tr.check_errors() == 0
        losses.append(run)
        tr.close()
    assert losses[0] == losses[1] and all(np.isfinite(losses[0]))

    exe = str(tmp_path / "ResNetMI")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "resnet_main.c"),
                           "-L", os.path.join(ROOT, "resnet_amd"), "-lresnet_mi", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "resnet_amd"), "-o", exe])
    log, dump = str(tmp_path / "loss.txt"), str(tmp_path / "dumps")
    outp = subprocess.check_output([exe, "--input", "32", "--blocks", "1", "--batch", "8", "--iters", "3", "--epochs", "2", "--loss-log", log,
                                    "--dump-root", dump], timeout=300).decode()
    assert re.findall(r"Epoch: (\d), Batch: (\d)", outp) == [(str(e), str(i)) for e in range(2) for i in range(3)]
    avg = [float(x) for x in open(log).read().split()]
    meta = open(os.path.join(dump, "my_custom", "%08d" % 77777777, "trainer_metadata.txt")).read().splitlines()
    per_epoch_loss = [float(x) for x in meta[-2].split(",")]
    per_epoch_acc = [float(x) for x in meta[-1].split(",")]
    assert len(per_epoch_loss) == 2 and len(per_epoch_acc) == 2
    for e in range(2):  # loss_per_epoch is the epoch's SUMMED loss (resnet.cu:3410)
        assert abs(per_epoch_loss[e] - 8 * sum(avg[3 * e:3 * e + 3])) <= 1e-2 * per_epoch_loss[e]
        assert 0.0 <= per_epoch_acc[e] <= 1.0
