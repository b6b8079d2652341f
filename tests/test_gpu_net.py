"""Whole-step parity through the drop-in trainer surface (init_* / load_new_batch / forward_pass /
backwards_pass / update_parameters) against the CPU oracle on the same seeded synthetic batch:
per-layer activations, loss, every gradient tensor, and parameters + Adam moments after the update.
Config 1 (BASELINE.json configs[0]) and a 3-block net with a striding block (3x3-s2 projection,
identity shortcut, toAdd)."""
import numpy as np
import pytest

import synth
from util import ACT_MAX_ABS, ACT_REL_L2, GRAD_REL_L2, LOSS_ABS, check_act, check_grad, nhwc, rel_l2

pytestmark = pytest.mark.gpu

# Adam itself is pinned in tests/test_gpu_state.py::test_adam_pinned_with_the_oracles_gradients (the oracle's gradients injected:
# parameters / moments at 1e-6).  End to end the first Adam steps are ~lr * sign(g), so an element whose gradient is at rounding
# level (|g| <~ eps = 1e-7 ... 1e-6) turns a 1e-7 gradient difference into an lr-sized parameter difference: gradient-sign noise,
# not Adam.  The whole-step check therefore compares the parameters where the gradient is RESOLVED (|g| above 1e-3 of the tensor's
# rms; >= 90 % of every tensor) at 1e-6, and all of them at the loose PARAM_REL_L2.
PARAM_REL_L2 = 2e-5
PARAM_REL_L2_RESOLVED = 1e-6


def check_params_after_update(got, ref, ref_grad, what):
    g = np.abs(np.asarray(ref_grad, np.float64).ravel())
    mask = g > 1e-3 * np.sqrt(np.mean(g * g) + 1e-300)
    assert mask.mean() >= 0.9, "%s: only %.0f %% of the gradient elements are resolved" % (what, 100 * mask.mean())
    r = rel_l2(np.asarray(got).ravel()[mask], np.asarray(ref).ravel()[mask])
    assert r <= PARAM_REL_L2_RESOLVED, "%s (elements with a resolved gradient): rel-L2 %.3e" % (what, r)
    assert rel_l2(got, ref) <= PARAM_REL_L2, "%s: rel-L2 %.3e" % (what, rel_l2(got, ref))
HYPER = dict(lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7)


def _make(dims, batch, oracle, full_store=False, wd=0.0):
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
    assert tr.sizes == net.sizes
    for i, p in enumerate(params):
        net.param(i)[:] = p
    tr.set_params(params)
    tr.source_host(B.MI_LAYOUT_NHWC)
    if full_store:
        tr.L.mi_trainer_set_full_store(tr.t, 1)
    return net, tr


def _step(net, tr, dims, batch, step):
    im, lab = synth.make_batch(dims, batch, step=step)
    net.set_batch(im, lab)
    tr.fill_host_batch(im, lab)
    tr.load_new_batch()
    net.forward()
    tr.forward()
    tr.check()


def _relu_gate_flips(tr, ref, dims):
    """ReLU gates on which the HIP forward and a reference forward disagree, and the largest magnitude involved.  A
    pre-activation within fp32 rounding of 0 (seen: 1.4e-6 where activations are O(1)) takes its gate from the summation
    order; one flipped gate moves every gradient upstream of it by ~1e-3..1e-2 (BN backward spreads it over the whole
    channel), so gradients can only be compared tightly against a reference with the SAME gates."""
    names = ["init_conv_activated"] + ["conv_blocks/%02d/%s" % (b, l) for b in range(dims["n_conv_blocks"])
                                       for l in ("reduction_activated", "spatial_activated", "output_activated")]
    flips, mag = 0, 0.0
    for nm in names:
        g, r = nhwc(tr.activation(nm)), ref.tensor(nm)
        d = (g > 0) != (r > 0)
        if d.any():
            flips += int(d.sum())
            mag = max(mag, float(np.abs(g[d]).max()), float(np.abs(r[d]).max()))
    return flips, mag


FWD_NAMES = ["init_conv_applied", "init_conv_activated", "init_convblock_input"]
BLOCK_FWD = ["reduction_applied", "reduction_activated", "spatial_applied", "spatial_activated", "expanded_applied",
             "output_activated"]


@pytest.mark.parametrize("cfg", ["C1", "C1S", "C1S_batch5", "C1_in48"])
def test_training_step_parity(oracle, oracle64, cfg):
    # batch 5: column counts of every layer are odd multiples (ragged last tiles, images straddling tiles) in a whole step
    # in48: a 48x48 input (24x24 stem output: not a multiple of 16 pixels per row) keeps the stem on the VALU kernels, and gives
    # 12x12 planes (144 pixels) to the block
    dims, batch = (synth.C1_DIMS, synth.C1_BATCH) if cfg == "C1" else (synth.resnet_dims(input=48, n_conv_blocks=1, reductions=(), final_depth=256), 3) \
        if cfg == "C1_in48" else (synth.C1S_DIMS, 5 if cfg.endswith("5") else 4)
    net, tr = _make(dims, batch, oracle)
    from oracle.oracle_py import OracleNet
    ref64 = OracleNet(oracle64, dims, batch)  # double-accumulation twin: arbiter for ReLU gates that sit on a rounding error
    ref64.set_hyper(HYPER["lr"], HYPER["wd"], HYPER["b1"], HYPER["b2"], HYPER["eps"])
    for i in range(net.n_locations):
        ref64.param(i)[:] = net.param(i)
    try:
        for step in range(2):
            _step(net, tr, dims, batch, step)
            im64, lab64 = synth.make_batch(dims, batch, step=step)
            ref64.set_batch(im64, lab64)
            ref64.forward()
            # ---- per-layer activations (the second step compounds the first update's ~1e-6 parameter
            # differences through 10 BN layers, hence the wider band there) ----
            rel = ACT_REL_L2 if step == 0 else 5 * ACT_REL_L2
            for nm in FWD_NAMES:
                check_act(nhwc(tr.activation(nm)), net.tensor(nm), "%s step %d" % (nm, step), rel=rel)
            for b in range(dims["n_conv_blocks"]):
                for leaf in BLOCK_FWD:
                    nm = "conv_blocks/%02d/%s" % (b, leaf)
                    check_act(nhwc(tr.activation(nm)), net.tensor(nm), "%s step %d" % (nm, step), rel=rel)
            check_act(tr.activation("final_avg_pool"), net.tensor("final_avg_pool").reshape(batch, -1), "avg pool")
            check_act(tr.activation("fc_output"), net.tensor("fc_output").reshape(batch, -1), "logits")
            check_act(tr.pred(), net.tensor("softmax").reshape(batch, -1), "softmax")
            # ---- loss (host loop of resnet.cu:3363-3383 on pred_cpu) ----
            (gl, gw), (ol, ow) = tr.loss(), net.loss()
            assert abs(gl - ol) <= LOSS_ABS * max(1.0, abs(ol)), (gl, ol)
            assert gw == ow
            # ---- gradients ----
            net.backward()
            tr.backward()
            tr.check()
            ref64.backward()
            # A pre-activation within rounding of 0 gets its ReLU gate from the summation order: the sequential-fp32 oracle and
            # its double-accumulation twin then disagree with each other by ~1e-2 on EVERY gradient (C1S, batch 5, second
            # batch).  For such a step the arbiter is the double-accumulation oracle (closer to exact arithmetic).
            tol = GRAD_REL_L2 if step == 0 else 3 * GRAD_REL_L2
            disputed = any(rel_l2(net.grad(i), ref64.grad(i)) > tol for i in range(net.n_locations))
            ref = ref64 if disputed else net
            flips, mag = _relu_gate_flips(tr, ref, dims)
            if flips:  # the HIP forward shares neither oracle's gates on a rounding-level element (see _relu_gate_flips)
                assert flips <= 4 and mag <= 1e-5, "ReLU gates differ on %d elements up to magnitude %.2e" % (flips, mag)
            for i in range(net.n_locations):
                check_grad(tr.get("grads", i), ref.grad(i), "gradient of location %d step %d%s" % (i, step, " (f64 oracle)" if disputed else ""),
                           rel=tol if flips == 0 else 3e-2)
            # ---- Adam ----
            ref_grads = [ref.grad(i).copy() for i in range(net.n_locations)]
            net.update()
            ref64.update()
            tr.update()
            tr.check()
            for i in range(net.n_locations):
                if flips == 0:  # (with a flipped gate the gradients fed to Adam differ: the update is pinned by the other steps)
                    # Adam's first steps are ~lr*sign(g): gradient elements near 0 amplify rounding differences
                    check_params_after_update(tr.get("params", i), ref.param(i), ref_grads[i], "param %d step %d" % (i, step))
                    check_grad(tr.get("means", i), ref.mean(i), "adam mean %d" % i)
                    check_grad(tr.get("vars", i), ref.var(i), "adam var %d" % i, rel=2 * GRAD_REL_L2)
                assert not np.any(tr.get("grads", i)), "gradients are zeroed after the update (resnet.cu:2972-2978)"
            assert not np.any(tr.activation("input")), "batch buffers are zeroed after the update (resnet.cu:2981)"
            # The second step starts from ONE state everywhere (the fp32 oracle's): the update itself is pinned above, and
            # Adam's first update (~lr * sign(g)) turns 1e-7 gradient differences into 1e-5 parameter differences that
            # flipped ReLU / max-pool decisions can blow up downstream -- a property of comparing fp32 trajectories, not of
            # a kernel.  Step 1 so tests every kernel with a second batch, nonzero moments and advanced decays.
            for i in range(net.n_locations):
                tr.set("params", i, net.param(i)); tr.set("means", i, net.mean(i)); tr.set("vars", i, net.var(i))
                ref64.param(i)[:] = net.param(i); ref64.mean(i)[:] = net.mean(i); ref64.var(i)[:] = net.var(i)
    finally:
        tr.close()
        net.close()
        ref64.close()


def test_full_store_matches_fast_path(oracle):
    """full-store mode (x-hat, BN-out and pre-ReLU sums kept, unfused add) gives the same step"""
    dims, batch = synth.C1S_DIMS, 4
    net, tr = _make(dims, batch, oracle, full_store=True, wd=1e-3)
    try:
        _step(net, tr, dims, batch, 0)
        for b in range(dims["n_conv_blocks"]):
            for leaf in ("expanded_post_norm", "combined_output", "output_activated"):
                nm = "conv_blocks/%02d/%s" % (b, leaf)
                check_act(nhwc(tr.activation(nm)), net.tensor(nm), nm)
        check_act(tr.activation("batch_norms/01/projected/means"), net.tensor("batch_norms/01/projected/means"), "proj means")
        net.backward(); tr.backward()
        for i in range(net.n_locations):
            check_grad(tr.get("grads", i), net.grad(i), "gradient %d" % i)
        net.update(); tr.update()
        for i in range(net.n_locations):
            assert rel_l2(tr.get("params", i), net.param(i)) <= PARAM_REL_L2
    finally:
        tr.close()
        net.close()


def test_synthetic_source_and_reproducibility():
    """the HBM-resident synthetic pool equals the numpy stream, and two runs give identical losses"""
    from resnet_amd import Trainer
    dims, batch = synth.C1_DIMS, 4
    losses = []
    for _ in range(2):
        tr = Trainer(dims, batch, seed=1236)
        tr.source_synthetic(1234, 1235, pool_batches=2)
        tr.load_new_batch()
        n = batch * 3 * 32 * 32
        assert np.array_equal(tr.activation("input").ravel(), synth.uniform(1234, n, -124.0, 152.0))
        assert np.array_equal(tr.labels(), synth.labels(1235, batch, 1000))
        run = []
        for _s in range(3):
            tr.forward(); run.append(tr.loss()[0]); tr.backward(); tr.update(); tr.load_new_batch()
        tr.check()
        losses.append(run)
        tr.close()
    assert losses[0] == losses[1], "bitwise reproducible (no atomics, fixed reduction order)"
    assert all(np.isfinite(losses[0]))


@pytest.mark.parametrize("cfg", ["C1S", "R50x8"])
def test_wgrad_scheduling_modes_bit_identical(cfg):
    """mi_trainer_set_overlap: serial (0), wgrad next to the following BN' (1) and free-running wgrads over the ring of
    derivative buffers (2) are schedules of the same kernels on the same data -- gradients, parameters and losses after
    three steps must be bit-identical.  A missing ring wait (a derivative buffer rewritten under a running weight
    gradient) shows up here as a difference."""
    from resnet_amd import Trainer
    dims, batch = (synth.C1S_DIMS, 4) if cfg == "C1S" else (synth.R50_DIMS, 8)
    outs = []
    for mode in (0, 1, 2, 21):
        tr = Trainer(dims, batch, seed=1236)
        tr.source_synthetic(1234, 1235, pool_batches=2)
        tr.L.mi_trainer_set_overlap(tr.t, min(mode, 2))
        losses = []
        tr.load_new_batch(); tr.forward(); losses.append(tr.loss()[0]); tr.backward()
        g0 = [tr.get("grads", i).copy() for i in range(len(tr.sizes))]
        tr.update()
        if mode == 21:  # switch 2 -> 1 between steps: the fixed buffer aliasing is restored
            tr.L.mi_trainer_set_overlap(tr.t, 1)
        for _ in range(2):
            losses.append(tr.step()[0])
        tr.check()
        outs.append((losses, [tr.get("params", i).copy() for i in range(len(tr.sizes))],
                     [tr.get("means", i).copy() for i in range(len(tr.sizes))], g0))
        tr.close()
    for o in outs[1:]:
        assert o[0] == outs[0][0]
        for k in (1, 2, 3):
            for a, b in zip(o[k], outs[0][k]):
                assert np.array_equal(a, b)


def test_fused_bn_statistics_match_separate_pass(tmp_path):
    """RESNET_MI_BNFUSE (default 1): the forward implicit-GEMM convolutions leave per-tile (count, mean, M2) partials of
    their output and batch norm merges those instead of reading the tensor a third time.  A child process per setting runs
    forward + backward of the same batch through the 3-block net with a striding block (every conv + BN unit but the stem
    takes the fused route); loss and every gradient tensor must agree to fp32 reduction-order noise.  (The 16-block
    network is not used here: at small batch ANY change of summation order -- e.g. RESNET_MI_IGEMM_TAIL=0 -- moves its
    gradients by ~2 % rel-L2; its yardstick is test_reference_resnet50_step_parity, which runs with the fused statistics.)"""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import numpy as np, synth; from resnet_amd import Trainer; "
            "tr = Trainer(synth.C1S_DIMS, 8, seed=1236); tr.source_synthetic(1234, 1235, pool_batches=2); "
            "tr.load_new_batch(); tr.forward(); l = tr.loss()[0]; tr.backward(); tr.check(); "
            "np.savez(sys.argv[1], loss=np.float64(l), **{'g%%03d' %% i: tr.get('grads', i) for i in range(len(tr.sizes))}); tr.close()")
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for fuse in ("1", "0"):
        f = str(tmp_path / ("g%s.npz" % fuse))
        r = subprocess.run([sys.executable, "-c", code % (os.path.dirname(here), here), f], env=dict(os.environ, RESNET_MI_BNFUSE=fuse),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
    a, b = outs
    assert np.isfinite(a["loss"]) and abs(a["loss"] - b["loss"]) <= 1e-5 * abs(b["loss"])
    for k in b.files:
        if k != "loss":
            assert rel_l2(a[k], b[k]) <= 5 * GRAD_REL_L2, k


def test_weight_init_matches_stream():
    from resnet_amd import Trainer
    dims = synth.C1_DIMS
    tr = Trainer(dims, 4, seed=1236)
    ref = synth.make_params(dims, seed=1236)
    for i, r in enumerate(ref):
        got = tr.get("params", i)
        assert np.allclose(got, r, rtol=1e-6, atol=1e-9), "location %d" % i
    tr.close()


def test_data_parallel_path_single_rank(oracle):
    """the RCCL bucket / second-stream / event path with a one-rank communicator: all-reduce(SUM) over one rank
    is the identity, so the step must still match the oracle -- exercises dlopen(librccl), ncclCommInitRank,
    bucketed ncclAllReduce on the comm stream and the Adam wait on this box's single GPU"""
    import ctypes as C
    dims, batch = synth.C1S_DIMS, 4
    net, tr = _make(dims, batch, oracle)
    try:
        nbytes = tr.L.mi_dp_unique_id_bytes()
        uid = (C.c_char * nbytes)()
        assert tr.L.mi_dp_get_unique_id(uid, nbytes) == 0, tr.error()
        assert tr.L.mi_dp_init(tr.t, 0, 1, uid, nbytes) == 0, tr.error()
        tr.L.mi_dp_set_bucket_bytes(tr.t, 64 << 10)  # small buckets: several all-reduces per backward
        for step in range(2):
            _step(net, tr, dims, batch, step)
            net.backward(); tr.backward(); tr.check()
            net.update(); tr.update(); tr.check()
            for i in range(net.n_locations):
                assert rel_l2(tr.get("params", i), net.param(i)) <= PARAM_REL_L2, "param %d step %d" % (i, step)
    finally:
        tr.close()
        net.close()


def test_reference_resnet50_step_parity(oracle, oracle64):
    """the benchmark architecture itself (16 bottleneck blocks, 160 tensors, 47.58 M parameters, every layer shape incl. the
    three 3x3-s2 projections) at batch 2.  53 BN layers over as few as 98 samples per channel amplify rounding, so the
    yardstick is the oracle's own fp32 error: the HIP path must be as close to the double-accumulation oracle as the
    sequential-fp32 oracle is (factor 3), for the last activation, the loss and every gradient tensor."""
    from oracle.oracle_py import OracleNet
    dims, batch = synth.R50_DIMS, 2
    net, tr = _make(dims, batch, oracle)
    ref = OracleNet(oracle64, dims, batch)
    try:
        assert net.n_locations == 160 and sum(net.sizes) == 47576128
        for i in range(net.n_locations):
            ref.param(i)[:] = net.param(i)
        im, lab = synth.make_batch(dims, batch, step=0)
        ref.set_batch(im, lab)
        _step(net, tr, dims, batch, 0)
        ref.forward()
        name = "conv_blocks/15/output_activated"
        e_gpu, e_f32 = rel_l2(nhwc(tr.activation(name)), ref.tensor(name)), rel_l2(net.tensor(name), ref.tensor(name))
        assert e_gpu <= 3 * e_f32 + 1e-6, (e_gpu, e_f32)
        (gl, gw), (ol, ow), (rl, rw) = tr.loss(), net.loss(), ref.loss()
        assert abs(gl - rl) <= 3 * abs(ol - rl) + LOSS_ABS * max(1.0, abs(rl)), (gl, ol, rl)
        net.backward(); tr.backward(); tr.check(); ref.backward()
        worst_gpu = worst_f32 = 0.0
        for i in range(net.n_locations):
            g_gpu, g_f32 = rel_l2(tr.get("grads", i), ref.grad(i)), rel_l2(net.grad(i), ref.grad(i))
            worst_gpu, worst_f32 = max(worst_gpu, g_gpu), max(worst_f32, g_f32)
            assert g_gpu <= 3 * g_f32 + GRAD_REL_L2, "gradient of location %d: HIP %.3e, fp32 oracle %.3e vs f64 oracle" % (i, g_gpu, g_f32)
        print("ResNet-50 batch 2 vs f64 oracle: activation err HIP %.2e / fp32-oracle %.2e; loss %.6f / %.6f / %.6f; "
              "worst gradient err HIP %.2e / fp32-oracle %.2e" % (e_gpu, e_f32, gl, ol, rl, worst_gpu, worst_f32))
    finally:
        tr.close()
        net.close()
        ref.close()


OTHER_DIMS = {
    # reductions in other places, a striding FIRST block (3x3-s2 projection from the 64-channel pool output), two striding
    # blocks in a row, 64x64 input: shapes the benchmark network does not contain
    "stride_first": (synth.resnet_dims(input=32, n_conv_blocks=2, reductions=(0,), final_depth=512), 3),
    "two_strides": (synth.resnet_dims(input=64, n_conv_blocks=4, reductions=(1, 2), final_depth=1024), 2),
    "six_blocks": (synth.resnet_dims(input=32, n_conv_blocks=6, reductions=(2, 4), final_depth=1024), 3),
}


@pytest.mark.parametrize("name", sorted(OTHER_DIMS))
def test_other_reference_defined_nets(oracle, oracle64, name):
    """`Dims` can express any bottleneck net (init_dimensions, resnet.cu:666): three more of them, one forward + backward
    against the oracle -- loss, last activation, every gradient (the f64 oracle arbitrates a step whose fp32 oracle sits on
    a flipped ReLU gate, as in test_training_step_parity)."""
    from oracle.oracle_py import OracleNet
    dims, batch = OTHER_DIMS[name]
    net, tr = _make(dims, batch, oracle)
    ref64 = OracleNet(oracle64, dims, batch)
    try:
        for i in range(net.n_locations):
            ref64.param(i)[:] = net.param(i)
        im, lab = synth.make_batch(dims, batch, step=0)
        ref64.set_batch(im, lab)
        _step(net, tr, dims, batch, 0)
        ref64.forward()
        last = "conv_blocks/%02d/output_activated" % (dims["n_conv_blocks"] - 1)
        check_act(nhwc(tr.activation(last)), net.tensor(last), last)
        (gl, gw), (ol, ow) = tr.loss(), net.loss()
        assert abs(gl - ol) <= LOSS_ABS * max(1.0, abs(ol)) and gw == ow
        net.backward(); tr.backward(); tr.check(); ref64.backward()
        # compare against the oracle whose ReLU gates the HIP forward shares; if it shares neither's (a gate sitting on a
        # rounding error: its magnitude must be at rounding level), the gradients can only agree loosely
        f32_flips, f32_mag = _relu_gate_flips(tr, net, dims)
        f64_flips, f64_mag = _relu_gate_flips(tr, ref64, dims)
        ref, flips, mag = (net, f32_flips, f32_mag) if f32_flips <= f64_flips else (ref64, f64_flips, f64_mag)
        if flips:
            assert flips <= 4 and mag <= 1e-5, "ReLU gates differ on %d elements up to magnitude %.2e" % (flips, mag)
        for i in range(net.n_locations):
            check_grad(tr.get("grads", i), ref.grad(i), "%s: gradient of location %d (%d flipped gates)" % (name, i, flips),
                       rel=GRAD_REL_L2 if flips == 0 else 3e-2)
    finally:
        tr.close()
        net.close()
        ref64.close()
