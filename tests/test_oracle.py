"""CPU tests that pin the oracle (no GPU): the reference's own self-test definitions and tolerances
(resnet.cu:2990-3218), its labels.buffer fixture, an independent torch-CPU autograd model of the
reference-defined network, the double-accumulation build, and the committed golden vectors."""
import os

import numpy as np
import pytest

import synth
import torch_ref
from util import rel_l2

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_labels_buffer_fixture():
    """labels.buffer of the reference: 32 x int32, first 457 (bow tie), last 546 (electric guitar)
    (inspect_input.ipynb cell 8) -- pins the labels file format of resnet.cu:1281-1285"""
    lab = np.fromfile(os.path.join(GOLD, "labels.buffer"), dtype=np.int32)
    assert lab.shape == (32,) and lab[0] == 457 and lab[31] == 546
    assert lab.min() >= 0 and lab.max() < 1000


def test_reference_selftest_matmul_transpose(oracle):
    """testMatMul (resnet.cu:3033-3107): 32x2048 . 2048x1000 vs a host triple loop, abs 1e-5;
    testTranspose (:2990-3030): exact"""
    a = synth.uniform(41, 32 * 2048, -1, 1).reshape(32, 2048)
    b = synth.uniform(42, 2048 * 1000, -1, 1).reshape(2048, 1000)
    got = oracle.matmul(a, b)
    acc = np.zeros((32, 1000), np.float32)  # same z-order sequential fp32 sum, product rounded first
    for z in range(2048):
        acc = acc + a[:, z:z + 1] * b[z:z + 1, :]
    assert np.max(np.abs(got - acc)) <= 1e-5 * 32  # fma vs mul+add over 2048 terms of magnitude <= 1
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    assert np.max(np.abs(got - ref64)) <= 2e-4
    t = oracle.transpose(b)
    assert np.array_equal(t, b.T)


@pytest.mark.parametrize("C,H,K,k,stride,N", [(3, 32, 8, 7, 2, 2), (8, 12, 16, 3, 1, 2), (8, 12, 16, 3, 2, 3), (16, 6, 8, 1, 1, 2)])
def test_reference_selftest_conv(oracle, C, H, K, k, stride, N):
    """testConvolution (resnet.cu:3109-3218, weight index as corrected in resnet_cudnn.cu:3453): 7-deep host
    loop nest, abs 1e-4"""
    x = synth.normal(5, N * H * H * C, 1.0).reshape(N, H, H, C)
    w = synth.normal(6, K * C * k * k, 0.1).reshape(K, C, k, k)
    got = oracle.conv_fwd(x, w, stride)
    Ho, half = H // stride, k // 2
    ref = np.zeros((N, Ho, Ho, K), np.float64)
    xp = np.pad(x.astype(np.float64), ((0, 0), (half, half), (half, half), (0, 0)))
    for r in range(k):
        for s in range(k):
            patch = xp[:, r:r + stride * Ho:stride, s:s + stride * Ho:stride, :]
            ref += np.einsum("nhwc,kc->nhwk", patch, w[:, :, r, s].astype(np.float64))
    assert np.max(np.abs(got - ref)) <= 1e-4


@pytest.mark.parametrize("cfg", ["C1", "C1S"])
def test_oracle_vs_torch_autograd(oracle, oracle64, cfg):
    from oracle.oracle_py import OracleNet
    dims, batch = (synth.C1_DIMS, 4) if cfg == "C1" else (synth.C1S_DIMS, 4)
    params = synth.make_params(dims, perturb_bn=True)
    im, lab = synth.make_batch(dims, batch)
    tn = torch_ref.TorchNet(dims, params)
    tl = float(tn.forward(torch_ref.nhwc_to_nchw(im), lab).detach())
    tg = tn.backward()
    for orc, gtol in ((oracle, 1e-4), (oracle64, 2e-6)):
        net = OracleNet(orc, dims, batch)
        for i, p in enumerate(params):
            net.param(i)[:] = p
        net.set_batch(im, lab)
        net.forward()
        loss, _ = net.loss()
        assert abs(loss - tl) <= 1e-4
        net.backward()
        for i, t in enumerate(tg):
            assert rel_l2(net.grad(i), t) <= gtol, "location %d" % i
        # selected activations
        a = torch_ref.nchw_to_nhwc(tn.acts["b0_out"].detach().numpy())
        assert rel_l2(net.tensor("conv_blocks/00/output_activated"), a) <= 1e-5
        net.close()


def test_softmax_stable_equals_reference_form(oracle):
    x = synth.normal(9, 4 * 1000, 4.0).reshape(4, 1000)
    assert rel_l2(oracle.softmax(x, True), oracle.softmax(x, False)) <= 1e-6  # identical unless exp overflows (h2)
    x[0, 0] = 100.0
    assert np.all(np.isfinite(oracle.softmax(x, True)))
    assert not np.all(np.isfinite(oracle.softmax(x, False)))


def test_adam_matches_textbook(oracle):
    n = 1000
    p, g = synth.normal(1, n, 1.0), synth.normal(2, n, 1.0)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    p0 = p.copy()
    oracle.lib.orc_adam(n, p, g, m, v, 1e-3, 0.0, 0.9, 0.999, 0.9, 0.999, 1e-7)
    assert np.allclose(m, 0.1 * g, rtol=1e-5) and np.allclose(v, 0.001 * g * g, rtol=1e-4)
    assert np.allclose(p, p0 - 1e-3 * np.sign(g), atol=1e-6)  # first bias-corrected step is lr*sign(g)


def test_golden_vectors(oracle):
    """committed config-1 fixtures (tests/golden/make_golden.py): the oracle reproduces them"""
    from oracle.oracle_py import OracleNet
    path = os.path.join(GOLD, "c1_golden.npz")
    g = np.load(path)
    dims, batch = synth.C1_DIMS, synth.C1_BATCH
    params = synth.make_params(dims)
    net = OracleNet(oracle, dims, batch)
    for i, p in enumerate(params):
        net.param(i)[:] = p
    im, lab = synth.make_batch(dims, batch)
    assert np.array_equal(im[0, :2, :2].ravel(), g["images_probe"])
    assert np.array_equal(lab, g["labels"])
    net.set_batch(im, lab)
    net.forward()
    loss, wrong = net.loss()
    assert abs(loss - float(g["loss"])) <= 1e-5 and wrong == int(g["n_wrong"])
    assert rel_l2(net.tensor("fc_output").reshape(batch, -1), g["logits"]) <= 1e-6
    net.backward()
    for i in range(net.n_locations):
        assert abs(float(np.abs(net.grad(i)).sum()) - float(g["grad_abs_sum"][i])) <= 1e-4 * float(g["grad_abs_sum"][i]) + 1e-12
    assert rel_l2(net.grad(0), g["grad_init_conv"]) <= 1e-5
    assert rel_l2(net.grad(net.n_locations - 1).reshape(256, 1000)[:, :16], g["grad_fc_probe"]) <= 1e-5
    net.update()
    assert rel_l2(net.param(3), g["param3_after"]) <= 1e-6
    net.close()
