"""Parity of every HIP kernel family against the CPU oracle, called through the C-ABI operator layer
(mi_op_*).  Shapes are the reference-defined ResNet-50's own layer shapes at a
small batch, plus config-1 shapes; the oracle works in NHWC (resnet.cu:145), the product in NCHW, so
tensors are permuted on the way.  Tolerances: util.py (fp32, reduction order differs)."""
import numpy as np
import pytest

from util import check_act, check_grad, nchw, nhwc, rand, rel_l2

pytestmark = pytest.mark.gpu

# (C, H, K, k, stride, N)
CONV_SHAPES = [
    (3, 224, 64, 7, 2, 2),      # stem
    (3, 32, 64, 7, 2, 4),       # stem, config 1
    (64, 56, 64, 3, 1, 2),      # stage 0 spatial
    (128, 56, 128, 3, 2, 2),    # b3 spatial (stride 2)
    (256, 56, 512, 3, 2, 1),    # b3 projection 3x3 s2
    (128, 28, 128, 3, 1, 3),
    (256, 14, 256, 3, 1, 3),
    (512, 14, 512, 3, 2, 2),    # b13 spatial
    (512, 7, 512, 3, 1, 5),
    (1024, 14, 2048, 3, 2, 1),  # b13 projection (18.9M weights)
    (512, 28, 1024, 3, 2, 3),   # b7 projection; 588 columns: ragged last tile of the implicit GEMM
    (64, 8, 64, 3, 1, 4),       # config 1 block
    (128, 8, 128, 3, 2, 4),     # config 1S strided block
    (64, 56, 64, 1, 1, 2),      # 1x1 reduce (MFMA GEMM)
    (64, 56, 256, 1, 1, 2),     # 1x1 expand / projection
    (256, 56, 64, 1, 1, 2),
    (1024, 14, 256, 1, 1, 3),
    (512, 7, 2048, 1, 1, 5),    # P = 49: odd plane
    (2048, 7, 512, 1, 1, 5),
    (64, 8, 256, 1, 1, 4),      # config 1
]
IDS = ["C%d_H%d_K%d_k%d_s%d_N%d" % s for s in CONV_SHAPES]


def _conv_data(C, H, K, k, stride, N, seed=7):
    x = rand((N, H, H, C), seed)
    w = rand((K, C, k, k), seed + 1, scale=(2.0 / (k * k * (C + K))) ** 0.5)
    dy = rand((N, H // stride, H // stride, K), seed + 2)
    return x, w, dy


@pytest.mark.parametrize("shape", CONV_SHAPES, ids=IDS)
def test_conv_fwd(ops, oracle, shape):
    C, H, K, k, stride, N = shape
    x, w, _ = _conv_data(*shape)
    ref = oracle.conv_fwd(x, w, stride)
    got = ops.conv_fwd(nchw(x), w, stride)
    check_act(nhwc(got), ref, "conv_fwd %s" % (shape,))


@pytest.mark.parametrize("shape", [s for s in CONV_SHAPES if s[0] != 3], ids=[i for s, i in zip(CONV_SHAPES, IDS) if s[0] != 3])
def test_conv_dgrad(ops, oracle, shape):
    C, H, K, k, stride, N = shape
    x, w, dy = _conv_data(*shape)
    ref = oracle.conv_dgrad(w, dy, H, stride)
    got = ops.conv_dgrad(w, nchw(dy), H, stride)
    check_grad(nhwc(got), ref, "conv_dgrad %s" % (shape,))
    if k == 1 or (k == 3 and stride == 2 and C >= 256):  # toAdd (residual join, resnet.cu:212-217)
        base = rand((N, H, H, C), 99)
        ref2 = oracle.conv_dgrad(w, dy, H, stride, dx_init=base)
        got2 = ops.conv_dgrad(w, nchw(dy), H, stride, dx_init=nchw(base))
        check_grad(nhwc(got2), ref2, "conv_dgrad+add %s" % (shape,))


@pytest.mark.parametrize("shape", CONV_SHAPES, ids=IDS)
def test_conv_wgrad(ops, oracle, shape):
    C, H, K, k, stride, N = shape
    x, w, dy = _conv_data(*shape)
    ref = oracle.conv_wgrad(x, dy, k, stride)
    got = ops.conv_wgrad(nchw(x), nchw(dy), k, stride)
    check_grad(got, ref, "conv_wgrad %s" % (shape,))


@pytest.mark.parametrize("shape", [(256, 14, 256, 3, 1, 3), (512, 7, 512, 3, 1, 5), (512, 14, 512, 3, 2, 2), (64, 8, 64, 3, 1, 4), (256, 14, 512, 3, 2, 3)])
def test_conv_kernels_do_not_read_unwritten_lds(ops, oracle, shape):
    """rows padded to whole pixel quads multiply stale LDS by dY = 0: with NaNs left in LDS by another kernel
    that must still be finite (regression: full-size step produced NaN gradients)"""
    C, H, K, k, stride, N = shape
    x, w, dy = _conv_data(*shape)
    for fn, ref in ((lambda: ops.conv_wgrad(nchw(x), nchw(dy), k, stride), oracle.conv_wgrad(x, dy, k, stride)),
                    (lambda: nhwc(ops.conv_fwd(nchw(x), w, stride)), oracle.conv_fwd(x, w, stride)),
                    (lambda: nhwc(ops.conv_dgrad(w, nchw(dy), H, stride)), oracle.conv_dgrad(w, dy, H, stride))):
        assert ops.L.mi_debug_poison_lds() == 0
        got = fn()
        assert np.all(np.isfinite(got))
        check_grad(got, ref, "after LDS poison %s" % (shape,))


BN_SHAPES = [(64, 112, 2), (64, 56, 3), (256, 56, 2), (512, 28, 3), (1024, 14, 4), (2048, 7, 6), (512, 7, 5), (64, 8, 4), (256, 8, 4)]


@pytest.mark.parametrize("C,H,N", BN_SHAPES)
@pytest.mark.parametrize("relu", [0, 1])
def test_bn_fwd_bwd(ops, oracle, C, H, N, relu):
    eps = 1e-7
    x = rand((N, H, H, C), 3, 2.0) + 0.5
    gamma = (1 + 0.2 * rand((C,), 4)).astype(np.float32)
    beta = (0.3 * rand((C,), 5)).astype(np.float32)
    dy = rand((N, H, H, C), 6)
    means, vars_, xhat, norm, act = oracle.bn_fwd(x, gamma, beta, eps, relu)
    gm, gv, gy = ops.bn_fwd(nchw(x), gamma, beta, eps, relu)
    check_act(gm, means, "bn means")
    check_act(gv, vars_, "bn vars")
    check_act(nhwc(gy), act, "bn out")
    rdx, rdg, rdb = oracle.bn_bwd(x, gamma, eps, means, vars_, xhat, act, dy, relu)
    # backward is fed the oracle's statistics: with the GPU's own (differently rounded) mean/var a handful of
    # elements with |y| ~ 1e-7 flip their ReLU gate, which is a property of the comparison, not of the kernel
    gdx, gdg, gdb = ops.bn_bwd(nchw(x), gamma, beta, means, vars_, nchw(dy), eps, 1 if relu else 0)
    check_grad(nhwc(gdx), rdx, "bn dx")
    check_grad(gdg, rdg, "bn dgamma")
    check_grad(gdb, rdb, "bn dbeta")


@pytest.mark.parametrize("C,H,N", [(256, 56, 2), (2048, 7, 5), (256, 8, 4)])
def test_bn_add_relu_and_external_mask(ops, oracle, C, H, N):
    """BN(expanded)+addVec+doActivation fused forward, and ReLU' fused into BN' as an external mask"""
    eps = 1e-7
    x = rand((N, H, H, C), 13)
    res = rand((N, H, H, C), 14)
    gamma = (1 + 0.2 * rand((C,), 15)).astype(np.float32)
    beta = (0.3 * rand((C,), 16)).astype(np.float32)
    up = rand((N, H, H, C), 17)
    means, vars_, xhat, norm, act = oracle.bn_fwd(x, gamma, beta, eps, 0)
    summ = act + res
    out = np.maximum(summ, 0)
    gm, gv, gy = ops.bn_fwd(nchw(x), gamma, beta, eps, 0, residual=nchw(res))
    check_act(nhwc(gy), out, "bn+add+relu")
    d_sum = np.where(summ > 0, up, 0).astype(np.float32)  # doActivationDeriv
    rdx, rdg, rdb = oracle.bn_bwd(x, gamma, eps, means, vars_, xhat, act, d_sum, 0)
    gdx, gdg, gdb = ops.bn_bwd(nchw(x), gamma, beta, gm, gv, nchw(up), eps, 2, mask_src=gy)
    check_grad(nhwc(gdx), rdx, "bn dx (external mask)")
    check_grad(gdg, rdg, "bn dgamma (external mask)")
    check_grad(gdb, rdb, "bn dbeta (external mask)")
    got = ops.relu_deriv(gy, nchw(up))
    assert np.array_equal(nhwc(got), d_sum)
    # the variant backwards_pass uses for identity blocks: same gradients, plus the gated upstream gradient itself
    hdx, hdg, hdb, gated = ops.bn_bwd_gate(nchw(x), gamma, beta, gm, gv, nchw(up), eps, gy)
    assert np.array_equal(nhwc(gated), d_sum)
    assert np.array_equal(hdx, gdx) and np.array_equal(hdg, gdg) and np.array_equal(hdb, gdb)


@pytest.mark.parametrize("C,H,N", [(64, 112, 2), (64, 16, 4), (64, 12, 3)])  # 12: rows that are not whole 8-element vectors -> the generic kernel
def test_maxpool(ops, oracle, C, H, N):
    x = np.maximum(rand((N, H, H, C), 21), 0)  # post-ReLU input: ties at 0 exercise "first max wins"
    y, idx = oracle.maxpool_fwd(x, 3, 2)
    gy, gidx = ops.maxpool_fwd(nchw(x), 3, 2)
    assert np.array_equal(nhwc(gy), y)
    # the product stores flat NCHW indices, the oracle flat NHWC: compare the (ih, iw) they point at
    Ho = H // 2
    n_, c_ = np.arange(N)[:, None, None, None], np.arange(C)[None, :, None, None]
    pos_g = gidx - (n_ * C + c_) * H * H
    pos_r = (nchw(idx) - n_ * H * H * C - c_) // C
    assert np.array_equal(pos_g, pos_r)
    dy = rand((N, Ho, Ho, C), 22)
    rdx = oracle.maxpool_bwd(idx, dy, H, 2)
    gdx = ops.maxpool_bwd(gidx, nchw(dy), H, 3, 2)
    assert np.array_equal(nhwc(gdx), rdx)  # overwrite scatter, last writer in scan order


def test_avgpool(ops, oracle):
    N, C, H = 5, 2048, 7
    x = rand((N, H, H, C), 31)
    ref = np.empty((N, C), np.float32)
    oracle.lib.orc_avgpool_fwd(x, H, C, N, ref)
    check_act(ops.avgpool_fwd(nchw(x)), ref, "avgpool")
    dy = rand((N, C), 32)
    rdx = np.empty((N, H, H, C), np.float32)
    oracle.lib.orc_avgpool_bwd(dy, C, N, H, rdx)
    check_act(nhwc(ops.avgpool_bwd(dy, H)), rdx, "avgpool bwd")


def test_matmul_reference_selftest_shapes(ops, oracle):
    """testMatMul / testTranspose of the reference (resnet.cu:2990-3107): 32x2048 . 2048x1000, data in +-1,
    abs tolerance 1e-5 (:3081) -- here scaled by the result magnitude because operands are not the
    reference's rand() stream."""
    import synth
    a = synth.uniform(41, 32 * 2048, -1, 1).reshape(32, 2048)
    b = synth.uniform(42, 2048 * 1000, -1, 1).reshape(2048, 1000)
    ref = oracle.matmul(a, b)
    got = ops.matmul(a, b, "nn")
    assert np.max(np.abs(got - ref)) <= 1e-5 * np.max(np.abs(ref)) * 10
    # FC backward forms (prepareAndDoMatMulLeftTranspose / RightTranspose, resnet.cu:1482-1509)
    d = synth.uniform(43, 32 * 1000, -1, 1).reshape(32, 1000)
    ref_w = oracle.matmul(oracle.transpose(a), d)          # [2048 x 1000]
    check_grad(ops.matmul(a, d, "lt"), ref_w, "fc wgrad")
    ref_x = oracle.matmul(d, oracle.transpose(b))          # [32 x 2048]
    check_grad(ops.matmul(d, b, "rt"), ref_x, "fc dgrad")


def test_softmax_ce_adam(ops, oracle):
    import synth
    x = (synth.normal(51, 8 * 1000, 9.0)).reshape(8, 1000)
    x[3, 17] = 95.0  # would overflow the reference's resnet.cu softmax (hazard h2); stable form required
    ref = oracle.softmax(x)
    got = ops.softmax(x)
    check_act(got, ref, "softmax")
    lab = synth.labels(52, 8, 1000)
    d = ops.ce_deriv(got, lab)
    refd = got.copy()
    oracle.lib.orc_ce_deriv(refd, lab, 1000, 8)
    assert np.array_equal(d, refd)
    n = 10007
    p, g = rand((n,), 61), rand((n,), 62, 0.01)
    m, v = rand((n,), 63, 0.001), np.abs(rand((n,), 64, 1e-4))
    g[5] = np.nan
    g[6] = np.inf
    rp, rm, rv = p.copy(), m.copy(), v.copy()
    oracle.lib.orc_adam(n, rp, g, rm, rv, 1e-4, 1e-3, 0.9, 0.999, 0.9 ** 3, 0.999 ** 3, 1e-7)
    gp, gm, gv, flag = ops.adam(p, g, m, v, 1e-4, 1e-3, 0.9, 0.999, 0.9 ** 3, 0.999 ** 3, 1e-7)
    assert flag == 1  # NaN/Inf gradient seen (check_errors semantics, resnet.cu:2879-2907)
    assert rel_l2(gp, rp) < 1e-6 and rel_l2(gm, rm) < 1e-6 and rel_l2(gv, rv) < 1e-6
    assert gm[5] == m[5] and gv[6] == v[6]  # guards keep the old moments (resnet.cu:610-617)


def test_layout(ops):
    x = rand((3, 10, 10, 7), 71)
    assert np.array_equal(ops.nhwc_to_nchw(x), nchw(x))


@pytest.mark.parametrize("mode", ["0", "1"])
def test_conv_parity_on_the_other_kernel_routes(mode):
    """RESNET_MI_IGEMM selects which kernels a convolution runs on (kernels_igemm.hip: mi_igemm_supported).  The default
    (2) sends every tiling 3x3 and 1x1 convolution to the MFMA implicit GEMM; 1 keeps the bottleneck's own 3x3
    convolutions on the direct VALU kernels and 1x1 forward/dgrad on gemm_mfma_kernel (only the projection shortcuts and
    1x1 wgrad on the implicit GEMM); 0 uses the direct kernels and gemm_mfma_kernel for everything.  The routes are read
    once per process, so the conv parity tests of
    this file are re-run in a child process per route: every kernel family stays pinned to the oracle at the
    ResNet-50 layer shapes whichever route is the default."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RESNET_MI_IGEMM=mode)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_ops.py"), "-x", "-q", "-m", "gpu",
                        "-k", "test_conv_fwd or test_conv_dgrad or test_conv_wgrad or unwritten_lds or config2", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def _basic_block(conv_fwd, conv_dgrad, conv_wgrad, bn_fwd, bn_bwd, x, w1, w2, g1, b1, g2, b2, up, eps=1e-7):
    """ResNet-18 BasicBlock out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x) and its backward, from the operator layer only
    (3x3 conv forward / dgrad / wgrad + BN forward / backward: BASELINE.json configs[1]).  Arrays NHWC; the callables adapt
    layouts.  Returns the tensors a trainer would keep."""
    c1 = conv_fwd(x, w1)
    m1, v1, a1 = bn_fwd(c1, g1, b1, 1, None)
    c2 = conv_fwd(a1, w2)
    m2, v2, out = bn_fwd(c2, g2, b2, 0, x)                   # + residual, ReLU
    d_c2, dg2, db2 = bn_bwd(c2, g2, b2, m2, v2, up, 2, out)  # ReLU' of the block output as an external mask
    dw2 = conv_wgrad(a1, d_c2)
    d_a1 = conv_dgrad(w2, d_c2, None)
    d_c1, dg1, db1 = bn_bwd(c1, g1, b1, m1, v1, d_a1, 1, None)
    dw1 = conv_wgrad(x, d_c1)
    shortcut = np.where(out > 0, up, 0).astype(np.float32)   # identity path: dL/dx += relu'(out) * up
    dx = conv_dgrad(w1, d_c1, shortcut)
    return dict(c1=c1, a1=a1, c2=c2, out=out, dw2=dw2, dg2=dg2, db2=db2, dw1=dw1, dg1=dg1, db1=db1, dx=dx)


@pytest.mark.parametrize("C,H,N", [(64, 56, 8), (128, 28, 8), (256, 14, 16), (512, 7, 16)])
def test_config2_resnet18_basic_block(ops, oracle, C, H, N):
    """BASELINE.json configs[1] ("ResNet-18 ... 3x3 conv + BN kernels only"): the reference can only express bottleneck
    nets, so a BasicBlock is composed from the C-ABI operators at the four ResNet-18 stage shapes and checked
    against the same composition of oracle operators, forward and backward.  On the default route the 3x3 run on the MFMA
    implicit GEMM; test_conv_parity_on_the_other_kernel_routes[0] pins the same operators on the MFMA-free direct kernels."""
    eps = 1e-7
    x = np.maximum(rand((N, H, H, C), 31), 0)  # a block input is post-ReLU
    w1 = rand((C, C, 3, 3), 32, scale=(2.0 / (9 * 2 * C)) ** 0.5)
    w2 = rand((C, C, 3, 3), 33, scale=(2.0 / (9 * 2 * C)) ** 0.5)
    g1, g2 = (1 + 0.2 * rand((C,), 34)).astype(np.float32), (1 + 0.2 * rand((C,), 35)).astype(np.float32)
    b1, b2 = (0.3 * rand((C,), 36)).astype(np.float32), (0.3 * rand((C,), 37)).astype(np.float32)
    up = rand((N, H, H, C), 38)

    def o_bn_fwd(t, g, b, relu, res):
        m, v, xh, norm, act = oracle.bn_fwd(t, g, b, eps, relu)
        return m, v, (np.maximum(act + res, 0) if res is not None else act)

    def o_bn_bwd(t, g, b, m, v, dy, mode, mask):
        _, _, xh, norm, act = oracle.bn_fwd(t, g, b, eps, 1 if mode == 1 else 0)
        d = np.where(mask > 0, dy, 0).astype(np.float32) if mode == 2 else dy
        return oracle.bn_bwd(t, g, eps, m, v, xh, act, d, 1 if mode == 1 else 0)

    ref = _basic_block(lambda a, w: oracle.conv_fwd(a, w, 1), lambda w, d, add: oracle.conv_dgrad(w, d, H, 1, dx_init=add),
                       lambda a, d: oracle.conv_wgrad(a, d, 3, 1), o_bn_fwd, o_bn_bwd, x, w1, w2, g1, b1, g2, b2, up)

    def g_bn_fwd(t, g, b, relu, res):
        m, v, y = ops.bn_fwd(nchw(t), g, b, eps, relu, residual=None if res is None else nchw(res))
        return m, v, nhwc(y)

    def g_bn_bwd(t, g, b, m, v, dy, mode, mask):
        dx, dg, db = ops.bn_bwd(nchw(t), g, b, m, v, nchw(dy), eps, mode, mask_src=None if mask is None else nchw(mask))
        return nhwc(dx), dg, db

    # each GPU operator is fed the ORACLE's upstream tensors (op-by-op pinning: a ReLU gate that flips on a 1e-7 difference
    # upstream would otherwise be charged to every operator downstream)
    got = {}
    got["c1"] = nhwc(ops.conv_fwd(nchw(x), w1, 1))
    om1, ov1, _ = o_bn_fwd(ref["c1"], g1, b1, 1, None)
    _, _, got["a1"] = g_bn_fwd(ref["c1"], g1, b1, 1, None)
    got["c2"] = nhwc(ops.conv_fwd(nchw(ref["a1"]), w2, 1))
    om2, ov2, _ = o_bn_fwd(ref["c2"], g2, b2, 0, x)
    _, _, got["out"] = g_bn_fwd(ref["c2"], g2, b2, 0, x)
    for k in ("c1", "a1", "c2", "out"):
        check_act(got[k], ref[k], "BasicBlock %s" % k)
    d_c2_ref, _, _ = o_bn_bwd(ref["c2"], g2, b2, om2, ov2, up, 2, ref["out"])
    d_c2, dg2, db2 = g_bn_bwd(ref["c2"], g2, b2, om2, ov2, up, 2, ref["out"])
    check_grad(d_c2, d_c2_ref, "BasicBlock d_c2"); check_grad(dg2, ref["dg2"], "dgamma2"); check_grad(db2, ref["db2"], "dbeta2")
    check_grad(ops.conv_wgrad(nchw(ref["a1"]), nchw(d_c2_ref), 3, 1), ref["dw2"], "BasicBlock dw2")
    d_a1_ref = oracle.conv_dgrad(w2, d_c2_ref, H, 1)
    check_grad(nhwc(ops.conv_dgrad(w2, nchw(d_c2_ref), H, 1)), d_a1_ref, "BasicBlock d_a1")
    d_c1_ref, _, _ = o_bn_bwd(ref["c1"], g1, b1, om1, ov1, d_a1_ref, 1, None)
    d_c1, dg1, db1 = g_bn_bwd(ref["c1"], g1, b1, om1, ov1, d_a1_ref, 1, None)
    check_grad(d_c1, d_c1_ref, "BasicBlock d_c1"); check_grad(dg1, ref["dg1"], "dgamma1"); check_grad(db1, ref["db1"], "dbeta1")
    check_grad(ops.conv_wgrad(nchw(x), nchw(d_c1_ref), 3, 1), ref["dw1"], "BasicBlock dw1")
    shortcut = np.where(ref["out"] > 0, up, 0).astype(np.float32)
    check_grad(nhwc(ops.conv_dgrad(w1, nchw(d_c1_ref), H, 1, dx_init=nchw(shortcut))), ref["dx"], "BasicBlock dx (+shortcut)")


SWEEP = [
    # (C, H, K, k, stride, N): ragged column counts, odd batches, every tile-height / tail-slice / two-tap / transposed route
    (64, 10, 64, 3, 1, 3), (64, 10, 192, 3, 1, 5), (128, 12, 64, 3, 2, 7), (192, 6, 320, 3, 2, 9), (256, 4, 256, 3, 2, 33),
    (96, 14, 128, 3, 1, 2), (64, 6, 128, 1, 1, 11), (64, 9, 256, 1, 1, 3), (320, 5, 64, 1, 1, 13), (128, 7, 1024, 1, 1, 6),
    (1024, 3, 128, 1, 1, 17), (32, 8, 64, 3, 1, 4), (64, 8, 96, 3, 1, 2), (128, 2, 128, 3, 1, 3), (128, 16, 128, 3, 2, 1),
    (512, 4, 512, 3, 1, 40), (64, 32, 64, 1, 1, 9),
]


@pytest.mark.parametrize("shape", SWEEP, ids=["C%d_H%d_K%d_k%d_s%d_N%d" % s for s in SWEEP])
def test_conv_shape_sweep(ops, oracle, shape):
    """shapes chosen to hit the planners' corners rather than the network's layers: column counts that are not multiples
    of 128 (ragged last tile, with and without a sliced tail round), 64- and 128-row tiles, channel counts that tile for
    one operator but not another (fallback to the direct / plain-GEMM kernels mid-layer), 2x2 and 3x3 images (every tap
    out of the image somewhere), odd batches"""
    C, H, K, k, stride, N = shape
    x, w, dy = _conv_data(*shape, seed=11)
    check_act(nhwc(ops.conv_fwd(nchw(x), w, stride)), oracle.conv_fwd(x, w, stride), "fwd %s" % (shape,))
    base = rand((N, H, H, C), 98)
    check_grad(nhwc(ops.conv_dgrad(w, nchw(dy), H, stride)), oracle.conv_dgrad(w, dy, H, stride), "dgrad %s" % (shape,))
    check_grad(nhwc(ops.conv_dgrad(w, nchw(dy), H, stride, dx_init=nchw(base))), oracle.conv_dgrad(w, dy, H, stride, dx_init=base),
               "dgrad+add %s" % (shape,))
    check_grad(ops.conv_wgrad(nchw(x), nchw(dy), k, stride), oracle.conv_wgrad(x, dy, k, stride), "wgrad %s" % (shape,))


# (C, H, K, k, stride, N): dgrads as backwards_pass chains them with the batch-norm backward in front of them
DGRAD_BN_SHAPES = [
    (64, 56, 256, 1, 1, 2),      # expansion dgrad -> spatial BN'
    (256, 56, 64, 1, 1, 2),      # reduction dgrad -> the expansion BN' of the identity block below (+ shortcut addend)
    (128, 28, 128, 3, 1, 3),     # spatial dgrad -> reduction BN'
    (512, 28, 128, 1, 1, 2),
    (1024, 14, 256, 1, 1, 3),
    (256, 14, 256, 3, 1, 3),
    (2048, 7, 512, 1, 1, 5),     # P = 49: the scalar-staging variant of the kernel; 245 columns: ragged last tile
    (512, 7, 512, 3, 1, 5),
    (128, 56, 128, 3, 2, 2),     # stride-2 dgrad: not fused, the separate reduction pass runs
    (64, 10, 64, 1, 1, 3),       # ragged: 300 columns
    (256, 4, 128, 1, 1, 4),      # 64 columns: half a tile
]


@pytest.mark.parametrize("shape", DGRAD_BN_SHAPES, ids=["C%d_H%d_K%d_k%d_s%d_N%d" % s for s in DGRAD_BN_SHAPES])
@pytest.mark.parametrize("with_addend", [0, 1])
def test_dgrad_with_the_bn_backward_reduction_in_its_epilogue_f32(ops, oracle, shape, with_addend):
    """prepreAndDoConvolutionDeriv + activationAndBatchNormDeriv chained (resnet.cu:1399-1429, 1455-1480): on the MFMA route a stride-1
    dgrad gates its output by the sign of the activation, stores the gated gradient and leaves per-tile sums of g and g (x - mean);
    the batch norm backward then only merges and applies.  Against the oracle's dgrad -> ReLU' -> BN' chain."""
    C, H, K, k, stride, N = shape
    eps = 1e-7
    _, w, dy = _conv_data(*shape)
    bn_x = (rand((N, H, H, C), 21, 1.5) + 0.3).astype(np.float32)   # the convolution output the batch norm normalised
    gamma = (1 + 0.2 * rand((C,), 22)).astype(np.float32)
    beta = (0.3 * rand((C,), 23)).astype(np.float32)
    means, vars_, xhat, norm, act = oracle.bn_fwd(bn_x, gamma, beta, eps, 1)
    addend = rand((N, H, H, C), 24) if with_addend else None
    ref_d = oracle.conv_dgrad(w, dy, H, stride, dx_init=addend) if with_addend else oracle.conv_dgrad(w, dy, H, stride)
    ref_g = np.where(act > 0, ref_d, 0).astype(np.float32)
    rdx, rdg, rdb = oracle.bn_bwd(bn_x, gamma, eps, means, vars_, xhat, act, ref_g, 0)   # dy already gated
    gated, bdx, dg, db, fused = ops.conv_dgrad_bn_bwd_f32(w, nchw(dy), H, stride, nchw(bn_x), nchw(act), gamma, beta, means, vars_, eps,
                                                          addend=nchw(addend) if with_addend else None)
    assert fused == (stride == 1), "which launches fuse"
    check_grad(nhwc(gated), ref_g, "gated dgrad %s" % (shape,))
    assert np.array_equal(nhwc(gated) == 0, ref_g == 0) or rel_l2(nhwc(gated), ref_g) < 1e-5  # the gates themselves
    check_grad(db, rdb, "dbeta")
    check_grad(dg, rdg, "dgamma")
    check_grad(nhwc(bdx), rdx, "bn dx")
