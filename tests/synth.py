"""Seeded synthetic inputs for the tests: splitmix64 counter streams.

images  U(-124, 152)  seed 1234   (range of mean-subtracted bytes, build_training_shards.c:118-126)
labels  uniform [0, n_classes)    seed 1235
weights N(0, var) with the reference's variances (resnet.cu:726-790, 831, 938)   seed 1236
The same streams are produced in C by resnet_amd/csrc/synth.c (checked in test_host.py).
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n, offset=0):
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _G
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n, offset=0):
    """float64 in [0,1) with 53 bits"""
    return (splitmix64(seed, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, n, lo, hi, offset=0):
    return (lo + (hi - lo) * uniform01(seed, n, offset)).astype(np.float32)


def normal(seed, n, var, offset=0):
    """Box-Muller on pairs (u[2i], u[2i+1]) -> one normal each (cos branch)"""
    u = uniform01(seed, 2 * n, 2 * offset)
    u1 = 1.0 - u[0::2]  # (0,1]
    u2 = u[1::2]
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (z * np.sqrt(var)).astype(np.float32)


def labels(seed, n, n_classes, offset=0):
    return (splitmix64(seed, n, offset) % np.uint64(n_classes)).astype(np.int32)


def resnet_dims(input=224, n_conv_blocks=16, reductions=(3, 7, 13), final_depth=2048, output=1000,
                init_conv_filters=64):
    flags = [1 if i in reductions else 0 for i in range(n_conv_blocks)]
    return dict(input=input, init_kernel_dim=7, init_conv_filters=init_conv_filters, init_conv_stride=2,
                init_maxpool_dim=3, init_maxpool_stride=2, n_conv_blocks=n_conv_blocks,
                is_block_spatial_reduction=flags, final_depth=final_depth, output=output)


# BASELINE.json configs[0]: 1 bottleneck block, batch 4, 32x32 (reference-defined net)
C1_DIMS = resnet_dims(input=32, n_conv_blocks=1, reductions=(), final_depth=256)
C1_BATCH = 4
# a 3-block net with one striding block: exercises the 3x3-s2 projection, identity residual, toAdd
C1S_DIMS = resnet_dims(input=32, n_conv_blocks=3, reductions=(1,), final_depth=512)
# 4 blocks, the third one striding: an identity block FOLLOWED by another block (the upper block's reduction dgrad feeds the lower one's
# expansion BN': the fused-reduction route of the bf16 trainer), at 8x8 and 4x4 planes
C4I_DIMS = resnet_dims(input=32, n_conv_blocks=4, reductions=(2,), final_depth=512)
R50_DIMS = resnet_dims()


def location_table(dims):
    """(size, kind, fan) per parameter tensor in the reference's locations[] order (resnet.cu:838-943).
    kind: 'w' conv weight (var = 2/fan), 'g' gamma (=1), 'b' beta (=0), 'fc' (var 1e-4)."""
    f = dims["init_conv_filters"]
    k = dims["init_kernel_dim"]
    t = [(k * k * f * 3, "w", 7 * 7 * (3 + f)), (f, "g", 0), (f, "b", 0)]
    inc, red, ex = f, f, 4 * f
    for i in range(dims["n_conv_blocks"]):
        stride = 1
        if dims["is_block_spatial_reduction"][i]:
            stride, red, ex = 2, red * 2, ex * 2
        t += [(inc * red, "w", inc + red), (red, "g", 0), (red, "b", 0)]
        t += [(red * red * 9, "w", 9 * (red + red)), (red, "g", 0), (red, "b", 0)]
        t += [(ex * red, "w", red + ex), (ex, "g", 0), (ex, "b", 0)]
        if inc != ex:
            if stride == 2:
                t += [(9 * inc * ex, "w", 9 * (inc + ex))]
            else:
                t += [(inc * ex, "w", inc + ex)]
            t += [(ex, "g", 0), (ex, "b", 0)]
        inc = ex
    t += [(ex * dims["output"], "fc", 0)]
    return t


def make_params(dims, seed=1236, perturb_bn=False):
    """list of float32 arrays in locations[] order.  perturb_bn: gamma/beta drawn near 1/0 so that
    BN parameter gradients are exercised non-trivially in op tests."""
    out, off = [], 0
    for size, kind, fan in location_table(dims):
        if kind == "w":
            a = normal(seed, size, 2.0 / fan, off)
        elif kind == "fc":
            a = normal(seed, size, 1e-4, off)
        elif kind == "g":
            a = np.ones(size, np.float32) if not perturb_bn else (1 + 0.1 * normal(seed, size, 1.0, off))
        else:
            a = np.zeros(size, np.float32) if not perturb_bn else (0.1 * normal(seed, size, 1.0, off))
        out.append(np.ascontiguousarray(a, np.float32))
        off += size
    return out


def make_batch(dims, batch, seed_img=1234, seed_lab=1235, step=0):
    """images NHWC float32 (the reference layout, resnet.cu:145), labels int32"""
    n = batch * dims["input"] * dims["input"] * 3
    im = uniform(seed_img, n, -124.0, 152.0, offset=step * n).reshape(batch, dims["input"], dims["input"], 3)
    lab = labels(seed_lab, batch, dims["output"], offset=step * batch)
    return im, lab
