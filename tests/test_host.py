"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/resnet_mi.h declares,
struct layouts match the header, host-only entry points behave like the reference's, and the N>1 path's
semantics (per-rank slices, per-replica BN, SUM of gradients) hold in a world_size-2 gloo run."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "resnet_mi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    body = src[src.index("typedef struct MiRng"):]
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", body)))


def test_library_exports_every_declared_symbol():
    from resnet_amd import binding
    lib = binding.load()  # raises if the .so is missing: there is no fallback
    declared = _header_functions()
    assert len(declared) >= 55
    for name in declared:
        assert hasattr(lib, name), "libresnet_mi.so does not export %s" % name
        assert name in binding.PROTOTYPES, "binding.py does not bind %s" % name
    assert sorted(binding.PROTOTYPES) == declared


def test_struct_layouts_match_reference_header_order():
    from resnet_amd import binding as B
    # field order of resnet.h (the cudnn handle slot of resnet_cudnn.h:213 sits between init_loaded and dump_dir)
    names = [f[0] for f in B.Train_ResNet._fields_]
    assert names[:4] == ["model", "cur_batch", "forward_buffer", "backprop_buffer"]
    assert names[-3:] == ["init_loaded", "backend_ctx", "dump_dir"]
    assert [f[0] for f in B.Cache_BatchNorm._fields_] == ["input_size", "feature_size", "means", "vars", "normalized_temp", "normalized"]
    assert C.sizeof(B.Dims) == 48 and C.sizeof(B.BatchNorm) == 24 and C.sizeof(B.Batch) == 72
    # compile a C probe against the real header and compare sizes/offsets
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "resnet_mi.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(Train_ResNet), offsetof(Train_ResNet, backend_ctx),
  sizeof(Activation_ConvBlock), offsetof(Activation_ConvBlock, output_activated), sizeof(Params), sizeof(Activations), sizeof(ConvBlock)); return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(probe)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "p"), os.path.join(d, "p.c")])
        out = subprocess.check_output([os.path.join(d, "p")]).decode().split()
    got = [C.sizeof(B.Train_ResNet), B.Train_ResNet.backend_ctx.offset, C.sizeof(B.Activation_ConvBlock),
           B.Activation_ConvBlock.output_activated.offset, C.sizeof(B.Params), C.sizeof(B.Activations), C.sizeof(B.ConvBlock)]
    assert [int(x) for x in out] == got


def test_init_dimensions_and_class_info():
    """host-only entry points: init_dimensions (resnet.cu:666) and populate_class_info (resnet.cu:1363) on the
    reference's own metadata format (one line per class; counts sum to 1,281,167 for ImageNet-1k)"""
    from resnet_amd import binding as B
    lib = B.load()
    flags = (C.c_int * 16)(*[1 if i in (3, 7, 13) else 0 for i in range(16)])
    d = lib.init_dimensions(224, 7, 64, 2, 3, 2, 16, flags, 2048, 1000).contents
    assert (d.input, d.init_kernel_dim, d.n_conv_blocks, d.final_depth, d.output) == (224, 7, 16, 2048, 1000)
    assert d.is_block_spatial_reduction[7] == 1
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for nm, rows in (("labels", ["tench", "goldfish", "shark"]), ("synsets", ["n01440764", "n01443537", "n01484850"]),
                         ("counts", ["1300", "1300", "1267"])):
            p = os.path.join(tmp, nm)
            open(p, "w").write("\n".join(rows) + "\n")
            paths.append(p.encode())

        class CM(C.Structure):
            _fields_ = [("labels", C.POINTER(C.c_char_p)), ("synsets", C.POINTER(C.c_char_p)), ("counts", C.POINTER(C.c_int)), ("n", C.c_int)]
        cm = C.cast(lib.populate_class_info(paths[0], paths[1], paths[2], 3), C.POINTER(CM)).contents
        assert cm.n == 3 and [cm.counts[i] for i in range(3)] == [1300, 1300, 1267]
        assert cm.labels[1].decode().strip() == "goldfish" and cm.synsets[2].decode().strip() == "n01484850"


def test_product_does_not_touch_the_oracle():
    """the oracle is test infrastructure: nothing under resnet_amd/ may import or link it"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "resnet_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("# the oracle", ""), os.path.join(dirpath, f)
    deps = subprocess.check_output(["ldd", os.path.join(ROOT, "resnet_amd", "libresnet_mi.so")]).decode()
    assert "liboracle" not in deps


DP_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
import torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import synth
from oracle.oracle_py import Oracle, OracleNet
from resnet_amd import dp
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
raw = dp.exchange_unique_id(dist, rank, 128, lambda: bytes(range(128)))
assert raw == bytes(range(128))
dims, per = synth.C1_DIMS, 2
o = Oracle("f32"); net = OracleNet(o, dims, per)
params = synth.make_params(dims, perturb_bn=True)
for i, p in enumerate(params): net.param(i)[:] = p
si, sl = dp.rank_seeds(rank)
im, lab = synth.make_batch(dims, per, seed_img=si, seed_lab=sl)
net.set_batch(im, lab); net.forward(); net.backward()
out = []
for i in range(net.n_locations):
    g = torch.from_numpy(net.grad(i).copy())
    dist.all_reduce(g, op=dist.ReduceOp.SUM)   # what ncclAllReduce(ncclSum) does to the gradient arena
    out.append(g.numpy())
if rank == 0:
    np.savez(sys.argv[2], *out)
dist.barrier(); dist.destroy_process_group()
'''


def test_data_parallel_semantics_gloo_world2(tmp_path):
    import synth
    from oracle.oracle_py import Oracle, OracleNet
    from resnet_amd import dp
    script, out = tmp_path / "w.py", tmp_path / "g.npz"
    script.write_text(DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29653", str(script), ROOT, str(out)], env=env, timeout=600)
    got = np.load(out)
    # single-process statement of the same thing: two BN groups of 2 images, gradients summed
    dims, per = synth.C1_DIMS, 2
    o = Oracle("f32")
    params = synth.make_params(dims, perturb_bn=True)
    total = None
    for rank in range(2):
        net = OracleNet(o, dims, per)
        for i, p in enumerate(params):
            net.param(i)[:] = p
        si, sl = dp.rank_seeds(rank)
        im, lab = synth.make_batch(dims, per, seed_img=si, seed_lab=sl)
        net.set_batch(im, lab)
        net.forward()
        net.backward()
        g = [net.grad(i).copy() for i in range(net.n_locations)]
        total = g if total is None else [a + b for a, b in zip(total, g)]
        net.close()
    for i, t in enumerate(total):
        assert np.array_equal(got["arr_%d" % i], t), "location %d" % i


def test_conv_launch_planners_on_the_benchmark_shapes():
    """host-only view of the implicit-GEMM planners (mi_debug_conv_plan; no GPU): for every convolution of the
    reference-defined ResNet-50 at batch 256 and every operator -- the MFMA route is taken, tiles / slices / splits are
    consistent, the sliced tail round fits the chip and its partial-tile buffer, and workgroup rounds are well filled."""
    import ctypes as C
    from resnet_amd import binding as B
    L = B.load()
    layers = [  # (C, H, K, k, stride)
        (64, 56, 64, 3, 1), (128, 56, 128, 3, 2), (256, 56, 512, 3, 2), (128, 28, 128, 3, 1), (256, 28, 256, 3, 2),
        (512, 28, 1024, 3, 2), (256, 14, 256, 3, 1), (512, 14, 512, 3, 2), (1024, 14, 2048, 3, 2), (512, 7, 512, 3, 1),
        (64, 56, 256, 1, 1), (256, 56, 64, 1, 1), (256, 56, 128, 1, 1), (512, 28, 128, 1, 1), (128, 28, 512, 1, 1),
        (512, 28, 256, 1, 1), (1024, 14, 256, 1, 1), (256, 14, 1024, 1, 1), (1024, 14, 512, 1, 1), (2048, 7, 512, 1, 1),
        (512, 7, 2048, 1, 1)]
    N = 256
    worst = 1.0
    for (Cc, H, K, k, s) in layers:
        for op in (0, 1, 2):
            out = (C.c_int * 9)()
            assert L.mi_debug_conv_plan(op, N, Cc, H, K, k, s, out) == 1, (op, Cc, H, K, k, s)
            route, bm, tiles, full, tsplit, tklen, splits, wgs, ksteps = list(out)
            assert route == 1 and bm in (64, 128) and tiles >= 1 and ksteps >= 1
            slots = 512 if bm == 128 else 768
            if op == 2:
                assert splits >= 1 and tklen * splits >= ksteps > tklen * (splits - 1)  # splits cover the reduction, none empty
                rounds = tiles * splits / slots
            else:
                assert 0 <= full <= tiles and tsplit >= 1 and (tiles - full) * tsplit <= slots        # the sliced round fits the chip
                assert (tiles - full) * tsplit * bm * 128 <= 512 * 128 * 128                          # and the partial-tile buffer
                assert tsplit == 1 or (tklen * tsplit >= ksteps > tklen * (tsplit - 1) and tklen >= 8)
                assert wgs == full + (tiles - full) * tsplit
                rounds = full / slots + (1.0 / tsplit if tiles > full else 0.0) if tsplit > 1 else tiles / slots
            fill = rounds / max(1, -(-int(rounds * 1e6) // 10 ** 6)) if tsplit == 1 or op == 2 else 1.0
            worst = min(worst, fill)
    assert worst > 0.45, "some layer leaves more than half of its last round of workgroups empty without slicing it"
    # the stem and a shape that does not tile stay on the other kernels
    out = (C.c_int * 9)()
    assert L.mi_debug_conv_plan(0, N, 3, 224, 64, 7, 2, out) == 0 and L.mi_debug_conv_plan(2, N, 48, 8, 80, 3, 1, out) == 0
