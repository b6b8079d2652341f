"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/resnet_mi.h declares,
struct layouts match the header, host-only entry points behave like the reference's, and the N>1 path's
semantics (per-rank slices, per-replica BN, SUM of gradients) hold in a world_size-2 gloo run."""
import ctypes as C
import json
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


def _arena_offsets(dims):
    """float offsets of every locations[] tensor in the product's parameter-shaped arenas: carve order of
    init_model_parameters (resnet.cu:838-943), each tensor aligned up to 64 floats (trainer.c ARENA_ALIGN)"""
    import synth
    sizes = [size for size, _, _ in synth.location_table(dims)]
    offs, off = [], 0
    for sz in sizes:
        offs.append(off)
        off += (sz + 63) // 64 * 64
    return offs, sizes, off


def _dp_plan(lib, dims, bucket_bytes):
    import ctypes as C
    from resnet_amd import binding as B
    flags = (C.c_int * max(dims["n_conv_blocks"], 1))(*dims["is_block_spatial_reduction"])
    d = lib.init_dimensions(dims["input"], dims["init_kernel_dim"], dims["init_conv_filters"], dims["init_conv_stride"],
                            dims["init_maxpool_dim"], dims["init_maxpool_stride"], dims["n_conv_blocks"], flags,
                            dims["final_depth"], dims["output"])
    fr, to = (C.c_size_t * 64)(), (C.c_size_t * 64)()
    n = lib.mi_debug_dp_plan(d, bucket_bytes, fr, to, 64)
    return [(int(fr[i]), int(to[i])) for i in range(n)], int(lib.mi_debug_arena_floats(d)), flags


def test_dp_bucket_plan_resnet50_32mb():
    """host-only (mi_debug_dp_plan, the arithmetic backwards_pass itself uses): for the reference-defined ResNet-50 with
    32 MB buckets the buckets tile the 190.3 MB gradient arena exactly once, FC side first (the order update_parameters
    walks, resnet.cu:2952), every cut falls on a tensor boundary (no tensor is split across two all-reduces), and every
    bucket but the last is at least one bucket long"""
    import synth
    from resnet_amd import binding as B
    lib = B.load()
    dims = synth.R50_DIMS
    plan, arena, _ = _dp_plan(lib, dims, 32 << 20)
    offs, sizes, total = _arena_offsets(dims)
    assert arena == total and sum(sizes) == 47576128 and len(sizes) == 160
    assert plan[0][1] == arena and plan[-1][0] == 0
    for (f0, t0), (f1, t1) in zip(plan, plan[1:]):
        assert t1 == f0 and f1 < f0  # contiguous, descending, no overlap, no gap
    for f, t in plan:
        assert f in offs, "bucket boundary %d splits a tensor" % f
    assert all((t - f) * 4 >= (32 << 20) for f, t in plan[:-1])
    # cuts exist only where a whole block's gradients are final, and b13's block holds the 75 MB projection weight: 4 buckets
    # (44 / 91 / 45 / 10 MB); the LAST one -- stem + blocks 0-2, what the next forward needs first -- is the small one
    assert 3 <= len(plan) <= 7, plan
    assert (plan[-1][1] - plan[-1][0]) * 4 < (32 << 20)
    assert plan[0][0] <= offs[-1]  # the first bucket holds the FC gradient
    # tiny buckets: one cut per block + FC + stem side, still an exact tiling
    plan2, _, _ = _dp_plan(lib, dims, 1)
    assert len(plan2) == dims["n_conv_blocks"] + 2 and plan2[0][1] == arena and plan2[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(plan2, plan2[1:]))


def test_dp_bucket_plan_more_cuts_than_event_slots():
    """a network with more cut points than bucket slots (MI_MAX_BUCKETS = 64, one event each): 70 blocks and 1-byte buckets would
    cut 72 times.  The plan keeps the last slot for the forced final cut, so the tail of the arena goes out as ONE larger bucket --
    still an exact tiling, no range lost (a lost range would silently miss its Adam step)."""
    import synth
    from resnet_amd import binding as B
    lib = B.load()
    dims = synth.resnet_dims(input=32, n_conv_blocks=70, reductions=(), final_depth=256)
    import ctypes as C
    flags = (C.c_int * 70)(*dims["is_block_spatial_reduction"])
    d = lib.init_dimensions(dims["input"], 7, 64, 2, 3, 2, 70, flags, 256, dims["output"])
    fr, to = (C.c_size_t * 128)(), (C.c_size_t * 128)()
    n = lib.mi_debug_dp_plan(d, 1, fr, to, 128)
    arena = int(lib.mi_debug_arena_floats(d))
    plan = [(int(fr[i]), int(to[i])) for i in range(n)]
    assert n == 64, n
    assert plan[0][1] == arena and plan[-1][0] == 0
    assert all(a[0] == b[1] and b[0] < a[0] for a, b in zip(plan, plan[1:]))
    offs, _, total = _arena_offsets(dims)
    assert total == arena and all(f in offs for f, _ in plan)
    assert plan[-1][1] - plan[-1][0] > plan[-2][1] - plan[-2][0]  # the merged tail: several blocks + the stem


def test_bench_self_launch_notices_a_failed_rank(monkeypatch, capsys):
    """the launcher of a bare `bench.py --gpus N`: a rank that dies must end the launch with ITS exit code while rank 0 is still
    blocked (in a real run: in a collective, waiting for the dead peer) -- the other ranks are terminated, nothing is printed;
    when every rank succeeds, rank 0's line is passed on unchanged"""
    import time
    import types
    sys.path.insert(0, ROOT)
    import bench
    py = sys.executable
    args = types.SimpleNamespace(launch_dry_run=False, gpus=2)
    monkeypatch.setattr(bench, "worker_plan", lambda a, v: [
        {"rank": 0, "cmd": [py, "-c", "import time; time.sleep(120)"], "env": {}},
        {"rank": 1, "cmd": [py, "-c", "import sys; sys.exit(3)"], "env": {}}])
    t0 = time.time()
    assert bench.launch_workers(args, []) == 3
    assert time.time() - t0 < 30
    assert capsys.readouterr().out == ""
    monkeypatch.setattr(bench, "worker_plan", lambda a, v: [
        {"rank": 0, "cmd": [py, "-c", "print('{\"value\": 1}')"], "env": {}},
        {"rank": 1, "cmd": [py, "-c", "pass"], "env": {}}])
    assert bench.launch_workers(args, []) == 0
    assert capsys.readouterr().out.strip() == '{"value": 1}'


def test_bench_traffic_is_stamped_with_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from committed rocprofv3 PMC passes; the record says which sources those were measured on and flags a
    figure that belongs to an older build (bench.py cannot run the profiler on itself)"""
    import types
    sys.path.insert(0, ROOT)
    import bench
    (tmp_path / "resnet_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "profiles").mkdir()
    src = tmp_path / "resnet_amd" / "csrc" / "k.hip"
    src.write_text("// kernel v1\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    sha1 = bench.source_sha16()
    fam = {"hbm_GB_per_step_raw": 57.0}
    (tmp_path / "profiles" / bench.PMC_FILE_F32).write_text(json.dumps({"source_sha16": sha1, "families": {bench.PMC_KEY[5]: fam}}))
    leg = {"dom": 5, "dom_stats": (570, 600.0, 7.1e13, 1.7e11), "fam_stats": None, "fam_serial": None}
    args = types.SimpleNamespace(steps=10)
    roof = bench.roofline_of(leg, args, False)
    assert roof["traffic"] == round(57.0e9 / 57) and roof["traffic_stale"] is False and "traffic_warning" not in roof
    src.write_text("// kernel v2\n")                      # the kernels change, the PMC file does not
    roof = bench.roofline_of(leg, args, False)
    assert roof["traffic_stale"] is True and roof["traffic_source_sha16"] == sha1 and roof["current_source_sha16"] != sha1
    assert "re-run tools/pmc_traffic.sh" in roof["traffic_warning"]
    (tmp_path / "profiles" / bench.PMC_FILE_F32).unlink()   # no file: no figure, and the record says so
    roof = bench.roofline_of(leg, args, False)
    assert roof["traffic"] is None and "not present" in roof["traffic_note"]


def test_bench_self_launch_dry_run():
    """`python bench.py --gpus N` with no launcher around it starts N fresh rank processes itself BEFORE anything touches a GPU
    (the library is not even loaded in the parent).  --launch-dry-run prints what it would start: N workers running this script
    with the same arguments, RANK / LOCAL_RANK = 0..N-1, WORLD_SIZE = N, one rendezvous address for all, dmabuf IPC on."""
    import json
    for n in (2, 8):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "7", "--warmup", "2", "--launch-dry-run"],
                           capture_output=True, text=True, timeout=120, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_workers"] == n and d["touches_gpu"] is False and len(d["workers"]) == n
        ports = set()
        for i, w in enumerate(d["workers"]):
            e = w["env"]
            assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"]) == (str(i), str(i), str(n))
            assert e["MASTER_ADDR"] == "127.0.0.1" and e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
            ports.add(e["MASTER_PORT"])
            assert w["cmd"][1].endswith("bench.py") and "--launch-dry-run" not in w["cmd"]
            assert w["cmd"][2:] == ["--gpus", str(n), "--steps", "7", "--warmup", "2"]
        assert len(ports) == 1
    # the parent of a real launch must not have loaded the HIP library: bench.py imports resnet_amd only in the worker path
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("sys.exit(launch_workers(args, argv))") < src.index("lib = B.load()")


DP_WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
import torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import synth, test_host
from oracle.oracle_py import Oracle, OracleNet
from resnet_amd import dp, binding as B
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
raw = dp.exchange_unique_id(dist, rank, 128, lambda: bytes(range(128)))
assert raw == bytes(range(128))
lib = B.load()
dims, per = synth.C1S_DIMS, 2
# PRODUCT code: the arena layout and the bucket plan every rank derives for itself must agree across ranks
plan, arena, _ = test_host._dp_plan(lib, dims, 256 << 10)
offs, sizes, total = test_host._arena_offsets(dims)
assert arena == total
t = torch.tensor([arena] + [x for b in plan for x in b], dtype=torch.int64)
t0 = t.clone(); dist.broadcast(t0, src=0)
assert torch.equal(t, t0), "ranks disagree on the bucket plan"
# the checker: this rank's gradients (its own slice of the global batch, its own BN statistics)
o = Oracle("f32"); net = OracleNet(o, dims, per)
params = synth.make_params(dims, perturb_bn=True)
for i, p in enumerate(params): net.param(i)[:] = p
si, sl = dp.rank_seeds(rank)
im, lab = synth.make_batch(dims, per, seed_img=si, seed_lab=sl)
net.set_batch(im, lab); net.forward(); net.backward()
g = torch.zeros(arena, dtype=torch.float32)
for i in range(net.n_locations):
    g[offs[i]:offs[i] + sizes[i]] = torch.from_numpy(net.grad(i).copy().ravel())
# what backwards_pass hands to ncclAllReduce(ncclSum): the product's buckets, in the product's order, in place
for f, to in plan:
    dist.all_reduce(g[f:to], op=dist.ReduceOp.SUM)
if rank == 0:
    np.savez(sys.argv[2], arena=g.numpy(), plan=np.array(plan))
dist.barrier(); dist.destroy_process_group()
"""


def test_dp_bucket_plan_allreduce_gloo_world2(tmp_path):
    """world_size 2 on gloo (CPU).  What runs through PRODUCT code here: the rendezvous helpers, the per-rank stream seeds,
    the gradient-arena layout and the bucket plan (mi_debug_dp_plan = the arithmetic of backwards_pass), which must be
    identical on every rank and tile the arena exactly once.  The gradients themselves come from the CPU oracle (there is no
    GPU here); all-reducing them bucket by bucket over the product's plan must equal the oracle's "2 BN groups, summed
    gradients" statement.  The RCCL launches themselves are covered on the GPU (tests/test_gpu_dp.py)."""
    import synth
    from oracle.oracle_py import Oracle, OracleNet
    from resnet_amd import dp
    script, out = tmp_path / "w.py", tmp_path / "g.npz"
    script.write_text(DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", OMP_NUM_THREADS="2")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29653", str(script), ROOT, str(out)], env=env, timeout=600)
    got = np.load(out)
    dims, per = synth.C1S_DIMS, 2
    offs, sizes, total = _arena_offsets(dims)
    assert len(got["plan"]) >= 3  # several buckets at this size: the per-bucket path was exercised
    o = Oracle("f32")
    params = synth.make_params(dims, perturb_bn=True)
    want = np.zeros(total, np.float32)
    for rank in range(2):
        net = OracleNet(o, dims, per)
        for i, p in enumerate(params):
            net.param(i)[:] = p
        si, sl = dp.rank_seeds(rank)
        im, lab = synth.make_batch(dims, per, seed_img=si, seed_lab=sl)
        net.set_batch(im, lab)
        net.forward()
        net.backward()
        for i in range(net.n_locations):
            want[offs[i]:offs[i] + sizes[i]] += net.grad(i).ravel()
        net.close()
    assert np.array_equal(got["arena"], want)


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
