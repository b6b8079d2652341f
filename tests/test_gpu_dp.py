"""Data-parallel path on the GPU, through PRODUCT code.

* DP-2 emulated on one GPU: two Trainers (rank 0 / rank 1 slices of the global batch, the seeds bench.py gives the ranks)
  run forward / backward; their gradient arenas are summed on the host -- what ncclAllReduce(ncclSum) does -- and injected
  into both; update_parameters then must give, on both replicas, what the oracle gives for "2 BN groups of N images,
  summed gradients, one Adam step" (the DP parity definition of SURVEY 8e; BN statistics per replica).
* The RCCL launches with a one-rank communicator at the benchmark network's real bucket geometry (32 MB buckets over the
  190.3 MB arena): the buckets issued by backwards_pass are exactly mi_debug_dp_plan's, per-bucket Adam gives bit-identical
  parameters to the single-launch path.
* Two real ranks over RCCL when the box has two GPUs (skipped VISIBLY otherwise: the builder's boxes have one)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from util import rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HYPER = dict(lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7)


def _trainer(dims, batch, params):
    from resnet_amd import Trainer
    from resnet_amd import binding as B
    tr = Trainer(dims, batch, **HYPER)
    if tr.L.mi_device_count() < 1:
        pytest.fail("no HIP device: this test must run on the MI355X box")
    tr.set_params(params)
    tr.source_host(B.MI_LAYOUT_NHWC)
    return tr


@pytest.mark.parametrize("cfg", ["C1", "C1S"])
def test_dp2_emulated_on_one_gpu_matches_two_bn_groups(oracle, cfg):
    from oracle.oracle_py import OracleNet
    from resnet_amd import dp
    dims, per = (synth.C1_DIMS, 4) if cfg == "C1" else (synth.C1S_DIMS, 3)
    params = synth.make_params(dims, perturb_bn=True)
    trs = [_trainer(dims, per, params) for _ in range(2)]
    nets = [OracleNet(oracle, dims, per) for _ in range(2)]
    try:
        for net in nets:
            net.set_hyper(HYPER["lr"], HYPER["wd"], HYPER["b1"], HYPER["b2"], HYPER["eps"])
            for i, p in enumerate(params):
                net.param(i)[:] = p
        for step in range(2):
            for rank in range(2):
                si, sl = dp.rank_seeds(rank)
                im, lab = synth.make_batch(dims, per, step=step, seed_img=si, seed_lab=sl)
                nets[rank].set_batch(im, lab); nets[rank].forward(); nets[rank].backward()
                trs[rank].fill_host_batch(im, lab); trs[rank].load_new_batch(); trs[rank].forward(); trs[rank].backward()
                trs[rank].check()
            n = nets[0].n_locations
            # the exchange step: SUM, no averaging (resnet.cu:1806-1811)
            gsum = [trs[0].get("grads", i) + trs[1].get("grads", i) for i in range(n)]
            osum = [nets[0].grad(i) + nets[1].grad(i) for i in range(n)]
            # second step: the replicas start from parameters that differ from the oracle's in the 7th digit (Adam's first steps
            # are ~lr * sign(g)), and a ReLU gate that sits on such a difference moves every gradient upstream of it by
            # 1e-3..1e-2 -- hence the wider band there; the parameters stay within 1e-5 (1e-4 after two updates: an Adam step is ~lr in every coordinate whatever the gradient's size)
            gtol = 1e-4 if step == 0 else 3e-2
            for i in range(n):
                assert rel_l2(gsum[i], osum[i]) <= gtol, "summed gradient %d step %d: %.3e" % (i, step, rel_l2(gsum[i], osum[i]))
            for rank in range(2):
                for i in range(n):
                    trs[rank].set("grads", i, gsum[i])
                    nets[rank].grad(i)[:] = osum[i]
                trs[rank].update(); nets[rank].update()
                assert trs[rank].check_errors() == 0
            for i in range(n):
                p0, p1 = trs[0].get("params", i), trs[1].get("params", i)
                assert np.array_equal(p0, p1), "replicas diverged at location %d" % i  # identical Adam on every replica
                assert rel_l2(p0, nets[0].param(i)) <= (1e-5 if step == 0 else 1e-4), "param %d step %d" % (i, step)
                assert np.all(trs[0].get("grads", i) == 0)  # the update cleared the arena (resnet.cu:2972-2978)
    finally:
        for t in trs:
            t.close()
        for net in nets:
            net.close()


def _plan(lib, tr, bucket_bytes):
    fr, to = (C.c_size_t * 64)(), (C.c_size_t * 64)()
    n = lib.mi_debug_dp_plan(tr.c_dims, bucket_bytes, fr, to, 64)
    return [(int(fr[i]), int(to[i])) for i in range(n)]


def test_rccl_one_rank_resnet50_bucket_geometry():
    """the benchmark network, batch 2, 32 MB buckets, one-rank RCCL communicator (all-reduce = identity): the buckets
    backwards_pass hands to RCCL are the planned ones, and per-bucket Adam == single-launch Adam bit for bit"""
    from resnet_amd import Trainer
    from resnet_amd import binding as B
    dims, batch = synth.R50_DIMS, 2
    params = synth.make_params(dims, perturb_bn=True)
    im, lab = synth.make_batch(dims, batch, step=0)
    results = []
    for with_comm in (False, True):
        tr = _trainer(dims, batch, params)
        try:
            if with_comm:
                nbytes = tr.L.mi_dp_unique_id_bytes()
                uid = (C.c_char * nbytes)()
                assert tr.L.mi_dp_get_unique_id(uid, nbytes) == 0, tr.error()
                assert tr.L.mi_dp_init(tr.t, 0, 1, uid, nbytes) == 0, tr.error()
                tr.L.mi_dp_set_bucket_bytes(tr.t, 32 << 20)
            tr.fill_host_batch(im, lab); tr.load_new_batch(); tr.forward(); tr.backward(); tr.check()
            if with_comm:
                fr, to = (C.c_size_t * 64)(), (C.c_size_t * 64)()
                n = tr.L.mi_debug_last_buckets(tr.t, fr, to, 64)
                issued = [(int(fr[i]), int(to[i])) for i in range(n)]
                assert issued == _plan(tr.L, tr, 32 << 20) and len(issued) >= 3
                assert issued[0][1] == tr.L.mi_debug_arena_floats(tr.c_dims) and issued[-1][0] == 0
            tr.L.mi_device_synchronize()
            grads = [tr.get("grads", i) for i in range(tr.n_locations)]
            tr.update()
            assert tr.check_errors() == 0
            results.append((grads, [tr.get("params", i) for i in range(tr.n_locations)]))
        finally:
            tr.close()
    for a, b in zip(results[0][0], results[1][0]):
        assert np.array_equal(a, b)
    for a, b in zip(results[0][1], results[1][1]):
        assert np.array_equal(a, b)


RANK_WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import synth
import torch.distributed as dist
from resnet_amd import Trainer, dp, binding as B
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo", rank=rank, world_size=world)
dims, per = synth.C1S_DIMS, 3
tr = Trainer(dims, per, lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7, device=rank)
tr.set_params(synth.make_params(dims, perturb_bn=True))
tr.source_host(B.MI_LAYOUT_NHWC)
dp.init_data_parallel(tr, dist, rank, world, bucket_mb=1)
tr.L.mi_dp_set_bucket_bytes(tr.t, 256 << 10)
si, sl = dp.rank_seeds(rank)
im, lab = synth.make_batch(dims, per, step=0, seed_img=si, seed_lab=sl)
tr.fill_host_batch(im, lab); tr.load_new_batch(); tr.forward(); tr.backward(); tr.check()
tr.L.mi_device_synchronize()
grads = [tr.get("grads", i) for i in range(tr.n_locations)]
tr.update(); assert tr.check_errors() == 0
np.savez(sys.argv[2] + ".%d.npz" % rank, *(grads + [tr.get("params", i) for i in range(tr.n_locations)]))
dist.barrier(); tr.close(); dist.destroy_process_group()
"""


def test_two_ranks_over_rccl(oracle, tmp_path):
    from oracle.oracle_py import OracleNet
    from resnet_amd import binding as B
    from resnet_amd import dp
    if B.load().mi_device_count() < 2:
        pytest.skip("needs 2 GPUs on one node: multi-rank ncclCommInitRank / ncclAllReduce over xGMI is NOT covered on this box")
    script, out = tmp_path / "w.py", str(tmp_path / "r")
    script.write_text(RANK_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                           "--master-port", "29655", str(script), ROOT, out], env=env, timeout=600)
    dims, per = synth.C1S_DIMS, 3
    params = synth.make_params(dims, perturb_bn=True)
    nets = [OracleNet(oracle, dims, per) for _ in range(2)]
    try:
        for rank, net in enumerate(nets):
            net.set_hyper(1e-4, 0.0, 0.9, 0.999, 1e-7)
            for i, p in enumerate(params):
                net.param(i)[:] = p
            si, sl = dp.rank_seeds(rank)
            im, lab = synth.make_batch(dims, per, step=0, seed_img=si, seed_lab=sl)
            net.set_batch(im, lab); net.forward(); net.backward()
        n = nets[0].n_locations
        osum = [nets[0].grad(i) + nets[1].grad(i) for i in range(n)]
        for i in range(n):
            nets[0].grad(i)[:] = osum[i]
        nets[0].update()
        got = [np.load(out + ".%d.npz" % r) for r in range(2)]
        for i in range(n):
            g0, g1 = got[0]["arr_%d" % i], got[1]["arr_%d" % i]
            assert np.array_equal(g0, g1), "ranks hold different reduced gradients at %d" % i
            assert rel_l2(g0, osum[i]) <= 1e-4
            assert np.array_equal(got[0]["arr_%d" % (n + i)], got[1]["arr_%d" % (n + i)])
            assert rel_l2(got[0]["arr_%d" % (n + i)], nets[0].param(i)) <= 1e-5
    finally:
        for net in nets:
            net.close()


def test_sync_bn_option_one_rank_is_the_identity():
    """mi_dp_enable_sync_bn (an option the reference does not have; default off): with a one-rank communicator the per-layer
    all-reduces of (mean), (var + (mean - mean_g)^2) and (dgamma, dbeta) run for real and must change nothing -- same
    predictions, gradients and parameters bit for bit.  The merge itself (equal counts per replica:
    mean_g = avg mean_r, var_g = avg (var_r + (mean_r - mean_g)^2)) is checked against whole-batch statistics in numpy."""
    rng = np.random.RandomState(3)
    x = rng.randn(2, 64, 5).astype(np.float64) * 2 + 0.7  # 2 replicas x 64 samples x 5 channels
    mr, vr = x.mean(axis=1), x.var(axis=1)
    mg = mr.mean(axis=0)
    vg = (vr + (mr - mg) ** 2).mean(axis=0)
    assert np.allclose(mg, x.reshape(-1, 5).mean(axis=0)) and np.allclose(vg, x.reshape(-1, 5).var(axis=0))
    dims, batch = synth.C1S_DIMS, 4
    params = synth.make_params(dims, perturb_bn=True)
    im, lab = synth.make_batch(dims, batch, step=0)
    res = []
    for sync in (False, True):
        tr = _trainer(dims, batch, params)
        try:
            if sync:
                nbytes = tr.L.mi_dp_unique_id_bytes()
                ids = [(C.c_char * nbytes)() for _ in range(2)]
                for u in ids:
                    assert tr.L.mi_dp_get_unique_id(u, nbytes) == 0, tr.error()
                assert tr.L.mi_dp_init(tr.t, 0, 1, ids[0], nbytes) == 0, tr.error()
                assert tr.L.mi_dp_enable_sync_bn(tr.t, ids[1], nbytes) == 0, tr.error()
            tr.fill_host_batch(im, lab); tr.load_new_batch(); tr.forward(); tr.backward(); tr.check()
            tr.L.mi_device_synchronize()
            grads = [tr.get("grads", i) for i in range(tr.n_locations)]
            tr.update()
            assert tr.check_errors() == 0
            res.append((tr.pred(), grads, [tr.get("params", i) for i in range(tr.n_locations)]))
        finally:
            if sync:
                tr.L.mi_dp_enable_sync_bn(tr.t, None, 0)
            tr.close()
    assert np.array_equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1] + res[0][2], res[1][1] + res[1][2]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(6, 64, 8), (4, 256, 7), (3, 2048, 2)])
def test_sync_bn_merge_of_two_different_replicas_equals_the_whole_batch(oracle, ops, shape):
    """Cross-replica batch norm with two replicas that hold DIFFERENT data, on one GPU: each half of a batch of 2N images goes through
    the product's own BN operators (per-replica statistics; backward sums taken with the merged statistics), and the merge in
    between is mi_debug_bn_merge -- the kernels the trainer launches around its sync-BN collectives (bn_sync_k2 / k3 / pack /
    unpack), with the all-reduce replaced by a sum over the two buffers.  Yardstick: the ORACLE's batch norm over the whole batch of
    2N (resnet.cu:289-426): means, variances, dgamma, dbeta, and dx of both halves."""
    from util import nhwc
    N, Cc, H = shape
    eps = 1e-7
    rng = np.random.RandomState(11)
    x = (rng.randn(2 * N, Cc, H, H) * 1.7 + rng.randn(1, Cc, 1, 1)).astype(np.float32)
    x[N:] += 0.8  # the replicas' means differ: the (mean_r - mean_g)^2 term of the variance merge is not negligible
    dy = rng.randn(2 * N, Cc, H, H).astype(np.float32)
    gamma = (1 + 0.1 * rng.randn(Cc)).astype(np.float32)
    beta = (0.1 * rng.randn(Cc)).astype(np.float32)
    # the oracle on the whole batch (NHWC)
    om, ov, oxhat, _onorm, oact = oracle.bn_fwd(nhwc(x), gamma, beta, eps, True)
    odx, odg, odb = oracle.bn_bwd(nhwc(x), gamma, eps, om, ov, oxhat, oact, nhwc(dy), True)
    # forward: per-replica statistics, then the device-side merge
    halves = [slice(0, N), slice(N, 2 * N)]
    stats = [ops.bn_fwd(x[h], gamma, beta, eps, True)[:2] for h in halves]
    means = ops.dev(np.stack([s[0] for s in stats]))
    vars_ = ops.dev(np.stack([s[1] for s in stats]))
    assert ops.L.mi_debug_bn_merge(2, Cc, means.ptr, vars_.ptr, None, None, None) == 0, ops.L.mi_last_error()
    gm, gv = means.get(), vars_.get()
    assert np.array_equal(gm[0], gm[1]) and np.array_equal(gv[0], gv[1])  # every replica ends with the same statistics
    assert rel_l2(gm[0], om) <= 1e-5 and rel_l2(gv[0], ov) <= 1e-5, (rel_l2(gm[0], om), rel_l2(gv[0], ov))
    assert rel_l2(stats[0][0], om) > 1e-2  # (a replica's own mean is NOT the whole-batch mean here)
    # backward: each replica's sums with the MERGED statistics and its own ReLU mask (mask_mode 1), merged, then its dx
    per = [ops.bn_bwd(x[h], gamma, beta, gm[0], gv[0], dy[h], eps, 1) for h in halves]
    dg = ops.dev(np.stack([p[1] for p in per]))
    db = ops.dev(np.stack([p[2] for p in per]))
    sums = ops.dev(shape=(2, 2 * Cc))
    assert ops.L.mi_debug_bn_merge(2, Cc, None, None, dg.ptr, db.ptr, sums.ptr) == 0, ops.L.mi_last_error()
    s = sums.get()
    assert np.array_equal(s[0], s[1])
    assert rel_l2(s[0][:Cc], odg) <= 1e-4 and rel_l2(s[0][Cc:], odb) <= 1e-4, (rel_l2(s[0][:Cc], odg), rel_l2(s[0][Cc:], odb))
    # what goes into the gradient arena is sum / world, so that the arena's own all-reduce SUM restores the global value
    arena_dg, arena_db = dg.get(), db.get()
    assert rel_l2(arena_dg[0] + arena_dg[1], odg) <= 1e-4 and rel_l2(arena_db[0] + arena_db[1], odb) <= 1e-4
    # dx of each half from the global sums: dx = gamma / sigma * (g - sum_g / M - xhat * sum_gxhat / M), M = 2 N H W
    M = 2 * N * H * H
    inv = gamma / np.sqrt(gv[0] + eps)
    for r, h in enumerate(halves):
        xh = (x[h] - gm[0][None, :, None, None]) / np.sqrt(gv[0] + eps)[None, :, None, None]
        act = np.maximum(gamma[None, :, None, None] * xh + beta[None, :, None, None], 0)
        g = np.where(act > 0, dy[h], 0).astype(np.float64)
        dx = inv[None, :, None, None] * (g - s[0][Cc:][None, :, None, None] / M - xh * s[0][:Cc][None, :, None, None] / M)
        assert rel_l2(dx, nchw_(odx[h])) <= 1e-4, "dx of replica %d: %.3e" % (r, rel_l2(dx, nchw_(odx[h])))


def nchw_(a):
    return np.ascontiguousarray(np.transpose(a, (0, 3, 1, 2)))


@pytest.mark.gpu
def test_bench_stdout_is_one_json_line_on_the_distributed_path():
    """the driver reads ONE JSON line from rank 0's stdout: gloo and RCCL print banners on fd 1 ("[Gloo] Rank 0 is connected ...", "RCCL
    version : ..."), which bench.py sends to stderr.  --force-dist runs the rendezvous + RCCL path with one rank."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["MASTER_PORT"] = "29617"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--dtype", "bf16", "--batch", "32", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extra", "--no-prof"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["dtype"] == "bf16" and rec["value"] > 0
