#!/usr/bin/env python3
"""bench.py -- images/sec of the reference-defined ResNet-50 training step (fp32, 224x224, batch 256 per GPU)
on N MI355X, through the drop-in C-ABI (load_new_batch -> forward_pass -> host loss -> backwards_pass ->
update_parameters: the loop of resnet.cu:3340-3402).  Synthetic seeded data resident in HBM, random-init
weights.  One JSON line on rank 0, with a `roofline` object for the dominant kernel family (HIP events
around every launch of it, on the launch stream) and a `cpu_baseline` object (the CPU oracle timed on
this host on a bounded sample; the reference has no CPU path of its own).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3  # gfx950 fp32: vector FMA rate == fp32 MFMA rate (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
FAMILIES = {0: "direct conv 7x7 stem / untiled 3x3 fwd+dgrad (dconv_kernel, VALU)", 1: "direct conv wgrad (wgradC/wgrad_kernel, VALU)",
            2: "1x1 conv / FC GEMM (igemm_kernel<*,1,1,*> / gemm_mfma_kernel, fp32 MFMA)", 3: "batch norm fwd+bwd",
            5: "3x3 conv fwd+dgrad+wgrad incl. the 3x3-s2 projections (igemm_kernel<*,3,*,*>, implicit GEMM on fp32 MFMA)"}


def cpu_baseline(seconds_budget=30.0):
    """The oracle (a CPU port: the reference has no CPU path) on a bounded sample of the same workload:
    reference-defined ResNet-50, 224x224, fwd + bwd + Adam, batch 6, OpenMP over independent outputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from oracle.oracle_py import Oracle, OracleNet
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 32)
    o = Oracle("f32")
    o.set_threads(cores)
    dims, batch = synth.R50_DIMS, 6  # ~10-15 s of CPU work on the GPU box's host cores
    net = OracleNet(o, dims, batch)
    params = synth.make_params(dims)
    for i, p in enumerate(params):
        net.param(i)[:] = p
    im, lab = synth.make_batch(dims, batch)
    net.set_batch(im, lab)
    t0 = time.time()
    net.forward()
    net.loss()
    net.backward()
    net.update()
    dt = time.time() - t0
    net.close()
    return {"value": batch / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "1 full training step (fwd+loss+bwd+Adam) of the reference-defined ResNet-50 fp32 224x224 at batch %d, "
                      "oracle/liboracle_f32.so with %d OpenMP threads, %.1f s" % (batch, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="disable the per-kernel HIP-event timing")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="storage type of activations (bf16 = BASELINE configs[4]); arithmetic is fp32 either way")
    ap.add_argument("--policy", choices=["fast", "recompute_bn"], default="fast", help="what backward keeps from forward (mi_trainer_set_store_policy)")
    ap.add_argument("--force-dist", action="store_true", help="run the torch.distributed + RCCL path even with one rank (self-test)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    from resnet_amd import Trainer, resnet_dims
    from resnet_amd import binding as B
    lib = B.load()
    if lib.mi_device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback exists)")

    dist = None
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29571")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # single node: the container hostname may not resolve
        import torch.distributed as dist  # rendezvous + barrier + max-reduce only (gloo); collectives are RCCL in C
        dist.init_process_group("gloo", rank=rank, world_size=world)

    dims = resnet_dims()
    tr = Trainer(dims, args.batch, lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7, seed=1236, device=local_rank)
    if args.policy == "recompute_bn":
        tr.set_store_policy(B.MI_STORE_RECOMPUTE_BN)
    if args.dtype == "bf16":
        tr.set_dtype(B.MI_DTYPE_BF16)
    # every rank draws its own slice of the global batch: distinct image/label streams per rank
    from resnet_amd import dp
    tr.source_synthetic(*dp.rank_seeds(rank), pool_batches=2)
    if dist is not None:
        dp.init_data_parallel(tr, dist, rank, world, args.bucket_mb)

    def barrier():
        lib.mi_device_synchronize()
        if dist is not None:
            dist.barrier()

    def read_prof():
        st = {}
        for fam in FAMILIES:
            n, ms, fl, by = C.c_long(0), C.c_double(0), C.c_double(0), C.c_double(0)
            lib.mi_prof_get(fam, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by))
            st[fam] = (n.value, ms.value, fl.value, by.value)
        return st

    losses = []
    prof = not args.no_prof
    warm_stats, warm_steps = {}, 0
    for w in range(args.warmup):
        # the last warm-up step is timed per kernel family (HIP events around every launch cost ~3% of a step, so in
        # the timed region only the dominant family found here is bracketed)
        if prof and w == args.warmup - 1:
            lib.mi_prof_enable(1)
            lib.mi_prof_reset()
            warm_steps = 1
        losses.append(tr.step()[0])
    tr.check()
    dom = 0
    if prof:
        if warm_steps:
            warm_stats = read_prof()
            dom = max(warm_stats, key=lambda f: warm_stats[f][1])
        lib.mi_prof_enable(1 << dom if dom else 2 | 1)  # family 0 -> mask 1 (value 1 means "all", so add family 1)
        lib.mi_prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.step()[0])
    lib.mi_device_synchronize()
    dt = time.perf_counter() - t0
    barrier()
    tr.check()
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])

    fam_stats = {}
    if prof:
        fam_stats = read_prof()
        lib.mi_prof_enable(0)
    timings = tr.timings()
    tr.close()

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        out = {"metric": "images/sec ResNet-50 fp32 224x224 batch256", "value": round(value, 2), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "reference-defined ResNet-50 (47.58M params, 3x3-s2 projections, Adam, sum loss), "
                                      "fp32, 224x224, full training step, batch %d per GPU" % args.batch,
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "phase_ms_last_step": {"forward": round(timings[1], 3), "backward": round(timings[2], 3),
                                                 "update": round(timings[3], 3)},
                          "final_loss_per_image": round(losses[-1] / args.batch, 4)}}
        if fam_stats:
            n, ms, fl, by = fam_stats[dom]
            if dom == 3:
                ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None}
            else:
                ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / FP32_PEAK_TFLOPS, 4), "traffic": None}
            # HBM traffic of that family from the committed rocprofv3 PMC passes (bench.py cannot run the profiler itself)
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
                key = {0: "dconv", 1: "wgradC", 2: "gemm", 3: "bn", 5: "igemm3x3"}[dom]
                # the PMC passes count kernel launches, this run counts logical launches (the four concurrent stride-2 dgrad
                # class kernels are one): convert through bytes per step
                roof["traffic"] = round(pmc["families"][key]["hbm_GB_per_step_raw"] * 1e9 / (n / args.steps))
                roof["traffic_note"] = ("HBM bytes per launch (FETCH_SIZE+WRITE_SIZE, separate --pmc passes, profiles/r1_pmc_traffic.json; "
                                        "4-B/lane gathers: gfx950 FETCH halving uncalibrated, raw value; Infinity-Cache hits are counted); algorithmic bytes per launch: %d" % round(by / max(n, 1)))
            except Exception:
                pass
            roof["kernel"] = FAMILIES[dom]
            roof["launches"] = n
            roof["avg_launch_ms"] = round(ms / max(n, 1), 4)
            roof["algorithmic_gflop_per_launch"] = round(fl / max(n, 1) / 1e9, 3)
            roof["note"] = "fp32 peak 157.3 TFLOP/s is both the vector-FMA and the fp32-MFMA rate on gfx950"
            if warm_stats:  # whole-step breakdown from the profiled warm-up step (all families bracketed)
                roof["families_ms_per_step_warmup"] = {FAMILIES[f].split(" (")[0]: round(warm_stats[f][1] / warm_steps, 3) for f in warm_stats}
                roof["families_tflops_warmup"] = {FAMILIES[f].split(" (")[0]: round(warm_stats[f][2] / max(warm_stats[f][1], 1e-9) / 1e9, 2)
                                                  for f in warm_stats if warm_stats[f][2] > 0}
                roof["families_gbs_warmup"] = {FAMILIES[f].split(" (")[0]: round(warm_stats[f][3] / max(warm_stats[f][1], 1e-9) / 1e6, 1)
                                               for f in warm_stats if warm_stats[f][3] > 0}
            out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
