#!/usr/bin/env python3
"""bench.py -- images/sec of the reference-defined ResNet-50 training step (224x224, batch 256 per GPU) on N MI355X,
through the drop-in C-ABI (load_new_batch -> forward_pass -> host loss -> backwards_pass -> update_parameters: the loop
of resnet.cu:3340-3402).  Synthetic seeded data resident in HBM, random-init weights.  One JSON line on rank 0.

  python bench.py --gpus 1 --steps 10 --warmup 3                      fp32: BASELINE.json's metric (configs[2])
  python bench.py --dtype bf16                                        bf16 activations / fp32 accumulate (configs[4])
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                       data parallel, RCCL all-reduce of the gradients
  python bench.py --gpus N                                            the same without a launcher: this process touches no GPU,
                                                                      starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
                                                                      MASTER_* set), waits, and passes rank 0's JSON line on
  python bench.py --gpus N --launch-dry-run                           prints the N worker commands + environments, starts nothing

What the line carries besides the contract's fields:
  roofline       the dominant kernel family, HIP events around every launch of it on its launch stream INSIDE the timed
                 region (the other families are not bracketed there: ~1200 event records per step cost 2-3 %);
                 roofline.families: every family, from K FURTHER steps with every launch bracketed, run right after the
                 timed region (their own ms/step is given: it shows what the bracketing costs)
  value_h2d_inclusive   K further steps with the batch handed over in pinned host memory every step (the reference's own
                 blocking copy, resnet.cu:1315-1316: 154 MB per step over PCIe) -- never `value`
  cpu_baseline   the CPU oracle (a port: the reference has no CPU path) on this host, bounded samples
  bf16           (fp32 runs only) BASELINE configs[4] in the SAME record: K timed steps of a second trainer with bf16 activations
                 after the fp32 one is gone -- its own value / ms_per_step / roofline; `value`, `dtype`, `config` stay the fp32 line's
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3   # gfx950 fp32: vector FMA rate == fp32 MFMA rate (MI355X_MICROARCH.md)
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md; AMD's 5 PF figure includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0
FAMILIES_F32 = {0: "7x7 stem fwd (st32_pad + st32_fwd_kernel, fp32 MFMA)", 1: "7x7 stem wgrad (st32_wgrad_kernel, fp32 MFMA)",
                2: "1x1 conv / FC GEMM (igemm_kernel<*,1,1,*> / gemm_mfma_kernel, fp32 MFMA)", 3: "batch norm fwd+bwd",
                5: "3x3 conv fwd+dgrad+wgrad incl. the 3x3-s2 projections (igemm_kernel<*,3,*,*>, implicit GEMM on fp32 MFMA)"}
FAMILIES_BF16 = {0: "7x7 stem fwd (st_pad + st_fwd_kernel, bf16 MFMA, bf16 output)", 1: "7x7 stem wgrad (st_wgrad_kernel, bf16 MFMA, bf16 dY)",
                 2: "1x1 conv (bgemm_kernel<*,1,1,*>, bf16 MFMA) + FC GEMM (fp32 MFMA)", 3: "batch norm fwd+bwd (bf16 tensors, fp32 math)",
                 5: "3x3 conv fwd+dgrad+wgrad incl. the 3x3-s2 projections (bgemm_kernel<*,3,*,*> on NCHW; the stride-2 layers on channel-last operands: cl_conv / cl_dgrad2 / cl_wgrad_kernel; implicit GEMM on bf16 MFMA)"}
PMC_KEY = {0: "dconv", 1: "wgradC", 2: "gemm", 3: "bn", 5: "igemm3x3"}
PMC_FILE_F32, PMC_FILE_BF16 = "r3_pmc_traffic.json", "r3_bf16_pmc_traffic.json"


def _time_oracle(o, dims, batch, threads):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from oracle.oracle_py import OracleNet
    o.set_threads(threads)
    net = OracleNet(o, dims, batch)
    for i, p in enumerate(synth.make_params(dims)):
        net.param(i)[:] = p
    im, lab = synth.make_batch(dims, batch)
    net.set_batch(im, lab)
    t0 = time.time()
    net.forward(); net.loss(); net.backward(); net.update()
    dt = time.time() - t0
    net.close()
    return dt


def cpu_baseline():
    """The oracle (a CPU port: the reference has no CPU path) on bounded samples of the same workload, timed on this
    host: the reference-defined ResNet-50 at batch 4 on every core the process may use (the headline `value`), config 1
    (BASELINE configs[0]) in full on one thread and on all of them.  The single-thread ResNet-50 sample is one image (batch 1):
    batch 4 on one thread would take about a minute."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    from oracle.oracle_py import Oracle
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = visible
    try:  # a container's CPU share (cgroup v2 quota) is what this process can really use
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(visible, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    cores = min(cores, 32)  # the oracle's OpenMP loops stop scaling (and a GPU box gives one job ~16 cores of the 256 it shows)
    o = Oracle("f32")
    c1_1 = min(_time_oracle(o, synth.C1_DIMS, synth.C1_BATCH, 1) for _ in range(3))
    c1_n = min(_time_oracle(o, synth.C1_DIMS, synth.C1_BATCH, cores) for _ in range(3))
    batch = 4
    dt = _time_oracle(o, synth.R50_DIMS, batch, cores)
    dt1 = _time_oracle(o, synth.R50_DIMS, 1, 1)  # one image on one thread: the scalar-port rate (~10-20 s)
    return {"value": round(batch / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "1 full training step (fwd+loss+bwd+Adam) of the reference-defined ResNet-50 fp32 224x224 at batch %d, "
                      "oracle/liboracle_f32.so with %d OpenMP threads (`cores` = threads used; the host shows %d CPUs), %.1f s" % (batch, cores, visible, dt),
            "config1_images_per_sec_1_thread": round(synth.C1_BATCH / c1_1, 2),
            "config1_images_per_sec_all_threads": round(synth.C1_BATCH / c1_n, 2),
            "config1_sample": "config 1 (1 block, batch 4, 32x32): one full step, best of 3, %.1f ms on 1 thread, %.1f ms on %d" % (c1_1 * 1e3, c1_n * 1e3, cores),
            "resnet50_images_per_sec_1_thread": round(1.0 / dt1, 4),
            "resnet50_1_thread_sample": "the same step at batch 1 on 1 thread, %.1f s" % dt1}


def source_sha16():
    """identity of the kernel + host sources a measurement belongs to (the GPU box has no .git): sha256 over resnet_amd/csrc"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "resnet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".c", ".h", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true", help="no per-kernel HIP-event timing at all (and no extra instrumented steps)")
    ap.add_argument("--no-extra", action="store_true", help="skip the further steps behind the timed region (families, H2D-inclusive)")
    ap.add_argument("--no-bf16", action="store_true", help="fp32 runs: skip the bf16 leg (the `bf16` key of the record)")
    ap.add_argument("--bucket-mb", type=int, default=32)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32", help="storage type of activations (bf16 = BASELINE configs[4]); arithmetic is fp32 either way")
    ap.add_argument("--policy", choices=["fast", "recompute_bn"], default="fast", help="what backward keeps from forward (mi_trainer_set_store_policy)")
    ap.add_argument("--force-dist", action="store_true", help="run the torch.distributed + RCCL path even with one rank (self-test)")
    ap.add_argument("--launch-dry-run", action="store_true", help="with --gpus N and no WORLD_SIZE: print the N worker commands and environments, start nothing")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` with no launcher around it: THIS process never touches a GPU (no HIP call, the library is not even
# loaded); it starts N fresh worker processes -- one per GPU, the environment torch.distributed.run would give them -- waits for
# all of them and passes rank 0's JSON line on.  A rank that fails takes the others down and the exit code is non-zero.
def worker_plan(args, argv):
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--launch-dry-run"]
    plan = []
    for r in range(args.gpus):
        env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
               "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "GLOO_SOCKET_IFNAME": "lo",
               "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}  # dmabuf IPC (RCCL across processes)
        plan.append({"rank": r, "cmd": cmd, "env": env})
    return plan


def launch_workers(args, argv):
    plan = worker_plan(args, argv)
    if args.launch_dry_run:
        print(json.dumps({"launcher": "bench.py", "n_workers": len(plan), "touches_gpu": False, "workers": plan}))
        return 0
    procs = []
    for w in plan:
        out = subprocess.PIPE if w["rank"] == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(w["cmd"], env=dict(os.environ, **w["env"]), stdout=out))
    line, rc = "", 0
    chunks = []
    import threading
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # (a blocking read here would never
    reader.start()                                                                                # see another rank fail)
    try:
        pending = list(procs)
        while pending:
            for pr in list(pending):
                r = pr.poll()
                if r is None:
                    continue
                pending.remove(pr)
                if r != 0:
                    rc = rc or r
                    for other in pending:  # a failed rank: the others would wait in a collective for ever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    reader.join(timeout=10)
    line = b"".join(chunks).decode()
    if rc == 0:
        sys.stdout.write(line)
        sys.stdout.flush()
    else:
        sys.stderr.write("bench.py: a rank exited with code %d\n" % rc)
    return rc


def run_leg(args, lib, dist, rank, local_rank, world, bf16, extras):
    """One configuration (fp32 or bf16 storage): warm-up, THE timed region of exactly K steps, then (extras) the further
    instrumented steps.  Returns everything the record is assembled from."""
    from resnet_amd import Trainer, resnet_dims, dp
    from resnet_amd import binding as B
    FAMILIES = FAMILIES_BF16 if bf16 else FAMILIES_F32
    dims = resnet_dims()
    tr = Trainer(dims, args.batch, lr=1e-4, wd=0.0, b1=0.9, b2=0.999, eps=1e-7, seed=1236, device=local_rank)
    if args.policy == "recompute_bn":
        tr.set_store_policy(B.MI_STORE_RECOMPUTE_BN)
    if bf16:
        tr.set_dtype(B.MI_DTYPE_BF16)
    act_bytes, dev_bytes = tr.activation_bytes(), tr.device_bytes()
    # every rank draws its own slice of the global batch: distinct image/label streams per rank
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

    def timed(k):
        barrier()
        t0 = time.perf_counter()
        ls = [tr.step()[0] for _ in range(k)]
        lib.mi_device_synchronize()
        dt = time.perf_counter() - t0
        barrier()
        tr.check()
        if dist is not None:
            import torch
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0])
        return dt, ls

    losses = []
    prof = not args.no_prof
    warm_stats = {}
    for w in range(args.warmup):
        # the last warm-up step is bracketed per family only to FIND the dominant one
        if prof and w == args.warmup - 1:
            lib.mi_prof_enable(1)
            lib.mi_prof_reset()
        losses.append(tr.step()[0])
    tr.check()
    dom = 5
    if prof:
        if args.warmup:
            warm_stats = read_prof()
            dom = max(warm_stats, key=lambda f: warm_stats[f][1])
        lib.mi_prof_enable(1 << dom if dom else 2 | 1)  # family 0 -> mask 1 (value 1 means "all", so add family 1)
        lib.mi_prof_reset()
    dt, ls = timed(args.steps)          # ---- THE timed region: exactly K steps ----
    losses += ls
    dom_stats = read_prof()[dom] if prof else None
    timings = tr.timings()

    fam_stats, dt_fam, dt_h2d = {}, None, None
    fam_serial, dt_serial = {}, None
    if prof and extras:
        lib.mi_prof_enable(1)
        lib.mi_prof_reset()
        dt_fam, _ = timed(args.steps)   # K further steps, every launch of every family bracketed
        fam_stats = read_prof()
        if not bf16 and world == 1:
            # ... and K more in the SERIAL schedule: in the default one the weight gradients run on a second stream beside the
            # next layer's BN', so a family's bracketed time there includes what it lost to its neighbour (HBM-bound next to
            # MFMA-bound: both slow down, the step gains ~1 %); this is the family on its own
            lib.mi_trainer_set_overlap(tr.t, 0)
            tr.step()
            lib.mi_prof_reset()
            dt_serial, _ = timed(args.steps)
            fam_serial = read_prof()
    lib.mi_prof_enable(0)
    if extras and world == 1:
        # K further steps with the reference's own data movement: the batch sits in pinned host memory and is copied over
        # PCIe at the top of every step (blocking, resnet.cu:1315-1316)
        import numpy as np
        tr.source_host(B.MI_LAYOUT_NCHW)
        b = tr.c_batch.contents
        rng = np.random.default_rng(1234)
        tr.fill_host_batch(rng.uniform(-124.0, 152.0, size=b.n_images * b.image_size).astype(np.float32),
                           rng.integers(0, dims["output"], size=b.n_images).astype(np.int32))
        tr.step()
        dt_h2d, _ = timed(args.steps)
    tr.close()
    return dict(dt=dt, losses=losses, dom=dom, dom_stats=dom_stats, timings=timings, fam_stats=fam_stats, dt_fam=dt_fam,
                fam_serial=fam_serial, dt_serial=dt_serial, dt_h2d=dt_h2d, act_bytes=act_bytes, dev_bytes=dev_bytes)


def roof_of(fam, stat, steps, bf16):
    FAMILIES = FAMILIES_BF16 if bf16 else FAMILIES_F32
    n, ms, fl, by = stat
    sec = max(ms, 1e-9) * 1e-3
    tf, gbs = fl / sec / 1e12, by / sec / 1e9
    # which roof binds the family: its arithmetic intensity against the machine balance of the pipe it runs on
    peak_tf = BF16_PEAK_TFLOPS if (bf16 and fam in (0, 1, 2, 5)) else FP32_PEAK_TFLOPS
    mfma_bound = fl > 0 and (fl / max(by, 1.0)) >= peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    r = {"kernel": FAMILIES[fam], "bound": "mfma" if mfma_bound else "hbm", "launches_per_step": round(n / steps, 1),
         "ms_per_step": round(ms / steps, 3)}
    if mfma_bound:
        r.update({"achieved": round(tf, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4), "algorithmic_GB_per_s": round(gbs, 1)})
    else:
        r.update({"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)})
        if fl > 0:
            r["algorithmic_TFLOP_per_s"] = round(tf, 2)
    return r


def roofline_of(leg, args, bf16):
    """the `roofline` object of a leg: the dominant family inside the timed region, plus every family from the extra steps"""
    dom, dom_stats = leg["dom"], leg["dom_stats"]
    if not dom_stats:
        return None
    n, ms, fl, by = dom_stats
    roof = roof_of(dom, dom_stats, args.steps, bf16)
    roof["launches"] = n
    roof["avg_launch_ms"] = round(ms / max(n, 1), 4)
    roof["algorithmic_gflop_per_launch"] = round(fl / max(n, 1) / 1e9, 3)
    roof["algorithmic_bytes_per_launch"] = round(by / max(n, 1))
    roof["traffic"] = None
    # HBM bytes of that family from the committed rocprofv3 PMC passes (bench.py cannot run the profiler on itself).  The file
    # names the sources it was measured on; a different hash now means the kernels changed since: the figure is then stale
    pmc_file = PMC_FILE_BF16 if bf16 else PMC_FILE_F32
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        # the PMC passes count kernel launches, this run counts logical launches: convert through bytes per step
        roof["traffic"] = round(pmc["families"][PMC_KEY[dom]]["hbm_GB_per_step_raw"] * 1e9 / (n / args.steps))
        roof["traffic_note"] = ("HBM bytes per launch = FETCH_SIZE + WRITE_SIZE of two separate rocprofv3 --pmc passes over this command "
                                "(profiles/%s, tools/pmc_traffic.sh); gathers narrower than 16 B per lane: gfx950 FETCH halving "
                                "uncalibrated, raw value; Infinity-Cache hits are counted" % pmc_file)
        roof["traffic_source_sha16"] = pmc.get("source_sha16")
        roof["current_source_sha16"] = source_sha16()
        roof["traffic_stale"] = roof["traffic_source_sha16"] != roof["current_source_sha16"]
        if roof["traffic_stale"]:
            roof["traffic_warning"] = ("resnet_amd/csrc changed since profiles/%s was measured (or the file predates the hash): `traffic` is "
                                       "that older build's figure -- re-run tools/pmc_traffic.sh" % pmc_file)
    except Exception:
        roof["traffic_note"] = "profiles/%s not present: no PMC traffic for this configuration" % pmc_file
    roof["note"] = ("fp32 peak 157.3 TFLOP/s is both the vector-FMA and the fp32-MFMA rate on gfx950; bf16 peak 2500 TFLOP/s dense; "
                    "`achieved` = algorithmic work of the family's launches / their HIP-event durations inside the timed region")
    fam_stats, fam_serial = leg["fam_stats"], leg["fam_serial"]
    if fam_stats:
        roof["families"] = [roof_of(f, fam_stats[f], args.steps, bf16) for f in sorted(fam_stats) if fam_stats[f][0] > 0]
        if fam_serial:
            roof["families_serial_schedule"] = [roof_of(f, fam_serial[f], args.steps, bf16) for f in sorted(fam_serial) if fam_serial[f][0] > 0]
            roof["families_serial_note"] = ("%d further steps with mi_trainer_set_overlap(0) (weight gradients on the compute stream, nothing runs "
                                            "beside anything): %.3f ms/step; the default schedule is the faster STEP, the serial one shows each "
                                            "family undisturbed" % (args.steps, leg["dt_serial"] / args.steps * 1e3))
        roof["families_note"] = ("%d further steps with every launch of every family bracketed by HIP events: %.3f ms/step "
                                 "(weight gradients run on a second stream next to batch norm, so family times overlap and do not add up to the step)"
                                 % (args.steps, leg["dt_fam"] / args.steps * 1e3))
    return roof


def workload(args, bf16):
    nice = "bf16 activations / fp32 accumulate" if bf16 else "fp32"
    return ("reference-defined ResNet-50 (47.58M params, 3x3-s2 projections, Adam, sum loss), %s, 224x224, full training step, batch %d per GPU%s"
            % (nice, args.batch, ", store policy RECOMPUTE_BN" if args.policy != "fast" else ""))


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            sys.exit(launch_workers(args, argv))  # no launcher around us: be it (before anything touches a GPU)
        args.gpus = world
    if args.launch_dry_run:
        sys.exit(launch_workers(args, argv))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it, and it must be there before HIP starts
    # stdout carries ONE JSON line and nothing else: gloo and RCCL print banners on fd 1 ("[Gloo] Rank 0 is connected ...", "RCCL version :
    # ...").  Everything written to fd 1 from here on goes to stderr; the record is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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

    bf16 = args.dtype == "bf16"
    leg = run_leg(args, lib, dist, rank, local_rank, world, bf16, extras=not args.no_extra)

    # fp32 runs carry BASELINE configs[4] in the same record: a second trainer with bf16 activations, its own warm-up and K timed
    # steps, after the fp32 trainer is gone.  One GPU: in a CHILD process (this one keeps its GPU context; a failure there cannot
    # take the fp32 line with it).  Data parallel: in process on every rank (the ranks are the launcher's).
    bf16_rec = None
    if not bf16 and not args.no_bf16:
        if world == 1 and dist is None:
            cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "bf16", "--steps", str(args.steps), "--warmup", str(args.warmup),
                   "--batch", str(args.batch), "--policy", args.policy, "--no-cpu-baseline", "--gpus", "1"]
            cmd += ["--no-prof"] if args.no_prof else []
            cmd += ["--no-extra"] if args.no_extra else []
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
                rec = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and rec:
                    d = json.loads(rec[-1])
                    bf16_rec = {k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "config", "roofline", "value_h2d_inclusive", "h2d_note") if k in d}
                    bf16_rec["how"] = "child process `bench.py --dtype bf16` with this run's --steps / --warmup / --batch, started after the fp32 trainer was destroyed"
                else:
                    bf16_rec = {"error": "bf16 leg exited with code %d: %s" % (r.returncode, (r.stderr or "")[-400:])}
            except Exception as e:  # noqa: BLE001
                bf16_rec = {"error": "bf16 leg: %r" % (e,)}
        else:
            try:
                bl = run_leg(args, lib, dist, rank, local_rank, world, True, extras=False)
                bf16_rec = {"metric": "images/sec ResNet-50 bf16 224x224 batch256", "value": round(args.batch * world * args.steps / bl["dt"], 2),
                            "unit": "images/sec", "ms_per_step": round(bl["dt"] / args.steps * 1e3, 3), "dtype": "bf16",
                            "config": {"workload": workload(args, True), "global_batch": args.batch * world, "parallelism": "dp%d" % world},
                            "how": "in process on every rank after the fp32 trainer was destroyed; same barrier / max-over-ranks timing"}
                r = roofline_of(bl, args, True)
                if r:
                    bf16_rec["roofline"] = r
            except Exception as e:  # noqa: BLE001
                bf16_rec = {"error": "bf16 leg: %r" % (e,)}

    if rank == 0:
        dt, losses, timings = leg["dt"], leg["losses"], leg["timings"]
        ms_step = dt / args.steps * 1e3
        value = args.batch * world * args.steps / dt
        out = {"metric": "images/sec ResNet-50 %s 224x224 batch256" % ("bf16" if bf16 else "fp32"), "value": round(value, 2), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
               "config": {"workload": workload(args, bf16),
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "phase_ms_last_step": {"forward": round(timings[1], 3), "backward": round(timings[2], 3),
                                                 "update": round(timings[3], 3)},
                          "final_loss_per_image": round(losses[-1] / args.batch, 4),
                          "activation_bytes_kept_for_backward": leg["act_bytes"], "device_bytes": leg["dev_bytes"]}}
        if leg["dt_h2d"] is not None:
            out["value_h2d_inclusive"] = round(args.batch * world * args.steps / leg["dt_h2d"], 2)
            out["h2d_note"] = ("%d further steps with the batch in pinned host memory, copied over PCIe at the top of every step "
                               "(blocking, as resnet.cu:1315-1316): %.3f ms/step; never `value`" % (args.steps, leg["dt_h2d"] / args.steps * 1e3))
        roof = roofline_of(leg, args, bf16)
        if roof:
            out["roofline"] = roof
        if bf16_rec is not None:
            out["bf16"] = bf16_rec
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
