#!/usr/bin/env python3
"""Throughput of the supervised conv-autoencoder train step (BASELINE.json metric) on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one full iteration of the reference's batch loop (R.md:646-654): forward, alpha*MSE + CE, backward, Adam, on a
device-resident synthetic batch of EuroSAT-shaped 64x64 RGB images (BASELINE.json configs[2]: batch 512 per GPU, joint
AE + classification head, 64-d latent, bf16 storage / fp32 accumulation).  With N > 1 every rank holds a replica and a
512-image shard of the global batch; gradients are averaged with one RCCL all-reduce per step (weak scaling).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 512
ALPHA, LR = 35.0, 5e-3          # the reference's best grid point (R.md:2407)
# algorithmic work per image (SURVEY.md 8d / BASELINE.md section 2)
FLOP_PER_IMG_TRAIN = 181.92e6
BYTES_PER_IMG_BF16 = 1.354e6
BYTES_PER_STEP_WEIGHTS = 44.7e6
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
MFMA_FP8_PEAK_TFLOPS = 5000.0   # dense fp8


def make_batch(b, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.rand((b, 3, 64, 64), generator=g, dtype=torch.float32)
    y = torch.randint(0, 10, (b,), generator=g, dtype=torch.int64)
    return x.to(device), y.to(device)


def host_cores():
    """(cores this job may use, logical CPUs visible, cgroup quota in cores or None).  The GPU boxes expose every logical CPU of
    the host to a 1-GPU job but give it a CPU-time quota (cgroup cpu.max) of a 16-core share; an OpenMP pool wider than the quota
    makes the torch CPU ops many times slower, so the thread count is min(visible, quota)."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    cores = visible if quota is None else max(1, min(visible, int(quota + 0.5)))
    return cores, visible, quota


def cpu_baseline(seconds_budget=10.0):
    """The CPU restatement (oracle/ae_torch_cpu.py, same module graph as the reference notebook, fp32, torch CPU ops on the
    host cores this job may use) timed on bounded samples of the same train step: B=32 (BASELINE configs[0]) and B=64 (the
    notebook's batch size, R.md:246), SURVEY.md 8d.  `value` is the B=64 leg."""
    from oracle import ae_torch_cpu as T
    cores, visible, quota = host_cores()
    cores = int(os.environ.get("EAE_CPU_THREADS", cores))
    torch.set_num_threads(cores)
    legs = {}
    for b in (32, 64):
        model = T.build(latent_dim=64, seed=0)
        opt = T.make_adam(model, LR)
        g = torch.Generator().manual_seed(1234)
        x = torch.rand((b, 3, 64, 64), generator=g)
        y = torch.randint(0, 10, (b,), generator=g)
        T.train_step(model, opt, x, y, ALPHA)       # warm-up (allocations, thread pool)
        n, t0 = 0, time.perf_counter()
        while True:
            T.train_step(model, opt, x, y, ALPHA)
            n += 1
            el = time.perf_counter() - t0
            if el > seconds_budget or n >= 5000:
                break
        legs[f"b{b}"] = {"images_per_s": round(b * n / el, 1), "ms_per_step": round(1e3 * el / n, 2), "steps": n}
    return {"value": legs["b64"]["images_per_s"], "unit": "images/s", "cores": cores, "kind": "port",
            "host_cpus_visible": visible, "cgroup_quota_cores": quota,
            "sample": f"{legs['b64']['steps']} train steps of batch 64 and {legs['b32']['steps']} of batch 32 (fp32, torch CPU ops, "
                      f"{cores} threads, oracle/ae_torch_cpu.py)",
            "legs": legs}


def kernel_roofline(eng, step_fn, batch, tag="b512", H=64, W=64, mfma_peak=None, steps=32):
    """Per-launch duration of the DOMINANT kernel of a workload -- the one with the largest total duration in the newest committed
    rocprofv3 summary of that workload under profiles/ -- measured live with HIP events on the stream it is launched on inside real
    train steps, priced against its bounding roofline with its own ALGORITHMIC bytes."""
    from eae_amd import profile_hooks as PH
    return PH.dominant_kernel_roofline(eng, step_fn, batch, HBM_PEAK_GBS, mfma_peak or MFMA_BF16_PEAK_TFLOPS, steps=steps, tag=tag, H=H, W=W)


def time_steps(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


C5_WORKLOAD = ("BASELINE configs[4], one GPU's share: 256x256x3 inputs, 256-d latent, joint train step; fp8 = e4m3 weights / "
               "activations, e5m2 gradients on v_mfma_f32_16x16x32_{fp8,bf8}_{fp8,bf8} for the six 3x3 layers (delayed scaling), "
               "bf16 = the same shape through the bf16 kernels")


def config5_engine(quant, batch=128):
    import eae_amd
    from eae_amd.engine import AEEngine
    g = torch.Generator(device="cuda")
    g.manual_seed(4321)
    x = torch.rand((batch, 3, 256, 256), generator=g, device="cuda")
    y = torch.randint(0, 10, (batch,), generator=g, device="cuda")
    torch.manual_seed(0)
    m = eae_amd.SupervisedAutoencoder(latent_dim=256, num_classes=10, image_size=256).cuda().train()
    eng = AEEngine(m, max_batch=batch, quant=quant)
    if quant == "fp8":
        eng.fp8_calibrate(x, y, ALPHA)
    return m, eng, (lambda: eng.train_step(x, y, ALPHA, LR))


def config5_leg(batch=128, steps=20, warmup=5, roofline=True):
    res = {"workload": C5_WORKLOAD, "per_gpu_batch": batch}
    for quant in ("fp8", "bf16"):
        m, eng, step = config5_engine(quant, batch)
        sec = time_steps(step, steps, warmup)
        res[quant] = {"ms_per_step": round(1e3 * sec, 3), "images_per_s": round(batch / sec, 1),
                      "final_loss": round(float(eng.loss_last.cpu()[0]), 4)}
        if roofline:
            try:
                res[quant]["roofline"] = kernel_roofline(eng, step, batch, tag=f"c5{quant}", H=256, W=256, steps=12,
                                                         mfma_peak=MFMA_FP8_PEAK_TFLOPS if quant == "fp8" else MFMA_BF16_PEAK_TFLOPS)
            except Exception as e:
                res[quant]["roofline"] = {"error": str(e)[:200]}
        del eng, m, step
        torch.cuda.empty_cache()
    return res


def grid_b64_leg(k=int(os.environ.get("EAE_GRID_K", "8")), steps=150, warmup=15):
    """The reference's REAL workload (R.md:246, 599-711): batch 64, a grid of independent configurations.  `grouped_one`: K engine
    contexts stepped by ONE sequence of grouped launches per step (include/eae.h eae_group_train_step); `grouped`: two such groups at a
    time from two host threads, one side stream per context -- what grid_search_autoencoder(grouped=K, concurrent_groups=2) does;
    `concurrent`: the round-3 way, 4 contexts on one stream + host thread each (train.run_concurrent); `k1`: one configuration.
    `images_per_s` is the aggregate over the 2 x K configurations of the `grouped` leg."""
    import eae_amd
    from eae_amd.engine import AEEngine, engine_for
    from eae_amd import train as T
    res = {"workload": "BASELINE configs[2]'s step at the notebook's batch size 64 (R.md:246): K independent (alpha, lr) configurations of "
                       "the grid R.md:599-711 trained at the same time on one GPU", "batch": 64}
    x, y = make_batch(64, torch.device("cuda"), seed=4321)

    def build(n, single_stream):
        engs = []
        for i in range(n):
            torch.manual_seed(100 + i)
            m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
            if single_stream:
                m._eae_side_streams = -1          # what grid_search_autoencoder(concurrent >= 3) does: one stream per context
            engs.append((m, engine_for(m, max_batch=64)))
        return engs

    def grouped_leg(ngroups, side_streams):
        groups = []
        for g in range(ngroups):
            es = []
            for i in range(k):
                torch.manual_seed(100 + g * k + i)
                m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).cuda().train()
                if side_streams:
                    m._eae_side_streams = side_streams
                es.append((m, engine_for(m, max_batch=64)))
            groups.append(es)

        def gjob(es, n):
            engs = [e for _, e in es]
            a = ([x] * k, [y] * k, [ALPHA + i for i in range(k)], [1e-3] * k)

            def job():
                for _ in range(n):
                    AEEngine.group_train_step(engs, *a)
            return job
        if ngroups == 1:
            gjob(groups[0], warmup)()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gjob(groups[0], steps)()
        else:
            T.run_concurrent([gjob(es, warmup) for es in groups], ngroups, static=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            T.run_concurrent([gjob(es, steps) for es in groups], ngroups, static=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out = {"groups": ngroups, "configs": ngroups * k, "steps_each": steps, "images_per_s": round(ngroups * k * steps * 64 / el, 1),
               "ms_per_round_of_group_steps": round(1e3 * el / steps, 4), "side_streams_per_context": side_streams,
               "gate_timeouts": sum(1 for es in groups for _, e in es if e.gate_timeouts())}
        del groups
        gc.collect()                # (engine contexts sit in reference cycles: destroy them, and their claims on hardware queues, now)
        torch.cuda.empty_cache()
        return out
    # what grid_search_autoencoder(grouped=K, concurrent_groups=2) does: two groups at a time from two host threads, one side stream per
    # context (2 x 2 streams = the four hardware queues; one group's forward beside the other's backward); then one group of K with
    # the default stream layout
    res["grouped"] = grouped_leg(2, 1)
    res["grouped_one"] = grouped_leg(1, 2)
    for name, kk in (("k1", 1), ("concurrent", 4)):
        engs = build(kk, kk >= 3)

        def job_of(e, n):
            def job():
                for _ in range(n):
                    e.train_step(x, y, ALPHA, 1e-3)
            return job
        T.run_concurrent([job_of(e, warmup) for _, e in engs], kk, static=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        T.run_concurrent([job_of(e, steps) for _, e in engs], kk, static=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        bad = [e.gate_timeouts() for _, e in engs]
        res[name] = {"configs": kk, "steps_each": steps, "images_per_s": round(kk * steps * 64 / el, 1),
                     "ms_per_step_per_config": round(1e3 * el / steps, 4), "gate_timeouts": sum(1 for b in bad if b)}
        del engs
        gc.collect()
        torch.cuda.empty_cache()
    res["images_per_s"] = res["grouped"]["images_per_s"]
    return res


C2_WORKLOAD = "BASELINE configs[1]: batch 256, encoder+decoder reconstruction (MSE) only, bf16"


def side_workload(args, device):
    """`--workload c2 | c5fp8 | c5bf16`: the SAME timed loop over one of the other single-GPU BASELINE configurations, so that
    tools/profile_round.sh can put rocprofv3 around it (kernel stats + PMC passes per workload) and the line carries its roofline."""
    import eae_amd
    from eae_amd.engine import engine_for
    if args.workload == "grid8":
        # the notebook's own batch size, eight grid configurations stepped as one group (configs.grid_b64's grouped leg, alone, for rocprofv3)
        from eae_amd.engine import AEEngine
        engs = []
        for i in range(8):
            torch.manual_seed(100 + i)
            m = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).to(device).train()
            m._eae_side_streams = 2               # what train.fit_autoencoder_group builds (a grouped step is fastest with two)
            engs.append((m, engine_for(m, max_batch=64)))
        es = [e for _, e in engs]
        x, y = make_batch(64, device, seed=4321)
        xs, ys, al, lr = [x] * 8, [y] * 8, [ALPHA + i for i in range(8)], [LR] * 8
        sec = time_steps(lambda: AEEngine.group_train_step(es, xs, ys, al, lr), args.steps, args.warmup)
        if any(e.gate_timeouts() for e in es):
            raise SystemExit("a gate time-out in the timed region")
        print(json.dumps({"metric": "EuroSAT 64x64 RGB images/sec (AE+MLP train step)", "value": round(8 * 64 / sec, 1), "unit": "images/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * sec, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": "BASELINE configs[2]'s step at the notebook's batch size 64, 8 grid configurations per grouped step "
                                                 "(eae_group_train_step)", "per_gpu_batch": 512, "global_batch": 512, "parallelism": "single",
                                     "note": "NOT the headline configuration; one 'step' = one grouped step of 8 x 64 images"}}), flush=True)
        return
    if args.workload == "c2":
        batch = 256 if args.batch == BATCH else args.batch
        torch.manual_seed(0)
        model = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).to(device)
        eng = engine_for(model, max_batch=batch)
        x, y = make_batch(batch, device, seed=1234)
        step = lambda: eng.train_step(x, y, 1.0, LR, head=False)
        desc, tag, hw, dtype, peak = C2_WORKLOAD, "c2", 64, "bf16", MFMA_BF16_PEAK_TFLOPS
    else:
        quant = args.workload[2:]
        batch = 128 if args.batch == BATCH else args.batch
        model, eng, step = config5_engine(quant, batch)
        desc, tag, hw, dtype = C5_WORKLOAD + f" [{quant} leg]", f"c5{quant}", 256, quant
        peak = MFMA_FP8_PEAK_TFLOPS if quant == "fp8" else MFMA_BF16_PEAK_TFLOPS
    sec = time_steps(step, args.steps, args.warmup)
    loss = float(eng.loss_last[0].item())
    if not np.isfinite(loss) or eng.gate_timeouts():
        raise SystemExit("non-finite loss or a gate time-out in the timed region")
    out = {"metric": "EuroSAT 64x64 RGB images/sec (AE+MLP train step)", "value": round(batch / sec, 1), "unit": "images/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * sec, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": dtype, "data": "synthetic",
           "config": {"workload": desc, "per_gpu_batch": batch, "global_batch": batch, "parallelism": "single", "final_loss": round(loss, 4),
                      "note": "NOT the headline configuration (that is the default run, configs[2]); images are %dx%d here" % (hw, hw)}}
    if not args.no_roofline:
        try:
            out["roofline"] = kernel_roofline(eng, step, batch, tag=tag, H=hw, W=hw, mfma_peak=peak, steps=32 if hw == 64 else 12)
        except Exception as e:
            out["roofline"] = {"error": str(e)[:200]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra BASELINE configs[1] measurement")
    ap.add_argument("--workload", choices=("c3", "c2", "c5fp8", "c5bf16", "grid8"), default="c3",
                    help="c3 = BASELINE configs[2], the headline (default); the others run the same loop over configs[1] / configs[4]'s shape")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    if args.workload != "c3":
        if world > 1:
            raise SystemExit("--workload other than c3 is a single-GPU measurement")
        return side_workload(args, device)

    import eae_amd
    from eae_amd.engine import engine_for
    from eae_amd import dp

    dist_on = world > 1 or os.environ.get("EAE_FORCE_DP") == "1"     # EAE_FORCE_DP: exercise the DP code path with one rank
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # no device_id: eager communicator binding slowed every kernel launch sequence by ~20 % on this stack (measured)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)

    torch.manual_seed(0)
    model = eae_amd.SupervisedAutoencoder(latent_dim=64, num_classes=10).to(device)
    eng = engine_for(model, max_batch=args.batch)
    trainer = dp.DataParallelTrainer(eng) if dist_on else None
    if trainer is not None:
        trainer.broadcast_parameters()
    x, y = make_batch(args.batch, device, seed=1234 + rank)

    def step():
        if trainer is None:
            eng.train_step(x, y, ALPHA, LR)
        else:
            trainer.train_step(x, y, ALPHA, LR)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier(device_ids=[local_rank])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([el], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss = float(eng.loss_last[0].item())
    if not np.isfinite(loss):
        raise SystemExit("non-finite loss in the timed region")
    if eng.gate_timeouts():
        raise SystemExit("a side-stream gate timed out in the timed region (results invalid)")

    if rank == 0:
        total_images = args.batch * world * args.steps
        value = total_images / el
        ms = 1e3 * el / args.steps
        out = {
            "metric": "EuroSAT 64x64 RGB images/sec (AE+MLP train step)",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: joint conv-AE + classification head train step (fwd, 35*MSE+CE, bwd, Adam), "
                                   "64x64x3 inputs, 64-d latent, bf16 storage / fp32 accumulate",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}" if world > 1 else "single", "final_loss": round(loss, 4)},
        }
        if trainer is not None:
            # ranks of the RCCL communicator the ENGINE owns (eae_dp_init): the driver's SCALE record can confirm that N ranks exchanged
            out["rccl_ranks"] = trainer.rccl_ranks()
            out["config"]["dp_exchange"] = ("engine-owned RCCL communicator, decoder-side bucket overlapped with the encoder backward"
                                            if trainer.native and os.environ.get("EAE_DP_OVERLAP", "0") == "1" else
                                            "engine-owned RCCL communicator, one all-reduce after the backward" if trainer.native else
                                            "torch.distributed all-reduce")
        # step-level roofline context
        step_bytes = args.batch * BYTES_PER_IMG_BF16 + BYTES_PER_STEP_WEIGHTS
        out["step_roofline"] = {"hbm_floor_us": round(step_bytes / (HBM_PEAK_GBS * 1e3), 1),
                                "mfma_floor_us": round(args.batch * FLOP_PER_IMG_TRAIN / (MFMA_BF16_PEAK_TFLOPS * 1e6), 1),
                                "frac_of_hbm_floor": round(step_bytes / (HBM_PEAK_GBS * 1e3) / (ms * 1e3), 4)}
        if not args.no_roofline:
            try:
                # local steps only: the other ranks have left the loop, so no collective may be issued from here
                out["roofline"] = kernel_roofline(eng, lambda: eng.train_step(x, y, ALPHA, LR), args.batch)
            except Exception as e:  # the headline number must still be printed
                out["roofline"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_configs:
            # the other single-GPU BASELINE config, measured the same way (not the headline: parity-tested in tests/):
            # configs[1] = batch 256, encoder + decoder reconstruction (MSE) only
            try:
                x2, y2 = x[:256].contiguous(), y[:256].contiguous()
                sec = time_steps(lambda: eng.train_step(x2, y2, 1.0, LR, head=False), min(args.steps, 200), min(args.warmup, 20))
                b2 = 256 * (BYTES_PER_IMG_BF16 - 0.0) + BYTES_PER_STEP_WEIGHTS
                out["configs"] = {"c2": {"workload": C2_WORKLOAD,
                                         "ms_per_step": round(1e3 * sec, 4), "images_per_s": round(256 / sec, 1),
                                         "hbm_floor_us": round(b2 / (HBM_PEAK_GBS * 1e3), 1),
                                         "frac_of_hbm_floor": round(b2 / (HBM_PEAK_GBS * 1e3) / (sec * 1e6), 4)}}
                if not args.no_roofline:
                    out["configs"]["c2"]["roofline"] = kernel_roofline(eng, lambda: eng.train_step(x2, y2, 1.0, LR, head=False), 256, tag="c2")
            except Exception as e:
                out["configs"] = {"error": str(e)[:200]}
            # configs[4]'s per-GPU shape and arithmetic (its 8-GPU form is the driver's): 256x256 inputs, 256-d latent, fp8 operands for
            # the GEMMs of the six 3x3 layers; the same shape in bf16 beside it.  Not the headline; parity in tests/test_gpu_fp8.py.
            try:
                out.setdefault("configs", {})["c5"] = config5_leg(roofline=not args.no_roofline)
            except Exception as e:
                out.setdefault("configs", {})["c5"] = {"error": str(e)[:200]}
            try:
                out.setdefault("configs", {})["grid_b64"] = grid_b64_leg()
                out["configs"]["grid_b64"]["vs_b512_single_model"] = round(out["configs"]["grid_b64"]["images_per_s"] / value, 3)
            except Exception as e:
                out.setdefault("configs", {})["grid_b64"] = {"error": str(e)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:
                out["cpu_baseline"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier(device_ids=[local_rank])      # every rank leaves together (rank 0 ran its local roofline pass meanwhile)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
