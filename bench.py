#!/usr/bin/env python3
"""bench.py -- ray-samples/sec of a train step of the Contrastive-Lift render hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Step = one full ``training_step`` of the reference (trainer/train_panopli_tensorf.py:148-228) at steady state:
main pass over 4096 rays (forward of all heads, MSE + TV + confidence-weighted CE + dist-reg, backward, Adam)
followed by the instance pass over 1024 rays (EMA, forward_instance_feature, slow-fast loss, backward, Adam).
Workload = BASELINE.json configs[1] ("ScanNet scene0423_02, 4096 rays/batch, fp32") as a synthetic stand-in of
the same shapes (no dataset offline): C = 22 classes, E = 3 (D = 6), grid 128^3 => S = 440 samples/ray.
Unit = nominal ray-sample (rays x S); value = world * (4096 + 1024) * S / step time, weak scaling
(every rank renders its own rays; one RCCL all-reduce of the gradient arena range per backward).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PMC_RECORD = "profiles/r03_pmc_k_layer_f32.json"
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_FP32_MFMA_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--inst-rays", type=int, default=1024)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--classes", type=int, default=22)
    ap.add_argument("--chunk", type=int, default=0, help="rays per renderer call; 0 = whole batch (reference: 2048)")
    ap.add_argument("--lean", action="store_true", help="skip the instance heads in the main pass (their output is discarded)")
    ap.add_argument("--dtype", choices=["fp32", "bf16", "fp32x6"], default="fp32",
                    help="MLP operand precision: fp32 (headline, BASELINE configs[1]) or bf16 operands / fp32 accumulate (configs[2])")
    ap.add_argument("--inference-probe", action="store_true",
                    help="also report frame-render throughput and the lean main-pass time (both change the launch mix of the dominant "
                         "kernel: keep them out of profiled runs)")
    ap.add_argument("--global-rays", type=int, default=0,
                    help="strong scaling (BASELINE configs[3]: 8192 rays per step over the job): rays per GPU = global / world, "
                         "instance rays stay per-GPU (one instance image per rank, as the reference's DDP)")
    ap.add_argument("--nosync", action="store_true",
                    help="sync-free steps: the active-sample count is never read back; buffers sized by a learnt capacity, kernels clamp to the "
                         "device-side count (exact fp32 only)")
    ap.add_argument("--inference-sharded", action="store_true",
                    help="BASELINE configs[4]: time full 1296x968 frames through inference.render_rays_sharded (row tiles over the ranks, one "
                         "all-gather per frame) instead of training steps; a 'step' is one frame, value = rays/s over the job")
    ap.add_argument("--launch-check", action="store_true",
                    help="only launch the ranks, form the process group, all-reduce one number and print the line's n_gpus / rccl_ranks "
                         "fields (no render work; runs without a GPU over gloo)")
    ap.add_argument("--no-extras", action="store_true", help="skip the bf16-mode and frame-render extras measured after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    return ap.parse_args()


def self_launch(a):
    """``python bench.py --gpus N`` without a launcher: re-exec this command under torch.distributed.run with N ranks on this node
    (one per GPU, RCCL).  On a box with fewer devices than ranks (the 1-GPU test box) the ranks share devices and the process group
    falls back to gloo -- the line then says so (`dist_backend`, `rccl_ranks: 0`)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < a.gpus:
        env.setdefault("CLIFT_DIST_BACKEND", "gloo")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(a, world, rank):
    """The launcher path alone: process group + one collective, no render work (CPU-runnable)."""
    backend = os.environ.get("CLIFT_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    total = 1.0
    if world > 1:
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            t = torch.ones(1, device="cuda")
        else:
            dist.init_process_group(backend)
            t = torch.ones(1)
        dist.all_reduce(t)
        total = float(t.item())
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "requested_gpus": a.gpus, "dist_backend": backend if world > 1 else None,
                          "rccl_ranks": (dist.get_world_size() if (world > 1 and backend == "nccl") else 0), "allreduce_of_ones": total}))
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.launch_check:
        return launch_check(a, world, rank)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the render path has no CPU fallback")
    local = local % torch.cuda.device_count()       # (lets a 2-rank gloo smoke test share one GPU; a no-op on a real node)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        backend = os.environ.get("CLIFT_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl" and torch.cuda.device_count() < world:
            backend = "gloo"                                        # ranks share a device (1-GPU test box): RCCL wants one device per rank
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if a.global_rays:
        a.rays = max(1, a.global_rays // world)
    import contrastive_lift_amd as cl
    from contrastive_lift_amd import engine, synthetic
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config

    model, renderer, pool = synthetic.make_scene(grid=a.grid, num_classes=a.classes, max_instances=3, seed=0, device=dev)
    S = int(renderer.n_samples)
    if a.inference_sharded:
        return bench_inference_sharded(a, model, renderer, dev, world, rank, backend)
    cfg = default_config(chunk=a.chunk, instance_optimization_epoch=0, late_semantic_optimization=0, mlp_dtype=a.dtype, nosync=a.nosync)
    tr = HotPathTrainer(model, renderer, cfg, current_epoch=4)
    n_batches = 4
    batches = [synthetic.make_batches(pool, a.rays, a.inst_rays, a.classes, 25, seed=100 + rank * 17 + i, device=dev) for i in range(n_batches)]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Everything the measurement itself uses for the first time -- a device-to-host copy (the generator state in the snapshot), timing
    # events, their read-back -- is used once BEFORE the warm-up: the runtime creates queues / signal pools lazily, and in the first GPU
    # process on a fresh box that one-off cost (44-51 ms) landed in the first timed step.
    _ = tr._rng_state()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record(); torch.zeros(8, device=dev).cpu(); ev[1].record(); torch.cuda.synchronize(); _ = ev[0].elapsed_time(ev[1])
    if world > 1 and tr.overlap_allreduce == "auto":
        # data-parallel: the trainer decides by measurement whether the asynchronous exchange (early range all-reduced under the density
        # backward, CUs reserved for RCCL) beats the synchronous one -- its 2 + 2 x 4 calibration passes synchronise the device, so
        # they run here, before the warm-up
        for i in range(tr.CAL_WARMUP + 2 * tr.CAL_STEPS):
            tr.training_step(batches[i % n_batches], lean=a.lean)
    for i in range(a.warmup):
        tr.training_step(batches[i % n_batches], lean=a.lean)
    # snapshot of the training state at the start of the timed region (parameters + both Adam states): the roofline pass below
    # replays exactly these steps with per-launch events
    snap = (model.param_flat.clone(), tr.opt_main.state_dict(), tr.opt_inst.state_dict(), tr._rng_state())
    sync_all()
    # one event per step boundary (recorded, never waited on inside the loop): the per-step GPU times afterwards show whether the K steps were
    # uniform or whether one of them carried a one-off stall (seen: ~40 ms once in the first GPU process on a fresh box)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(a.steps):
        tr.training_step(batches[i % n_batches], lean=a.lean)
        marks[i + 1].record()
    sync_all()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    if os.environ.get("CLIFT_BENCH_DUMP"):
        print("[steps] " + " ".join(f"{x:.2f}" for x in step_ms), file=sys.stderr)
    step_ms.sort()
    # ---- roofline pass: the SAME steps again from the same state (the active-sample count drifts while the field trains, so
    # any other step would see a different launch mix), every clift_gemm launch bracketed by two HIP events on its launch
    # stream (torch's current stream; the side-stream mode is off by default).  Kept out of the timed region because creating
    # and recording ~80 events per step makes the step host-bound (+20 %); kernel durations are unaffected by that.
    real_gemm, real_first2, real_last2, real_app_last2, real_first2_bwd, real_first2_wgrad = (engine.gemm, engine.first2, engine.last2, engine.app_last2,
                                                                                              engine.first2_bwd, engine.first2_wgrad)
    real_first2_x6 = engine.first2_x6

    def replay(select):
        """Re-run the timed steps from the snapshot with the selected matrix-core launches bracketed by HIP events.  Two launch sites:
        engine.gemm (clift_gemm) and engine.first2 (clift_xyz_head_first2_fwd: the K = 3 layer generated inside the persistent kernel
        of the first 256 x 256 layer -- kind "fwd_gen", FLOPs of both layers)."""
        def restore():
            model.param_flat.copy_(snap[0])
            tr.opt_main.load_state_dict(snap[1])
            tr.opt_inst.load_state_dict(snap[2])
            tr.load_rng_state(snap[3])          # the per-ray jitter: same samples, same launch sizes
        restore()
        if os.environ.get("CLIFT_BENCH_PRESTEP"):
            tr.training_step(batches[0], lean=a.lean)
            sync_all()
            restore()
        out = []

        def bracket(kind, M, N, K, extra_flops, fn):
            if not select(kind, N):
                return fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            out.append((kind, M, N, K, e0, e1, extra_flops))

        def recorded_gemm(M, N, K, *args, **kw):
            kind = "wgrad" if kw.get("a_trans") else "dgrad" if kw.get("b_trans") else "fwd"
            return bracket(kind, M, N, K, 0.0, lambda: real_gemm(M, N, K, *args, **kw))

        def recorded_first2(M, *args):
            return bracket("fwd_gen", M, 256, 256, 2.0 * M * 256 * 3, lambda: real_first2(M, *args))

        def recorded_last2(M, h, W, b, Wo, *args):        # last hidden layer + narrow output layer (clift_xyz_head_last2_fwd)
            return bracket("fwd_out", M, 256, 256, 2.0 * M * 256 * Wo.shape[0], lambda: real_last2(M, h, W, b, Wo, *args))
        def recorded_app_last2(M, H1, W2, b2, W3, *args):  # appearance: last hidden layer + output layer + sigmoid (clift_app_head_last2_fwd)
            return bracket("fwd_out", M, 128, 128, 2.0 * M * 128 * W3.shape[0], lambda: real_app_last2(M, H1, W2, b2, W3, *args))
        def recorded_first2_bwd(M, *args):                 # second layer's masked dgrad + the K = 3 layer's weight gradient (clift_xyz_head_first2_bwd)
            return bracket("dgrad", M, 256, 256, 2.0 * M * 256 * 4, lambda: real_first2_bwd(M, *args))
        def recorded_first2_wgrad(M, *args):               # second layer's weight gradient over the regenerated first activation (clift_xyz_head_first2_wgrad)
            return bracket("wgrad", 256, 256, M, 2.0 * M * 256 * 3, lambda: real_first2_wgrad(M, *args))
        engine.gemm, engine.first2, engine.last2, engine.app_last2 = recorded_gemm, recorded_first2, recorded_last2, recorded_app_last2
        engine.first2_bwd, engine.first2_wgrad = recorded_first2_bwd, recorded_first2_wgrad

        def recorded_first2_x6(M, *args):                  # fp32x6 mode: the same fusion on the split kernel (clift_xyz_head_first2_x6_fwd)
            return bracket("fwd_gen", M, 256, 256, 2.0 * M * 256 * 3, lambda: real_first2_x6(M, *args))
        engine.first2_x6 = recorded_first2_x6
        try:
            for i in range(a.steps):
                tr.training_step(batches[i % n_batches], lean=a.lean)
            sync_all()
        finally:
            engine.gemm, engine.first2, engine.last2, engine.app_last2 = real_gemm, real_first2, real_last2, real_app_last2
            engine.first2_bwd, engine.first2_wgrad, engine.first2_x6 = real_first2_bwd, real_first2_wgrad, real_first2_x6
        return out
    # pass 1: only the dominant kernel's launches (k_layer_f32 forward: the 11 256 x 256 forward layers of a step, in its three
    # instantiations -- plain, K = 3 input generated in-kernel, narrow output layer fused) -- few enough events that the step stays
    # GPU-bound, so an event pair measures the kernel and not a host gap; pass 2: every matrix-core launch, for the informational all_gemm
    # split (the ~60 events per step make that pass host-bound, so its per-launch times are upper bounds).
    # The pass runs TWICE and each launch keeps the shorter of its two measurements: an event pair also contains any moment the GPU
    # sat idle between the two records, and a single host hiccup (seen: 40 ms inside one bracket of the first process on a fresh box)
    # would otherwise pass for kernel time.  The replays are deterministic (same state, same batches), so launch i is the same work.
    def resolve(recs):
        return [(kind, M, N, K, e0.elapsed_time(e1), xf) for kind, M, N, K, e0, e1, xf in recs]
    dom_sel = lambda kind, N: kind in ("fwd", "fwd_gen", "fwd_out") and N > 128
    rec, rec_b = resolve(replay(dom_sel)), resolve(replay(dom_sel))
    # (same launch = same kind and shape at the same position; the row count may differ by a few samples between replays -- the gradient
    # atomics are not order-deterministic, so a later step's active-sample count can move by one or two)
    same = lambda x, y: x[0] == y[0] and x[2:4] == y[2:4] and abs(x[1] - y[1]) <= 0.01 * max(x[1], y[1])
    if len(rec) == len(rec_b) and all(same(x, y) for x, y in zip(rec, rec_b)):
        rec = [x[:4] + (min(x[4], y[4]),) + x[5:] for x, y in zip(rec, rec_b)]
    rec_all = resolve(replay(lambda kind, N: True))
    # every bracket is an UPPER bound of its kernel's time (it also contains any moment the GPU idled between the two records), and the third
    # replay bracketed the same launches once more: keep the shortest of the three per launch (seen once: all OUTV brackets of both dominant
    # passes 3x too long in a run right behind a 20-minute test session -- 0.45 instead of 0.77 -- while the all-launch pass read 111 TFLOP/s)
    sub = [x for x in rec_all if dom_sel(x[0], x[2])]
    if len(sub) == len(rec) and all(same(x, y) for x, y in zip(rec, sub)):
        rec = [x[:4] + (min(x[4], y[4]),) + x[5:] for x, y in zip(rec, sub)]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_step = dt / a.steps * 1e3
    samples_step = world * (a.rays + a.inst_rays) * S
    value = samples_step / (dt / a.steps)

    extra = {}
    roof = None
    cpu = None
    if rank == 0:
        roof = roofline(rec, rec_all, a.steps, engine, a.dtype, ms_step)
    if rank == 0 and world == 1:
        # ---- per-pass split and sample statistics (outside the timed region)
        def timed(fn, n=n_batches):          # cycles through the same batches as the timed loop
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(n):
                fn(batches[i % n_batches])
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n
        t_main = timed(lambda b: tr.main_pass(b[0], lean=a.lean))
        t_inst = timed(lambda b: tr.instance_pass(b[1]))
        ctxs = tr.main_pass(batches[0][0], lean=a.lean)
        M = sum((int(c.ray_start[-1]) if getattr(c, "capped", False) else c.M) for c in ctxs)
        inbox = sum(int((c.alpha > 0).sum()) for c in ctxs)
        extra = dict(main_pass_ms=round(t_main * 1e3, 3), instance_pass_ms=round(t_inst * 1e3, 3),
                     main_pass_samples_per_s=a.rays * S / t_main, instance_pass_samples_per_s=a.inst_rays * S / t_inst,
                     f_active=M / (a.rays * S), f_inbox_alpha_gt0=inbox / (a.rays * S), samples_per_ray=S)
        extra["active_samples_per_s_main_pass"] = M / t_main            # cost follows the ACTIVE samples; comparable across scenes
        if not a.no_extras and a.dtype == "fp32":
            # the reference's main pass computes the instance heads and discards their output (T:155); without that dead work:
            t_lean = timed(lambda b: tr.main_pass(b[0], lean=True))
            extra["lean_main_pass_ms"] = round(t_lean * 1e3, 3)
            extra["lean_step_ms_estimate"] = round((t_lean + t_inst) * 1e3, 3)
            # BASELINE configs[2] (bf16 MLP operands) and configs[4] (frame render) on the same scene, AFTER the timed fp32 region, so
            # that the driver-run record carries them: same step definition, 10 steps after 3 warm-up steps / one 262144-ray tile
            extra.update(inference_probe(cl, model, renderer, pool))
            extra.update(bf16_probe(a, dev, batches, S))
            extra.update(x6_probe(a, dev, batches, S))
            extra.update(small_batch_probe(a, dev, pool, S))
        if not a.no_cpu_baseline:
            cpu = cpu_baseline(model, renderer, batches[0], a, S)
    if rank == 0:
        line = {"metric": "ray-samples/sec (train step) at 4096 rays", "value": value, "unit": "ray-samples/s",
                "main_pass_samples_per_s": extra.get("main_pass_samples_per_s"), "n_gpus": world,
                "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": ("strong" if a.global_rays else "weak"),
                "vs_baseline": None,
                "value_definition": "nominal ray-samples of one full training_step = (main-pass rays + instance-pass rays) x S, over all ranks, / step time; the "
                                    "main pass alone (the 4096 rays of the metric's name) is `main_pass_samples_per_s`, and because cost follows the ACTIVE "
                                    "samples (f_active of the nominal ones) `active_samples_per_s_main_pass` is the scene-independent figure",
                "dtype": {"fp32": "f32", "bf16": "bf16", "fp32x6": "f32 (fp32-faithful 6-product bf16 split on the matrix cores)"}[a.dtype],
                "data": "synthetic",
                "config": {"workload": ("BASELINE configs[1] stand-in: ScanNet-shaped scene (C=22, E=3/D=6, grid 128^3, S=440), "
                                        "full training_step = main pass 4096 rays + slow-fast instance pass 1024 rays, fp32") if a.dtype == "fp32" else
                                       (f"BASELINE configs[2]-style bf16 mode on the configs[1] shapes (C={a.classes}, E=3/D=6, grid {a.grid}^3): MLP "
                                        "operands bf16 (weights rounded in-kernel, hidden activations / gradients bf16-stored), fp32 accumulate; everything else fp32"),
                           "rays_per_gpu": a.rays, "instance_rays_per_gpu": a.inst_rays, "grid": a.grid, "classes": a.classes,
                           "samples_per_ray": S, "chunk": a.chunk or a.rays, "lean_main_pass": bool(a.lean), "sync_free": bool(a.nosync),
                           "parallelism": f"dp{world} (rays sharded, 1 all-reduce per backward)"},
                "roofline": roof, "cpu_baseline": cpu,
                "dist_backend": backend, "rccl_ranks": (dist.get_world_size() if backend == "nccl" else 0),
                "devices_visible": torch.cuda.device_count(),
                "allreduce_overlap": (tr.overlap_decision or {"overlap": tr.overlap_allreduce}) if world > 1 else None,
                "allreduce_cu_reserve": tr.allreduce_cu_reserve if world > 1 else None}
        line.update(extra)
        line["step_ms_median"] = step_ms[len(step_ms) // 2]
        line["step_ms_min_max"] = [step_ms[0], step_ms[-1]]
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


FRAME_W, FRAME_H = 1296, 968             # BASELINE configs[4]: ScanNet colour frame


def head_flops_per_active_sample(model):
    """Forward matrix-core FLOPs of all heads per ACTIVE sample: 2 x (sum of in x out over every Linear of the appearance MLP + basis,
    the semantic MLP and both instance MLPs)."""
    tot = 0
    for name, t in model.named_views().items():
        if name.endswith(".weight") and t.dim() == 2 and ("mlp" in name or "basis_mat" in name):
            tot += 2 * t.shape[0] * t.shape[1]
    return float(tot)


def bench_inference_sharded(a, model, renderer, dev, world, rank, backend):
    """configs[4]: full 1296 x 968 frames (is_train=False, step ratio halved as RP:104, chunked like RP:114-120) through
    ``inference.render_rays_sharded``: contiguous row tiles, one per rank, ONE all-gather of the packed outputs per frame.  A step is one
    frame; value = rays/s over the job (strong scaling: the frame is fixed)."""
    import numpy as np
    from contrastive_lift_amd import inference as inf, synthetic
    from contrastive_lift_amd.rays import generate_ray_table
    K = np.array([[1170.0, 0, 647.75], [0, 1170.0, 483.75], [0, 0, 1]], np.float32)
    rays = generate_ray_table(FRAME_H, FRAME_W, K, synthetic.look_at((0.0, 0.0, -0.9)), device=dev)
    renderer.update_step_ratio(renderer.step_ratio * 0.5)
    S = int(renderer.n_samples)
    chunk = a.chunk or 65536

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    for _ in range(max(1, a.warmup)):
        out = inf.render_rays_sharded(model, renderer, rays, chunk)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = inf.render_rays_sharded(model, renderer, rays, chunk)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        P = rays.shape[0]
        # active samples of the frame (cost follows them): counted once on this rank's tile, outside the timed region
        b = inf.tile_bounds(P, world)
        mine = rays[b[0]:b[1]]
        act = 0
        for i in range(0, mine.shape[0], chunk):
            _, ctx = engine_render_forward(model, renderer, mine[i:i + chunk])
            act += int(ctx.M)
        fl = head_flops_per_active_sample(model)
        t_frame = dt / a.steps
        tf = act * world * fl / t_frame / 1e12          # tiles are equal-sized; this rank's active fraction stands for the frame
        print(json.dumps({"metric": "rays/sec (full-frame 1296x968 render, row tiles sharded over the ranks)", "value": P / t_frame,
                          "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": max(1, a.warmup), "ms_per_step": t_frame * 1e3,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": {"fp32": "f32"}.get(a.dtype, a.dtype),
                          "data": "synthetic",
                          "config": {"workload": "BASELINE configs[4] stand-in: one 1296x968 frame (1,254,528 rays), is_train=False, S=%d, "
                                                 "chunk %d rays, rgb/semantics/instances/distance outputs" % (S, chunk),
                                     "rays_per_frame": P, "samples_per_ray": S, "classes": a.classes, "grid": a.grid,
                                     "parallelism": f"row tiles over {world} rank(s), 1 all-gather per frame"},
                          "ray_samples_per_s": P * S / t_frame,
                          "inference_roofline": {"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                 "frac": tf / PEAK_FP32_MFMA_TFLOPS, "active_samples_per_ray": act * world / P,
                                                 "flops_per_active_sample": fl,
                                                 "note": "head FLOPs of the frame's active samples / frame time / exact-fp32 MFMA peak"},
                          "dist_backend": backend, "rccl_ranks": (dist.get_world_size() if backend == "nccl" else 0),
                          "devices_visible": torch.cuda.device_count()}))
    if world > 1:
        dist.destroy_process_group()


def engine_render_forward(model, renderer, rays):
    from contrastive_lift_amd import engine
    with torch.no_grad():
        return engine.render_forward(model, renderer, rays, None, False, grad_heads=())


def bf16_probe(a, dev, batches, S, steps=10, warmup=3):
    """The same training_step in bf16 mode (mlp_dtype "bf16": bf16 MLP operands, bf16-stored hidden activations, fp32 accumulate;
    everything else fp32) on a fresh copy of the scene."""
    from contrastive_lift_amd import engine, synthetic
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    prev = engine.set_mlp_precision("bf16")
    try:
        model, renderer, pool = synthetic.make_scene(grid=a.grid, num_classes=a.classes, max_instances=3, seed=0, device=dev)
        tr = HotPathTrainer(model, renderer, default_config(chunk=a.chunk, instance_optimization_epoch=0, late_semantic_optimization=0,
                                                            mlp_dtype="bf16"), current_epoch=4)
        for i in range(warmup):
            tr.training_step(batches[i % len(batches)], lean=a.lean)
        torch.cuda.synchronize()
        dt = extras_step_time(tr, batches, steps, a.lean)
        bf16_roof = bf16_roofline_pass(tr, batches, a.lean)
        # frame render in bf16 mode (the xyz heads run as one fused kernel each: csrc/head_bf16.hip)
        from contrastive_lift_amd import inference as inf
        ratio = renderer.step_ratio
        renderer.update_step_ratio(ratio * 0.5)
        try:
            rays = pool[:262144].contiguous()
            inf.render_rays(model, renderer, rays[:32768], 32768)
            torch.cuda.synchronize()
            t = time.perf_counter()
            inf.render_rays(model, renderer, rays, 32768)
            torch.cuda.synchronize()
            inf_rate = rays.shape[0] / (time.perf_counter() - t)
        finally:
            renderer.update_step_ratio(ratio)
    finally:
        engine.set_mlp_precision(prev)
    return dict(bf16_ms_per_step=round(dt * 1e3, 3), bf16_ray_samples_per_s=(a.rays + a.inst_rays) * S / dt, bf16_inference_rays_per_s=inf_rate,
                bf16_roofline=bf16_roof)


def extras_step_time(tr, batches, steps, lean):
    """Seconds per training_step for the EXTRAS (not the headline, whose contract is wall time over all K steps): per-step GPU times from events
    recorded at the step boundaries, MEDIAN -- a one-off stall of a few ms inside a ten-step probe (seen: fp32x6 reported 6.98 instead of 6.5 ms
    once) would otherwise pass for the mode's speed."""
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    for i in range(steps):
        tr.training_step(batches[i % len(batches)], lean=lean)
        marks[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    return ts[steps // 2] * 1e-3


def x6_probe(a, dev, batches, S, steps=10, warmup=3):
    """The same training_step with mlp_dtype "fp32x6": the 256 x 256 hidden layers (forward / dgrad) as fp32-FAITHFUL six-product bf16 splits on
    the bf16 matrix cores (csrc/layer_x6.hip); every other launch on its exact-fp32 kernel.  Reported beside the exact-fp32 headline."""
    from contrastive_lift_amd import engine, synthetic
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    prev = engine.set_mlp_precision("fp32x6")
    try:
        model, renderer, pool = synthetic.make_scene(grid=a.grid, num_classes=a.classes, max_instances=3, seed=0, device=dev)
        tr = HotPathTrainer(model, renderer, default_config(chunk=a.chunk, instance_optimization_epoch=0, late_semantic_optimization=0,
                                                            mlp_dtype="fp32x6"), current_epoch=4)
        for i in range(warmup):
            tr.training_step(batches[i % len(batches)], lean=a.lean)
        torch.cuda.synchronize()
        dt = extras_step_time(tr, batches, steps, a.lean)
        # frame render in fp32x6 mode (same rays / chunking as the bf16 probe's and the exact path's `inference_rays_per_s`)
        from contrastive_lift_amd import inference as inf
        ratio = renderer.step_ratio
        renderer.update_step_ratio(ratio * 0.5)
        try:
            rays = pool[:262144].contiguous()
            inf.render_rays(model, renderer, rays[:32768], 32768)
            torch.cuda.synchronize()
            t = time.perf_counter()
            inf.render_rays(model, renderer, rays, 32768)
            torch.cuda.synchronize()
            inf_rate = rays.shape[0] / (time.perf_counter() - t)
        finally:
            renderer.update_step_ratio(ratio)
    finally:
        engine.set_mlp_precision(prev)
    return dict(fp32x6_ms_per_step=round(dt * 1e3, 3), fp32x6_ray_samples_per_s=(a.rays + a.inst_rays) * S / dt, fp32x6_inference_rays_per_s=inf_rate,
                fp32x6_note="fp32-faithful (6 bf16 products of exactly split operands, fp32 accumulate): outputs / gradients meet the exact path's "
                            "test tolerances (tests/test_gpu_round3.py, CLIFT_FORCE_MLP_DTYPE=fp32x6 runs of the suite); not the headline")


def small_batch_probe(a, dev, pool, S, rays=1024, steps=10, warmup=3):
    """BASELINE configs[3] per-GPU shape: 8192 rays per step over 8 GPUs = 1024 main-pass rays per rank (+ one 1024-ray instance image per rank,
    as the reference's DDP), exact fp32, on this one GPU.  The ideal is a quarter of the 4096-ray main pass + the unchanged instance pass."""
    from contrastive_lift_amd import synthetic
    from contrastive_lift_amd.trainer import HotPathTrainer, default_config
    model, renderer, _ = synthetic.make_scene(grid=a.grid, num_classes=a.classes, max_instances=3, seed=0, device=dev)
    tr = HotPathTrainer(model, renderer, default_config(chunk=a.chunk, instance_optimization_epoch=0, late_semantic_optimization=0), current_epoch=4)
    bs = [synthetic.make_batches(pool, rays, a.inst_rays, a.classes, 25, seed=300 + i, device=dev) for i in range(4)]
    for i in range(warmup):
        tr.training_step(bs[i % 4], lean=a.lean)
    torch.cuda.synchronize()
    dt = extras_step_time(tr, bs, steps, a.lean)
    return dict(rays1024_ms_per_step=round(dt * 1e3, 3), rays1024_ray_samples_per_s=(rays + a.inst_rays) * S / dt,
                rays1024_note="configs[3] per-GPU shape (8192 global rays / 8 ranks): 1024 main-pass rays + 1024 instance rays per step, exact fp32")


def bf16_roofline_pass(tr, batches, lean, steps=3):
    """bf16 mode is HBM-bound (bf16 MFMAs run 16x the fp32 rate): per kernel family of the xyz heads, ALGORITHMIC bytes per launch / launch
    time from HIP events around the launch sites (engine.head_bf16 = the fused whole-head forward, engine.gemm = per-layer forward / dgrad
    / wgrad launches), over `steps` training steps after the timed ones.  The family with the largest time share is reported as the
    dominant kernel, against the 8 TB/s peak and the ~6.3 TB/s a pure stream reaches on this chip (MI355X_MICROARCH.md)."""
    from contrastive_lift_amd import engine
    real_head, real_gemm = engine.head_bf16, engine.gemm
    rec = []

    def bracket(kind, nbytes, flops, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        rec.append((kind, nbytes, flops, e0, e1))

    def head(M, xa, l0, l1, l2, lout, h1, h2, h3, out, ldo, col_off):
        kept = sum(x is not None for x in (h1, h2, h3))
        nb = M * (16.0 + (4.0 * lout[0].shape[0] if lout is not None else 0.0) + kept * 512.0) + 3 * 256 * 256 * 4.0
        fl = 2.0 * M * (3 * 256 + 2 * 256 * 256 + (256 * lout[0].shape[0] if lout is not None else 0))
        return bracket("k_head_bf16_fwd (whole xyz head forward)", nb, fl, lambda: real_head(M, xa, l0, l1, l2, lout, h1, h2, h3, out, ldo, col_off))

    def gemm(M, N, K, A, lda, B, ldb, Cm, ldc, **kw):
        es = lambda t: float(t.element_size())
        if kw.get("a_trans"):       # wgrad: both operands streamed over K rows, result (M x N) accumulated
            kind, nb = "wgrad", K * (M * es(A) + N * es(B)) + M * N * 4.0
        else:
            kind = "dgrad" if kw.get("b_trans") else "fwd"
            nb = M * K * es(A) + M * N * es(Cm) + N * K * es(B) + (M * N * es(kw["mask"]) if kw.get("mask") is not None else 0.0)
        kind = f"{kind} {N}x{K}" if not kw.get("a_trans") else f"wgrad {M}x{N}"
        return bracket(kind, nb, 2.0 * M * N * K, lambda: real_gemm(M, N, K, A, lda, B, ldb, Cm, ldc, **kw))
    engine.head_bf16, engine.gemm = head, gemm
    try:
        for i in range(steps):
            tr.training_step(batches[i % len(batches)], lean=lean)
        torch.cuda.synchronize()
    finally:
        engine.head_bf16, engine.gemm = real_head, real_gemm
    fam = {}
    for kind, nb, fl, e0, e1 in rec:
        f = fam.setdefault(kind, [0.0, 0.0, 0, 0.0])
        f[0] += nb; f[1] += e0.elapsed_time(e1); f[2] += 1; f[3] += fl
    if not fam:
        return None
    table = {k: {"GBps": v[0] / (v[1] * 1e-3) / 1e9, "tflops": v[3] / (v[1] * 1e-3) / 1e12, "ms_per_step": v[1] / steps,
                 "launches_per_step": v[2] / steps} for k, v in fam.items() if v[1] > 0}
    dom = max(table, key=lambda k: table[k]["ms_per_step"])
    g, tfl = table[dom]["GBps"], table[dom]["tflops"]
    hbm_frac, mfma_frac = g / PEAK_HBM_GBS, tfl / 2500.0
    # The fused head forward moves 16 B in / 4 E B out per row (+ 512 B per kept activation): it is NOT a stream -- its ceiling is the bf16
    # matrix pipe / LDS (every wave re-reads the 64-row tile from LDS); the per-layer launches (activations through HBM) are streams.
    return {"bound": "hbm" if hbm_frac >= mfma_frac else "mfma", "kernel": dom,
            "achieved": g if hbm_frac >= mfma_frac else tfl, "peak": PEAK_HBM_GBS if hbm_frac >= mfma_frac else 2500.0,
            "unit": "GB/s" if hbm_frac >= mfma_frac else "TFLOP/s (dense bf16 MFMA)", "frac": max(hbm_frac, mfma_frac),
            "hbm_GBps": g, "hbm_frac": hbm_frac, "hbm_frac_of_achievable_stream_6300": g / 6300.0, "mfma_tflops": tfl, "mfma_frac": mfma_frac,
            "families": table,
            "note": "algorithmic bytes per launch (operands + results as stored: bf16 activations, fp32 weights / narrow tensors) / HIP-event time "
                    "(events make the step host-bound, so per-launch times are upper bounds)"}


def inference_probe(cl, model, renderer, pool, n_rays=262144, chunk=32768):
    """Frame-render throughput (BASELINE configs[4] shape: is_train=False, halved step ratio as RP:104, chunked like
    RP:114-120) on a bounded 262144-ray tile -- reported beside the training metric, outside the timed region."""
    from contrastive_lift_amd import inference as inf
    ratio = renderer.step_ratio
    renderer.update_step_ratio(ratio * 0.5)
    try:
        rays = pool[:n_rays].contiguous()
        inf.render_rays(model, renderer, rays[:chunk], chunk)
        torch.cuda.synchronize()
        t = time.perf_counter()
        inf.render_rays(model, renderer, rays, chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        S = int(renderer.n_samples)
        act = 0
        for i in range(0, rays.shape[0], chunk):
            act += int(engine_render_forward(model, renderer, rays[i:i + chunk])[1].M)
    finally:
        renderer.update_step_ratio(ratio)
    fl = head_flops_per_active_sample(model)
    tf = act * fl / dt / 1e12
    return dict(inference_roofline={"bound": "mfma", "achieved": tf, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_MFMA_TFLOPS,
                                    "active_samples_per_ray": act / rays.shape[0], "flops_per_active_sample": fl,
                                    "note": "forward head FLOPs of the tile's active samples / render time / exact-fp32 MFMA peak"},
                inference_rays_per_s=rays.shape[0] / dt, inference_ray_samples_per_s=rays.shape[0] * S / dt,
                inference_samples_per_ray=S, inference_probe=f"{rays.shape[0]} rays, chunk {chunk}, fp32 outputs rgb/sem/inst/dist")


def measured_traffic_ratio():
    """HBM bytes of the dominant kernel from counters, as a ratio to its algorithmic bytes: profiles/r03_pmc_k_layer_f32.json holds
    the rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    16-byte-per-lane streaming reads on gfx950) over this very command and over the torch-free single-kernel harness."""
    try:
        with open(os.path.join(REPO, PMC_RECORD)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def roofline(rec_dom, rec, nb, engine, dtype="fp32", ms_step=None):
    """``rec`` = the (kind, M, N, K, start event, end event) records of every clift_gemm launch of the ``nb`` timed steps.  The
    dominant kernel by time is k_layer_f32<false> (csrc/layer_f32.hip) = the 256x256 forward layers of the semantic / fast /
    slow instance MLPs as a persistent kernel (rocprofv3 lists it under exactly that name, profiles/r01_v12_*): achieved = its
    algorithmic FLOPs (2*M*256*256 per launch, M = active samples of the pass) / its summed launch durations; peak = dense fp32
    MFMA.  ``all_gemm`` is the same ratio over every matrix-core launch (forward, dgrad, wgrad, narrow layers).  The active-sample
    count drifts while the field trains, which is why the records come from a replay of exactly the timed steps."""
    tot_f, tot_ms, by = 0.0, 0.0, {}
    dom_f, dom_ms, dom_n = 0.0, 0.0, 0
    for kind, M, N, K, ms, xf in rec:
        fl = 2.0 * M * N * K + xf
        tot_f += fl
        tot_ms += ms
        b = by.setdefault(kind, [0.0, 0.0, 0])
        b[0] += fl; b[1] += ms; b[2] += 1
    inst = {}
    dump = os.environ.get("CLIFT_BENCH_DUMP")
    for kind, M, N, K, ms, xf in rec_dom:
        if dump:
            print(f"[dom] {kind} M={M} N={N} K={K} {ms:.4f} ms", file=sys.stderr)
        dom_f += 2.0 * M * N * K + xf; dom_ms += ms; dom_n += 1
        b = inst.setdefault(kind, [0.0, 0.0, 0])
        b[0] += 2.0 * M * N * K + xf; b[1] += ms; b[2] += 1
    tf = lambda f, ms: f / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    ach = tf(dom_f, dom_ms)
    if dtype == "bf16":
        # bf16 operands run the same layers at 16x the MFMA rate: the dominant kernel is then bound by streaming its
        # activations.  Algorithmic bytes per launch = A read + C written (M x 256 x 2 bytes each, bf16-stored) + weights (256 KB, L2).
        esz = 2.0 if engine.act_dtype() == torch.bfloat16 else 4.0          # hidden activations are bf16-stored in bf16 mode
        dom_b = sum(esz * M * K + esz * M * N + 4.0 * N * K for kind, M, N, K, _, _ in rec_dom)
        gbs = dom_b / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        return {"bound": "hbm", "kernel": "k_layer_bf16<false,3> (persistent streamed 256x256 forward layers: weights in registers, LDS-DMA ring, v_mfma_f32_32x32x16_bf16; bf16-stored activations)",
                "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                "launches_per_step": dom_n // nb, "avg_launch_ms": dom_ms / max(1, dom_n), "mfma_tflops_of_same_kernel": ach,
                "all_gemm": {"achieved_tflops": tf(tot_f, tot_ms), "launches_per_step": len(rec) // nb, "ms_per_step": tot_ms / nb,
                             "gflop_per_step": tot_f / 1e9 / nb,
                             "by_kind": {k: {"tflops": tf(v[0], v[1]), "ms": v[1] / nb, "launches": v[2] // nb} for k, v in by.items()}}}
    if dtype == "fp32x6":
        # fp32x6: the 256 x 256 forward layers run as six bf16 products per fp32 product on the bf16 matrix cores (csrc/layer_x6.hip).  Two ceilings,
        # both reported: the bf16 MFMA pipe (6 x 2MNK FLOPs per launch against the 2.5 PFLOP/s dense peak) and HBM (a row read + a row written).
        plain = inst.get("fwd", [dom_f, dom_ms, dom_n])
        rows = plain[0] / (2.0 * 256 * 256)                                  # summed rows of the plain launches
        bytes_plain = rows * 2048.0 + plain[2] * 256 * 256 * 4.0
        bf16_tf = 6.0 * ach
        gbs = bytes_plain / (plain[1] * 1e-3) / 1e9 if plain[1] > 0 else 0.0
        return {"bound": "mfma", "kernel": "k_layer_x6<false, *> (persistent fp32-faithful split kernel: pair of workgroups per row range, the weights' three "
                                           "bf16 planes in registers, cooperative activation split through LDS-DMA staging, v_mfma_f32_32x32x16_bf16; "
                                           "eight waves = 4 column groups x 2 k-halves meeting through LDS)",
                "achieved": bf16_tf, "peak": 2500.0, "unit": "TFLOP/s (bf16 MFMA FLOPs = 6 x the fp32-equivalent 2MNK)", "frac": bf16_tf / 2500.0,
                "fp32_equivalent_tflops": ach, "fp32_equivalent_vs_exact_fp32_peak": ach / PEAK_FP32_MFMA_TFLOPS,
                "hbm_GBps_plain_launches": gbs, "hbm_frac": gbs / PEAK_HBM_GBS, "traffic": None,
                "note": "a bare stream of these MFMAs (no memory, no VALU) runs 113 us per 249 k-row launch on this chip = 1.73 PFLOP/s at the ~1.7 GHz it "
                        "holds under that load; with the 255 MB of output writes the kernel sits at ~190 us (profiles/r03_x6_notes.txt)",
                "launches_per_step": dom_n // nb, "avg_launch_ms": dom_ms / max(1, dom_n),
                "instantiations": {k: {"fp32_equiv_tflops": tf(v[0], v[1]), "launches_per_step": v[2] // nb, "avg_launch_ms": v[1] / max(1, v[2])} for k, v in inst.items()},
                "all_gemm": {"fp32_equiv_tflops": tf(tot_f, tot_ms), "launches_per_step": len(rec) // nb, "ms_per_step": tot_ms / nb, "gflop_per_step": tot_f / 1e9 / nb,
                             "by_kind": {k: {"tflops": tf(v[0], v[1]), "ms": v[1] / nb, "launches": v[2] // nb} for k, v in by.items()}}}
    # algorithmic bytes of the average launch: the plain instantiation reads a 1 KB row and writes one per sample; the generating one reads
    # 16 B and writes 1 KB (2 KB when the first layer's activation is kept); the output-fused one reads 1 KB and writes 16 B (+ 1 KB when kept).
    # Reported for the plain instantiation, the one the counter passes measured (8 B per output element + 256 KB weights).
    plain = inst.get("fwd", [dom_f, dom_ms, dom_n])
    alg_bytes = (plain[0] / max(1, plain[2])) / (2.0 * 256.0) * 8.0 + 256.0 * 256.0 * 4.0
    pmc = measured_traffic_ratio()
    names = {"fwd": "k_layer_f32<false, false, false, false>", "fwd_gen": "k_layer_f32<false, true, false, false>", "fwd_out": "k_layer_f32<false, false, true, false>"}
    out = {"bound": "mfma", "kernel": "k_layer_f32<false, *, *, false> (persistent fp32 v_mfma_f32_32x32x2_f32 kernel: the 256x256 forward MLP layers of the xyz heads, "
                                      "11 launches per step in three instantiations -- input streamed from memory / K = 3 input layer generated in-kernel / narrow "
                                      "output layer applied in-kernel; rocprofv3 lists them separately: see `instantiations`)",
           "instantiations": {names[k]: {"tflops": tf(v[0], v[1]), "frac": tf(v[0], v[1]) / PEAK_FP32_MFMA_TFLOPS, "launches_per_step": v[2] // nb,
                                         "avg_launch_ms": v[1] / max(1, v[2])} for k, v in inst.items()},
           "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
           # HBM bytes per (average) launch of the dominant kernel: algorithmic bytes of THIS run's average launch (A read once + C
           # written once = 8 B per output element, + 256 KB weights) x the counter/algorithmic ratio measured by rocprofv3 --pmc on
           # the same command (profiles/r03_pmc_k_layer_f32.json); null if that record is missing
           "traffic": (alg_bytes * pmc["bench_ratio"]) if pmc else None,
           "traffic_is": "ESTIMATE = this run's algorithmic bytes x the counter/algorithmic ratio RECORDED in " + PMC_RECORD + " (not re-measured by this run: "
                         "PMC passes need rocprofv3 around the process; tools/gpu_pmc_bench.sh re-records it)",
           "traffic_unit": "bytes/launch (average launch of this run)",
           "traffic_algorithmic": alg_bytes,
           "traffic_over_algorithmic": pmc["bench_ratio"] if pmc else None,
           "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH_SIZE x2, gfx950 correction of MI355X_MICROARCH.md) "
                              "over `python bench.py --steps 3 --warmup 1`: profiles/r03_pmc_k_layer_f32.json") if pmc else None,
           "mfma_busy_frac_counters": pmc.get("mfma_busy_frac") if pmc else None,
           "launches_per_step": dom_n // nb, "avg_launch_ms": dom_ms / max(1, dom_n), "gflop_per_launch_avg": dom_f / max(1, dom_n) / 1e9,
           "all_gemm": {"achieved": tf(tot_f, tot_ms), "frac": tf(tot_f, tot_ms) / PEAK_FP32_MFMA_TFLOPS, "launches_per_step": len(rec) // nb,
                        "ms_per_step": tot_ms / nb, "gflop_per_step": tot_f / 1e9 / nb,
                        "by_kind": {k: {"tflops": tf(v[0], v[1]), "ms": v[1] / nb, "launches": v[2] // nb} for k, v in by.items()}}}
    if ms_step:
        # whole-step fraction of the fp32-MFMA roof: every matrix-core FLOP of a step / the step's wall time / peak
        out["step_gflop"] = tot_f / 1e9 / nb
        out["step_frac"] = (tot_f / nb) / (ms_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS
    return out


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(model, renderer, batch, a, S):
    """The oracle (CPU-PyTorch restatement of the reference path, same ATen ops / shapes / chunk 2048) timed on the
    host cores of this box: 1 warm-up + up to 3 full-size steps within the budget."""
    from oracle import render as orender
    from oracle.train_step import CpuTrainer
    cores = usable_cores()
    torch.set_num_threads(cores)
    P = {k: v.detach().cpu() for k, v in model.export_state_dict().items()}
    cfg = orender.RenderCfg(renderer.bbox_aabb.cpu(), tuple(int(x) for x in renderer.grid_dim.tolist()), density_shift=-3.0)
    ct = CpuTrainer(P, cfg, chunk=2048, epoch=4)
    b0, b1 = batch[0], batch[1][0]
    rays, rgbs, probs, conf = (b0[k].cpu() for k in ("rays", "rgbs", "probabilities", "confidences"))
    ir, il, ic = b1["rays"].cpu(), b1["instances"].cpu(), b1["confidences"].cpu()
    g = torch.Generator().manual_seed(1)

    def step():
        jit = torch.rand(rays.shape[0], generator=g)
        ct.main_pass(rays, rgbs, probs, conf, jit, [False] * ((rays.shape[0] + 2047) // 2048))
        ct.instance_pass(ir, il, ic, torch.rand(ir.shape[0], generator=g))
    t0 = time.perf_counter()
    step()
    warm = time.perf_counter() - t0
    ts = []
    while len(ts) < 3 and (time.perf_counter() - t0) + warm < a.cpu_budget_s:
        t = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t)
    if not ts:
        ts = [warm]
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": (a.rays + a.inst_rays) * S / med, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"{len(ts)} full-size training_step(s) after 1 warm-up ({a.rays}+{a.inst_rays} rays x S={S}, chunk 2048, "
                      f"torch {torch.__version__} CPU, {cores} threads), median {med:.2f} s/step",
            "s_per_step": med,
            "port_vs_reference": "build container, 8 threads, same step: oracle port 6.28 s, imported reference trainer 5.22 s (port / reference time "
                                 "= 1.20; tools/cpu_port_vs_reference.py, profiles/r02_cpu_port_vs_reference.json) -- the reference's own CPU "
                                 "path is ~1.2x FASTER than this port number"}


if __name__ == "__main__":
    main()
