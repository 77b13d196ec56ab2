#!/usr/bin/env python3
"""Headline benchmark: img/s of the full G+D StyleGAN training iteration (optimize_discriminator +
optimize_generator: logistic loss + R1, both Adam steps, grad clip, EMA) on synthetic data, on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json `metric` / configs[2..3]): FFHQ-1024 model (8 mapping layers, no truncation), progressive
depth index 8 (1024x1024), batch 4 per GPU, alpha 0.5 (fade-in active: both branches live), bf16 activations with
fp32 accumulation / parameters.  Random-init weights, synthetic N(0,1) latents and images.

The LAST stdout line is ONE compact JSON object (< 4 KB: the contract's keys, `roofline`, `cpu_baseline`, a scalar `b32` block);
the full record with every table goes to gpurun_out/bench_detail.json (`--detail-file`).  `roofline` is measured live with HIP events around every launch of the dominant kernel (the
instantiation with the largest total time in a surveyed single-stream step), on the launch stream, by the library's own
per-launch profiler (sgx_prof_*), in a single-stream eager re-run of the timed steps right after the timed region: each
launch alone on the GPU, the number `rocprofv3 --kernel-trace --stats -- python bench.py --graphs off --streams 00`
reproduces (profiles/).  `b32` (default invocation, N=1): a second measured block at batch 32 on the one GPU -- the
configuration BASELINE.json's north-star target is stated on -- with its own throughput and `roofline`.  `cpu_baseline`
times the CPU oracle (a port of the reference step, oracle/stylegan_oracle.py) -- or the reference itself where its
sources exist -- on this host's cores on a bounded sample (rank 0, N=1 only).  Two short extra blocks follow the
measured ones in the default invocation (N=1, batch 4; `--no-extras` skips them): `sweep_top_depths` = BASELINE
configs[4]'s depth indices 6,7,8 at the reference's batch sizes with hipGraph replay (the mode `StyleGAN.train` runs),
and `ffhq128_fp32_b64` = BASELINE configs[1] as its own bench line from a child process.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

CONFIGS = {
    # name: (resolution, mapping_layers, truncation_psi, depth index)
    "ffhq1024": dict(resolution=1024, mapping_layers=8, truncation_psi=-1.0, depth=8, flops_per_img=1223.27e9),
    "ffhq128": dict(resolution=128, mapping_layers=4, truncation_psi=0.7, depth=5, flops_per_img=692.85e9),
}
PEAK = {"bf16": 2500e12, "fp32": 157.3e12}      # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8e12                                 # HBM3E bytes/s, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="ffhq1024", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--alpha", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-timeout", type=float, default=150.0)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--graphs", default="auto", choices=["auto", "on", "off"],
                    help="replay each half-iteration as captured hipGraphs (with data parallelism: two graphs around the eager all-reduce)")
    ap.add_argument("--streams", default="auto", choices=["auto", "11", "01", "00"],
                    help="extra streams of the step: auxiliary (fake branch) / side (weight gradients); auto = measured on this box")
    ap.add_argument("--dry-run-ranks-on-one-gpu", action="store_true",
                    help="TEST ONLY: every rank on cuda:0 over a gloo group (RCCL refuses two ranks per device): exercises the N>1 "
                         "control flow (calibration votes, sharded step, all-reduce, max-over-ranks timing) on a 1-GPU box; the "
                         "printed throughput is meaningless")
    ap.add_argument("--layer-table", default=None, help="write a per-layer conv/wgrad timing table (TSV) to this path (+ .b32.tsv for the batch-32 block)")
    ap.add_argument("--no-b32", action="store_true", help="skip the second measured block (batch 32 on one GPU)")
    ap.add_argument("--b32-steps", type=int, default=8)
    ap.add_argument("--rccl-group-of-one", action="store_true",
                    help="N=1 only: run the data-parallel code path over an RCCL process group of size 1 (bucketed all-reduce launched for real, "
                         "replay split into [graph | all-reduce | update]) -- measures what that path costs per step on one GPU; not a scaling run")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the two short extra blocks of the default invocation: BASELINE configs[4]'s top depths (6,7,8) replayed, "
                         "and BASELINE configs[1] (ffhq128, fp32, batch 64) in a child process")
    ap.add_argument("--detail-file", default="gpurun_out/bench_detail.json",
                    help="where the full record (per-layer tables, calibration, sweep rows, the configs[1] block) is written, relative to the repo root")
    ap.add_argument("--no-detail-file", action="store_true")
    ap.add_argument("--sweep", action="store_true",
                    help="BASELINE configs[4]: the progressive-growing sweep, depth index 0..8 of the 1024 model with the reference's per-depth "
                         "batch sizes (config.py:40-41), the fade-in ramp of models/GAN.py:748-753 and style mixing on; per-depth img/s")
    ap.add_argument("--sweep-steps", type=int, default=8, help="timed iterations per depth (first half on the alpha ramp, second half at alpha = 1)")
    ap.add_argument("--sweep-depths", default=None, help="comma-separated depth indices to measure (default: all)")
    ap.add_argument("--sweep-batch-scale", type=float, default=1.0,
                    help="multiply the reference's per-depth batch sizes (its schedule fits an 11 GB card; kept a multiple of 4 where >= 4)")
    return ap.parse_args()


# reference config.py:37-42 (cfg.sched): per-depth batch sizes for resolutions 4 .. 1024 and the fade-in share of a depth's iterations
REF_BATCH_SIZES = [128, 128, 128, 64, 32, 16, 8, 4, 2]
REF_FADE_IN_PERCENTAGE = 50
# algorithmic conv+GEMM GFLOP per image of one full G+D iteration (logistic + R1, 8 mapping layers), by depth index
SWEEP_GFLOP_PER_IMG = [1.60, 13.10, 59.05, 242.87, 518.72, 692.85, 867.73, 1044.05, 1223.27]


def sweep(sg, a, cfg, dev):
    """The progressive-growing schedule as a measured workload (one GPU): for every depth index the reference's batch size, real
    batches at the FULL resolution (the reference's loader yields 1024^2 images at every depth and the step average-pools them
    down, models/GAN.py:557-589), alpha = ticker / fade_point up to the fade point and 1 after it (:748-753) with the fade point
    in the middle of the timed iterations (fade_in_percentage 50), style mixing on, eager launches (what StyleGAN.train runs).
    -> list of per-depth dicts."""
    from stylegan.pytorch_amd import native
    from stylegan.pytorch_amd.GAN import StyleGAN
    res = cfg["resolution"]
    K = max(2, a.sweep_steps)
    fade_point = StyleGAN.fade_point_of(REF_FADE_IN_PERCENTAGE, 1, K)
    rows = []
    gen = torch.Generator(device=dev); gen.manual_seed(4321)
    import random
    random.seed(4321)
    depths = range(cfg["depth"] + 1) if not a.sweep_depths else [int(d) for d in a.sweep_depths.split(",")]
    for depth in depths:
        B = REF_BATCH_SIZES[depth]
        if a.sweep_batch_scale != 1.0:
            B = max(1, int(round(B * a.sweep_batch_scale)))
            if B >= 4:
                B -= B % 4
        reals = [torch.randn(B, res, res, 3, device=dev, generator=gen).permute(0, 3, 1, 2) for _ in range(2)]
        lats = [torch.randn(B, 512, device=dev, generator=gen) for _ in range(2)]
        alphas = [StyleGAN.alpha_at(t, fade_point) for t in range(1, K + 1)]

        def step(i, alpha):
            sg.optimize_discriminator(lats[i % 2], reals[i % 2], depth, alpha)
            sg.optimize_generator(lats[i % 2], reals[i % 2], depth, alpha)
        replay = bool(sg.use_graphs)
        for i in range(1 if replay else 2):                  # untimed: first use of this depth's kernels, packs, gradient buffers
            step(i, alphas[0])
        torch.cuda.synchronize()
        native.prof_start(1)                                 # one surveyed iteration: launches per step, kernel time (replay: the second
        step(0, alphas[0])                                   # of the two eager calls before the capture -- events cannot be captured)
        torch.cuda.synchronize()
        native.prof_start(0)
        recs = native.prof_records()
        torch.cuda.synchronize()
        if replay:
            step(1, alphas[0])                               # untimed: the capture of both half-iterations and their first replay
            torch.cuda.synchronize()
        prof = None
        if os.environ.get("SGX_SWEEP_PROFILE"):              # where the host spends the timed loop (stderr)
            import cProfile
            prof = cProfile.Profile(); prof.enable()
        t0 = time.perf_counter()
        for i in range(K):
            step(i, alphas[i])
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if prof is not None:
            import pstats
            prof.disable()
            st = pstats.Stats(prof, stream=sys.stderr)
            st.sort_stats("tottime").print_stats(12)
            st.print_callers("copy_|pin_memory")
        img_s = B * K / dt
        useful = img_s * SWEEP_GFLOP_PER_IMG[depth] * 1e9
        rows.append({"depth": depth, "resolution": 4 << depth, "batch": B, "alphas": [round(float(x), 4) for x in alphas],
                     "img_per_s": round(img_s, 2), "ms_per_step": round(dt / K * 1e3, 3), "host_enqueue_ms_per_step": round(t_enq / K * 1e3, 3),
                     "useful_tflops": round(useful / 1e12, 2), "frac_of_mfma_peak": round(useful / PEAK[a.dtype], 5),
                     "library_launches_per_step": len(recs), "library_kernel_ms_per_step": round(sum(r[1] for r in recs), 3),
                     "reference_gflop_per_img": SWEEP_GFLOP_PER_IMG[depth]})
        del reals, lats
        torch.cuda.empty_cache()
    return rows


def extra_sweep_top_depths(sg, a, cfg, dev):
    """Default invocation, N=1: BASELINE configs[4]'s three top depths (index 6,7,8 = 256^2, 512^2, 1024^2 at the reference's batch sizes
    8, 4, 2) with hipGraph replay -- what `StyleGAN.train` runs since round 5 -- 6 timed iterations each, after the headline blocks
    (never inside their timed regions).  A failure here is recorded, it does not take the line down."""
    import copy
    b = copy.copy(a)
    b.sweep_depths, b.sweep_steps, b.sweep_batch_scale = "6,7,8", 6, 1.0
    was = sg.use_graphs
    try:
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        sg._step_graphs.clear()
        sg.use_graphs = True
        t0 = time.perf_counter()
        rows = sweep(sg, b, cfg, dev)
        return {"config": {"workload": "ffhq1024 model, depth index 6,7,8 at the reference's batch sizes 8,4,2 (config.py:40-41), fade-in over the "
                                       "first half of the timed iterations, style mixing on, hipGraph replay"},
                "steps": b.sweep_steps, "rows": rows, "wall_s": round(time.perf_counter() - t0, 1)}
    except Exception as e:                                   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        sg.use_graphs = was
        sg._step_graphs.clear()
        torch.cuda.synchronize(); torch.cuda.empty_cache()


def extra_ffhq128_fp32_b64(timeout_s=150.0):
    """Default invocation, N=1: BASELINE configs[1] (FFHQ-128 model, depth index 5, fp32, batch 64) as its own bench line from a child
    process (own model, own dtype), embedded without its evidence text."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "ffhq128", "--dtype", "fp32", "--batch-per-gpu", "64",
           "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-extras", "--no-detail-file"]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"child rc {r.returncode}: {r.stderr.strip()[-300:]}"}
        j = json.loads(line[-1])
        keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "host_enqueue_ms_per_step", "hip_graphs",
                "useful_tflops", "executed_tflops", "executed_frac_of_mfma_peak", "roofline")
        out = {k: j[k] for k in keep if k in j}
        out["wall_s"] = round(time.perf_counter() - t0, 1)
        return out
    except subprocess.TimeoutExpired:
        return {"error": f"child exceeded {timeout_s:.0f} s"}
    except Exception as e:                                   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def cpu_baseline_subprocess(config, timeout_s):
    """Run the CPU-oracle leg in a child process with a hard time limit, so a slow host cannot eat the run."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--config", config]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "error": (r.stderr or r.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "error": f"CPU oracle leg exceeded {timeout_s} s on this host; see BASELINE.md section 2 for "
                                         "the reference's own CPU numbers"}


CPU_BATCH = 4            # the CPU legs run the metric's batch (4 = one minibatch-stddev group of four, as on the GPU)

# the REFERENCE ITSELF timed in the build container (its sources do not exist on the GPU boxes, where the
# port below is what can be timed) -- carried in the line as a labelled, not-measured-here field
REFERENCE_CONTAINER = {"value": 0.126, "unit": "img/s", "cores": 8, "kind": "reference",
                       "sample": "reference models/GAN.py optimize_discriminator + optimize_generator, 1024x1024 depth index 8, batch 2, fp32, "
                                 "8-core Xeon @ 2.1 GHz of the build container (BASELINE.md section 2; NOT measured in this run)"}


def cpu_baseline_reference(cfg, batch=CPU_BATCH):
    """One iteration of the REFERENCE ITSELF on this host's cores (kind "reference"), when its sources are present
    (/root/reference: the build container; the GPU boxes do not have it -> None, and the port below is timed instead).
    Imported with the shims of tests/golden/make_golden.py: a stub ``data`` module (models/GAN.py:25 pulls torchvision in
    through it), no bytecode written into the read-only mount, float Adam betas (config.py:81 gives an int)."""
    ref = os.environ.get("SGX_REFERENCE_DIR", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "models", "GAN.py")):
        return None
    import random
    import types
    sys.dont_write_bytecode = True
    sys.path.insert(0, ref)
    stub = types.ModuleType("data"); stub.get_data_loader = None; sys.modules["data"] = stub
    from models.GAN import StyleGAN as RefStyleGAN
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    res, depth = cfg["resolution"], cfg["depth"]
    torch.manual_seed(0); random.seed(0)
    opt = dict(learning_rate=0.003, beta_1=0.0, beta_2=0.99, eps=1e-8)
    sg = RefStyleGAN("linear", res, 3, 512,
                     g_args=dict(latent_size=512, mapping_layers=cfg["mapping_layers"], blur_filter=[1, 2, 1],
                                 truncation_psi=cfg["truncation_psi"], truncation_cutoff=8),
                     d_args=dict(use_wscale=True, blur_filter=[1, 2, 1]), g_opt_args=opt, d_opt_args=opt,
                     loss="logistic", d_repeats=1, use_ema=True, ema_decay=0.999, device=torch.device("cpu"))
    sg.gen.train(); sg.dis.train(); sg.gen_shadow.train()
    z = torch.randn(batch, 512); real = torch.randn(batch, 3, res, res)
    t0 = time.time()
    sg.optimize_discriminator(z, real, depth, 0.5)
    sg.optimize_generator(z, real, depth, 0.5)
    dt = time.time() - t0
    return {"value": batch / dt, "unit": "img/s", "cores": cores, "kind": "reference",
            "sample": f"1 full G+D iteration of the reference's own StyleGAN.optimize_discriminator + optimize_generator "
                      f"(models/GAN.py:591-659), batch {batch}, {res}x{res} depth index {depth}, fp32 CPU, {cores} threads, {dt:.1f} s"}


def cpu_baseline(cfg, batch=CPU_BATCH):
    """One iteration of the CPU oracle (fp32, torch CPU ops) at the metric's batch (4).  Threads are capped: the step is
    thousands of small ATen ops, and a 256-thread fork/join per op is slower than 16 threads."""
    import random
    from oracle import stylegan_oracle as O
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    res, depth = cfg["resolution"], cfg["depth"]
    total_depth = int(torch.log2(torch.tensor(float(res))).item()) - 1
    torch.manual_seed(0)
    gp = O.make_generator_params(res, cfg["mapping_layers"], truncation=cfg["truncation_psi"] > 0)
    dp = O.make_discriminator_params(res)
    shadow = {k: v.detach().clone() for k, v in gp.items()}
    noises = [torch.randn(s) for s in O.noise_shapes(batch, total_depth - 1)]
    z = torch.randn(batch, 512); real = torch.randn(batch, 3, res, res)
    kw = dict(total_depth=total_depth, mapping_layers=cfg["mapping_layers"], noises=noises,
              truncation_psi=cfg["truncation_psi"])
    random.seed(0)
    t0 = time.time()
    l2, cut = O.draw_mixing(z.shape, depth)
    O.d_step(gp, dp, O.AdamState(), z, real, depth, 0.5, latents2=l2, mixing_cutoff=cut, **kw)
    l2, cut = O.draw_mixing(z.shape, depth)
    O.g_step(gp, dp, O.AdamState(), z, depth, 0.5, latents2=l2, mixing_cutoff=cut, shadow=shadow, **kw)
    dt = time.time() - t0
    return {"value": batch / dt, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"1 full G+D iteration, batch {batch}, {res}x{res} depth index {depth}, fp32 torch-CPU oracle "
                      f"(oracle/stylegan_oracle.py; the reference's sources are not on this box), {cores} threads, {dt:.1f} s"}


def flush_c_stdio():
    """fflush(NULL): push out what C libraries (RCCL's banner) hold in stdio buffers now, not at exit."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                        # noqa: BLE001 -- cosmetic
        pass


def make_stylegan(a, cfg, dev, dp):
    from stylegan.pytorch_amd.GAN import StyleGAN
    act_dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    torch.manual_seed(0)                                    # identical random-init weights on every rank
    opt = dict(learning_rate=0.003, beta_1=0, beta_2=0.99, eps=1e-8)
    sg = StyleGAN("linear", cfg["resolution"], 3, 512,
                  g_args=dict(latent_size=512, mapping_layers=cfg["mapping_layers"], blur_filter=[1, 2, 1],
                              truncation_psi=cfg["truncation_psi"], truncation_cutoff=8),
                  d_args=dict(use_wscale=True, blur_filter=[1, 2, 1]),
                  g_opt_args=opt, d_opt_args=opt, loss="logistic", d_repeats=1, use_ema=True, ema_decay=0.999,
                  device=dev, act_dtype=act_dtype, data_parallel=dp,
                  use_graphs=(a.graphs != "off"))
    sg.deferred_losses = True                                # losses are not read inside the timed loop: no per-half-step host wait
    sg.gen.train(); sg.dis.train(); sg.gen_shadow.train()
    return sg


RING = 4                                                     # pre-generated synthetic batches resident in HBM ("a ring of 4")


def measure(sg, a, cfg, dev, B, steps, warmup, rank, world, want_graphs, stream_opts, layer_table=None, traffic_ok=False):
    """One measured block: launch-structure calibration, warm-up, surveyed step (single stream), timed region, roofline of the
    dominant kernel on ONE stream.  -> dict (the fields of the JSON line that depend on the batch size)."""
    from stylegan.pytorch_amd import native
    res, depth = cfg["resolution"], cfg["depth"]
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    reals = [torch.randn(B, res, res, 3, device=dev, generator=gen).permute(0, 3, 1, 2) for _ in range(RING)]  # NHWC storage
    lats = [torch.randn(B, 512, device=dev, generator=gen) for _ in range(RING)]
    import random
    random.seed(1234)                                        # same mixing cutoffs on every rank

    def step(i):
        z, x = lats[i % RING], reals[i % RING]
        sg.optimize_discriminator(z, x, depth, a.alpha)
        sg.optimize_generator(z, x, depth, a.alpha)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def max_over_ranks(values):
        """element-wise MAX of a list of floats over all ranks (a device tensor over RCCL; host memory in the gloo dry run)"""
        t = torch.tensor(values, dtype=torch.float64, device="cpu" if a.dry_run_ranks_on_one_gpu else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return [float(v) for v in t]

    # Launch structure by measurement on THIS box (setup, not warmup).  Two knobs: hipGraph replay of each half-iteration
    # vs stream launches, and the step's extra streams (fake branch of the D step on an auxiliary stream / weight gradients on
    # a side stream).  Replay takes the host out of the loop but costs the ROCm runtime more per node; extra streams overlap
    # the latency-bound low-resolution kernels but cost host time per fork -- which combination wins depends on whether this
    # box's host keeps up with the GPU.  Every candidate runs the same arithmetic (the parity tests pin graph vs eager and
    # multi- vs single-stream); best of two interleaved 4-step rounds each.
    def timed(n):
        """-> (ms per step, host enqueue ms per step)"""
        barrier(); torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(n):
            step(i)
        t_host = time.perf_counter() - t
        torch.cuda.synchronize(); barrier()
        return (time.perf_counter() - t) / n * 1e3, t_host / n * 1e3

    def apply(c):
        sg.use_graphs, sg.aux_stream, sg.param_stream = c

    cands = [(g, ax, pr) for g in want_graphs for (ax, pr) in stream_opts if not (g and (ax, pr) == (False, True))]
    HOST_MARGIN = 1.25        # a mode whose host enqueue time is within 25 % of its step time is one host hiccup away from
    #                           being host-bound for the whole timed region (measured, round 4: eager launches calibrated at 14.2 ms
    #                           with 13.9 ms of host time, then ran a 20.0 ms timed region on the same box where replay held 14.8):
    #                           it is ranked by max(step time, 1.25 x host time)
    calib, best = {}, None
    if len(cands) > 1:
        times = {c: float("inf") for c in cands}
        hosts = {c: float("inf") for c in cands}
        dead = set()
        for c in cands:                                      # graphs: two eager calls, then the capture
            apply(c)
            for i in range(3 if c[0] else 1):
                step(i)
            if c[0] and not sg.use_graphs:                   # a failed capture falls back to eager for good: the candidate stays
                dead.add(c)                                  # at +inf (on every rank after the MAX below: the keys never diverge)
        for _ in range(2):
            for c in [c for c in cands if c not in dead]:
                apply(c)
                timed(1)
                t, h = timed(4)
                if t < times[c]:
                    times[c], hosts[c] = t, h
        if world > 1:                                        # one decision for all ranks: the slowest rank's time per candidate
            keys = sorted(times)
            tt = max_over_ranks([times[k] for k in keys] + [hosts[k] for k in keys])
            times = dict(zip(keys, tt[:len(keys)]))
            hosts = dict(zip(keys, tt[len(keys):]))
        best = min(times, key=lambda c: max(times[c], HOST_MARGIN * hosts[c]))
        calib = {("graph" if g else "eager") + f"_aux{int(ax)}_side{int(pr)}": {"ms_per_step": round(t, 3), "host_ms_per_step": round(hosts[(g, ax, pr)], 3)}
                 for (g, ax, pr), t in sorted(times.items()) if t != float("inf")}
    else:
        best = cands[0]
        apply(best)
        for i in range(3 if best[0] else 1):
            step(i)
        if best[0] and not sg.use_graphs:
            best = (False,) + best[1:]
    apply(best)
    graphs = best[0]
    for i in range(warmup):
        step(i)

    def single_stream(fn):
        """Run ``fn`` with the step on ONE stream, eagerly (every launch through the library's hook, no kernel sharing the GPU with
        another: per-kernel durations are the kernel's own)."""
        keep = (sg.use_graphs, sg.aux_stream, sg.param_stream)
        apply((False, False, False))
        try:
            return fn()
        finally:
            apply(keep)

    # Roofline leg, part 1 (untimed): ONE surveyed single-stream step with the library's per-launch profiler on every kernel, to
    # find the dominant kernel (largest total time) and the per-layer table.
    survey = None
    if not a.no_kernel_timing:
        def do_survey():
            step(warmup)                                     # (the mode switch re-packs nothing; one untimed step settles allocations)
            torch.cuda.synchronize()
            native.prof_start(1)
            step(warmup + 1)
            torch.cuda.synchronize()
            native.prof_start(0)
            return native.prof_records()
        survey = single_stream(do_survey)
        agg = {}
        for idx, (name, ms, fl, nb, desc) in enumerate(survey):
            e = agg.setdefault(name, [0.0, 0, idx])
            e[0] += ms; e[1] += 1
        # the dominant kernel: the instantiation with the largest total time AMONG those that have a roofline at all -- an instantiation
        # whose average launch sits within 2x of the launch floor of this step (5th percentile of the surveyed durations: ~6.6 us) is
        # bound by launch latency, not by HBM or MFMA (at batch 4 the 52 mapping / dense-head GEMMs of M = 4..8 rows, 8-14 us each,
        # tie with the 3x3 convolutions for the largest total from one survey to the next); they stay in `top_kernels_ms_per_step`
        durs = sorted(r[1] for r in survey)
        floor_ms = durs[max(0, int(0.05 * len(durs)) - 1)] if durs else 0.0
        ranked = {k: v for k, v in agg.items() if v[0] / v[1] > 2.0 * floor_ms} or agg
        dom_name, (dom_ms, dom_n, dom_idx) = max(ranked.items(), key=lambda kv: kv[1][0])
        step(warmup + 2)                                     # back in the timed mode before the clock starts
    import gc
    gc.collect(); gc.disable()                              # no collector pause inside the timed region
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + 3 + i)
    t_enq = time.perf_counter() - t0                        # host time to enqueue the region (launch-bound if ~= dt)
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        dt = max_over_ranks([dt])[0]

    roof = None
    if survey is not None:
        # part 2: the dominant kernel's launches bracketed by HIP events (on their launch stream, inside libsgx_hip.so) in a
        # single-stream eager re-run of the timed steps: the same launches as in the timed region, each alone on the GPU -- the
        # number `rocprofv3 --kernel-trace --stats` of `bench.py --graphs off --streams 00` reproduces (profiles/)
        def rerun():
            native.prof_start(2, dom_idx)
            for i in range(min(steps, 4)):
                step(warmup + 3 + i)
            torch.cuda.synchronize()
            native.prof_start(0)
            return native.prof_records()
        recs = single_stream(rerun)
        graphs = graphs and all(g.graph is not None for g in sg._step_graphs.values())   # False if a capture fell back
        assert recs and all(r[0] == dom_name for r in recs)
        ms = sum(r[1] for r in recs); fl = sum(r[2] for r in recs); nb = sum(r[3] for r in recs)
        peak_f, peak_b = PEAK[a.dtype], HBM_PEAK
        if fl / peak_f >= nb / peak_b:                       # which roof the kernel's algorithmic work sits under
            bound, achieved, peak, unit = "mfma", fl / (ms * 1e-3) / 1e12, peak_f / 1e12, "TFLOP/s"
        else:
            bound, achieved, peak, unit = "hbm", nb / (ms * 1e-3) / 1e9, peak_b / 1e9, "GB/s"
        per_kernel = sorted(((k, v[0], v[1]) for k, v in agg.items()), key=lambda kv: -kv[1])

        def roof_row(desc, name, n, t_ms, f, nbytes):
            """One (layer, kernel) row of the surveyed step under its own roof: a single `frac` for an instantiation hides
            that it serves shapes from HBM-bound to latency-bound."""
            us = t_ms * 1e3 / n
            mf = f / peak_f >= nbytes / peak_b
            ach = (f / n / us / 1e6) if mf else (nbytes / n / us / 1e3)
            pk = (peak_f / 1e12) if mf else (peak_b / 1e9)
            return {"layer": desc, "kernel": name.split("(")[0].replace("void ", ""), "launches": n, "avg_us": round(us, 1),
                    "bound": "mfma" if mf else "hbm", "achieved": round(ach, 1), "unit": "TFLOP/s" if mf else "GB/s",
                    "frac": round(ach / pk, 4) if (f or nbytes) else None}
        by_layer = {}
        for name, t, f, n, desc in survey:
            L = by_layer.setdefault((desc, name), [0, 0.0, 0.0, 0.0])
            L[0] += 1; L[1] += t; L[2] += f; L[3] += n
        rows = sorted(by_layer.items(), key=lambda kv: -kv[1][1])
        dom_layers = [roof_row(d, nm, *v) for (d, nm), v in rows if nm == dom_name]
        top_layers = [roof_row(d, nm, *v) for (d, nm), v in rows[:12]]
        executed_flops = sum(r[2] for r in survey)            # what the kernels of one step actually execute (their own notes)
        algorithmic_bytes = sum(r[3] for r in survey)         # ... and the bytes their algorithms must move (operands once)
        traffic, traffic_src = None, None
        if traffic_ok:
            # HBM bytes per launch of this instantiation from the committed PMC passes of this same workload (rocprofv3
            # cannot run inside the benchmark): tools/gpu_final3.sh -> tools/pmc_traffic.py, corrected as the guide prescribes
            import glob
            for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_bf16_b{B}.json")), reverse=True):   # newest round first
                ent = json.load(open(pmc))["kernels"].get(dom_name)
                if ent:
                    traffic, traffic_src = ent["hbm_bytes_per_launch"], "profiles/" + os.path.basename(pmc)
                    break
        roof = {"bound": bound, "kernel": dom_name, "launches": len(recs), "avg_us": ms * 1e3 / len(recs),
                "selection": f"largest total time in the surveyed step among instantiations averaging more than 2x the launch floor ({floor_ms * 1e3:.1f} us)",
                "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak if (fl or nb) else None, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": nb / len(recs), "flops_per_launch": fl / len(recs),
                "avg_us_note": "HIP events around each launch on its stream: ~3 us more than rocprofv3's kernel duration on 8-20 us launches "
                               "(profiles/*_kernel_stats.csv holds the profiler's figure for the same command)",
                "library_kernels_ms_per_step": round(sum(v[0] for v in agg.values()), 3),
                "library_launches_per_step": len(survey),
                "algorithmic_gbytes_per_step": round(algorithmic_bytes / 1e9, 2),
                "events_over": "single-stream eager re-run of the timed steps (each launch alone on the GPU; the timed region is "
                               + ("hipGraph replay" if graphs else "eager") + f", aux stream {int(bool(sg.aux_stream))}, side stream {int(bool(sg.param_stream))})",
                "top_kernels_ms_per_step": {k: round(t, 3) for k, t, _ in per_kernel[:8]},
                "layers": dom_layers, "top_layers": top_layers,
                "layers_note": "per (layer, kernel) of ONE surveyed eager step on a single stream, every launch bracketed by HIP "
                               "events: a launch's duration is the kernel's own"}
        if layer_table and rank == 0:
            with open(layer_table, "w") as fh:
                fh.write("layer\tkernel\tcalls_per_step\tavg_us\tTFLOP/s\tGB/s(algorithmic)\tms_per_step\n")
                for (desc, name), (n, t, f, nbytes) in rows:
                    us = t * 1e3 / n
                    fh.write(f"{desc}\t{name}\t{n}\t{us:.1f}\t{f / n / us / 1e6:.1f}\t{nbytes / n / us / 1e3:.0f}\t{t:.3f}\n")

    imgs = B * world * steps
    value = imgs / dt
    out = {"value": value, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
           "batch_per_gpu": B, "global_batch": B * world, "input_ring": RING,
           "host_enqueue_ms_per_step": t_enq / steps * 1e3,
           "hip_graphs": bool(graphs), "aux_stream": bool(sg.aux_stream), "side_stream": bool(sg.param_stream),
           # SURVEY 8d asks for a median over per-iteration syncs; this is the MEAN of one region of `steps` iterations with the
           # losses deferred and a single final synchronize (a per-iteration sync would stall the launch queue every 16 ms)
           "timing": "mean over one timed region: barrier + synchronize, `steps` iterations enqueued back to back (losses deferred), synchronize + barrier",
           "launch_mode_calibration": calib or None,
           # useful-work convention: the reference step's algorithmic conv+GEMM FLOPs per image, whatever
           # the kernels execute; beside it the FLOPs the kernels really execute (one D(real) forward instead of two, no
           # D weight gradients in the G step, 4x4 stride-2 instead of 3x3 + pool) from their own per-launch notes
           "useful_tflops": value * cfg["flops_per_img"] / 1e12,
           "mfma_frac_of_step": value * cfg["flops_per_img"] / (PEAK[a.dtype] * world)}
    if roof:
        out["executed_tflops"] = executed_flops / (dt / steps) / 1e12 * world
        out["executed_frac_of_mfma_peak"] = executed_flops / (dt / steps) / PEAK[a.dtype]
        out["executed_gflop_per_img"] = executed_flops / B / 1e9
        out["roofline"] = roof
    del reals, lats
    return out


LINE_LIMIT = 4096        # the driver keeps ~8 KB of stdout: the LAST line must sit wholly inside it (round 5's 21 KB line did not parse)
ROOF_KEEP = ("bound", "kernel", "launches", "avg_us", "achieved", "peak", "unit", "frac", "traffic",
             "algorithmic_bytes_per_launch", "flops_per_launch")


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def _short_kernel(name):
    return name.split("(")[0].replace("void ", "") if isinstance(name, str) else name


def compact_roofline(roof):
    if not roof:
        return None
    out = {k: _r(roof.get(k)) for k in ROOF_KEEP}
    out["kernel"] = _short_kernel(out["kernel"])
    return out


def compact_block(blk):
    """The scalars of one measured block (no tables)."""
    keep = ("value", "unit", "steps", "warmup", "ms_per_step", "batch_per_gpu", "global_batch", "host_enqueue_ms_per_step", "hip_graphs",
            "useful_tflops", "mfma_frac_of_step", "executed_tflops", "executed_frac_of_mfma_peak")
    out = {k: _r(blk[k]) for k in keep if k in blk}
    roof = blk.get("roofline")
    if roof:
        for k in ("library_launches_per_step", "library_kernels_ms_per_step", "algorithmic_gbytes_per_step"):
            if k in roof:
                out[k] = roof[k]
        out["roofline"] = compact_roofline(roof)
    return out


def compact_line(full, detail_path=None):
    """The ONE JSON line the driver parses: the contract's keys + `roofline` + `cpu_baseline` + a scalar `b32` block, under LINE_LIMIT
    characters.  Everything else (per-layer tables, calibration, the sweep rows, the whole configs[1] block, notes) lives in the detail
    file named by `detail`."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
           "config")
    out = {k: _r(full[k]) for k in top if k in full}
    for k, v in compact_block(full).items():
        out.setdefault(k, v)
    cb = full.get("cpu_baseline")
    if cb is not None:
        out["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
        if cb.get("value") is None and "error" in cb:
            out["cpu_baseline"]["error"] = str(cb["error"])[:160]
    if full.get("b32"):
        b = compact_block(full["b32"])
        b["config"] = full["b32"].get("config")
        out["b32"] = b
    sw = full.get("sweep_top_depths")
    if sw:
        out["sweep_top_depths"] = ({"error": str(sw["error"])[:120]} if "error" in sw else
                                   [{"depth": r["depth"], "batch": r["batch"], "img_per_s": r["img_per_s"]} for r in sw.get("rows", [])])
    f128 = full.get("ffhq128_fp32_b64")
    if f128:
        out["ffhq128_fp32_b64"] = ({"error": str(f128["error"])[:120]} if "error" in f128 else
                                   {"value": _r(f128.get("value")), "unit": "img/s", "ms_per_step": _r(f128.get("ms_per_step")),
                                    "roofline_frac": _r((f128.get("roofline") or {}).get("frac"))})
    if "rccl_group_of_one" in full:
        out["rccl_group_of_one"] = True
    if detail_path:
        out["detail"] = detail_path
    line = json.dumps(out)
    for k in ("ffhq128_fp32_b64", "sweep_top_depths", "detail"):          # never let an optional field push the line over
        if len(line) < LINE_LIMIT:
            break
        out.pop(k, None)
        line = json.dumps(out)
    if len(line) >= LINE_LIMIT and "cpu_baseline" in out:
        out["cpu_baseline"].pop("sample", None)
        line = json.dumps(out)
    assert len(line) < LINE_LIMIT, len(line)
    return line


def write_detail(full, path):
    """The full record (every table) beside the line: a file, never stdout."""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def main():
    a = parse()
    cfg = CONFIGS[a.config]
    if a.cpu_baseline_child:
        out = None
        try:
            out = cpu_baseline_reference(cfg)                # the reference itself where its sources exist
        except Exception as e:                               # noqa: BLE001 -- fall back to the port, say why
            sys.stderr.write(f"reference CPU leg failed ({type(e).__name__}: {e}); timing the port\n")
        out = out or cpu_baseline(cfg)
        if a.config == "ffhq1024":
            out["reference_container"] = REFERENCE_CONTAINER
        out["timing"] = "wall clock of ONE iteration (D step + G step) after model construction, no warm-up iteration"
        print(json.dumps(out))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE {world} (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if a.dry_run_ranks_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from stylegan.pytorch_amd import native
    native.lib()
    dp = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dry_run_ranks_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        from stylegan.pytorch_amd.dist import DataParallelGroup
        dp = DataParallelGroup()
    elif a.rccl_group_of_one:
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1, device_id=dev)
        from stylegan.pytorch_amd.dist import DataParallelGroup
        dp = DataParallelGroup(force_collectives=True)

    flush_c_stdio()
    sg = make_stylegan(a, cfg, dev, dp)
    B, res, depth = a.batch_per_gpu, cfg["resolution"], cfg["depth"]
    if a.sweep:
        assert world == 1 and a.config == "ffhq1024", "--sweep: one GPU, the 1024 model (BASELINE configs[4] on one device)"
        sg.use_graphs = a.graphs == "on"          # eager unless asked (--graphs on: hipGraph replay, what StyleGAN.train runs since round 5)
        rows = sweep(sg, a, cfg, dev)
        worst = min(rows, key=lambda r: r["frac_of_mfma_peak"])
        total_img = sum(r["batch"] * a.sweep_steps for r in rows)
        total_s = sum(r["ms_per_step"] * a.sweep_steps for r in rows) / 1e3
        print(json.dumps({"metric": "img/s per progressive depth, full G+D train step, 4x4 -> 1024x1024 sweep", "unit": "img/s",
                          "value": total_img / total_s, "n_gpus": 1, "steps": a.sweep_steps, "warmup": 2, "higher_is_better": True,
                          "dtype": a.dtype, "data": "synthetic (full-resolution real batches, down-sampled by the step as the reference does)",
                          "config": {"workload": "ffhq1024 model, depth index 0..8, batch sizes " + str([r["batch"] for r in rows])
                                                 + " (reference config.py:40-41), fade-in over the first half of each depth's timed iterations, "
                                                   "style mixing on, " + ("hipGraph replay" if a.graphs == "on" else "eager launches")},
                          "value_note": "images of all depths / time of all depths (equal iteration counts per depth: not the reference's epoch mix)",
                          "worst_depth": {"depth": worst["depth"], "frac_of_mfma_peak": worst["frac_of_mfma_peak"]},
                          "sweep": rows}))
        return
    want_graphs = [False] if a.graphs == "off" else ([True] if a.graphs == "on" else [False, True])
    stream_opts = {"auto": [(True, True), (False, True), (False, False)], "11": [(True, True)], "01": [(False, True)], "00": [(False, False)]}[a.streams]
    headline = a.config == "ffhq1024" and a.dtype == "bf16"
    blk = measure(sg, a, cfg, dev, B, a.steps, a.warmup, rank, world, want_graphs, stream_opts, layer_table=a.layer_table,
                  traffic_ok=headline and B in (4, 32))
    b32 = None
    if headline and world == 1 and B != 32 and not a.no_b32:
        # second measured block: the north-star target configuration (BASELINE.json: >= 40 % of the bf16 MFMA peak at batch 32
        # on ONE MI355X), same model, same weights object, 8 timed steps.  Stream launches only: at 80 ms per step the host is
        # never the limit, and a captured graph of this size would hold a second private copy of every activation.
        torch.cuda.synchronize(); torch.cuda.empty_cache()
        sg._step_graphs.clear()
        b32 = measure(sg, a, cfg, dev, 32, a.b32_steps, 2, rank, world, [False],
                      [(True, True), (False, False)] if a.streams == "auto" else stream_opts,
                      layer_table=(a.layer_table + ".b32.tsv") if a.layer_table else None, traffic_ok=True)

    if rank == 0:
        out = {"metric": "img/s full G+D train step, 1024x1024 depth-9 bf16" if headline
               else f"img/s full G+D train step, {a.config} {a.dtype}",
               "value": blk["value"], "unit": "img/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": blk["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": a.dtype, "data": f"synthetic (ring of {RING} pre-generated batches resident in HBM)",
               "config": {"workload": f"{a.config}: StyleGAN {res}x{res}, progressive depth index {depth}, logistic+R1, "
                                      f"alpha {a.alpha}, batch {B}/GPU, global batch {B * world}",
                          "global_batch": B * world, "parallelism": f"dp{world}"}}
        for k, v in blk.items():
            if k not in out and k != "roofline":
                out[k] = v
        if "roofline" in blk:
            out["roofline"] = blk["roofline"]
        if a.rccl_group_of_one:
            out["rccl_group_of_one"] = ("data-parallel code path over an RCCL group of size 1: bucketed all-reduce launched for real (no peer to exchange "
                                        "with), replay split into [graph | all-reduce | update]; NOT a scaling measurement")
        if b32 is not None:
            b32["config"] = {"workload": f"{a.config}: same model, batch 32 on one GPU (the north-star target configuration)"}
            out["b32"] = b32
        if headline and world == 1 and not a.no_extras and B == 4:
            out["sweep_top_depths"] = extra_sweep_top_depths(sg, a, cfg, dev)
            out["ffhq128_fp32_b64"] = extra_ffhq128_fp32_b64()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(a.config, a.cpu_baseline_timeout)
        detail = None if a.no_detail_file else write_detail(out, os.path.join(ROOT, a.detail_file))
        line = compact_line(out, detail)
    # The JSON line must be the LAST line of the job's stdout.  RCCL prints its version banner through C stdio, which is fully buffered
    # when stdout is a pipe: left alone it comes out when the process exits, AFTER Python's own output (seen with --rccl-group-of-one:
    # `tail -1` was "Librccl path : ...").  So: tear the group down, flush C stdio on every rank, let the other ranks' processes end,
    # and only then print.
    if world > 1 or a.rccl_group_of_one:
        if world > 1:
            torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        if world > 1:
            time.sleep(1.0)
        sys.stdout.flush()
        print(line, flush=True)


if __name__ == "__main__":
    main()
