#!/usr/bin/env python3
"""Headline benchmark of the backdoored-diffusion hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one optimiser step of the poisoned fine-tune loop at per-GPU batch 128 on synthetic CIFAR10-shaped data
(config #2 of BASELINE.json): GPU trigger stamping -> q-sample+backdoor target -> UNet fwd -> MSE -> UNet bwd ->
(RCCL all-reduce) -> clip + Adam.  Inputs (uint8 dataset, weights) are resident in HBM before the timed region.
After the timed region the same process measures 1000-step DDPM sampling (second half of the metric), times the
MFMA kernels of one extra step with HIP events (roofline) and, on rank 0 at N=1, the CPU oracle (cpu_baseline).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_IMG = 12.444          # SURVEY.md §8d (2 x MAC), DDPM-CIFAR10-32 UNet
TRAIN_GFLOP_PER_IMG = 3 * FWD_GFLOP_PER_IMG
PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= f32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16, 32 cycles / instruction)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (BASELINE config #2)")
    ap.add_argument("--sample-steps", type=int, default=1000)
    ap.add_argument("--sample-images", type=int, default=1024,
                    help="per-GPU images of the DDPM sampling leg, in chunks of --batch (SURVEY 8d: 1024 = 8 chunks of 128; 0 = skip)")
    ap.add_argument("--sample-streams", type=int, default=4,
                    help="chunks denoised concurrently, each on its own stream / HIP graph (pipelines.sample_concurrent); 1: one after the other")
    ap.add_argument("--mode", choices=("all", "train", "sample"), default="all",
                    help="train: training legs only (no sampler) / sample: sampler only -- so that a rocprofv3 summary covers ONE dispatch population")
    ap.add_argument("--cpu-batch", type=int, default=128, help="batch of the CPU-oracle training baseline (SURVEY 8d: 128)")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-f32 leg (profiling runs: the summary then covers the default arithmetic only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU-oracle baseline leg")
    ap.add_argument("--no-f16", action="store_true",
                    help="skip the opt-in mixed-precision legs (conv_math = 'f16': the reference's GPU arithmetic is fp16 autocast; reported beside the headline)")
    ap.add_argument("--cpu-true-1000", action="store_true",
                    help="CPU leg: also one TRUE 1000-step DDPM run of one image on the oracle and on the product with the same noise (about a minute of CPU)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-ddp-path", action="store_true",
                    help="skip the extra leg that times the MULTI-RANK schedule on this one GPU (see --ddp-path)")
    ap.add_argument("--ddp-path", action="store_true",
                    help="(default at --gpus 1; the flag makes a failure fatal) also time the K steps on the schedule rank 0 of an N-GPU run executes: "
                         "an RCCL communicator of size 1, gradient-bucket hooks fired from the explicit backward, graphed micro-step off -> "
                         "ddp_path_ms_per_step beside ms_per_step")
    ap.add_argument("--graph-step", type=int, default=None, choices=(0, 1),
                    help="1: replay q-sample + forward + loss + backward of the training step as ONE HIP graph (trainer.GraphedMicroStep; single process only); "
                         "default: the library default (graphs only under gradient accumulation)")
    ap.add_argument("--eager-sample", action="store_true", help="also time the sampling loop with eager launches instead of the HIP-graph replay")
    ap.add_argument("--no-secondary", action="store_true", help="skip the DDIM-50 / DPM-Solver++-20 / UniPC-20 legs (counter-collection passes)")
    ap.add_argument("--serial-wgrad", action="store_true",
                    help="keep the weight gradients on the launch stream for the WHOLE run (profiling: every kernel's duration is then its own; "
                         "the default overlaps them with the backward pass on a side stream)")
    ap.add_argument("--config", choices=("cifar10", "celebahq256", "ldm64"), default="cifar10",
                    help="cifar10 (default, the driver's line): BASELINE config #2 + DDPM-1000 sampling.  celebahq256 / ldm64: SECONDARY lines for BASELINE "
                         "configs #4 / #5 (DDPM-CELEBA-HQ-256 UNet at 3x256x256, LDM-CELEBA-HQ UNet on 3x64x64 latents; per-GPU batch 8 = 64 over 8 GPUs): "
                         "training step + one denoising step, analytic FLOPs from the architecture, dominant kernel and its roofline fraction")
    ap.add_argument("--conv-math", choices=("bf16x3", "f32"), default=None,
                    help="arithmetic of the eligible 3x3 convolutions (default: the library default, bf16x3 split products)")
    return ap.parse_args()


def cpu_baseline(batch: int = 128, n_steps: int = 3, state_dict=None, gpu_side=None, true_1000: bool = False):
    """The oracle (plain torch fp32 = what the reference's diffusers path executes on a CPU) on the host cores:
    fwd + bwd + clip(1.0) + Adam on a bounded sample of the same workload.

    With `state_dict` (the product network's current weights) and `gpu_side` (callables that run the product path on
    the same inputs) the same CPU work doubles as the parity gate printed with the number (BASELINE.md §2;
    workload VillanDiffusion.py:1141-1176): the oracle's FIRST step at batch `batch` gives loss and gradient norm on
    identical weights / batch / timesteps / noise, its 20 DDPM steps on 8 images give the denoised images."""
    from oracle.loss_ref import LossFnRef, SDE_VP
    from oracle.schedulers_ref import DDPMSchedulerRef, sample_loop
    from oracle.unet_ref import UNet2DModelRef
    # cores this process may run on, capped at 16: on the 256-thread GPU host oneDNN gets SLOWER with more threads for these 32x32
    # convolutions -- the B = 128 oracle step takes 6.0 s on 16 threads, 6.9 s on 32, 11.0 s on 64, 21 s on 128 and more than 15 minutes on
    # 256 (profiles/r06_cpu_threads.txt, tools/cpu_thread_sweep.py); the line states the count it used (`cores`)
    ncpu = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(max(1, ncpu))
    torch.manual_seed(0)
    net = UNet2DModelRef()
    if state_dict is not None:
        net.load_state_dict(state_dict)
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)
    sched = DDPMSchedulerRef()
    lf = LossFnRef(sched, SDE_VP, psi=1, solver_type="sde")

    def make_batch(b, seed):
        g = torch.Generator().manual_seed(seed)
        x0 = torch.rand(b, 3, 32, 32, generator=g) * 2 - 1
        R = torch.rand(b, 3, 32, 32, generator=g) * 2 - 1
        R[: b - max(1, b // 10)] = 0            # poison_rate 0.1: ~10 % of the rows carry a trigger residual, the rest are clean (R = 0)
        t = torch.randint(0, 1000, (b,), generator=g)
        return x0, R, t, torch.randn(x0.shape, generator=g)

    def step(b, seed):
        x0, R, t, eps = make_batch(b, seed)
        loss = lf.p_loss(net, x0, R, t, noise=eps)
        opt.zero_grad()
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)     # returns the norm BEFORE clipping
        opt.step()
        return float(loss), float(gnorm)

    parity = None
    if gpu_side is not None:                     # the product runs the very same step-0 inputs first (weights not yet touched by `opt`)
        parity = gpu_side["train"](*make_batch(batch, 1000))
    warm = torch.optim.Adam(net.parameters(), lr=0.0)        # warm-up (allocator, oneDNN primitives) that leaves the weights alone
    opt, opt_real = warm, opt
    step(2, 1)
    opt = opt_real
    t0 = time.perf_counter()
    for i in range(n_steps):
        l_, g_ = step(batch, 1000 + i)
        if i == 0 and parity is not None:
            parity["loss_step0_rel_err"] = abs(parity.pop("loss") - l_) / abs(l_)
            parity["grad_norm_rel_err"] = abs(parity.pop("grad_norm") - g_) / g_
            if "loss_f16" in parity:          # the opt-in mixed-precision mode on the same inputs (reported, not gated: never the headline arithmetic)
                parity["f16_mode_loss_rel_err"] = abs(parity.pop("loss_f16") - l_) / abs(l_)
                parity["f16_mode_grad_norm_rel_err"] = abs(parity.pop("grad_norm_f16") - g_) / g_
    dt_train = time.perf_counter() - t0
    train_ips = n_steps * batch / dt_train
    # sampling: 8 images x the last 20 steps of the 1000-step DDPM chain, extrapolated x50 (stated in "sample")
    if state_dict is not None:
        net.load_state_dict(state_dict)          # the denoising parity runs on the product's weights, not on the 3-steps-older ones
    init = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        t0 = time.perf_counter()
        x = sample_loop(net, sched, init.clone(), 1000, generator=torch.Generator().manual_seed(5), start_from=980)
        dt_s = time.perf_counter() - t0
    sample_ips = 8 / (dt_s * 50.0)
    if parity is not None:
        img_ref = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
        img, ts_equal = gpu_side["sample"](init, 5, 980, sched.timesteps)
        parity["denoised_max_rel_err"] = float(abs(img - img_ref).max() / abs(img_ref).max())
        parity["timestep_indices_bit_exact"] = bool(ts_equal)
        parity = {k: (v if isinstance(v, bool) else float(f"{v:.3e}")) for k, v in parity.items()}
        parity["gates"] = {"loss_step0_rel_err": 1e-5, "grad_norm_rel_err": 1e-4, "denoised_max_rel_err": 1e-3}
        parity["pass"] = bool(parity["loss_step0_rel_err"] <= 1e-5 and parity["grad_norm_rel_err"] <= 1e-4
                              and parity["denoised_max_rel_err"] <= 1e-3 and parity["timestep_indices_bit_exact"])
        parity["what"] = (f"product vs CPU oracle on identical weights and inputs: one fwd+bwd at batch {batch} (loss, gradient norm); "
                          "8 images x the last 20 steps of DDPM-1000 with the same CPU-generator noise (denoised images); DDPM-1000 timestep table")
    full = None
    if true_1000:          # SURVEY 8d "one true 1000-step run at n=1 if budget allows": the whole chain, product vs oracle on the same noise (opt-in: ~1 min of CPU)
        init1 = torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(12))
        with torch.no_grad():
            t0 = time.perf_counter()
            x1 = sample_loop(net, sched, init1.clone(), 1000, generator=torch.Generator().manual_seed(6), start_from=0)
            dt1 = time.perf_counter() - t0
        full = {"images_per_sec": round(1.0 / dt1, 5), "seconds": round(dt1, 2)}
        if gpu_side is not None:
            ref1 = (x1 / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
            img1, _ = gpu_side["sample"](init1, 6, 0, sched.timesteps)
            full["denoised_max_rel_err_after_1000_steps"] = float(f"{abs(img1 - ref1).max() / abs(ref1).max():.3e}")
    cpu = {"value": round(train_ips, 3), "unit": "train images/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n_steps} optimiser steps at batch {batch} (fwd+bwd+clip+Adam, fp32 torch CPU oracle); "
                     f"sampling: 8 images x 20 DDPM steps extrapolated x50",
           "sample_ddpm1000_images_per_sec": round(sample_ips, 5), "host_cpus": os.cpu_count(),
           "affinity": len(os.sched_getaffinity(0))}
    if full is not None:
        cpu["true_1000_step_run_n1"] = full
    return cpu, parity


# ---- roofline: per-launch HIP-event timing (on the launch stream) of one extra training step and of one denoising step ----
def is_split(kname, kind="mfma"):     # split-precision kernels run on the bf16 MFMA (3 instructions per algorithmic product term)
    from villandiffusion_amd.flops import SPLIT_PRECISION_FAMILIES, is_split_precision
    if kind != "mfma":                # HBM-bound passes: no matrix peak is quoted for them
        return any(f in kname for f in SPLIT_PRECISION_FAMILIES)
    return is_split_precision(kname)  # raises for a matrix kernel nobody classified


def peak_of(kname):
    return PEAK_BF16_MFMA_TFLOPS if is_split(kname) else PEAK_F32_MFMA_TFLOPS


def summarise(rec):
    """Per kernel symbol: launches, summed event time, algorithmic TFLOP/s and GB/s, and the fraction of the BINDING roofline.  Which
    resource binds is decided on what the hardware does: the matrix pipe's occupancy is the EXECUTED MFMA work (three instructions per
    split-precision product term) over the dtype's dense peak, HBM's is algorithmic bytes over 8 TB/s; `frac` is then quoted on
    ALGORITHMIC work against that resource's peak (frac_mfma = algorithmic TFLOP/s / peak, frac_hbm = GB/s / 8000)."""
    agg = {}
    for r_ in rec:
        a = agg.setdefault(r_["name"], {"n": 0, "flops": 0.0, "bytes": 0.0, "ms": 0.0, "kind": r_["kind"]})
        a["n"] += 1
        a["flops"] += r_["flops"]
        a["bytes"] += r_["bytes"]
        a["ms"] += r_["e0"].elapsed_time(r_["e1"])
    rows = []
    for k, v in agg.items():
        tf = v["flops"] / v["ms"] / 1e9 if v["ms"] > 0 else 0.0
        gbs = v["bytes"] / v["ms"] / 1e6 if v["ms"] > 0 else 0.0
        split = is_split(k, v["kind"])
        f_mfma = tf / (PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS) if v["kind"] == "mfma" else 0.0
        f_hbm = gbs / PEAK_HBM_GBS
        rows.append({"kernel": k, "launches": v["n"], "ms": round(v["ms"], 3), "avg_us": round(1e3 * v["ms"] / v["n"], 1),
                     "tflops": round(tf, 2), "gbs": round(gbs, 1), "gflop": round(v["flops"] / 1e9, 1), "mbytes": round(v["bytes"] / 1e6, 1),
                     "frac_mfma": round(f_mfma, 4), "frac_mfma_executed": round((3 if split else 1) * f_mfma, 4), "frac_hbm": round(f_hbm, 4),
                     "bound": "hbm" if round(f_hbm, 4) >= round((3 if split else 1) * f_mfma, 4) else "mfma",
                     "frac": round(f_hbm if round(f_hbm, 4) >= round((3 if split else 1) * f_mfma, 4) else f_mfma, 4),
                     "mfma_peak": (PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS) if v["kind"] == "mfma" else None})
    return sorted(rows, key=lambda r: -r["ms"])



def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(args):
    """`python bench.py --gpus N` (N > 1, no torchrun around it): start N fresh rank processes -- one per GPU -- BEFORE this process makes any
    GPU call (it never does: it only waits), relay their output (rank 0 prints the one JSON line on the inherited stdout) and exit with the
    launcher's code.  The reference's multi-GPU entry is `--gpu "0,1,..."` (VillanDiffusion.py:241-244, 440)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus}: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.run(cmd, env=env).returncode)



def bench_secondary_config(args):
    """BASELINE configs #4 / #5 as secondary bench lines (never the driver's default): the fine-tune step of the 256x256 pixel UNet
    (reference model.py:706-776 `DDPM-CELEBA-HQ-256`, run_celeba_hq_script.py) or of the latent UNet (`LDM-CELEBA-HQ-256`,
    run_ldm_celeba_hq_script.py:10,68-70) at per-GPU batch 8 (global 64 over 8 GPUs), plus one no-grad denoising forward (UniPC-20 is 20 of
    them).  FLOPs are computed from the architecture (villandiffusion_amd.flops)."""
    from villandiffusion_amd import ops
    from villandiffusion_amd.flops import unet_forward_flops
    from villandiffusion_amd.loss import LossFn
    from villandiffusion_amd.model import DDPM_256_ARCH, LDM_CELEBA_UNET_ARCH
    from villandiffusion_amd.schedulers import DDPMScheduler
    from villandiffusion_amd.trainer import Trainer
    from villandiffusion_amd.unet import UNet2DModel
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1:
        dist.init_process_group(os.environ.get("VD_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    B = 8 if args.batch == 128 else args.batch
    if args.config == "celebahq256":
        arch, S, sde, tag = dict(DDPM_256_ARCH), 256, "SDE-VP", "DDPM-CELEBA-HQ-256 UNet (113.7 M), 3x256x256, STOP_SIGN_14->CAT recipe"
        sched = DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, clip_sample=False)
    else:
        arch, S, sde, tag = dict(LDM_CELEBA_UNET_ARCH), 64, "SDE-LDM", "LDM-CELEBA-HQ-256 UNet (274 M), 3x64x64 latents, GLASSES->CAT recipe"
        sched = DDPMScheduler(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear", clip_sample=False)
    arch.pop("in_channels", None), arch.pop("out_channels", None), arch.pop("sample_size", None)
    net = UNet2DModel(in_channels=3, out_channels=3, sample_size=S, **arch)
    net.reset_parameters(seed=0)
    if args.conv_math:
        net.conv_math = args.conv_math
    if args.serial_wgrad:
        net.wgrad_stream = False
        net.sc_stream = False                          # (and no shortcut launches on the auxiliary stream: every kernel's duration is its own)
    fwd_flop = unet_forward_flops(net)
    n_par = sum(p.numel() for p in net.parameters())
    lf = LossFn(sched, sde, psi=1, solver_type="sde")
    lf.noise_seed = 1234 + rank
    trainer = Trainer(net, lf, lr=6e-5, total_steps=10000, warmup_steps=500, grad_accum=1)
    g = torch.Generator(device=dev).manual_seed(rank)
    x0 = torch.rand((B, 3, S, S), device=dev, generator=g) * 2 - 1
    R = torch.zeros_like(x0)
    R[: max(1, B // 8)] = torch.rand((max(1, B // 8), 3, S, S), device=dev, generator=g) * 2 - 1
    tgen = torch.Generator(device=dev).manual_seed(100 + rank)

    def one_step():
        t = torch.randint(0, 1000, (B,), device=dev, generator=tgen)
        return trainer.train_step({"target": x0, "pixel_values": R}, t)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(max(2, args.warmup)):
        one_step()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    torch.cuda.synchronize()
    barrier()
    tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    ips = world * B * args.steps / dt
    log(f"{args.config}: {ips:.2f} img/s ({1e3 * dt / args.steps:.2f} ms/step at per-GPU batch {B}), loss {float(loss):.4f}")
    # one no-grad denoising forward (graph replay as the samplers run it)
    from villandiffusion_amd.pipelines import sampler_forward
    fwd_ms = None
    with torch.no_grad():
        fwd = sampler_forward(net, B)
        tq = torch.full((B,), 500.0, device=dev)
        for _ in range(2):
            fwd(x0, tq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fwd(x0, tq)
        torch.cuda.synchronize()
        fwd_ms = 1e3 * (time.perf_counter() - t0) / 5
    from villandiffusion_amd.pipelines import drop_sampler_graphs
    drop_sampler_graphs(net)
    kernels = None
    if not args.no_roofline:
        ws0, net.wgrad_stream = net.wgrad_stream, False
        sc0, net.sc_stream = net.sc_stream, False
        for _ in range(2):
            ops.profile_start()
            one_step()
            torch.cuda.synchronize()
            rec = ops.profile_stop()
        net.wgrad_stream, net.sc_stream = ws0, sc0
        kernels = summarise(rec)
    if rank == 0:
        split = net.conv_math == "bf16x3"
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        ms = 1e3 * dt / args.steps
        out = {"metric": "train imgs/sec (secondary line: BASELINE config #%d)" % (4 if args.config == "celebahq256" else 5),
               "value": round(ips, 3), "unit": "train images/s", "n_gpus": world, "steps": args.steps, "warmup": max(2, args.warmup),
               "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16x3/f32" if split else "f32", "data": "synthetic",
               "config": {"workload": f"{tag}; fine-tune step at per-GPU batch {B} ({sde}, Adam + clip 1.0)", "global_batch": B * world,
                          "image": f"3x{S}x{S}", "parallelism": f"dp{world}", "parameters": n_par},
               "forward_gflop_per_image": round(fwd_flop / 1e9, 2), "train_gflop_per_image": round(3 * fwd_flop / 1e9, 2),
               "train_tflops": round(ips * 3 * fwd_flop / 1e12, 2), f"train_frac_of_{'bf16' if split else 'f32'}_peak": round(ips * 3 * fwd_flop / 1e12 / (peak * world), 4),
               "denoise_forward_ms": None if fwd_ms is None else round(fwd_ms, 3),
               "denoise_forward_tflops": None if fwd_ms is None else round(B * fwd_flop / fwd_ms / 1e9, 2),
               "final_loss": round(float(loss), 5)}
        if kernels:
            mf = [k for k in kernels if k["mfma_peak"]]
            k0 = mf[0]
            sp = is_split(k0["kernel"])
            # HBM bytes per launch of the dominant kernel from this configuration's committed PMC passes (tools/collect_profiles_r06.sh:
            # FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc runs of this same command; rocprofv3 cannot run inside this process)
            traffic, traffic_src = None, None
            tag = 'cfg4' if args.config == 'celebahq256' else 'cfg5'
            pmc = next((f for f in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic_{tag}.json") for r in (6, 5)) if os.path.exists(f)), "")
            if pmc:
                with open(pmc) as f:
                    traffic = json.load(f)["kernels"].get(k0["kernel"].split("(+")[0].split("@")[0], {}).get("traffic_bytes_per_launch")
                traffic_src = os.path.relpath(pmc, ROOT) if traffic else None
            out["roofline"] = {"bound": k0["bound"], "kernel": k0["kernel"], "achieved": k0["tflops"] if k0["bound"] == "mfma" else k0["gbs"],
                               "peak": k0["mfma_peak"] if k0["bound"] == "mfma" else PEAK_HBM_GBS, "unit": "TFLOP/s" if k0["bound"] == "mfma" else "GB/s",
                               "frac": k0["frac"], "frac_executed": k0["frac_mfma_executed"], "executed_tflops": round((3 if sp else 1) * k0["tflops"], 2),
                               "traffic": traffic, "traffic_source": traffic_src, "launches_per_step": k0["launches"], "avg_launch_us": k0["avg_us"],
                               "ms_per_step": k0["ms"], "algorithmic_gflop_per_launch": round(k0["gflop"] / k0["launches"], 2),
                               "algorithmic_mbytes_per_launch": round(k0["mbytes"] / k0["launches"], 2)}
            out["top_kernels"] = [{"kernel": k["kernel"], "launches": k["launches"], "ms": k["ms"], "tflops": k["tflops"], "bound": k["bound"], "frac": k["frac"]}
                                  for k in kernels[:10]]
            out["profiled_kernels_ms"] = round(sum(k["ms"] for k in kernels), 2)
            for k in kernels[:14]:
                log(f"{k['ms']:8.3f} ms {k['launches']:4d}x {k['avg_us']:8.1f} us  {k['tflops']:7.1f} TF {k['bound']:4s} frac {k['frac']:.3f}  {k['kernel']}")
            path = os.environ.get("VD_BENCH_DETAIL")       # full per-kernel table (tools/update_profiles_r06.py joins it with the PMC passes)
            if path:
                with open(path, "w") as f:
                    json.dump(dict(out, train_step_kernels=kernels), f)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)
    if args.config != "cifar10":
        return bench_secondary_config(args)
    if int(os.environ.get("WORLD_SIZE", 1)) != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', 1)}: launch with "
                 f"`python bench.py --gpus N` or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    ndev = torch.cuda.device_count()
    dev_id = local_rank % max(1, ndev)                    # one GPU per rank on a real node (local_rank < ndev)
    rendezvous_only = os.environ.get("VD_BENCH_RENDEZVOUS_ONLY") == "1"      # launcher test on a box without GPUs: process group proof, no compute
    if not rendezvous_only:
        torch.cuda.set_device(dev_id)
    dev = torch.device("cpu") if rendezvous_only else torch.device("cuda", dev_id)
    ranks_seen = None
    if world > 1:
        # "nccl" IS RCCL on ROCm.  VD_BENCH_BACKEND=gloo exists only to exercise the multi-process path on a 1-GPU box.
        backend = os.environ.get("VD_BENCH_BACKEND", "nccl")
        rccl_log = None
        if backend == "nccl":
            # RCCL's own account of the communicator (rank count, transports): INFO lines of the INIT subsystem.  They go to a FILE per process --
            # RCCL would otherwise write them to stdout, where the one JSON line of this script has to stand alone -- and rank 0 quotes the
            # communicator lines in its stderr log and in the JSON (`process_group.rccl_init`).
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT")
            if "NCCL_DEBUG_FILE" not in os.environ:
                rccl_log = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"villan_rccl_rank{rank}_{os.getpid()}.log")
                os.environ["NCCL_DEBUG_FILE"] = rccl_log
        dist.init_process_group(backend, rank=rank, world_size=world)
        # proof that the collective really spans N ranks on N devices: every rank contributes (rank, device index, PCI bus id)
        pci = -1 if rendezvous_only else int(getattr(torch.cuda.get_device_properties(dev_id), "pci_bus_id", -1))
        mine = torch.tensor([rank, dev_id, pci, 1], device=dev, dtype=torch.int64)
        seen = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(seen, mine)
        tot = mine.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        ranks_seen = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_counted_by_all_reduce": int(tot[3]),
                      "rank_device_pci": [[int(v) for v in t_[:3]] for t_ in seen],
                      "distinct_devices": len({(int(t_[1]), int(t_[2])) for t_ in seen})}
        if rank == 0 and rccl_log and os.path.exists(rccl_log):
            with open(rccl_log, errors="replace") as f:
                lines = [ln.strip() for ln in f if "nranks" in ln or "Init COMPLETE" in ln or "comm 0x" in ln]
            ranks_seen["rccl_init"] = lines[:4]
        if rank == 0:
            print(f"[bench] process group: {ranks_seen}", file=sys.stderr, flush=True)
        assert ranks_seen["ranks_counted_by_all_reduce"] == world == ranks_seen["world_size"]
    if rendezvous_only:
        if rank == 0:
            print(json.dumps({"metric": "rendezvous only (VD_BENCH_RENDEZVOUS_ONLY=1): no compute ran", "value": None, "n_gpus": world,
                              "process_group": ranks_seen}))
        if world > 1:
            dist.destroy_process_group()
        return

    from villandiffusion_amd import ops
    from villandiffusion_amd.dataset import DatasetLoader
    from villandiffusion_amd.loss import LossFn
    from villandiffusion_amd.model import DDPM_32_ARCH
    from villandiffusion_amd.pipelines import DDPMPipeline
    from villandiffusion_amd.schedulers import DDPMScheduler
    from villandiffusion_amd.trainer import Trainer
    from villandiffusion_amd.unet import UNet2DModel

    B = args.batch
    torch.manual_seed(0)
    net = UNet2DModel(in_channels=3, out_channels=3, sample_size=32, **DDPM_32_ARCH)
    net.reset_parameters(seed=0)                          # identical replicas on every rank
    if args.conv_math:
        net.conv_math = args.conv_math
    if args.serial_wgrad:
        net.wgrad_stream = False
        net.sc_stream = False                          # (and no shortcut launches on the auxiliary stream: every kernel's duration is its own)
    sched = DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02, clip_sample=False)
    loss_fn = LossFn(sched, "SDE-VP", psi=1, solver_type="sde")
    loss_fn.noise_seed = 1234 + rank
    dsl = DatasetLoader("SYNTHETIC-CIFAR10", root=ROOT, batch_size=B, seed=0)
    dsl.set_poison("BOX_14", "HAT", poison_rate=0.1).prepare_dataset(mode="FIXED")
    n_batches = (len(dsl) + B * world - 1) // (B * world)
    trainer = Trainer(net, loss_fn, lr=2e-4, total_steps=n_batches * 50, warmup_steps=500, grad_accum=1,
                      graph_micro_step=None if args.graph_step is None else bool(args.graph_step))
    from villandiffusion_amd.trainer import shard_indices
    ids = shard_indices(len(dsl), 0, rank, world, seed=0)
    if os.environ.get("VD_BENCH_HOST_POSITIONS", "0") != "1":
        ids = ids.to(dev)                                    # device positions: a batch then needs no host -> device copy (dataset.make_batch)
    tgen = torch.Generator(device=dev).manual_seed(100 + rank)

    def one_step(i):
        s = (i * B) % (len(ids) - B)
        batch = dsl.make_batch(ids[s:s + B], full=False)
        t = torch.randint(0, 1000, (B,), device=dev, generator=tgen)
        return trainer.train_step(batch, t)

    def barrier():
        if world > 1:
            dist.barrier()

    do_train, do_sample = args.mode in ("all", "train"), args.mode in ("all", "sample") and args.sample_images > 0
    log(f"setup done (world={world}, B={B}, mode={args.mode}); warm-up {args.warmup} steps")
    train_ips = dt = final_loss = None
    host_submit_ms = host_sync_ms = None
    exact = None
    f16 = None
    ddp_path = None
    if do_train:
        for i in range(args.warmup):
            one_step(i)
            if i == 0:
                torch.cuda.synchronize()
                log("first step done")
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = one_step(args.warmup + i)
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
        train_ips = world * B * args.steps / dt
        final_loss = float(loss)
        log(f"train: {train_ips:.1f} img/s ({1e3 * dt / args.steps:.2f} ms/step), loss {final_loss:.4f}")
        # host side of one step: time until the last launch of a step has been SUBMITTED (queue drained first) vs until it has run
        sub = []
        for i in range(3):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            one_step(args.warmup + args.steps + i)
            h1 = time.perf_counter()
            torch.cuda.synchronize()
            sub.append((h1 - h0, time.perf_counter() - h0))
        host_submit_ms, host_sync_ms = 1e3 * min(a for a, _ in sub), 1e3 * min(b for _, b in sub)
        log(f"host: one step submitted in {host_submit_ms:.2f} ms, finished in {host_sync_ms:.2f} ms (drained queue)")

        # ---- the MULTI-RANK schedule on this one GPU (reference VillanDiffusion.py:440, 1161-1166): what rank 0 of an 8-GPU run
        #      executes minus the wire time -- a size-1 RCCL communicator, one all-reduce per gradient bucket fired from the backward pass ----
        if world == 1 and not args.no_ddp_path:
            # RCCL prints its version banner / INFO lines through C stdio: they must not land on the stdout that carries the ONE JSON line.
            # Its debug file takes what honours NCCL_DEBUG_FILE; fd 1 points at stderr for the length of the leg and C stdio is flushed before it
            # is restored (the banner sat in libc's buffer until exit and came out AFTER the JSON line on the first run of this leg).
            import ctypes
            import tempfile
            if "NCCL_DEBUG_FILE" not in os.environ:
                os.environ["NCCL_DEBUG_FILE"] = os.path.join(tempfile.gettempdir(), "vd_bench_rccl_ddp_path.%h.%p.log")
            sys.stdout.flush()
            saved_fd1 = os.dup(1)
            os.dup2(2, 1)
            try:
                import socket
                if not dist.is_initialized():
                    with socket.socket() as sk:
                        sk.bind(("127.0.0.1", 0))
                        port = sk.getsockname()[1]
                    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                    os.environ.setdefault("MASTER_PORT", str(port))
                    dist.init_process_group(os.environ.get("VD_BENCH_BACKEND", "nccl"), rank=0, world_size=1)
                trainer.ddp_path, net.bucket_ready_hook = True, trainer._bucket_ready
                for i in range(3):
                    one_step(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    one_step(args.warmup + i)
                torch.cuda.synchronize()
                ddp_ms = 1e3 * (time.perf_counter() - t0) / args.steps
                ddp_path = {"ddp_path_ms_per_step": round(ddp_ms, 3), "vs_single_process": round(ddp_ms / (1e3 * dt / args.steps), 4),
                            "what": "same K steps, process group of size 1 (backend %s), 4 bucket all-reduces fired from the explicit backward on the "
                                    "weight-gradient side stream, graphed micro-step off" % dist.get_backend()}
                log(f"ddp path: {ddp_ms:.2f} ms/step ({ddp_path['vs_single_process']:.3f} x the single-process step)")
            except Exception as e:  # noqa: BLE001 -- a box without a usable RCCL must not cost the headline line
                if args.ddp_path:
                    raise
                ddp_path = {"ddp_path_ms_per_step": None, "error": f"{type(e).__name__}: {e}"[:200]}
            finally:
                trainer.ddp_path, net.bucket_ready_hook = False, None
                trainer._pending, trainer._sync_now = [], False
                if dist.is_initialized():
                    dist.destroy_process_group()
                try:
                    ctypes.CDLL(None).fflush(None)
                except OSError:
                    pass
                os.dup2(saved_fd1, 1)
                os.close(saved_fd1)

        # ---- the same K steps with every contraction on the exact-f32 MFMA (reported beside the headline, not as `value`) ----
        if net.conv_math != "f32" and not args.no_exact:
            mode0, net.conv_math = net.conv_math, "f32"
            for i in range(2):
                one_step(i)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                one_step(args.warmup + i)
            torch.cuda.synchronize()
            barrier()
            te = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
            exact = {"train_images_per_sec": round(world * B * args.steps / float(te), 2), "ms_per_step": round(1e3 * float(te) / args.steps, 3),
                     "note": "same run, net.conv_math = 'f32': all contractions on v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s peak)"}
            net.conv_math = mode0
            log(f"exact-f32 mode: {exact}")

        # ---- ... and in the opt-in mixed-precision mode (one f16 product per term in the full-size 3x3 / 1x1 forward and input-gradient
        # contractions, loss scaling in the trainer): the reference trains this config under fp16 autocast (VillanDiffusion.py:260-264).
        # Reported beside the headline, never as `value`. ----
        if net.conv_math == "bf16x3" and not args.no_f16:
            net.conv_math = "f16"
            for i in range(3):
                one_step(i)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                one_step(args.warmup + i)
            torch.cuda.synchronize()
            barrier()
            tf = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tf, op=dist.ReduceOp.MAX)
            f16 = {"train_images_per_sec": round(world * B * args.steps / float(tf), 2), "ms_per_step": round(1e3 * float(tf) / args.steps, 3),
                   "loss_scale": trainer.loss_scale,
                   "note": "same run, net.conv_math = 'f16': 3x3 / 1x1 forward + input gradients as single f16 products (v_mfma_f32_16x16x32_f16), weight "
                           "gradients and attention as in the default arithmetic, f32 master weights and accumulation, loss scaling"}
            net.conv_math = "bf16x3"
            log(f"f16 mixed-precision mode: {f16}")

    # ---- 1000-step DDPM sampling (second half of the metric), embarrassingly parallel ----
    sample_ips, sample_s, secondary, sample_eager = None, None, None, None
    pipe = DDPMPipeline(net, sched)
    n_img = args.sample_images
    if do_sample:
        sched.device_rng_seed = 99 + rank                 # throughput mode: Philox noise fused into the step kernel
        init = torch.empty((n_img, 3, 32, 32), device=dev)
        ops.randn(init, 7 + rank, 0)
        pp = torch.empty((n_img, 32, 32, 3), device=dev)

        def sample_all(steps):
            chunks = torch.split(init, B)
            if args.sample_streams > 1 and len(chunks) > 1 and net.sampler_graph:
                outs = pipe.sample_concurrent(list(chunks), num_inference_steps=steps, n_streams=args.sample_streams)
            else:
                outs = [pipe(batch_size=len(c), init=c, num_inference_steps=steps, return_tensor=True) for c in chunks]
            final = torch.cat(outs)
            ops.postprocess(final, pp, 0.5, 0.5, 0.0, 1.0, True)      # (x/2+0.5).clamp(0,1), NHWC; PNG encode excluded
            return final

        def timed_sampling():
            with torch.no_grad():                             # short warm-up of the inference path: one graph capture per distinct chunk size
                for nb_ in sorted({len(c) for c in torch.split(init, B)}):      # (a remainder chunk must not capture inside the timed region)
                    pipe(batch_size=nb_, init=init[:nb_], num_inference_steps=1000, start_from=995, return_tensor=True)
                if args.sample_streams > 1 and n_img > B and net.sampler_graph:   # ... nor the second stream's graph
                    pipe.sample_concurrent(list(torch.split(init, B))[:args.sample_streams], num_inference_steps=3, n_streams=args.sample_streams)
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sample_all(args.sample_steps)
            torch.cuda.synchronize()
            barrier()
            ts = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            return float(ts)

        sample_s = timed_sampling()
        sample_ips = world * n_img / sample_s
        assert bool(torch.isfinite(pp).all())
        log(f"sample: {sample_ips:.4f} img/s ({sample_s:.2f} s for {n_img} images x {args.sample_steps} steps; hip graph = {net.sampler_graph})")
        if net.sampler_graph and args.eager_sample:      # the same loop launched eagerly (what the graph replay replaces), reported beside it
            net.sampler_graph = False
            sched._rng_offset = 0
            s_eager = timed_sampling()
            net.sampler_graph = True
            sample_eager = {"images_per_sec": round(world * n_img / s_eager, 4), "seconds": round(s_eager, 2)}
            log(f"sample (eager launches): {sample_eager}")
        if net.conv_math == "bf16x3" and not args.no_f16 and n_img >= 4 * B:      # the same loop in the f16 mode on 4 chunks (one round of streams)
            net.conv_math = "f16"
            init_full, init = init, init[:4 * B]
            s16 = timed_sampling()
            f16 = dict(f16 or {}, sample_ddpm1000_images_per_sec=round(world * len(init) / s16, 4))
            init = init_full
            net.conv_math = "bf16x3"
            log(f"f16 mode sampling: {f16.get('sample_ddpm1000_images_per_sec')} img/s")
        sched.device_rng_seed = None
        # secondary samplers of SURVEY.md §8d (configs #2-#4): same network, same init, whole loop incl. post-processing
        from villandiffusion_amd.pipelines import DDIMPipeline, PNDMPipeline
        from villandiffusion_amd.schedulers import DDIMScheduler, DPMSolverMultistepScheduler, UniPCMultistepScheduler
        secondary = {}
        for tag, mk, pcls, nst in (() if args.no_secondary else
                                   (("ddim50", lambda: DDIMScheduler(clip_sample=False), DDIMPipeline, 50),
                                    ("dpm_solver_pp_o2_20", lambda: DPMSolverMultistepScheduler(), PNDMPipeline, 20),
                                    ("unipc20", lambda: UniPCMultistepScheduler(), PNDMPipeline, 20))):
            p2 = pcls(net, mk())
            conc = args.sample_streams > 1 and n_img > B and net.sampler_graph
            c0 = init if conc else init[:B]

            def run2():
                if conc:
                    return torch.cat(p2.sample_concurrent(list(torch.split(c0, B)), num_inference_steps=nst, n_streams=args.sample_streams))
                return p2(batch_size=len(c0), init=c0, num_inference_steps=nst, return_tensor=True)

            run2()                                                                                 # warm-up
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o = run2()
            ops.postprocess(o, pp[:len(c0)], 0.5, 0.5, 0.0, 1.0, True)
            torch.cuda.synchronize()
            barrier()
            tsec = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tsec, op=dist.ReduceOp.MAX)
            secondary[tag] = round(world * len(c0) / float(tsec), 2)
        log(f"secondary samplers (img/s): {secondary}")

    roofline, kernels, sample_kernels = None, None, None
    if not args.no_roofline and do_train:
        ws0, net.wgrad_stream = net.wgrad_stream, False   # per-launch durations: no weight-gradient kernels running beside the bracketed launch
        sc0, net.sc_stream = net.sc_stream, False         # ... and no shortcut 1x1 on the auxiliary stream
        for _ in range(2):                       # every rank runs the step (it contains the all-reduce); rank 0 reports
            ops.profile_start()
            one_step(10_000)
            torch.cuda.synchronize()
            rec = ops.profile_stop()
        net.wgrad_stream, net.sc_stream = ws0, sc0
        barrier()
        if rank == 0:
            kernels = summarise(rec)
    if not args.no_roofline and do_sample:
        g0, net.sampler_graph = net.sampler_graph, False          # events bracket eager launches
        sched.device_rng_seed = 99 + rank
        xs = init[:B]
        for _ in range(2):
            ops.profile_start()
            pipe(batch_size=len(xs), init=xs, num_inference_steps=1000, start_from=999, return_tensor=True)
            torch.cuda.synchronize()
            rec_s = ops.profile_stop()
        sched.device_rng_seed = None
        net.sampler_graph = g0
        if rank == 0:
            sample_kernels = summarise(rec_s)
    pmc_file = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")) if os.path.exists(f)), "")

    def traffic_of(kname):      # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process)
        try:
            with open(pmc_file) as f:
                return json.load(f)["kernels"].get(kname.split("(+")[0].split("@")[0], {}).get("traffic_bytes_per_launch")
        except OSError:
            return None

    def roof_entry(k, rule):
        split = is_split(k["kernel"], "mfma" if k["mfma_peak"] else "hbm")
        tr = traffic_of(k["kernel"])
        if k["bound"] == "mfma":
            e = {"bound": "mfma", "kernel": k["kernel"], "achieved": k["tflops"], "peak": k["mfma_peak"], "unit": "TFLOP/s", "frac": k["frac_mfma"],
                 "executed_tflops": round((3 if split else 1) * k["tflops"], 2), "frac_executed": k["frac_mfma_executed"]}
        else:
            e = {"bound": "hbm", "kernel": k["kernel"], "achieved": k["gbs"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": k["frac_hbm"],
                 "tflops": k["tflops"], "frac_mfma_executed": k["frac_mfma_executed"]}
        e.update({"traffic": tr, "traffic_source": (os.path.relpath(pmc_file, ROOT) + " (separate rocprofv3 --pmc passes)") if tr else None,
                  "launches_per_step": k["launches"], "avg_launch_us": k["avg_us"], "ms_per_step": k["ms"],
                  "algorithmic_mbytes_per_launch": round(k["mbytes"] / k["launches"], 2), "algorithmic_gflop_per_launch": round(k["gflop"] / k["launches"], 2),
                  "selection_rule": rule})
        return e

    roofline_by_flops = None
    src = kernels if kernels else sample_kernels
    if src:
        mf = [k for k in src if k["mfma_peak"]]
        # dominant kernel: the kernel TEMPLATE (all its instantiations: one source kernel) with the largest total time in the profiled step, and of
        # it the instantiation with the largest total time -- a symbol that collects every shape of a small family (the 1x1 kernel: 35 launches)
        # would otherwise outrank the instantiations of the family that takes three times as long (the 3x3 convolution)
        fam = {}
        for k in mf:
            fam[k["kernel"].split("<")[0].split("(")[0]] = fam.get(k["kernel"].split("<")[0].split("(")[0], 0.0) + k["ms"]
        top_fam = max(fam, key=fam.get)
        dom = next(k for k in mf if k["kernel"].split("<")[0].split("(")[0] == top_fam)
        roofline = roof_entry(dom, "the MFMA kernel template with the largest TOTAL TIME in the profiled step (HIP events around every launch; all its "
                                   f"instantiations: {round(fam[top_fam], 3)} ms), and its instantiation with the largest total time")
        roofline["family_ms_per_step"] = round(fam[top_fam], 3)
        roofline["all_mfma_kernels_ms"] = round(sum(k["ms"] for k in mf), 2)
        roofline["note"] = ("per-launch durations are taken with the weight gradients on the launch stream (bench.py --serial-wgrad is the same setting for "
                            "a whole run, used for the rocprofv3 summaries); the timed region overlaps them with the backward pass on a side stream")
        roofline_by_flops = roof_entry(max(mf, key=lambda r: r["gflop"]), "the MFMA kernel symbol carrying the most algorithmic FLOPs in the profiled step")

    log("roofline leg done")

    # ---- CPU oracle on the host cores (rank 0, N = 1): the reported baseline AND the parity gates printed with the number ----
    def gpu_train_side(x0, R, t, eps):
        out_ = {}
        gs0 = loss_fn.grad_scale
        for tag, mode in (("", net.conv_math),) + ((("_f16", "f16"),) if (net.conv_math == "bf16x3" and not args.no_f16) else ()):
            mode0, net.conv_math = net.conv_math, mode
            loss_fn.grad_scale = 4096.0 if mode == "f16" else 1.0          # the trainer's loss scale, divided out below
            net.zero_grad()
            batch = {"target": x0.to(dev), "pixel_values": R.to(dev)}
            loss = loss_fn.p_loss_by_keys(batch, net, "target", "pixel_values", t.to(dev), noise=eps.to(dev))
            loss.backward()
            out_["loss" + tag] = float(loss)
            out_["grad_norm" + tag] = float(torch.sqrt((net.flat_grad.double() ** 2).sum())) / loss_fn.grad_scale
            net.zero_grad()
            net.conv_math = mode0
        loss_fn.grad_scale = gs0
        return out_

    def gpu_sample_side(init_cpu, seed, start_from, ref_timesteps):
        sch = DDPMScheduler(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02)
        o = DDPMPipeline(net, sch)(batch_size=len(init_cpu), generator=torch.Generator().manual_seed(seed), init=init_cpu,
                                   num_inference_steps=1000, start_from=start_from, output_type=None)
        return o.images, torch.equal(sch.timesteps.cpu(), ref_timesteps) and sch.timesteps.dtype == torch.int64

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        cpu, parity = cpu_baseline(args.cpu_batch, args.cpu_steps, state_dict=sd,
                                   gpu_side={"train": gpu_train_side, "sample": gpu_sample_side}, true_1000=args.cpu_true_1000)
        log(f"cpu baseline: {cpu}")
        log(f"parity: {parity}")

    if rank == 0:
        split = net.conv_math == "bf16x3"
        ms = None if dt is None else 1e3 * dt / args.steps
        slim = ("bound", "kernel", "achieved", "peak", "unit", "frac", "executed_tflops", "frac_executed", "traffic", "launches_per_step",
                "avg_launch_us", "ms_per_step", "algorithmic_gflop_per_launch", "algorithmic_mbytes_per_launch", "all_mfma_kernels_ms")

        def slim_roof(r):
            return None if r is None else {k: r[k] for k in slim if k in r}

        pg = None if ranks_seen is None else {k: v for k, v in ranks_seen.items() if k != "rccl_init"}
        top = None if not kernels else [{"kernel": k["kernel"], "launches": k["launches"], "ms": k["ms"], "bound": k["bound"], "frac": k["frac"]}
                                        for k in kernels[:5]]
        non_mfma_ms = None if not kernels else round(sum(k["ms"] for k in kernels if not k["mfma_peak"]), 3)
        out = {
            "metric": "train imgs/sec + 1000-step DDPM sample imgs/sec, CIFAR10 bs128",
            "value": None if train_ips is None else round(train_ips, 2), "unit": "train images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None if ms is None else round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3/f32" if split else "f32", "data": "synthetic",
            # `workload` stays under 128 characters (the driver's record truncates there); the details live in the keys beside it
            "config": {"workload": "DDPM-CIFAR10-32 poisoned fine-tune B=%d (BOX_14->HAT pr0.1 SDE-VP) + DDPM-%d sampling %d img/GPU (%dx%d, %d streams)"
                                   % (B, args.sample_steps, args.sample_images, -(-args.sample_images // B) if B else 0, B,
                                      min(args.sample_streams, max(1, -(-args.sample_images // B)))),
                       "train": "BASELINE config #2: poison_rate 0.1, trigger BOX_14, target HAT, SDE-VP psi=1 sde, per-GPU batch %d, Adam + clip 1.0" % B,
                       "sampling": "%d-step DDPM, %d images/GPU in chunks of %d, %d chunks at a time on their own streams / HIP graphs"
                                   % (args.sample_steps, args.sample_images, B, min(args.sample_streams, max(1, -(-args.sample_images // B)))),
                       "global_batch": B * world, "image": "3x32x32", "parallelism": f"dp{world}", "mode": args.mode},
            "exact_f32_mode": None if exact is None else {k: exact[k] for k in ("train_images_per_sec", "ms_per_step")},
            "f16_mode": None if f16 is None else {k: f16[k] for k in ("train_images_per_sec", "ms_per_step", "sample_ddpm1000_images_per_sec", "loss_scale") if k in f16},
            "sample_ddpm1000_images_per_sec": None if sample_ips is None else round(sample_ips, 4),
            "sample_seconds": None if sample_s is None else round(sample_s, 2),
            "sample_hip_graph": bool(net.sampler_graph), "sample_streams": min(args.sample_streams, -(-args.sample_images // B)) if args.sample_images > B else 1,
            "sample_secondary_images_per_sec": secondary,
            "train_tflops": None if train_ips is None else round(train_ips * TRAIN_GFLOP_PER_IMG / 1e3, 2),
            "sample_tflops": None if sample_ips is None else round(sample_ips * FWD_GFLOP_PER_IMG * args.sample_steps / 1e3, 2),
            "final_loss": None if final_loss is None else round(final_loss, 5),
            "host_submit_ms_per_step": None if host_submit_ms is None else round(host_submit_ms, 3),
            "ddp_path_ms_per_step": None if not ddp_path else ddp_path.get("ddp_path_ms_per_step"),
            "ddp_path": None if not ddp_path else {k: v for k, v in ddp_path.items() if k in ("vs_single_process", "error")},
            "non_mfma_ms_per_step": non_mfma_ms,
            "roofline": slim_roof(roofline), "roofline_largest_flops": slim_roof(roofline_by_flops), "top_kernels": top,
            "cpu_baseline": cpu, "parity": parity, "process_group": pg,
        }
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS          # whole-job fractions against the peak of the arithmetic that ran
        key = "bf16" if split else "f32"
        if train_ips is not None:
            out[f"train_frac_of_{key}_peak"] = round(train_ips * TRAIN_GFLOP_PER_IMG / 1e3 / (peak * world), 4)
        if sample_ips is not None:
            out[f"sample_frac_of_{key}_peak"] = round(sample_ips * FWD_GFLOP_PER_IMG * args.sample_steps / 1e3 / (peak * world), 4)
        # everything that does not fit a short line (per-kernel tables of the training and the sampler step, selection rules, notes, RCCL's
        # communicator lines) goes to a side file and to stderr: the driver keeps only the tail of stdout, and the ONE line must survive it
        detail = dict(out)
        detail.update({"roofline": roofline, "roofline_largest_flops": roofline_by_flops, "train_step_kernels": kernels,
                       "sampler_step_kernels": sample_kernels, "exact_f32_mode": exact, "f16_mode": f16, "ddp_path": ddp_path, "sample_eager_launches": sample_eager,
                       "process_group": ranks_seen,
                       "dtype_note": ("f32 tensors and accumulation; 3x3 / 1x1 convolutions and the attention contractions (forward, input and weight "
                                      "gradients) as hi*hi + hi*lo + lo*hi over bf16 halves on the bf16 MFMA (~1e-5 of exact f32 per contraction; the "
                                      "reference trains this config under fp16 autocast); stride-2 convolutions, linears, conv_in / conv_out on the "
                                      "exact f32 MFMA") if split else "every contraction on the f32-input MFMA (exact f32)"})
        path = os.environ.get("VD_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(detail, f)
            out["detail_file"] = os.path.relpath(path, ROOT)
        except OSError as e:
            log(f"detail file not written: {e}")
        for tag, rows in (("train", kernels), ("sampler", sample_kernels)):
            for k in (rows or [])[:12]:
                log(f"{tag:7s} {k['ms']:8.3f} ms {k['launches']:4d}x {k['avg_us']:8.1f} us  {k['bound']:4s} frac {k['frac']:.3f}  {k['kernel']}")
        line = json.dumps(out)
        assert len(line) < 4096, len(line)
        print(line, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
