#!/usr/bin/env python3
"""MI355X-native driver with the reference's command line (VillanDiffusion.py:74-116), modes, config overlay rules,
result-directory naming and JSON side files (args.json / config.json / sampling.json), so existing recipes
(`--mode train --dataset CIFAR10 --batch 128 --epoch 50 --poison_rate 0.1 --trigger BOX_14 --target HAT
--ckpt DDPM-CIFAR10-32 --fclip o -o --gpu 0`) run unchanged.  `--gpu K` runs on device K; `--gpu "0,1,2,3"` starts one rank process per listed
GPU (gpu_plan below; the reference drives nn.DataParallel over them, :240-245, 440); `torchrun --nproc-per-node N VillanDiffusion.py ...` works too.

Not reproduced on purpose: module-level side effects at import (the reference parses argv and calls wandb.init on import,
:323), swallowed training exceptions (:1189-1191), nn.DataParallel (:440).  `measure` writes the clean / backdoor PNG
sets, MSE + SSIM against the target and the FID of the clean set against the dataset into score.json (reference key naming); FID
runs on the HIP InceptionV3 (villandiffusion_amd/inception.py) and needs the published pt_inception weights as a LOCAL file
($VILLAN_FID_WEIGHTS): without it the score is recorded as null.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from dataclasses import asdict, dataclass, field
from typing import List, Optional

MODE_TRAIN, MODE_RESUME, MODE_SAMPLING, MODE_MEASURE, MODE_TRAIN_MEASURE = "train", "resume", "sampling", "measure", "train+measure"
TASK_GENERATE = "generate"
TASK_UNPOISONED_DENOISE, TASK_POISONED_DENOISE = "unpoisoned_denoise", "poisoned_denoise"
TASK_UNPOISONED_INPAINT_BOX, TASK_POISONED_INPAINT_BOX = "unpoisoned_inpaint_box", "poisoned_inpaint_box"
TASK_UNPOISONED_INPAINT_LINE, TASK_POISONED_INPAINT_LINE = "unpoisoned_inpaint_line", "poisoned_inpaint_line"
# the reference's DEFAULT_* constants (VillanDiffusion.py:20-60), pinned by tests/golden/driver_defaults.json
DEFAULT = dict(project="Default", batch=512, eval_max_batch=1500, epoch=50, learning_rate=None, clean_rate=1.0, poison_rate=0.007,
               ext_poison_rate=0.0, trigger="SM_BOX", target="CORNER", dataset_load_mode="FIXED", solver_type="sde", sde_type="SDE-VP",
               psi=1, ve_scale=1.0, vp_scale=1.0, gpu="0", ckpt=None, overwrite=False, postfix="", fclip="w", save_image_epochs=5,
               save_model_epochs=5, is_save_all_model_epochs=False, sample_ep=None, result=".", dataset="CIFAR10", sched=None,
               ddim_eta=None, infer_steps=1000, infer_start=0, inpaint_mul=1.0, task=TASK_GENERATE, R_trigger_only=False)
# per-mode whitelists of options that may be overridden from the command line (reference :17-72)
NOT_MODE_TRAIN = {"sample_ep"}
NOT_MODE_TRAIN_MEASURE = {"sample_ep"}
MODE_RESUME_OPTS = {"project", "task", "sched", "ddim_eta", "infer_steps", "mode", "gpu", "ckpt"}
MODE_SAMPLING_OPTS = {"project", "mode", "eval_max_batch", "gpu", "fclip", "ckpt", "sample_ep", "sched", "ddim_eta", "infer_steps",
                      "infer_start", "inpaint_mul", "task"}
MODE_MEASURE_OPTS = MODE_SAMPLING_OPTS
IGNORE_ARGS = {"overwrite", "is_save_all_model_epochs", "R_trigger_only"}


def parse_args(argv: Optional[List[str]] = None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description=globals()["__doc__"])
    a = p.add_argument
    a("--project", "-pj", type=str); a("--mode", "-m", type=str, required=True,
                                       choices=[MODE_TRAIN, MODE_RESUME, MODE_SAMPLING, MODE_MEASURE, MODE_TRAIN_MEASURE])
    # option strings, types and choices as the reference declares them (VillanDiffusion.py:77-111; pinned by
    # tests/golden/cli_flags.json); --dataset additionally accepts the synthetic stand-ins of this build
    a("--task", "-t", type=str, choices=[TASK_GENERATE, TASK_UNPOISONED_DENOISE, TASK_POISONED_DENOISE, TASK_UNPOISONED_INPAINT_BOX,
                                         TASK_POISONED_INPAINT_BOX, TASK_UNPOISONED_INPAINT_LINE, TASK_POISONED_INPAINT_LINE])
    a("--dataset", "-ds", type=str, choices=["MNIST", "CIFAR10", "CELEBA", "CELEBA-HQ", "CELEBA-HQ-LATENT_PR05", "CELEBA-HQ-LATENT",
                                             "SYNTHETIC-CIFAR10", "SYNTHETIC-CELEBA-HQ"])
    a("--sched", "-sc", type=str, choices=["DDPM-SCHED", "DDIM-SCHED", "DPM_SOLVER_PP_O1-SCHED", "DPM_SOLVER_O1-SCHED",
                                           "DPM_SOLVER_PP_O2-SCHED", "DPM_SOLVER_O2-SCHED", "DPM_SOLVER_PP_O3-SCHED", "DPM_SOLVER_O3-SCHED",
                                           "UNIPC-SCHED", "PNDM-SCHED", "DEIS-SCHED", "HEUN-SCHED", "LMSD-SCHED", "SCORE-SDE-VE-SCHED",
                                           "EDM-VE-SDE-SCHED", "EDM-VE-ODE-SCHED"])
    a("--ddim_eta", "-det", type=float); a("--infer_steps", "-is", type=int)
    a("--infer_start", "-ist", type=float); a("--inpaint_mul", "-im", type=float)
    a("--batch", "-b", type=int); a("--eval_max_batch", "-eb", type=int); a("--epoch", "-e", type=int)
    a("--learning_rate", "-lr", type=float); a("--clean_rate", "-cr", type=float); a("--poison_rate", "-pr", type=float)
    a("--ext_poison_rate", "-epr", type=float); a("--trigger", "-tr", type=str); a("--target", "-ta", type=str)
    a("--dataset_load_mode", "-dlm", type=str, choices=["FIXED", "FLEX", "EXTEND", "NONE"])
    a("--solver_type", "-solt", type=str, choices=["sde", "ode"]); a("--sde_type", "-sdet", type=str, choices=["SDE-VP", "SDE-VE", "SDE-LDM"])
    a("--psi", "-ps", type=float); a("--ve_scale", "-ves", type=float); a("--vp_scale", "-vps", type=float)
    a("--gpu", "-g", type=str); a("--ckpt", "-c", type=str); a("--overwrite", "-o", action="store_true")
    a("--R_trigger_only", "-trigonly", action="store_true"); a("--postfix", "-p", type=str); a("--fclip", "-fc", type=str, choices=["w", "o"])
    a("--save_image_epochs", "-sie", type=int); a("--save_model_epochs", "-sme", type=int)
    a("--is_save_all_model_epochs", "-isame", action="store_true"); a("--sample_ep", "-se", type=int); a("--result", "-res", type=str)
    return p.parse_args(argv)


@dataclass
class TrainingConfig:
    project: str = DEFAULT["project"]; batch: int = DEFAULT["batch"]; epoch: int = DEFAULT["epoch"]
    eval_max_batch: int = DEFAULT["eval_max_batch"]; learning_rate: Optional[float] = None
    clean_rate: float = 1.0; poison_rate: float = DEFAULT["poison_rate"]; ext_poison_rate: float = 0.0
    trigger: str = DEFAULT["trigger"]; target: str = DEFAULT["target"]; dataset_load_mode: str = "FIXED"
    solver_type: str = "sde"; sde_type: str = "SDE-VP"; psi: float = 1; ve_scale: float = 1.0; vp_scale: float = 1.0
    gpu: str = "0"; ckpt: Optional[str] = None; overwrite: bool = False; postfix: str = ""; fclip: str = DEFAULT["fclip"]
    save_image_epochs: int = DEFAULT["save_image_epochs"]; save_model_epochs: int = 5; is_save_all_model_epochs: bool = False
    sample_ep: Optional[int] = None; result: str = DEFAULT["result"]; dataset: str = "CIFAR10"; sched: Optional[str] = None
    ddim_eta: Optional[float] = None; infer_steps: int = 1000; infer_start: int = 0; inpaint_mul: float = 1.0
    task: str = TASK_GENERATE; R_trigger_only: bool = False; mode: str = MODE_TRAIN
    eval_sample_n: int = 16; measure_sample_n: int = 10000; measure_inpaint_sample_n: int = 1024; batch_32: int = 128; batch_256: int = 64
    gradient_accumulation_steps: int = 1; learning_rate_32_scratch: float = 2e-4; learning_rate_256_scratch: float = 2e-5
    lr_warmup_steps: int = 500; mixed_precision: str = "no"; seed: int = 0; dataset_path: str = "datasets"
    ckpt_dir: str = "ckpt"; data_ckpt_dir: str = "data.ckpt"; ep_model_dir: str = "epochs"
    clip: bool = False; output_dir: str = ""; ckpt_path: Optional[str] = None; data_ckpt_path: Optional[str] = None
    extra: dict = field(default_factory=dict)


def naming_fn(c: TrainingConfig) -> str:
    """reference :186-190."""
    add_on = f"_{c.postfix}" if c.postfix else ""                    # (the sampler is NOT part of the name: pinned by driver_defaults.json)
    return (f"res_{c.ckpt}_{c.dataset}_ep{c.epoch}_{c.solver_type}_c{c.clean_rate}_p{c.poison_rate}_epr{c.ext_poison_rate}_"
            f"{c.trigger}-{c.target}_psi{c.psi}_lr{c.learning_rate}_vp{c.vp_scale}_ve{c.ve_scale}{add_on}")


def setup(args: argparse.Namespace, preflight: bool = False) -> TrainingConfig:
    """Config overlay of reference :200-321.  preflight (what main() passes): a --dataset that cannot be read from local files fails
    before the run directory and its side files are created (the reference would download it instead)."""
    cfg = TrainingConfig()
    for k, v in json.loads(os.environ.get("VILLAN_CFG_OVERRIDES", "{}")).items():     # test hook: shrink e.g. measure_sample_n (10000)
        setattr(cfg, k, v)
    given = {k: v for k, v in vars(args).items() if v is not None and not (isinstance(v, bool) and v is False)}
    mode = args.mode
    if mode in (MODE_RESUME, MODE_SAMPLING, MODE_MEASURE):
        base = os.path.join(given.get("result", cfg.result), given["ckpt"]) if not os.path.isdir(given.get("ckpt", "")) else given["ckpt"]
        with open(os.path.join(base, "args.json")) as f:
            for k, v in json.load(f).items():
                if v is not None and hasattr(cfg, k):
                    setattr(cfg, k, v)
        allowed = {MODE_RESUME: MODE_RESUME_OPTS, MODE_SAMPLING: MODE_SAMPLING_OPTS, MODE_MEASURE: MODE_MEASURE_OPTS}[mode]
        for k, v in given.items():
            if k in allowed:
                setattr(cfg, k, v)
            elif k not in IGNORE_ARGS and k != "result":
                raise NotImplementedError(f"Argument: {k}={v} isn't supported in mode: {mode}")
        cfg.output_dir = base
    else:
        banned = NOT_MODE_TRAIN if mode == MODE_TRAIN else NOT_MODE_TRAIN_MEASURE
        for k, v in given.items():
            if k in banned:
                raise NotImplementedError(f"Argument: {k}={v} isn't supported in mode: {mode}")
            setattr(cfg, k, v)
    cfg.mode = mode
    cfg.clip = cfg.fclip == "w"                                        # :252-258
    # f32 tensors everywhere by default (reference :260-264: fp16 autocast + GradScaler for SDE-VP / SDE-LDM).  VILLAN_MIXED_PRECISION=fp16 opts
    # into the library's mixed-precision arithmetic (UNet2DModel.conv_math = "f16": single f16 products in the full-size convolutions, loss scaling)
    mp = os.environ.get("VILLAN_MIXED_PRECISION", "no")
    if mp not in ("no", "fp16"):
        raise ValueError(f"VILLAN_MIXED_PRECISION={mp!r}: 'no' or 'fp16'")
    cfg.mixed_precision = mp if cfg.sde_type in ("SDE-VP", "SDE-LDM") else "no"
    cfg.device_ids = [int(i) for i in range(len(cfg.gpu.split(",")))]  # :245 (indices into the visible set main() exported from --gpu)
    if isinstance(cfg.sample_ep, int) and cfg.sample_ep < 0:           # :248-251
        cfg.sample_ep = None
    small = cfg.dataset in ("CIFAR10", "MNIST", "SYNTHETIC-CIFAR10", "CELEBA-HQ-LATENT_PR05", "CELEBA-HQ-LATENT")
    big = cfg.dataset in ("CELEBA", "CELEBA-HQ", "LSUN-CHURCH", "LSUN-BEDROOM", "SYNTHETIC-CELEBA-HQ")
    if not (small or big):
        raise NotImplementedError()
    bs = cfg.batch_32 if small else cfg.batch_256                      # :266-287 (every mode, like the reference)
    if cfg.learning_rate is None:                                      # the from-scratch rate applies only when --ckpt is omitted
        scratch = cfg.ckpt is None
        cfg.learning_rate = (2e-4 if small else 6e-5) if not scratch else (cfg.learning_rate_32_scratch if small else cfg.learning_rate_256_scratch)
    if bs % cfg.batch != 0:
        raise ValueError(f"batch size {cfg.batch} should be divisible to {bs} for dataset {cfg.dataset}")
    if bs < cfg.batch:
        raise ValueError(f"batch size {cfg.batch} should be smaller or equal to {bs} for dataset {cfg.dataset}")
    cfg.gradient_accumulation_steps = int(bs // cfg.batch)
    # One process per GPU (torchrun): every rank computes the same config, but the existing-directory check, the directory
    # creation and the JSON side files belong to rank 0 alone -- main() puts a barrier behind it.  (The reference is a single
    # process driving nn.DataParallel, :440, so it has no such distinction.)
    rank0 = int(os.environ.get("RANK", "0")) == 0
    if mode in (MODE_TRAIN, MODE_TRAIN_MEASURE):
        cfg.output_dir = os.path.join(cfg.result, naming_fn(cfg))
        if rank0 and preflight:
            from dataset import DatasetLoader
            DatasetLoader.check_available(cfg.dataset, cfg.dataset_path)
        if rank0:
            if os.path.isdir(cfg.output_dir) and not cfg.overwrite:
                raise ValueError(f"Output directory: {cfg.output_dir} has already been created, please set overwrite flag --overwrite or -o")
            os.makedirs(cfg.output_dir, exist_ok=True)
            with open(os.path.join(cfg.output_dir, "args.json"), "w") as f:
                json.dump({k: v for k, v in vars(args).items()}, f, indent=2)
            with open(os.path.join(cfg.output_dir, "config.json"), "w") as f:
                json.dump({k: v for k, v in vars(cfg).items() if k != "extra"}, f, indent=2)
    elif mode in (MODE_SAMPLING, MODE_MEASURE) and rank0:              # :303-306 sampling.json / measure.json = the effective config
        with open(os.path.join(cfg.output_dir, f"{mode}.json"), "w") as f:
            json.dump({k: v for k, v in vars(cfg).items() if k != "extra"}, f, indent=2)
    cfg.ckpt_path = os.path.join(cfg.output_dir, cfg.ckpt_dir)
    cfg.data_ckpt_path = os.path.join(cfg.output_dir, cfg.data_ckpt_dir)
    if rank0:
        os.makedirs(cfg.ckpt_path, exist_ok=True)                      # :312-315
    return cfg


# ----------------------------------------------------------------------------------------------------------------- run
def _dist():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get("VILLAN_DIST_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm; gloo only for launcher tests on a box without GPUs
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def get_data_loader(cfg: TrainingConfig):
    from dataset import DatasetLoader
    vmin, vmax = (0.0, 1.0) if cfg.sde_type == "SDE-VE" else (-1.0, 1.0)          # reference :398-405
    dsl = DatasetLoader(root=cfg.dataset_path, name=cfg.dataset, batch_size=cfg.batch, vmin=vmin, vmax=vmax)
    dsl.set_poison(trigger_type=cfg.trigger, target_type=cfg.target, clean_rate=cfg.clean_rate, poison_rate=cfg.poison_rate,
                   ext_poison_rate=cfg.ext_poison_rate)
    return dsl.prepare_dataset(mode=cfg.dataset_load_mode, R_trigger_only=cfg.R_trigger_only)


def make_grid(images, path):
    """auto_grid of the reference (:586-613): uint8 = round(img*255), near-square PNG grid."""
    import numpy as np
    from PIL import Image
    arr = (np.asarray(images) * 255).round().astype("uint8")
    n = len(arr)
    cols = int(np.ceil(np.sqrt(n)))
    rows = int(np.ceil(n / cols))
    h, w, c = arr.shape[1:]
    grid = np.zeros((rows * h, cols * w, c), dtype="uint8")
    for i, im in enumerate(arr):
        grid[(i // cols) * h:(i // cols + 1) * h, (i % cols) * w:(i % cols + 1) * w] = im
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(grid.squeeze()).save(path)


def sampling(cfg: TrainingConfig, file_name, pipeline, dsl):
    """Clean + backdoor sample grids from the same seeded CPU noise (reference :570-676, TASK_GENERATE)."""
    import torch
    g = torch.Generator().manual_seed(cfg.seed)
    n = cfg.eval_sample_n
    noise = torch.randn((n, pipeline.unet.in_channels, pipeline.unet.sample_size, pipeline.unet.sample_size), generator=g)
    tag = f"{file_name:04d}" if isinstance(file_name, int) else str(file_name)
    kw = {} if cfg.ddim_eta is None else {"eta": cfg.ddim_eta}
    if cfg.task != TASK_GENERATE:
        return _special_sampling(cfg, tag, pipeline, dsl, noise, kw)
    for name, init in (("samples", noise), ("backdoor_samples", noise + pipeline.encode(dsl.trigger.unsqueeze(0)).to(noise.device))):
        res = pipeline(batch_size=n, generator=torch.Generator().manual_seed(cfg.seed), init=init, output_type=None,
                       num_inference_steps=cfg.infer_steps, start_from=int(cfg.infer_start), save_every_step=True, **kw)
        make_grid(res.images, os.path.join(cfg.output_dir, name, f"{tag}.png"))
        make_grid(res.movie[0], os.path.join(cfg.output_dir, name, f"{tag}_sample_t0.png"))


def _special_sampling(cfg, tag, pipeline, dsl, noise, kw):
    """Denoise / inpaint tasks (reference :637-678): start from the last N dataset images (+0.3*noise or masked) at
    step --infer_start."""
    import torch
    n = cfg.eval_sample_n
    ids = torch.tensor([(len(dsl) - i) % len(dsl) for i in range(n)])
    imgs = dsl.make_batch(ids, flip_bits=torch.zeros(n, dtype=torch.bool), full=False)["image"].cpu()
    poisoned = pipeline.encode(dsl.get_poisoned(imgs))
    ext = f"_{cfg.sched}_{cfg.infer_steps}_st{cfg.infer_start}_m{cfg.inpaint_mul}"
    noise_sp, mul = noise * 0.3, cfg.inpaint_mul
    table = {
        TASK_UNPOISONED_DENOISE: ("unpoisoned_noisy_samples", lambda: (imgs + noise_sp) * mul),
        TASK_POISONED_DENOISE: ("poisoned_noisy_samples", lambda: (poisoned + noise_sp) * mul),
        TASK_UNPOISONED_INPAINT_BOX: ("inpaint_box_unpoisoned_samples", lambda: dsl.get_inpainted_by_type(imgs, dsl.INPAINT_BOX) * mul),
        TASK_POISONED_INPAINT_BOX: ("inpaint_box_poisoned_samples", lambda: dsl.get_inpainted_by_type(poisoned, dsl.INPAINT_BOX) * mul),
        TASK_UNPOISONED_INPAINT_LINE: ("inpaint_line_unpoisoned_samples", lambda: dsl.get_inpainted_by_type(imgs, dsl.INPAINT_LINE) * mul),
        TASK_POISONED_INPAINT_LINE: ("inpaint_line_poisoned_samples", lambda: dsl.get_inpainted_by_type(poisoned, dsl.INPAINT_LINE) * mul),
    }
    if cfg.task not in table:
        raise NotImplementedError(f"Sampling task: {cfg.task} isn't implemented")
    folder, make_init = table[cfg.task]
    res = pipeline(batch_size=n, generator=torch.Generator().manual_seed(cfg.seed), init=make_init(), output_type=None,
                   num_inference_steps=cfg.infer_steps, start_from=int(cfg.infer_start), save_every_step=True, **kw)
    make_grid(res.images, os.path.join(cfg.output_dir, folder + ext, f"{tag}.png"))
    make_grid(res.movie[0], os.path.join(cfg.output_dir, folder + ext, f"{tag}_sample_t0.png"))


def score_key(cfg, key: str) -> str:
    """reference :726-738."""
    res = f"{key}_ep{cfg.sample_ep}" if cfg.sample_ep is not None else key
    res += "_noclip" if not cfg.clip else ""
    if cfg.sched is not None:
        res += f"_{cfg.sched}-{cfg.infer_steps}"
    if cfg.sched == "DDIM-SCHED" and cfg.ddim_eta is not None:
        res += f"-eta{cfg.ddim_eta}"
    if cfg.task != TASK_GENERATE:
        return res + f"_{cfg.measure_inpaint_sample_n}_{cfg.task}"
    return res + f"_{cfg.measure_sample_n}"


def measure_inpaints(cfg, pipeline, dsl):
    """Denoise / inpaint measurement (reference :875-949 + measure_inpaint :874-892): recover the last N dataset images from
    their corrupted version (start at --infer_start, init * --inpaint_mul) and score MSE / SSIM against the task's target.
    Reproduced as written: the unpoisoned tasks compare the recovered [0,1] images with the dataset tensors in their
    NORMALISED range (reference passes `target_imgs=imgs`), the poisoned ones with the backdoor target mapped to [0,1].
    LPIPS runs on the HIP AlexNet taps when its two weight files are present locally, else None."""
    import numpy as np
    import torch
    from villandiffusion_amd.metrics import mse_batch, ssim_batch
    n = min(cfg.measure_inpaint_sample_n, len(dsl))
    noise = torch.randn((n, pipeline.unet.in_channels, pipeline.unet.sample_size, pipeline.unet.sample_size),
                        generator=torch.Generator().manual_seed(cfg.seed))
    noise_sp = noise * 0.3
    ids = torch.tensor([(len(dsl) - i) % len(dsl) for i in range(n)])
    imgs = torch.cat([dsl.make_batch(ids[s:s + 256], flip_bits=torch.zeros(len(ids[s:s + 256]), dtype=torch.bool), full=False)["image"].cpu()
                      for s in range(0, n, 256)])
    tgt01 = dsl.target.clamp(0, 1) if cfg.sde_type == "SDE-VE" else (dsl.target / 2 + 0.5).clamp(0, 1)
    bd_targets = tgt01[None].expand(n, -1, -1, -1)
    poisoned = dsl.get_poisoned(imgs)
    table = {
        TASK_UNPOISONED_DENOISE: lambda: (imgs, imgs + noise_sp),
        TASK_POISONED_DENOISE: lambda: (bd_targets, poisoned + noise_sp),
        TASK_UNPOISONED_INPAINT_LINE: lambda: (imgs, dsl.get_inpainted_by_type(imgs, dsl.INPAINT_LINE)),
        TASK_POISONED_INPAINT_LINE: lambda: (bd_targets, dsl.get_inpainted_by_type(poisoned, dsl.INPAINT_LINE)),
        TASK_UNPOISONED_INPAINT_BOX: lambda: (imgs, dsl.get_inpainted_by_type(imgs, dsl.INPAINT_BOX)),
        TASK_POISONED_INPAINT_BOX: lambda: (bd_targets, dsl.get_inpainted_by_type(poisoned, dsl.INPAINT_BOX)),
    }
    if cfg.task not in table:
        raise NotImplementedError(f"Measurement task: {cfg.task} isn't implemented")
    target_imgs, corrupt = table[cfg.task]()
    rec = []
    for s in range(0, n, cfg.eval_max_batch):
        init = corrupt[s:s + cfg.eval_max_batch] * cfg.inpaint_mul
        out = pipeline(batch_size=len(init), generator=torch.Generator().manual_seed(cfg.seed), init=init, output_type=None,
                       num_inference_steps=cfg.infer_steps, start_from=int(cfg.infer_start), save_every_step=False)
        rec.append(out.images)
    recover = torch.from_numpy(np.vstack(rec)).permute(0, 3, 1, 2).float()
    return {"LPIPS": measure_lpips(recover, target_imgs.float(), cfg.eval_max_batch), "MSE": mse_batch(recover, target_imgs.float()),
            "SSIM": ssim_batch(recover, target_imgs.float(), device=pipeline.device)}


def measure_lpips(recover, target, max_batch: int = 256):
    """reference :892  float(torch.mean(lpips.LPIPS(net='alex')(recover_imgs, target_imgs))) -- on the HIP AlexNet taps
    (villandiffusion_amd/lpips.py); None when the two weight files are not present locally (no network on the box)."""
    import torch
    import lpips
    try:
        model = lpips.LPIPS(net="alex")
    except FileNotFoundError as e:
        print(f"measure: LPIPS not computed -- {e}")
        return None
    vals = [model(recover[s:s + max_batch], target[s:s + max_batch]).flatten().cpu() for s in range(0, len(recover), max_batch)]
    return float(torch.cat(vals).mean())


def measure(cfg, pipeline, dsl, rank: int = 0, world: int = 1):
    """reference :1017-1096: N clean + N backdoor samples as PNGs (chunks of --eval_max_batch, split over ranks),
    then FID of the clean samples against the dataset (when the InceptionV3 weights file is present) and MSE / SSIM of the backdoor
    samples against the target -> score.json (same key naming)."""
    import numpy as np
    import torch
    from PIL import Image
    from villandiffusion_amd.metrics import mse_batch, ssim_batch
    from villandiffusion_amd.sampling_io import batch_sampling_save
    if cfg.task != TASK_GENERATE:
        if rank != 0:
            return None
        sc = measure_inpaints(cfg, pipeline, dsl)
        path = os.path.join(cfg.output_dir, "score.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        for k, v in sc.items():
            data[score_key(cfg, k)] = v
        with open(path, "w") as f:
            json.dump(data, f, indent=2, sort_keys=True)
        print(f"measure[{cfg.task}]: LPIPS {sc['LPIPS']} MSE {sc['MSE']:.5f} SSIM {sc['SSIM']:.5f}")
        return sc
    n = cfg.measure_sample_n
    step = f"{cfg.sample_ep}" if cfg.sample_ep is not None else ""
    sub = ("" if cfg.clip else "_noclip") + ("" if cfg.sched is None else f"_{cfg.sched}-{cfg.infer_steps}")
    clean_path = os.path.join(cfg.output_dir, f"clean{step}{sub}_{n}")
    bd_path = os.path.join(cfg.output_dir, f"backdoor{step}{sub}_{n}")
    g = torch.Generator().manual_seed(cfg.seed)
    noise = torch.randn((n, pipeline.unet.in_channels, pipeline.unet.sample_size, pipeline.unet.sample_size), generator=g)
    bd_noise = noise + pipeline.encode(dsl.trigger.unsqueeze(0)).to(noise.device)
    # Throughput mode of the 2 x N-image sampling job: the per-step noise of the stochastic samplers comes from the in-kernel Philox stream seeded with
    # cfg.seed (not from a CPU generator whose draws would cross PCIe every step), which also lets the chunks of --eval_max_batch run concurrently
    # (sampling_io.batch_sampling_save -> pipelines.sample_concurrent).  VILLAN_HOST_RNG=1: the reference's CPU-generator noise, one chunk at a time.
    sch = pipeline.scheduler
    if os.environ.get("VILLAN_HOST_RNG", "0") != "1" and hasattr(sch, "device_rng_seed") and pipeline.device.type == "cuda":
        sch.device_rng_seed = int(cfg.seed)
    for path, init in ((clean_path, noise), (bd_path, bd_noise)):
        batch_sampling_save(n, pipeline, path, init=init, max_batch_n=cfg.eval_max_batch, rng=torch.Generator().manual_seed(cfg.seed),
                            num_inference_steps=cfg.infer_steps, eta=cfg.ddim_eta, rank=rank, world=world)
    if world > 1:
        torch.distributed.barrier()
    if rank != 0:
        return None
    imgs = np.stack([np.asarray(Image.open(os.path.join(bd_path, f"{i}.png")).convert("RGB")) for i in range(n)])
    gen = torch.from_numpy(imgs).permute(0, 3, 1, 2).float() / 255.0
    # reference :1081-1085: SDE-VE data lives in [0, 1] (vmin, vmax = 0, 1), the other SDE types in [-1, 1]
    tgt01 = dsl.target.clamp(0, 1) if cfg.sde_type == "SDE-VE" else (dsl.target / 2 + 0.5).clamp(0, 1)
    tgt = tgt01[None].expand(n, -1, -1, -1)
    fid_sc = measure_fid(cfg, dsl, clean_path, n)
    sc = {"FID": fid_sc, "MSE": mse_batch(gen, tgt), "SSIM": ssim_batch(gen, tgt, device=pipeline.device)}
    path = os.path.join(cfg.output_dir, "score.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    for k, v in sc.items():
        data[score_key(cfg, k)] = v
    with open(path, "w") as f:
        json.dump(data, f, indent=2, sort_keys=True)
    print(f"measure: FID {sc['FID']} MSE {sc['MSE']:.5f} SSIM {sc['SSIM']:.5f}")
    return sc


def measure_fid(cfg, dsl, clean_path: str, n: int, folder_name: str = "measure"):
    """reference :1032,1072: FID between `<folder_name>/<dataset>` (the first n dataset images as PNGs -- written here when the
    directory is missing, which is what the reference's commented-out block :1043-1049 did) and the clean samples.  The InceptionV3
    weights are a local file (villandiffusion_amd.inception.load_fid_weights: no network on the box); without it FID stays None."""
    import numpy as np
    from PIL import Image
    from fid_score import fid
    from villandiffusion_amd.inception import load_fid_weights
    try:
        load_fid_weights()
    except FileNotFoundError as e:
        print(f"measure: FID not computed -- {e}")
        return None
    dataset_img_dir = os.path.join(folder_name, cfg.dataset)
    imgs = getattr(dsl, "_images", None)
    if (not os.path.isdir(dataset_img_dir) or len(os.listdir(dataset_img_dir)) < n) and imgs is not None:
        os.makedirs(dataset_img_dir, exist_ok=True)
        order = np.random.default_rng(cfg.seed).permutation(len(imgs))[:n]       # reference: ds.shuffle(seed=config.seed)[:n]
        for i, j in enumerate(order):
            Image.fromarray(np.asarray(imgs[j]).squeeze()).save(os.path.join(dataset_img_dir, f"{i}.png"))
    if not os.path.isdir(dataset_img_dir):
        print(f"measure: FID not computed -- {dataset_img_dir} does not exist")
        return None
    return float(fid(path=[dataset_img_dir, clean_path], num_workers=4, batch_size=cfg.eval_max_batch))


def checkpoint(cfg, trainer, pipeline, epoch, step, dsl=None):
    import torch
    os.makedirs(cfg.ckpt_path, exist_ok=True)
    torch.save(trainer.state_dict(), os.path.join(cfg.ckpt_path, "trainer.pt"))
    data = {"epoch": epoch, "step": step}
    if dsl is not None and hasattr(dsl, "loader_state"):
        data["loader"] = dsl.loader_state()                           # device flip generator: a resumed run draws the same flips
    torch.save(data, cfg.data_ckpt_path)
    pipeline.save_pretrained(cfg.output_dir)
    if cfg.is_save_all_model_epochs:                                   # reference :1110-1114: a copy per checkpointed epoch
        pipeline.save_pretrained(get_ep_model_path(cfg, cfg.output_dir, epoch))


def get_ep_model_path(cfg, dir: str, epoch: int) -> str:
    """reference :1099-1100."""
    return os.path.join(dir, cfg.ep_model_dir, f"ep{epoch}")


# batch key regressed against (reference VillanDiffusion.py:1159 'target'); rm_backdoor_VillanDiffusion.py sets 'image'
TARGET_LATENT_KEY = "target"


def train_loop(cfg: TrainingConfig, dsl, rank: int, world: int):
    """reference :1117-1196.  Kept as written there: grids and checkpoints carry the 0-based epoch index (`sampling(config, epoch,
    ...)`, `checkpoint(cur_epoch=epoch)`), a grid `0000.png` is sampled before the first step (on resume too), training ends with a
    checkpoint and the `final` grid, and --mode resume restarts AT the recorded epoch (`range(start_epoch, epoch)` with the epoch
    index the last checkpoint stored, :457-461) with the model taken from the run directory.  Not kept: the reference swallows
    training exceptions (:1189-1191); here they propagate."""
    import torch
    from loss import LossFn
    from model import DiffuserModelSched
    from villandiffusion_amd.trainer import Trainer
    if cfg.mode == MODE_RESUME:                                        # the run directory setup() resolved (reference: config.ckpt IS that path)
        src = cfg.output_dir
    else:
        src = cfg.ckpt if cfg.ckpt is not None else DiffuserModelSched.MODEL_DEFAULT
    model, vae, noise_sched, get_pipeline = DiffuserModelSched.get_model_sched(
        image_size=dsl.image_size, channels=dsl.channel, ckpt=src, sde_type=cfg.sde_type, clip_sample=cfg.clip, noise_sched_type=cfg.sched)
    if cfg.mixed_precision == "fp16" and hasattr(model, "conv_math"):
        model.conv_math = "f16"
    if world > 1:                                                      # identical replicas
        torch.distributed.broadcast(model.flat_param, src=0)
    loss_fn = LossFn(noise_sched=noise_sched, sde_type=cfg.sde_type, loss_type="l2", psi=cfg.psi, solver_type=cfg.solver_type,
                     vp_scale=cfg.vp_scale, ve_scale=cfg.ve_scale)
    n_batch = (len(dsl) + cfg.batch * world - 1) // (cfg.batch * world)
    trainer = Trainer(model, loss_fn, lr=cfg.learning_rate, total_steps=n_batch * cfg.epoch, warmup_steps=cfg.lr_warmup_steps,
                      grad_accum=cfg.gradient_accumulation_steps)
    start_epoch, step = 0, 0
    if cfg.mode == MODE_RESUME:
        trainer.load_state_dict(torch.load(os.path.join(cfg.ckpt_path, "trainer.pt"), map_location=model.device))
        d = torch.load(cfg.data_ckpt_path)
        start_epoch, step = int(d["epoch"]), int(d["step"])
        if hasattr(dsl, "load_loader_state"):
            dsl.load_loader_state(d.get("loader"))
    pipeline = get_pipeline(None, model, vae, noise_sched)
    if rank == 0:
        sampling(cfg, 0, pipeline, dsl)
    T = noise_sched.config.num_train_timesteps
    epoch = start_epoch
    for epoch in range(start_epoch, cfg.epoch):
        loader = dsl.get_dataloader(rank=rank, world=world, epoch=epoch, full=False)
        nb = len(loader)
        for i, batch in enumerate(loader):
            bs = batch["pixel_values"].shape[0]
            t = torch.randint(0, T, (bs,), device=model.device).long()            # reference :1151
            loss = trainer.train_step(batch, t, last_batch=(i == nb - 1), target_key=TARGET_LATENT_KEY)
            step += 1
            if rank == 0 and (step % 50 == 0 or i == nb - 1):
                print(f"epoch {epoch} step {step} loss {float(loss):.5f} lr {trainer.lr:.3e}", flush=True)
        if rank == 0:
            if (epoch + 1) % cfg.save_image_epochs == 0 or epoch == cfg.epoch - 1:
                sampling(cfg, epoch, pipeline, dsl)
                if dsl.image_size >= 128:                              # the captured forwards pin their peak memory: release it before training resumes
                    from villandiffusion_amd.pipelines import drop_sampler_graphs
                    drop_sampler_graphs(model)
            if (epoch + 1) % cfg.save_model_epochs == 0 or epoch == cfg.epoch - 1:
                checkpoint(cfg, trainer, pipeline, epoch, step, dsl)
    if rank == 0:                                                      # reference :1192-1195
        checkpoint(cfg, trainer, pipeline, epoch, step, dsl)
        sampling(cfg, "final", pipeline, dsl)
    return pipeline


def effective_gpu(args: argparse.Namespace) -> str:
    """The --gpu string that applies to this run BEFORE the config overlay runs (it must be known before anything touches the GPU):
    the command line, else (resume / sampling / measure) the value the run directory's args.json recorded, else the default "0"."""
    if getattr(args, "gpu", None):
        return str(args.gpu)
    if args.mode in (MODE_RESUME, MODE_SAMPLING, MODE_MEASURE) and getattr(args, "ckpt", None):
        base = args.ckpt if os.path.isdir(args.ckpt) else os.path.join(args.result or DEFAULT["result"], args.ckpt)
        try:
            with open(os.path.join(base, "args.json")) as f:
                g = json.load(f).get("gpu")
            if g:
                return str(g)
        except OSError:
            pass
    return DEFAULT["gpu"]


def gpu_plan(gpu: str, env: dict) -> dict:
    """What `--gpu` means here (reference VillanDiffusion.py:240-245, 440: CUDA_VISIBLE_DEVICES = config.gpu, then nn.DataParallel over
    device_ids 0..n-1 of the visible set).  Pure function of the flag and the environment, so that it can be tested without a GPU.

    * the listed indices select from the devices visible NOW (an already-set HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES is composed with,
      not overridden: `--gpu 1` inside HIP_VISIBLE_DEVICES=4,5 is physical device 5);
    * one index  -> this process runs on that device (`visible` is exported before the HIP runtime starts);
    * n indices, no WORLD_SIZE -> the parent starts n rank processes, one per listed GPU, through torch.distributed.run BEFORE making any GPU
      call (it never makes one) and exits with the launcher's code -- the one-process-per-GPU form of the reference's DataParallel;
    * under a launcher (WORLD_SIZE set) the launcher's device assignment stands: LOCAL_RANK picks the device, nothing is changed."""
    ids = [i.strip() for i in str(gpu).split(",") if i.strip() != ""]
    if not ids or not all(i.isdigit() for i in ids):
        raise ValueError(f"--gpu {gpu!r}: expected a device index or a comma-separated list of indices")
    if len(set(ids)) != len(ids):
        raise ValueError(f"--gpu {gpu!r}: a device is listed twice")
    if "WORLD_SIZE" in env:
        return {"action": "rank", "visible": None, "n": int(env["WORLD_SIZE"])}
    cur = env.get("HIP_VISIBLE_DEVICES") or env.get("CUDA_VISIBLE_DEVICES")
    if cur:
        pool = [c.strip() for c in cur.split(",") if c.strip() != ""]
        for i in ids:
            if int(i) >= len(pool):
                raise ValueError(f"--gpu {gpu!r}: index {i} is outside the {len(pool)} visible device(s) ({cur})")
        vis = [pool[int(i)] for i in ids]
    else:
        vis = ids
    return {"action": "single" if len(ids) == 1 else "spawn", "visible": ",".join(vis), "n": len(ids)}


_VISIBLE0: Optional[str] = None


def apply_gpu_flag(args: argparse.Namespace, argv: Optional[List[str]]):
    """Honour --gpu (see gpu_plan).  Called first thing in main(): no torch import, no HIP call has happened yet."""
    global _VISIBLE0
    if _VISIBLE0 is None:                                              # main() called again in this process: --gpu indexes the ORIGINAL set
        _VISIBLE0 = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES") or ""
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    if _VISIBLE0:
        env["HIP_VISIBLE_DEVICES"] = _VISIBLE0
    plan = gpu_plan(effective_gpu(args), env)
    if plan["action"] == "rank":
        return plan
    os.environ["HIP_VISIBLE_DEVICES"] = plan["visible"]
    os.environ["CUDA_VISIBLE_DEVICES"] = plan["visible"]              # the name the reference sets (:240); HIP honours both
    if plan["action"] == "single":
        return plan
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={plan['n']}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    print(f"[VillanDiffusion] --gpu {effective_gpu(args)}: one rank per GPU ({plan['visible']}): {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main(argv: Optional[List[str]] = None):
    args = parse_args(argv)
    apply_gpu_flag(args, argv)
    rank, world, _ = _dist()
    if os.environ.get("VILLAN_RENDEZVOUS_ONLY") == "1":               # launcher test on a box without GPUs: process-group proof, no compute
        import torch
        import torch.distributed as dist
        n = torch.ones(1)
        if world > 1:
            dist.all_reduce(n)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "world_size": world, "ranks_counted": int(n),
                              "visible": os.environ.get("HIP_VISIBLE_DEVICES")}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    err = None
    try:
        cfg = setup(args, preflight=True)
    except Exception as e:                                             # noqa: BLE001  (re-raised below, on every rank)
        err = e
    if world > 1:
        # rank 0 alone checks / creates the run directory and writes its side files; if it fails there (directory exists without
        # --overwrite, dataset not found) every rank must exit, not sit in a collective until the RCCL timeout: agree on a flag first
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([0 if err is None else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag) and err is None:
            raise SystemExit(f"rank {rank}: another rank failed in setup(); exiting")
    if err is not None:
        raise err
    dsl = get_data_loader(cfg)
    if cfg.mode in (MODE_TRAIN, MODE_RESUME, MODE_TRAIN_MEASURE):
        pipeline = train_loop(cfg, dsl, rank, world)
    else:
        from model import DiffuserModelSched
        src = cfg.output_dir                                           # reference :424-431: --sample_ep N loads epochs/epN
        if cfg.sample_ep is not None:
            src = get_ep_model_path(cfg, cfg.output_dir, cfg.sample_ep)
        model, vae, noise_sched, get_pipeline = DiffuserModelSched.get_pretrained(ckpt=src, clip_sample=cfg.clip,
                                                                                  noise_sched_type=cfg.sched, sde_type=cfg.sde_type)
        pipeline = get_pipeline(None, model, vae, noise_sched)
    if cfg.mode == MODE_SAMPLING and rank == 0:
        sampling(cfg, cfg.sample_ep if cfg.sample_ep is not None else "final", pipeline, dsl)
    if cfg.mode in (MODE_MEASURE, MODE_TRAIN_MEASURE):
        measure(cfg, pipeline, dsl, rank, world)


if __name__ == "__main__":
    main()
