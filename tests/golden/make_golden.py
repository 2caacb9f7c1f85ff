"""Generate golden fixtures by importing the REFERENCE's own loss.py / dataset.py.

Run in the build container only (needs /root/reference; it never travels):

    python tests/golden/make_golden.py

Recipe: the reference modules import torchvision /
diffusers / wandb / comet_ml at module level, none of which is installed; they
are satisfied with MagicMock stubs that are never *called* by the functions
used here (R-coefficient tables, LossFn with a fake scheduler object and an
analytic stand-in model, box triggers / box targets / masks).  Only numbers
produced by the reference's own arithmetic are stored; no reference source.

Outputs (committed): loss_tables.npz, loss_batch.npz, noise_scheduler.npz, backdoor_boxes.npz, ... (see main()).
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    sys.path.insert(0, REF)
    import datasets  # noqa: F401  (must precede the torchvision stub: it probes torchvision.__spec__)
    for m in ["torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets",
              "diffusers", "comet_ml", "wandb"]:
        sys.modules[m] = MagicMock()
    import torch  # noqa: F401
    import loss as ref_loss
    import dataset as ref_dataset
    return ref_loss, ref_dataset


class FakeVPSched:
    """Scheduler-like object exposing only what loss.py:830-834,924 touches."""

    def __init__(self, betas):
        import torch
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, 0)

    def add_noise(self, x0, eps, t):
        ac = self.alphas_cumprod.to(x0.device)
        sa = (ac[t] ** 0.5).flatten()
        sb = ((1 - ac[t]) ** 0.5).flatten()
        while sa.dim() < x0.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * x0 + sb * eps


class FakeVESched:
    def __init__(self, sigmas_desc):
        self.sigmas = sigmas_desc


def stand_in_model(x, t, return_dict=False):
    """Analytic stand-in for the UNet (fixed, documented): 0.5*x - 0.25*roll(x) + 0.1."""
    import torch
    return (0.5 * x - 0.25 * torch.roll(x, 1, -1) + 0.1,)


def main():
    import torch
    ref_loss, ref_dataset = import_reference()
    LossFn = ref_loss.LossFn

    lin = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float32)
    sl = torch.linspace(0.0015 ** 0.5, 0.0195 ** 0.5, 1000, dtype=torch.float32) ** 2
    ts = torch.linspace(1, 1e-5, 2000)
    sig_desc = torch.tensor([0.01 * (380.0 / 0.01) ** t for t in ts])

    tables = {}
    for name, betas in (("vp_linear", lin), ("ldm_scaled_linear", sl)):
        sched = FakeVPSched(betas)
        tables[f"{name}/hs"] = ref_loss.get_hs_vp(sched.alphas, sched.alphas_cumprod).numpy()
        for psi in (0.0, 0.5, 1.0):
            for solver in ("sde", "ode"):
                lf = LossFn(sched, "SDE-VP", psi=psi, solver_type=solver)
                step, coef = lf._LossFn__get_R_step_coef()
                tables[f"{name}/psi{psi}/{solver}/step"] = step.numpy()
                tables[f"{name}/psi{psi}/{solver}/coef"] = coef.numpy()
    sched_ve = FakeVESched(sig_desc)
    tables["ve/sigmas_desc"] = sig_desc.numpy()
    for solver in ("sde", "ode"):
        lf = LossFn(sched_ve, "SDE-VE", psi=0, solver_type=solver)
        step, coef = lf._LossFn__get_R_step_coef()
        tables[f"ve/psi0/{solver}/step"] = step.numpy()
        tables[f"ve/psi0/{solver}/coef"] = coef.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_tables.npz"), **tables)

    # seeded [4,3,32,32] batch through __get_inputs_targets and p_loss
    g = torch.Generator().manual_seed(1234)
    x0 = torch.rand((4, 3, 32, 32), generator=g) * 2 - 1
    R = torch.rand((4, 3, 32, 32), generator=g) * 2 - 1
    R[0] = 0                                   # a clean sample has R = 0
    eps = torch.randn((4, 3, 32, 32), generator=g)
    t_vp = torch.tensor([0, 10, 500, 999])
    t_ve = torch.tensor([0, 10, 1000, 1999])
    batch = {"x0": x0.numpy(), "R": R.numpy(), "eps": eps.numpy(), "t_vp": t_vp.numpy(), "t_ve": t_ve.numpy()}
    for name, sched, sde, t, psis in (("vp", FakeVPSched(lin), "SDE-VP", t_vp, (0.0, 0.5, 1.0)),
                                      ("ldm", FakeVPSched(sl), "SDE-LDM", t_vp, (1.0,)),
                                      ("ve", sched_ve, "SDE-VE", t_ve, (0.0,))):
        for psi in psis:
            for solver in ("sde", "ode"):
                lf = LossFn(sched, sde, psi=psi, solver_type=solver)
                xt, y = lf._LossFn__get_inputs_targets(x_start=x0, R=R, timesteps=t, noise=eps)
                loss = lf.p_loss(stand_in_model, x0, R, t, noise=eps)
                key = f"{name}/psi{psi}/{solver}"
                batch[key + "/x_t"], batch[key + "/y"] = xt.numpy(), y.numpy()
                batch[key + "/loss"] = np.float32(loss.item())
    np.savez_compressed(os.path.join(OUT, "loss_batch.npz"), **batch)

    # the reference's own NoiseScheduler (loss.py:62-160) and forward-diffusion samplers q_sample_clean / q_sample_backdoor
    # (loss.py:175-196): beta / alpha-bar / posterior-variance / R_coef tables for its four beta schedules, and (x_t, target) on the
    # seeded batch above.  Pins the oracle's DDPM variance table, `add_noise` and the psi = 1 (BadDiffusion) correction.
    ns = {}
    for tag, kind in (("linear", "SC_LIN"), ("quadratic", "SC_QUAD"), ("cosine", "SC_COS"), ("sigmoid", "SC_SIGM")):
        sch = ref_loss.NoiseScheduler(timesteps=1000, scheduler=kind)
        for attr in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas", "sqrt_alphas_cumprod",
                     "sqrt_one_minus_alphas_cumprod", "R_coef", "posterior_variance"):
            ns[f"{tag}/{attr}"] = getattr(sch, attr).numpy()
        if tag in ("linear", "quadratic"):
            xc, _ = ref_loss.q_sample_clean(sch, x0, t_vp, noise=eps)
            xb, yb = ref_loss.q_sample_backdoor(sch, x0, R, t_vp, noise=eps)
            ns[f"{tag}/q_sample_clean"], ns[f"{tag}/q_sample_backdoor_x"], ns[f"{tag}/q_sample_backdoor_y"] = xc.numpy(), xb.numpy(), yb.numpy()
    np.savez_compressed(os.path.join(OUT, "noise_scheduler.npz"), **ns)

    # box triggers / targets / masks from dataset.Backdoor
    Backdoor = ref_dataset.Backdoor
    bd = Backdoor(root=REF)
    boxes = {}
    trig_types = ["SM_BOX", "XSM_BOX", "XXSM_BOX", "XXXSM_BOX", "BIG_BOX",
                  "BOX_18", "BOX_14", "BOX_11", "BOX_8", "BOX_4", "NONE"]
    for S in (32, 256):
        for (vmin, vmax) in ((-1.0, 1.0), (0.0, 1.0)):
            for tt in trig_types:
                trig = bd.get_trigger(type=tt, channel=3, image_size=S, vmin=vmin, vmax=vmax)
                key = f"S{S}/v{vmin}_{vmax}/{tt}"
                boxes[key + "/trigger"] = trig.numpy().astype(np.float32)
                boxes[key + "/mask"] = torch.where(trig > vmin, 0, 1).numpy().astype(np.int64)   # dataset.py:472-473
                if tt in ("BOX_14", "SM_BOX", "NONE"):
                    for tg in ("CORNER", "NOSHIFT", "SHIFT"):
                        tgt = bd.get_target(type=tg, trigger=trig, vmin=vmin, vmax=vmax)
                        boxes[key + f"/target_{tg}"] = tgt.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "backdoor_boxes.npz"), **boxes)

    # inpainting corruptions (dataset.py:547-579), the poisoning blend (dataset.py:472-473, 540-545) and util.normalize
    # (util.py:119-147).  The methods touch no dataset state: call them on an uninitialised DatasetLoader.
    import util as ref_util
    DL = ref_dataset.DatasetLoader
    dl = DL.__new__(DL)
    misc = {}
    g = torch.Generator().manual_seed(7)
    for S in (32, 50):
        imgs = torch.rand(2, 3, S, S, generator=g) * 2 - 1
        misc[f"inpaint/S{S}/imgs"] = imgs.numpy()
        for it in ("INPAINT_BOX", "INPAINT_LINE"):
            misc[f"inpaint/S{S}/{it}"] = dl.get_inpainted_by_type(imgs=imgs, inpaint_type=it).numpy()
    trig = bd.get_trigger(type="BOX_14", channel=3, image_size=32, vmin=-1.0, vmax=1.0)
    setattr(dl, "_DatasetLoader__trigger", trig)
    setattr(dl, "_DatasetLoader__vmin", -1.0)
    imgs = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    misc["poisoned/imgs"] = imgs.numpy()
    misc["poisoned/out"] = dl.get_poisoned(imgs).numpy()
    x = torch.rand(2, 3, 8, 8, generator=g) * 255
    misc["normalize/x"] = x.numpy()
    misc["normalize/t_0_255_to_m1_1"] = ref_util.normalize(x, 0, 255, -1, 1).numpy()
    misc["normalize/t_auto_to_0_1"] = ref_util.normalize(x).numpy()
    misc["normalize/t_0_1_to_m1_1"] = ref_util.normalize(x / 255, 0, 1, -1, 1).numpy()
    misc["normalize/np_auto_to_m1_1"] = ref_util.normalize(x.numpy(), None, None, -1, 1)
    misc["normalize/np_0_255_keepmax"] = ref_util.normalize(x.numpy(), 0, 255, 0, None)
    np.savez_compressed(os.path.join(OUT, "inpaint_normalize.npz"), **misc)

    # API-surface constants: every upper-case str / number class attribute of the three drop-in classes
    import json
    import model as ref_model

    def consts(cls):
        return {k: v for k, v in vars(cls).items() if k.isupper() and isinstance(v, (str, int, float, bool))}

    api = {"DiffuserModelSched": consts(ref_model.DiffuserModelSched), "Backdoor": consts(Backdoor), "DatasetLoader": consts(DL)}
    with open(os.path.join(OUT, "api_constants.json"), "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)

    # CLI surface of the reference driver (VillanDiffusion.py:77-111): its module is imported with every missing third-party
    # module stubbed, parse_args() is intercepted at ArgumentParser.parse_args and the declared options are dumped
    import argparse
    import builtins

    real_import = builtins.__import__

    def lenient_import(name, globals=None, locals=None, fromlist=(), level=0):
        try:
            return real_import(name, globals, locals, fromlist, level)
        except ModuleNotFoundError:
            parts = name.split(".")
            for i in range(1, len(parts) + 1):
                sys.modules.setdefault(".".join(parts[:i]), MagicMock())
            return sys.modules[name] if fromlist else sys.modules[parts[0]]

    class _Stop(Exception):
        pass

    captured = {}

    def capture(self, *a, **k):
        captured["parser"] = self
        mod = sys.modules.get("VillanDiffusion")          # half-imported reference driver: constants and TrainingConfig exist already
        import dataclasses
        captured["module_consts"] = {k: v for k, v in vars(mod).items()
                                     if k.isupper() and isinstance(v, (str, int, float, bool, list, type(None)))}
        tc = getattr(mod, "TrainingConfig", None)
        if tc is not None and dataclasses.is_dataclass(tc):
            captured["training_config"] = {f.name: f.default for f in dataclasses.fields(tc)
                                           if isinstance(f.default, (str, int, float, bool, type(None)))}
            names = []                                 # result-directory names (naming_fn, :186-190) for a few configurations
            for kw in ({"dataset": "CIFAR10"}, {"ckpt": "DDPM-CIFAR10-32", "dataset": "CIFAR10", "epoch": 50, "poison_rate": 0.1, "trigger": "BOX_14",
                            "target": "HAT", "learning_rate": 0.0002, "sched": "DDIM-SCHED", "postfix": "new"},
                       {"ckpt": "NCSNPP-CIFAR10-32", "dataset": "CELEBA-HQ", "sde_type": "SDE-VE", "psi": 0.0, "solver_type": "ode", "ve_scale": 2.0,
                        "ext_poison_rate": 0.5, "learning_rate": 2e-05}):
                c = tc()
                for k, v in kw.items():
                    setattr(c, k, v)
                names.append({"overrides": kw, "name": mod.naming_fn(config=c)})
            captured["naming"] = names
        raise _Stop()

    builtins.__import__ = lenient_import
    orig_parse = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = capture
    try:
        try:
            import VillanDiffusion  # noqa: F401   (the reference runs setup() -> parse_args() at import time, :323)
        except _Stop:
            pass
    finally:
        builtins.__import__ = real_import
        argparse.ArgumentParser.parse_args = orig_parse
    # --- the reference's setup() (config overlay, :200-321) and score-file keys (:724-778) on a handful of command lines.
    # Second import of the driver: parse_args() now returns real namespaces, setup() runs for real in a temp result dir, and the
    # script body is stopped at its first DatasetLoader(...) (by then every function of the module is defined).
    import shutil
    import tempfile
    import copy
    tmp = tempfile.mkdtemp(prefix="villan_golden_")
    state = {"argv": None, "mod": None}

    def parse_real(self, *a, **k):
        return orig_parse(self, state["argv"])

    def stop_dsl(self, *a, **k):
        state["mod"] = sys.modules.get("VillanDiffusion")
        raise _Stop()

    train_argv = ["--project", "default", "--mode", "train", "--dataset", "CIFAR10", "--batch", "4", "--epoch", "1", "--poison_rate", "0.1",
                  "--trigger", "BOX_14", "--target", "HAT", "--ckpt", "DDPM-CIFAR10-32", "--fclip", "o", "-o", "--gpu", "0",
                  "--result", tmp, "--sched", "DDIM-SCHED"]
    cases = [("train_cfg1", train_argv),
             ("train_ve", ["--mode", "train", "--dataset", "CIFAR10", "--batch", "128", "--sde_type", "SDE-VE", "--psi", "0", "--solver_type", "ode",
                           "--ve_scale", "2.0", "--result", tmp, "-o", "--postfix", "x"]),
             ("train_256", ["--mode", "train+measure", "--dataset", "CELEBA-HQ", "--batch", "16", "--ckpt", "DDPM-CELEBA-HQ-256", "--result", tmp,
                            "-o", "--learning_rate", "1e-5", "--R_trigger_only", "--dataset_load_mode", "EXTEND", "--ext_poison_rate", "0.3"])]
    orig_init = ref_dataset.DatasetLoader.__init__
    builtins.__import__ = lenient_import
    argparse.ArgumentParser.parse_args = parse_real
    ref_dataset.DatasetLoader.__init__ = stop_dsl
    sys.modules.pop("VillanDiffusion", None)
    for m in ("accelerate", "torchmetrics", "lpips", "fid_score"):     # installed-but-unusable here (accelerate probes wandb's spec) or absent
        sys.modules[m] = MagicMock()
    setup_out = []
    keep = ("mode", "clip", "mixed_precision", "learning_rate", "batch", "gradient_accumulation_steps", "epoch", "poison_rate", "sde_type",
            "psi", "solver_type", "ve_scale", "sched", "fclip", "R_trigger_only", "dataset_load_mode", "ext_poison_rate", "task",
            "infer_steps", "infer_start", "eval_max_batch", "sample_ep", "ddim_eta", "ckpt", "dataset", "trigger", "target", "device_ids")

    def rel(pth):
        return None if pth is None else os.path.relpath(pth, tmp)

    def record(tag, argv, cfg):
        setup_out.append({"tag": tag, "argv": [("<RESULT>" if x == tmp else x.replace(tmp, "<RESULT>")) for x in argv],
                          "config": {k: (v.replace(tmp, "<RESULT>") if isinstance(v, str) else v)
                                     for k, v in ((k, getattr(cfg, k, "<absent>")) for k in keep)},
                          "output_dir": rel(cfg.output_dir), "ckpt_path": rel(cfg.ckpt_path), "data_ckpt_path": rel(cfg.data_ckpt_path),
                          "files": sorted(os.listdir(cfg.output_dir))})

    try:
        state["argv"] = train_argv
        try:
            import VillanDiffusion  # noqa: F401,F811
        except _Stop:
            pass
        mod = state["mod"]
        record("train_cfg1", train_argv, mod.config)
        run_dir = mod.config.output_dir
        for tag, argv in cases[1:]:
            state["argv"] = argv
            record(tag, argv, mod.setup())
        for tag, extra in (("sampling", ["--mode", "sampling", "--sched", "UNIPC-SCHED", "--infer_steps", "20", "--eval_max_batch", "64"]),
                           ("measure_inpaint", ["--mode", "measure", "--task", "poisoned_inpaint_box", "--infer_start", "10", "--inpaint_mul", "1.5",
                                                "--fclip", "w", "--sample_ep", "3"]),
                           ("resume", ["--mode", "resume"])):
            argv = extra + ["--ckpt", run_dir]
            state["argv"] = argv
            record(tag, argv, mod.setup())
        errors = []
        for tag, argv in (("train_with_sample_ep", train_argv + ["--sample_ep", "2"]),
                          ("sampling_with_epoch", ["--mode", "sampling", "--ckpt", run_dir, "--epoch", "3"]),
                          ("batch_not_divisor", ["--mode", "train", "--dataset", "CIFAR10", "--batch", "48", "--result", tmp, "-o"]),
                          ("batch_too_big", ["--mode", "train", "--dataset", "CIFAR10", "--batch", "256", "--result", tmp, "-o"]),
                          ("exists_no_overwrite", [x for x in train_argv if x != "-o"])):
            state["argv"] = argv
            try:
                mod.setup()
                errors.append({"tag": tag, "error": None, "argv": [x.replace(tmp, "<RESULT>") for x in argv]})
            except Exception as e:  # noqa: BLE001
                errors.append({"tag": tag, "error": type(e).__name__, "argv": [x.replace(tmp, "<RESULT>") for x in argv]})
        # score.json keys
        keys = []
        for tag, over, kw in (("generate", {}, dict(fid_sc=1.0, mse_sc=2.0, ssim_sc=3.0)),
                              ("generate_ep_clip_eta", {"sample_ep": 7, "clip": True, "sched": "DDIM-SCHED", "infer_steps": 50, "ddim_eta": 0.5},
                               dict(fid_sc=1.0, mse_sc=2.0, ssim_sc=3.0)),
                              ("inpaint", {"task": "poisoned_inpaint_line", "sched": "UNIPC-SCHED", "infer_steps": 20, "clip": False},
                               dict(lpips_sc=1.0, mse_sc=2.0, ssim_sc=3.0))):
            c = copy.copy(mod.config)
            c.clip, c.sched, c.sample_ep, c.ddim_eta, c.task = False, None, None, None, "generate"
            for k, v in over.items():
                setattr(c, k, v)
            d = tempfile.mkdtemp(dir=tmp)
            c.output_dir = d
            sc = mod.update_score_file(config=c, score_file="score.json", **kw)
            keys.append({"tag": tag, "overrides": over, "keys": sorted(sc.keys())})
        a_, b_ = torch.rand(5, 3, 8, 8, generator=torch.Generator().manual_seed(3)), torch.rand(5, 3, 8, 8, generator=torch.Generator().manual_seed(4))
        metric = {"inputs": "torch.rand(5, 3, 8, 8) with manual_seed 3 (a) and 4 (b)", "mse_batch": mod.Metric.mse_batch(a=a_, b=b_, max_batch_n=2),
                  "mse_thres_batch": mod.Metric.mse_thres_batch(a=a_, b=b_, thres=0.17, max_batch_n=2)}
    finally:
        builtins.__import__ = real_import
        argparse.ArgumentParser.parse_args = orig_parse
        ref_dataset.DatasetLoader.__init__ = orig_init
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(OUT, "driver_setup.json"), "w") as f:
        json.dump({"setup": setup_out, "errors": errors, "score_keys": keys, "metric": metric}, f, indent=1, sort_keys=True, default=str)

    # --- the scheduler / pipeline factory (model.py:599-776): with diffusers mocked, every scheduler / pipeline class is a distinct
    # MagicMock, so calling the reference's private per-SDE factories records WHICH class each --sched builds and WITH WHAT kwargs
    DMS = ref_model.DiffuserModelSched
    sched_cls = ["DDPMScheduler", "DDIMScheduler", "DPMSolverMultistepScheduler", "UniPCMultistepScheduler", "PNDMScheduler",
                 "DEISMultistepScheduler", "HeunDiscreteScheduler", "LMSDiscreteScheduler", "ScoreSdeVeScheduler", "KarrasVeScheduler"]
    pipe_cls = ["DDPMPipeline", "DDIMPipeline", "PNDMPipeline", "ScoreSdeVePipeline", "LDMPipeline", "KarrasVePipeline", "DiffusionPipeline"]

    def plain(v):
        return v if isinstance(v, (str, int, float, bool, type(None))) else repr(type(v).__name__)

    def probe(fn_name, sched, **kw):
        for n in sched_cls + pipe_cls:
            getattr(ref_model, n).reset_mock()
        fn = getattr(DMS, "_DiffuserModelSched__" + fn_name)
        try:
            model, vae, ns, get_pipeline = fn("some/ckpt", noise_sched_type=sched, **kw)
        except Exception as e:  # noqa: BLE001
            return {"error": type(e).__name__}
        built = [(n, getattr(ref_model, n).call_args) for n in sched_cls if getattr(ref_model, n).call_args is not None]
        out = {"scheduler": None, "kwargs": None, "has_vae": vae is not None}
        if built:
            out["scheduler"] = built[0][0]
            out["kwargs"] = {k: plain(v) for k, v in built[0][1].kwargs.items()}
        acc = MagicMock()
        try:
            get_pipeline(acc, MagicMock(), vae, ns)
        except TypeError:      # model.py:767-768: the LDM + LMSD branch returns the 2-argument generator -> the driver's 4-argument call fails
            out["pipeline"], out["pipeline_kwargs"] = "TypeError", None
            return out
        used = [(n, getattr(ref_model, n).call_args) for n in pipe_cls if getattr(ref_model, n).call_args is not None]
        out["pipeline"] = used[0][0] if used else None
        out["pipeline_kwargs"] = sorted(k for k in used[0][1].kwargs) if used else None
        return out

    vp_names = ["DDPM_SCHED", "DDIM_SCHED", "DPM_SOLVER_PP_O1_SCHED", "DPM_SOLVER_O1_SCHED", "DPM_SOLVER_PP_O2_SCHED", "DPM_SOLVER_O2_SCHED",
                "DPM_SOLVER_PP_O3_SCHED", "DPM_SOLVER_O3_SCHED", "UNIPC_SCHED", "PNDM_SCHED", "DEIS_SCHED", "HEUN_SCHED", "LMSD_SCHED"]
    ve_names = ["SCORE_SDE_VE_SCHED", "EDM_VE_SCHED", "EDM_VE_SDE_SCHED", "EDM_VE_ODE_SCHED"]
    factory = {"vp": {}, "ldm": {}, "ve": {}}
    for clip in (True, False):
        for n in vp_names + [None]:
            v = getattr(DMS, n) if n else None
            factory["vp"][f"{v}|clip={clip}"] = probe("get_model_sched_vp", v, clip_sample=clip)
            factory["ldm"][f"{v}|clip={clip}"] = probe("get_model_sched_ldm", v, clip_sample=clip)
        for n in ve_names + [None]:
            v = getattr(DMS, n) if n else None
            factory["ve"][f"{v}|clip={clip}"] = probe("get_model_sched_ve", v, clip_sample=clip)
    factory["vp"]["LDM-SCHED-like unknown|clip=False"] = probe("get_model_sched_vp", "NO-SUCH-SCHED", clip_sample=False)
    factory["ve"]["DDPM-SCHED|clip=False"] = probe("get_model_sched_ve", DMS.DDPM_SCHED, clip_sample=False)
    with open(os.path.join(OUT, "factory_table.json"), "w") as f:
        json.dump(factory, f, indent=1, sort_keys=True)

    flags = []
    for a in captured["parser"]._actions:
        if a.option_strings and a.dest != "help":
            flags.append({"dest": a.dest, "opts": sorted(a.option_strings), "type": getattr(a.type, "__name__", None),
                          "choices": list(a.choices) if a.choices else None, "required": bool(a.required),
                          "store_true": isinstance(a, argparse._StoreTrueAction)})
    with open(os.path.join(OUT, "cli_flags.json"), "w") as f:
        json.dump(sorted(flags, key=lambda d: d["dest"]), f, indent=1)
    with open(os.path.join(OUT, "driver_defaults.json"), "w") as f:
        json.dump({"module_consts": captured.get("module_consts", {}), "training_config": captured.get("training_config", {}),
                   "naming": captured.get("naming", [])}, f, indent=1, sort_keys=True)
    print("wrote", len(tables), len(batch), len(boxes), len(misc), "arrays")


if __name__ == "__main__":
    main()
