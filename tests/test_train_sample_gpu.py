"""End-to-end parity on the GPU against the CPU oracle: optimiser steps (accumulate / clip / Adam / LR), full sampling
loops for every native sampler (north_star: denoised images within 1e-3 relative, timestep indices bit-exact), the GPU
data path, and the drop-in CLI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import backdoor_ref as BR  # noqa: E402
from oracle import schedulers_ref as R  # noqa: E402
from oracle.loss_ref import LossFnRef, SDE_VP  # noqa: E402
from oracle.unet_ref import UNet2DModelRef  # noqa: E402
from villandiffusion_amd import schedulers as S  # noqa: E402
from villandiffusion_amd.dataset import DatasetLoader, synthetic_images  # noqa: E402
from villandiffusion_amd.loss import LossFn  # noqa: E402
from villandiffusion_amd.pipelines import DDIMPipeline, DDPMPipeline, PNDMPipeline  # noqa: E402
from villandiffusion_amd.trainer import Trainer  # noqa: E402
from villandiffusion_amd.unet import UNet2DModel  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def nets():
    torch.manual_seed(0)
    ref = UNet2DModelRef()
    net = UNet2DModel()
    net.load_state_dict(ref.state_dict())
    return ref, net


@pytest.mark.parametrize("conv_math", ["bf16x3", "f32"])
def test_training_steps_match_oracle(nets, conv_math):
    """3 optimiser steps with gradient accumulation 2 (6 micro-batches of 4): loss per micro-step, LR, parameters -- in both
    convolution arithmetics (split-precision bf16 products, the default, and exact f32)."""
    ref, _ = nets
    import copy
    ref = copy.deepcopy(ref)
    net = UNet2DModel()
    net.load_state_dict(ref.state_dict())
    net.conv_math = conv_math
    theta0 = {k: v.clone() for k, v in ref.state_dict().items()}
    G, lr, warm, total = 2, 2e-4, 2, 10
    opt = torch.optim.Adam(ref.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: R.cosine_with_warmup_lambda(s, warm, total))
    lf_ref = LossFnRef(R.DDPMSchedulerRef(), SDE_VP, psi=0.5, solver_type="ode")
    lf = LossFn(S.DDPMScheduler(), "SDE-VP", psi=0.5, solver_type="ode")
    tr = Trainer(net, lf, lr=lr, total_steps=total, warmup_steps=warm, grad_accum=G)
    g = torch.Generator().manual_seed(7)
    for micro in range(6):
        x0 = torch.rand(4, 3, 32, 32, generator=g) * 2 - 1
        Rr = torch.rand(4, 3, 32, 32, generator=g) * 2 - 1
        Rr[:2] = 0
        eps = torch.randn(4, 3, 32, 32, generator=g)
        t = torch.randint(0, 1000, (4,), generator=g)
        l_ref = lf_ref.p_loss(ref, x0, Rr, t, noise=eps)
        (l_ref / G).backward()
        lr_now = tr.lr
        l = tr.train_step({"target": x0.cuda(), "pixel_values": Rr.cuda()}, t.cuda(), noise=eps.cuda())
        assert abs(float(l) - float(l_ref)) <= 2e-5 * abs(float(l_ref)), (micro, float(l), float(l_ref))
        if (micro + 1) % G == 0:
            assert abs(lr_now - opt.param_groups[0]["lr"]) < 1e-12
            torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
            opt.step()
            sched.step()
            opt.zero_grad()
    worst = max(rel(net.state_dict()[k], v) for k, v in ref.state_dict().items())
    # the trajectory as a whole: the 3-step update of all 35.7 M parameters against the oracle's, in L2
    num = sum(float(((net.state_dict()[k].cpu().double() - v.double()) ** 2).sum()) for k, v in ref.state_dict().items())
    den = sum(float(((v.double() - theta0[k].double()) ** 2).sum()) for k, v in ref.state_dict().items())
    upd = (num / den) ** 0.5
    print(f"[parity] parameters after 3 optimiser steps ({conv_math}): worst rel_err {worst:.3e}, update L2 error {upd:.3e}")
    # Early Adam steps are ~lr*sign(g): an element whose gradient is at rounding level may move by up to 2*lr per step the other way,
    # i.e. O(lr / max|w|) ~ 1e-3 in this per-tensor metric whatever the arithmetic; the split-precision gradients (5e-5 of scale
    # instead of 2e-5) put a few more elements in that band.
    assert worst < (2e-3 if conv_math == "bf16x3" else 1e-3)
    assert upd < 1e-3          # measured 6e-5 (bf16x3) / 1.2e-5 (f32)
    assert tr.sched_step == 3 and tr.opt.step_count == 3


SAMPLERS = [
    ("DDPM-20", lambda: S.DDPMScheduler(clip_sample=False), lambda: R.DDPMSchedulerRef(clip_sample=False), DDPMPipeline, 20),
    ("DDPM-clip-20", lambda: S.DDPMScheduler(clip_sample=True), lambda: R.DDPMSchedulerRef(clip_sample=True), DDPMPipeline, 20),
    ("DDIM-50", lambda: S.DDIMScheduler(clip_sample=False), lambda: R.DDIMSchedulerRef(clip_sample=False), DDIMPipeline, 50),
    ("DPM_PP_O2-20", lambda: S.DPMSolverMultistepScheduler(), lambda: R.DPMSolverMultistepSchedulerRef(), PNDMPipeline, 20),
    ("DPM_O3-20", lambda: S.DPMSolverMultistepScheduler(solver_order=3, algorithm_type="dpmsolver"),
     lambda: R.DPMSolverMultistepSchedulerRef(solver_order=3, algorithm_type="dpmsolver"), PNDMPipeline, 20),
    ("UNIPC-20", lambda: S.UniPCMultistepScheduler(), lambda: R.UniPCMultistepSchedulerRef(), PNDMPipeline, 20),
    # SURVEY §8f.3 (reference model.py:641-652)
    ("PNDM-20", lambda: S.PNDMScheduler(), lambda: R.PNDMSchedulerRef(), PNDMPipeline, 20),
    ("DEIS-20", lambda: S.DEISMultistepScheduler(), lambda: R.DEISMultistepSchedulerRef(), PNDMPipeline, 20),
    ("HEUN-10", lambda: S.HeunDiscreteScheduler(), lambda: R.HeunDiscreteSchedulerRef(), PNDMPipeline, 10),
    ("LMSD-20", lambda: S.LMSDiscreteScheduler(), lambda: R.LMSDiscreteSchedulerRef(), PNDMPipeline, 20),
]


@pytest.mark.parametrize("name,mk,mkref,pipe_cls,n", SAMPLERS)
def test_sampling_loop_matches_oracle(nets, name, mk, mkref, pipe_cls, n):
    ref, net = nets
    init = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(11))
    sched, sref = mk(), mkref()
    pipe = pipe_cls(net, sched)
    out = pipe(batch_size=2, generator=torch.Generator().manual_seed(5), init=init, num_inference_steps=n, output_type=None,
               save_every_step=True)
    with torch.no_grad():
        x_ref = R.sample_loop(ref, sref, init.clone(), n, generator=torch.Generator().manual_seed(5))
    assert torch.equal(sched.timesteps, sref.timesteps) and sched.timesteps.dtype == sref.timesteps.dtype
    assert hasattr(sref, "sigmas") or sched.timesteps.dtype == torch.int64
    img_ref = (x_ref / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    err = float(np.abs(out.images - img_ref).max() / np.abs(img_ref).max())
    print(f"[parity] {name}: denoised image max-rel-err {err:.3e}")
    assert out.images.shape == (2, 32, 32, 3) and out.images.dtype == np.float32
    assert err <= 1e-3
    assert len(out.movie) == len(sched.timesteps) + 1 and np.allclose(out.movie[0], (init / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy())
    assert np.array_equal(out.movie[-1], out.images)


def test_start_from_and_device_rng(nets):
    _, net = nets
    sched = S.DDPMScheduler(clip_sample=False)
    pipe = DDPMPipeline(net, sched)
    init = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    sched.device_rng_seed = 42
    a = pipe(batch_size=2, init=init, num_inference_steps=1000, start_from=990, return_tensor=True)
    sched._rng_offset = 0
    b = pipe(batch_size=2, init=init, num_inference_steps=1000, start_from=990, return_tensor=True)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())          # in-kernel Philox stream is reproducible
    sched.device_rng_seed = None


def test_gpu_data_path_matches_oracle():
    imgs = synthetic_images(n=512)
    dsl = DatasetLoader("X", root=ROOT, images=imgs, batch_size=64, seed=3)
    dsl.set_poison("STOP_SIGN_14", "HAT", poison_rate=0.25).prepare_dataset("FIXED")
    ids = torch.arange(300, 428)
    flips = (torch.arange(128) % 3 == 0)
    batch = dsl.make_batch(ids, flip_bits=flips, full=True)
    pos = ids.numpy()
    pv_ref, tg_ref = BR.poison_batch_ref(torch.from_numpy(imgs[dsl._index[pos]]), torch.from_numpy(dsl._flags[pos] & 1), dsl.trigger,
                                         dsl.target, -1.0, 1.0, flip=flips.to(torch.uint8))
    assert torch.equal(batch["pixel_values"].cpu(), pv_ref) and torch.equal(batch["target"].cpu(), tg_ref)
    assert set(batch.keys()) == {"image", "pixel_values", "pixel_values_trigger", "trigger", "target", "label", "is_clean"}
    assert int((~batch["is_clean"]).sum()) == int((dsl._flags[pos] & 1).sum()) > 0
    n = sum(b["pixel_values"].shape[0] for b in dsl.get_dataloader(full=False))
    assert n == 512


def test_cli_train_and_sampling_end_to_end(tmp_path):
    """BASELINE config #1 plumbing at toy size: --batch 4 style accumulation is covered above; here the drop-in CLI runs a
    1-epoch fine-tune on the synthetic set and re-samples from the saved diffusers-layout checkpoint."""
    res = str(tmp_path / "exp")
    env = dict(os.environ, PYTHONPATH=ROOT)
    code = ("import sys, numpy as np; sys.argv=['VillanDiffusion.py']+%r; import villandiffusion_amd.dataset as D;"
            "D.synthetic_images=(lambda f: (lambda n=60000, **k: f(n=256, **k)))(D.synthetic_images);"
            "import VillanDiffusion as V; V.TrainingConfig.eval_sample_n=4; V.main()")
    argv = ["--mode", "train", "--dataset", "SYNTHETIC-CIFAR10", "--batch", "64", "--epoch", "1", "--poison_rate", "0.1", "--trigger", "BOX_14",
            "--target", "HAT", "--ckpt", "DDPM-32-DEFAULT", "--fclip", "o", "-o", "--result", res, "--sched", "DDIM-SCHED", "--infer_steps", "5",
            "--save_image_epochs", "1", "--save_model_epochs", "1"]
    out = subprocess.run([sys.executable, "-c", code % (argv,)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    run = os.path.join(res, os.listdir(res)[0])
    for f in ("args.json", "config.json", "model_index.json", "unet/config.json", "unet/diffusion_pytorch_model.safetensors",
              "scheduler/scheduler_config.json", "samples/0000.png", "samples/final.png", "backdoor_samples/final.png", "ckpt/trainer.pt", "data.ckpt"):
        assert os.path.exists(os.path.join(run, f)), f                 # grids carry the 0-based epoch index + 'final' (reference :1177-1195)
    assert not os.path.exists(os.path.join(run, "samples", "0001.png"))
    assert json.load(open(os.path.join(run, "config.json")))["gradient_accumulation_steps"] == 2
    dck = torch.load(os.path.join(run, "data.ckpt"))
    assert (dck["epoch"], dck["step"]) == (0, 4) and dck["loader"]["flip_gen"] is not None      # + the device flip generator's state (resume)
    # --mode resume with --result that is not the cwd: the model comes from the run directory setup() resolved; the loop restarts AT
    # the recorded epoch index like the reference (:457-461 + range(start_epoch, epoch))
    os.remove(os.path.join(run, "samples", "final.png"))
    argv_r = ["--mode", "resume", "--ckpt", os.path.basename(run), "--result", res]
    out = subprocess.run([sys.executable, "-c", code % (argv_r,)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    dck = torch.load(os.path.join(run, "data.ckpt"))
    assert (dck["epoch"], dck["step"]) == (0, 8) and os.path.exists(os.path.join(run, "samples", "final.png"))
    assert torch.load(os.path.join(run, "ckpt", "trainer.pt"), map_location="cpu")["optimizer"]["step"] == 4      # 2 + 2 sync steps at G = 2
    argv2 = ["--mode", "sampling", "--ckpt", run, "--sched", "DPM_SOLVER_PP_O2-SCHED", "--infer_steps", "5"]
    out = subprocess.run([sys.executable, "-c", code % (argv2,)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert os.path.exists(os.path.join(run, "samples", "final.png")) and os.path.exists(os.path.join(run, "sampling.json"))
    # measure (MSE / SSIM vs target -> score.json, reference key naming) and one inpaint task from the saved checkpoint
    env_m = dict(env, VILLAN_CFG_OVERRIDES=json.dumps({"measure_sample_n": 16, "measure_inpaint_sample_n": 6}))   # reference defaults: 10000 / 1024
    argv3 = ["--mode", "measure", "--ckpt", run, "--sched", "DDIM-SCHED", "--infer_steps", "4", "--eval_max_batch", "4"]
    out = subprocess.run([sys.executable, "-c", code % (argv3,)], cwd=ROOT, env=env_m, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    sc = json.load(open(os.path.join(run, "score.json")))
    assert set(sc) == {"FID_noclip_DDIM-SCHED-4_16", "MSE_noclip_DDIM-SCHED-4_16", "SSIM_noclip_DDIM-SCHED-4_16"}
    assert sc["FID_noclip_DDIM-SCHED-4_16"] is None and 0 <= sc["MSE_noclip_DDIM-SCHED-4_16"] <= 1 and -1 <= sc["SSIM_noclip_DDIM-SCHED-4_16"] <= 1
    assert len(os.listdir(os.path.join(run, "backdoor_noclip_DDIM-SCHED-4_16"))) == 16
    # ... and the scores themselves are held to the ORACLE: the same 16 backdoor samples (checkpoint weights, seeded CPU noise + trigger,
    # 4 DDIM steps in chunks of --eval_max_batch) from the CPU UNet and sampler, quantised to 8 bits like the saved PNGs, scored by
    # oracle/metrics_ref.py (VillanDiffusion.py:1050-1091)
    from safetensors.torch import load_file
    from oracle.metrics_ref import mse_ref, ssim_ref
    from villandiffusion_amd.dataset import Backdoor
    ref = UNet2DModelRef()
    ref.load_state_dict(load_file(os.path.join(run, "unet", "diffusion_pytorch_model.safetensors")))
    bd = Backdoor(root=ROOT)
    trig = bd.get_trigger("BOX_14", 3, 32, -1.0, 1.0)
    tgt01 = (bd.get_target("HAT", trig, vmin=-1.0, vmax=1.0) / 2 + 0.5).clamp(0, 1)
    noise = torch.randn((16, 3, 32, 32), generator=torch.Generator().manual_seed(0)) + trig[None]
    outs = []
    with torch.no_grad():
        for c in torch.split(noise, 4):
            x = R.sample_loop(ref, R.DDIMSchedulerRef(clip_sample=False), c.clone(), 4)
            outs.append(((x / 2 + 0.5).clamp(0, 1) * 255).round() / 255)
    gen_ref = torch.cat(outs).numpy()
    tg = tgt01[None].expand(16, -1, -1, -1).numpy()
    mse_o, ssim_o = mse_ref(gen_ref, tg), ssim_ref(gen_ref, tg)
    print(f"[parity] measure: MSE {sc['MSE_noclip_DDIM-SCHED-4_16']:.6f} (oracle {mse_o:.6f}), SSIM {sc['SSIM_noclip_DDIM-SCHED-4_16']:.6f} (oracle {ssim_o:.6f})")
    assert abs(sc["MSE_noclip_DDIM-SCHED-4_16"] - mse_o) <= 1e-4 + 1e-3 * mse_o         # 8-bit quantisation: a pixel at a rounding boundary may flip by 1/255
    assert abs(sc["SSIM_noclip_DDIM-SCHED-4_16"] - ssim_o) <= 2e-3
    from PIL import Image
    png = np.stack([np.asarray(Image.open(os.path.join(run, "backdoor_noclip_DDIM-SCHED-4_16", f"{i}.png")).convert("RGB")) for i in range(16)])
    diff = np.abs(png.astype(np.int32) - (gen_ref * 255).round().astype(np.int32).transpose(0, 2, 3, 1))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.01                       # the saved images ARE the oracle's, up to rounding-boundary pixels
    # the same for a denoise task: MSE / SSIM of the recovered images (reference measure_inpaints; LPIPS needs AlexNet weights)
    argv5 = ["--mode", "measure", "--ckpt", run, "--sched", "DDIM-SCHED", "--infer_steps", "10", "--infer_start", "6", "--eval_max_batch", "4",
             "--task", "poisoned_denoise"]
    out = subprocess.run([sys.executable, "-c", code % (argv5,)], cwd=ROOT, env=env_m, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    sc = json.load(open(os.path.join(run, "score.json")))
    k = "MSE_noclip_DDIM-SCHED-10_6_poisoned_denoise"             # key carries measure_inpaint_sample_n
    assert k in sc and 0 <= sc[k] <= 4 and "SSIM_noclip_DDIM-SCHED-10_6_poisoned_denoise" in sc
    assert sc["LPIPS_noclip_DDIM-SCHED-10_6_poisoned_denoise"] is None
    argv4 = ["--mode", "sampling", "--ckpt", run, "--sched", "DDIM-SCHED", "--infer_steps", "10", "--infer_start", "6", "--task", "poisoned_inpaint_box"]
    out = subprocess.run([sys.executable, "-c", code % (argv4,)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert os.path.exists(os.path.join(run, "inpaint_box_poisoned_samples_DDIM-SCHED_10_st6.0_m1.0", "final.png"))


def test_ve_loss_and_score_sde_sampler_match_oracle(nets):
    """SDE-VE row: the VE loss (model fed sigma_t, prediction scaled by -sigma) and the predictor-corrector sampler, with the
    DDPM-style UNet standing in for NCSN++."""
    from oracle.loss_ref import SDE_VE
    from villandiffusion_amd.pipelines import ScoreSdeVePipeline
    ref, net = nets
    sref = R.ScoreSdeVeSchedulerRef(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, snr=0.075)
    s = S.ScoreSdeVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, snr=0.075)
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(4, 3, 32, 32, generator=g)
    Rr = torch.rand(4, 3, 32, 32, generator=g)
    Rr[:2] = 0
    eps = torch.randn(4, 3, 32, 32, generator=g)
    t = torch.tensor([0, 100, 1000, 1999])
    l_ref = LossFnRef(sref, SDE_VE, psi=0).p_loss(ref, x0, Rr, t, noise=eps)
    net.zero_grad()
    l = LossFn(s, "SDE-VE", psi=0).p_loss(net, x0.cuda(), Rr.cuda(), t.cuda(), noise=eps.cuda())
    assert abs(float(l) - float(l_ref)) <= 2e-5 * abs(float(l_ref)), (float(l), float(l_ref))
    l.backward()
    assert bool(torch.isfinite(net.flat_grad).all()) and float(net.flat_grad.abs().max()) > 0
    net.zero_grad()
    # sampler: 6 predictor-corrector steps with fixed noises
    n = 6
    init = torch.randn(2, 3, 32, 32, generator=g) * 380.0
    zs = [torch.randn(2, 3, 32, 32, generator=g) for _ in range(2 * n)]
    sref.set_timesteps(n); sref.set_sigmas(n)
    x = init.clone()
    with torch.no_grad():
        for i, tt in enumerate(sref.timesteps):
            sig = sref.sigmas[i] * torch.ones(2)
            x = sref.step_correct(ref(x, sig)[0], x, noise=zs[2 * i]).prev_sample
            out = sref.step_pred(ref(x, sig)[0], tt, x, noise=zs[2 * i + 1])
            x, mean_ref = out.prev_sample, out.prev_sample_mean
    s.set_timesteps(n); s.set_sigmas(n)
    xd = init.cuda()
    with torch.no_grad():
        for i, tt in enumerate(s.timesteps):
            sig = (s.sigmas[i] * torch.ones(2)).cuda()
            xd = s.step_correct(net(xd, sig)[0], xd, noise=zs[2 * i].cuda()).prev_sample
            out = s.step_pred(net(xd, sig)[0], tt, xd, noise=zs[2 * i + 1].cuda())
            xd, mean = out.prev_sample, out.prev_sample_mean
    err = rel(mean, mean_ref)
    print(f"[parity] ScoreSDE-VE {n} PC steps: rel_err {err:.3e}")
    assert err <= 1e-3
    res = ScoreSdeVePipeline(net, s)(batch_size=2, generator=torch.Generator().manual_seed(0), num_inference_steps=3, output_type=None)
    assert res.images.shape == (2, 32, 32, 3) and float(res.images.min()) >= 0 and float(res.images.max()) <= 1


@pytest.mark.parametrize("churn", [80.0, 0.0])
def test_karras_ve_pipeline_matches_oracle(nets, churn):
    """EDM_VE / EDM_VE_ODE samplers (reference model.py:685-693) with the DDPM-style UNet standing in for NCSN++."""
    from villandiffusion_amd.pipelines import KarrasVePipeline
    ref, net = nets
    n = 8
    s = S.KarrasVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, s_churn=churn)
    sref = R.KarrasVeSchedulerRef(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, s_churn=churn)
    init = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(4)) * 380.0
    out = KarrasVePipeline(net, s)(batch_size=2, generator=torch.Generator().manual_seed(9), init=init, num_inference_steps=n,
                                   return_tensor=True)
    x_ref = R.karras_ve_loop(ref, sref, init.clone(), n, generator=torch.Generator().manual_seed(9))
    err = float((out.cpu() - x_ref).abs().max() / x_ref.abs().max())
    print(f"[parity] KarrasVe churn={churn}: final sample max-rel-err {err:.3e}")
    assert err <= 1e-3


def test_celebahq256_config_train_step_with_image_trigger():
    """BASELINE config #4 shape of work at reduced width: 256x256 images, STOP_SIGN_14 -> CAT (image-based trigger / target from
    static/), GPU stamping vs the oracle's per-sample rule, one fine-tune step of a 6-level UNet and a UniPC sampling call."""
    from villandiffusion_amd.pipelines import PNDMPipeline
    dsl = DatasetLoader("SYNTHETIC-CELEBA-HQ", root=ROOT, batch_size=4, seed=0, images=synthetic_images(n=16, size=256))
    dsl.set_poison("STOP_SIGN_14", "CAT", poison_rate=0.5).prepare_dataset(mode="FIXED")
    assert tuple(dsl.trigger.shape) == (3, 256, 256) and tuple(dsl.target.shape) == (3, 256, 256)
    ids = torch.arange(16)
    batch = dsl.make_batch(ids, flip_bits=torch.zeros(16, dtype=torch.bool))
    imgs_u8 = torch.from_numpy(dsl._images[dsl._index])                      # NHWC uint8
    pv_ref, tg_ref = BR.poison_batch_ref(imgs_u8, torch.from_numpy((dsl._flags & 1).astype(bool)), dsl.trigger, dsl.target, -1.0, 1.0)
    assert torch.equal(batch["pixel_values"].cpu(), pv_ref) and torch.equal(batch["target"].cpu(), tg_ref)
    assert int((dsl._flags & 1).sum()) == 8
    net = UNet2DModel(sample_size=256, block_out_channels=(32, 32, 64, 64, 128, 128),
                      down_block_types=("DownBlock2D",) * 4 + ("AttnDownBlock2D", "DownBlock2D"),
                      up_block_types=("UpBlock2D", "AttnUpBlock2D") + ("UpBlock2D",) * 4, norm_num_groups=8)
    sched = S.UniPCMultistepScheduler()
    tr = Trainer(net, LossFn(S.DDPMScheduler(), "SDE-VP"), lr=6e-5, total_steps=100, warmup_steps=10)
    b4 = dsl.make_batch(torch.tensor([0, 1, 14, 15]), full=False)
    l = tr.train_step(b4, torch.tensor([5, 300, 600, 999], device="cuda"))
    assert float(l) == float(l) and bool(torch.isfinite(net.flat_param).all())
    out = PNDMPipeline(net, sched)(batch_size=2, generator=torch.Generator().manual_seed(0), num_inference_steps=3, output_type=None)
    assert out.images.shape == (2, 256, 256, 3)


def test_batches_addressed_by_device_positions_need_no_host_copy_and_equal_the_host_path():
    """Training batches (full=False) for device-resident positions: indices / poison flags gathered on the GPU (dataset._make_batch_resident);
    same tensors as the host-position path for the same flip bits; random flips are per-sample mirror images; the loader feeds device positions."""
    dsl = DatasetLoader("SYNTHETIC-CIFAR10", root=ROOT, batch_size=32, seed=0, images=synthetic_images(n=256, size=32))
    dsl.set_poison("BOX_14", "HAT", poison_rate=0.25).prepare_dataset(mode="FIXED")
    ids = torch.randperm(256, generator=torch.Generator().manual_seed(3))[:64]
    flips = torch.rand(64, generator=torch.Generator().manual_seed(4)) < 0.5
    host = dsl.make_batch(ids, flip_bits=flips, full=False)
    resident = dsl.make_batch(ids.cuda(), flip_bits=flips, full=False)
    assert set(host) == set(resident)
    for k in host:
        assert torch.equal(host[k], resident[k]), k
    assert int((dsl._flags[ids.numpy()] & 1).sum()) > 0                         # poisoned samples in the batch
    rnd = dsl.make_batch(ids.cuda(), full=False)
    noflip = dsl.make_batch(ids.cuda(), flip_bits=torch.zeros(64, dtype=torch.bool), full=False)
    same = (rnd["image"] == noflip["image"]).flatten(1).all(1)
    mirrored = (rnd["image"] == noflip["image"].flip(-1)).flatten(1).all(1)
    assert bool((same | mirrored).all()) and 8 < int(mirrored.sum()) < 56
    loader = dsl.get_dataloader(batch_size=32, full=False)
    assert loader._ids_dev is not None and loader._ids_dev.is_cuda
    n = sum(b["pixel_values"].shape[0] for b in loader)
    assert n == 256
    # R_trigger_only partitions (the per-batch flag check needs the host flags) and full batches take the host-position path even for device positions
    rto = DatasetLoader("SYNTHETIC-CIFAR10", root=ROOT, batch_size=32, seed=0, images=synthetic_images(n=256, size=32))
    rto.set_poison("BOX_14", "HAT", poison_rate=1.0).prepare_dataset(mode="FIXED", R_trigger_only=True)
    a = rto.make_batch(ids, flip_bits=flips, full=False)
    b = rto.make_batch(ids.cuda(), flip_bits=flips, full=False)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(b["pixel_values"][0], rto.trigger.to(b["pixel_values"].device))       # R = the trigger alone
    fa, fb = dsl.make_batch(ids, flip_bits=flips, full=True), dsl.make_batch(ids.cuda(), flip_bits=flips, full=True)
    assert set(fa) == set(fb) and all(torch.equal(fa[k], fb[k]) for k in fa)


def test_measure_sampling_job_runs_its_chunks_concurrently_with_the_same_images(tmp_path, monkeypatch):
    """sampling_io.batch_sampling_save (the 2 x N-image job of measure(), reference VillanDiffusion.py:1062-1067 / model.py:504-527): a rank's chunks go
    through pipelines.sample_concurrent where that gives the same PNGs -- deterministic samplers always, DDPM with the in-kernel noise the driver selects."""
    from PIL import Image
    from villandiffusion_amd import sampling_io as SIO
    from villandiffusion_amd.pipelines import DDIMPipeline, DDPMPipeline
    net = UNet2DModel(block_out_channels=(32, 64), down_block_types=("DownBlock2D", "AttnDownBlock2D"),
                      up_block_types=("AttnUpBlock2D", "UpBlock2D"), norm_num_groups=8)
    net.reset_parameters(0)
    init = torch.randn(10, 3, 32, 32, generator=torch.Generator().manual_seed(0))              # chunks of 4, 4, 2

    def pngs(d):
        return np.stack([np.asarray(Image.open(os.path.join(d, f"{i}.png"))) for i in range(10)])

    calls = []
    orig = DDIMPipeline.sample_concurrent

    def spy(self, inits, **kw):
        calls.append(len(inits))
        return orig(self, inits, **kw)

    monkeypatch.setattr(DDIMPipeline, "sample_concurrent", spy)
    pipe = DDIMPipeline(net, S.DDIMScheduler(clip_sample=False))
    SIO.batch_sampling_save(10, pipe, str(tmp_path / "conc"), init=init, max_batch_n=4, rng=torch.Generator().manual_seed(1), num_inference_steps=5)
    assert calls == [3]
    monkeypatch.setenv("VILLAN_SAMPLER_STREAMS", "1")
    SIO.batch_sampling_save(10, pipe, str(tmp_path / "seq"), init=init, max_batch_n=4, rng=torch.Generator().manual_seed(1), num_inference_steps=5)
    assert calls == [3]
    assert np.array_equal(pngs(tmp_path / "conc"), pngs(tmp_path / "seq"))
    monkeypatch.delenv("VILLAN_SAMPLER_STREAMS")
    # two ranks: each takes its own chunks and writes its own index range
    for r in range(2):
        SIO.batch_sampling_save(10, pipe, str(tmp_path / "ddp"), init=init, max_batch_n=4, num_inference_steps=5, rank=r, world=2)
    assert np.array_equal(pngs(tmp_path / "ddp"), pngs(tmp_path / "seq"))
    # DDPM: host-generator noise keeps the sequential path; with the in-kernel stream (what measure() selects) the chunks run concurrently
    dd = DDPMPipeline(net, S.DDPMScheduler())
    assert not SIO._concurrent_ok(dd, [(init[:4], 4), (init[4:8], 4)], None)
    dd.scheduler.device_rng_seed = 7
    assert SIO._concurrent_ok(dd, [(init[:4], 4), (init[4:8], 4)], None)
    SIO.batch_sampling_save(10, dd, str(tmp_path / "ddpm_a"), init=init, max_batch_n=4, num_inference_steps=6)
    dd.scheduler._rng_offset = 0
    SIO.batch_sampling_save(10, dd, str(tmp_path / "ddpm_b"), init=init, max_batch_n=4, num_inference_steps=6)
    a = pngs(tmp_path / "ddpm_a")
    assert np.array_equal(a, pngs(tmp_path / "ddpm_b")) and a.std() > 0                           # seeded: reproducible


def test_cli_resume_continues_from_the_checkpoint(tmp_path):
    """--mode resume (reference :454-461, 1103-1115): model from the run directory, optimiser / LR-schedule / counters from
    ckpt/trainer.pt and data.ckpt.  As in the reference the checkpoint records the 0-based index of the epoch it was written AFTER
    (`checkpoint(cur_epoch=epoch)`, :1181) and the loop restarts AT that index (`range(start_epoch, config.epoch)`, :1132): the last
    checkpointed epoch is run again."""
    res = str(tmp_path / "exp")
    env = dict(os.environ, PYTHONPATH=ROOT)
    code = ("import sys; sys.argv=['VillanDiffusion.py']+%r; import villandiffusion_amd.dataset as D;"
            "D.synthetic_images=(lambda f: (lambda n=60000, **k: f(n=256, **k)))(D.synthetic_images);"
            "import VillanDiffusion as V; V.TrainingConfig.eval_sample_n=4; V.main()")
    argv = ["--mode", "train", "--dataset", "SYNTHETIC-CIFAR10", "--batch", "128", "--epoch", "1", "--poison_rate", "0.1", "--trigger", "BOX_14",
            "--target", "HAT", "--ckpt", "DDPM-32-DEFAULT", "--fclip", "o", "-o", "--result", res, "--sched", "DDIM-SCHED", "--infer_steps", "3",
            "--save_image_epochs", "1", "--save_model_epochs", "1"]
    out = subprocess.run([sys.executable, "-c", code % (argv,)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    run = os.path.join(res, os.listdir(res)[0])
    st1 = torch.load(os.path.join(run, "ckpt", "trainer.pt"), map_location="cpu")
    d1 = torch.load(os.path.join(run, "data.ckpt"))
    assert d1["epoch"] == 0 and d1["step"] == 2 and st1["sched_step"] == 2 and st1["optimizer"]["step"] == 2
    w1 = {}
    from safetensors.torch import load_file
    w1 = load_file(os.path.join(run, "unet", "diffusion_pytorch_model.safetensors"))["conv_in.weight"].clone()
    args = json.load(open(os.path.join(run, "args.json")))
    args["epoch"] = 2                                   # ask for one more epoch
    json.dump(args, open(os.path.join(run, "args.json"), "w"))
    out = subprocess.run([sys.executable, "-c", code % (["--mode", "resume", "--ckpt", run],)], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    st2 = torch.load(os.path.join(run, "ckpt", "trainer.pt"), map_location="cpu")
    d2 = torch.load(os.path.join(run, "data.ckpt"))
    assert d2["epoch"] == 1 and d2["step"] == 6 and st2["sched_step"] == 6 and st2["optimizer"]["step"] == 6      # epochs 0 (again) and 1
    w2 = load_file(os.path.join(run, "unet", "diffusion_pytorch_model.safetensors"))["conv_in.weight"]
    assert not torch.equal(w1, w2) and bool(torch.isfinite(w2).all())
    assert float(st2["optimizer"]["exp_avg_sq"].abs().max()) > 0


@pytest.mark.parametrize("side_stream", [True, False])
def test_training_step_is_run_to_run_deterministic(side_stream):
    """Every reduction on the path has a fixed order (split-K slabs, grouped weight gradients, row / column sums, the loss, the gradient norm), so two
    runs from the same state give bit-identical parameters -- also with the weight gradients on their side stream.  Between the two settings only
    the grouping of the weight-gradient launches differs (the side stream flushes every 24 jobs, so the K ranges a layer is split into, and with
    them the summation order, change): same result to rounding."""
    def run(stream):
        net = UNet2DModel()
        net.reset_parameters(seed=3)
        net.wgrad_stream = stream
        tr = Trainer(net, LossFn(S.DDPMScheduler(), "SDE-VP", psi=1), lr=1e-3, total_steps=10, warmup_steps=0, grad_accum=2)
        g = torch.Generator().manual_seed(5)
        for i in range(4):
            x0 = (torch.rand(16, 3, 32, 32, generator=g) * 2 - 1).cuda()
            R_ = (torch.rand(16, 3, 32, 32, generator=g) * 2 - 1).cuda()
            R_[::2] = 0
            eps = torch.randn(16, 3, 32, 32, generator=g).cuda()
            t = torch.randint(0, 1000, (16,), generator=g).cuda()
            tr.train_step({"target": x0, "pixel_values": R_}, t, noise=eps)
        torch.cuda.synchronize()
        return net.flat_param.detach().clone()

    a, b = run(side_stream), run(side_stream)
    assert torch.equal(a, b)
    c = run(not side_stream)
    err = float((a - c).abs().max() / a.abs().max())
    print(f"[parity] side stream {side_stream} vs {not side_stream}: parameters after 2 optimiser steps differ by {err:.2e} (rounding of a different K split)")
    assert err < 1e-5


def test_concurrent_sampler_chunks_equal_sequential_chunks():
    """pipelines.sample_concurrent: chunks denoised two at a time, each on its own stream / HIP graph / split-K workspace / scheduler copy, stepping
    in lock-step on the host.  A chunk's result must not depend on what runs beside it: bit-identical to a sequential pipeline call started at the
    same Philox offset (DDPM: in-kernel noise), and for a multistep solver (DPM-Solver++: per-chunk history) with no noise at all."""
    from villandiffusion_amd.pipelines import DDPMPipeline, PNDMPipeline
    net = UNet2DModel()
    net.reset_parameters(seed=1)
    g = torch.Generator().manual_seed(3)
    inits = [torch.randn(8, 3, 32, 32, generator=g), torch.randn(8, 3, 32, 32, generator=g), torch.randn(5, 3, 32, 32, generator=g)]
    steps = 12
    sched = S.DDPMScheduler(clip_sample=False)
    sched.device_rng_seed = 42
    pipe = DDPMPipeline(net, sched)
    outs = pipe.sample_concurrent(inits, num_inference_steps=steps, n_streams=2)
    torch.cuda.synchronize()
    assert len(outs) == 3 and all(bool(torch.isfinite(o).all()) for o in outs)
    for ci, x0 in enumerate(inits):
        sched._rng_offset = pipe.chunk_rng_offset(ci, steps, max(c.numel() for c in inits))
        ref = pipe(batch_size=len(x0), init=x0, num_inference_steps=steps, return_tensor=True)
        assert torch.equal(ref, outs[ci]), ci
    assert not torch.equal(outs[0][:5], outs[2])                      # different chunks, different noise ranges
    p2 = PNDMPipeline(net, S.DPMSolverMultistepScheduler())
    o2 = p2.sample_concurrent(inits[:2], num_inference_steps=10, n_streams=2)
    for ci in range(2):
        ref = p2(batch_size=8, init=inits[ci], num_inference_steps=10, return_tensor=True)
        assert torch.equal(ref, o2[ci]), ci
    with pytest.raises(RuntimeError):                                  # noise from a CPU generator would depend on the interleaving
        DDPMPipeline(net, S.DDPMScheduler()).sample_concurrent(inits[:2], num_inference_steps=4)
