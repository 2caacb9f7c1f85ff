"""NCSN++: oracle known-answer tests on the CPU, HIP forward / backward vs the oracle on the GPU."""
import pytest
import torch

from oracle.ncsnpp_ref import NCSNppRef, downsample_2d, upsample_2d
from villandiffusion_amd.ncsnpp import NCSNppModel

SMALL = dict(sample_size=16, block_out_channels=(32, 64, 64),
             down_block_types=("SkipDownBlock2D", "AttnSkipDownBlock2D", "SkipDownBlock2D"),
             up_block_types=("SkipUpBlock2D", "AttnSkipUpBlock2D", "SkipUpBlock2D"), layers_per_block=2)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_fir_resampling_known_answers():
    """(1,3,3,1) FIR, factor 2: constants are preserved in the interior; the closed forms of the module docstring; the two
    operators are adjoint up to the factor 4 the backward pass uses."""
    x = torch.ones(1, 2, 8, 8)
    u, d = upsample_2d(x), downsample_2d(x)
    assert u.shape == (1, 2, 16, 16) and d.shape == (1, 2, 4, 4)
    assert torch.allclose(u[..., 2:-2, 2:-2], torch.ones(1, 2, 12, 12)) and abs(float(u[0, 0, 0, 0]) - 0.5625) < 1e-7
    assert torch.allclose(d[..., 1:-1, 1:-1], torch.ones(1, 2, 2, 2)) and abs(float(d[0, 0, 0, 0]) - 0.765625) < 1e-7
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(1, 1, 6, 6, generator=g), torch.randn(1, 1, 12, 12, generator=g)
    assert abs(float((upsample_2d(a) * b).sum()) - 4.0 * float((a * downsample_2d(b)).sum())) < 1e-4
    r = torch.arange(8.0).view(1, 1, 1, 8).expand(1, 1, 8, 8).contiguous()
    uu = upsample_2d(r)[0, 0, 8]
    assert abs(float(uu[6]) - (2 + 3 * 3) / 4) < 1e-5 and abs(float(uu[7]) - (3 * 3 + 4) / 4) < 1e-5


def test_state_dict_surface_and_size():
    """reference model.py:839-857 architecture: 61.9 M parameters (the 'NCSN++ cont.' size of Song et al. 2021)."""
    ref, net = NCSNppRef(), NCSNppModel(device="cpu")
    sr, sn = ref.state_dict(), net.state_dict()
    assert set(sr) == set(sn) and all(tuple(sr[k].shape) == tuple(sn[k].shape) for k in sr)
    assert sum(p.numel() for p in net.parameters()) == 61894924
    assert not net.time_proj.weight.requires_grad
    sd = dict(sr)
    sd["time_proj.W"] = sr["time_proj.weight"]                     # upstream stores the alias too
    net.load_state_dict(sd)
    assert torch.equal(net.state_dict()["up_blocks.0.skip_conv.weight"], sr["up_blocks.0.skip_conv.weight"])


def test_oracle_output_is_divided_by_sigma():
    torch.manual_seed(0)
    ref = NCSNppRef(**SMALL)
    x = torch.randn(2, 3, 16, 16)
    with torch.no_grad():
        y1 = ref(x, torch.tensor([2.0, 2.0]))[0]
        emb_same = ref.time_proj(torch.tensor([2.0]))
        assert emb_same.shape == (1, 64)
        assert y1.shape == x.shape and bool(torch.isfinite(y1).all())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,B,wscale", [(SMALL, 3, 0.1), (dict(layers_per_block=1), 2, 0.1), (SMALL, 3, 1.0)])
def test_hip_forward_backward_match_oracle(cfg, B, wscale):
    torch.manual_seed(1)
    ref = NCSNppRef(**cfg)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
        ref.time_proj.weight.mul_(wscale)         # 1.0 = the real scale (16): sin/cos arguments up to ~1e3 rad
    net = NCSNppModel(**cfg)
    net.load_state_dict(ref.state_dict())
    S = ref.config.sample_size
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(2))
    sig = torch.tensor([0.05, 1.7, 120.0][:B])
    y_ref = ref(x, sig)[0]
    w = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(3))
    (y_ref * w).sum().backward()
    net.zero_grad()
    y = net(x.cuda(), sig.cuda())[0]
    ef = rel(y, y_ref)
    (y * w.cuda()).sum().backward()
    gref = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    gmax = max(float(g.abs().max()) for g in gref.values())
    worst = (0.0, "")
    for n, p in net.named_parameters():
        if n not in gref:
            continue
        a, b = p.grad.detach().double().cpu(), gref[n].double()
        e = float((a - b).abs().max() / (b.abs().max() + 1e-4 * gmax))
        if e > worst[0]:
            worst = (e, n)
    print(f"[parity] NCSN++ fwd rel_err={ef:.3e}; worst param-grad rel_err={worst[0]:.3e} at {worst[1]}")
    assert ef < 1e-4 and worst[0] < 1e-3, (ef, worst)


@pytest.mark.gpu
def test_ve_training_step_sampling_and_disk_roundtrip_with_ncsnpp(tmp_path):
    """The SDE-VE row end to end with its own network: VE loss (model fed sigma_t, output already divided by sigma) vs the
    oracle, two optimiser steps, predictor-corrector sampling, save_pretrained -> from_pretrained picks NCSNppModel."""
    from oracle.loss_ref import LossFnRef
    from oracle.schedulers_ref import ScoreSdeVeSchedulerRef
    from villandiffusion_amd import schedulers as S
    from villandiffusion_amd.loss import LossFn
    from villandiffusion_amd.pipelines import DiffusionPipeline, ScoreSdeVePipeline
    from villandiffusion_amd.trainer import Trainer
    torch.manual_seed(0)
    ref = NCSNppRef(**SMALL)
    net = NCSNppModel(**SMALL)
    net.load_state_dict(ref.state_dict())
    kw = dict(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, snr=0.075)
    sched, sref = S.ScoreSdeVeScheduler(**kw), ScoreSdeVeSchedulerRef(**kw)
    g = torch.Generator().manual_seed(3)
    x0 = torch.rand(3, 3, 16, 16, generator=g)
    Rr = torch.rand(3, 3, 16, 16, generator=g)
    Rr[:1] = 0
    eps = torch.randn(3, 3, 16, 16, generator=g)
    t = torch.tensor([3, 900, 1999])
    l_ref = LossFnRef(sref, "SDE-VE", psi=0).p_loss(ref, x0, Rr, t, noise=eps)
    lf = LossFn(sched, "SDE-VE", psi=0)
    tr = Trainer(net, lf, lr=1e-4, total_steps=100, warmup_steps=10)
    w0 = net.time_proj.weight.detach().clone()
    l = tr.train_step({"target": x0.cuda(), "pixel_values": Rr.cuda()}, t.cuda(), noise=eps.cuda())
    assert abs(float(l) - float(l_ref)) <= 5e-5 * abs(float(l_ref)), (float(l), float(l_ref))
    l2 = tr.train_step({"target": x0.cuda(), "pixel_values": Rr.cuda()}, t.cuda(), noise=eps.cuda())
    assert bool(torch.isfinite(net.flat_param).all()) and float(l2) == float(l2)
    assert torch.equal(net.time_proj.weight, w0)                       # the Fourier features stay fixed
    pipe = ScoreSdeVePipeline(net, sched)
    out = pipe(batch_size=2, generator=torch.Generator().manual_seed(1), num_inference_steps=4, output_type=None)
    assert out.images.shape == (2, 16, 16, 3) and bool((out.images >= 0).all()) and bool((out.images <= 1).all())
    d = str(tmp_path / "ncsnpp")
    pipe.save_pretrained(d)
    pipe2 = DiffusionPipeline.from_pretrained(d)
    assert type(pipe2.unet).__name__ == "NCSNppModel" and torch.equal(pipe2.unet.flat_param, net.flat_param)


@pytest.mark.gpu
def test_cli_sde_ve_ncsnpp_train_and_sample(tmp_path):
    """The reference's score-based flow (run_score-basde_model_script.py): --sde_type SDE-VE --psi 0 with the NCSN++
    from-scratch id; fine-tune one tiny epoch, sample with the predictor-corrector sampler, re-load the checkpoint."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = str(tmp_path / "exp")
    env = dict(os.environ, PYTHONPATH=root)
    code = ("import sys; sys.argv=['VillanDiffusion.py']+%r; import villandiffusion_amd.dataset as D;"
            "D.synthetic_images=(lambda f: (lambda n=60000, **k: f(n=128, **k)))(D.synthetic_images);"
            "import villandiffusion_amd.model as M; M.NCSNPP_32_ARCH.update(block_out_channels=[32, 64, 64], layers_per_block=1,"
            " down_block_types=['SkipDownBlock2D', 'AttnSkipDownBlock2D', 'SkipDownBlock2D'],"
            " up_block_types=['SkipUpBlock2D', 'AttnSkipUpBlock2D', 'SkipUpBlock2D']);"
            "import VillanDiffusion as V; V.TrainingConfig.eval_sample_n=4; V.main()")
    argv = ["--mode", "train", "--dataset", "SYNTHETIC-CIFAR10", "--batch", "32", "--epoch", "1", "--poison_rate", "0.3", "--trigger", "BOX_14",
            "--target", "HAT", "--ckpt", "NCSNPP-32-DEFAULT", "--sde_type", "SDE-VE", "--psi", "0", "--fclip", "o", "-o", "--result", res,
            "--sched", "SCORE-SDE-VE-SCHED", "--infer_steps", "6", "--save_image_epochs", "1", "--save_model_epochs", "1"]
    out = subprocess.run([sys.executable, "-c", code % (argv,)], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    run = os.path.join(res, os.listdir(res)[0])
    cfg = json.load(open(os.path.join(run, "unet", "config.json")))
    assert cfg["time_embedding_type"] == "fourier" and cfg["down_block_types"][0] == "SkipDownBlock2D"
    assert json.load(open(os.path.join(run, "model_index.json")))["_class_name"] == "ScoreSdeVePipeline"
    assert os.path.exists(os.path.join(run, "samples", "0000.png")) and os.path.exists(os.path.join(run, "backdoor_samples", "final.png"))
    # sampling mode takes sde_type from the run's args.json (the reference's mode whitelist rejects --sde_type here)
    argv2 = ["--mode", "sampling", "--ckpt", run, "--sched", "SCORE-SDE-VE-SCHED", "--infer_steps", "4"]
    out = subprocess.run([sys.executable, "-c", code % (argv2,)], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert os.path.exists(os.path.join(run, "samples", "final.png"))
