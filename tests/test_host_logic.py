"""Host side of the product path (no GPU): coefficient tables against the reference-generated fixtures, Backdoor /
DatasetLoader partition logic, the model/scheduler factory surface and errors, the CLI's config overlay."""
import json
import os

import numpy as np
import pytest
import torch

from villandiffusion_amd import dataset as D
from villandiffusion_amd import loss as PL
from villandiffusion_amd.model import DiffuserModelSched as DMS
from villandiffusion_amd.schedulers import DDPMScheduler
from villandiffusion_amd.unet import UNet2DModel

G = os.path.join(os.path.dirname(__file__), "golden")
TAB = np.load(os.path.join(G, "loss_tables.npz"))
BOX = np.load(os.path.join(G, "backdoor_boxes.npz"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _VE:
    def __init__(self):
        ts = torch.linspace(1, 1e-5, 2000)
        self.sigmas = torch.tensor([0.01 * (380.0 / 0.01) ** t for t in ts])


def test_product_tables_bit_exact_vs_reference_fixtures():
    for name, kw in (("vp_linear", {}), ("ldm_scaled_linear", dict(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear"))):
        s = DDPMScheduler(**kw)
        np.testing.assert_array_equal(PL.get_hs_vp(s.alphas, s.alphas_cumprod).numpy(), TAB[f"{name}/hs"])
        for psi in (0.0, 0.5, 1.0):
            for solver in ("sde", "ode"):
                step, coef = PL.LossFn(s, "SDE-VP", psi=psi, solver_type=solver).get_R_step_coef()
                np.testing.assert_array_equal(step.numpy(), TAB[f"{name}/psi{psi}/{solver}/step"])
                np.testing.assert_array_equal(coef.numpy(), TAB[f"{name}/psi{psi}/{solver}/coef"])
    for solver in ("sde", "ode"):
        step, coef = PL.LossFn(_VE(), "SDE-VE", psi=0, solver_type=solver).get_R_step_coef()
        np.testing.assert_array_equal(step.numpy(), TAB[f"ve/psi0/{solver}/step"])
        np.testing.assert_array_equal(coef.numpy(), TAB[f"ve/psi0/{solver}/coef"])
    with pytest.raises(NotImplementedError):
        PL.LossFn(_VE(), "SDE-VE", psi=1).get_R_step_coef()
    with pytest.raises(NotImplementedError):
        PL.LossFn(DDPMScheduler(), "SDE-XX")
    with pytest.raises(NotImplementedError):
        PL.LossFn(DDPMScheduler(), "SDE-VP", solver_type="xyz").get_R_step_coef()
    assert PL.LossFn(DDPMScheduler(), "SDE-VP").p_loss(None, torch.zeros(0, 3, 4, 4), torch.zeros(0, 3, 4, 4), torch.zeros(0)) == 0


def test_product_backdoor_bit_exact_vs_reference_fixtures():
    bd = D.Backdoor(root=ROOT)
    n = 0
    for key in BOX.files:
        if not key.endswith("/trigger"):
            continue
        S_, v_, tt, _ = key.split("/")
        vmin, vmax = (float(z) for z in v_[1:].split("_"))
        trig = bd.get_trigger(tt, 3, int(S_[1:]), vmin, vmax)
        np.testing.assert_array_equal(trig.numpy(), BOX[key])
        for tg in ("CORNER", "NOSHIFT", "SHIFT"):
            k2 = key[:-7] + f"target_{tg}"
            if k2 in BOX.files:
                np.testing.assert_array_equal(bd.get_target(tg, trig, vmin=vmin, vmax=vmax).numpy(), BOX[k2])
                n += 1
    assert n > 0
    from oracle import backdoor_ref as BR
    for tt in ("STOP_SIGN_14", "GLASSES"):
        assert torch.equal(bd.get_trigger(tt, 3, 32), BR.get_trigger(ROOT, tt, 3, 32))
    trig = bd.get_trigger("BOX_14", 3, 32)
    for tg in ("HAT", "BWHAT", "CAT"):
        assert torch.equal(bd.get_target(tg, trig), BR.get_target(ROOT, tg, trig))
    with pytest.raises(ValueError):
        bd.get_trigger("NOPE", 3, 32)
    assert D.Backdoor.TRIGGER_SM_BOX_MED == "BOX_14" and D.Backdoor.TARGET_FEDORA_HAT == "HAT" and D.Backdoor.TARGET_HAT == "BWHAT"


def _dsl(n=1000, **kw):
    imgs = D.synthetic_images(n=n)
    return D.DatasetLoader("X", root=ROOT, images=imgs, labels=np.arange(n) % 10, device="cpu", **kw)


def test_dataset_partition_modes():
    d = _dsl().set_poison("BOX_14", "CORNER", poison_rate=0.1).prepare_dataset("FIXED")
    assert len(d) == 1000 and int((d._flags & 1).sum()) == 100 and len(set(d._index.tolist())) == 1000
    assert d.num_batch == 2 and d.image_size == 32 and d.channel == 3
    d = _dsl().set_poison("BOX_14", "CORNER", clean_rate=0.5, poison_rate=0.2).prepare_dataset("FLEX")
    assert len(d) == 700 and int((d._flags & 1).sum()) == 200
    d = _dsl().set_poison("BOX_14", "CORNER", poison_rate=1.0).prepare_dataset("FLEX")
    assert len(d) == 2000 and int((d._flags & 1).sum()) == 1000
    d = _dsl().set_poison("BOX_14", "CORNER", poison_rate=2.5, ext_poison_rate=0.1).prepare_dataset("EXTEND", ext_R_trigger_only=True)
    assert len(d) == 900 + 100 + 2500 and int(((d._flags & 4) != 0).sum()) == 100
    d = _dsl().set_poison("BOX_14", "CORNER", poison_rate=0.3).prepare_dataset("NONE")
    assert len(d) == 1000 and int(d._flags.sum()) == 0
    with pytest.raises(ValueError):
        _dsl().set_poison("BOX_14", "CORNER", poison_rate=1.5).prepare_dataset("FIXED")
    with pytest.raises(NotImplementedError):
        _dsl().set_poison("BOX_14", "CORNER").prepare_dataset("WHAT")
    with pytest.raises(ValueError):
        D.DatasetLoader("X", images=D.synthetic_images(8), device="cpu").set_poison("BOX_14", "CORNER")
    d = _dsl(label=3).set_poison("BOX_14", "CORNER", poison_rate=0.5).prepare_dataset("FIXED")
    assert len(d) == 100 and set((d._labels[d._index]).tolist()) == {3}
    x = torch.rand(2, 3, 32, 32) * 2 - 1
    p = d.get_poisoned(x)
    assert bool((p[:, :, 16:30, 16:30] == 0).all()) and bool((p[:, :, :16] == x[:, :, :16]).all())
    m = d.get_inpainted_by_type(x, "INPAINT_BOX")
    assert bool((m[:, :, 11:21, 11:21] == x.min()).all())


def test_unet_structure_and_factory_errors():
    net = UNet2DModel(device="cpu")
    assert sum(p.numel() for p in net.parameters()) == 35746307
    from oracle.unet_ref import UNet2DModelRef
    ref = UNet2DModelRef()
    assert list(sorted(net.state_dict())) == list(sorted(ref.state_dict()))
    assert all(p.grad is not None and p.grad.shape == p.shape for p in net.parameters())
    assert net.Wt_all.shape == (4992, 512) and net.flat_param.numel() == net.flat_grad.numel()
    with pytest.raises(NotImplementedError):
        UNet2DModel(device="cpu", time_embedding_type="fourier")
    with pytest.raises(NotImplementedError):
        UNet2DModel(device="cpu", down_block_types=("SkipDownBlock2D",) * 4)
    with pytest.raises(ValueError):
        DMS.get_model_sched(ckpt=DMS.DDPM_32_DEFAULT)
    with pytest.raises(NotImplementedError):
        DMS.get_model_sched(image_size=32, channels=3, ckpt=DMS.DDPM_32_DEFAULT, noise_sched_type="NOPE-SCHED")
    with pytest.raises(NotImplementedError):
        DMS.get_model_sched(image_size=32, channels=3, ckpt=DMS.DDPM_32_DEFAULT, sde_type="SDE-XX")
    with pytest.raises(FileNotFoundError):
        DMS.get_pretrained(DMS.DDPM_CIFAR10_32)
    assert DMS.HUB_IDS[DMS.DDPM_CIFAR10_32] == "google/ddpm-cifar10-32" and DMS.SDE_VP == "SDE-VP"
    assert DMS.DPM_SOLVER_PP_O2_SCHED == "DPM_SOLVER_PP_O2-SCHED" and DMS.UNIPC_SCHED == "UNIPC-SCHED"


def test_pretrained_directory_roundtrip(tmp_path):
    from villandiffusion_amd.pipelines import DDPMPipeline
    net = UNet2DModel(device="cpu")
    net.reset_parameters(seed=3)
    pipe = DDPMPipeline(net, DDPMScheduler(clip_sample=False))
    pipe.save_pretrained(str(tmp_path / "ck"))
    idx = json.load(open(tmp_path / "ck" / "model_index.json"))
    assert idx["unet"] == ["diffusers", "UNet2DModel"] and idx["scheduler"] == ["diffusers", "DDPMScheduler"]
    assert os.path.exists(tmp_path / "ck" / "unet" / "diffusion_pytorch_model.safetensors")
    pipe2 = DDPMPipeline.from_pretrained(str(tmp_path / "ck"))
    assert torch.equal(pipe2.unet.flat_param.cpu(), net.flat_param.cpu()) and pipe2.scheduler.config.clip_sample is False
    m, vae, sched, gp = DMS.get_pretrained(str(tmp_path / "ck"), clip_sample=True, noise_sched_type=DMS.DDIM_SCHED)
    assert vae is None and sched.config.clip_sample is True and type(sched).__name__ == "DDIMScheduler"
    assert type(gp(None, m, None, sched)).__name__ == "DDIMPipeline"


def test_cli_config_overlay(tmp_path):
    import VillanDiffusion as V
    res = str(tmp_path / "exp")
    argv = ["--project", "default", "--mode", "train", "--dataset", "CIFAR10", "--batch", "4", "--epoch", "1", "--poison_rate", "0.1",
            "--trigger", "BOX_14", "--target", "HAT", "--ckpt", "DDPM-CIFAR10-32", "--fclip", "o", "-o", "--gpu", "0", "--result", res,
            "--sched", "DDPM-SCHED"]
    cfg = V.setup(V.parse_args(argv))
    assert cfg.gradient_accumulation_steps == 32 and cfg.learning_rate == 2e-4 and cfg.clip is False
    # `psi1`: DEFAULT_PSI is the int 1 in the reference (":45"), so the directory name of BASELINE config #1 has no ".0"
    assert os.path.basename(cfg.output_dir) == ("res_DDPM-CIFAR10-32_CIFAR10_ep1_sde_c1.0_p0.1_epr0.0_BOX_14-HAT_psi1_lr0.0002_vp1.0_ve1.0")
    assert json.load(open(os.path.join(cfg.output_dir, "args.json")))["trigger"] == "BOX_14"
    assert os.path.exists(os.path.join(cfg.output_dir, "config.json"))
    with pytest.raises(ValueError):                    # existing directory without -o
        V.setup(V.parse_args([a for a in argv if a != "-o"]))
    with pytest.raises(ValueError):                    # batch must divide 128
        V.setup(V.parse_args([("48" if a == "4" else a) for a in argv]))
    with pytest.raises(NotImplementedError):           # option outside the sampling whitelist
        V.setup(V.parse_args(["--mode", "sampling", "--ckpt", cfg.output_dir, "--epoch", "3"]))
    c2 = V.setup(V.parse_args(["--mode", "sampling", "--ckpt", cfg.output_dir, "--sched", "DDIM-SCHED", "--infer_steps", "50", "--fclip", "w"]))
    assert c2.trigger == "BOX_14" and c2.sched == "DDIM-SCHED" and c2.infer_steps == 50 and c2.clip is True
    assert os.path.exists(os.path.join(cfg.output_dir, "sampling.json"))
    c3 = V.setup(V.parse_args(argv[:-2] + ["--ckpt", "DDPM-32-DEFAULT", "--dataset", "CELEBA-HQ"][0:0] + ["--sched", "DDPM-SCHED", "-o"]))
    assert c3.batch == 4


def test_metrics_known_answers():
    from villandiffusion_amd.metrics import mse_batch, mse_thres_batch, ssim_batch
    a = torch.rand(3, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    assert ssim_batch(a, a) == pytest.approx(1.0, abs=1e-6) and mse_batch(a, a) == 0.0
    b = (a + 0.1).clamp(0, 1)
    assert 0.5 < ssim_batch(a, b) < 1.0 and mse_batch(a, b) == pytest.approx(float(((a - b) ** 2).mean()), rel=1e-6)
    assert ssim_batch(a, 1 - a) < 0.1
    assert mse_thres_batch(a, a, 1e-3) == 1.0 and mse_thres_batch(a, 1 - a, 1e-3) == 0.0
    # constant images: mu terms only -> (2ab+c1)/(a^2+b^2+c1)
    x, y = torch.full((1, 1, 16, 16), 0.2), torch.full((1, 1, 16, 16), 0.6)
    c1 = 0.01 ** 2
    assert ssim_batch(x, y) == pytest.approx((2 * 0.2 * 0.6 + c1) / (0.04 + 0.36 + c1), rel=1e-4)


def test_latent_dataset_disk_format_and_poison_by_index(tmp_path):
    """dataset.py:1037-1371: `<root>/target..pt` dict, `<root>/<type>/<idx>..pt` (double dot), item i poisoned iff
    i < int(len * poison_rate); clean items carry zeros as the poison latent and the raw latent as target."""
    import os
    from types import SimpleNamespace
    from dataset import LatentDataset

    class FakeVae:                         # stand-in with the VQModel call surface
        device = torch.device("cpu")

        def encode(self, x):
            return SimpleNamespace(latents=x[:, :, ::2, ::2] * 2.0)

        def decode(self, z):
            return SimpleNamespace(sample=z.repeat_interleave(2, 2).repeat_interleave(2, 3) / 2.0)

    root = str(tmp_path / "lat")
    ds = LatentDataset(root).set_vae(FakeVae())
    imgs = torch.randn(10, 3, 8, 8, generator=torch.Generator().manual_seed(0))
    pois = imgs + 1.0
    tgt = torch.randn(3, 8, 8, generator=torch.Generator().manual_seed(1))
    ds.update_target_by_key("CAT", tgt)
    ds.update_data_by_idxs(LatentDataset.RAW_LATENTS_FILE_NAME, list(range(10)), imgs)
    ds.update_data_by_idxs("GLASSES", list(range(10)), pois)
    assert os.path.exists(os.path.join(root, "target..pt")) and os.path.exists(os.path.join(root, "raw", "7..pt"))
    assert os.path.exists(os.path.join(root, "GLASSES", "0..pt"))
    assert LatentDataset.add_ext("x") == "x..pt"
    ds.set_poison(target_key="CAT", poison_key="GLASSES", raw="raw", poison_rate=0.3).set_use_names("target", "pixel_values", "image")
    assert len(ds) == 10
    enc = lambda x: x[:, ::2, ::2] * 2.0
    for i in range(10):
        it = ds[i]
        assert torch.equal(it["image"], enc(imgs[i]))
        if i < 3:
            assert torch.equal(it["target"], enc(tgt)) and torch.equal(it["pixel_values"], enc(pois[i]))
        else:
            assert torch.equal(it["target"], enc(imgs[i])) and float(it["pixel_values"].abs().max()) == 0.0
    assert torch.equal(ds[13]["image"], ds[3]["image"])                  # index wraps (dataset.py:1345)
    assert torch.allclose(ds.get_target_by_key("CAT")[:, ::2, ::2], tgt[:, ::2, ::2])
    with pytest.raises(ValueError):
        LatentDataset(str(tmp_path / "x")).update_target_by_key("k", tgt)


def test_lossfn_captures_the_schedule_at_construction():
    """reference loss.py:829-834 reads noise_sched.sigmas / alphas in LossFn.__init__; a VE pipeline's set_sigmas(n) later swaps
    noise_sched.sigmas for the n-step inference table (VillanDiffusion.py samples before training) -- the 2000-entry training
    tables must not follow it (regression: indexing a 6-entry table with t < 2000 faulted on the device)."""
    from villandiffusion_amd import schedulers as S
    from villandiffusion_amd.loss import LossFn
    s = S.ScoreSdeVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0)
    lf = LossFn(s, "SDE-VE", psi=0)
    step0, coef0 = lf.get_R_step_coef()
    s.set_timesteps(6)
    s.set_sigmas(6)
    assert len(s.sigmas) == 6
    step1, coef1 = lf.get_R_step_coef()
    assert len(step1) == 2000 and torch.equal(step0, step1) and torch.equal(coef0, coef1)


def test_dataset_loader_on_latent_dataset(tmp_path):
    """--dataset CELEBA-HQ-LATENT (BASELINE config #5): DatasetLoader wraps LatentDataset, poisons by index, batches are index
    gathers of the resident latents; trigger / target stay image-space 256x256 tensors for the pipeline to encode."""
    import os
    from dataset import DatasetLoader, LatentDataset
    root = str(tmp_path)
    lds = LatentDataset(os.path.join(root, "celeba_hq_256_latents"))
    g = torch.Generator().manual_seed(0)
    raw = torch.randn(10, 3, 8, 8, generator=g)
    poi = torch.randn(10, 3, 8, 8, generator=g)
    tgt = torch.randn(3, 8, 8, generator=g)
    lds.update_target_latent_by_key("CORNER", tgt)
    lds.update_data_latents_by_idxs("raw", list(range(10)), raw)
    lds.update_data_latents_by_idxs("BOX_14", list(range(10)), poi)
    dsl = DatasetLoader("CELEBA-HQ-LATENT", root=root, batch_size=4, device="cpu").set_poison("BOX_14", "CORNER", poison_rate=0.5) \
        .prepare_dataset(mode="NONE")
    assert len(dsl) == 10 and dsl.image_size == 256 and tuple(dsl.trigger.shape) == (3, 256, 256)
    b = dsl.make_batch(torch.tensor([0, 4, 5, 9]))
    assert b["is_clean"].tolist() == [False, False, True, True]
    assert torch.equal(b["image"], raw[[0, 4, 5, 9]])
    assert torch.equal(b["pixel_values"][:2], poi[[0, 4]]) and float(b["pixel_values"][2:].abs().max()) == 0.0
    assert torch.equal(b["target"][0], tgt) and torch.equal(b["target"][1], tgt) and torch.equal(b["target"][2:], raw[[5, 9]])
    assert sum(x["image"].shape[0] for x in dsl.get_dataloader(full=False)) == 10


def test_frechet_distance_known_answers():
    """FID's closed form on Gaussians (fid_score.py:150-203): 0 for identical statistics, |dmu|^2 for a mean shift,
    Tr(S1 + S2 - 2 sqrt(S1 S2)) = sum (sqrt(a_i) - sqrt(b_i))^2 for commuting (diagonal) covariances."""
    import numpy as np
    from villandiffusion_amd.metrics import activation_statistics, frechet_distance
    rng = np.random.default_rng(0)
    act = rng.normal(size=(500, 6))
    mu, sig = activation_statistics(act)
    assert abs(frechet_distance(mu, sig, mu, sig)) < 1e-8
    assert abs(frechet_distance(mu + 2.0, sig, mu, sig) - 6 * 4.0) < 1e-6
    a, b = np.array([1.0, 4.0, 9.0]), np.array([4.0, 1.0, 16.0])
    want = float(((np.sqrt(a) - np.sqrt(b)) ** 2).sum())
    assert abs(frechet_distance(np.zeros(3), np.diag(a), np.zeros(3), np.diag(b)) - want) < 1e-8


def _write_idx(path, arr, gz=False):
    import gzip
    import struct
    arr = np.asarray(arr, dtype=np.uint8)
    head = struct.pack(">I", 2051 if arr.ndim == 3 else 2049) + b"".join(struct.pack(">I", d) for d in arr.shape)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with (gzip.open(path + ".gz", "wb") if gz else open(path, "wb")) as f:
        f.write(head + arr.tobytes())


def test_image_folder_dataset_with_resize_and_channel_conversion(tmp_path):
    """D4 (dataset.py:118-122, 160-176): CELEBA-HQ / CELEBA from a local image folder; RGB conversion + Resize([S, S]) (PIL bilinear)
    applied once at load time.  (torchvision is absent here, so the transform's PIL restatement is the comparison.)"""
    from PIL import Image
    rng = np.random.default_rng(5)
    root = tmp_path / "datasets"
    d = root / "celeba_hq_256" / "sub"
    d.mkdir(parents=True)
    srcs = []
    for i, (h, w, mode) in enumerate([(300, 200, "RGB"), (64, 64, "RGB"), (100, 130, "L"), (90, 90, "RGBA")]):
        a = rng.integers(0, 256, size=(h, w) + ({"RGB": (3,), "L": (), "RGBA": (4,)}[mode]), dtype=np.uint8)
        im = Image.fromarray(a, mode=mode)
        im.save(d / f"{i:03d}.png")
        srcs.append(im)
    (d / "notes.txt").write_text("not an image")
    dsl = D.DatasetLoader(name="CELEBA-HQ", root=str(root), image_size=64, device="cpu")
    assert dsl._images.shape == (4, 64, 64, 3) and dsl._images.dtype == np.uint8 and dsl.image_size == 64 and dsl.channel == 3
    for i, im in enumerate(srcs):
        want = np.asarray(im.convert("RGB").resize((64, 64), Image.BILINEAR))
        assert np.array_equal(dsl._images[i], want), i
    assert np.array_equal(dsl._images[1], np.asarray(srcs[1]))              # already at the training size: untouched
    assert D.DatasetLoader(name="CELEBA-HQ", root=str(root), device="cpu")._images.shape == (4, 256, 256, 3)       # dataset.py:143-147 defaults
    g1 = D.DatasetLoader(name="CELEBA-HQ", root=str(root), image_size=32, channel=1, device="cpu")
    assert np.array_equal(g1._images[0, ..., 0], np.asarray(srcs[0].convert("L").resize((32, 32), Image.BILINEAR)))
    # the same set stored as an array file, and the 64-pixel CelebA default
    np.savez(root / "celeba.npz", images=dsl._images)
    c = D.DatasetLoader(name="CELEBA", root=str(root), device="cpu")
    assert c.image_size == 64 and np.array_equal(c._images, dsl._images)
    c.set_poison("STOP_SIGN_14", "CAT", poison_rate=0.5).prepare_dataset(mode="FIXED")
    assert len(c) == 4 and int((c._flags & 1).sum()) == 2 and c.trigger.shape == (3, 64, 64)
    with pytest.raises(FileNotFoundError):
        D.DatasetLoader(name="CELEBA-HQ", root=str(tmp_path / "nowhere"), device="cpu")
    for name in ("LSUN-CHURCH", "LSUN-BEDROOM", "CELEBA-HQ-LATENT_PR05", "IMAGENET"):     # the reference has no loader for these either
        with pytest.raises(NotImplementedError):
            D.DatasetLoader(name=name, root=str(root), device="cpu")


def test_mnist_from_idx_files_and_idx_derived_triggers(tmp_path):
    """MNIST = train + test from local idx files, 28x28 'L' -> Resize([32, 32]) -> 1 channel (dataset.py:111-114, 131-149); the
    FASHION / MNIST triggers and the SHOE target read torchvision's raw-file layout (dataset.py:791-812, 947-951)."""
    from PIL import Image
    rng = np.random.default_rng(9)
    root = str(tmp_path / "datasets")
    tr, te = rng.integers(0, 256, size=(150, 28, 28), dtype=np.uint8), rng.integers(0, 256, size=(7, 28, 28), dtype=np.uint8)
    _write_idx(os.path.join(root, "MNIST", "raw", "train-images-idx3-ubyte"), tr)
    _write_idx(os.path.join(root, "MNIST", "raw", "t10k-images-idx3-ubyte"), te, gz=True)
    _write_idx(os.path.join(root, "MNIST", "raw", "train-labels-idx1-ubyte"), np.arange(150) % 10)
    _write_idx(os.path.join(root, "MNIST", "raw", "t10k-labels-idx1-ubyte"), np.arange(7) % 10, gz=True)
    _write_idx(os.path.join(root, "FashionMNIST", "raw", "train-images-idx3-ubyte"), tr[::-1].copy(), gz=True)
    dsl = D.DatasetLoader(name="MNIST", root=root, device="cpu")
    assert dsl._images.shape == (157, 32, 32, 1) and dsl.channel == 1 and dsl.image_size == 32
    assert np.array_equal(dsl._images[150, ..., 0], np.asarray(Image.fromarray(te[0], "L").resize((32, 32), Image.BILINEAR)))
    assert np.array_equal(dsl._labels[:12], np.arange(12) % 10) and len(dsl._labels) == 157
    only3 = D.DatasetLoader(name="MNIST", root=root, label=3, device="cpu").set_poison("BOX_14", "CORNER", poison_rate=0.0).prepare_dataset()
    assert len(only3) == 16                                               # 15 in train + 1 in test carry label 3
    bd = D.Backdoor(root=root)

    def ref_item(arr, ch, size):
        im = Image.fromarray(arr, "L").convert("RGB" if ch == 3 else "L").resize((size, size), Image.BILINEAR)
        t = torch.from_numpy(np.asarray(im).copy())
        t = (t.permute(2, 0, 1) if t.dim() == 3 else t[None]).float() / 255.0
        return D.normalize(t, 0.0, 1.0, -1.0, 1.0)

    for typ, (arr, dx, dy) in {"FASHION": (tr[::-1][0], 0, 2), "FASHION_EZ": (tr[::-1][144], 0, 4), "MNIST": (tr[3], 10, 3),
                               "MNIST_EZ": (tr[6], 10, 3)}.items():
        for ch in (1, 3):
            want = ref_item(arr, ch, 32)
            want[want <= -0.4] = -1.0                                      # __bg2black
            want = torch.roll(want, shifts=(dy, dx), dims=(1, 2))
            assert torch.equal(bd.get_trigger(typ, ch, 32), want), (typ, ch)
    shoe = ref_item(tr[::-1][0], 3, 32)
    shoe[shoe <= -0.4] = -0.4                                              # __bg2grey
    assert torch.equal(bd.get_target("SHOE", bd.get_trigger("BOX_14", 3, 32)), shoe)
    with pytest.raises(FileNotFoundError):
        D.Backdoor(root=str(tmp_path / "nowhere")).get_target("SHOE", bd.get_trigger("BOX_14", 3, 32))


def test_driver_preflight_rejects_unreadable_dataset_before_side_effects(tmp_path, monkeypatch):
    """A --dataset with no local copy must fail before the run directory / args.json exist (the reference would download it)."""
    import VillanDiffusion as V
    monkeypatch.chdir(tmp_path)
    res = tmp_path / "exp"
    args = V.parse_args(["--mode", "train", "--dataset", "CELEBA-HQ", "--batch", "64", "--result", str(res), "-o"])
    with pytest.raises(FileNotFoundError):
        V.setup(args, preflight=True)
    assert not res.exists()
    args = V.parse_args(["--mode", "train", "--dataset", "SYNTHETIC-CIFAR10", "--batch", "64", "--result", str(res), "-o"])
    cfg = V.setup(args, preflight=True)
    assert os.path.exists(os.path.join(cfg.output_dir, "args.json"))
    # every rank but 0 computes the same config without touching the file system
    monkeypatch.setenv("RANK", "1")
    args = V.parse_args(["--mode", "train", "--dataset", "SYNTHETIC-CIFAR10", "--batch", "64", "--result", str(tmp_path / "exp2")])
    cfg1 = V.setup(args, preflight=True)
    assert cfg1.output_dir.startswith(str(tmp_path / "exp2")) and not (tmp_path / "exp2").exists()
    args = V.parse_args(["--mode", "train", "--dataset", "SYNTHETIC-CIFAR10", "--batch", "64", "--result", str(res)])     # exists, no -o:
    assert V.setup(args).output_dir == cfg.output_dir                                                                     # only rank 0 raises
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(ValueError):
        V.setup(args)


def test_scheduler_variance_and_prediction_types(tmp_path):
    """`variance_type` / `prediction_type` of a loaded scheduler_config.json are honoured or rejected, never ignored."""
    from villandiffusion_amd import schedulers as S
    from villandiffusion_amd.pipelines import DDPMPipeline
    for vt in ("fixed_large_log", "learned", "learned_range"):
        with pytest.raises(NotImplementedError):
            S.DDPMScheduler(variance_type=vt)
    with pytest.raises(NotImplementedError):
        S.DDPMScheduler(prediction_type="v_prediction")
    with pytest.raises(NotImplementedError):
        S.DDIMScheduler(prediction_type="sample")
    a = S.DDPMScheduler(variance_type="fixed_large")
    ac = a.alphas_cumprod
    assert a._noise_scale(ac[500], ac[499], 1 - ac[500] / ac[499]) == float((1 - ac[500] / ac[499]) ** 0.5)
    # the from-scratch ids take the scheduler of google/ddpm-cifar10-32 when --sched is unset (model.py:654, 813)
    _, _, sched, _ = DMS._get_model_sched_vp(DMS.HUB_IDS[DMS.DDPM_CIFAR10_32], None, noise_sched_type=None, build_model=False)
    assert sched.config.variance_type == "fixed_large" and sched.config.clip_sample is False      # CLIP_SAMPLE_DEFAULT (model.py:601,657-659)
    _, _, sched, _ = DMS._get_model_sched_vp(DMS.HUB_IDS[DMS.DDPM_CIFAR10_32], False, noise_sched_type=DMS.DDPM_SCHED, build_model=False)
    assert sched.config.variance_type == "fixed_small" and sched.config.clip_sample is False          # model.py:615
    # a diffusers directory whose scheduler asks for something unsupported fails loudly at load time
    net = UNet2DModel(block_out_channels=(32, 64), down_block_types=("DownBlock2D", "AttnDownBlock2D"),
                      up_block_types=("AttnUpBlock2D", "UpBlock2D"), layers_per_block=1, sample_size=8, device="cpu")
    DDPMPipeline(net, S.DDPMScheduler(variance_type="fixed_large")).save_pretrained(str(tmp_path / "ck"))
    cfgp = tmp_path / "ck" / "scheduler" / "scheduler_config.json"
    cfg = json.load(open(cfgp))
    assert cfg["variance_type"] == "fixed_large"
    assert DDPMPipeline.from_pretrained(str(tmp_path / "ck")).scheduler.config.variance_type == "fixed_large"
    cfg["prediction_type"] = "v_prediction"
    json.dump(cfg, open(cfgp, "w"))
    with pytest.raises(NotImplementedError):
        DDPMPipeline.from_pretrained(str(tmp_path / "ck"))


def test_metrics_agree_with_the_oracle_restatement():
    """villandiffusion_amd/metrics.py (one grouped convolution) vs oracle/metrics_ref.py (numpy loops): MSE / SSIM of the measure pipeline."""
    from oracle.metrics_ref import mse_ref, ssim_ref
    from villandiffusion_amd.metrics import mse_batch, ssim_batch
    rng = np.random.default_rng(3)
    for shape in ((5, 3, 32, 32), (2, 1, 40, 28), (3, 3, 16, 16)):
        a = rng.random(shape).astype(np.float32)
        b = np.clip(a + 0.2 * rng.standard_normal(shape).astype(np.float32), 0, 1)
        assert abs(mse_batch(torch.from_numpy(a), torch.from_numpy(b)) - mse_ref(a, b)) < 1e-7
        assert abs(ssim_batch(torch.from_numpy(a), torch.from_numpy(b)) - ssim_ref(a, b)) < 1e-5
    assert abs(ssim_ref(a, a) - 1.0) < 1e-12


def test_gpu_flag_plan():
    """--gpu (reference VillanDiffusion.py:240-245, 440): one index selects the device, a list starts one rank per listed GPU, an already
    restricted visible set is indexed into, a launcher's assignment is left alone."""
    import VillanDiffusion as V
    assert V.gpu_plan("5", {}) == {"action": "single", "visible": "5", "n": 1}
    assert V.gpu_plan("0,1,2,3", {}) == {"action": "spawn", "visible": "0,1,2,3", "n": 4}
    assert V.gpu_plan("1", {"HIP_VISIBLE_DEVICES": "4,5"}) == {"action": "single", "visible": "5", "n": 1}
    assert V.gpu_plan("1,0", {"CUDA_VISIBLE_DEVICES": "6,7"}) == {"action": "spawn", "visible": "7,6", "n": 2}
    assert V.gpu_plan("0,1", {"WORLD_SIZE": "2", "HIP_VISIBLE_DEVICES": "0,1"})["action"] == "rank"
    for bad in ("", "a", "0,0", "1,x"):
        with pytest.raises(ValueError):
            V.gpu_plan(bad, {})
    with pytest.raises(ValueError):
        V.gpu_plan("2", {"HIP_VISIBLE_DEVICES": "0,1"})
    # the flag of a resumed / sampled run comes from the run directory when the command line omits it
    a = V.parse_args(["--mode", "train", "--gpu", "3"])
    assert V.effective_gpu(a) == "3"
    assert V.effective_gpu(V.parse_args(["--mode", "train"])) == "0"


def test_gpu_flag_from_run_directory(tmp_path):
    import VillanDiffusion as V
    (tmp_path / "run").mkdir()
    with open(tmp_path / "run" / "args.json", "w") as f:
        json.dump({"gpu": "2,3", "trigger": "BOX_14"}, f)
    assert V.effective_gpu(V.parse_args(["--mode", "sampling", "--ckpt", str(tmp_path / "run")])) == "2,3"
    assert V.effective_gpu(V.parse_args(["--mode", "sampling", "--ckpt", str(tmp_path / "run"), "--gpu", "1"])) == "1"


def test_gpu_list_spawns_one_rank_per_gpu():
    """`--gpu "0,1"` without a launcher: the parent (which never touches a GPU) starts two rank processes through torch.distributed.run and
    exits with its code; here on gloo with VILLAN_RENDEZVOUS_ONLY=1 (no compute).  The ranks see the composed visible set."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VILLAN_RENDEZVOUS_ONLY="1", VILLAN_DIST_BACKEND="gloo", HIP_VISIBLE_DEVICES="3,4,5", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("CUDA_VISIBLE_DEVICES", None)
    r = subprocess.run([sys.executable, os.path.join(root, "VillanDiffusion.py"), "--mode", "train", "--gpu", "2,0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out == {"rendezvous_only": True, "world_size": 2, "ranks_counted": 2, "visible": "5,3"}
    # one index: same process, device selected through the environment before torch starts
    r1 = subprocess.run([sys.executable, os.path.join(root, "VillanDiffusion.py"), "--mode", "train", "--gpu", "1"], env=env,
                        capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1]) == \
        {"rendezvous_only": True, "world_size": 1, "ranks_counted": 1, "visible": "4"}


def test_rank_sharded_measure_chunks_own_disjoint_philox_ranges(monkeypatch):
    """Advisor r4: two consecutive chunked sampling calls (clean set, then backdoor set -- what `measure()` does) on a world of 2 must give every
    (call, chunk) its own Philox offset range whatever rank draws it, including a rank that owns only the short last chunk and a rank with no
    chunk at all: the stride and the advance of the scheduler's offset come from the WHOLE job's chunk list, not from the rank's own."""
    from types import SimpleNamespace
    from villandiffusion_amd import pipelines as P
    from villandiffusion_amd import sampling_io as SIO

    steps, calls = 7, []

    def fake_call(self, batch_size=None, init=None, num_inference_steps=None, return_tensor=False, **kw):
        calls.append((self.rank, self.scheduler._rng_offset, init.numel()))
        return init

    monkeypatch.setattr(P.DiffusionPipeline, "__call__", fake_call)
    monkeypatch.setattr(SIO, "save_imgs", lambda *a, **k: None)
    monkeypatch.setattr(P, "_post", lambda x: x)

    def job(n_images, batch, world):
        calls.clear()
        ranges, ends = [], []
        for rank in range(world):
            pipe = object.__new__(P.DiffusionPipeline)
            pipe.rank, pipe.unet = rank, None                    # (no UNet2DModel: the chunks take the sequential walk)
            pipe.scheduler = SimpleNamespace(device_rng_seed=5, _rng_offset=0, set_timesteps=lambda n: None, timesteps=list(range(steps)))
            pipe.default_steps = steps
            for _ in range(2):                                   # clean set, then backdoor set
                init = torch.zeros(n_images, 3, 4, 4)
                SIO.batch_sampling_save(n_images, pipe, "unused", init=init, max_batch_n=batch, num_inference_steps=steps, rank=rank, world=world)
            ends.append(pipe.scheduler._rng_offset)
        per_elem = 2 * steps                                     # draws per element a chunk may make (chunk_rng_offset's stride)
        for _, off, numel in calls:
            ranges.append((off, off + per_elem * ((numel + 3) // 4)))
        return sorted(ranges), ends

    for n_images, batch, world in ((384, 256, 2), (2048, 256, 2), (256, 256, 2), (1024, 128, 4)):
        ranges, ends = job(n_images, batch, world)
        n_chunks = -(-n_images // batch)
        assert len(ranges) == 2 * n_chunks
        for (a0, a1), (b0, b1) in zip(ranges, ranges[1:]):
            assert a1 <= b0, (n_images, batch, world, ranges)     # no two (call, chunk) pairs share an offset
        assert len(set(ends)) == 1                                # every rank (also one without chunks) ends at the same offset
        ranges1, ends1 = job(n_images, batch, 1)
        assert ranges1 == ranges and ends1[0] == ends[0]          # ... and the layout does not depend on the world size
