"""VQ-VAE of the latent-diffusion path: structure on the CPU, HIP forward vs the oracle on the GPU."""
import pytest
import torch

from oracle.vqmodel_ref import VQModelRef
from villandiffusion_amd.vqmodel import VQModel

SMALL = dict(block_out_channels=(32, 64), down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2,
             layers_per_block=1, norm_num_groups=8, num_vq_embeddings=200, latent_channels=3, sample_size=32)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_state_dict_surface_matches_oracle_and_published_size():
    """Key names / shapes are the diffusers ones; the CompVis/ldm-celebahq-256 vqvae config has 55.3 M parameters."""
    ref, net = VQModelRef(), VQModel(device="cpu")
    sr, sn = ref.state_dict(), net.state_dict()
    assert set(sr) == set(sn)
    assert all(tuple(sr[k].shape) == tuple(sn[k].shape) for k in sr)
    assert sum(p.numel() for p in net.parameters()) == sum(p.numel() for p in ref.parameters()) == 55322782
    assert not any(p.requires_grad for p in net.parameters())          # model.py:790 vae.requires_grad_(False)
    legacy = {k.replace("to_q", "query").replace("to_k", "key").replace("to_v", "value").replace("to_out.0", "proj_attn"): v
              for k, v in sr.items()}
    net.load_state_dict(legacy)
    assert torch.equal(net.state_dict()["decoder.mid_block.attentions.0.to_k.weight"], sr["decoder.mid_block.attentions.0.to_k.weight"])


def test_oracle_quantiser_picks_the_nearest_code():
    torch.manual_seed(0)
    ref = VQModelRef(**SMALL)
    with torch.no_grad():
        ref.quantize.embedding.weight.normal_()
        z = torch.randn(2, 3, 8, 8)
        zq, idx = ref.quantize(z)
        w = ref.quantize.embedding.weight
        d = ((z.permute(0, 2, 3, 1).reshape(-1, 1, 3) - w[None]) ** 2).sum(-1)
        assert torch.equal(idx, d.argmin(1))
        assert torch.allclose(zq.permute(0, 2, 3, 1).reshape(-1, 3), w[idx])
        lat = ref.encode(torch.randn(1, 3, 32, 32)).latents
        assert lat.shape == (1, 3, 16, 16) and ref.decode(lat).sample.shape == (1, 3, 32, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,B,S", [(SMALL, 3, 32), (dict(num_vq_embeddings=8192), 1, 64)])
def test_hip_encode_quantise_decode_match_oracle(cfg, B, S):
    torch.manual_seed(1)
    ref = VQModelRef(**cfg)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
        ref.quantize.embedding.weight.normal_(0, 0.5)
    net = VQModel(**cfg)
    net.load_state_dict(ref.state_dict())
    x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        lat_ref = ref.encode(x).latents
        zq_ref, idx_ref = ref.quantize(lat_ref)
        dec_ref = ref.decode(lat_ref).sample
        dec_nq_ref = ref.decode(lat_ref, force_not_quantize=True).sample
    lat = net.encode(x.cuda()).latents
    e_enc = rel(lat, lat_ref)
    zq, idx = net.quantize_latents(lat_ref.cuda(), return_indices=True)
    same = float((idx.cpu() == idx_ref).float().mean())
    # a differing index must still be a (near-)tie: its distance is within rounding of the minimum
    w = ref.quantize.embedding.weight
    zf = lat_ref.permute(0, 2, 3, 1).reshape(-1, w.shape[1]).double()
    d_mine = ((zf - w[idx.cpu()].double()) ** 2).sum(1)
    d_best = ((zf - w[idx_ref].double()) ** 2).sum(1)
    assert float(((d_mine - d_best) / (d_best + 1e-12)).max()) < 1e-4
    e_nq = rel(net.decode(lat_ref.cuda(), force_not_quantize=True).sample, dec_nq_ref)
    e_dec = rel(net.decode(lat_ref.cuda()).sample, dec_ref)
    print(f"[parity] VQModel encode {e_enc:.2e}, decode(no quant) {e_nq:.2e}, decode {e_dec:.2e}, identical code indices {same:.4f}")
    assert e_enc < 1e-4 and e_nq < 1e-4 and same > 0.999
    assert e_dec < (1e-4 if same == 1.0 else 5e-2)


@pytest.mark.gpu
def test_ldm_pipeline_matches_oracle_and_roundtrips_on_disk(tmp_path):
    """BASELINE config #5 shape of work at a small size: UniPC-20 in the latent space, then the VQ-VAE decode
    (reference model.py:706-776 + fork LDMPipeline); also .encode(), save_pretrained/from_pretrained with vqvae/, and the
    on-the-fly latent encoding of LossFn.p_loss_by_keys(vae=...) (loss.py:942-976)."""
    import numpy as np
    from oracle import schedulers_ref as R
    from oracle.loss_ref import LossFnRef
    from oracle.unet_ref import UNet2DModelRef
    from villandiffusion_amd import schedulers as S
    from villandiffusion_amd.loss import LossFn
    from villandiffusion_amd.pipelines import DiffusionPipeline, LDMPipeline
    from villandiffusion_amd.unet import UNet2DModel

    torch.manual_seed(0)
    ucfg = dict(sample_size=16, block_out_channels=(32, 64), down_block_types=("DownBlock2D", "AttnDownBlock2D"),
                up_block_types=("AttnUpBlock2D", "UpBlock2D"), layers_per_block=1, norm_num_groups=8, downsample_padding=1,
                flip_sin_to_cos=True, freq_shift=0)
    uref, vref = UNet2DModelRef(**ucfg), VQModelRef(**SMALL)
    with torch.no_grad():
        vref.quantize.embedding.weight.normal_(0, 0.5)
    unet, vq = UNet2DModel(**ucfg), VQModel(**SMALL)
    unet.load_state_dict(uref.state_dict())
    vq.load_state_dict(vref.state_dict())
    beta = dict(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear")
    pipe = LDMPipeline(vqvae=vq, unet=unet, scheduler=S.UniPCMultistepScheduler(**beta))
    init = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    out = pipe(batch_size=2, init=init, num_inference_steps=20, output_type=None)
    with torch.no_grad():
        lat_ref = R.sample_loop(uref, R.UniPCMultistepSchedulerRef(**beta), init.clone(), 20)
        img_ref = (vref.decode(lat_ref).sample / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    err = float(np.abs(out.images - img_ref).max() / np.abs(img_ref).max())
    print(f"[parity] LDM UniPC-20 + VQ decode: image max-rel-err {err:.3e}")
    assert out.images.shape == (2, 32, 32, 3) and err <= 1e-3 and len(out.movie) == 2
    # encode contract
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        assert rel(pipe.encode(x), vref.encode(x).latents) < 1e-4
    # disk round trip in the diffusers layout
    d = str(tmp_path / "ldm")
    pipe.save_pretrained(d)
    import os
    assert os.path.exists(os.path.join(d, "vqvae", "config.json")) and os.path.exists(os.path.join(d, "unet", "config.json"))
    pipe2 = DiffusionPipeline.from_pretrained(d)
    assert isinstance(pipe2, LDMPipeline)
    assert torch.equal(pipe2.vqvae.flat_param, vq.flat_param) and torch.equal(pipe2.unet.flat_param, unet.flat_param)
    # loss with on-the-fly encoding
    sched = S.DDPMScheduler(**beta)
    t = torch.tensor([5, 900])
    eps = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(5))
    batch = {"target": x, "pixel_values": torch.zeros_like(x)}
    l = LossFn(sched, "SDE-LDM").p_loss_by_keys({k: v.cuda() for k, v in batch.items()}, unet, "target", "pixel_values", t.cuda(),
                                                vae=vq, noise=eps.cuda(), scaling_factor=0.5)
    with torch.no_grad():
        x0 = vref.encode(x).latents * 0.5
        Rr = vref.encode(torch.zeros_like(x)).latents * 0.5
        l_ref = LossFnRef(R.DDPMSchedulerRef(**beta), "SDE-LDM").p_loss(uref, x0, Rr, t, noise=eps)
    assert abs(float(l) - float(l_ref)) <= 1e-4 * abs(float(l_ref)), (float(l), float(l_ref))


@pytest.mark.gpu
def test_cli_sde_ldm_on_latent_dataset(tmp_path):
    """BASELINE config #5 flow at toy size through the drop-in CLI: a local latent-diffusion checkpoint (unet/ vqvae/
    scheduler/), the precomputed-latent dataset (--dataset CELEBA-HQ-LATENT), SDE-LDM loss with vae=None, UniPC sampling in
    latent space + VQ-VAE decode, checkpoint written back with its vqvae/ folder."""
    import json, os, subprocess, sys
    from dataset import LatentDataset
    from villandiffusion_amd import schedulers as S
    from villandiffusion_amd.pipelines import LDMPipeline
    from villandiffusion_amd.unet import UNet2DModel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck, data, res = str(tmp_path / "ldm_ckpt"), str(tmp_path / "datasets"), str(tmp_path / "exp")
    vq = VQModel(block_out_channels=(16, 32, 32), down_block_types=("DownEncoderBlock2D",) * 3, up_block_types=("UpDecoderBlock2D",) * 3,
                 layers_per_block=1, norm_num_groups=8, num_vq_embeddings=64, latent_channels=3, sample_size=256)
    unet = UNet2DModel(sample_size=64, block_out_channels=(32, 64), down_block_types=("DownBlock2D", "DownBlock2D"),
                       up_block_types=("UpBlock2D", "UpBlock2D"), layers_per_block=1, norm_num_groups=8, downsample_padding=1)
    LDMPipeline(vqvae=vq, unet=unet, scheduler=S.DDIMScheduler(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear",
                                                               clip_sample=False)).save_pretrained(ck)
    lds = LatentDataset(os.path.join(data, "celeba_hq_256_latents"))
    g = torch.Generator().manual_seed(0)
    lds.update_target_latent_by_key("CORNER", torch.randn(3, 64, 64, generator=g))
    lds.update_data_latents_by_idxs("raw", list(range(16)), torch.randn(16, 3, 64, 64, generator=g))
    lds.update_data_latents_by_idxs("BOX_14", list(range(16)), torch.randn(16, 3, 64, 64, generator=g))
    env = dict(os.environ, PYTHONPATH=root)
    code = "import sys; sys.argv=['VillanDiffusion.py']+%r; import VillanDiffusion as V; V.TrainingConfig.eval_sample_n=2; V.main()"
    argv = ["--mode", "train", "--dataset", "CELEBA-HQ-LATENT", "--dataset_load_mode", "NONE", "--sde_type", "SDE-LDM", "--ckpt", ck,
            "--batch", "8", "--epoch", "1", "--poison_rate", "0.5", "--trigger", "BOX_14", "--target", "CORNER", "--fclip", "o", "-o",
            "--result", res, "--sched", "UNIPC-SCHED", "--infer_steps", "3", "--save_image_epochs", "1", "--save_model_epochs", "1"]
    # cwd = tmp dir: the driver's dataset_path is the relative 'datasets' (reference TrainingConfig.dataset_path)
    out = subprocess.run([sys.executable, "-c", code % (argv,)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    # the result-dir name embeds --ckpt (reference :186-190); with a path ckpt that nests directories: locate the run by its index
    run = [d for d, _, files in os.walk(res) if "model_index.json" in files][0]
    idx = json.load(open(os.path.join(run, "model_index.json")))
    assert idx["_class_name"] == "LDMPipeline" and "vqvae" in idx
    for f in ("vqvae/config.json", "vqvae/diffusion_pytorch_model.safetensors", "unet/config.json", "samples/0000.png", "backdoor_samples/final.png"):
        assert os.path.exists(os.path.join(run, f)), f
