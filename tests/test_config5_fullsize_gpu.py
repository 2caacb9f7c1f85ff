"""BASELINE config #5 (LDM-CELEBA-HQ-256, SDE-LDM) at its REAL size on the GPU, held to the CPU oracle:

* the 274 056 163-parameter latent UNet of CompVis/ldm-celebahq-256 (block_out_channels [224, 448, 672, 896], head_dim 32, attention in
  3 of 4 levels; reference model.py:706-776) -- one forward + backward at B = 1 on a 3x64x64 latent, in both arithmetics;
* the 55 322 782-parameter VQ-VAE decoding a 64x64 latent to 256x256 (LDMPipeline's last step, loss.py:951-962);
* the config's poisoning: GLASSES -> CAT at 256x256, poison_rate 0.9 (dataset.py:1343-1371 poisons by index; the image-space stamping
  rule of D5 is what make_latent_dataset.py encodes) -- GPU stamping bit-exact against the oracle's per-sample rule.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import backdoor_ref as BR  # noqa: E402
from oracle.unet_ref import UNet2DModelRef  # noqa: E402
from oracle.vqmodel_ref import VQModelRef  # noqa: E402
from villandiffusion_amd.dataset import DatasetLoader, synthetic_images  # noqa: E402
from villandiffusion_amd.model import LDM_CELEBA_UNET_ARCH, LDM_CELEBA_VQ_ARCH  # noqa: E402
from villandiffusion_amd.unet import UNet2DModel  # noqa: E402
from villandiffusion_amd.vqmodel import VQModel  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.timeout(900)
def test_ldm_unet_274m_forward_backward_matches_oracle():
    cfg = {k: (tuple(v) if isinstance(v, list) else v) for k, v in LDM_CELEBA_UNET_ARCH.items()}
    torch.manual_seed(0)
    ref = UNet2DModelRef(**cfg)
    assert sum(p.numel() for p in ref.parameters()) == 274056163
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([417])
    y_ref = ref(x, t)[0]
    w = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(3))
    (y_ref * w).sum().backward()
    gref = {n: p.grad for n, p in ref.named_parameters()}
    gmax = max(float(g.abs().max()) for g in gref.values())
    net = UNet2DModel(**cfg)
    assert sum(p.numel() for p in net.parameters()) == 274056163
    net.load_state_dict(ref.state_dict())
    for conv_math, tol_f, tol_g in (("bf16x3", 1e-4, 1e-3), ("f32", 1e-4, 1e-3)):
        net.conv_math = conv_math
        net.zero_grad()
        y = net(x.cuda(), t.cuda())[0]
        ef = rel(y, y_ref)
        (y * w.cuda()).sum().backward()
        worst = (0.0, "")
        for n, p in net.named_parameters():
            a, b = p.grad.detach().double().cpu(), gref[n].double()
            e = float((a - b).abs().max() / (b.abs().max() + 1e-4 * gmax))
            if e > worst[0]:
                worst = (e, n)
        print(f"[parity] LDM UNet 274M B=1 ({conv_math}): fwd rel_err={ef:.3e}; worst param-grad rel_err={worst[0]:.3e} at {worst[1]}")
        assert ef < tol_f, (conv_math, ef)
        assert worst[0] < tol_g, (conv_math, worst)


@pytest.mark.timeout(900)
def test_vqvae_55m_decodes_a_64x64_latent_to_256x256_like_the_oracle():
    cfg = {k: (tuple(v) if isinstance(v, list) else v) for k, v in LDM_CELEBA_VQ_ARCH.items()}
    torch.manual_seed(2)
    ref = VQModelRef(**cfg)
    assert sum(p.numel() for p in ref.parameters()) == 55322782
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
        ref.quantize.embedding.weight.normal_(0, 0.5)
    net = VQModel(**cfg)
    net.load_state_dict(ref.state_dict())
    lat = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(5)) * 0.5
    with torch.no_grad():
        dec_nq_ref = ref.decode(lat, force_not_quantize=True).sample
        zq_ref, idx_ref = ref.quantize(lat)
        dec_ref = ref.decode(lat).sample
    assert dec_ref.shape == (1, 3, 256, 256)
    e_nq = rel(net.decode(lat.cuda(), force_not_quantize=True).sample, dec_nq_ref)
    zq, idx = net.quantize_latents(lat.cuda(), return_indices=True)
    same = float((idx.cpu() == idx_ref).float().mean())
    e_dec = rel(net.decode(lat.cuda()).sample, dec_ref)
    print(f"[parity] VQ-VAE 55M decode 64x64 -> 256x256: no-quant {e_nq:.2e}, quantised {e_dec:.2e}, identical code indices {same:.4f}")
    assert e_nq < 1e-4 and same > 0.999
    assert e_dec < (1e-4 if same == 1.0 else 5e-2)


def test_glasses_cat_poison_rate_09_stamping_at_256_is_bit_exact():
    n = 20
    dsl = DatasetLoader("SYNTHETIC-CELEBA-HQ", root=ROOT, batch_size=n, seed=0, images=synthetic_images(n=n, size=256, seed=3))
    dsl.set_poison("GLASSES", "CAT", poison_rate=0.9).prepare_dataset(mode="FIXED")
    assert int((dsl._flags & 1).sum()) == int(n * 0.9) == 18
    assert tuple(dsl.trigger.shape) == (3, 256, 256) and float((dsl.trigger > -1.0).float().mean()) > 0.01      # the 160-px glasses image
    flips = torch.rand(n, generator=torch.Generator().manual_seed(1)) < 0.5
    batch = dsl.make_batch(torch.arange(n), flip_bits=flips)
    imgs_u8 = torch.from_numpy(dsl._images[dsl._index])
    pv_ref, tg_ref = BR.poison_batch_ref(imgs_u8, torch.from_numpy((dsl._flags & 1).astype(bool)), dsl.trigger, dsl.target, -1.0, 1.0,
                                         flip=flips.to(torch.uint8))
    assert torch.equal(batch["pixel_values"].cpu(), pv_ref) and torch.equal(batch["target"].cpu(), tg_ref)
    assert int((~batch["is_clean"]).sum()) == 18
    # by-index poisoning of the latent dataset at the same rate (dataset.py:1352-1359): first int(n * 0.9) items
    assert [int(i < int(n * 0.9)) for i in range(n)].count(1) == 18
