"""FID feature extractor on the HIP kernels (reference fid_score.py:91-148,264-284 -> pytorch-fid InceptionV3) against the CPU
oracle (oracle/inception_ref.py) on seeded random weights -- the published weights are a download the box cannot make."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle.inception_ref import InceptionV3Ref  # noqa: E402
from villandiffusion_amd import ops  # noqa: E402
from villandiffusion_amd.inception import InceptionV3  # noqa: E402

DEV = "cuda"


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("B,Cin,Cout,H,W,kh,kw,stride,ph,pw", [
    (3, 3, 32, 37, 37, 3, 3, 2, 0, 0),       # Conv2d_1a_3x3 (stride 2, 3 input channels)
    (2, 48, 64, 35, 35, 5, 5, 1, 2, 2),      # Mixed_5b.branch5x5_2
    (2, 128, 128, 17, 17, 1, 7, 1, 0, 3), (2, 128, 192, 17, 17, 7, 1, 1, 3, 0),   # Mixed_6b 1x7 / 7x1
    (2, 96, 96, 35, 35, 3, 3, 2, 0, 0),      # Mixed_6a.branch3x3dbl_3
    (2, 384, 384, 8, 8, 1, 3, 1, 0, 1), (2, 384, 384, 8, 8, 3, 1, 1, 1, 0),       # Mixed_7b (1,3) / (3,1)
    (5, 80, 192, 12, 9, 3, 3, 1, 0, 0),      # ragged, non-square
    (2, 288, 64, 35, 35, 1, 1, 1, 0, 0),     # 1x1 -> plain GEMM, ragged pixel count (1225)
])
def test_general_convolution_with_bias_and_relu(B, Cin, Cout, H, W, kh, kw, stride, ph, pw):
    x = torch.randn(B, Cin, H, W, generator=g(0))
    w = torch.randn(Cout, Cin, kh, kw, generator=g(1)) / math.sqrt(Cin * kh * kw)
    b = torch.randn(Cout, generator=g(2)) * 0.3
    y_ref = F.relu(F.conv2d(x, w, b, stride=stride, padding=(ph, pw)))
    OH, OW = y_ref.shape[-2:]
    xbuf = torch.zeros(B, Cin + 2, H, W, device=DEV)            # channel slices of wider buffers on both sides
    xbuf[:, 1:1 + Cin] = x.to(DEV)
    obuf = torch.full((B, Cout + 3, OH, OW), 7.0, device=DEV)
    ops.conv2d_general(xbuf[:, 1:1 + Cin], w.to(DEV).view(Cout, -1), b.to(DEV), obuf[:, 2:2 + Cout], kh, kw, stride, ph, pw, relu=True)
    e = rel(obuf[:, 2:2 + Cout], y_ref)
    print(f"[parity] conv {kh}x{kw} s{stride} p({ph},{pw}) {Cin}->{Cout}@{H}x{W}: rel_err {e:.2e}")
    assert e < 2e-5
    assert float((obuf[:, :2] - 7).abs().max()) == 0 and float((obuf[:, -1] - 7).abs().max()) == 0
    o2 = torch.empty(B, Cout, OH, OW, device=DEV)
    ops.conv2d_general(xbuf[:, 1:1 + Cin], w.to(DEV).view(Cout, -1), b.to(DEV), o2, kh, kw, stride, ph, pw, relu=False)
    assert rel(o2, F.conv2d(x, w, b, stride=stride, padding=(ph, pw))) < 2e-5 and float(o2.min()) < 0


@pytest.mark.parametrize("H,W,stride,pad,mode", [(35, 35, 1, 1, "avg"), (17, 17, 1, 1, "avg"), (8, 8, 1, 1, "max"), (147, 147, 2, 0, "max"),
                                                (71, 71, 2, 0, "max"), (35, 35, 2, 0, "max"), (9, 12, 1, 1, "avg")])
def test_pool3_matches_torch(H, W, stride, pad, mode):
    x = torch.randn(3, 20, H, W, generator=g(3))
    ref = F.max_pool2d(x, 3, stride, pad) if mode == "max" else F.avg_pool2d(x, 3, stride, pad, count_include_pad=False)
    out = torch.full((3, 24) + tuple(ref.shape[-2:]), 5.0, device=DEV)
    ops.pool3(x.to(DEV), out[:, 2:22], stride=stride, pad=pad, mode=mode)
    if mode == "max":
        assert torch.equal(out[:, 2:22].cpu(), ref)
    else:
        assert rel(out[:, 2:22], ref) < 1e-6
    assert float((out[:, :2] - 5).abs().max()) == 0 and float((out[:, 22:] - 5).abs().max()) == 0


@pytest.mark.parametrize("H,W", [(32, 32), (64, 48), (299, 299), (400, 310)])
def test_bilinear_resize_to_299_matches_torch(H, W):
    x = torch.rand(2, 3, H, W, generator=g(4))
    ref = 2 * F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False) - 1
    out = ops.resize_bilinear(x.to(DEV), torch.empty(2, 3, 299, 299, device=DEV), 2.0, -1.0)
    assert rel(out, ref) < 5e-5          # downscaling 400 -> 299: the source coordinate itself carries ~1e-5 of f32 rounding


@pytest.mark.timeout(600)
def test_inception_blocks_match_oracle_on_random_weights():
    ref = InceptionV3Ref((0, 1, 2, 3)).randomize(1)
    net = InceptionV3((0, 1, 2, 3), state_dict=ref.state_dict())
    x = torch.rand(4, 3, 32, 32, generator=g(5))                # CIFAR10-sized samples in [0, 1], as fid_score.py feeds them
    want = ref(x)
    got = net(x)
    assert [tuple(o.shape) for o in got] == [tuple(o.shape) for o in want]
    for i, (a, b) in enumerate(zip(got, want)):
        e = rel(a, b)
        print(f"[parity] InceptionV3 block {i}: rel_err {e:.2e}")
        assert e < 1e-4, (i, e)
    # single-block request (what fid() builds): pool3 only; no resize / no normalisation switches
    p3 = InceptionV3([3], state_dict=ref.state_dict())(x)
    assert len(p3) == 1 and torch.equal(p3[0], got[3])
    r2 = InceptionV3Ref((3,), resize_input=False, normalize_input=False)
    r2.load_state_dict(ref.state_dict())
    x2 = torch.rand(2, 3, 139, 139, generator=g(6))
    n2 = InceptionV3([3], resize_input=False, normalize_input=False, state_dict=ref.state_dict())
    assert rel(n2(x2)[0], r2(x2)[0]) < 1e-4


@pytest.mark.timeout(600)
def test_fid_of_two_png_directories_matches_the_oracle(tmp_path):
    """fid(path=[dir_a, dir_b]) end to end (PNG decode -> pool3 activations on the GPU -> float64 statistics -> Frechet distance) against
    the same statistics taken from the CPU oracle's activations.  dims = 192 keeps the covariance full-rank at 256 images."""
    from PIL import Image
    from villandiffusion_amd import fid_score
    ref = InceptionV3Ref((1,)).randomize(2)
    rng = np.random.default_rng(0)
    sets = []
    for name, shift in (("a", 0), ("b", 40)):
        d = tmp_path / name
        d.mkdir()
        imgs = np.clip(rng.integers(0, 216, size=(256, 32, 32, 3)) + shift, 0, 255).astype(np.uint8)
        for i, im in enumerate(imgs):
            Image.fromarray(im).save(d / f"{i}.png")
        order = sorted(range(256), key=lambda i: str(d / f"{i}.png"))            # the reference sorts the file paths
        sets.append(imgs[order])
    model = InceptionV3([1], state_dict=ref.state_dict())
    got = fid_score.calculate_fid_given_paths([str(tmp_path / "a"), str(tmp_path / "b")], 64, "cuda", 192, num_workers=4, model=model)
    stats = []
    for imgs in sets:
        x = torch.from_numpy(imgs).permute(0, 3, 1, 2).float() / 255.0
        act = torch.cat([ref(x[i:i + 64])[0].mean(dim=(2, 3)) for i in range(0, 256, 64)]).double().numpy()
        stats.append((act.mean(0), np.cov(act, rowvar=False)))
    want = fid_score.calculate_frechet_distance(*stats[0], *stats[1])
    print(f"[parity] FID {got:.6f} vs oracle {want:.6f}")
    assert want > 1e-3 and abs(got - want) <= 1e-3 * abs(want)
    assert fid_score.fid([str(tmp_path / "a"), str(tmp_path / "a")], batch_size=64, dims=192, model=model) == pytest.approx(0.0, abs=1e-6 * max(1.0, want))


def test_lpips_matches_oracle_on_random_weights():
    """lpips.LPIPS(net='alex')(a, b) on the HIP kernels (11x11 stride-4 / 5x5 / 3x3 convolutions + ReLU, max pooling, per-tap normalise /
    difference / lin / mean in one kernel) against oracle/lpips_ref.py, 32x32 (the CIFAR10 measure) and 64x64 inputs in [0, 1]."""
    from oracle.lpips_ref import LPIPSRef
    from villandiffusion_amd.lpips import LPIPS
    ref = LPIPSRef().randomize(3)
    net = LPIPS(net="alex", state_dict=ref.flat_state_dict())
    for S in (32, 64, 96):
        a, b = torch.rand(5, 3, S, S, generator=g(7)), torch.rand(5, 3, S, S, generator=g(8))
        want, got = ref(a, b), net(a, b)
        assert tuple(got.shape) == (5, 1, 1, 1)
        e = rel(got, want)
        print(f"[parity] LPIPS {S}x{S}: rel_err {e:.2e}")
        assert e < 1e-4 and float(net(a, a).abs().max()) == 0.0
    fa, fr = net.features(torch.rand(2, 3, 64, 64, generator=g(9)).cuda()), ref.net((torch.rand(2, 3, 64, 64, generator=g(9)) - ref.shift) / ref.scale)
    for k, (x, y) in enumerate(zip(fa, fr)):
        assert rel(x, y) < 2e-5, k


def test_measure_wiring_fills_fid_and_lpips_when_local_weights_exist(tmp_path, monkeypatch):
    """measure() / measure_inpaints() of the drop-in driver (reference VillanDiffusion.py:1072 and :892): with the weight files present locally the
    scores are numbers (here: seeded random weights in the published key layout), without them None plus the reason -- never a silent fallback."""
    from types import SimpleNamespace
    from PIL import Image
    import VillanDiffusion as V
    from oracle.lpips_ref import LPIPSRef
    rng = np.random.default_rng(1)
    clean = tmp_path / "clean"
    clean.mkdir()
    for i in range(24):
        Image.fromarray(rng.integers(0, 255, size=(32, 32, 3), dtype=np.uint8)).save(clean / f"{i}.png")
    data = rng.integers(0, 255, size=(64, 32, 32, 3), dtype=np.uint8)
    cfg = SimpleNamespace(dataset="SYNTHETIC-CIFAR10", seed=0, eval_max_batch=8)
    dsl = SimpleNamespace(_images=data)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("VILLAN_CKPT_ROOT", str(tmp_path / "nothing-here"))
    for k in ("VILLAN_FID_WEIGHTS", "VILLAN_ALEXNET_WEIGHTS", "VILLAN_LPIPS_WEIGHTS"):
        monkeypatch.delenv(k, raising=False)
    assert V.measure_fid(cfg, dsl, str(clean), 24) is None
    a, b = torch.rand(6, 3, 32, 32, generator=g(1)), torch.rand(6, 3, 32, 32, generator=g(2))
    assert V.measure_lpips(a, b, 4) is None
    # the published files' key layout, random values
    ref = InceptionV3Ref((3,)).randomize(4)
    sd = dict(ref.state_dict())
    sd["fc.weight"], sd["fc.bias"] = torch.zeros(1008, 2048), torch.zeros(1008)          # present in pt_inception-2015-12-05: must be ignored
    torch.save(sd, tmp_path / "inc.pth")
    monkeypatch.setenv("VILLAN_FID_WEIGHTS", str(tmp_path / "inc.pth"))
    fid_sc = V.measure_fid(cfg, dsl, str(clean), 24)
    assert isinstance(fid_sc, float) and np.isfinite(fid_sc) and fid_sc >= -1e-3
    assert len(os.listdir(tmp_path / "measure" / "SYNTHETIC-CIFAR10")) == 24              # the dataset side was written as PNGs (reference :1043-1049)
    lp = LPIPSRef().randomize(5)
    fsd = lp.flat_state_dict()
    torch.save({k: v for k, v in fsd.items() if k.startswith("features.")}, tmp_path / "alexnet.pth")
    torch.save({k: v for k, v in fsd.items() if k.startswith("lin")}, tmp_path / "lins.pth")
    monkeypatch.setenv("VILLAN_ALEXNET_WEIGHTS", str(tmp_path / "alexnet.pth"))
    monkeypatch.setenv("VILLAN_LPIPS_WEIGHTS", str(tmp_path / "lins.pth"))
    got = V.measure_lpips(a, b, 4)
    want = float(lp(a, b).mean())
    assert abs(got - want) <= 1e-4 * abs(want)


@pytest.mark.parametrize("N,C,H,W", [(5, 3, 32, 32), (3, 1, 28, 28), (2, 3, 64, 48), (4, 3, 8, 8)])
def test_ssim_kernel_matches_oracle(N, C, H, W):
    """vd_ssim (measure(): VillanDiffusion.py:1001-1007, torchmetrics SSIM) against oracle/metrics_ref.py; 8 x 8 images are smaller than the
    cropped border, so the reflect-indexed window is exercised too; a CUDA input never takes the host expression."""
    from oracle.metrics_ref import ssim_ref
    from villandiffusion_amd.metrics import ssim_batch
    a = torch.rand((N, C, H, W), generator=g(1))
    b = (a + 0.2 * torch.randn((N, C, H, W), generator=g(2))).clamp(0, 1)
    want = ssim_ref(a.numpy(), b.numpy())
    got = ssim_batch(a.to(DEV), b.to(DEV))
    got_dev_arg = ssim_batch(a, b[:1].expand(N, -1, -1, -1) if False else b, device=DEV, chunk=2)
    assert abs(got - want) < 2e-5 and abs(got_dev_arg - want) < 2e-5, (got, got_dev_arg, want)
    assert abs(ssim_batch(a.to(DEV), a.to(DEV)) - 1.0) < 1e-6
