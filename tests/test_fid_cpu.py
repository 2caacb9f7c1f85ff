"""FID (reference fid_score.py:91-284): the oracle restatement of pytorch-fid's InceptionV3 and the Frechet closed form,
checked against known answers on the CPU.  (pytorch-fid / torchvision are third-party and absent: parity unpinned, see oracle/inception_ref.py.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.inception_ref import InceptionV3Ref


def test_inception_oracle_parameter_count_and_block_shapes():
    m = InceptionV3Ref((0, 1, 2, 3))
    # torchvision Inception3: 27 161 264 parameters = this trunk + AuxLogits (3 326 696) + fc 2048 -> 1000 (2 049 000)
    assert sum(p.numel() for p in m.parameters()) == 27_161_264 - 3_326_696 - 2_049_000 == 21_785_568
    keys = set(m.state_dict())
    assert {"Conv2d_1a_3x3.conv.weight", "Conv2d_1a_3x3.bn.running_var", "Mixed_5b.branch5x5_2.conv.weight", "Mixed_6a.branch3x3dbl_3.bn.bias",
            "Mixed_6e.branch7x7dbl_5.conv.weight", "Mixed_7a.branch7x7x3_4.conv.weight", "Mixed_7c.branch3x3dbl_3b.bn.weight"} <= keys
    sd = m.state_dict()
    assert tuple(sd["Mixed_6b.branch7x7_2.conv.weight"].shape) == (128, 128, 1, 7) and tuple(sd["Mixed_6b.branch7x7_3.conv.weight"].shape) == (192, 128, 7, 1)
    assert tuple(sd["Mixed_5b.branch5x5_2.conv.weight"].shape) == (64, 48, 5, 5) and tuple(sd["Mixed_7c.branch1x1.conv.weight"].shape) == (320, 2048, 1, 1)
    outs = m.randomize(0)(torch.rand(2, 3, 40, 24))                # any input size: resized to 299 x 299 first
    assert [tuple(o.shape) for o in outs] == [(2, 64, 73, 73), (2, 192, 35, 35), (2, 768, 17, 17), (2, 2048, 1, 1)]
    assert all(bool(torch.isfinite(o).all()) and float(o.min()) >= 0 for o in outs)            # every block ends in ReLU / pooling of ReLUs
    assert InceptionV3Ref.BLOCK_INDEX_BY_DIM == {64: 0, 192: 1, 768: 2, 2048: 3}


def test_fid_pooling_variants_exclude_the_padding():
    from oracle.inception_ref import _avg3
    x = torch.ones(1, 2, 5, 5)
    assert torch.equal(_avg3(x), x)                                 # zero padding is not averaged in (count_include_pad=False)
    assert float(F.avg_pool2d(x, 3, 1, 1)[0, 0, 0, 0]) == pytest.approx(4 / 9)     # ... which the default would do
    m = InceptionV3Ref()
    assert m.Mixed_7b.pool == "avg" and m.Mixed_7c.pool == "max"   # FIDInceptionE_1 / FIDInceptionE_2


def test_frechet_distance_known_answers():
    from villandiffusion_amd.fid_score import calculate_frechet_distance
    rng = np.random.default_rng(0)
    a = rng.standard_normal((400, 6))
    mu, s = a.mean(0), np.cov(a, rowvar=False)
    assert abs(calculate_frechet_distance(mu, s, mu, s)) < 1e-6
    # commuting (diagonal) covariances: d^2 = |dmu|^2 + sum (sqrt(a) - sqrt(b))^2
    d1, d2 = np.array([1.0, 4.0, 9.0]), np.array([4.0, 1.0, 16.0])
    m1, m2 = np.array([0.0, 1.0, 2.0]), np.array([1.0, 1.0, 0.0])
    want = ((m1 - m2) ** 2).sum() + ((np.sqrt(d1) - np.sqrt(d2)) ** 2).sum()
    assert calculate_frechet_distance(m1, np.diag(d1), m2, np.diag(d2)) == pytest.approx(want, rel=1e-9)
    # symmetric, and invariant under a common rotation of both Gaussians
    b = rng.standard_normal((400, 6)) * 1.5 + 0.3
    mu2, s2 = b.mean(0), np.cov(b, rowvar=False)
    f12, f21 = calculate_frechet_distance(mu, s, mu2, s2), calculate_frechet_distance(mu2, s2, mu, s)
    assert f12 == pytest.approx(f21, rel=1e-8) and f12 > 0
    q, _ = np.linalg.qr(rng.standard_normal((6, 6)))
    assert calculate_frechet_distance(q @ mu, q @ s @ q.T, q @ mu2, q @ s2 @ q.T) == pytest.approx(f12, rel=1e-7)


def test_fid_module_surface_and_missing_weights_error(tmp_path, monkeypatch):
    import fid_score as F0                       # the reference's module name (VillanDiffusion.py:340)
    for name in ("get_activations", "calculate_frechet_distance", "calculate_activation_statistics", "compute_statistics_of_path",
                 "calculate_fid_given_paths", "fid", "InceptionV3", "IMAGE_EXTENSIONS"):
        assert hasattr(F0, name), name
    assert F0.InceptionV3.BLOCK_INDEX_BY_DIM[2048] == 3 and F0.InceptionV3.DEFAULT_BLOCK_INDEX == 3
    from villandiffusion_amd.inception import load_fid_weights
    monkeypatch.setenv("VILLAN_CKPT_ROOT", str(tmp_path))
    monkeypatch.delenv("VILLAN_FID_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError, match="pt_inception-2015-12-05-6726825d.pth"):
        load_fid_weights()
    np.savez(tmp_path / "stats.npz", mu=np.zeros(4), sigma=np.eye(4))
    m, s = F0.compute_statistics_of_path(str(tmp_path / "stats.npz"), None, 8, 4, "cuda")          # precomputed statistics need no network
    assert m.shape == (4,) and s.shape == (4, 4)
    with pytest.raises(RuntimeError, match="Invalid path"):
        F0.calculate_fid_given_paths([str(tmp_path / "nope"), str(tmp_path)], 8, "cuda", 2048, model=object())


def test_lpips_oracle_known_answers(tmp_path, monkeypatch):
    """lpips.LPIPS(net='alex') restated (oracle/lpips_ref.py; reference VillanDiffusion.py:892): AlexNet feature widths / parameter count,
    d(x, x) = 0, symmetry, non-negativity with non-negative lin weights, and the loud failure of the product loader without local weights."""
    from oracle.lpips_ref import CHNS, LPIPSRef
    m = LPIPSRef().randomize(0)
    assert sum(p.numel() for p in m.net.parameters()) == 61_100_840 - 58_631_144 == 2_469_696       # torchvision AlexNet minus its classifier
    x, y = torch.rand(3, 3, 64, 64, generator=torch.Generator().manual_seed(0)), torch.rand(3, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    assert [t.shape[1] for t in m.net(x)] == list(CHNS) and [t.shape[-1] for t in m.net(x)] == [15, 7, 3, 3, 3]
    dxy, dyx, dxx = m(x, y), m(y, x), m(x, x)
    assert tuple(dxy.shape) == (3, 1, 1, 1) and float(dxx.abs().max()) == 0.0
    assert torch.allclose(dxy, dyx, rtol=1e-6) and float(dxy.min()) > 0
    assert set(m.flat_state_dict()) == {f"features.{i}.{k}" for i in (0, 3, 6, 8, 10) for k in ("weight", "bias")} | {f"lin{k}.model.1.weight" for k in range(5)}
    import lpips as L0                                # the package name the reference imports
    monkeypatch.setenv("VILLAN_CKPT_ROOT", str(tmp_path))
    monkeypatch.delenv("VILLAN_ALEXNET_WEIGHTS", raising=False)
    monkeypatch.delenv("VILLAN_LPIPS_WEIGHTS", raising=False)
    with pytest.raises(FileNotFoundError, match="alexnet-owt-7be5be79.pth"):
        L0.load_lpips_weights()
    with pytest.raises(NotImplementedError):
        L0.LPIPS(net="vgg", state_dict={})
