"""Drop-in module name of the third-party package the reference imports (`import lpips`, reference VillanDiffusion.py:337;
`lpips.LPIPS(net='alex')` at :892) -- the implementation lives in villandiffusion_amd/lpips.py (AlexNet taps on the HIP kernels)."""
from villandiffusion_amd.lpips import LPIPS, load_lpips_weights  # noqa: F401
