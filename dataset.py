"""Drop-in module name of the reference (`from dataset import DatasetLoader, Backdoor`, reference dataset.py:42,639) --
the implementation lives in villandiffusion_amd/dataset.py."""
from villandiffusion_amd.dataset import (DEFAULT_VMAX, DEFAULT_VMIN, Backdoor, DatasetLoader, LatentDataset,  # noqa: F401
                                         normalize, synthetic_images)
