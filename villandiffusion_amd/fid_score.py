"""Frechet Inception Distance with the reference's function surface (fid_score.py:91-284: `get_activations`,
`calculate_frechet_distance`, `calculate_activation_statistics`, `compute_statistics_of_path`, `calculate_fid_given_paths`, `fid`),
consumed by the measure pipeline at VillanDiffusion.py:1072:

    fid(path=[dataset_img_dir, clean_path], device=..., num_workers=4, batch_size=config.eval_max_batch)

The pool3 activations come from `villandiffusion_amd.inception.InceptionV3` (HIP kernels; there is no CPU fallback); mean / covariance /
matrix square root are float64 numpy / scipy on the host exactly as the reference does them.  Images are decoded with PIL in a thread
pool (`ToTensor()` semantics: RGB, CHW, /255).  Same quirks kept: `batch_size` is clamped to the number of files; a `.npz` path supplies
precomputed `mu` / `sigma`.
"""
from __future__ import annotations

import os
import pathlib
from concurrent.futures import ThreadPoolExecutor
from typing import List

import numpy as np
import torch

from .inception import InceptionV3
from .metrics import frechet_distance

IMAGE_EXTENSIONS = {"bmp", "jpg", "jpeg", "pgm", "png", "ppm", "tif", "tiff", "webp"}


def _load(path) -> np.ndarray:
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def get_activations(files, model, batch_size=50, dims=2048, device="cuda", num_workers=1):
    """[len(files), dims] float64 activations of the requested Inception block (global-average-pooled when it is a feature map)."""
    if batch_size > len(files):
        print("Warning: batch size is bigger than the data size. Setting batch size to data size")
        batch_size = len(files)
    pred_arr = np.empty((len(files), dims))
    start = 0
    with ThreadPoolExecutor(max_workers=max(1, num_workers or 1)) as pool:
        for i in range(0, len(files), batch_size):
            imgs = list(pool.map(_load, files[i:i + batch_size]))
            batch = torch.from_numpy(np.stack(imgs)).permute(0, 3, 1, 2).float().div_(255.0)
            pred = model(batch)[0]
            if pred.size(2) != 1 or pred.size(3) != 1:
                pred = pred.mean(dim=(2, 3), keepdim=True)
            pred = pred.squeeze(3).squeeze(2).cpu().numpy()
            pred_arr[start:start + pred.shape[0]] = pred
            start += pred.shape[0]
    return pred_arr


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    return frechet_distance(mu1, sigma1, mu2, sigma2, eps)


def calculate_activation_statistics(files, model, batch_size=50, dims=2048, device="cuda", num_workers=1):
    act = get_activations(files, model, batch_size, dims, device, num_workers)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def compute_statistics_of_path(path, model, batch_size, dims, device, num_workers=1):
    if str(path).endswith(".npz"):
        with np.load(path) as f:
            return f["mu"][:], f["sigma"][:]
    path = pathlib.Path(path)
    files = sorted([file for ext in IMAGE_EXTENSIONS for file in path.glob("*.{}".format(ext))])
    return calculate_activation_statistics(files, model, batch_size, dims, device, num_workers)


def calculate_fid_given_paths(paths, batch_size, device, dims, num_workers=1, model=None):
    for p in paths:
        if not os.path.exists(p):
            raise RuntimeError("Invalid path: %s" % p)
    if model is None:
        model = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]], device=device)
    m1, s1 = compute_statistics_of_path(paths[0], model, batch_size, dims, device, num_workers)
    m2, s2 = compute_statistics_of_path(paths[1], model, batch_size, dims, device, num_workers)
    return calculate_frechet_distance(m1, s1, m2, s2)


def fid(path: List[str], batch_size: int = 50, dims: int = 2048, device=None, num_workers: int = None, model=None):
    if device is None or (isinstance(device, str) and not device.startswith("cuda")):
        device = torch.device("cuda", torch.cuda.current_device())
    elif isinstance(device, int):
        device = torch.device("cuda", device)
    if num_workers is None:
        num_workers = min(len(os.sched_getaffinity(0)), 8)
    fid_value = calculate_fid_given_paths(path, batch_size, device, dims, num_workers, model=model)
    print("FID: ", fid_value)
    return fid_value
