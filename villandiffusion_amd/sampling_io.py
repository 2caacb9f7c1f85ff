"""Chunked sampling + PNG output helpers with the reference's names (model.py:468-527)."""
from __future__ import annotations

import os
from typing import Optional, Union

import numpy as np
import torch


def save_imgs(imgs: np.ndarray, file_dir: Union[str, os.PathLike], file_name: Union[str, os.PathLike] = "", start_cnt: int = 0) -> None:
    """uint8 = round(img*255); one PNG per sample named <file_name><index>.png (model.py:496-502)."""
    from PIL import Image
    os.makedirs(file_dir, exist_ok=True)
    arr = (np.asarray(imgs) * 255).round().astype("uint8")
    for i, im in enumerate(arr):
        Image.fromarray(np.squeeze(im)).save(os.path.join(file_dir, f"{file_name}{start_cnt + i}.png"))


def _chunks(sample_n: int, init: Optional[torch.Tensor], max_batch_n: int):
    if init is None:
        sizes = [max_batch_n] * (sample_n // max_batch_n) + ([sample_n % max_batch_n] if sample_n % max_batch_n else [])
        return [(None, s) for s in sizes]
    return [(c, len(c)) for c in torch.split(init, max_batch_n)]


def batch_sampling(sample_n: int, pipeline, init: torch.Tensor = None, max_batch_n: int = 256, rng: torch.Generator = None):
    outs = [pipeline(batch_size=n, generator=rng, init=c, output_type=None).images for c, n in _chunks(sample_n, init, max_batch_n)]
    return np.concatenate(outs)


def batch_sampling_save(sample_n: int, pipeline, path: Union[str, os.PathLike], init: torch.Tensor = None, max_batch_n: int = 256,
                        rng: torch.Generator = None, num_inference_steps: Optional[int] = None, eta: Optional[float] = None,
                        rank: int = 0, world: int = 1):
    """model.py:504-527; with `world` > 1 each rank samples its own contiguous slice of the chunk list (replicas only)."""
    cnt = 0
    for k, (c, n) in enumerate(_chunks(sample_n, init, max_batch_n)):
        if k % world == rank:
            kw = {} if eta is None else {"eta": eta}
            res = pipeline(batch_size=n, generator=rng, init=c, output_type=None, num_inference_steps=num_inference_steps, **kw)
            save_imgs(res.images, path, start_cnt=cnt)
        cnt += n
    return None
