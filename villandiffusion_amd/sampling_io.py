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


def sampler_streams() -> int:
    """Chunks the measure / sampling loops denoise at a time (pipelines.sample_concurrent); VILLAN_SAMPLER_STREAMS=1: one after the other like the reference."""
    return max(1, int(os.environ.get("VILLAN_SAMPLER_STREAMS", "4")))


def _concurrent_ok(pipeline, chunks, eta) -> bool:
    """The chunk list can go through pipelines.sample_concurrent with the SAME images as one pipeline call per chunk: a plain pipeline on the
    HIP-graph forward, explicit inits, and either a deterministic sampler or in-kernel noise (scheduler.device_rng_seed: the drivers set it)."""
    from .pipelines import DiffusionPipeline
    from .unet import UNet2DModel
    if sampler_streams() < 2 or len(chunks) < 2 or any(c is None for c, _ in chunks):
        return False
    if type(pipeline).__call__ is not DiffusionPipeline.__call__ or not hasattr(pipeline, "sample_concurrent"):
        return False                                           # LDM / ScoreSDE-VE / Karras-VE pipelines have their own loops
    unet, sch = pipeline.unet, pipeline.scheduler
    if not (isinstance(unet, UNet2DModel) and getattr(unet, "sampler_graph", False) and unet.device.type == "cuda"):
        return False
    seeded = getattr(sch, "device_rng_seed", None) is not None
    stochastic = type(sch).__name__ in ("DDPMScheduler", "ScoreSdeVeScheduler") or (eta is not None and float(eta) != 0.0)
    return seeded or not stochastic


def _seeded_scheduler(pipeline) -> bool:
    from .pipelines import DiffusionPipeline
    return (type(pipeline).__call__ is DiffusionPipeline.__call__ and hasattr(pipeline, "sample_sequential")
            and getattr(pipeline.scheduler, "device_rng_seed", None) is not None)


def _seeded_chunks(pipeline, chunks) -> bool:
    """Explicit inits + in-kernel noise on a plain pipeline: the per-chunk Philox offsets of pipelines.chunk_rng_offset apply."""
    from .pipelines import DiffusionPipeline
    if not chunks or any(c is None for c, _ in chunks) or type(pipeline).__call__ is not DiffusionPipeline.__call__:
        return False
    return getattr(pipeline.scheduler, "device_rng_seed", None) is not None and hasattr(pipeline, "sample_sequential")


def batch_sampling_save(sample_n: int, pipeline, path: Union[str, os.PathLike], init: torch.Tensor = None, max_batch_n: int = 256,
                        rng: torch.Generator = None, num_inference_steps: Optional[int] = None, eta: Optional[float] = None,
                        rank: int = 0, world: int = 1):
    """model.py:504-527; with `world` > 1 each rank samples its own slice of the chunk list (replicas only).  A rank's chunks are denoised
    sampler_streams() at a time on their own streams where that gives the same images (see _concurrent_ok), else one after the other."""
    cnt, mine, n_total, max_numel = 0, [], 0, 0
    for k, (c, n) in enumerate(_chunks(sample_n, init, max_batch_n)):
        if k % world == rank:
            mine.append((c, n, cnt, k))
        cnt += n
        n_total += 1
        if c is not None:
            max_numel = max(max_numel, c.numel())
    # the Philox offset ranges of the chunks are laid out over the WHOLE job (every rank's chunks, the largest chunk's size): rank-local values
    # made consecutive calls (clean set, then backdoor set) overlap between ranks
    plan = dict(n_chunks_total=n_total, max_numel=max_numel)
    from .pipelines import _post
    if not mine and n_total and max_numel and _seeded_scheduler(pipeline):
        pipeline.sample_sequential([], num_inference_steps=num_inference_steps, eta=eta, chunk_ids=[], **plan)   # no chunk here: advance the offset only
        return None
    if _concurrent_ok(pipeline, [(c, n) for c, n, _, _ in mine], eta):
        xs = pipeline.sample_concurrent([c for c, _, _, _ in mine], num_inference_steps=num_inference_steps, n_streams=sampler_streams(), eta=eta,
                                        chunk_ids=[k for _, _, _, k in mine], **plan)
        for (c, n, start, _), x in zip(mine, xs):
            save_imgs(_post(x), path, start_cnt=start)
        return None
    if _seeded_chunks(pipeline, [(c, n) for c, n, _, _ in mine]):
        # in-kernel noise, one chunk at a time: the same per-chunk Philox ranges as the concurrent walk, so the images (and every score computed
        # from them) do not depend on VILLAN_SAMPLER_STREAMS or on the number of ranks
        xs = pipeline.sample_sequential([c for c, _, _, _ in mine], num_inference_steps=num_inference_steps, eta=eta,
                                        chunk_ids=[k for _, _, _, k in mine], **plan)
        for (c, n, start, _), x in zip(mine, xs):
            save_imgs(_post(x), path, start_cnt=start)
        return None
    for c, n, start, _ in mine:
        kw = {} if eta is None else {"eta": eta}
        res = pipeline(batch_size=n, generator=rng, init=c, output_type=None, num_inference_steps=num_inference_steps, **kw)
        save_imgs(res.images, path, start_cnt=start)
    return None
