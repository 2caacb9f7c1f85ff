"""Sampling pipelines with the call contract of the reference's diffusers FORK (call sites
VillanDiffusion.py:579-583, 843-852; model.py:482-488, 519-523):

    pipeline(batch_size, generator, init=None, num_inference_steps=N, start_from=0, save_every_step=False,
             output_type=None[, eta]) -> result with .images (numpy NHWC float in [0,1]) and .movie (list of frames,
             movie[0] = the init frame)

plus ``.unet``, ``.scheduler``, ``.device``, ``.to(device)``, ``.encode(x)`` (identity for pixel models) and
``save_pretrained`` / ``from_pretrained`` in the diffusers directory layout.
The denoising loop is the reference's: ``x = step(unet(x, t), t, x)``; every tensor op in it is a HIP kernel
(UNet launch sequence + one fused scheduler kernel), the final ``(x/2+0.5).clamp(0,1)`` + NCHW->NHWC is one kernel.
[UPSTREAM, unverified] where the fork's exact behaviour is not visible in the reference tree (movie contents).
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import ops
from .schedulers import SCHEDULER_CLASSES, DDIMScheduler, DDPMScheduler, ScoreSdeVeScheduler
from .unet import UNet2DModel


def _post(x: torch.Tensor) -> np.ndarray:
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    ops.postprocess(x.contiguous(), out, 0.5, 0.5, 0.0, 1.0, True)
    return out.cpu().numpy()


class GraphedForward:
    """The no-grad forward of a native UNet captured ONCE into a HIP graph per batch shape and replayed every sampler step
    (the ~250 launches of a denoising step are 5-20 us kernels at the 4x4 / 8x8 stages; eager launch leaves the GPU idle
    between them once the host falls behind).  The graph reads the network's own parameter / packed-operand buffers, so it stays
    valid across optimiser steps; the split-precision operands are refreshed (outside the graph) when the weights have changed.
    Inputs are copied into static buffers, the output out of one."""

    def __init__(self, unet, batch, dtype=torch.float32, slot: int = 0):
        dev = unet.device
        self.unet = unet
        self.slot = slot                                  # split-K workspace slot: graphs replayed concurrently must not share one (ops.ws_slot)
        self.x = torch.zeros((batch, unet.in_channels, unet.sample_size, unet.sample_size), device=dev, dtype=dtype)
        self.t = torch.zeros((batch,), device=dev, dtype=torch.float32)
        self.key = self._key()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with ops.ws_slot(slot):
            with torch.cuda.stream(side):                 # warm-up on a side stream: workspaces, packed operands, allocator pools
                for _ in range(2):
                    unet._run_forward(self.x, self.t, save=False)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._keep = (ops.gemm_ws_buffer(dev, slot), getattr(unet, "_packed", None), getattr(unet, "_packed16", None))     # buffers whose addresses are baked into the graph
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: a HIP call from ANOTHER thread during capture (the RCCL watchdog of a DDP run whose rank 0 samples) must not abort it
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = unet._run_forward(self.x, self.t, save=False)[0]

    def _key(self):
        u = self.unet
        return (u.conv_math, u.fuse_gn_inference, getattr(u, "fused_attention", None), u.sample_size, u.in_channels, u.flat_param.data_ptr())

    def valid(self) -> bool:
        return self.key == self._key() and self._keep[0] is ops.gemm_ws_buffer(self.unet.device, self.slot)

    def __call__(self, x, t):
        u = self.unet
        if hasattr(u, "refresh_packed"):                  # weights changed since the last replay: rebuild the packed operands (one launch)
            u.refresh_packed(False)
        self.x.copy_(x)
        self.t.copy_(t)
        self.graph.replay()
        return self.out.clone()          # multistep samplers keep model outputs across steps: never hand out the static buffer itself


def sampler_forward(unet, batch, slot: int = 0):
    """eps = f(x, t) for the sampler loops: the HIP-graph replay where the network supports it (UNet2DModel, `unet.sampler_graph`), else
    the eager launch sequence.  `slot` > 0: a second, independent graph of the same batch size for a chunk that runs concurrently on another stream."""
    use = getattr(unet, "sampler_graph", False) and isinstance(unet, UNet2DModel) and unet.device.type == "cuda"
    if not use:
        if slot:
            raise RuntimeError("concurrent sampler chunks need the HIP-graph forward (UNet2DModel with sampler_graph)")
        return lambda x, t: unet(x, t, return_dict=False)[0]
    cache = unet.__dict__.setdefault("_fwd_graphs", {})
    g = cache.pop((batch, slot), None)
    if g is None or not g.valid():
        g = None
        while len(cache) >= MAX_CACHED_GRAPHS:            # each graph's private pool pins the peak memory of a no-grad forward: keep the newest few
            cache.pop(next(iter(cache)))
        g = GraphedForward(unet, batch, slot=slot)
    cache[(batch, slot)] = g                              # (re-)inserted last: the dict is the LRU order
    return g


MAX_CACHED_GRAPHS = 4


def drop_sampler_graphs(unet):
    """Release the captured forwards (and the memory pools they pin) -- called when training resumes after a periodic sampling() pass:
    for the 256x256 models the pinned forward peak can otherwise make the next training step run out of memory."""
    unet.__dict__.pop("_fwd_graphs", None)


class DiffusionPipeline:
    _class_name = "DiffusionPipeline"
    default_steps = 1000

    def __init__(self, unet: UNet2DModel, scheduler, vqvae=None, clip_sample=None, clip_sample_range=None):
        self.unet, self.scheduler, self.vqvae = unet, scheduler, vqvae
        self.clip_sample, self.clip_sample_range = clip_sample, clip_sample_range

    @property
    def device(self):
        return self.unet.device

    def to(self, device=None):
        return self

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def _step_kwargs(self, generator, eta):
        return {"generator": generator}

    @torch.no_grad()
    def __call__(self, batch_size: int = 1, generator: Optional[torch.Generator] = None, init: Optional[torch.Tensor] = None,
                 num_inference_steps: Optional[int] = None, start_from: int = 0, save_every_step: bool = False,
                 output_type: Optional[str] = None, eta: Optional[float] = None, return_dict: bool = True,
                 return_tensor: bool = False):
        unet, sched, dev = self.unet, self.scheduler, self.device
        n = num_inference_steps if num_inference_steps is not None else self.default_steps
        shape = (batch_size, unet.in_channels, unet.sample_size, unet.sample_size)
        if init is None:
            if generator is not None and generator.device.type == "cpu":
                x = torch.randn(shape, generator=generator).to(dev)
            else:
                x = torch.randn(shape, generator=generator, device=dev)
        else:
            x = init.to(dev).float().contiguous()
            batch_size = x.shape[0]
        sched.set_timesteps(n)
        ts = sched.timesteps[start_from:]
        # one [n, B] fp32 table of timesteps on the device: row k is the UNet's timestep argument at step k
        t_tab = ts.to(torch.float32).to(dev)[:, None].expand(len(ts), batch_size).contiguous()
        movie = [_post(x)] if (save_every_step or init is not None) else []
        kw = self._step_kwargs(generator, eta)
        sigma_space = float(sched.init_noise_sigma) != 1.0      # Heun / LMSD: state lives in sigma space (schedulers._SigmaSpace)
        if sigma_space and start_from == 0:
            x = ops.lincomb(torch.empty_like(x), [x], [float(sched.init_noise_sigma)])
        fwd = sampler_forward(unet, batch_size)
        for k, t in enumerate(ts if sigma_space else ts.tolist()):
            x_in = sched.scale_model_input(x, t) if sigma_space else x
            eps = fwd(x_in, t_tab[k])
            x = sched.step(eps, t, x, **kw).prev_sample
            if save_every_step:
                movie.append(_post(x))
        if return_tensor:
            return x
        images = _post(x)
        if output_type == "pil":
            from PIL import Image
            images = [Image.fromarray(im.squeeze()) for im in (images * 255).round().astype("uint8")]
        return SimpleNamespace(images=images, movie=movie)

    @torch.no_grad()
    def sample_concurrent(self, inits, num_inference_steps: Optional[int] = None, n_streams: int = 2, eta: Optional[float] = None,
                          chunk_ids=None, n_chunks_total: Optional[int] = None, max_numel: Optional[int] = None):
        """Throughput mode of the measure / sampling loops (reference VillanDiffusion.py:1062-1067 walks its chunks of `eval_max_batch` one after
        the other): the chunks of `inits` are denoised `n_streams` at a time, each on its own HIP stream with its own captured forward, its own
        split-K workspace and its own copy of the scheduler state, stepping in lock-step on the host.  A denoising step is ~250 kernels, a fifth
        of them grids that cannot fill the chip (the 8x8 / 4x4 stages, GroupNorm of small maps, the time-embedding linears) plus a dependency gap
        after every kernel: a second chunk's launches fill those holes.  Noise comes from the in-kernel Philox stream (`scheduler.device_rng_seed`
        must be set); chunk c owns the offset range starting at `chunk_rng_offset(c, ...)`, so a chunk's result does not depend on what runs beside
        it: it is bit-identical to a sequential `__call__` started at the same offset (`sample_sequential` does exactly that).  `chunk_ids`:
        GLOBAL index of every chunk in the caller's chunk list (rank-sharded measure jobs; default 0, 1, ...), so that two ranks never draw the
        same noise and a chunk's images do not depend on the world size; `n_chunks_total` / `max_numel`: chunk count and largest chunk of the
        WHOLE job (all ranks) -- the stride of the offset ranges and the amount the scheduler's offset advances by are then the same on every
        rank, including ranks that own only the short last chunk or none at all (defaults: this call's own list).  Returns the list of final
        states (device tensors)."""
        import copy
        unet, dev = self.unet, self.device
        n = num_inference_steps if num_inference_steps is not None else self.default_steps
        base = self.scheduler
        if getattr(base, "device_rng_seed", None) is None and type(base).__name__ in ("DDPMScheduler", "ScoreSdeVeScheduler"):
            raise RuntimeError("sample_concurrent draws its noise on the device: set scheduler.device_rng_seed")
        base.set_timesteps(n)
        sigma_space = float(base.init_noise_sigma) != 1.0
        ts = base.timesteps
        outs = [None] * len(inits)
        off0 = getattr(base, "_rng_offset", 0)
        main = torch.cuda.current_stream(dev)
        streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        if hasattr(unet, "refresh_packed"):
            # ONE rebuild of the packed split-precision weights, on `main`, before any chunk stream branches off it: GraphedForward.__call__
            # refreshes on whichever stream is current and marks the version fresh, so a refresh left to chunk 0's stream would race with the
            # replays of chunks 1.. on their own streams (they would skip it and read the buffer while it is being rewritten).
            unet.refresh_packed(False)
        numel, chunk_ids, n_total = self._chunk_plan(inits, chunk_ids, n_chunks_total, max_numel)
        for first in range(0, len(inits), n_streams):
            group = list(range(first, min(first + n_streams, len(inits))))
            states = []
            for k, ci in enumerate(group):
                sch = copy.deepcopy(base)
                if hasattr(sch, "_rng_offset"):
                    sch._rng_offset = off0 + self.chunk_rng_offset(chunk_ids[ci], len(ts), numel)
                streams[k].wait_stream(main)
                with torch.cuda.stream(streams[k]), ops.ws_slot(k):
                    # upload on the chunk's OWN stream: the block is then owned by the stream that reads it (a block allocated on `main` and
                    # dropped below could be handed to the next chunk's host -> device copy while this stream still reads it)
                    x = inits[ci].to(dev).float().contiguous()
                    if x.data_ptr() == inits[ci].data_ptr():
                        x.record_stream(streams[k])       # a caller-owned device tensor: keep its block alive for this stream's reads
                    if sigma_space:
                        x = ops.lincomb(torch.empty_like(x), [x], [float(sch.init_noise_sigma)])
                    fwd = sampler_forward(unet, x.shape[0], slot=k)
                    t_tab = ts.to(torch.float32).to(dev)[:, None].expand(len(ts), x.shape[0]).contiguous()
                states.append([sch, x, fwd, t_tab])
            kw = self._step_kwargs(None, eta)
            for step, t in enumerate(ts if sigma_space else ts.tolist()):
                for k, st in enumerate(states):
                    sch, x, fwd, t_tab = st
                    with torch.cuda.stream(streams[k]), ops.ws_slot(k):
                        x_in = sch.scale_model_input(x, t) if sigma_space else x
                        eps = fwd(x_in, t_tab[step])
                        st[1] = sch.step(eps, t, x, **kw).prev_sample
            for k, ci in enumerate(group):
                main.wait_stream(streams[k])
                outs[ci] = states[k][1]
        if hasattr(base, "_rng_offset"):
            base._rng_offset = off0 + self.chunk_rng_offset(n_total, len(ts), numel)
        return outs

    @staticmethod
    def _chunk_plan(inits, chunk_ids, n_chunks_total, max_numel):
        """(stride numel, global chunk ids, global chunk count) of a chunked sampling call; see `sample_concurrent`."""
        chunk_ids = list(range(len(inits))) if chunk_ids is None else [int(c) for c in chunk_ids]
        own = max((c.numel() for c in inits), default=0)
        numel = own if max_numel is None else int(max_numel)
        n_total = (max(chunk_ids) + 1 if chunk_ids else 0) if n_chunks_total is None else int(n_chunks_total)
        if numel < own or any(c >= n_total for c in chunk_ids):
            raise ValueError("chunk plan: max_numel / n_chunks_total smaller than this call's own chunks")
        return numel, chunk_ids, n_total

    @torch.no_grad()
    def sample_sequential(self, inits, num_inference_steps: Optional[int] = None, eta: Optional[float] = None, chunk_ids=None,
                          n_chunks_total: Optional[int] = None, max_numel: Optional[int] = None):
        """The chunks of `inits` one after the other with the SAME per-chunk Philox offsets as `sample_concurrent`: seeded stochastic samplers
        then give identical images whatever VILLAN_SAMPLER_STREAMS / the world size is.  Returns the list of final states (device tensors)."""
        base = self.scheduler
        n = num_inference_steps if num_inference_steps is not None else self.default_steps
        base.set_timesteps(n)
        off0 = getattr(base, "_rng_offset", 0)
        numel, chunk_ids, n_total = self._chunk_plan(inits, chunk_ids, n_chunks_total, max_numel)
        outs = []
        kw = {} if eta is None else {"eta": eta}
        for ci, c in enumerate(inits):
            if hasattr(base, "_rng_offset"):
                base._rng_offset = off0 + self.chunk_rng_offset(chunk_ids[ci], n, numel)
            outs.append(self(batch_size=len(c), init=c, num_inference_steps=n, return_tensor=True, **kw))
        if hasattr(base, "_rng_offset"):
            base._rng_offset = off0 + self.chunk_rng_offset(n_total, n, numel)
        return outs

    @staticmethod
    def chunk_rng_offset(chunk: int, n_steps: int, numel: int) -> int:
        """First Philox offset of chunk `chunk` relative to the scheduler's offset at the call: disjoint ranges of 2 draws per step (no sampler
        here draws more) of the largest chunk."""
        return chunk * 2 * n_steps * ((numel + 3) // 4)

    # ---- diffusers on-disk layout ----
    def save_pretrained(self, save_directory: str, safe_serialization: bool = True):
        os.makedirs(os.path.join(save_directory, "unet"), exist_ok=True)
        os.makedirs(os.path.join(save_directory, "scheduler"), exist_ok=True)
        index = {"_class_name": self._class_name, "_diffusers_version": "0.16.1",
                 "unet": ["diffusers", "UNet2DModel"], "scheduler": ["diffusers", self.scheduler._class_name]}
        if self.vqvae is not None:
            index["vqvae"] = ["diffusers", "VQModel"]
            os.makedirs(os.path.join(save_directory, "vqvae"), exist_ok=True)
            vcfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.vqvae.config).items()}
            vcfg["_class_name"] = "VQModel"
            with open(os.path.join(save_directory, "vqvae", "config.json"), "w") as f:
                json.dump(vcfg, f, indent=2)
            vsd = {k: v.detach().cpu().contiguous().clone() for k, v in self.vqvae.state_dict().items()}
            if safe_serialization:
                from safetensors.torch import save_file
                save_file(vsd, os.path.join(save_directory, "vqvae", "diffusion_pytorch_model.safetensors"))
            else:
                torch.save(vsd, os.path.join(save_directory, "vqvae", "diffusion_pytorch_model.bin"))
        with open(os.path.join(save_directory, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)
        cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self.unet.config).items()}
        cfg["_class_name"] = "UNet2DModel"
        with open(os.path.join(save_directory, "unet", "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        sd = {k: v.detach().cpu().contiguous().clone() for k, v in self.unet.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "unet", "diffusion_pytorch_model.safetensors"))
        else:
            torch.save(sd, os.path.join(save_directory, "unet", "diffusion_pytorch_model.bin"))
        with open(os.path.join(save_directory, "scheduler", "scheduler_config.json"), "w") as f:
            json.dump({k: v for k, v in self.scheduler.scheduler_config().items() if v is None or isinstance(v, (int, float, str, bool, list))},
                      f, indent=2)

    @classmethod
    def from_pretrained(cls, path: str, **kwargs):
        if not os.path.isdir(path):
            raise FileNotFoundError(
                f"{path}: not a local diffusers checkpoint directory (hub ids such as 'google/ddpm-cifar10-32' need a "
                f"network/HF cache; download the repo and pass its directory)")
        with open(os.path.join(path, "unet", "config.json")) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        if cfg.get("time_embedding_type", "positional") == "fourier":         # NCSN++ (fusing/cifar10-ncsnpp-ve, google/ncsnpp-*)
            from .ncsnpp import NCSNppModel
            unet = NCSNppModel(**cfg)
        else:
            unet = UNet2DModel(**cfg)
        st = os.path.join(path, "unet", "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "unet", "diffusion_pytorch_model.bin"), map_location="cpu")
        unet.load_state_dict(sd)
        with open(os.path.join(path, "scheduler", "scheduler_config.json")) as f:
            scfg = json.load(f)
        name = scfg.pop("_class_name", "DDPMScheduler")
        scfg = {k: v for k, v in scfg.items() if not k.startswith("_")}
        if name not in SCHEDULER_CLASSES:
            raise NotImplementedError(f"{path}: scheduler class '{name}' (implemented: {sorted(SCHEDULER_CLASSES)})")
        sched = SCHEDULER_CLASSES[name](**scfg)       # unsupported variance_type / prediction_type raise in the constructor
        vdir = os.path.join(path, "vqvae")
        if os.path.isdir(vdir):
            from .vqmodel import VQModel
            with open(os.path.join(vdir, "config.json")) as f:
                vcfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
            vq = VQModel(**vcfg)
            vst = os.path.join(vdir, "diffusion_pytorch_model.safetensors")
            if os.path.exists(vst):
                from safetensors.torch import load_file
                vq.load_state_dict(load_file(vst))
            else:
                vq.load_state_dict(torch.load(os.path.join(vdir, "diffusion_pytorch_model.bin"), map_location="cpu"))
            return LDMPipeline(vqvae=vq, unet=unet, scheduler=sched)
        return cls(unet, sched)


class DDPMPipeline(DiffusionPipeline):
    _class_name = "DDPMPipeline"
    default_steps = 1000


class DDIMPipeline(DiffusionPipeline):
    _class_name = "DDIMPipeline"
    default_steps = 50

    def _step_kwargs(self, generator, eta):
        return {"generator": generator, "eta": 0.0 if eta is None else eta}


class ScoreSdeVePipeline(DiffusionPipeline):
    """[UPSTREAM] ScoreSdeVePipeline: x = z * sigma_max; per step `correct_steps` Langevin corrections then the
    predictor; the UNet is called with sigma_t; result = sample_mean.clamp(0, 1) (reference model.py:683-684)."""
    _class_name = "ScoreSdeVePipeline"
    default_steps = 2000

    @torch.no_grad()
    def __call__(self, batch_size: int = 1, generator=None, init=None, num_inference_steps: Optional[int] = None, start_from: int = 0,
                 save_every_step: bool = False, output_type=None, return_dict: bool = True, return_tensor: bool = False, **_):
        unet, sched, dev = self.unet, self.scheduler, self.device
        n = num_inference_steps if num_inference_steps is not None else self.default_steps
        shape = (batch_size, unet.in_channels, unet.sample_size, unet.sample_size)
        if init is None:
            z = torch.randn(shape, generator=generator) if (generator is None or generator.device.type == "cpu") else \
                torch.randn(shape, generator=generator, device=dev)
            x = (z * sched.init_noise_sigma).to(dev)
        else:
            x = init.to(dev).float().contiguous()
            batch_size = x.shape[0]
        sched.set_timesteps(n)
        sched.set_sigmas(n)
        mean = x

        def post(t):
            B, C, H, W = t.shape
            out = torch.empty((B, H, W, C), device=t.device, dtype=torch.float32)
            ops.postprocess(t.contiguous(), out, 1.0, 0.0, 0.0, 1.0, True)
            return out.cpu().numpy()

        movie = [post(x)] if (save_every_step or init is not None) else []
        sig_tab = sched.sigmas.to(torch.float32).to(dev)[:, None].expand(n, batch_size).contiguous()
        for i in range(start_from, n):
            t = sched.timesteps[i]
            for _k in range(sched.config.correct_steps):
                score = unet(x, sig_tab[i], return_dict=False)[0]
                x = sched.step_correct(score, x, generator=generator).prev_sample
            score = unet(x, sig_tab[i], return_dict=False)[0]
            out = sched.step_pred(score, t, x, generator=generator)
            x, mean = out.prev_sample, out.prev_sample_mean
            if save_every_step:
                movie.append(post(mean))
        if return_tensor:
            return mean
        return SimpleNamespace(images=post(mean), movie=movie)


class KarrasVePipeline(DiffusionPipeline):
    """[UPSTREAM] KarrasVePipeline (reference model.py:685-693): stochastic 2nd-order sampler of Karras et al. 2022; the
    UNet is called as  (sigma/2) * unet((x+1)/2, sigma/2)  twice per step (predictor + Heun-style correction)."""
    _class_name = "KarrasVePipeline"
    default_steps = 50

    @torch.no_grad()
    def __call__(self, batch_size: int = 1, generator=None, init=None, num_inference_steps: Optional[int] = None, start_from: int = 0,
                 save_every_step: bool = False, output_type=None, return_dict: bool = True, return_tensor: bool = False, **_):
        unet, sched, dev = self.unet, self.scheduler, self.device
        n = num_inference_steps if num_inference_steps is not None else self.default_steps
        shape = (batch_size, unet.in_channels, unet.sample_size, unet.sample_size)
        if init is None:
            z = torch.randn(shape, generator=generator) if (generator is None or generator.device.type == "cpu") else \
                torch.randn(shape, generator=generator, device=dev)
            x = (z * sched.init_noise_sigma).to(dev)
        else:
            x = init.to(dev).float().contiguous()
            batch_size = x.shape[0]
        sched.set_timesteps(n)
        B, C, H, W = x.shape
        inf = float("inf")

        def half(t):          # (t + 1) / 2 on the device
            return ops.postprocess(t.contiguous(), torch.empty_like(t), 0.5, 0.5, -inf, inf, False)

        def model(xx, sigma):
            sig = torch.full((batch_size,), float(sigma) / 2, device=dev, dtype=torch.float32)
            mo = unet(half(xx), sig, return_dict=False)[0]
            return ops.lincomb(torch.empty_like(mo), [mo.contiguous()], [float(sigma) / 2])

        movie = [_post(x)] if (save_every_step or init is not None) else []
        for t in sched.timesteps[start_from:].tolist():
            sigma = sched.schedule[t]
            sigma_prev = sched.schedule[t - 1] if t > 0 else 0
            x_hat, sigma_hat = sched.add_noise_to_input(x, sigma, generator=generator)
            out = sched.step(model(x_hat, sigma_hat), sigma_hat, sigma_prev, x_hat)
            if sigma_prev != 0:
                out = sched.step_correct(model(out.prev_sample, sigma_prev), sigma_hat, sigma_prev, x_hat, out.prev_sample,
                                         out.derivative)
            x = out.prev_sample
            if save_every_step:
                movie.append(_post(x))
        if return_tensor:
            return x
        return SimpleNamespace(images=_post(x), movie=movie)


class LDMPipeline(DiffusionPipeline):
    """[UPSTREAM] LDMPipeline with the fork's call contract (reference model.py:713-769): the generic sampler loop in the
    latent space of the VQ-VAE, then `vqvae.decode(latents)`; `.encode(x)` maps pixels to (unquantised) latents
    (VillanDiffusion.py:632,648,1054: trigger / poisoned images are encoded before they are added to the latent noise).
    `.movie` frames are decoded lazily only for the frames kept (decode of every step would dominate the loop)."""
    _class_name = "LDMPipeline"
    default_steps = 50

    def __init__(self, vqvae=None, unet=None, scheduler=None, clip_sample=None, clip_sample_range=None):
        if vqvae is None:
            raise ValueError("LDMPipeline needs a vqvae")
        super().__init__(unet, scheduler, vqvae=vqvae, clip_sample=clip_sample, clip_sample_range=clip_sample_range)

    def encode(self, x: torch.Tensor) -> torch.Tensor:
        return self.vqvae.encode(x.to(self.device).float()).latents

    def _step_kwargs(self, generator, eta):
        import inspect
        kw = {}
        params = inspect.signature(self.scheduler.step).parameters
        if "eta" in params:
            kw["eta"] = 0.0 if eta is None else eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    @torch.no_grad()
    def __call__(self, batch_size: int = 1, generator=None, init=None, num_inference_steps: Optional[int] = None, start_from: int = 0,
                 save_every_step: bool = False, output_type=None, eta: Optional[float] = None, return_dict: bool = True,
                 return_tensor: bool = False):
        lat = super().__call__(batch_size=batch_size, generator=generator, init=init, num_inference_steps=num_inference_steps,
                               start_from=start_from, save_every_step=False, eta=eta, return_tensor=True)
        img = self.vqvae.decode(lat).sample
        if return_tensor:
            return img
        movie = []
        if save_every_step or init is not None:
            first = init if init is not None else lat
            movie = [_post(self.vqvae.decode(first.to(self.device).float()).sample), _post(img)]
        images = _post(img)
        if output_type == "pil":
            from PIL import Image
            images = [Image.fromarray(im.squeeze()) for im in (images * 255).round().astype("uint8")]
        return SimpleNamespace(images=images, movie=movie)


class PNDMPipeline(DiffusionPipeline):
    """The fork's generic loop used for DPM-Solver / UniPC / ... (model.py:620-652): step(eps, t, x) only."""
    _class_name = "PNDMPipeline"
    default_steps = 50

    def _step_kwargs(self, generator, eta):
        return {}
