"""One optimiser step of the poisoned fine-tune loop (T1/T2 in SURVEY.md §8a; reference VillanDiffusion.py:1141-1176
plus the accelerate semantics of SURVEY Appendix B), one process per GPU.

* backward of micro-step k accumulates into the model's flat gradient buffer; the loss gradient is pre-divided by the
  gradient-accumulation count G inside the MSE kernel (``accelerator.backward``);
* on a sync step: ONE RCCL all-reduce (SUM) of the flat fp32 gradient bucket over xGMI (replaces nn.DataParallel's
  per-step parameter broadcast + gradient reduce, VillanDiffusion.py:440), then ONE l2-norm kernel and ONE fused
  clip + Adam kernel over the flat param/grad/m/v buffers (``clip_grad_norm_(…, 1.0)`` + ``torch.optim.Adam``);
  the 1/world averaging and the clip coefficient are folded into the Adam kernel's gradient scale;
* LR = base_lr * cosine-with-warmup(sync_step_count), advanced only on sync steps (accelerate's scheduler wrapper).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist

from . import ops
from .schedulers import get_cosine_schedule_with_warmup_lambda


class FusedAdam:
    """torch.optim.Adam(betas=(0.9, 0.999), eps=1e-8, weight_decay=0) on one flat buffer, with the global-norm clip
    fused in.  ``state_dict`` is flat too (exp_avg / exp_avg_sq / step)."""

    def __init__(self, model, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, max_grad_norm: Optional[float] = 1.0):
        self.model = model
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, betas, eps, max_grad_norm
        self.exp_avg = torch.zeros_like(model.flat_param)
        self.exp_avg_sq = torch.zeros_like(model.flat_param)
        self.step_count = 0
        self._partial = torch.empty(1024, device=model.flat_param.device, dtype=torch.float32)
        self.grad_norm_sq = torch.zeros(1, device=model.flat_param.device, dtype=torch.float32)
        # steps the Adam kernel skipped because the gradient norm was not finite (device word, bumped by the kernel; read lazily by Trainer)
        self.skipped = torch.zeros(1, device=model.flat_param.device, dtype=torch.int32)

    def step(self, lr: Optional[float] = None, grad_inv_scale: float = 1.0, need_norm: bool = False):
        m = self.model
        self.step_count += 1
        nsq = None
        if self.max_grad_norm is not None or need_norm:          # need_norm: the overflow guard of the f16 mode reads it (no clipping: max_norm = inf)
            ops.l2norm_sq(m.flat_grad, self._partial, self.grad_norm_sq)
            nsq = self.grad_norm_sq
        ops.adam_step(m.flat_param, m.flat_grad, self.exp_avg, self.exp_avg_sq, nsq,
                      float(self.max_grad_norm if self.max_grad_norm is not None else 3.0e38), grad_inv_scale, self.lr if lr is None else lr, self.betas[0],
                      self.betas[1], self.eps, self.step_count, skipped=self.skipped if nsq is not None else None)

    def grad_norm(self, grad_inv_scale: float = 1.0) -> float:
        """Global L2 norm of the (averaged) gradient of the last step -- synchronises; for logging/tests only."""
        return math.sqrt(float(self.grad_norm_sq)) * grad_inv_scale

    def state_dict(self) -> Dict:
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count, "lr": self.lr}

    def load_state_dict(self, sd: Dict):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.lr = float(sd.get("lr", self.lr))


def allreduce_flat_grad(flat_grad: torch.Tensor, n_buckets: int = 4, even_alone: bool = False):
    """SUM all-reduce of the flat gradient in a few large buckets (RCCL over xGMI: ring all-reduce is per-link bound,
    so few large messages; backend "nccl" IS RCCL on ROCm, "gloo" in the CPU tests)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not even_alone):
        return
    n = flat_grad.numel()
    per = (n + n_buckets - 1) // n_buckets
    per = (per + 1023) // 1024 * 1024
    works = []
    for s in range(0, n, per):
        works.append(dist.all_reduce(flat_grad[s:min(n, s + per)], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()


def shard_indices(n_items: int, epoch: int, rank: int, world: int, seed: int = 0, shuffle: bool = True) -> torch.Tensor:
    """DistributedSampler-style deterministic sharding: a seeded epoch permutation, padded to a multiple of `world`,
    rank r takes the contiguous slice r.  Poison flags travel with the sample index."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        perm = torch.randperm(n_items, generator=g)
    else:
        perm = torch.arange(n_items)
    total = (n_items + world - 1) // world * world
    if total > n_items:
        perm = torch.cat([perm, perm[: total - n_items]])
    per = total // world
    return perm[rank * per:(rank + 1) * per]


_DEBUG = bool(__import__("os").environ.get("VD_GRAPH_DEBUG"))


class GraphedMicroStep:
    """q-sample -> UNet forward -> loss (+ its gradient) -> UNet backward of ONE micro-batch, captured once into a HIP graph per batch shape
    and replayed (BASELINE config #1: `--batch 4` with gradient accumulation 32, VillanDiffusion.py:287 -- a micro-step is ~470 launches of
    5-20 us kernels, so eager launch is host-bound: the GPU idles between them).  The graph calls the network's explicit launch sequences
    directly (no autograd tape inside the capture), accumulates into the flat gradient buffer like `loss.backward()` does and leaves the
    un-divided micro-batch loss in `self.loss`.  Static inputs: clean image, poison residual, noise, timesteps.  Single-process VP / LDM
    training only (the bucketed all-reduce hooks of a multi-rank step fire from inside the backward pass and are not captured)."""

    def __init__(self, trainer, x0, R, noise, t):
        import torch.cuda
        net, lf = trainer.model, trainer.loss_fn
        self.trainer, self.net = trainer, net
        dev = net.device
        self.x0, self.R, self.noise = (torch.zeros_like(v, device=dev, dtype=torch.float32) for v in (x0, R, noise))
        self.t = torch.zeros(t.shape, device=dev, dtype=torch.int64)
        self.tf = torch.zeros(t.shape, device=dev, dtype=torch.float32)
        self.loss = torch.zeros(1, device=dev, dtype=torch.float32)
        self.key = self._key()
        for buf, v in ((self.x0, x0), (self.R, R), (self.noise, noise), (self.t, t)):
            buf.copy_(v)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        keep = net.flat_grad.clone()
        with torch.cuda.stream(side):                     # warm-up off the capture: workspaces, job tables, packed operands, allocator pools
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if _DEBUG:
            print("[graph] warm-up done", flush=True)
        net.flat_grad.copy_(keep)                         # the warm-up passes accumulated gradients: undo
        self._keep = (ops.gemm_ws_buffer(dev, 0), ops._WG_WS.get(dev), getattr(net, "_packed", None), net.wgrad_ws, getattr(net, "_packed16", None),
                      ops.gemm_ws_buffer(dev, ops.AUX_WS_SLOT))      # (the auxiliary stream's workspace: its address is baked into the capture too)
        self.graph = torch.cuda.CUDAGraph()
        ops.capture_begin(dev)                            # job tables built during the capture are uploaded right after it (ops.upload_table)
        try:
            with torch.cuda.graph(self.graph):
                self._body()
        finally:
            self._tables = ops.capture_end()              # device job tables the captured launches read: owned by this graph from now on
        if _DEBUG:
            torch.cuda.synchronize(dev)
            print(f"[graph] captured; {len(self._tables)} job tables", flush=True)

    def _key(self):
        n = self.net
        return (n.conv_math, n.wgrad_stream, n.group_wgrad, n.flat_param.data_ptr(), n.flat_grad.data_ptr(), self.trainer.loss_fn.grad_scale)

    def valid(self) -> bool:
        dev = self.net.device
        return (self.key == self._key() and self._keep[0] is ops.gemm_ws_buffer(dev, 0) and self._keep[1] is ops._WG_WS.get(dev)
                and self._keep[3] is self.net.wgrad_ws and self._keep[5] is ops.gemm_ws_buffer(dev, ops.AUX_WS_SLOT))

    def _body(self):
        net, lf = self.net, self.trainer.loss_fn
        x_t, y = lf.get_inputs_targets(self.x0, self.R, self.t, self.noise)
        self.tf.copy_(self.t)                             # int64 -> float32 timesteps, as UNet2DModel.forward does
        pred, st = net._run_forward(x_t, self.tf, save=True)
        dpred = torch.empty_like(pred)
        ops.mse_fwd_bwd(pred, y, dpred, self.loss, lf._partial, pscale=None, gscale=lf.grad_scale, kind=lf._loss_type)
        net._run_backward(st, dpred)

    def __call__(self, x0, R, noise, t):
        net = self.net
        if hasattr(net, "refresh_packed"):                # weights changed since the last replay: rebuild the packed operands (one launch each)
            net.refresh_packed(False)
            net.refresh_packed(True)
        elif getattr(net, "_packed", None) is not None:
            net._packed.refresh(False)
            net._packed.refresh(True)
        self.x0.copy_(x0)
        self.R.copy_(R)
        self.noise.copy_(noise)
        self.t.copy_(t)
        self.graph.replay()
        if _DEBUG:
            torch.cuda.synchronize(net.device)
            print("[graph] replayed", flush=True)
        return self.loss[0].clone()       # the static buffer is overwritten by the next replay: callers keep losses across micro-steps


class Trainer:
    def __init__(self, model, loss_fn, lr: float, total_steps: int, warmup_steps: int = 500, grad_accum: int = 1,
                 max_grad_norm: Optional[float] = 1.0, n_allreduce_buckets: int = 4, graph_micro_step: Optional[bool] = None,
                 force_ddp_path: bool = False):
        self.model, self.loss_fn = model, loss_fn
        self.base_lr = lr
        self.opt = FusedAdam(model, lr, max_grad_norm=max_grad_norm)
        self.lr_lambda: Callable[[int], float] = get_cosine_schedule_with_warmup_lambda(warmup_steps, total_steps)
        self.grad_accum = max(1, int(grad_accum))
        self.micro = 0
        self.sched_step = 0                       # LambdaLR.last_epoch
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.n_buckets = n_allreduce_buckets
        # f16 mixed-precision mode (model.conv_math == "f16", opt-in): the loss gradient is multiplied by `loss_scale` (a power of two: exact in
        # f32) so that the input gradients survive their conversion to f16 operands, and divided out again inside the Adam kernel; a step whose
        # gradient is not finite is skipped by the kernel and the scale halved at the next check (GradScaler semantics with a lazy host check:
        # the norm is read back every `scale_check_every` optimiser steps, doubled again after `scale_growth_every` clean ones)
        self.loss_scale = float(2 ** 12)
        self.scale_check_every, self.scale_growth_every = 100, 2000
        self._since_growth = 0
        self.overflow_steps_seen = 0
        self._skipped_seen = 0                    # value of opt.skipped at the last lazy check
        self._recheck = False                     # the last lazy check found skipped steps: read the counter after every step until one passes clean
        self.loss_fn.grad_scale = self._scale() / self.grad_accum
        self._pending = []                        # async all-reduce handles of the current optimiser step
        self._sync_now = False
        # HIP-graph replay of the micro-step: on by default where a step is launch-bound (gradient accumulation = small micro-batches), single process
        self.graph_micro_step = (self.grad_accum > 1) if graph_micro_step is None else bool(graph_micro_step)
        self._graphs: Dict = {}
        # force_ddp_path: the schedule of a multi-rank step (bucket hooks fired from the explicit backward, eager micro-step, one all-reduce per
        # bucket) on a process group of ANY size, including 1 -- what rank 0 of an 8-GPU run executes minus the wire time (bench.py --ddp-path)
        self.ddp_path = (self.world > 1 or (force_ddp_path and dist.is_available() and dist.is_initialized())) and hasattr(model, "grad_buckets")
        if self.ddp_path:
            model.bucket_ready_hook = self._bucket_ready      # overlap the all-reduce with the rest of backward
        model.zero_grad()

    def _bucket_ready(self, i: int):
        """Called from the UNet's explicit backward when gradient bucket i is final (order: up|out, mid, down, temb)."""
        if not self._sync_now:
            return
        s, e = self.model.grad_buckets[i]
        self._pending.append(dist.all_reduce(self.model.flat_grad[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def _scale(self) -> float:
        return self.loss_scale if getattr(self.model, "conv_math", None) == "f16" else 1.0

    def check_skipped(self, force: bool = False, count_step: bool = True):
        """Lazy finite-gradient check, every arithmetic: the Adam kernel leaves the parameters alone when the gradient norm is not finite and
        counts the step on the device; that counter is read back every `scale_check_every` optimiser steps (and from `state_dict`).

        * f16 mode (GradScaler semantics, reference VillanDiffusion.py:260-264 -> accelerate): ANY skipped step since the last check halves the
          loss scale and restarts the growth interval; skipped steps are taken back out of Adam's bias-correction count and the LR schedule
          (GradScaler / accelerate skip both), at the check rather than at the step -- the price of not synchronising every step.  After a check
          that found skipped steps the counter is read after EVERY step until a step passes clean (a persistent overflow then costs one step
          per halving, as under GradScaler, not `scale_check_every` steps);
        * every other arithmetic has no scale to lower: a non-finite gradient norm is a broken run (NaN statistics out of a GroupNorm poll
          timeout, a diverged model, ...) and raises instead of training on silently.

        count_step: this call follows an optimiser step (train_step).  A forced read from `state_dict` passes False: it applies what the
        counter already holds (a checkpoint never records steps the kernel refused) but advances neither the growth interval nor the scale."""
        if count_step:
            self._since_growth += 1
        if not force and not self._recheck and self.sched_step % self.scale_check_every:
            return
        if self.opt.max_grad_norm is None and self._scale() == 1.0:
            return                                   # no norm kernel in this configuration: nothing was counted
        skipped = int(self.opt.skipped)
        new = skipped - self._skipped_seen
        self._skipped_seen = skipped
        if new > 0 and self._scale() == 1.0:
            raise FloatingPointError(f"{new} optimiser step(s) had a non-finite gradient norm and were skipped by the Adam kernel "
                                     f"(arithmetic {getattr(self.model, 'conv_math', '?')!r}; asynchronous kernel errors, e.g. GroupNorm poll "
                                     f"timeouts: {ops.L.load().vd_async_errors(0)}; last library error: {ops.L.last_error()!r})")
        self._recheck = new > 0
        if new > 0:
            self.loss_scale = max(1.0, self.loss_scale * 0.5)
            self.overflow_steps_seen += new
            self._since_growth = 0
            self.opt.step_count = max(0, self.opt.step_count - new)
            self.sched_step = max(0, self.sched_step - new)
        elif count_step and self._scale() != 1.0 and self._since_growth >= self.scale_growth_every:
            self.loss_scale = min(float(2 ** 24), self.loss_scale * 2.0)
            self._since_growth = 0

    _check_scale = check_skipped

    @property
    def lr(self) -> float:
        return self.base_lr * self.lr_lambda(self.sched_step)

    def train_step(self, batch, timesteps: torch.Tensor, noise: Optional[torch.Tensor] = None, last_batch: bool = False,
                   target_key: str = "target", poison_key: str = "pixel_values"):
        """One micro-step; returns the (un-divided) loss tensor of this micro-batch."""
        sync = (self.micro + 1) % self.grad_accum == 0 or last_batch   # accelerate: sync on every G-th and on the last batch
        self._sync_now = sync and self.model.bucket_ready_hook is not None
        if self.micro % self.grad_accum == 0:                          # first micro-step of an accumulation window (the scale never changes inside one)
            self._step_scale = self._scale()
        self.loss_fn.grad_scale = self._step_scale / self.grad_accum
        loss = self._graphed(batch, timesteps, noise, target_key, poison_key)
        if loss is None:
            loss = self.loss_fn.p_loss_by_keys(batch, self.model, target_latent_key=target_key, poison_latent_key=poison_key,
                                               timesteps=timesteps, noise=noise)
            if torch.is_tensor(loss):
                loss.backward()
        self.micro = 0 if last_batch else self.micro + 1          # accelerate `_do_sync`: the counter restarts at end_of_dataloader
        if sync:
            if self._sync_now and self._pending:
                for w in self._pending:                           # bucketed all-reduces were launched during backward
                    w.wait()
                self._pending = []
            else:
                allreduce_flat_grad(self.model.flat_grad, self.n_buckets, even_alone=self.ddp_path)
            self._sync_now = False
            self.opt.step(lr=self.lr, grad_inv_scale=1.0 / (self.world * self._step_scale), need_norm=self._step_scale != 1.0)
            self.sched_step += 1
            self._check_scale()
            self.model.zero_grad()
        return loss

    def _graphed(self, batch, timesteps, noise, target_key, poison_key):
        """The micro-step as a HIP-graph replay, or None when this step has to run eagerly."""
        m, lf = self.model, self.loss_fn
        if not self.graph_micro_step or self.ddp_path or not hasattr(m, "_run_backward") or getattr(lf, "_sde", None) == "SDE-VE":
            return None
        x0, R = batch[target_key], batch[poison_key]
        if len(x0) == 0 or getattr(m, "device", torch.device("cpu")).type != "cuda" or not hasattr(m, "_packed"):
            return None
        dev = m.device
        x0, R = x0.to(dev).float(), R.to(dev).float()
        if noise is None:
            noise = lf._noise(x0)
        lf._tables(dev)
        key = (tuple(x0.shape),)
        g = self._graphs.get(key)
        if g is None or not g.valid():
            self._graphs.clear()                          # one shape at a time: a graph's private pool pins the activations of a whole step
            g = self._graphs[key] = GraphedMicroStep(self, x0, R, noise.to(dev), timesteps.to(dev))
        return g(x0, R, noise.to(dev), timesteps.to(dev))

    def state_dict(self) -> Dict:
        if getattr(self.model, "device", torch.device("cpu")).type == "cuda":
            self.check_skipped(force=True, count_step=False)      # a checkpoint never records steps the kernel refused
        return {"optimizer": self.opt.state_dict(), "micro": self.micro, "sched_step": self.sched_step,
                "loss_scale": self.loss_scale, "since_growth": self._since_growth, "overflow_steps_seen": self.overflow_steps_seen}

    def load_state_dict(self, sd: Dict):
        self.opt.load_state_dict(sd["optimizer"])
        self.micro, self.sched_step = int(sd["micro"]), int(sd["sched_step"])
        self.loss_scale = float(sd.get("loss_scale", self.loss_scale))          # (absent from checkpoints written before round 5)
        self._since_growth = int(sd.get("since_growth", 0))
        self.overflow_steps_seen = int(sd.get("overflow_steps_seen", 0))
        self._skipped_seen = int(self.opt.skipped)
