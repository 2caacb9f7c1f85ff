"""CPU oracle for the backdoored-diffusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.

What it restates (plain fp32 ``torch`` ops on the CPU, i.e. exactly the
``conv2d / group_norm / silu / linear / softmax / bmm`` sequence the
reference's diffusers path executes on a CPU):

* ``unet_ref``       UNet2DModel forward             -- parity UNPINNED
* ``schedulers_ref`` DDPM/DDIM/DPM-Solver/UniPC/ScoreSDE-VE        -- parity UNPINNED
* ``loss_ref``       R-coefficient tables + LossFn (loss.py)       -- pinned by tests/golden
* ``backdoor_ref``   Backdoor triggers/targets/masks (dataset.py)  -- box types pinned by tests/golden

"parity unpinned": the UNet and sampler arithmetic of the reference lives in an
un-vendored third-party fork (FrankCCCCC/diffusers@3784fd43, requirement.txt:37)
that is absent from /root/reference and from this image; no golden vectors for
it exist upstream.  Those two modules restate the published upstream
algorithms and are checked by known-answer tests (parameter count, state-dict
key list, analytic identities between samplers, fp64 gradient checks).
"""
