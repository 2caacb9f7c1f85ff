"""MI355X-native backdoored-diffusion hot path (IBM/VillanDiffusion's train + sample path on gfx950).

Layout:  csrc/ (HIP kernels + C ABI, include/villan_hip.h)  ·  lib/ops (ctypes binding)  ·  unet (UNet2DModel)  ·
schedulers / pipelines (samplers)  ·  loss (LossFn)  ·  dataset (Backdoor, DatasetLoader)  ·  model (DiffuserModelSched)
·  trainer (one-process-per-GPU data-parallel step).
"""
__version__ = "0.1.0"
