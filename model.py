"""Drop-in module name of the reference (`from model import DiffuserModelSched`, reference model.py:529) -- the
implementation lives in villandiffusion_amd/model.py."""
from villandiffusion_amd.model import DDPM_32_ARCH, DiffuserModelSched  # noqa: F401
from villandiffusion_amd.pipelines import DDIMPipeline, DDPMPipeline, DiffusionPipeline, PNDMPipeline  # noqa: F401
from villandiffusion_amd.schedulers import (DDIMScheduler, DDPMScheduler, DPMSolverMultistepScheduler,  # noqa: F401
                                            UniPCMultistepScheduler)
from villandiffusion_amd.unet import UNet2DModel  # noqa: F401
from villandiffusion_amd.sampling_io import batch_sampling, batch_sampling_save, save_imgs  # noqa: F401
