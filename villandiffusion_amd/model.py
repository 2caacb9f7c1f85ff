"""``DiffuserModelSched`` -- the reference's model/scheduler factory (model.py:529-932) over the MI355X-native UNet,
samplers and pipelines.  Same class constants, same ``get_model_sched(...) -> (model, vae, noise_sched, get_pipeline)``
and ``get_pretrained`` signatures, same errors (NotImplementedError for an unknown sampler / SDE, ValueError when a
from-scratch model lacks size/channels).

Differences, all forced by the environment or scope: hub ids (``google/ddpm-cifar10-32`` ...) resolve
only to LOCAL diffusers-layout directories (no network): set ``VILLAN_CKPT_ROOT`` or pass a directory; the
``*-DEFAULT`` / from-scratch ids build the architecture of model.py:816-834 with torch-default init.  SDE-VE / SDE-LDM
model families (NCSN++, VQ-VAE) are "next" rows: the VE *loss* and tables are implemented, the NCSN++ network is not.
"""
from __future__ import annotations

import os
from functools import partial
from typing import Optional

import torch

from .pipelines import (DDIMPipeline, DDPMPipeline, DiffusionPipeline, KarrasVePipeline, LDMPipeline, PNDMPipeline,
                        ScoreSdeVePipeline)
from .schedulers import (DDIMScheduler, DDPMScheduler, DEISMultistepScheduler, DPMSolverMultistepScheduler,
                         HeunDiscreteScheduler, KarrasVeScheduler, LMSDiscreteScheduler, PNDMScheduler, ScoreSdeVeScheduler,
                         UniPCMultistepScheduler)
from .ncsnpp import NCSNppModel
from .unet import UNet2DModel
from .vqmodel import VQModel

# model.py:816-834
DDPM_32_ARCH = dict(act_fn="silu", attention_head_dim=None, block_out_channels=[128, 256, 256, 256], center_input_sample=False,
                    down_block_types=["DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"], downsample_padding=0,
                    flip_sin_to_cos=False, freq_shift=1, layers_per_block=2, mid_block_scale_factor=1, norm_eps=1e-06,
                    norm_num_groups=32, time_embedding_type="positional",
                    up_block_types=["UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D"])


# google/ddpm-ema-{celebahq,church,bedroom}-256 (113 673 219 parameters)
DDPM_256_ARCH = dict(DDPM_32_ARCH, block_out_channels=[128, 128, 256, 256, 512, 512],
                     down_block_types=["DownBlock2D"] * 4 + ["AttnDownBlock2D", "DownBlock2D"],
                     up_block_types=["UpBlock2D", "AttnUpBlock2D"] + ["UpBlock2D"] * 4)
# CompVis/ldm-celebahq-256: unet (274 056 163 parameters) on 3x64x64 latents + vqvae (55 322 782); [UPSTREAM config, from memory]
LDM_CELEBA_UNET_ARCH = dict(act_fn="silu", attention_head_dim=32, block_out_channels=[224, 448, 672, 896], center_input_sample=False,
                            down_block_types=["DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"],
                            downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, in_channels=3, layers_per_block=2,
                            mid_block_scale_factor=1, norm_eps=1e-05, norm_num_groups=32, out_channels=3, sample_size=64,
                            time_embedding_type="positional",
                            up_block_types=["AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"])
LDM_CELEBA_VQ_ARCH = dict(act_fn="silu", block_out_channels=[128, 256, 512], down_block_types=["DownEncoderBlock2D"] * 3,
                          in_channels=3, latent_channels=3, layers_per_block=2, num_vq_embeddings=8192, out_channels=3,
                          sample_size=256, up_block_types=["UpDecoderBlock2D"] * 3)


# model.py:839-857 / 876-894: the NCSN++ architecture of the NCSNPP-*-DEFAULT ids (61 894 924 parameters at 32x32)
NCSNPP_32_ARCH = dict(act_fn="silu", attention_head_dim=None, block_out_channels=[128, 256, 256, 256], center_input_sample=False,
                      down_block_types=["SkipDownBlock2D", "AttnSkipDownBlock2D", "SkipDownBlock2D", "SkipDownBlock2D"],
                      downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, layers_per_block=4,
                      mid_block_scale_factor=1.41421356237, norm_eps=1e-06, norm_num_groups=None, time_embedding_type="fourier",
                      up_block_types=["SkipUpBlock2D", "SkipUpBlock2D", "AttnSkipUpBlock2D", "SkipUpBlock2D"])


class DiffuserModelSched:
    LR_SCHED_CKPT, OPTIM_CKPT = "lr_sched.pth", "optim.pth"
    SDE_VP, SDE_VE, SDE_LDM = "SDE-VP", "SDE-VE", "SDE-LDM"
    CLIP_SAMPLE_DEFAULT = False
    MODEL_DEFAULT = "DEFAULT"
    DDPM_32_DEFAULT, DDPM_256_DEFAULT = "DDPM-32-DEFAULT", "DDPM-256-DEFAULT"
    NCSNPP_32_DEFAULT, NCSNPP_256_DEFAULT = "NCSNPP-32-DEFAULT", "NCSNPP-256-DEFAULT"
    DDPM_CIFAR10_DEFAULT, DDPM_CELEBA_HQ_DEFAULT = "DDPM-CIFAR10-DEFAULT", "DDPM-CELEBA-HQ-DEFAULT"
    DDPM_CHURCH_DEFAULT, DDPM_BEDROOM_DEFAULT = "DDPM-CHURCH-DEFAULT", "DDPM-BEDROOM-DEFAULT"
    LDM_CELEBA_HQ_DEFAULT = "LDM-CELEBA-HQ-DEFAULT"
    NCSNPP_CIFAR10_DEFAULT, NCSNPP_CELEBA_HQ_DEFAULT, NCSNPP_CHURCH_DEFAULT = \
        "NCSNPP-CIFAR10-DEFAULT", "NCSNPP-CELEBA-HQ-DEFAULT", "NCSNPP-CHURCH-DEFAULT"
    DDPM_CIFAR10_32, DDPM_CELEBA_HQ_256, DDPM_CHURCH_256, DDPM_BEDROOM_256 = \
        "DDPM-CIFAR10-32", "DDPM-CELEBA-HQ-256", "DDPM-CHURCH-256", "DDPM-BEDROOM-256"
    LDM_CELEBA_HQ_256 = "LDM-CELEBA-HQ-256"
    NCSNPP_CIFAR10_32, NCSNPP_CELEBA_HQ_256, NCSNPP_CHURCH_256 = "NCSNPP-CIFAR10-32", "NCSNPP-CELEBA-HQ-256", "NCSNPP-CHURCH-256"

    DDPM_SCHED, DDIM_SCHED = "DDPM-SCHED", "DDIM-SCHED"
    DPM_SOLVER_PP_O1_SCHED, DPM_SOLVER_O1_SCHED = "DPM_SOLVER_PP_O1-SCHED", "DPM_SOLVER_O1-SCHED"
    DPM_SOLVER_PP_O2_SCHED, DPM_SOLVER_O2_SCHED = "DPM_SOLVER_PP_O2-SCHED", "DPM_SOLVER_O2-SCHED"
    DPM_SOLVER_PP_O3_SCHED, DPM_SOLVER_O3_SCHED = "DPM_SOLVER_PP_O3-SCHED", "DPM_SOLVER_O3-SCHED"
    UNIPC_SCHED, PNDM_SCHED, DEIS_SCHED, HEUN_SCHED, LMSD_SCHED, LDM_SCHED = \
        "UNIPC-SCHED", "PNDM-SCHED", "DEIS-SCHED", "HEUN-SCHED", "LMSD-SCHED", "LDM-SCHED"
    SCORE_SDE_VE_SCHED, EDM_VE_SCHED, EDM_VE_ODE_SCHED, EDM_VE_SDE_SCHED = \
        "SCORE-SDE-VE-SCHED", "EDM-VE-SCHED", "EDM-VE-ODE-SCHED", "EDM-VE-SDE-SCHED"

    HUB_IDS = {DDPM_CIFAR10_32: "google/ddpm-cifar10-32", DDPM_CELEBA_HQ_256: "google/ddpm-ema-celebahq-256",
               DDPM_CHURCH_256: "google/ddpm-ema-church-256", DDPM_BEDROOM_256: "google/ddpm-ema-bedroom-256",
               LDM_CELEBA_HQ_256: "CompVis/ldm-celebahq-256", NCSNPP_CIFAR10_32: "fusing/cifar10-ncsnpp-ve",
               NCSNPP_CELEBA_HQ_256: "google/ncsnpp-celebahq-256", NCSNPP_CHURCH_256: "google/ncsnpp-church-256"}

    @staticmethod
    def get_sample_clip(clip_sample: bool, clip_sample_default: bool):
        return clip_sample if clip_sample is not None else clip_sample_default

    @staticmethod
    def _pipeline_factory(pipeline_cls):
        def get_pipeline(accelerate, unet, vae, scheduler):
            unwrap = getattr(accelerate, "unwrap_model", None)
            unet = unwrap(unet) if unwrap else getattr(unet, "module", unet)
            if vae is not None:
                return pipeline_cls(vqvae=vae, unet=unet, scheduler=scheduler)
            return pipeline_cls(unet=unet, scheduler=scheduler)
        return get_pipeline

    # sampler table: --sched -> (scheduler ctor kwargs, pipeline family)
    @classmethod
    def _make_sched(cls, noise_sched_type, clip, clip_range, beta):
        dpm = lambda order, algo: (partial(DPMSolverMultistepScheduler, solver_order=order, algorithm_type=algo, **beta), "pndm")
        table = {
            cls.DDPM_SCHED: (partial(DDPMScheduler, clip_sample=clip, **beta), "ddpm"),
            cls.DDIM_SCHED: (partial(DDIMScheduler, clip_sample=clip, **beta), "ddim"),
            cls.DPM_SOLVER_PP_O1_SCHED: dpm(1, "dpmsolver++"), cls.DPM_SOLVER_O1_SCHED: dpm(1, "dpmsolver"),
            cls.DPM_SOLVER_PP_O2_SCHED: dpm(2, "dpmsolver++"), cls.DPM_SOLVER_O2_SCHED: dpm(2, "dpmsolver"),
            cls.DPM_SOLVER_PP_O3_SCHED: dpm(3, "dpmsolver++"), cls.DPM_SOLVER_O3_SCHED: dpm(3, "dpmsolver"),
            cls.UNIPC_SCHED: (partial(UniPCMultistepScheduler, **beta), "pndm"),
            cls.PNDM_SCHED: (partial(PNDMScheduler, **beta), "pndm"),                      # model.py:641-652
            cls.DEIS_SCHED: (partial(DEISMultistepScheduler, **beta), "pndm"),
            cls.HEUN_SCHED: (partial(HeunDiscreteScheduler, **beta), "pndm"),
            cls.LMSD_SCHED: (partial(LMSDiscreteScheduler, **beta), "pndm"),
        }
        if noise_sched_type not in table:
            raise NotImplementedError()
        ctor, fam = table[noise_sched_type]
        pipe = {"ddpm": DDPMPipeline, "ddim": DDIMPipeline,
                "pndm": partial(PNDMPipeline, clip_sample=clip, clip_sample_range=clip_range)}[fam]
        return ctor(), pipe

    @classmethod
    def _resolve_dir(cls, ckpt_id: str) -> Optional[str]:
        if os.path.isdir(ckpt_id):
            return ckpt_id
        root = os.environ.get("VILLAN_CKPT_ROOT")
        if root:
            for cand in (os.path.join(root, ckpt_id), os.path.join(root, ckpt_id.split("/")[-1])):
                if os.path.isdir(cand):
                    return cand
        return None

    @classmethod
    def _get_model_sched_vp(cls, ckpt_id, clip_sample, noise_sched_type=None, clip_sample_range=None, build_model=True):
        clip = cls.get_sample_clip(clip_sample, cls.CLIP_SAMPLE_DEFAULT)
        clip_range = 1.0 if clip_sample_range is None else clip_sample_range
        beta = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02)         # model.py:606-608
        model = None
        loaded_sched = None
        if build_model:
            d = cls._resolve_dir(ckpt_id)
            if d is None:
                raise FileNotFoundError(
                    f"pretrained checkpoint '{ckpt_id}' is not available locally (no network / HF cache). Pass a "
                    f"diffusers-layout directory as --ckpt, set VILLAN_CKPT_ROOT, or use a from-scratch id such as "
                    f"'{cls.DDPM_32_DEFAULT}'.")
            pipe = DDPMPipeline.from_pretrained(d)
            model, loaded_sched = pipe.unet, pipe.scheduler
        if noise_sched_type is None:
            # model.py:654 keeps the checkpoint's own scheduler; for the from-scratch ids that is the scheduler of
            # google/ddpm-cifar10-32 (model.py:813), whose scheduler_config.json [UPSTREAM, from memory: not fetchable here] is
            # {linear 1e-4..0.02, T 1000, clip_sample true, variance_type "fixed_large"}
            noise_sched = loaded_sched if loaded_sched is not None else DDPMScheduler(clip_sample=True, variance_type="fixed_large", **beta)
            pipe_cls = DDPMPipeline
        else:
            noise_sched, pipe_cls = cls._make_sched(noise_sched_type, clip, clip_range, beta)
        if clip is not None:
            noise_sched.config.clip_sample = clip
        return model, None, noise_sched, cls._pipeline_factory(pipe_cls)

    @classmethod
    def _get_model_sched_ve(cls, ckpt_id, clip_sample, noise_sched_type=None, build_model=True):
        """model.py:668-703: VE-SDE with T=2000, sigma in [0.01, 380], snr 0.075, one corrector step."""
        model = None
        if build_model:
            d = cls._resolve_dir(ckpt_id)
            if d is None:
                raise FileNotFoundError(
                    f"pretrained SDE-VE checkpoint '{ckpt_id}' is not available locally (no network / HF cache). Pass a "
                    f"diffusers-layout directory as --ckpt, set VILLAN_CKPT_ROOT, or use a from-scratch id such as "
                    f"'{cls.NCSNPP_32_DEFAULT}'.")
            model = ScoreSdeVePipeline.from_pretrained(d).unet
        karras = {cls.EDM_VE_SCHED: {}, cls.EDM_VE_SDE_SCHED: {"s_churn": 100}, cls.EDM_VE_ODE_SCHED: {"s_churn": 0}}   # model.py:685-693
        if noise_sched_type in karras:
            sched = KarrasVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, **karras[noise_sched_type])
            pipe_cls = KarrasVePipeline
        elif noise_sched_type in (None, cls.SCORE_SDE_VE_SCHED):
            sched = ScoreSdeVeScheduler(num_train_timesteps=2000, sigma_min=0.01, sigma_max=380.0, sampling_eps=1e-05, correct_steps=1,
                                        snr=0.075)
            pipe_cls = ScoreSdeVePipeline
        else:
            raise NotImplementedError()
        clip = cls.get_sample_clip(clip_sample, cls.CLIP_SAMPLE_DEFAULT)
        if clip is not None:
            sched.config.clip_sample = clip
        return model, None, sched, cls._pipeline_factory(pipe_cls)

    @classmethod
    def _get_model_sched(cls, ckpt_id, clip_sample, clip_sample_range=None, noise_sched_type=None, sde_type=SDE_VP,
                         build_model=True):
        if sde_type == cls.SDE_VP:
            model, vae, sched, gp = cls._get_model_sched_vp(ckpt_id, clip_sample, noise_sched_type, clip_sample_range, build_model)
        elif sde_type == cls.SDE_VE:
            model, vae, sched, gp = cls._get_model_sched_ve(ckpt_id, clip_sample, noise_sched_type, build_model)
        elif sde_type == cls.SDE_LDM:
            model, vae, sched, gp = cls._get_model_sched_ldm(ckpt_id, clip_sample, noise_sched_type, build_model)
        else:
            raise NotImplementedError(f"sde_type {sde_type} not implemented")
        if model is not None:
            model.requires_grad_(True)
        if vae is not None:
            vae.requires_grad_(False)                                                # model.py:790
        return model, vae, sched, gp

    @classmethod
    def _get_model_sched_ldm(cls, ckpt_id, clip_sample, noise_sched_type=None, build_model=True):
        """model.py:706-776: latent diffusion = UNet on VQ-VAE latents; beta scaled_linear [0.0015, 0.0195], T=1000."""
        clip = cls.get_sample_clip(clip_sample, cls.CLIP_SAMPLE_DEFAULT)
        beta = dict(num_train_timesteps=1000, beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear")
        model = vae = loaded_sched = None
        if build_model:
            d = cls._resolve_dir(ckpt_id)
            if d is None:
                raise FileNotFoundError(
                    f"pretrained latent-diffusion checkpoint '{ckpt_id}' is not available locally (no network / HF cache). Pass a "
                    f"diffusers-layout directory (unet/ vqvae/ scheduler/) as --ckpt or set VILLAN_CKPT_ROOT.")
            pipe = DiffusionPipeline.from_pretrained(d)
            if pipe.vqvae is None:
                raise ValueError(f"{d}: no vqvae/ folder -- not a latent-diffusion checkpoint")
            model, vae, loaded_sched = pipe.unet, pipe.vqvae, pipe.scheduler
        ldm_clip = partial(LDMPipeline, clip_sample=clip)
        if noise_sched_type is None:
            noise_sched = loaded_sched if loaded_sched is not None else DDIMScheduler(clip_sample=False, **beta)
            pipe_cls = LDMPipeline
        elif noise_sched_type == cls.DDPM_SCHED:
            noise_sched, pipe_cls = DDPMScheduler(clip_sample=clip, **beta), LDMPipeline
        elif noise_sched_type == cls.DDIM_SCHED:
            noise_sched, pipe_cls = DDIMScheduler(clip_sample=clip, **beta), LDMPipeline
        else:
            noise_sched, _ = cls._make_sched(noise_sched_type, clip, 1.0, beta)
            pipe_cls = ldm_clip
        if clip is not None:
            noise_sched.config.clip_sample = clip
        return model, vae, noise_sched, cls._pipeline_factory(pipe_cls)

    @staticmethod
    def check_image_size_channel(image_size: int, channels: int):
        if image_size is None or channels is None:
            raise ValueError(f"Arguement image_size and channels shouldn't be {image_size} and {channels}")

    @classmethod
    def get_model_sched(cls, image_size: int = None, channels: int = None, ckpt: str = MODEL_DEFAULT, sde_type: str = SDE_VP,
                        clip_sample: bool = None, clip_sample_range: float = None, noise_sched_type: str = None, **kwargs):
        if ckpt in (cls.MODEL_DEFAULT, cls.DDPM_32_DEFAULT):
            cls.check_image_size_channel(image_size, channels)
            _, vae, sched, gp = cls._get_model_sched(cls.HUB_IDS[cls.DDPM_CIFAR10_32], clip_sample, clip_sample_range,
                                                     noise_sched_type, sde_type, build_model=False)
            model = UNet2DModel(in_channels=channels, out_channels=channels, sample_size=image_size, **DDPM_32_ARCH)
            model.requires_grad_(True)
            return model, vae, sched, gp
        scratch = {cls.DDPM_CIFAR10_DEFAULT: cls.DDPM_CIFAR10_32, cls.DDPM_CELEBA_HQ_DEFAULT: cls.DDPM_CELEBA_HQ_256,
                   cls.DDPM_CHURCH_DEFAULT: cls.DDPM_CHURCH_256, cls.DDPM_BEDROOM_DEFAULT: cls.DDPM_BEDROOM_256}
        if ckpt == cls.DDPM_CIFAR10_DEFAULT and cls._resolve_dir(cls.HUB_IDS[cls.DDPM_CIFAR10_32]) is None:
            # architecture of google/ddpm-cifar10-32 == model.py:816-834; weights are re-initialised anyway (weight_reset)
            return cls.get_model_sched(image_size=32, channels=3, ckpt=cls.DDPM_32_DEFAULT, sde_type=sde_type,
                                       clip_sample=clip_sample, clip_sample_range=clip_sample_range,
                                       noise_sched_type=noise_sched_type)
        if ckpt in scratch and cls._resolve_dir(cls.HUB_IDS[scratch[ckpt]]) is None:
            # weights are re-initialised anyway (weight_reset): build the published 256x256 architecture directly
            _, vae, sched, gp = cls._get_model_sched(cls.HUB_IDS[scratch[ckpt]], clip_sample, clip_sample_range, noise_sched_type,
                                                     sde_type, build_model=False)
            model = UNet2DModel(in_channels=3, out_channels=3, sample_size=256, **DDPM_256_ARCH)
            model.requires_grad_(True)
            return model, vae, sched, gp
        if ckpt in scratch:
            model, vae, sched, gp = cls.get_pretrained(scratch[ckpt], clip_sample, clip_sample_range, noise_sched_type, sde_type=sde_type)
            model.reset_parameters()
            return model, vae, sched, gp
        if ckpt == cls.LDM_CELEBA_HQ_DEFAULT:
            if cls._resolve_dir(cls.HUB_IDS[cls.LDM_CELEBA_HQ_256]) is not None:
                model, vae, sched, gp = cls.get_pretrained(cls.LDM_CELEBA_HQ_256, clip_sample, clip_sample_range, noise_sched_type,
                                                           sde_type=cls.SDE_LDM)
                model.reset_parameters()
                return model, vae, sched, gp
            # No local copy of CompVis/ldm-celebahq-256: the UNet is re-initialised anyway, but the VQ-VAE would be the
            # PRETRAINED one -- a random VQ-VAE is only good for plumbing / throughput runs, so say so loudly.
            import warnings
            warnings.warn("LDM-CELEBA-HQ-DEFAULT without a local CompVis/ldm-celebahq-256: the VQ-VAE is RANDOMLY initialised "
                          "(published architecture); decoded images are meaningless until real vqvae weights are loaded")
            _, _, sched, gp = cls._get_model_sched_ldm(cls.HUB_IDS[cls.LDM_CELEBA_HQ_256], clip_sample, noise_sched_type, build_model=False)
            model, vae = UNet2DModel(**LDM_CELEBA_UNET_ARCH), VQModel(**LDM_CELEBA_VQ_ARCH)
            model.requires_grad_(True)
            vae.requires_grad_(False)
            return model, vae, sched, gp
        if ckpt in (cls.NCSNPP_32_DEFAULT, cls.NCSNPP_CIFAR10_DEFAULT):                    # model.py:836-858, 876-898
            if ckpt == cls.NCSNPP_32_DEFAULT:
                cls.check_image_size_channel(image_size, channels)
            else:
                image_size, channels = 32, 3
            _, vae, sched, gp = cls._get_model_sched(cls.HUB_IDS[cls.NCSNPP_CELEBA_HQ_256], clip_sample, clip_sample_range,
                                                     noise_sched_type, sde_type, build_model=False)
            model = NCSNppModel(in_channels=channels, out_channels=channels, sample_size=image_size, **NCSNPP_32_ARCH)
            model.requires_grad_(True)
            model.time_proj.weight.requires_grad_(False)
            return model, vae, sched, gp
        if ckpt in (cls.NCSNPP_CELEBA_HQ_DEFAULT, cls.NCSNPP_CHURCH_DEFAULT):
            src = cls.NCSNPP_CELEBA_HQ_256 if ckpt == cls.NCSNPP_CELEBA_HQ_DEFAULT else cls.NCSNPP_CHURCH_256
            model, vae, sched, gp = cls.get_pretrained(src, clip_sample, clip_sample_range, noise_sched_type, sde_type=sde_type)
            model.reset_parameters()
            return model, vae, sched, gp
        return cls.get_pretrained(ckpt, clip_sample, clip_sample_range, noise_sched_type, sde_type=sde_type)

    @classmethod
    def get_pretrained(cls, ckpt: str, clip_sample: bool = None, clip_sample_range: float = None, noise_sched_type: str = None,
                       num_inference_steps: int = 1000, sde_type: str = SDE_VP):
        return cls._get_model_sched(cls.HUB_IDS.get(ckpt, ckpt), clip_sample, clip_sample_range, noise_sched_type, sde_type)

    @staticmethod
    def get_optim(ckpt: str, optim, lr_sched):
        lr_sched.load_state_dict(torch.load(DiffuserModelSched.LR_SCHED_CKPT, map_location="cpu"))
        optim.load_state_dict(torch.load(DiffuserModelSched.OPTIM_CKPT, map_location="cpu"))
        return optim, lr_sched
