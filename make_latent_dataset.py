"""Precompute the VQ-VAE latents of a (poisoned) dataset -- the reference's make_latent_dataset.py:
the LDM fine-tune consumes `LatentDataset` items and passes vae=None to the loss (VillanDiffusion.py:1159).

    python make_latent_dataset.py --dataset CELEBA-HQ --ckpt <dir with unet/ vqvae/ scheduler/> --trigger GLASSES --target CAT \
        --out celeba_hq_256_latents [--batch 32] [--limit N]

Layout written (dataset.LatentDataset): <out>/target..pt {target type -> latent}, <out>/raw/<i>..pt, <out>/<trigger>/<i>..pt.
The encoder runs on the GPU through the HIP VQModel (vae.encode(x).latents, unquantised)."""
import argparse

import torch

from dataset import Backdoor, DatasetLoader, LatentDataset
from model import DiffuserModelSched


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default=DatasetLoader.CELEBA_HQ)
    ap.add_argument("--ckpt", default=DiffuserModelSched.LDM_CELEBA_HQ_256)
    ap.add_argument("--trigger", default=Backdoor.TRIGGER_GLASSES)
    ap.add_argument("--target", default=Backdoor.TARGET_CAT)
    ap.add_argument("--out", default="celeba_hq_256_latents")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--limit", type=int, default=None)
    ap.add_argument("--dataset_path", default="datasets")
    a = ap.parse_args(argv)
    _, vae, _, _ = DiffuserModelSched.get_model_sched(ckpt=a.ckpt, sde_type=DiffuserModelSched.SDE_LDM)
    dsl = DatasetLoader(root=a.dataset_path, name=a.dataset, batch_size=a.batch, shuffle=False) \
        .set_poison(trigger_type=a.trigger, target_type=a.target, clean_rate=1.0, poison_rate=0.0).prepare_dataset(mode=DatasetLoader.MODE_NONE)
    lds = LatentDataset(a.out).set_vae(vae)
    lds.update_target_by_key(a.target, dsl.target)
    n = len(dsl) if a.limit is None else min(a.limit, len(dsl))
    for s in range(0, n, a.batch):
        ids = torch.arange(s, min(n, s + a.batch))
        batch = dsl.make_batch(ids.to(vae.device), flip_bits=torch.zeros(len(ids), dtype=torch.bool, device=vae.device))
        imgs = batch[DatasetLoader.IMAGE]
        lds.update_data_by_idxs(LatentDataset.RAW_LATENTS_FILE_NAME, ids.tolist(), imgs)
        lds.update_data_by_idxs(a.trigger, ids.tolist(), dsl.get_poisoned(imgs))
    print(f"wrote {n} raw + {n} poisoned latents and target '{a.target}' to {a.out}")


if __name__ == "__main__":
    main()
