"""Backdoor-removal fine-tune: the reference's rm_backdoor_VillanDiffusion.py is its training driver with
ONE changed line -- the regression target of the poisoned loss is the CLEAN image, `target_latent_key="image"`
(rm_backdoor_VillanDiffusion.py:1113) instead of the backdoor target (VillanDiffusion.py:1159) -- so a poisoned sample
x0=image, R=pixel_values teaches the network to ignore the trigger.  Same CLI as VillanDiffusion.py."""
import VillanDiffusion as _V

_V.TARGET_LATENT_KEY = "image"
main = _V.main

if __name__ == "__main__":
    main()
