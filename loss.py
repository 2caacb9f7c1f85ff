"""Drop-in module name of the reference (`from loss import LossFn`, reference loss.py:825) -- the implementation lives
in villandiffusion_amd/loss.py."""
from villandiffusion_amd.loss import (LossFn, get_hs_ve, get_hs_vp, get_R_coef_gen_ve_reduce, get_R_coef_gen_vp,  # noqa: F401
                                      get_ws_ve)
