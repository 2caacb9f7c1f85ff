"""Drop-in module name of the reference (`from fid_score import fid`, reference VillanDiffusion.py:340,348) -- the implementation
lives in villandiffusion_amd/fid_score.py (InceptionV3 pool3 activations on the HIP kernels)."""
from villandiffusion_amd.fid_score import (IMAGE_EXTENSIONS, calculate_activation_statistics, calculate_fid_given_paths,  # noqa: F401
                                           calculate_frechet_distance, compute_statistics_of_path, fid, get_activations)
from villandiffusion_amd.inception import InceptionV3  # noqa: F401
