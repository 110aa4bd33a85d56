"""fetalsyngen_amd -- MI355X (gfx950) implementation of FetalSynthGen's per-volume synthesis path.

Drop-in surface (same names / signatures as the reference's `fetalsyngen` package):
    fetalsyngen_amd.generator.model.FetalSynthGen
    fetalsyngen_amd.generator.intensity.rand_gmm.ImageFromSeeds
    fetalsyngen_amd.generator.deformation.affine_nonrigid.SpatialDeformation
    fetalsyngen_amd.generator.augmentation.synthseg.{RandResample,RandBiasField,RandNoise,RandGamma}
    fetalsyngen_amd.generator.augmentation.artifacts.{BlurCortex,StructNoise,SimulateMotion,SimulatedBoundaries}
    fetalsyngen_amd.generator.artifacts.{utils,simulate_reco,svort}   (Scanner, PSFReconstructor, slice_acquisition, ...)
    fetalsyngen_amd.data.datasets.FetalSynthDataset
    fetalsyngen_amd.utils.generation.{make_affine_matrix,make_gaussian_kernel,gaussian_blur_3d,
                                      fast_3D_interp_torch,myzoom_torch}
`fetalsyngen_amd.compat.install()` registers these under the `fetalsyngen.*` module paths so existing
Hydra `_target_` strings resolve to this package.

All arithmetic on volumes runs in hand-written HIP kernels (libfsg_hip.so, C ABI in
include/fsg_hip.h); PyTorch is used for device memory, streams and host-side RNG only.
"""
from .rng import get_mode as get_rng_mode, set_mode as set_rng_mode  # noqa: F401

__version__ = "0.1.0"

import os as _os

if not _os.environ.get("FSG_KEEP_TORCH_THREADS"):
    # torch's CPU pool sized by the machine inside a container's CPU share stalls the host side for ~90 ms at a time
    # (hostenv.py); only ever lowers the count
    from .hostenv import cap_host_threads as _cap

    _cap()


def build(force: bool = False):
    """Compile libfsg_hip.so in-tree (hipcc, --offload-arch=gfx950)."""
    from ._build import build as _b

    return _b(force=force)
