"""The reference's default generator configuration as Python objects (it ships it as Hydra YAML:
configs/dataset/generator/default.yaml:58-142 for the SR-artifact stages).  Used by the tests, `bench.py` and the tools;
a Hydra user instantiates the same classes from the YAML through `fetalsyngen_amd.compat.install()`."""
from __future__ import annotations


def default_artifacts(prob=0.4, merge_type="perlin"):
    """The four SR-artifact stages with the reference's default YAML values
    (configs/dataset/generator/default.yaml:58-142); `prob` replaces the stage gates (0.4 there)."""
    from .artifacts.utils import ReconMergeParams, ReconParams, ScannerParams, StructNoiseMergeParams
    from .augmentation.artifacts import BlurCortex, SimulatedBoundaries, SimulateMotion, StructNoise

    sn_merge = StructNoiseMergeParams(merge_type=merge_type, gauss_nloc_min=5, gauss_nloc_max=15, gauss_sigma_mu=25,
                                      gauss_sigma_std=5, perlin_res_list=[1, 2], perlin_octaves_list=[1, 2, 4],
                                      perlin_persistence=0.5, perlin_lacunarity=2, perlin_increase_size=0.1)
    scanner = ScannerParams(resolution_slice_fac_min=0.5, resolution_slice_fac_max=2, resolution_slice_max=1.5,
                            slice_thickness_min=1.5, slice_thickness_max=3.5, gap_min=1.5, gap_max=5.5, min_num_stack=2,
                            max_num_stack=6, max_num_slices=250, noise_sigma_min=0, noise_sigma_max=0.1, TR_min=1, TR_max=2,
                            prob_void=0.2, prob_gamma=0.1, gamma_std=0.05, slice_size=None, restrict_transform=False, txy=3.0)
    recon = ReconParams(prob_misreg_slice=0.1, slices_misreg_ratio=0.1, prob_misreg_stack=0.1, txy=3.0, prob_merge=1.0,
                        merge_params=ReconMergeParams(merge_type=merge_type, perlin_res_list=[1, 2], perlin_octaves_list=[1, 2, 4],
                                                      perlin_persistence=0.5, perlin_lacunarity=2, gauss_ngaussians_min=2,
                                                      gauss_ngaussians_max=4, perlin_increase_size=0.25),
                        prob_smooth=0.2, prob_rm_slices=0.3, rm_slices_min=0.1, rm_slices_max=0.4)
    return dict(
        blur_cortex=BlurCortex(prob=prob, cortex_label=2, nblur_min=50, nblur_max=200),
        struct_noise=StructNoise(prob=prob, wm_label=3, std_min=0.2, std_max=0.4, merge_params=sn_merge, nstages_min=1,
                                 nstages_max=5),
        simulate_motion=SimulateMotion(prob=prob, scanner_params=scanner, recon_params=recon),
        boundaries=SimulatedBoundaries(prob_no_mask=0.5 if prob < 1 else 0.0, prob_if_mask_halo=0.5, prob_if_mask_fuzzy=0.5),
    )
