"""Mirror of `fetalsyngen.generator.artifacts.svort` (reference svort/__init__.py:1-10)."""
from .rigid import (RigidTransform, ax_update_resolution, axisangle2mat, init_stack_transform, init_zero_transform,  # noqa: F401
                    mat2axisangle, mat_transform_points, mat_update_resolution, random_angle,
                    random_init_stack_transforms, reset_transform, transform_points)
from .scan import (get_PSF, get_trajectory, interleave_index, random_stack, resolution2sigma, sample_motion,  # noqa: F401
                   set_trajectory_bank, synthetic_trajectory_bank)
from .slice_acq import get_semantics, set_semantics, slice_acquisition, slice_acquisition_adjoint  # noqa: F401
