"""Spatial deformation: mirror of
`fetalsyngen.generator.deformation.affine_nonrigid.SpatialDeformation` (reference
affine_nonrigid.py:12-366) on the fused `fsg_coords_minmax_f32` / `fsg_warp_*` kernels.

Same constructor, same public methods and return tuples.  What differs underneath:
  * nothing of size H*W*D is built on the host or kept on `self` (the reference rebuilds three
    meshgrid volumes per call and uploads them, ref :64-84);
  * the nonlinear field is kept at its coarse size; `deform()` evaluates it inside the warp kernel;
  * `generate_deformation_and_flip` still returns materialised coordinate volumes for callers that
    want them (one `fsg_coords_f32` launch); `deform()` never materialises them.
"""
from __future__ import annotations

from typing import Iterable

import numpy as np
import torch

from ... import kernels as K
from ... import tables as T
from ...utils.generation import fast_3D_interp_torch, make_affine_matrix


class DeformPlan:
    """All random draws of one deformation, host side."""

    __slots__ = ("active", "flip", "A", "c2", "field_small", "params", "A_np", "c2_np")

    def __init__(self):
        self.active, self.flip, self.A, self.c2, self.field_small = False, False, None, None, None
        self.A_np, self.c2_np = None, None  # numpy twins of A (float32) / c2 (float64) when the planner has them at hand
        self.params = {"affine": None, "non_rigid": None, "flip": False}


class SpatialDeformation:
    def __init__(
        self,
        max_rotation: float,
        max_shear: float,
        max_scaling: float,
        size: Iterable[int],
        prob: float,
        nonlinear_transform: bool,
        nonlin_scale_min: float,
        nonlin_scale_max: float,
        nonlin_std_max: float,
        flip_prb: float,
        device: str,
    ):
        self.size = size
        self.prob = prob
        self.flip_prb = flip_prb
        self.max_rotation = max_rotation
        self.max_shear = max_shear
        self.max_scaling = max_scaling
        self.nonlinear_transform = nonlinear_transform
        self.nonlin_scale_min = nonlin_scale_min
        self.nonlin_scale_max = nonlin_scale_max
        self.nonlin_std_max = nonlin_std_max
        self.device = device

    def _shape_constants(self, shape3):
        """Per-shape constants of the plan (cached: the plan runs once per sample on the host's critical path)."""
        cache = self.__dict__.setdefault("_shape_cache", {})
        hit = cache.get(shape3)
        if hit is None:
            shp = np.array(shape3)
            centre32 = ((shp - 1) / 2).astype(np.float32)
            room = np.maximum((shp - np.asarray(self.size)).astype(np.float32) / np.float32(2), np.float32(0))
            hit = cache[shape3] = (shp, centre32, room.astype(np.float64))
            if len(cache) > 16:
                cache.pop(next(iter(cache)))
        return hit

    def _shape_lists(self, shape3):
        """`_shape_constants` as Python floats: (centre as float32 values, room for the random shift as float64 values)."""
        cache = self.__dict__.setdefault("_shape_cache_l", {})
        hit = cache.get(shape3)
        if hit is None:
            _shp, centre32, room64 = self._shape_constants(shape3)
            hit = cache[shape3] = (centre32.astype(np.float64).tolist(), room64.tolist())
            if len(cache) > 16:
                cache.pop(next(iter(cache)))
        return hit

    # ---- host: random draws in the reference's order (ref :140-145, :248-263, :284, :303-318) ----
    def plan(self, image_shape, random_shift=True, genparams: dict = {}) -> DeformPlan:
        p = DeformPlan()
        gate = np.random.rand() < self.prob
        if not (gate or len(genparams.keys()) > 0):
            return p
        p.active = True
        p.flip = genparams["flip"] if "flip" in genparams else bool(np.random.rand() < self.flip_prb)
        shp, centre32, room64 = self._shape_constants(tuple(image_shape)[0:3])

        ga = genparams.get("affine", {})
        rot = ga["rotations"] if "rotations" in ga else (
            (2 * self.max_rotation * np.random.rand(3) - self.max_rotation) / 180.0 * np.pi)
        shr = ga["shears"] if "shears" in ga else 2 * self.max_shear * np.random.rand(3) - self.max_shear
        scl = ga["scalings"] if "scalings" in ga else 1 + (2 * self.max_scaling * np.random.rand(3) - self.max_scaling)
        p.A = torch.from_numpy(make_affine_matrix(rot, shr, scl).astype(np.float32))
        if random_shift:
            u = torch.rand(3, dtype=torch.float64).numpy()  # float64 draw, always consumed
            # fp32 centre + fp64 shift promotes to float64 in the reference; with no room the shift is exactly 0
            centre = centre32.astype(np.float64) + (2 * (room64 * u) - room64)
        else:
            centre = centre32
        p.c2 = torch.from_numpy(np.ascontiguousarray(centre))
        aff_params = {"rotations": rot, "shears": shr, "scalings": scl}

        nr_params = {}
        if self.nonlinear_transform:
            gn = genparams.get("non_rigid", {})
            scale = gn["nonlin_scale"] if "nonlin_scale" in gn else (
                self.nonlin_scale_min + np.random.rand(1) * (self.nonlin_scale_max - self.nonlin_scale_min))
            small = gn["size_F_small"] if "size_F_small" in gn else np.round(scale * shp).astype(int).tolist()
            std = gn["nonlin_std"] if "nonlin_std" in gn else self.nonlin_std_max * np.random.rand()
            p.field_small = std * torch.randn([*small, 3], dtype=torch.float32)
            nr_params = {"nonlin_scale": scale, "nonlin_std": std, "size_F_small": small}
        p.params = {"affine": aff_params, "non_rigid": nr_params, "flip": p.flip}
        return p

    # ---- device ---------------------------------------------------------------------------------
    def make_spec(self, plan: DeformPlan, image_shape, flip_in_kernel: bool, arena: T.Arena | None = None):
        """DeformSpec (+ the arena that must be uploaded before launch when one is passed in)."""
        shape = tuple(int(v) for v in tuple(image_shape)[0:3])
        centre = self.__dict__.get("_centre")
        if centre is None:
            centre = self._centre = (np.array(self.size) - 1) / 2
        field_dev, tabs = None, None
        own = arena is None
        arena = arena or T.Arena()
        pending = None
        if plan.field_small is not None:
            fs = plan.field_small
            host_tabs, new = T.zoom_tables_between(tuple(fs.shape[:3]), shape)  # factor = shape / coarse shape
            if new != shape:
                raise ValueError(f"coarse field {tuple(fs.shape[:3])} does not zoom to {shape} (got {new})")
            tabs = K.device_tables_for(host_tabs, self.device)
            pending = (arena.add(fs.numpy()), tuple(fs.shape))
        if own:
            arena.upload(self.device)
        return _SpecBuilder(self, plan, shape, centre, flip_in_kernel, arena, tabs, pending)

    def generate_deformation_and_flip(self, image_shape, random_shift=True, genparams={}):
        plan = self.plan(image_shape, random_shift=random_shift, genparams=genparams)
        if not plan.active:
            return None, None, None, False, plan.params
        spec = self.make_spec(plan, image_shape, flip_in_kernel=False).build()
        mm6 = K.coords_minmax(spec)
        xx2, yy2, zz2 = K.coords(spec, mm6)
        return xx2, yy2, zz2, plan.flip, plan.params

    def apply_deformation_and_flip(self, image, segmentation, output, xx2, yy2, zz2, flip):
        if flip:
            segmentation = torch.flip(segmentation, [0])
            output = torch.flip(output, [0])
            image = torch.flip(image, [0]) if image is not None else None
        if xx2 is not None:
            output = fast_3D_interp_torch(output.contiguous(), xx2, yy2, zz2, "linear")
            segmentation = fast_3D_interp_torch(segmentation.to(self.device).contiguous(), xx2, yy2, zz2, "nearest")
            if image is not None:
                image = fast_3D_interp_torch(image.to(self.device).contiguous(), xx2, yy2, zz2, "linear")
        return image, segmentation, output

    def run(self, plan: DeformPlan, image, segmentation, output, spec=None, mm6=None, gamma=None, bias=None,
            bias_tabs=None, segmentation_u8=None):
        """Fused execution: one min/max launch + one warp launch (+ one more if `image` is given)."""
        if not plan.active:
            return image, segmentation, output
        if spec is None:
            spec = self.make_spec(plan, output.shape, flip_in_kernel=True).build()
        if mm6 is None:
            mm6 = K.coords_floormin(spec)
        seg = segmentation.to(self.device).contiguous()
        if seg.dtype not in (torch.float32, torch.uint8):
            seg = seg.float()
        if segmentation_u8 is not None and tuple(segmentation_u8.shape) == tuple(seg.shape):
            seg = segmentation_u8.contiguous()  # caller asked for uint8 labels in and out (1 B/voxel each way)
        if not spec.c.rows:
            spec.prepare_rows(bias, bias_tabs)
        out, seg = K.warp(spec, mm6, src_lin=output.contiguous(), src_nn=seg, gamma=gamma, bias=bias,
                          bias_tabs=bias_tabs)
        if image is not None:
            image, _ = K.warp(spec, mm6, src_lin=image.to(self.device).float().contiguous())
        return image, seg, out

    def deform(self, image, segmentation, output, genparams: dict = {}):
        plan = self.plan(output.shape, random_shift=True, genparams=genparams)
        image, segmentation, output = self.run(plan, image, segmentation, output)
        return image, segmentation, output, plan.params


class _SpecBuilder:
    """Defers pointer resolution until the arena holding the tables/field has been uploaded."""

    def __init__(self, owner, plan, shape, centre, flip_in_kernel, arena, tabs, pending):
        self.o, self.plan, self.shape, self.centre = owner, plan, shape, centre
        self.flip_in_kernel, self.arena, self.tabs, self.pending = flip_in_kernel, arena, tabs, pending

    def build(self) -> K.DeformSpec:
        field = None
        if self.pending is not None:
            off, fshape = self.pending
            field = self.arena.f32(off, fshape)
        c2 = self.plan.c2.to(torch.float32).numpy()  # the kernels add it as fp32, like ATen does
        return K.DeformSpec(self.shape, self.plan.A.numpy(), self.centre, c2,
                            self.plan.flip and self.flip_in_kernel, field, self.tabs, device=self.o.device)
