"""GMM intensity generator: mirror of `fetalsyngen.generator.intensity.rand_gmm.ImageFromSeeds`
(reference rand_gmm.py:9-154) on the `fsg_gmm_sample_*` kernel.

Same constructor arguments, same `load_seeds` / `sample_intensities` signatures, return values and
`ValueError`s.  Differences that do not change results: labels may stay uint8 (all values are < 50)
and may already be device resident; the Gaussian field comes from `fetalsyngen_amd.rng`.
"""
from __future__ import annotations

from pathlib import Path
from typing import Iterable

import numpy as np
import torch

from ... import kernels as K
from ... import rng
from ...utils.image_reading import NiftiReader


class GMMPlan:
    """Everything `sample_intensities` draws, before any kernel runs."""

    __slots__ = ("mus", "sigmas", "field")

    def __init__(self, mus, sigmas, field):
        self.mus, self.sigmas, self.field = mus, sigmas, field


class ImageFromSeeds:
    def __init__(
        self,
        min_subclusters: int,
        max_subclusters: int,
        seed_labels: Iterable[int],
        generation_classes: Iterable[int],
        meta_labels: int = 4,
    ):
        seed_labels, generation_classes = list(seed_labels), list(generation_classes)
        if len(set(seed_labels)) != len(seed_labels):
            raise ValueError("Parameter seed_labels should have unique values.")
        if len(seed_labels) != len(generation_classes):
            raise ValueError("Parameters seed_labels and generation_classes should have the same lengths.")
        self.min_subclusters = min_subclusters
        self.max_subclusters = max_subclusters
        self.seed_labels = seed_labels
        self.generation_classes = generation_classes
        self.meta_labels = meta_labels
        self.loader = NiftiReader()
        self._idx = None

    # -- seeds ------------------------------------------------------------------------------
    def draw_subclusters(self, genparams: dict = {}) -> dict:
        """One `np.random.randint` per meta label, always consumed (ref :81-87)."""
        chosen = {
            m: np.random.randint(self.min_subclusters, self.max_subclusters + 1)
            for m in range(1, self.meta_labels + 1)
        }
        return genparams["mlabel2subclusters"] if "mlabel2subclusters" in genparams else chosen

    def load_seeds(self, seeds, mlabel2subclusters=None, genparams: dict = {}):
        """Sum of the selected per-meta-label seed volumes as a LongTensor (H,W,D).

        `seeds[n_sub][mlabel]` is a path (decoded with the built-in NIfTI reader) or an
        already decoded array / tensor."""
        if mlabel2subclusters is None:
            mlabel2subclusters = self.draw_subclusters(genparams)
        elif "mlabel2subclusters" in genparams:
            mlabel2subclusters = genparams["mlabel2subclusters"]
        total = None
        for m in range(1, self.meta_labels + 1):
            item = seeds[mlabel2subclusters[m]][m]
            vol = self.loader(item) if isinstance(item, (str, Path)) else torch.as_tensor(np.asarray(item))
            total = vol.clone() if total is None else total + vol
        return total.long(), {"mlabel2subclusters": mlabel2subclusters}

    # -- intensities ------------------------------------------------------------------------
    def plan_intensities(self, shape, genparams: dict = {}) -> GMMPlan:
        """torch draws in the reference's order: rand(nlabels), rand(nlabels), randn(nsamp), field.
        The arithmetic on the 50-entry tables is done in numpy float32 (same IEEE operations as ATen,
        a few times cheaper per call on arrays this small)."""
        nlabels = max(self.seed_labels) + 1
        if "mus" in genparams:
            mus = torch.as_tensor(genparams["mus"]).detach().to("cpu", torch.float32).numpy().copy()
        else:
            mus = np.float32(25) + np.float32(200) * torch.rand(nlabels, dtype=torch.float32).numpy()
        if "sigmas" in genparams:
            sigmas = torch.as_tensor(genparams["sigmas"]).detach().to("cpu", torch.float32).numpy().copy()
        else:
            sigmas = np.float32(5) + np.float32(20) * torch.rand(nlabels, dtype=torch.float32).numpy()
        if self.generation_classes != self.seed_labels:
            if self._idx is None:
                self._idx = (np.asarray(self.generation_classes), np.asarray(self.seed_labels))
            z = torch.randn(len(self.seed_labels), dtype=torch.float32).numpy()
            tied = mus[self._idx[0]] + np.float32(25) * z
            mus[self._idx[1]] = np.minimum(np.maximum(tied, np.float32(0)), np.float32(225))
        return GMMPlan(torch.from_numpy(mus), torch.from_numpy(sigmas), rng.normal_field(shape, stream_id=1))

    def run_intensities(self, seeds, device, plan: GMMPlan):
        if seeds.dtype not in (torch.uint8, torch.int64):
            seeds = seeds.long()
        labels = seeds.to(device).contiguous()
        mus, sigmas = plan.mus.to(device), plan.sigmas.to(device)
        f = plan.field
        noise = f.device_tensor(device) if f.host is not None else None
        img = K.gmm_sample(labels, mus, sigmas, noise=noise, seed=f.seed or 0, stream_id=f.stream_id)
        return img, {"mus": mus, "sigmas": sigmas}

    def sample_intensities(self, seeds: torch.Tensor, device: str, genparams: dict = {}):
        plan = self.plan_intensities(tuple(seeds.shape), genparams)
        return self.run_intensities(seeds, device, plan)
