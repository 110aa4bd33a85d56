"""Where the two large Gaussian fields of a sample come from.

The reference draws them with `torch.randn(shape, device=device)` from the global generator of
`device` (rand_gmm.py:146-148, synthseg.py:230-232).  Two modes here:

  "reference": draw with `torch.randn(shape)` from the CPU global generator at the same point of the
               draw order and upload -- bit-identical noise to the reference's CPU path under the
               same `torch.manual_seed`; costs a 4 B/voxel host draw + PCIe copy per field.
  "device":    in-kernel Philox4x32-10; the 64-bit key of each field is taken from the CPU global
               generator (so runs are still reproducible under `torch.manual_seed`), nothing else
               touches the host.  `fsg_randn_f32` regenerates the identical field for checking.

  "keyed":     (fetalsyngen_amd/keyed.py) a sample is a function of its 64-bit key: every draw, small and large, from
               Philox4x32-10 under that key, inside one native call.  Where the fused keyed path does not apply (image as
               intensity prior, SR-artifact stages, genparams, stage-by-stage API) the global generators are seeded from
               the key and the sample is made as in "device" mode.

All SMALL draws (GMM tables, coarse displacement grid, bias grid, scalars) use numpy's / torch's CPU
global generators with the reference's calls in the reference's order in the first two modes.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np
import torch

_MODE = os.environ.get("FSG_RNG", "device")
_VALID = ("reference", "device", "keyed")


def get_mode() -> str:
    return _MODE


def set_mode(mode: str) -> None:
    global _MODE
    if mode not in _VALID:
        raise ValueError(f"rng mode must be one of {_VALID}")
    _MODE = mode


@contextlib.contextmanager
def use(mode: str | None):
    global _MODE
    if mode is None:
        yield
        return
    prev = _MODE
    set_mode(mode)
    try:
        yield
    finally:
        _MODE = prev


class Field:
    """A standard-normal field that is either host values or a Philox key."""

    __slots__ = ("shape", "host", "seed", "stream_id")

    def __init__(self, shape, host=None, seed=None, stream_id=0):
        self.shape, self.host, self.seed, self.stream_id = tuple(shape), host, seed, stream_id

    def device_tensor(self, device):
        """Materialise on the device (only needed by the un-fused API paths and by tests)."""
        from . import kernels as K

        if self.host is not None:
            h = self.host.pin_memory() if torch.cuda.is_available() else self.host
            return h.to(device, non_blocking=True)
        return K.randn(self.shape, self.seed, self.stream_id, device)


def normal_field(shape, stream_id: int = 0) -> Field:
    if _MODE == "reference":
        return Field(shape, host=torch.randn(tuple(shape), dtype=torch.float32))
    key = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
    return Field(shape, seed=key, stream_id=stream_id)


def distinct_ranks(count: int, k: int) -> torch.Tensor:
    """k distinct uniform integers in [0, count) (all of them, shuffled, if k >= count).

    "reference": `torch.randperm(count)[:k]`, the reference's own draw (simulate_reco.py:662, artifacts.py:200, :566) --
    O(count) host work, count = voxels of a mask.  "device": sequential draws with rejection of repeats from the same
    CPU generator -- the same distribution, O(k) work."""
    count, k = int(count), int(k)
    if _MODE == "reference" or k >= count // 2:
        return torch.randperm(count)[:k]
    seen, out = set(), []
    while len(out) < k:
        for v in torch.randint(0, count, (2 * (k - len(out)) + 8,), dtype=torch.int64).tolist():
            if v not in seen:
                seen.add(v)
                out.append(v)
                if len(out) == k:
                    break
    return torch.tensor(out, dtype=torch.int64)


def multinomial_distinct(prob: torch.Tensor, k: int) -> torch.Tensor:
    """k distinct indices drawn sequentially with probability proportional to `prob` (host tensor, need not be normalised).

    "reference": `torch.multinomial(prob, k)` (artifacts.py:110).  "device": inverse-CDF draws with rejection of repeats
    (sequential sampling without replacement, the same distribution) -- one cumulative sum instead of torch's
    per-element exponential race."""
    if _MODE == "reference" or k >= prob.numel() // 2:
        return torch.multinomial(prob, k)
    # numpy for the million-element pass: single-threaded by construction (a torch CPU op of this size wakes the whole
    # intra-op pool, see hostenv.py); the same sequential float64 running sum as torch.cumsum(prob.double())
    cdf = np.cumsum(prob.numpy(), dtype=np.float64)
    total = float(cdf[-1])
    seen, out = set(), []
    while len(out) < k:
        u = torch.rand(2 * (k - len(out)) + 8, dtype=torch.float64).numpy() * total
        for v in np.minimum(np.searchsorted(cdf, u, side="right"), prob.numel() - 1).tolist():
            if v not in seen:
                seen.add(v)
                out.append(v)
                if len(out) == k:
                    break
    return torch.tensor(out, dtype=torch.int64)
