"""Multi-GPU story of this path: independent volumes, one process per GPU, no collective.

The reference has no distributed code at all; volumes have no cross-volume dependency
(generator/model.py:231-276 touches only its arguments), so a batch / epoch is partitioned by sample
index.  Per-sample RNG keys depend only on (base_seed, sample index), so results do not depend on the
number of GPUs or on which rank produced a sample.
"""
from __future__ import annotations

import os

import numpy as np
import torch

_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def sample_key(base_seed: int, index: int) -> int:
    """64-bit key of sample `index` under `base_seed` (independent of world size)."""
    return splitmix64(splitmix64(base_seed & _M64) ^ (index & _M64))


_PENDING_KEY = [None]


def take_key():
    """The key of the sample announced by the last `seed_for_sample` / `announce_key` call, handed out once (keyed mode:
    `FetalSynthGen._pipeline` picks it up when the caller passes no explicit key)."""
    k, _PENDING_KEY[0] = _PENDING_KEY[0], None
    return k


def announce_key(base_seed: int, index: int) -> int:
    """Keyed mode's `seed_for_sample`: the sample's key without re-seeding the global generators (an MT19937 re-seed costs
    ~10 us, more than the rest of a keyed sample's Python side)."""
    k = _PENDING_KEY[0] = sample_key(base_seed, index)
    return k


def seed_for_sample(base_seed: int, index: int) -> int:
    """Seed numpy's and torch's CPU global generators for one sample; returns the key."""
    k = sample_key(base_seed, index)
    return seed_from_key(k)


def seed_from_key(k: int) -> int:
    _PENDING_KEY[0] = k
    np.random.seed(k & 0xFFFFFFFF)
    # CPU generator only: torch.manual_seed() would also walk every accelerator backend's lazy
    # seeding hook (~0.1 ms per call); all host draws of this package use the CPU generator.
    torch.default_generator.manual_seed(k >> 1)
    return k


def shard(n_items: int, rank: int, world: int) -> range:
    """Round-robin partition: item i belongs to rank i % world."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return range(rank, n_items, world)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class ShardedSynthStream(torch.utils.data.IterableDataset):
    """Streams `n_items` synthetic samples; this rank (and DataLoader worker) produces its share.

    `make_sample(index) -> dict` is called after seeding the global generators with
    `seed_for_sample(base_seed, index)`."""

    def __init__(self, make_sample, n_items: int, base_seed: int = 0, rank: int | None = None, world: int | None = None):
        r, w, _ = env_rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.make_sample, self.n_items, self.base_seed = make_sample, n_items, base_seed

    def __iter__(self):
        info = torch.utils.data.get_worker_info()
        wid, nw = (info.id, info.num_workers) if info is not None else (0, 1)
        for i in shard(self.n_items, self.rank * nw + wid, self.world * nw):
            seed_for_sample(self.base_seed, i)
            yield self.make_sample(i)
