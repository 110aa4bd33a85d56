"""CPU, world_size 2, gloo: the N>1 path of bench.py / ShardedSynthStream -- index sharding, per-sample
keys independent of the world size, barrier + max-over-ranks timing exchange.  No GPU, no kernels: the
per-sample work is the host plan (all RNG draws of one sample), whose digest must not depend on which
rank produced it."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _plan_digest(index, base_seed=77, shape=(32, 32, 32)):
    """Digest of every random quantity the product draws on the host for sample `index`."""
    from fetalsyngen_amd import rng, sharding
    from fetalsyngen_amd.generator.augmentation.synthseg import RandBiasField, RandGamma, RandNoise, RandResample
    from fetalsyngen_amd.generator.deformation.affine_nonrigid import SpatialDeformation
    from fetalsyngen_amd.generator.intensity.rand_gmm import ImageFromSeeds

    labels = [0] + list(range(10, 50))
    classes = [0] + [10] * 10 + [20] * 10 + [30] * 10 + list(range(40, 50))
    key = sharding.seed_for_sample(base_seed, index)
    with rng.use("device"):
        ig = ImageFromSeeds(1, 6, labels, classes)
        m2s = ig.draw_subclusters({})
        gp = ig.plan_intensities(shape, {})
        dp = SpatialDeformation(20, 0.02, 0.1, list(shape), 0.9, True, 0.1, 0.3, 4, 0.5, "cuda:0").plan(shape)
        g = RandGamma(0.9, 0.1).plan({})
        bp = RandBiasField(0.9, 0.05, 0.2, 0.01, 0.3).plan(shape, {})
        rp = RandResample(0.9, 0.5, 1.5).plan(shape, np.array([0.5] * 3), {})
        npn = RandNoise(0.9, 5, 15).plan(rp.new_size if rp.active else shape, {})
    parts = [float(key % 1000003), float(sum(m2s.values())), float(gp.mus.sum()), float(gp.field.seed % 1000003),
             float(dp.A.sum()) if dp.active else -1.0, float(g or -1.0), float(bp.grid.sum()) if bp.active else -1.0,
             float(rp.new_size[0]) if rp.active else -1.0, float(npn.std32 or -1.0) if npn.active else -1.0]
    return np.array(parts, dtype=np.float64)


def _worker(rank, world, port, n_items, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fetalsyngen_amd import sharding

    stream = sharding.ShardedSynthStream(lambda i: (i, None), n_items, base_seed=77)
    mine = [i for i, _ in stream]
    assert mine == list(sharding.shard(n_items, rank, world))
    digests = {i: _plan_digest(i) for i in mine}
    dist.barrier()
    t = torch.tensor([0.5 + rank], dtype=torch.float64)  # stand-in for the measured time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 0.5 + (world - 1)
    gathered = [None] * world
    dist.all_gather_object(gathered, digests)
    if rank == 0:
        merged = {}
        for d in gathered:
            assert not (set(d) & set(merged)), "an index was produced by two ranks"
            merged.update(d)
        np.save(Path(out_dir) / "digests.npy", np.stack([merged[i] for i in range(n_items)]))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_sharding_matches_single_process(tmp_path):
    n_items = 7  # ragged: rank 0 gets 4 items, rank 1 gets 3
    mp.spawn(_worker, args=(2, _free_port(), n_items, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "digests.npy")
    want = np.stack([_plan_digest(i) for i in range(n_items)])  # world size 1, same process
    assert np.array_equal(got, want), "per-sample draws depend on the world size / rank"
    assert len({tuple(r) for r in got}) == n_items, "samples are not distinct"


def test_shard_partition_properties():
    from fetalsyngen_amd import sharding

    for n in (0, 1, 5, 32, 10_000):
        for world in (1, 2, 3, 8):
            parts = [list(sharding.shard(n, r, world)) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert sharding.sample_key(1, 2) != sharding.sample_key(2, 1)
    assert sharding.sample_key(5, 9) == sharding.sample_key(5, 9)
