"""The drop-in must survive the reference's DataLoader pattern: `DataLoader(ds, batch_size=2, num_workers=2,
multiprocessing_context="spawn")` (reference fetalsyngen/test_dl.py:17-24, docs/datasets.md:4-6) pickles the dataset --
generator, cached label volumes and all -- into fresh worker processes.  Nothing process-local (host addresses, device
tensors, HIP handles, caches keyed by `id()`) may cross; everything is rebuilt lazily in the worker.

CPU part: the pickled state of every object on that path, walked attribute by attribute.
GPU part (`-m gpu`): a dataset that has already produced a sample, iterated through spawned workers, equals `num_workers=0`.
"""
import pickle

import numpy as np
import pytest
import torch

from tests.util_cases import make_generator


def _walk(obj, seen=None, path="obj"):
    """Yield (path, leaf) for everything reachable through __dict__ / containers of a pickled-and-restored object."""
    seen = set() if seen is None else seen
    if id(obj) in seen:
        return
    seen.add(id(obj))
    yield path, obj
    if isinstance(obj, dict):
        for k, v in obj.items():
            yield from _walk(v, seen, f"{path}[{k!r}]")
    elif isinstance(obj, (list, tuple, set)):
        for i, v in enumerate(obj):
            yield from _walk(v, seen, f"{path}[{i}]")
    elif hasattr(obj, "__dict__") and not isinstance(obj, type):
        for k, v in vars(obj).items():
            yield from _walk(v, seen, f"{path}.{k}")


def _assert_clean(obj, forbidden_ints=()):
    for path, leaf in _walk(obj):
        if torch.is_tensor(leaf):
            assert not leaf.is_cuda, f"{path}: a device tensor survived pickling"
        if isinstance(leaf, int) and not isinstance(leaf, bool):
            assert leaf not in forbidden_ints, f"{path}: an address of the parent process survived pickling"


def test_generator_pickle_drops_every_process_local_thing():
    shape = (16, 16, 16)
    gen = make_generator(shape, "cuda:0", rng="device")
    fb = gen._flat_buffers()
    parent_addresses = {fb["ivp"], fb["fvp"], fb["tbp"]}
    # what a generator that has produced samples carries (device tensors stood in for by CPU tensors: no GPU here)
    gen._ws[(shape, 0, 0)] = {"ws0": torch.zeros(8), "ws1": torch.zeros(8), "low": torch.zeros(8), "rows": None, "stride": 0}
    gen.__dict__["_twins"] = {"by_id": {1: [None, 0, torch.zeros(8, dtype=torch.uint8), 2]}, "bytes": 8}
    gen.__dict__["_seen_parts"] = {123: (None, 0, 0)}
    gen.__dict__["_arena_next"] = {(0, 0): torch.zeros(8, dtype=torch.uint8)}
    gen.__dict__["_rs_dt"] = {((8, 8, 8), shape): object()}
    gen.blur_events = [(1, 2, [])]
    fb["validated"][42] = (None, shape)

    clone = pickle.loads(pickle.dumps(gen))
    for name in ("_flat", "_twins", "_seen_parts", "_arena_next", "_rs_dt", "_batch_streams", "_keyed"):
        assert name not in clone.__dict__, name
    assert clone._ws == {} and clone.blur_events is None
    _assert_clean(clone, parent_addresses)
    # configuration travelled
    assert clone.shape == gen.shape and clone.resolution == gen.resolution and clone.device == gen.device
    assert clone.spatial_deform.max_rotation == gen.spatial_deform.max_rotation
    assert clone.intensity_generator.seed_labels == gen.intensity_generator.seed_labels
    # and the rebuilt flat buffers point at the clone's own arrays
    fb2 = clone._flat_buffers()
    assert fb2["ivp"] == fb2["iv"].ctypes.data and fb2["fvp"] == fb2["fv"].ctypes.data and fb2["tbp"] == fb2["tb"].ctypes.data
    assert fb2["validated"] == {}
    # the host plans of the clone are those of the original under the same seeds
    for g in (gen, clone):
        np.random.seed(3)
        torch.manual_seed(3)
        g._plan = g.plan_only(shape)
    a, b = gen._plan, clone._plan
    assert a[0] == b[0] and torch.equal(a[1].mus, b[1].mus) and torch.equal(a[2].A, b[2].A)
    assert torch.equal(a[2].field_small, b[2].field_small) and a[5].new_size == b[5].new_size


def test_seed_bank_pickles_as_host_arrays():
    from fetalsyngen_amd.data.datasets import SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes

    _seg, seeds = make_seed_volumes((12, 12, 12))
    bank = SeedBank(seeds, "cpu")  # "cpu" stands in for the device: the class only ever calls .to(device)
    m2s = {1: 2, 2: 1, 3: 3, 4: 1}
    ref = [p.clone() for p in bank.parts(m2s)]
    blob = pickle.dumps(bank)
    clone = pickle.loads(blob)
    assert clone._vol is None and clone._host is not None  # nothing uploaded until the worker asks
    for n, d in clone._host.items():
        for m, v in d.items():
            assert isinstance(v, np.ndarray) and v.dtype == np.uint8
    assert clone.nbytes == bank.nbytes == sum(len(d) for d in seeds.values()) * 12 ** 3
    got = clone.parts(m2s)
    assert clone._host is None and all(torch.equal(a, b) for a, b in zip(ref, got))
    assert clone.shape == (12, 12, 12)
    # a bank that was never touched after unpickling pickles again (worker of a worker)
    again = pickle.loads(pickle.dumps(pickle.loads(blob)))
    assert all(torch.equal(a, b) for a, b in zip(ref, again.parts(m2s)))


def test_datasets_pickle_without_cached_volumes(tmp_path):
    from fetalsyngen_amd.data.datasets import FetalSynthDataset, LabelCache, MemorySynthDataset, SeedBank
    from fetalsyngen_amd.phantom import make_seed_volumes
    from tests.util_bids import write_tree

    shape = (12, 12, 12)
    bids, seed_dir = write_tree(tmp_path, shape, ["sub-a", "sub-b"])
    gen = make_generator(shape, "cuda:0", rng="device")
    ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None, cache_bytes=12345, base_seed=5)
    gen.device = "cpu"  # stand-in device for the cache (no GPU here); the ctor's device check has passed
    ds._labels.device = "cpu"
    ds.generator.register_label_twin = lambda *_a: None
    bank, seg, twin = ds._subject(1)
    assert isinstance(bank, SeedBank) and seg.dtype == torch.float32 and twin.dtype == torch.uint8
    assert len(ds._labels) == 1 and ds._labels.misses == 1
    del ds.generator.register_label_twin
    clone = pickle.loads(pickle.dumps(ds))
    assert isinstance(clone._labels, LabelCache) and len(clone._labels) == 0 and clone._labels.budget_bytes == 12345
    assert clone.base_seed == 5 and clone.seed_paths.keys() == ds.seed_paths.keys()
    assert [str(p) for p in clone.segm_paths] == [str(p) for p in ds.segm_paths]
    _assert_clean(clone)

    segs, banks = [], []
    for v in range(2):
        s_, seeds = make_seed_volumes(shape, v)
        segs.append(s_)
        banks.append(seeds)
    gen2 = make_generator(shape, "cuda:0", rng="device")
    gen2.device = "cpu"
    mem = MemorySynthDataset(gen2, segs, banks, base_seed=9)
    clone = pickle.loads(pickle.dumps(mem))
    for bank, seg, twin in clone._mem:
        assert isinstance(seg, np.ndarray) and twin is None and bank._vol is None
    _assert_clean(clone)
    b, s_, t_ = clone._subject(1)
    assert torch.equal(s_, mem._subject(1)[1]) and torch.equal(t_, mem._subject(1)[2])
    assert all(torch.equal(x, y) for x, y in zip(b.parts({1: 1, 2: 2, 3: 3, 4: 4}), mem._subject(1)[0].parts({1: 1, 2: 2, 3: 3, 4: 4})))


def test_label_cache_lru_and_budget():
    from fetalsyngen_amd.data.datasets import LabelCache

    c = LabelCache("cpu", budget_bytes=250)
    built = []

    def make(k):
        def build():
            built.append(k)
            return f"v{k}", 100
        return build

    assert c.get(0, make(0)) == "v0" and c.get(1, make(1)) == "v1" and c.bytes == 200
    assert c.get(0, make(0)) == "v0" and c.hits == 1          # 0 is now the most recent
    assert c.get(2, make(2)) == "v2"                           # 300 > 250: the least recent (1) goes
    assert 1 not in c and 0 in c and 2 in c and c.bytes == 200 and c.evictions == 1
    assert c.get(1, make(1)) == "v1" and 0 not in c            # re-built from its source, 0 evicted
    assert built == [0, 1, 2, 1]
    tiny = LabelCache("cpu", budget_bytes=10)                  # smaller than one entry: keeps exactly the latest
    tiny.get("a", make("a"))
    tiny.get("b", make("b"))
    assert len(tiny) == 1 and "b" in tiny


# ---- GPU ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_spawned_dataloader_workers_reproduce_the_in_process_batches(tmp_path):
    """Reference pattern (fetalsyngen/test_dl.py:17-24).  The dataset has ALREADY produced a sample in this process (so the
    generator holds flat plan buffers, workspaces, cached label volumes), then goes through two spawned workers."""
    from fetalsyngen_amd.data.datasets import FetalSynthDataset
    from tests.util_bids import write_tree

    shape = (32, 32, 32)
    subs = ["sub-a", "sub-b", "sub-c", "sub-d", "sub-e", "sub-f"]
    bids, seed_dir = write_tree(tmp_path, shape, subs)
    gen = make_generator(shape, "cuda:0", rng="device", prob=0.9, nonlin_scale=(0.1, 0.3), bf_scale=(0.05, 0.2))
    ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None, base_seed=77)
    first = ds[0]
    assert "_flat" in gen.__dict__ and len(gen._ws) == 1 and len(ds._labels) == 1
    ref = list(torch.utils.data.DataLoader(ds, batch_size=2, num_workers=0))
    got = list(torch.utils.data.DataLoader(ds, batch_size=2, num_workers=2, multiprocessing_context="spawn"))
    assert len(ref) == len(got) == 3
    for a, b in zip(ref, got):
        assert a["name"] == b["name"]
        assert a["image"].shape == (2, 1, *shape) and b["image"].dtype == torch.float32 and not b["image"].is_cuda
        assert b["label"].dtype == torch.int64
        assert torch.equal(a["image"], b["image"]) and torch.equal(a["label"], b["label"])
    assert torch.equal(first["image"], ref[0]["image"][0])  # keyed by (base_seed, index): the very first call too
    ds.set_epoch(1)
    assert not torch.equal(ds[0]["image"], first["image"])


@pytest.mark.gpu
def test_memory_dataset_through_spawned_workers():
    from fetalsyngen_amd.data.datasets import MemorySynthDataset
    from fetalsyngen_amd.phantom import make_seed_volumes

    shape = (32, 32, 32)
    segs, banks = [], []
    for v in range(4):
        s_, seeds = make_seed_volumes(shape, v)
        segs.append(s_)
        banks.append(seeds)
    gen = make_generator(shape, "cuda:0", rng="device")
    ds = MemorySynthDataset(gen, segs, banks, base_seed=5)
    ds[1]
    ref = list(torch.utils.data.DataLoader(ds, batch_size=2, num_workers=0))
    got = list(torch.utils.data.DataLoader(ds, batch_size=2, num_workers=2, multiprocessing_context="spawn"))
    for a, b in zip(ref, got):
        assert torch.equal(a["image"], b["image"]) and torch.equal(a["label"], b["label"])


@pytest.mark.gpu
def test_label_cache_budget_smaller_than_the_subjects(tmp_path):
    """More subjects than the HBM budget holds: samples equal those of an unbounded cache and of no cache at all
    (the reference's behaviour: re-read per sample, data/datasets.py:280-296)."""
    from fetalsyngen_amd.data.datasets import FetalSynthDataset
    from tests.util_bids import write_tree

    shape = (32, 32, 32)
    subs = [f"sub-{c}" for c in "abcde"]
    bids, seed_dir = write_tree(tmp_path, shape, subs)
    per_subject = (24 + 4 + 1) * 32 ** 3
    order = [0, 1, 2, 3, 4, 0, 2, 4, 1, 3, 0, 0, 4]
    runs = {}
    for name, kw in (("bounded", dict(cache_bytes=int(2.5 * per_subject))), ("unbounded", dict(cache_bytes=1 << 40)),
                     ("uncached", dict(cache_on_device=False))):
        gen = make_generator(shape, "cuda:0", rng="device", prob=0.9)
        ds = FetalSynthDataset(str(bids), gen, str(seed_dir), None, base_seed=3, **kw)
        runs[name] = [ds[i] for i in order]
        if name == "bounded":
            c = ds._labels
            assert len(c) == 2 and c.bytes <= c.budget_bytes and c.evictions >= 8 and c.hits >= 1
        if name == "unbounded":
            assert len(ds._labels) == 5 and ds._labels.evictions == 0
    for k in range(len(order)):
        for other in ("unbounded", "uncached"):
            assert torch.equal(runs["bounded"][k]["image"], runs[other][k]["image"]), (k, other)
            assert torch.equal(runs["bounded"][k]["label"], runs[other][k]["label"]), (k, other)
