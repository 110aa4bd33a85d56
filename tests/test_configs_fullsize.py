"""BASELINE configs[2] and configs[4] at their stated volume size (256^3), on the GPU, against the pinned oracle.

configs[2]: "Batch of 32 x 256^3 volumes sharded over 8 MI355X (embarrassingly parallel, no RCCL)" -- independent samples,
            reference generator/model.py:231-276.  One GPU here: the eight ranks are simulated (eight generators, each
            producing the indices `i % 8 == rank`), which is exactly what the N-rank bench does minus the devices.
configs[4]: "Streaming 10k-volume epoch at 256^3 ... feeding a dummy PyTorch-ROCm DataLoader consumer" -- the hand-over
            contract of reference data/datasets.py:310-325 (float32 image (1,H,W,D) + int64 labels on the CPU), and the
            device-resident mode.  64 volumes of the stream are checked sample by sample.
"""
import functools

import numpy as np
import pytest
import torch

from oracle import fsg_oracle as O
from tests.util_cases import make_generator

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPE = (256, 256, 256)


@functools.lru_cache(maxsize=None)
def _phantom(variant):
    from fetalsyngen_amd.phantom import make_seed_volumes

    return make_seed_volumes(SHAPE, variant)


def _host_subject(k):
    """Label volumes of synthetic subject k as host arrays: base variant k % 4, rolled 8 * (k // 4) voxels along y and
    mirrored in z for odd k // 4 (the transform bench.py's config3 applies on the device)."""
    seg, seeds = _phantom(k % 4)
    r = k // 4
    if r == 0:
        return seg, seeds

    def fn(a):
        a = np.roll(a, 8 * r, 1)
        return np.ascontiguousarray(a[:, :, ::-1] if r % 2 else a)

    return fn(seg), {n: {m: fn(v) for m, v in d.items()} for n, d in seeds.items()}


def _device_subject(k, banks, segs):
    from fetalsyngen_amd.data.datasets import SeedBank

    b, r = k % 4, k // 4
    if b not in banks:
        seg, seeds = _phantom(b)
        banks[b], segs[b] = SeedBank(seeds, DEV), torch.from_numpy(seg).to(DEV)
    if r == 0:
        return segs[b], banks[b]
    fn = (lambda x: torch.roll(x, 8 * r, 1).flip(2).contiguous()) if r % 2 else (lambda x: torch.roll(x, 8 * r, 1).contiguous())
    return fn(segs[b]), banks[b].transformed(fn)


def _oracle_for_key(base_seed, i, seg, seeds, K):
    """The oracle's sample under the key (base_seed, i), fed the Philox fields the kernels generate (the keys are drawn from
    torch's CPU generator at the same points of the order as the product draws them)."""
    from fetalsyngen_amd import sharding

    streams = iter((1, 2))

    def philox_field(shp):
        key = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
        return K.randn(shp, key, next(streams), DEV).cpu()

    sharding.seed_for_sample(base_seed, i)
    return O.run_sample(O.Config(SHAPE, prob=1.0), torch.from_numpy(seg), seeds, noise_gmm=philox_field,
                        noise_lowres=philox_field)


@pytest.fixture(scope="module")
def K():
    from fetalsyngen_amd import kernels

    return kernels


def test_config3_32_volumes_at_256_do_not_depend_on_the_sharding(K):
    from fetalsyngen_amd import sharding

    B, base_seed, world = 32, 4321, 8
    banks, segs = {}, {}
    subj = {i: _device_subject(i, banks, segs) for i in range(B)}

    gen = make_generator(SHAPE, DEV, rng="device")
    ref = {}
    for i in range(B):  # world = 1: one generator makes the whole batch
        sharding.seed_for_sample(base_seed, i)
        out, seg, _img, _p = gen._pipeline(None, subj[i][0], subj[i][1], {}, scale01=True)
        ref[i] = (out, seg)
    torch.cuda.synchronize()
    sums = {i: float(o.double().sum()) for i, (o, _s) in ref.items()}
    assert len({round(v, 3) for v in sums.values()}) == B  # 32 different volumes

    seen = set()
    for rank in range(world):  # eight simulated ranks, each with its own generator (own workspaces, caches, tables)
        g = make_generator(SHAPE, DEV, rng="device")
        mine = list(sharding.shard(B, rank, world))
        assert mine == [i for i in range(B) if i % world == rank]
        for i in mine:
            sharding.seed_for_sample(base_seed, i)
            out, seg, _img, _p = g._pipeline(None, subj[i][0], subj[i][1], {}, scale01=True)
            assert torch.equal(out, ref[i][0]) and torch.equal(seg, ref[i][1]), f"volume {i} differs on rank {rank}"
            seen.add(i)
        del g
    assert seen == set(range(B))

    for i in (3, 21):  # one untransformed subject, one rolled + mirrored: against the pinned oracle
        seg_h, seeds_h = _host_subject(i)
        assert np.array_equal(subj[i][0].cpu().numpy(), seg_h)
        r = _oracle_for_key(base_seed, i, seg_h, seeds_h, K)
        assert np.array_equal(ref[i][1].cpu().numpy().astype(np.uint8), r["seg"].numpy().astype(np.uint8)), i
        np.testing.assert_allclose(ref[i][0].cpu().numpy(), r["scaled"].numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("mode", ["cpu_contract", "device_resident", "cpu_contract_batched"])
def test_config5_stream_of_64_volumes_at_256_equals_direct_calls(K, mode):
    from fetalsyngen_amd import sharding
    from fetalsyngen_amd.data.datasets import MemorySynthDataset
    from fetalsyngen_amd.data.staging import PrefetchingStream

    n, base_seed = 64, 99
    segs_h, seeds_h = zip(*[_phantom(v) for v in range(4)])
    gen = make_generator(SHAPE, DEV, rng="device")
    ds = MemorySynthDataset(gen, list(segs_h), list(seeds_h))
    idx = list(range(5, 5 + n))
    kw = {"cpu_contract": dict(to_host=True, depth=3), "device_resident": dict(to_host=False),
          "cpu_contract_batched": dict(to_host=True, depth=2, batch_size=4, batch_streams=2, keep=2)}[mode]
    direct = make_generator(SHAPE, DEV, rng="device")
    ds2 = MemorySynthDataset(direct, list(segs_h), list(seeds_h))
    to_host = kw.get("to_host", True)

    def direct_sample(i):
        sharding.seed_for_sample(base_seed, i)
        k = i % len(ds2)
        bank, seg, _twin = ds2._subject(k)
        out, lab, _img, _p = direct._pipeline(None, seg, bank, {}, scale01=True, labels_u8=not to_host)
        return out, lab, ds2._sub_ses_idx(k)

    pos, checked_oracle = 0, False
    for item in PrefetchingStream(ds, idx, base_seed=base_seed, **kw):
        B = item["image"].shape[0] if kw.get("batch_size") else 1
        for b in range(B):
            i = idx[pos]
            out, lab, name = direct_sample(i)
            img_i = item["image"][b] if kw.get("batch_size") else item["image"]
            lab_i = item["label"][b] if kw.get("batch_size") else item["label"]
            name_i = item["name"][b] if kw.get("batch_size") else item["name"]
            assert name_i == name
            assert tuple(img_i.shape) == (1, *SHAPE) and img_i.dtype == torch.float32
            if to_host:
                assert not img_i.is_cuda and lab_i.dtype == torch.int64 and not lab_i.is_cuda
                assert torch.equal(img_i[0], out.cpu()), f"sample {i}: image differs from the direct call"
                assert torch.equal(lab_i[0], lab.cpu().long()), f"sample {i}: labels differ from the direct call"
            else:
                assert img_i.is_cuda and lab_i.dtype == torch.uint8
                assert torch.equal(img_i[0], out) and torch.equal(lab_i[0], lab)
            if not checked_oracle and pos == 9:
                r = _oracle_for_key(base_seed, i, segs_h[i % 4], seeds_h[i % 4], K)
                assert np.array_equal(lab_i[0].cpu().numpy().astype(np.uint8), r["seg"].numpy().astype(np.uint8))
                np.testing.assert_allclose(img_i[0].cpu().numpy(), r["scaled"].numpy(), rtol=0, atol=2e-5)
                checked_oracle = True
            pos += 1
    assert pos == n and checked_oracle
