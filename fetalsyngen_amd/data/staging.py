"""Output side of the path: getting samples to the host without stalling the generator.

The reference's dataset ends every sample with `.cpu()` on the image and the labels (data/datasets.py:315-317)
-- a synchronous pageable copy of 64 MiB + (after `.long()`) 128 MiB at 256^3 that idles the GPU.  `HostStager`
keeps a ring of PINNED host buffers; a sample's device tensors are copied on a side stream while the next
samples are being generated, and handed out when their copy event has completed.  `PrefetchingStream` drives a
`FetalSynthDataset` that way (BASELINE config 5: streaming epoch into a DataLoader-style consumer).
"""
from __future__ import annotations

from collections import deque

import torch

from .. import sharding


class HostStager:
    """Ring of pinned host slots.  `depth` samples may be in flight (submitted, not collected); `keep` more
    slots stay untouched behind them, so the views `collect` hands out remain valid until `keep` further
    samples have been collected (default 2: the sample just yielded and the one before it)."""

    def __init__(self, shape, device, depth: int = 3, label_dtype=torch.int64, keep: int = 2, batch: int | None = None,
                 image_dtype=torch.float32):
        """shape: (H,W,D) of one volume.  batch=None: slots hold one sample, (1,H,W,D) as `FetalSynthDataset.__getitem__`
        returns it; batch=B: slots hold B samples, (B,1,H,W,D) as a collated DataLoader batch, moved with ONE copy per
        tensor.  image_dtype=torch.float16 halves the image bytes (conversion by fsg_cast_f32_to_f16 on the device)."""
        self.device = torch.device(device)
        self.depth = depth
        self.keep = max(int(keep), 1)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        full = (1, *shape) if batch is None else (int(batch), 1, *shape)
        self.slots = []
        for _ in range(depth + self.keep):
            self.slots.append({
                "image": torch.empty(full, dtype=image_dtype, pin_memory=True),
                "label": torch.empty(full, dtype=label_dtype, pin_memory=True),
                "event": torch.cuda.Event(),
                "busy": False,
            })
        self._next = 0
        self.label_dtype = label_dtype
        self.image_dtype = image_dtype

    def submit(self, image_dev: torch.Tensor, label_dev: torch.Tensor) -> int:
        """Enqueue the D2H copies of one sample; returns a ticket for `collect`."""
        slot_id = self._next
        slot = self.slots[slot_id]
        if slot["busy"]:
            raise RuntimeError("HostStager ring overrun: collect() a ticket before submitting more than `depth`")
        self._next = (self._next + 1) % len(self.slots)
        # the dtype conversion runs on the PRODUCING stream (the caching allocator then knows who reads
        # `label_dev`); the copy stream only ever touches tensors that carry a record_stream mark
        lab = label_dev if label_dev.dtype == self.label_dtype else label_dev.to(self.label_dtype)
        if image_dev.dtype != self.image_dtype:
            if self.image_dtype != torch.float16:
                raise TypeError("HostStager image_dtype must be torch.float32 or torch.float16")
            from .. import kernels as K

            image_dev = K.cast_f16(image_dev)
        produced = torch.cuda.Event()
        produced.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.copy_stream):
            self.copy_stream.wait_event(produced)
            slot["image"].view(image_dev.shape).copy_(image_dev, non_blocking=True)
            slot["label"].view(lab.shape).copy_(lab, non_blocking=True)
            # the device tensors must outlive the copy although their Python references may not
            image_dev.record_stream(self.copy_stream)
            lab.record_stream(self.copy_stream)
            slot["event"].record(self.copy_stream)
        slot["busy"] = True
        return slot_id

    def collect(self, ticket: int, clone: bool = False):
        """Wait for the sample's copies; returns (image, label) views of the pinned slot (valid until `keep`
        further tickets have been collected AND submitted over) or private copies when `clone`."""
        slot = self.slots[ticket]
        slot["event"].synchronize()
        slot["busy"] = False
        if clone:
            return slot["image"].clone(), slot["label"].clone()
        return slot["image"], slot["label"]


class PrefetchingStream:
    """Iterates `indices` of a FetalSynthDataset keeping `depth` samples in flight.

    to_host=True : yields the reference contract (image float32 (1,H,W,D) and int64 labels on the CPU, pinned)
    to_host=False: yields device-resident tensors (float32 image, uint8 labels).

    Lifetime of a yielded host sample: it is a view of a pinned ring slot and stays intact while the consumer
    holds it and the `keep - 1` samples yielded after it (default keep=2: `prev, cur` patterns are safe;
    collating B samples needs keep >= B, or `.clone()`)."""

    def __init__(self, dataset, indices, base_seed: int = 0, depth: int = 3, to_host: bool = True,
                 label_dtype=torch.int64, keep: int = 2, batch_size: int | None = None, image_dtype=torch.float32,
                 batch_streams: int = 1):
        """batch_size=B: B consecutive indices are generated by one `FetalSynthGen.sample_batch` call (one parameter
        upload, one native call, `batch_streams` HIP streams) and yielded as one collated item -- image (B,1,H,W,D),
        label (B,1,H,W,D), name: list of B -- moved to the host with one copy per tensor.  Sample i is seeded with
        (base_seed, i) whatever the batch size, so the stream of samples does not depend on it."""
        self.ds, self.indices, self.base_seed, self.depth, self.to_host = dataset, list(indices), base_seed, depth, to_host
        self.label_dtype = label_dtype
        self.keep = keep
        self.batch_size, self.image_dtype, self.batch_streams = batch_size, image_dtype, batch_streams
        self._stager = None
        # uint8 labels straight from the fused warp wherever the consumer gets uint8 anyway (the int64 contract keeps the
        # float32 labels and the reference's `.long()` semantics for any label value)
        self._labels_u8 = (not to_host) or label_dtype == torch.uint8

    def __len__(self):
        return len(self.indices)

    def _produce(self, i, i_next=None):
        # keyed mode: the sample is a function of its key alone -- no re-seeding of the global generators; the stream knows the
        # next index, whose draw job then rides in this sample's launches (FetalSynthGen._pipeline_keyed: next_key)
        next_key = None
        if self.ds.generator._is_keyed():
            sharding.announce_key(self.base_seed, i)
            if i_next is not None:
                next_key = sharding.sample_key(self.base_seed, i_next)
        else:
            sharding.seed_for_sample(self.base_seed, i)
        idx = i % len(self.ds)
        segm = self.ds._segmentation(idx)
        name = self.ds._sub_ses_idx(idx)
        seeds = self.ds._seeds_for(name, idx)
        # device-resident hand-over: the fused warp writes the uint8 labels itself (no float32 labels, no conversion pass)
        out, seg, _img, params = self.ds.generator._pipeline(None, segm, seeds, {}, scale01=True, labels_u8=self._labels_u8,
                                                             next_key=next_key)
        return out, seg, name

    def _produce_batch(self, idxs):
        """B samples, each seeded with its own (base_seed, i) key: the generator's global RNGs are re-seeded between the
        samples' host draws, so `sample_batch` cannot be handed the whole list at once -- the draws of sample b are made
        by `_prepare` right after its seeding, through the items iterator."""
        ds, gen = self.ds, self.ds.generator
        names = []
        if gen._is_keyed():
            items = []
            for i in idxs:
                idx = i % len(ds)
                name = ds._sub_ses_idx(idx)
                names.append(name)
                items.append((None, ds._segmentation(idx), ds._seeds_for(name, idx)))
            out, seg, _imgs, _params = gen.sample_batch(items, scale01=True, streams=self.batch_streams, labels_u8=self._labels_u8,
                                                        keys=[sharding.sample_key(self.base_seed, i) for i in idxs])
            return out, seg, names

        def items():
            for i in idxs:
                sharding.seed_for_sample(self.base_seed, i)
                idx = i % len(ds)
                name = ds._sub_ses_idx(idx)
                names.append(name)
                yield (None, ds._segmentation(idx), ds._seeds_for(name, idx))

        out, seg, _imgs, _params = gen.sample_batch(items(), scale01=True, streams=self.batch_streams, lazy_items=len(idxs),
                                                    labels_u8=self._labels_u8)
        return out, seg, names

    def _iter_batches(self):
        B = int(self.batch_size)
        chunks = [self.indices[s:s + B] for s in range(0, len(self.indices), B)]
        if not self.to_host:
            for ch in chunks:
                out, seg, names = self._produce_batch(ch)
                yield {"image": out.unsqueeze(1), "label": seg.unsqueeze(1), "name": names}
            return
        pending = deque()
        stagers = {}  # a ragged last batch gets its own (smaller) ring
        for ch in chunks:
            out, seg, names = self._produce_batch(ch)
            st = stagers.get(len(ch))
            if st is None:
                st = stagers[len(ch)] = HostStager(tuple(out.shape[1:]), out.device, self.depth, self.label_dtype, self.keep,
                                                   batch=len(ch), image_dtype=self.image_dtype)
            if len(pending) >= self.depth:
                s_, t_, n_ = pending.popleft()
                img, lab = s_.collect(t_)
                yield {"image": img, "label": lab, "name": n_}
            pending.append((st, st.submit(out.unsqueeze(1), seg.unsqueeze(1)), names))
        while pending:
            s_, t_, n_ = pending.popleft()
            img, lab = s_.collect(t_)
            yield {"image": img, "label": lab, "name": n_}

    def __iter__(self):
        if self.batch_size:
            yield from self._iter_batches()
            return
        order = list(self.indices)
        following = order[1:] + [None]
        if not self.to_host:
            for i, i_next in zip(order, following):
                out, seg, name = self._produce(i, i_next)
                yield {"image": out.unsqueeze(0), "label": seg.unsqueeze(0), "name": name}
            return
        pending = deque()
        for i, i_next in zip(order, following):
            out, seg, name = self._produce(i, i_next)
            if self._stager is None:
                self._stager = HostStager(tuple(out.shape), out.device, self.depth, self.label_dtype, self.keep,
                                          image_dtype=self.image_dtype)
            if len(pending) >= self.depth:
                t, n = pending.popleft()
                img, lab = self._stager.collect(t)
                yield {"image": img, "label": lab, "name": n}
            pending.append((self._stager.submit(out, seg), name))
        while pending:
            t, n = pending.popleft()
            img, lab = self._stager.collect(t)
            yield {"image": img, "label": lab, "name": n}
