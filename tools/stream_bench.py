#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: sustained volumes/s of a streaming epoch into a consumer that touches every
image, for (a) the reference's hand-over (synchronous `.cpu()` of float32 image + int64 labels),
(b) the same contract through the pinned double-buffered stager, (c) pinned stager with uint8 labels,
(d) device-resident outputs.  python tools/stream_bench.py [--size 256] [--n 60]"""
import argparse, json, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
from fetalsyngen_amd.data.datasets import SeedBank
from fetalsyngen_amd.data.staging import PrefetchingStream
from fetalsyngen_amd.phantom import make_seed_volumes

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--n", type=int, default=60)
args = ap.parse_args()
dev = "cuda:0"
shape = (args.size,) * 3


class MiniDataset:
    """FetalSynthDataset's device-side state without the file system (4 cached subjects)."""
    def __init__(self):
        self.generator = bench.build_generator(shape, dev, "device")
        self.generator.prewarm()
        self._banks, self._segs = [], []
        for v in range(4):
            seg, seeds = make_seed_volumes(shape, v)
            self._banks.append(SeedBank(seeds, dev)); self._segs.append(torch.from_numpy(seg).to(dev))
    def __len__(self): return 4
    def _segmentation(self, idx): return self._segs[idx]
    def _sub_ses_idx(self, idx): return f"sub-{idx}"
    def _seeds_for(self, name, idx=None): return self._banks[int(name.split("-")[1])]

ds = MiniDataset()

def consume(it):
    acc, n = 0.0, 0
    for item in it:
        acc += float(item["image"].reshape(-1)[::4097].sum())  # touch the data where it lives
        n += 1
    return n

def reference_handover():
    s = PrefetchingStream(ds, range(args.n), to_host=False)
    for item in s:
        yield {"image": item["image"].cpu(), "label": item["label"].cpu().long()}

modes = {
    "a_sync_cpu_f32_i64 (reference hand-over)": reference_handover,
    "b_pinned_stager_f32_i64": lambda: PrefetchingStream(ds, range(args.n), to_host=True, depth=3),
    "c_pinned_stager_f32_u8": lambda: PrefetchingStream(ds, range(args.n), to_host=True, depth=3, label_dtype=torch.uint8),
    "d_device_resident": lambda: PrefetchingStream(ds, range(args.n), to_host=False),
}
res = {}
for name, mk in modes.items():
    consume(list(zip(range(4), mk())) and [])  # warm-up a few samples
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = consume(mk())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[name] = round(n / dt, 1)
print(json.dumps({"stream_bench": {"size": args.size, "n": args.n, "volumes_per_s": res}}))
