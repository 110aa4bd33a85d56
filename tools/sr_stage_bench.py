#!/usr/bin/env python3
"""Time the SR-artifact stages on one MI355X at BASELINE config 4 scale (384^3, 0.5 mm) -- whole-stage wall time with the
default YAML parameters, device RNG, CUDA-kernel slice-acquisition semantics.  One JSON line.

    python tools/sr_stage_bench.py [--size 384] [--reps 3] [--stages simulate_motion,blur_cortex,struct_noise,boundaries]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fetalsyngen_amd import rng  # noqa: E402
from fetalsyngen_amd.phantom import make_segmentation  # noqa: E402
from fetalsyngen_amd.generator.defaults import default_artifacts  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=384)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--stages", default="simulate_motion,blur_cortex,struct_noise,boundaries")
    a = ap.parse_args()
    dev = "cuda:0"
    rng.set_mode("device")
    shape = (a.size,) * 3
    seg = torch.from_numpy(make_segmentation(shape)[0].astype(np.float32)).to(dev)
    img = (0.1 * seg + 0.05 * torch.rand(shape, device=dev)) * (seg > 0)
    img = img / img.max()
    arts = default_artifacts(prob=1.0)
    out = {"shape": list(shape), "reps": a.reps}
    for name in a.stages.split(","):
        st = arts[name]
        ts, metas = [], []
        for rep in range(a.reps + 1):
            np.random.seed(100 + rep)
            torch.manual_seed(100 + rep)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            y, meta = st(img, seg, dev, {}, resolution=[0.5, 0.5, 0.5])
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            metas.append({k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in meta.items()
                          if isinstance(v, (int, float, np.floating, np.integer, bool, str)) or v is None})
        out[name] = {"ms": [round(t, 2) for t in ts[1:]], "first_call_ms": round(ts[0], 2), "meta_last": {k: (v if not isinstance(v, (np.integer,)) else int(v)) for k, v in metas[-1].items()}}
    print(json.dumps(out, default=lambda o: o.item() if hasattr(o, "item") else str(o)))


if __name__ == "__main__":
    main()
