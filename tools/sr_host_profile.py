#!/usr/bin/env python3
"""cProfile of SimulateMotion's host side at 384^3 (5 calls after a warm-up): top functions by own time and by cumulative time."""
import cProfile
import io
import pstats
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from fetalsyngen_amd import rng  # noqa: E402
from fetalsyngen_amd.generator.defaults import default_artifacts  # noqa: E402
from fetalsyngen_amd.phantom import make_segmentation  # noqa: E402

dev = "cuda:0"
rng.set_mode("device")
shape = (384,) * 3
seg = torch.from_numpy(make_segmentation(shape)[0].astype(np.float32)).to(dev)
img = (0.1 * seg + 0.05 * torch.rand(shape, device=dev)) * (seg > 0)
img = img / img.max()
st = default_artifacts(prob=1.0)["simulate_motion"]
for rep in range(2):
    np.random.seed(100 + rep); torch.manual_seed(100 + rep)
    st(img, seg, dev, {}, resolution=[0.5, 0.5, 0.5])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for rep in range(2, 7):
    np.random.seed(100 + rep); torch.manual_seed(100 + rep)
    st(img, seg, dev, {}, resolution=[0.5, 0.5, 0.5])
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).strip_dirs().sort_stats(key).print_stats(28)
    print(buf.getvalue())
