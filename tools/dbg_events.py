import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch, numpy as np
import bench
from fetalsyngen_amd import _lib, sharding
from fetalsyngen_amd.data.datasets import SeedBank
from fetalsyngen_amd.phantom import make_seed_volumes
shape=(64,64,64)
seg, seeds = make_seed_volumes(shape)
bank, segd = SeedBank(seeds,"cuda:0"), torch.from_numpy(seg).to("cuda:0")
gen = bench.build_generator(shape,"cuda:0","keyed")
gen.blur_events=[]; gen.blur_events_every=1
for i in range(3):
    gen._pipeline(None, segd, bank, {}, scale01=True, key=sharding.sample_key(1,i))
torch.cuda.synchronize()
lib=_lib.load(); ms=ctypes.c_float()
print("n events", len(gen.blur_events))
for e0,e1,pl in gen.blur_events:
    print(e0, e1, pl, lib.fsg_event_elapsed_ms(e0,e1,ctypes.byref(ms)), ms.value)
