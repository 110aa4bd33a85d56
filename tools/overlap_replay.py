"""GPU-side value of fsg_sample_plan::overlap with the host out of the way: ONE prepared sample's packed plan is replayed
N times through fsg_sample_pack_run (host cost ~10 us per call), with the upload + head on the launch stream (overlap 0)
or on the library's side stream beside the previous call's resampling tail (overlap 1).  python tools/overlap_replay.py"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from bench import build_generator
from fetalsyngen_amd import sharding, tables as T, kernels as K, _lib
from fetalsyngen_amd.data.datasets import SeedBank
from fetalsyngen_amd.phantom import make_seed_volumes
from fetalsyngen_amd.generator import model as M

dev = "cuda:0"; shape = (256,) * 3; N = 300
seg, seeds = make_seed_volumes(shape); bank, segd = SeedBank(seeds, dev), torch.from_numpy(seg).to(dev)
gen = build_generator(shape, dev, "device"); gen.prewarm()
for i in range(5):
    sharding.seed_for_sample(1, i); gen._pipeline(None, segd, bank, {}, scale01=True)
torch.cuda.synchronize()
lib = _lib.load()
I = gen._I
res = {}
for mode in (0, 1, 0, 1):
    # one prepared sample, staged into two alternating (pinned slot, device block) pairs
    sharding.seed_for_sample(1, 77)
    K._EPOCH[0] += 1
    with M._rng.use(gen.rng):
        arena = T.Arena()
        c = gen._prepare(None, segd, bank, {}, arena)
    c.arena_early = True
    blocks = [torch.empty(1 << 16, dtype=torch.uint8, device=dev) for _ in range(2)]
    pinned = [torch.empty(1 << 16, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    assert arena.stage(dev, blocks[0])
    ring, slot, hp, nbytes = arena.pending
    src = np.frombuffer((__import__("ctypes").c_uint8 * nbytes).from_address(hp), dtype=np.uint8)
    for q in range(2):
        pinned[q].numpy()[:nbytes] = src
    assert gen._fast_operands(c)
    f2 = int(c.sb.pending[1][2]); b2 = int(c.bplan.grid.shape[2])
    ws = gen._workspace(c.shape, 3 * f2 + b2)
    out = torch.empty(c.shape, dtype=torch.float32, device=dev); seg_out = torch.empty_like(c.seg)
    assert gen._flat_plan(c, True, out, seg_out, ws, None)
    arena.flush(False)
    fb = gen._flat
    iv = fb["iv"]
    base0 = blocks[0].data_ptr()
    ptr_idx = [k for k in range(I["COUNT"]) if base0 <= iv[k] < base0 + (1 << 16)]  # every pointer into the arena block
    offs = {k: int(iv[k]) - base0 for k in ptr_idx}
    torch.cuda.synchronize()
    seq = int(iv[I["WS_SEQ"]])
    t0 = time.perf_counter()
    for it in range(N):
        q = it & 1
        b = blocks[q].data_ptr()
        for k in ptr_idx:
            iv[k] = b + offs[k]
        iv[I["ARENA_HOST"]], iv[I["ARENA_DEV"]], iv[I["ARENA_BYTES"]] = pinned[q].data_ptr(), b, nbytes
        seq += 1
        iv[I["OVERLAP"]], iv[I["WS_SEQ"]] = mode, seq
        rc = lib.fsg_sample_pack_run(fb["ivp"], I["COUNT"], fb["fvp"], 17, fb["tbp"], K._stream(torch.device(dev)))
        assert rc == 0, rc
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    res.setdefault(mode, []).append({"host_us": round((t1 - t0) / N * 1e6, 1), "wall_us": round((t2 - t0) / N * 1e6, 1),
                                    "checksum": float(out.double().sum())})
print(json.dumps({"overlap_replay": res}))
