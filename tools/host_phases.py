"""Host time of one sample by phase (perf_counter around the phases of FetalSynthGen._pipeline), small volume so that
the GPU never back-pressures the host.  python tools/host_phases.py [--size 64] [--n 2000]"""
import argparse, sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch
from bench import build_generator
from fetalsyngen_amd import sharding, tables as T, kernels as K, _lib, rng as _rng
from fetalsyngen_amd.data.datasets import SeedBank
from fetalsyngen_amd.phantom import make_seed_volumes

ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=64); ap.add_argument("--n", type=int, default=2000)
a = ap.parse_args()
dev = "cuda:0"; shape = (a.size,) * 3
seg, seeds = make_seed_volumes(shape); bank, segd = SeedBank(seeds, dev), torch.from_numpy(seg).to(dev)
gen = build_generator(shape, dev, "device"); gen.prewarm()
for i in range(100):
    sharding.seed_for_sample(1, i); gen._pipeline(None, segd, bank, {}, scale01=True)
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
lib = _lib.load()
t_all = time.perf_counter()
for i in range(a.n):
    t = time.perf_counter()
    sharding.seed_for_sample(1, 100 + i); t = tick("seed_for_sample", t)
    with _rng.use(gen.rng):
        arena = T.Arena(); t = tick("arena_new", t)
        ig = gen.intensity_generator
        m2s = ig.draw_subclusters({}); parts = bank.parts(m2s); t = tick("draw_subclusters+parts", t)
        gp = ig.plan_intensities(shape, {}); t = tick("plan_intensities", t)
        plans = gen._draw_plans(shape, {}); t = tick("draw_plans(deform,gamma,bias,resample,noise)", t)
    # the rest through the real code path, timed as a whole minus the above (re-seeded so the draws repeat)
    sharding.seed_for_sample(1, 100 + i)
    t = time.perf_counter()
    with _rng.use(gen.rng):
        arena = T.Arena()
        c = gen._prepare(None, segd, bank, {}, arena); t = tick("_prepare(total incl. draws)", t)
        arena.upload(dev); t = tick("arena.upload", t)
        gen._resolve(c); t = tick("_resolve", t)
        gen._native_operands(c); ws = gen._workspace(c.shape, gen._rows_needed(c)); t = tick("operands+workspace", t)
        out = torch.empty(c.shape, dtype=torch.float32, device=dev); so = torch.empty_like(c.seg); t = tick("torch.empty x2", t)
        p = _lib.SamplePlan(); ok = gen._fill_native_plan(p, c, True, ws, out, so); t = tick("fill_native_plan", t)
        rc = lib.fsg_sample_run(C.byref(p), K._stream(torch.device(dev))); t = tick("fsg_sample_run (C: launches)", t)
        prm = gen._synth_params(c, {}); t = tick("synth_params", t)
tot = time.perf_counter() - t_all
torch.cuda.synchronize()
print(f"per-sample host phases, us (n={a.n}, size={a.size}); draws are executed twice in this harness")
for k, v in acc.items():
    print(f"  {k:48s} {v / a.n * 1e6:8.1f}")
print(f"  harness total per sample {tot / a.n * 1e6:.1f} us")
t0 = time.perf_counter()
for i in range(a.n):
    sharding.seed_for_sample(1, 100 + i); gen._pipeline(None, segd, bank, {}, scale01=True)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"_pipeline + seed: {(t1 - t0) / a.n * 1e6:.1f} us per sample (enqueue), {(time.perf_counter() - t0) / a.n * 1e6:.1f} incl. drain")

# ---- keyed mode (r03): the whole host side of a sample is the key, three allocations and ONE native call ----------------
from fetalsyngen_amd import keyed as _keyed
genk = build_generator(shape, dev, "keyed")
kc = genk.keyed_context(shape); kc.register_tables()
for i in range(100):
    genk._pipeline(None, segd, bank, {}, scale01=True, key=sharding.sample_key(1, i))
torch.cuda.synchronize()
acck = {}
def tickk(name, t0):
    t1 = time.perf_counter(); acck[name] = acck.get(name, 0.0) + (t1 - t0); return t1
twin = genk._label_twin(segd)
for i in range(a.n):
    t = time.perf_counter()
    key = sharding.sample_key(1, 100 + i); t = tickk("sample_key (splitmix64, Python ints)", t)
    ent = kc.subject(bank, segd, twin); ws = genk._workspace(shape, kc.rows_need); t = tickk("subject pointers + workspace", t)
    out = torch.empty(shape, dtype=torch.float32, device=dev); so = torch.empty(shape, dtype=torch.float32, device=dev)
    block = torch.empty(kc.block_bytes, dtype=torch.uint8, device=dev); t = tickk("torch.empty x3", t)
    iv = kc.iv
    iv[0] = key if key < (1 << 63) else key - (1 << 64); iv[1] = out.data_ptr(); iv[2], iv[3] = so.data_ptr(), 0
    iv[4], iv[5], iv[6] = ent[2], ent[1], block.data_ptr()
    iv[7], iv[8], iv[9] = ws["ws0"].data_ptr(), ws["ws1"].data_ptr(), ws["low"].data_ptr()
    iv[10], iv[11], iv[12] = (ws["rows"].data_ptr() if ws["rows"] is not None else 0), ws["stride"], 1
    iv[13] = iv[14] = iv[15] = 0; iv[16:80] = ent[0]; iv[80] = iv[81] = 0; t = tickk("fill the int64 argument array", t)
    d = _lib.KeyedDraws()
    rc = kc.lib.fsg_keyed_sample_run(kc.handle, kc.ivp, len(kc.iv), C.byref(d), K._stream(torch.device(dev))); t = tickk("fsg_keyed_sample_run (C: draws + 8 launches)", t)
    prm = _keyed.params_of(d, block); t = tickk("params_of (synth_params dict)", t)
torch.cuda.synchronize()
print(f"keyed mode, per-sample host phases, us (n={a.n}, size={a.size})")
for k, v in acck.items():
    print(f"  {k:48s} {v / a.n * 1e6:8.1f}")
t0 = time.perf_counter()
for i in range(a.n):
    genk._pipeline(None, segd, bank, {}, scale01=True, key=sharding.sample_key(1, 100 + i))
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"keyed _pipeline + key: {(t1 - t0) / a.n * 1e6:.1f} us per sample (enqueue), {(time.perf_counter() - t0) / a.n * 1e6:.1f} incl. drain")
