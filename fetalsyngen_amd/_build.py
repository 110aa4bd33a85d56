"""Build recipe for libfsg_hip.so (hipcc, gfx950 only, in-tree so it travels with `gpurun`)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libfsg_hip.so"
SOURCES = ["fsg_deform.hip", "fsg_warp_lean.hip", "fsg_zoom.hip", "fsg_intensity.hip", "fsg_blur.hip", "fsg_blur_rs.hip", "fsg_reduce.hip", "fsg_slice_acq.hip", "fsg_artifacts.hip", "fsg_keyed.hip", "fsg_codes.hip",
           "fsg_pipeline.cpp"]
EXTRA = os.environ.get("FSG_EXTRA_FLAGS", "").split()
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libfsg_hip.so cannot be built")
    return exe


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "fsg_common.h", PKG.parent / "include" / "fsg_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cc = hipcc()
    objs, jobs = [], []
    build_dir = PKG / "build"
    build_dir.mkdir(exist_ok=True)
    for s in SOURCES:
        o = build_dir / (s + ".o")
        lang = ["-x", "hip"] if s.endswith(".cpp") else []
        cmd = [cc, *FLAGS, *EXTRA, *lang, "-c", str(CSRC / s), "-o", str(o)]
        objs.append(str(o))
        deps = [CSRC / s, CSRC / "fsg_common.h", PKG.parent / "include" / "fsg_hip.h", Path(__file__)]
        if not force and not EXTRA and o.exists() and all(d.stat().st_mtime < o.stat().st_mtime for d in deps):
            continue  # object is newer than its source and the shared headers
        jobs.append(cmd)
    # the translation units are independent: compile up to four at a time (hipcc is single-threaded per file)
    from concurrent.futures import ThreadPoolExecutor

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(len(jobs), 1))) as pool:
        list(pool.map(run, jobs))
    tmp = LIB.with_suffix(".so.tmp")
    subprocess.run([cc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", str(tmp)], check=True)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
