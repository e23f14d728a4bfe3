"""Builds libpixelsplat_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pixelsplat_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpixelsplat_hip.so")

# (source, extra flags).  The preprocess TU carries the integer-valued arithmetic (radius,
# tile rect, depth key): no FMA contraction there so the bins are bit-reproducible.
SOURCES = [
    ("raster_preprocess.hip", ["-ffp-contract=off"]),
    ("epipolar_geometry.hip", ["-ffp-contract=off"]),
    ("raster_sort.hip", []),
    ("raster_tiles.hip", []),
    ("raster_backward.hip", []),
    ("raster_api.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
          "-I", CSRC, "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps(src: str) -> list[str]:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(ROOT, "include", "pixelsplat_hip.h"))
    return [src] + hdrs + [os.path.abspath(__file__)]


def unit_hashes(units, extra_flags=()) -> str:
    """"<file>:<hash12> ..." -- per translation unit, sha256 over its source, every shared header
    and its compile flags.  Baked into the library (ps_build_info) and recorded with every counter
    summary under profiles/, so that bench.py never pairs a kernel's live time with counters taken
    on different code."""
    import hashlib
    shared = hashlib.sha256()
    for h in sorted(f for f in os.listdir(CSRC) if f.endswith(".h")):
        shared.update(open(os.path.join(CSRC, h), "rb").read())
    shared.update(open(os.path.join(ROOT, "include", "pixelsplat_hip.h"), "rb").read())
    out = []
    for name, flags in units:
        h = hashlib.sha256(shared.digest())
        h.update(open(os.path.join(CSRC, name), "rb").read())
        h.update(" ".join([*COMMON[:4], *flags, *extra_flags]).encode())
        out.append(f"{name[:-4]}:{h.hexdigest()[:12]}")
    return " ".join(out)


def build(force: bool = False, verbose: bool = False) -> str:
    extra = [(f, []) for f in sorted(os.listdir(CSRC))
             if f.endswith(".hip") and f not in {s for s, _ in SOURCES}]
    objs = []
    rebuilt = False
    procs = []
    units = SOURCES + extra
    # every unit, raster_api.hip included: a unit's hash covers its source, the shared headers and its BASE flags,
    # not the -DPS_BUILD_HASHES stamp raster_api.hip is compiled with, so the stamp is not self-referential
    hashes = unit_hashes(units)
    stamp = os.path.join(CSRC, ".build_hashes")
    hashes_changed = (not os.path.exists(stamp)) or open(stamp).read() != hashes
    for name, flags in units:
        src = os.path.join(CSRC, name)
        obj = os.path.join(CSRC, name.replace(".hip", ".o"))
        objs.append(obj)
        stale = force or not os.path.exists(obj) or any(
            os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src))
        if name == "raster_api.hip":      # carries every unit's hash (ps_build_info)
            stale = stale or hashes_changed
            flags = [*flags, f'-DPS_BUILD_HASHES="{hashes}"']
        if stale:
            cmd = [_hipcc(), *COMMON, *flags, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE,
                                                 stderr=subprocess.STDOUT, text=True)))
            rebuilt = True
    for name, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{out}")
        if verbose and out.strip():
            print(out)
    if rebuilt or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(hashes)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
