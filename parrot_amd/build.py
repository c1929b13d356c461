"""Builds libparrot_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

Usage: ``python -m parrot_amd.build [--force]``.  The library has no torch dependency: it is plain
``hipcc -shared -fPIC`` over parrot_amd/csrc/*.hip and is loaded with ctypes (parrot_amd/_lib.py).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libparrot_hip.so")
OBJDIR = os.path.join(CSRC, "build")
ARCH = "gfx950"

SOURCES = ["skinny.hip", "biggemm.hip", "attention.hip", "elementwise.hip", "quantize.hip",
           "plans.hip", "plans_decode.hip", "capi.hip", "samplernn.hip", "persist.hip", "sr_persist.hip", "rowgru.hip", "trainops.hip"]
EXTRA_FLAGS = {"quantize.hip": ["-ffp-contract=off"]}
if os.environ.get("PARROT_PM_DEPTH"):  # development: ring depth of the persistent machine's K loop
    EXTRA_FLAGS["persist.hip"] = ["-DPM_DEPTH=" + os.environ["PARROT_PM_DEPTH"]]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, timers: bool = False) -> str:
    """timers=True: the development build with in-kernel phase stamps (-DSK_TIMERS) -> libparrot_hip_timers.so, loaded
    instead of the product library when PARROT_LIB points at it (tools/att_timing.py)."""
    global OUT, OBJDIR
    if timers:
        OUT = os.path.join(HERE, "libparrot_hip_timers.so")
        OBJDIR = os.path.join(CSRC, "build_timers")
    tag, extra = os.environ.get("PARROT_BUILD_TAG"), os.environ.get("PARROT_BUILD_FLAGS", "").split()
    if tag:  # development: a variant library (compile-time knobs, e.g. -DWK_PB_DEPTH=4), loaded through PARROT_HIP_LIB
        OUT = os.path.join(HERE, f"libparrot_hip_{tag}.so")
        OBJDIR = os.path.join(CSRC, f"build_{tag}")
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(HERE, "..", "include", "parrot_hip.h"))
    stamp = os.path.join(OBJDIR, "stamp.txt")
    dig = _digest(deps)
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + (["-DSK_TIMERS"] if timers else []) + (extra if tag else [])

    def compile_one(src):
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        cmd = common + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"[parrot_amd.build] built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, timers="--timers" in sys.argv)
