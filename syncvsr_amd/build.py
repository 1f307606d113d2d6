"""Builds syncvsr_amd/libsyncvsr_hip.so from syncvsr_amd/csrc/*.hip with hipcc for gfx950 (in-tree).

    python -m syncvsr_amd.build [--force]
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libsyncvsr_hip.so")
OBJ = os.path.join(PKG, "csrc", "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-value", "-ffp-contract=fast"]


def _hipcc() -> str:
    for c in ("hipcc", "/opt/rocm/bin/hipcc"):
        try:
            subprocess.run([c, "--version"], capture_output=True, check=True)
            return c
        except (OSError, subprocess.CalledProcessError):
            continue
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers() -> list[str]:
    """Every header a source may include: csrc/*.h and the public C ABI (steplist.hip derives its call thunks from its declarations)."""
    pub = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(pub, "*.h"))


# build variants: "" = the product library; "syncdbg" = -DSVSR_SYNC_DEBUG (every counted s_waitcnt vmcnt(n) of the LDS-DMA pipelines is
# vmcnt(0): csrc/common.h), a TEST library that tests/test_gpu_syncdbg.py compares the product's outputs with, bit for bit
VARIANTS = {"": [], "syncdbg": ["-DSVSR_SYNC_DEBUG"], "p8stamp": ["-DSVSR_P8_STAMP"]}      # p8stamp: s_memtime stamps around k_igemm_p8's K loops (scripts/probes/p8_stamps.py)


def lib_path(variant: str = "") -> str:
    return os.path.join(PKG, f"libsyncvsr_hip_{variant}.so") if variant else LIB


def stale(variant: str = "") -> bool:
    lib = lib_path(variant)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + headers()
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, variant: str = "") -> str:
    LIB = lib_path(variant)
    OBJ = os.path.join(PKG, "csrc", "_obj" + ("_" + variant if variant else ""))
    if not force and not stale(variant):
        return LIB
    cc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    hdr_t = max([os.path.getmtime(h) for h in headers()] + [0.0])

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [cc, *FLAGS, *VARIANTS[variant], "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, variant=_variant))
