"""ctypes binding of libsyncvsr_hip.so.  Signatures are derived from include/syncvsr_hip.h, the single source
of truth for the C ABI.  There is no fallback: if the library is missing the product path raises."""
from __future__ import annotations

import ctypes
import os
import re
from functools import lru_cache

PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(PKG), "include", "syncvsr_hip.h")
# SVSR_LIB_VARIANT=syncdbg: the test-only build whose counted LDS-DMA waits are all vmcnt(0) (csrc/common.h SVSR_SYNC_DEBUG; built by
# `python -m syncvsr_amd.build --variant syncdbg`, loaded by the child process of tests/test_gpu_syncdbg.py).  Anything else: the product library.
_VARIANT = os.environ.get("SVSR_LIB_VARIANT", "")
LIB_PATH = os.path.join(PKG, f"libsyncvsr_hip_{_VARIANT}.so" if _VARIANT else "libsyncvsr_hip.so")

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "int64_t": ctypes.c_int64,
    "unsigned": ctypes.c_uint,
    "hipStream_t": ctypes.c_void_p,
}


class SvsrError(RuntimeError):
    pass


def parse_header(path: str = HEADER) -> dict[str, list[tuple[str, str]]]:
    """-> {function name: [(c type, arg name), ...]} for every `int svsr_*(...);` declaration."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out: dict[str, list[tuple[str, str]]] = {}
    for m in re.finditer(r"\b(int|int64_t|void\*)\s+(svsr_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in m.group(3).split(","):
            a = " ".join(a.split())
            if a in ("void", ""):
                continue
            mm = re.match(r"(.+?)\s*(\w+)$", a)
            args.append((mm.group(1).strip(), mm.group(2)))
        out[m.group(2)] = args
        _RESTYPE[m.group(2)] = m.group(1)
    return out


_RESTYPE: dict[str, str] = {}


def _ctype(ctype: str):
    if "*" in ctype:
        return ctypes.c_void_p
    return _CTYPES[ctype.replace("const", "").strip()]


@lru_cache(maxsize=1)
def load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise SvsrError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m syncvsr_amd.build` or __graft_entry__.build()); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in parse_header().items():
        fn = getattr(lib, name)
        fn.restype = _ctype(_RESTYPE.get(name, "int"))
        fn.argtypes = [_ctype(t) for t, _ in args]
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise SvsrError(f"{what} failed with code {rc}" + (" (unsupported shape/argument)" if rc == 1001 else " (hipError_t)"))
