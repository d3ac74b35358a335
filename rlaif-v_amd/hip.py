"""ctypes binding of librlaifv_hip.so (C ABI declared in include/rlaifv_hip.h).

The argument types are parsed from the header itself so the binding cannot drift from the ABI.
There is NO fallback: if the library is missing or fails to load, importing callers get a loud
RuntimeError (the product path never routes through the CPU oracle or eager PyTorch).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("RV_HIP_LIB") or os.path.join(_HERE, "librlaifv_hip.so")   # RV_HIP_LIB: A/B experiments only
_HEADER = os.path.normpath(os.path.join(_HERE, "..", "include", "rlaifv_hip.h"))

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float}


def parse_header(path: str = _HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """{name: (return type, [(ctype string, arg name), ...])} for every ``rv_*`` declaration."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(const char\*|int|long)\s+(rv_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                parsed.append((mm.group(1).strip(), mm.group(2)))
        out[name] = (ret, parsed)
    return out


def _to_ctype(t: str):
    if "*" in t:
        return ctypes.c_void_p
    return _CTYPES[t.replace("const ", "").strip()]


class HipLib:
    def __init__(self, path: str = _LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the DPO hot path.")
        self.path = path
        self.lib = ctypes.CDLL(path)
        self.decls = parse_header()
        for name, (ret, args) in self.decls.items():
            fn = getattr(self.lib, name)          # AttributeError = header/library drift, fail loudly
            fn.restype = ctypes.c_char_p if ret.startswith("const char") else (ctypes.c_long if ret == "long" else ctypes.c_int)
            fn.argtypes = [_to_ctype(t) for t, _ in args]
        if self.lib.rv_abi_version() != 7:
            raise RuntimeError("librlaifv_hip.so ABI version mismatch")

    def last_error(self) -> str:
        return (self.lib.rv_last_error() or b"").decode()

    def call(self, name: str, *args) -> None:
        fn = getattr(self.lib, name)
        conv = []
        for a in args:
            if a is None:
                conv.append(None)
            elif hasattr(a, "data_ptr"):
                conv.append(a.data_ptr())
            else:
                conv.append(a)
        rc = fn(*conv)
        if rc != 0:
            raise RuntimeError(f"{name} failed (rc={rc}): {self.last_error()}")


_LIB = None


def lib() -> HipLib:
    global _LIB
    if _LIB is None:
        _LIB = HipLib()
    return _LIB


def stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream on ``device`` (default: the current device)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


_DEBUG_ARGS = os.environ.get("RV_DEBUG_ARGS", "0") != "0"


def call(name: str, *args) -> None:
    """Launch ``name`` on torch's current HIP stream OF THE TENSORS' DEVICE (appended as the last argument).  A kernel
    launch goes to the calling thread's current HIP device, so when the first tensor argument lives on another device the
    call is wrapped in ``torch.cuda.device`` - raw pointers of cuda:1 never meet a stream of cuda:0.  RV_DEBUG_ARGS=1
    additionally checks that every tensor argument is on that device and dense in its last dimension."""
    import torch
    dev = None
    for a in args:
        if hasattr(a, "data_ptr"):
            if dev is None:
                dev = a.device
                if not _DEBUG_ARGS:
                    break
            elif a.device != dev:
                raise ValueError(f"{name}: tensor arguments on different devices ({dev} vs {a.device})")
            if _DEBUG_ARGS and a.dim() > 0 and a.stride(-1) != 1 and a.numel() > 1:
                raise ValueError(f"{name}: tensor argument with non-unit innermost stride {a.stride()}")
    if dev is None or dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
        lib().call(name, *args, stream_ptr(dev if dev is not None and dev.type == "cuda" else None))
        return
    with torch.cuda.device(dev):
        lib().call(name, *args, stream_ptr(dev))
