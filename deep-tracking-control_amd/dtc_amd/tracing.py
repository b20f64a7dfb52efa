"""Optional roctx ranges around the phases of the hot path (DTC_ROCTX=1): `rocprofv3 --marker-trace` then shows the
planner / compute_returns / per-mini-batch VAE and policy steps as named ranges above the kernel rows.  Off by default:
a no-op context manager, no library is loaded."""
from __future__ import annotations

import contextlib
import ctypes
import os

_ON = os.environ.get("DTC_ROCTX", "0") == "1"
_lib = None


def _load():
    global _lib, _ON
    if _lib is None and _ON:
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                _lib = ctypes.CDLL(name)
                _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                break
            except OSError:
                _lib = None
        if _lib is None:
            _ON = False
    return _lib


@contextlib.contextmanager
def span(name: str):
    lib = _load() if _ON else None
    if lib is None:
        yield
        return
    lib.roctxRangePushA(name.encode())
    try:
        yield
    finally:
        lib.roctxRangePop()
