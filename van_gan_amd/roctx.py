"""roctx ranges around the phases of a train step ("build adds roctx ranges"; the reference has no tracing).

Off unless VG_ROCTX=1 (two C calls per phase are free, but the marker library should only be mapped when a profiler asked for it):
`rocprofv3 --kernel-trace --marker-trace -- python bench.py ...` then shows, per step, the host intervals in which each phase was
ENQUEUED (the kernels themselves run asynchronously on the lanes' streams; tools/trace_streams.py joins the two by time)."""
from __future__ import annotations

import ctypes
import os

_lib = None
ON = os.environ.get('VG_ROCTX', '0') == '1'
_depth = 0


def _load():
    global _lib, ON
    if _lib is None:
        for name in ('librocprofiler-sdk-roctx.so', 'libroctx64.so'):
            try:
                _lib = ctypes.CDLL(os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'lib', name))
                _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                break
            except OSError:
                continue
        else:
            raise RuntimeError('VG_ROCTX=1 but neither librocprofiler-sdk-roctx.so nor libroctx64.so can be loaded')
    return _lib


def phase(name: str):
    """Close the open phase range (if any) and open `name`; phase(None) only closes."""
    global _depth
    if not ON:
        return
    lib = _load()
    if _depth:
        lib.roctxRangePop()
        _depth = 0
    if name is not None:
        lib.roctxRangePushA(name.encode())
        _depth = 1
