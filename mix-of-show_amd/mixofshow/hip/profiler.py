"""Per-kernel timing of libmos_hip's kernels with HIP events recorded on the launch stream
(mos_profile_begin / mos_profile_end / mos_profile_get, include/mos_hip.h). Used by bench.py for `roofline`."""
import ctypes
from contextlib import contextmanager

from . import lib as _lib


def begin():
    _lib.check(_lib.load().mos_profile_begin(), 'mos_profile_begin')


def end():
    """Returns a list of dicts: name, calls, total_ms, avg_us, flops (algorithmic, per call), bytes (per call)."""
    L = _lib.load()
    n = L.mos_profile_end()
    out = []
    buf = ctypes.create_string_buffer(256)
    for i in range(n):
        ms, fl, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        calls = ctypes.c_longlong()
        _lib.check(L.mos_profile_get(i, buf, 256, ctypes.byref(ms), ctypes.byref(calls), ctypes.byref(fl),
                                     ctypes.byref(by)), 'mos_profile_get')
        c = max(1, calls.value)
        out.append(dict(name=buf.value.decode(), calls=calls.value, total_ms=ms.value, avg_us=ms.value * 1e3 / c,
                        flops=fl.value / c, bytes=by.value / c))
    out.sort(key=lambda r: -r['total_ms'])
    return out


@contextmanager
def profile(result):
    begin()
    try:
        yield
    finally:
        result.extend(end())
