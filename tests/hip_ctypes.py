"""A caller's own device memory for the -m gpu tests: hipMalloc / hipMemcpy through ctypes on the HIP runtime the engine
itself is linked with (same library object in the process - a second HIP runtime, e.g. the one inside a PyTorch wheel,
would fight it for the device)."""
import ctypes as C

import numpy as np

_H2D, _D2H = 1, 2


def _runtime():
    from adcraft_amd import _ffi
    _ffi.lib()                                   # the engine (and with it its libamdhip64) is loaded first
    for name in ("libamdhip64.so.7", "libamdhip64.so"):
        try:
            return C.CDLL(name)
        except OSError:
            continue
    raise OSError("libamdhip64 not found")


class DeviceArray:
    """a device copy of a numpy array, owned by the test (the engine only borrows the pointer)"""

    def __init__(self, host):
        self._hip = _runtime()
        self.host = np.ascontiguousarray(host)
        p = C.c_void_p()
        assert self._hip.hipMalloc(C.byref(p), C.c_size_t(max(self.host.nbytes, 1))) == 0
        self.ptr = p.value
        assert self._hip.hipMemcpy(C.c_void_p(self.ptr), C.c_void_p(self.host.ctypes.data), C.c_size_t(self.host.nbytes), _H2D) == 0

    def free(self):
        if self.ptr:
            self._hip.hipFree(C.c_void_p(self.ptr))
            self.ptr = None


def read_device(ptr, nbytes, dtype, shape):
    """copy `nbytes` at device address `ptr` to a numpy array (what a zero-copy consumer would see)"""
    out = np.empty(shape, dtype=dtype)
    assert out.nbytes == nbytes
    hip = _runtime()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(nbytes), _D2H) == 0
    return out
