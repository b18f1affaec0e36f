"""Device buffers for tests that hand DEVICE pointers to the C ABI: hipMalloc / hipMemcpy / hipFree through ctypes on
the HIP runtime the product library is linked against (same soname, so the loader hands back the copy that is
already mapped).  PyTorch is not used for this: its wheel carries a second copy of the runtime, and two runtimes in
one process cannot both open the GPU."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        import ptts_lib
        ptts_lib.load()                       # maps libamdhip64.so.7 first
        _hip = C.CDLL("libamdhip64.so.7")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipDeviceSynchronize.argtypes = []
    return _hip


class DevArray:
    """float32 array in device memory."""

    def __init__(self, shape, fill_byte=None):
        self.shape = tuple(shape)
        self.nbytes = int(np.prod(self.shape)) * 4
        p = C.c_void_p()
        rc = hip().hipMalloc(C.byref(p), max(self.nbytes, 4))
        if rc != 0 or not p.value:
            raise RuntimeError(f"hipMalloc failed ({rc})")
        self.ptr = p.value
        if fill_byte is not None:
            assert hip().hipMemset(self.ptr, fill_byte, self.nbytes) == 0

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a, np.float32)
        d = cls(a.shape)
        assert hip().hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, 1) == 0      # hipMemcpyHostToDevice
        return d

    def numpy(self):
        out = np.empty(self.shape, np.float32)
        assert hip().hipDeviceSynchronize() == 0
        assert hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0   # hipMemcpyDeviceToHost
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
