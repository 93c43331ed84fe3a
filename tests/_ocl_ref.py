"""Runs the REFERENCE's own OpenCL kernels (kernels/ocl_program.cl, compiled as it stands to gfx950 code objects by
oracle/build_ref_ocl.py -> oracle/_ref/ocl_<config>_<flavour>.co) on the GPU through the HIP module API.

Test infrastructure: the golden generator (tests/golden/make_ocl_golden.py) and the live `-m gpu` checks use it; the
product never does.  Nothing here reads /root/reference - the code objects were built in the build container.

A kernel is launched the way the reference's host enqueues it: global size = populationLength work-items (the window
kernel: audioLength), local size = WRKGRPSIZE (Evolutionary_Strategy_OpenCL.hpp:471-533), explicit arguments in the
order of the kernel's signature.  The hidden OpenCL arguments (block counts, group sizes, global offsets) are filled in
by the HIP runtime from the code object's metadata.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

_hip = None


def hip():
    global _hip
    if _hip is None:
        # the HIP runtime this process already runs on (libsots_hip.so, or torch's bundled copy, brought one in): a second
        # runtime beside it would not share its device context - and may not even load against the first one's ROCr
        path = "/opt/rocm/lib/libamdhip64.so"
        with open("/proc/self/maps") as maps:
            for line in maps:
                if "libamdhip64.so" in line:
                    path = line.split()[-1]
                    break
        _hip = ctypes.CDLL(path)
        _hip.hipGetErrorString.restype = ctypes.c_char_p
    return _hip


def check(err, what):
    if err != 0:
        raise RuntimeError("%s: hip error %d (%s)" % (what, err, hip().hipGetErrorString(err).decode()))


def code_object(tag, flavour):
    return os.path.join(REF_DIR, "ocl_%s_%s.co" % (tag, flavour))


class DeviceBuffer:
    """hipMalloc'd bytes with numpy in / out."""

    def __init__(self, array=None, nbytes=None):
        if array is not None:
            array = np.ascontiguousarray(array)
            nbytes = array.nbytes
        self.nbytes = int(nbytes)
        self.ptr = ctypes.c_void_p()
        check(hip().hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(max(self.nbytes, 4))), "hipMalloc")
        if array is not None:
            check(hip().hipMemcpy(self.ptr, array.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(self.nbytes), 1), "hipMemcpy H2D")
        else:
            check(hip().hipMemset(self.ptr, 0, ctypes.c_size_t(max(self.nbytes, 4))), "hipMemset")

    def read(self, dtype, shape=None):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        check(hip().hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), self.ptr, ctypes.c_size_t(self.nbytes), 2), "hipMemcpy D2H")
        return out if shape is None else out.reshape(shape)

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = ctypes.c_void_p()


class RefProgram:
    """One compiled configuration of the reference's program."""

    def __init__(self, tag, flavour):
        self.path = code_object(tag, flavour)
        if not os.path.exists(self.path):
            raise FileNotFoundError(self.path + " (python oracle/build_ref_ocl.py in the build container makes it)")
        self.module = ctypes.c_void_p()
        check(hip().hipModuleLoad(ctypes.byref(self.module), self.path.encode()), "hipModuleLoad " + self.path)
        self.functions = {}

    def launch(self, name, global_size, local_size, buffers):
        """buffers: DeviceBuffer per explicit kernel argument (every argument of every kernel is a pointer)."""
        assert global_size % local_size == 0 and 1 <= local_size <= 256
        if name not in self.functions:
            f = ctypes.c_void_p()
            check(hip().hipModuleGetFunction(ctypes.byref(f), self.module, name.encode()), "hipModuleGetFunction " + name)
            self.functions[name] = f
        args = [ctypes.c_void_p(b.ptr.value) for b in buffers]
        params = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
        check(hip().hipModuleLaunchKernel(self.functions[name], global_size // local_size, 1, 1, local_size, 1, 1, 0, None, params, None),
              "hipModuleLaunchKernel " + name)
        check(hip().hipDeviceSynchronize(), "hipDeviceSynchronize after " + name)

    def unload(self):
        if self.module:
            hip().hipModuleUnload(self.module)
            self.module = ctypes.c_void_p()
