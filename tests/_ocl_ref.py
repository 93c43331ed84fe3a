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

    def launch(self, name, global_size, local_size, buffers, sync=True):
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
        if sync:
            check(hip().hipDeviceSynchronize(), "hipDeviceSynchronize after " + name)

    def time_ms(self, name, global_size, local_size, buffers, launches=3):
        """mean duration of `launches` back-to-back launches between two HIP events on the null stream (after one warm-up)"""
        self.launch(name, global_size, local_size, buffers)
        start, stop = ctypes.c_void_p(), ctypes.c_void_p()
        check(hip().hipEventCreate(ctypes.byref(start)), "hipEventCreate")
        check(hip().hipEventCreate(ctypes.byref(stop)), "hipEventCreate")
        check(hip().hipEventRecord(start, None), "hipEventRecord")
        for _ in range(launches):
            self.launch(name, global_size, local_size, buffers, sync=False)
        check(hip().hipEventRecord(stop, None), "hipEventRecord")
        check(hip().hipEventSynchronize(stop), "hipEventSynchronize")
        ms = ctypes.c_float()
        check(hip().hipEventElapsedTime(ctypes.byref(ms), start, stop), "hipEventElapsedTime")
        hip().hipEventDestroy(start), hip().hipEventDestroy(stop)
        return ms.value / launches

    def unload(self):
        if self.module:
            hip().hipModuleUnload(self.module)
            self.module = ctypes.c_void_p()


class RefGenerationLoop:
    """The reference's generation loop (Evolutionary_Strategy_OpenCL.hpp:471-541) driven through its own kernels: recombine,
    mutate, synthesise, window, FFT, fitness, sort, rotate.  The FFT is the one stage that is not the reference's code (clFFT is
    not in the image): the windowed rows come to the host, numpy transforms them in double, and bins 0 .. N/2-1 go back as
    the interleaved fp32 rows of N + 8 floats the fitness kernel reads (the bins behind them stay zero, and so does the
    target behind N/2: the kernel's three extra bins - ocl_program.cl:607 - then add nothing, DESIGN 6 deviation 4)."""

    SYNTH = {4: "synthesisePopulation", 6: "synthesisePopulationDoubleSeries", 12: "synthesisePopulationTripleParallel"}

    def __init__(self, tag, flavour, wg, d, log2n, parents, offspring, pmin, pmax, table, target_mag):
        self.prog = RefProgram(tag, flavour)
        self.wg, self.d, self.n, self.p = wg, d, 1 << log2n, parents + offspring
        p, n = self.p, self.n
        self.rot = 0
        self.b_rot = DeviceBuffer(np.zeros(1, np.uint32))
        self.values, self.steps = DeviceBuffer(nbytes=2 * p * d * 4), DeviceBuffer(nbytes=2 * p * d * 4)
        self.fitness, self.states = DeviceBuffer(nbytes=2 * p * 4), DeviceBuffer(nbytes=p * 8)
        self.audio, self.spectrum = DeviceBuffer(nbytes=p * n * 4), DeviceBuffer(nbytes=p * (n + 8) * 4)
        tgt = np.zeros(n // 2 + 8, np.float32)
        tgt[: n // 2] = target_mag
        self.target = DeviceBuffer(tgt)
        self.pmin, self.pmax = DeviceBuffer(np.asarray(pmin, np.float32)), DeviceBuffer(np.asarray(pmax, np.float32))
        self.table = DeviceBuffer(np.concatenate([np.asarray(table, np.float32), np.zeros(64, np.float32)]))
        self.rows = np.zeros((p, n + 8), np.float32)

    def _write(self, buf, array):
        array = np.ascontiguousarray(array)
        check(hip().hipMemcpy(buf.ptr, array.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(array.nbytes), 1), "hipMemcpy H2D")

    def init(self, states):
        """states: uint32 [P][2] MWC64X (x, c) - the reference's host seeds them from the clock"""
        self.rot = 0
        self._write(self.b_rot, np.zeros(1, np.uint32))
        self._write(self.states, np.asarray(states, np.uint32))
        self.prog.launch("initPopulation", self.p, self.wg, [self.values, self.steps, self.fitness, self.states, self.b_rot])

    def generation(self):
        p, n, wg, L = self.p, self.n, self.wg, self.prog.launch
        L("recombinePopulation", p, wg, [self.values, self.steps, self.b_rot], sync=False)
        L("mutatePopulation", p, wg, [self.values, self.steps, self.states, self.b_rot], sync=False)
        L(self.SYNTH[self.d], p, wg, [self.audio, self.values, self.pmin, self.pmax, self.b_rot, self.table], sync=False)
        L("applyWindowPopulation", n, wg, [self.audio])
        z = np.fft.rfft(self.audio.read(np.float32, (p, n)).astype(np.float64), axis=1)[:, : n // 2]
        self.rows[:, 0:n:2], self.rows[:, 1:n:2] = z.real, z.imag
        self._write(self.spectrum, self.rows)
        L("fitnessPopulation", p, wg, [self.fitness, self.spectrum, self.target, self.b_rot], sync=False)
        L("sortPopulation", p, wg, [self.values, self.steps, self.fitness, self.b_rot])
        self.rot ^= 1
        self._write(self.b_rot, np.array([self.rot], np.uint32))

    def best_fitness(self):
        return float(self.fitness.read(np.float32, (2, self.p))[self.rot, 0])

    def population(self):
        v = self.values.read(np.float32, (2, self.p, self.d))[self.rot]
        s = self.steps.read(np.float32, (2, self.p, self.d))[self.rot]
        return v, s, self.fitness.read(np.float32, (2, self.p))[self.rot]

    def close(self):
        for b in (self.b_rot, self.values, self.steps, self.fitness, self.states, self.audio, self.spectrum, self.target, self.pmin, self.pmax, self.table):
            b.free()
        self.prog.unload()
