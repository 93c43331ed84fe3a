"""ctypes binding of libsots_hip.so (include/sots_hip.h).

Thin by design: every method is one C-ABI call.  There is no CPU fallback; if the
shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsots_hip.so")

MAX_DIMS = 16
WAVETABLE_SIZE = 32768
SAMPLE_RATE = 44100

SYNTH_2OP, SYNTH_3OP_SERIES, SYNTH_TRIPLE_PAR, SYNTH_4OP_SERIES = 0, 1, 2, 3
SYNTH_DIMS = {SYNTH_2OP: 4, SYNTH_3OP_SERIES: 6, SYNTH_TRIPLE_PAR: 12, SYNTH_4OP_SERIES: 8}
SYNTH_NAMES = {"2op": SYNTH_2OP, "3op_series": SYNTH_3OP_SERIES,
               "triple_parallel": SYNTH_TRIPLE_PAR, "4op_series": SYNTH_4OP_SERIES}

(STAGE_INIT, STAGE_RECOMBINE, STAGE_MUTATE, STAGE_SYNTHESISE, STAGE_WINDOW, STAGE_FFT,
 STAGE_FITNESS, STAGE_SORT, STAGE_ROTATE, STAGE_FUSED_VARIATION, STAGE_FUSED_SYNTH,
 STAGE_FUSED_SPECTRAL, STAGE_SORT_TAIL, STAGE_COUNT) = range(14)
SORT_LAZY_TAIL, SORT_FULL, SORT_TOP_ONLY = 0, 1, 2
ARITH_CPU_PATH, ARITH_DEVICE_KERNELS = 0, 1

# the reference's Benchmarker timer names, Evolutionary_Strategy_OpenCL.hpp:117
STAGE_NAMES = ["initPopulation", "recombinePopulation", "mutatePopulation", "synthesisePopulation",
               "applyWindowPopulation", "hipFFT", "fitnessPopulation", "sortPopulation",
               "rotatePopulation", "fused:recombine+mutate", "fused:synthesise+window",
               "fused:FFT+fitness", "sortPopulation:tail"]

EXPORTS = [
    "sots_create", "sots_destroy", "sots_last_error", "sots_set_stream", "sots_synchronize",
    "sots_set_target_audio", "sots_set_target_spectrum", "sots_init_population",
    "sots_write_population", "sots_read_population", "sots_read_population_other",
    "sots_write_synth", "sots_read_synth",
    "sots_stage_recombine", "sots_stage_mutate", "sots_stage_synthesise", "sots_stage_window",
    "sots_stage_fft", "sots_stage_fitness", "sots_stage_sort", "sots_stage_select", "sots_stage_rotate",
    "sots_set_sort_mode",
    "sots_set_synth_arithmetic",
    "sots_execute_generation", "sots_execute_generations", "sots_get_generation",
    "sots_set_generation", "sots_timing_enable", "sots_timing_reset", "sots_stage_time_ms",
    "sots_stage_launch_times_ms",
    "sots_pack_elites_device", "sots_inject_immigrants_device", "sots_inject_gathered_device", "sots_fuse_exchange_next_sort",
    "sots_pack_elites_host",
    "sots_inject_immigrants_host", "sots_get_info",
    "sots_group_create", "sots_group_destroy", "sots_group_last_error", "sots_group_size", "sots_group_uses_rccl",
    "sots_group_island", "sots_group_set_target_audio", "sots_group_set_target_spectrum", "sots_group_init_population",
    "sots_group_execute_generations", "sots_group_synchronize", "sots_group_best",
]
GROUP_OVERLAP, GROUP_FORCE_RCCL, GROUP_UNFUSED, GROUP_EVENT_WAITS = 1, 2, 4, 8
MAX_GROUP_DEVICES = 16


class SotsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libsots_hip error {code}: {text}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("num_parents", C.c_uint32), ("num_offspring", C.c_uint32),
        ("num_dimensions", C.c_uint32), ("audio_length_log2", C.c_uint32),
        ("num_generations", C.c_uint32), ("synth_kind", C.c_uint32), ("workgroup_size", C.c_uint32),
        ("device", C.c_int32), ("gid_base", C.c_uint32), ("seed", C.c_uint64),
        ("param_min", C.c_float * MAX_DIMS), ("param_max", C.c_float * MAX_DIMS),
    ]


class Info(C.Structure):
    _fields_ = [
        ("population_length", C.c_uint32), ("num_dimensions", C.c_uint32),
        ("audio_length", C.c_uint32), ("spectrum_row_floats", C.c_uint32),
        ("rotation_index", C.c_uint32), ("generation", C.c_uint32),
        ("compute_units", C.c_uint32), ("reserved", C.c_uint32),
        ("device_name", C.c_char * 128), ("arch", C.c_char * 32),
    ]


_lib = None


def load():
    """dlopen libsots_hip.so.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SOTS_LIB_PATH", LIB_PATH)  # development override (kernel variants)
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C <package dir>`; there is no CPU fallback")
    L = C.CDLL(path)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    L.sots_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.sots_destroy.argtypes = [vp]
    L.sots_destroy.restype = None
    L.sots_last_error.argtypes = [vp]
    L.sots_last_error.restype = C.c_char_p
    L.sots_set_stream.argtypes = [vp, vp]
    L.sots_synchronize.argtypes = [vp]
    L.sots_set_target_audio.argtypes = [vp, vp, u32]
    L.sots_set_target_spectrum.argtypes = [vp, vp, u32]
    L.sots_init_population.argtypes = [vp, u32]
    for name in ("sots_write_population", "sots_read_population", "sots_read_population_other"):
        getattr(L, name).argtypes = [vp, vp, sz, vp, sz, vp, sz]
    L.sots_write_synth.argtypes = [vp, vp, sz, vp, sz]
    L.sots_read_synth.argtypes = [vp, vp, sz, vp, sz, vp, sz]
    L.sots_set_sort_mode.argtypes = [vp, u32]
    L.sots_set_synth_arithmetic.argtypes = [vp, u32]
    for name in ("recombine", "mutate", "synthesise", "window", "fft", "fitness", "sort", "select", "rotate"):
        getattr(L, "sots_stage_" + name).argtypes = [vp]
    L.sots_execute_generation.argtypes = [vp]
    L.sots_execute_generations.argtypes = [vp, u32]
    L.sots_get_generation.argtypes = [vp, C.POINTER(u32)]
    L.sots_set_generation.argtypes = [vp, u32]
    L.sots_timing_enable.argtypes = [vp, C.c_int]
    L.sots_timing_reset.argtypes = [vp]
    L.sots_stage_time_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sots_stage_launch_times_ms.argtypes = [vp, C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.sots_pack_elites_device.argtypes = [vp, vp, u32]
    L.sots_inject_immigrants_device.argtypes = [vp, vp, u32]
    L.sots_inject_gathered_device.argtypes = [vp, vp, u32, u32, u32]
    L.sots_fuse_exchange_next_sort.argtypes = [vp, vp, u32, vp, u32, u32, u32, vp]
    L.sots_pack_elites_host.argtypes = [vp, vp, u32]
    L.sots_inject_immigrants_host.argtypes = [vp, vp, u32]
    L.sots_get_info.argtypes = [vp, C.POINTER(Info)]
    L.sots_group_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_int32), u32, u32, u32, u32, C.POINTER(vp)]
    L.sots_group_destroy.argtypes = [vp]
    L.sots_group_destroy.restype = None
    L.sots_group_last_error.argtypes = [vp]
    L.sots_group_last_error.restype = C.c_char_p
    L.sots_group_size.argtypes = [vp]
    L.sots_group_size.restype = u32
    L.sots_group_uses_rccl.argtypes = [vp]
    L.sots_group_island.argtypes = [vp, u32]
    L.sots_group_island.restype = vp
    L.sots_group_set_target_audio.argtypes = [vp, vp, u32]
    L.sots_group_set_target_spectrum.argtypes = [vp, vp, u32]
    L.sots_group_init_population.argtypes = [vp, u32]
    L.sots_group_execute_generations.argtypes = [vp, u32]
    L.sots_group_synchronize.argtypes = [vp]
    L.sots_group_best.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float)]
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _nbytes(a):
    return 0 if a is None else a.nbytes


def make_config(num_parents, num_offspring, synth_kind=SYNTH_2OP, audio_log2=10, param_min=None, param_max=None,
                seed=0x5EED0001, workgroup_size=32, device=0, gid_base=0, num_generations=0):
    d = SYNTH_DIMS[synth_kind]
    cfg = Config()
    cfg.struct_size = C.sizeof(Config)
    cfg.num_parents, cfg.num_offspring, cfg.num_dimensions = num_parents, num_offspring, d
    cfg.audio_length_log2, cfg.num_generations = audio_log2, num_generations
    cfg.synth_kind, cfg.workgroup_size = synth_kind, workgroup_size
    cfg.device, cfg.gid_base, cfg.seed = device, gid_base, seed
    pmin = list(param_min) if param_min is not None else [0.0] * d
    pmax = list(param_max)
    for i in range(MAX_DIMS):
        cfg.param_min[i] = float(pmin[i]) if i < len(pmin) else 0.0
        cfg.param_max[i] = float(pmax[i]) if i < len(pmax) else 0.0
    return cfg


class HipES:
    """One evolutionary-strategy context on one MI355X (mirror of the C-ABI)."""

    @classmethod
    def borrowed(cls, handle, cfg):
        """A view of a context owned by somebody else (an island of a HipGroup): close() does not destroy it."""
        self = cls.__new__(cls)
        self.L = load()
        self.cfg = cfg
        self.P, self.D, self.N = cfg.num_parents + cfg.num_offspring, cfg.num_dimensions, 1 << cfg.audio_length_log2
        self.num_parents = cfg.num_parents
        self._h = C.c_void_p(handle)
        self._borrowed = True
        return self

    def __init__(self, num_parents, num_offspring, synth_kind=SYNTH_2OP, audio_log2=10,
                 param_min=None, param_max=None, seed=0x5EED0001, workgroup_size=32,
                 device=0, gid_base=0, num_generations=0):
        self.L = load()
        self._borrowed = False
        d = SYNTH_DIMS[synth_kind]
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.num_parents, cfg.num_offspring, cfg.num_dimensions = num_parents, num_offspring, d
        cfg.audio_length_log2, cfg.num_generations = audio_log2, num_generations
        cfg.synth_kind, cfg.workgroup_size = synth_kind, workgroup_size
        cfg.device, cfg.gid_base, cfg.seed = device, gid_base, seed
        pmin = list(param_min) if param_min is not None else [0.0] * d
        pmax = list(param_max)
        for i in range(MAX_DIMS):
            cfg.param_min[i] = float(pmin[i]) if i < len(pmin) else 0.0
            cfg.param_max[i] = float(pmax[i]) if i < len(pmax) else 0.0
        self.cfg = cfg
        self.P, self.D, self.N = num_parents + num_offspring, d, 1 << audio_log2
        self.num_parents = num_parents
        h = C.c_void_p()
        rc = self.L.sots_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise SotsError(rc, self.L.sots_last_error(None).decode())
        self._h = h

    # -- plumbing --
    def _check(self, rc):
        if rc != 0:
            raise SotsError(rc, self.L.sots_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self.L.sots_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        self._check(self.L.sots_set_stream(self._h, C.c_void_p(stream_handle)))

    def synchronize(self):
        self._check(self.L.sots_synchronize(self._h))

    def info(self):
        i = Info()
        self._check(self.L.sots_get_info(self._h, C.byref(i)))
        return i

    # -- target --
    def set_target_audio(self, audio):
        a = _f32(audio)
        self._check(self.L.sots_set_target_audio(self._h, _ptr(a), a.size))

    def set_target_spectrum(self, mag):
        m = _f32(mag)
        self._check(self.L.sots_set_target_spectrum(self._h, _ptr(m), m.size))

    # -- population --
    def init_population(self, chunk=0):
        self._check(self.L.sots_init_population(self._h, chunk))

    def write_population(self, values=None, steps=None, fitness=None):
        v = None if values is None else _f32(values)
        s = None if steps is None else _f32(steps)
        f = None if fitness is None else _f32(fitness)
        self._check(self.L.sots_write_population(self._h, _ptr(v), _nbytes(v), _ptr(s), _nbytes(s),
                                                 _ptr(f), _nbytes(f)))

    def read_population(self, other=False):
        v = np.empty((self.P, self.D), np.float32)
        s = np.empty((self.P, self.D), np.float32)
        f = np.empty(self.P, np.float32)
        fn = self.L.sots_read_population_other if other else self.L.sots_read_population
        self._check(fn(self._h, _ptr(v), v.nbytes, _ptr(s), s.nbytes, _ptr(f), f.nbytes))
        return v, s, f

    def read_fitness(self):
        f = np.empty(self.P, np.float32)
        self._check(self.L.sots_read_population(self._h, None, 0, None, 0, _ptr(f), f.nbytes))
        return f

    # -- synthesiser buffers --
    def write_audio(self, audio):
        a = _f32(audio)
        self._check(self.L.sots_write_synth(self._h, _ptr(a), a.nbytes, None, 0))

    def write_spectrum(self, spectrum):
        s = _f32(spectrum)
        self._check(self.L.sots_write_synth(self._h, None, 0, _ptr(s), s.nbytes))

    def read_audio(self):
        a = np.empty((self.P, self.N), np.float32)
        self._check(self.L.sots_read_synth(self._h, _ptr(a), a.nbytes, None, 0, None, 0))
        return a

    def read_spectrum(self):
        """complex64 [P][N/2+4]; bins 0..N/2 valid."""
        s = np.empty((self.P, self.N + 8), np.float32)
        self._check(self.L.sots_read_synth(self._h, None, 0, _ptr(s), s.nbytes, None, 0))
        return s.view(np.complex64)

    def read_target(self):
        t = np.empty(self.N // 2, np.float32)
        self._check(self.L.sots_read_synth(self._h, None, 0, None, 0, _ptr(t), t.nbytes))
        return t

    # -- stages --
    def recombine(self):
        self._check(self.L.sots_stage_recombine(self._h))

    def mutate(self):
        self._check(self.L.sots_stage_mutate(self._h))

    def synthesise(self):
        self._check(self.L.sots_stage_synthesise(self._h))

    def window(self):
        self._check(self.L.sots_stage_window(self._h))

    def fft(self):
        self._check(self.L.sots_stage_fft(self._h))

    def fitness(self):
        self._check(self.L.sots_stage_fitness(self._h))

    def sort(self):
        self._check(self.L.sots_stage_sort(self._h))

    def select(self):
        self._check(self.L.sots_stage_select(self._h))

    def rotate(self):
        self._check(self.L.sots_stage_rotate(self._h))

    def set_sort_mode(self, mode):
        self._check(self.L.sots_set_sort_mode(self._h, mode))

    def set_synth_arithmetic(self, arith):
        """ARITH_CPU_PATH (default) or ARITH_DEVICE_KERNELS: the reference's OpenCL kernels' arithmetic (enum sots_synth_arith)"""
        self._check(self.L.sots_set_synth_arithmetic(self._h, arith))

    def execute_generation(self):
        self._check(self.L.sots_execute_generation(self._h))

    def execute_generations(self, n):
        self._check(self.L.sots_execute_generations(self._h, n))

    @property
    def generation(self):
        g = C.c_uint32()
        self._check(self.L.sots_get_generation(self._h, C.byref(g)))
        return g.value

    @generation.setter
    def generation(self, g):
        self._check(self.L.sots_set_generation(self._h, g))

    # -- timing --
    def timing_enable(self, on=True):
        self._check(self.L.sots_timing_enable(self._h, 1 if on else 0))

    def timing_reset(self):
        self._check(self.L.sots_timing_reset(self._h))

    def stage_time_ms(self, stage):
        t, c = C.c_double(), C.c_uint64()
        self._check(self.L.sots_stage_time_ms(self._h, stage, C.byref(t), C.byref(c)))
        return t.value, c.value

    def stage_launch_times_ms(self, stage, capacity=65536):
        out = np.empty(capacity, np.float32)
        n = C.c_uint64()
        self._check(self.L.sots_stage_launch_times_ms(self._h, stage, _ptr(out), capacity, C.byref(n)))
        return out[:n.value].copy()

    # -- island exchange --
    def pack_elites_device(self, dev_ptr, n_rows):
        self._check(self.L.sots_pack_elites_device(self._h, C.c_void_p(dev_ptr), n_rows))

    def inject_immigrants_device(self, dev_ptr, n_rows):
        self._check(self.L.sots_inject_immigrants_device(self._h, C.c_void_p(dev_ptr), n_rows))

    def inject_gathered_device(self, dev_ptr, world, rank, elites):
        self._check(self.L.sots_inject_gathered_device(self._h, C.c_void_p(dev_ptr), world, rank, elites))

    def sort_places(self, n_rows):
        """True when every generation's sortPopulation places at least the first n_rows rows (the selection places
        rows 0..S-1 only, S = the whole parent blocks and never fewer than the parents; enum sots_sort_mode)."""
        block = max(1, self.cfg.workgroup_size)
        breeding = max(1, self.cfg.num_parents // block) * block
        return n_rows <= max(breeding, self.cfg.num_parents)

    def fuse_exchange_next_sort(self, elite_ptr, n_elite_rows, gathered_ptr, world, rank, elites, host_gate_event=None):
        """pack + inject folded into the sort of the last generation of the next execute_generations call; the host
        waits for host_gate_event (a raw hipEvent_t) right before it enqueues that sort"""
        self._check(self.L.sots_fuse_exchange_next_sort(self._h, C.c_void_p(elite_ptr) if elite_ptr else None, n_elite_rows,
                                                        C.c_void_p(gathered_ptr) if gathered_ptr else None, world, rank, elites,
                                                        C.c_void_p(host_gate_event) if host_gate_event else None))

    def pack_elites(self, n_rows):
        rows = np.empty((n_rows, 2 * self.D + 1), np.float32)
        self._check(self.L.sots_pack_elites_host(self._h, _ptr(rows), n_rows))
        return rows

    def inject_immigrants(self, rows):
        r = _f32(rows)
        self._check(self.L.sots_inject_immigrants_host(self._h, _ptr(r), r.shape[0]))


class HipGroup:
    """Islands inside the library: one process, one island per listed device (sots_group_* of the C-ABI)."""

    def __init__(self, devices, num_elites, num_parents, num_offspring, synth_kind=SYNTH_2OP, audio_log2=10,
                 param_min=None, param_max=None, seed=0x5EED0001, workgroup_size=32, gid_base=0,
                 migration_interval=1, overlap=False, force_rccl=False, unfused=False, event_waits=False):
        self.L = load()
        self.cfg = make_config(num_parents, num_offspring, synth_kind, audio_log2, param_min, param_max, seed, workgroup_size,
                               0, gid_base)
        devs = (C.c_int32 * len(devices))(*devices)
        flags = (GROUP_OVERLAP if overlap else 0) | (GROUP_FORCE_RCCL if force_rccl else 0) | (GROUP_UNFUSED if unfused else 0) | (GROUP_EVENT_WAITS if event_waits else 0)
        h = C.c_void_p()
        rc = self.L.sots_group_create(C.byref(self.cfg), devs, len(devices), num_elites, migration_interval, flags, C.byref(h))
        if rc != 0:
            raise SotsError(rc, self.L.sots_group_last_error(None).decode())
        self._h = h
        self.size = self.L.sots_group_size(h)
        self.uses_rccl = bool(self.L.sots_group_uses_rccl(h))

    def _check(self, rc):
        if rc != 0:
            raise SotsError(rc, self.L.sots_group_last_error(self._h).decode())

    def island(self, i):
        h = self.L.sots_group_island(self._h, i)
        if not h:
            raise IndexError(i)
        return HipES.borrowed(h, self.cfg)

    def set_target_audio(self, audio):
        a = _f32(audio)
        self._check(self.L.sots_group_set_target_audio(self._h, _ptr(a), a.size))

    def init_population(self, chunk=0):
        self._check(self.L.sots_group_init_population(self._h, chunk))

    def execute_generations(self, n):
        self._check(self.L.sots_group_execute_generations(self._h, n))

    def synchronize(self):
        self._check(self.L.sots_group_synchronize(self._h))

    def best(self):
        i, f = C.c_uint32(), C.c_float()
        self._check(self.L.sots_group_best(self._h, C.byref(i), C.byref(f)))
        return i.value, f.value

    def close(self):
        if getattr(self, "_h", None):
            self.L.sots_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
