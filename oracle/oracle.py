"""ctypes loader for the CPU oracle (oracle/sots_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product path never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsots_oracle.so")

MAX_DIMS = 16
WAVETABLE_SIZE = 32768
SYNTH_2OP, SYNTH_3OP_SERIES, SYNTH_TRIPLE_PAR, SYNTH_4OP_SERIES = 0, 1, 2, 3
SYNTH_DIMS = {SYNTH_2OP: 4, SYNTH_3OP_SERIES: 6, SYNTH_TRIPLE_PAR: 12, SYNTH_4OP_SERIES: 8}
TAG_INIT, TAG_MUTATE = 0x494E4954, 0x4D555441


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sots_oracle.c")
    hdr = os.path.join(_HERE, "sots_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsots_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Config(C.Structure):
    _fields_ = [
        ("num_parents", C.c_uint32), ("num_offspring", C.c_uint32),
        ("num_dims", C.c_uint32), ("audio_log2", C.c_uint32),
        ("synth_kind", C.c_uint32), ("recomb_block", C.c_uint32),
        ("gid_base", C.c_uint32), ("reserved", C.c_uint32),
        ("seed", C.c_uint64),
        ("param_min", C.c_float * MAX_DIMS), ("param_max", C.c_float * MAX_DIMS),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    L.sots_or_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.sots_or_wavetable.argtypes = [f32p]
    L.sots_or_window.argtypes = [f64p, C.c_uint32]
    L.sots_or_window.restype = C.c_float
    L.sots_or_synth_dims.argtypes = [C.c_uint32]
    L.sots_or_synth_dims.restype = C.c_uint32
    L.sots_or_synth.argtypes = [C.c_uint32, f32p, f32p, f32p, f32p, C.c_uint32, f32p]
    L.sots_or_synth_ocl.argtypes = [C.c_uint32, f32p, f32p, f32p, f32p, C.c_uint32, f32p, C.c_uint32]
    L.sots_or_spectrum.argtypes = [f32p, C.c_uint32, f64p, C.c_float, f32p]
    L.sots_or_rfft.argtypes = [f32p, C.c_uint32, f64p, f64p, f64p]
    L.sots_or_rfft_naive.argtypes = [f32p, C.c_uint32, f64p, f64p, f64p]
    L.sots_or_fitness.argtypes = [f32p, f32p, C.c_uint32]
    L.sots_or_fitness.restype = C.c_float
    L.sots_or_init_population.argtypes = [f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]
    L.sots_or_recombine.argtypes = [f32p, f32p, f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.sots_or_mutate.argtypes = [f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32]
    L.sots_or_sort_perm.argtypes = [f32p, C.c_uint32, u32p]
    L.sots_or_draw_unit.argtypes = [C.c_uint32]
    L.sots_or_draw_unit.restype = C.c_float
    L.sots_or_mutate_gene.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, u32p]
    L.sots_or_es_create.argtypes = [C.POINTER(Config)]
    L.sots_or_es_create.restype = C.c_void_p
    L.sots_or_es_destroy.argtypes = [C.c_void_p]
    L.sots_or_es_set_target_audio.argtypes = [C.c_void_p, f32p]
    L.sots_or_es_set_target_spectrum.argtypes = [C.c_void_p, f32p]
    L.sots_or_es_init_population.argtypes = [C.c_void_p, C.c_uint32]
    L.sots_or_es_write_population.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sots_or_es_read_population.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sots_or_es_set_generation.argtypes = [C.c_void_p, C.c_uint32]
    for name in ("recombine", "mutate", "evaluate", "sort", "generation"):
        getattr(L, "sots_or_es_" + name).argtypes = [C.c_void_p]
    L.sots_or_es_inject.argtypes = [C.c_void_p, f32p, C.c_uint32]
    L.sots_or_es_pack_elites.argtypes = [C.c_void_p, f32p, C.c_uint32]
    L.sots_or_es_audio.argtypes = [C.c_void_p]
    L.sots_or_es_audio.restype = C.POINTER(C.c_float)
    L.sots_or_es_spectrum.argtypes = [C.c_void_p]
    L.sots_or_es_spectrum.restype = C.POINTER(C.c_float)
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def philox(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().sots_or_philox4x32_10(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
    return out


def wavetable():
    t = np.zeros(WAVETABLE_SIZE, np.float32)
    lib().sots_or_wavetable(t)
    return t


def window(n):
    w = np.zeros(n, np.float64)
    f = lib().sots_or_window(w, n)
    return w, np.float32(f)


def pad_params(p):
    out = np.zeros(MAX_DIMS, np.float32)
    out[: len(p)] = p
    return out


def set_threads(n):
    """Threads of OracleES.evaluate()/generation() (all-cores CPU baseline only; default 1)."""
    lib().sots_or_set_threads(int(n))


def synth(kind, values, pmin, pmax, n, table=None):
    table = wavetable() if table is None else table
    out = np.zeros(n, np.float32)
    lib().sots_or_synth(kind, _f32(values), pad_params(pmin), pad_params(pmax), table, n, out)
    return out


def synth_ocl(kind, values, pmin, pmax, n, table_padded, contract):
    """one individual with the arithmetic of the reference's OpenCL kernels (table_padded: W + 1 entries)"""
    assert len(table_padded) >= WAVETABLE_SIZE + 1
    out = np.zeros(n, np.float32)
    lib().sots_or_synth_ocl(kind, _f32(values), pad_params(pmin), pad_params(pmax), _f32(table_padded), n, out, contract)
    return out


def spectrum(audio, win=None, wf=None):
    n = len(audio)
    if win is None:
        win, wf = window(n)
    mag = np.zeros(n // 2, np.float32)
    lib().sots_or_spectrum(_f32(audio), n, win, wf, mag)
    return mag


def rfft(audio, win=None, naive=False):
    n = len(audio)
    if win is None:
        win, _ = window(n)
    re = np.zeros(n // 2 + 1)
    im = np.zeros(n // 2 + 1)
    (lib().sots_or_rfft_naive if naive else lib().sots_or_rfft)(_f32(audio), n, win, re, im)
    return re + 1j * im


def fitness(mag, target):
    return np.float32(lib().sots_or_fitness(_f32(mag), _f32(target), len(target)))


def init_population(p, d, seed, gid_base=0, chunk=0):
    v = np.zeros((p, d), np.float32)
    s = np.zeros((p, d), np.float32)
    lib().sots_or_init_population(v, s, p, d, seed, gid_base, chunk)
    return v, s


def recombine(v, s, num_parents, block):
    v, s = _f32(v), _f32(s)
    p, d = v.shape
    vo, so = np.zeros_like(v), np.zeros_like(s)
    lib().sots_or_recombine(v, s, vo, so, p, d, num_parents, block)
    return vo, so


def mutate(v, s, seed, gid_base, generation):
    v, s = _f32(v).copy(), _f32(s).copy()
    p, d = v.shape
    lib().sots_or_mutate(v, s, p, d, seed, gid_base, generation)
    return v, s


def draw_unit(word):
    return np.float32(lib().sots_or_draw_unit(int(word) & 0xFFFFFFFF))


def mutate_gene(value, step, d, words):
    """(new value, new step) of one gene from the 13 random words its mutation consumes"""
    v, s = C.c_float(float(value)), C.c_float(float(step))
    lib().sots_or_mutate_gene(C.byref(v), C.byref(s), d, np.ascontiguousarray(words, dtype=np.uint32))
    return np.float32(v.value), np.float32(s.value)


def sort_perm(fit):
    fit = _f32(fit)
    perm = np.zeros(len(fit), np.uint32)
    lib().sots_or_sort_perm(fit, len(fit), perm)
    return perm


class OracleES:
    """Evolutionary_Strategy_CPU restatement (stage order of Evolutionary_Strategy_CPU.hpp:353-418)."""

    def __init__(self, num_parents, num_offspring, synth_kind=SYNTH_2OP, audio_log2=10,
                 param_min=None, param_max=None, seed=0x5EED0001, recomb_block=32, gid_base=0):
        d = SYNTH_DIMS[synth_kind]
        cfg = Config()
        cfg.num_parents, cfg.num_offspring, cfg.num_dims = num_parents, num_offspring, d
        cfg.audio_log2, cfg.synth_kind, cfg.recomb_block = audio_log2, synth_kind, recomb_block
        cfg.gid_base, cfg.seed = gid_base, seed
        pmin = pad_params(param_min if param_min is not None else [0.0] * d)
        pmax = pad_params(param_max)
        for i in range(MAX_DIMS):
            cfg.param_min[i] = float(pmin[i])
            cfg.param_max[i] = float(pmax[i])
        self.cfg = cfg
        self.P, self.D, self.N = num_parents + num_offspring, d, 1 << audio_log2
        self._h = lib().sots_or_es_create(C.byref(cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sots_or_es_destroy(self._h)
            self._h = None

    def set_target_audio(self, audio):
        lib().sots_or_es_set_target_audio(self._h, _f32(audio))

    def set_target_spectrum(self, mag):
        lib().sots_or_es_set_target_spectrum(self._h, _f32(mag))

    def init_population(self, chunk=0):
        lib().sots_or_es_init_population(self._h, chunk)

    def write_population(self, values=None, steps=None, fitness=None):
        def ptr(a):
            return None if a is None else _f32(a).ctypes.data_as(C.c_void_p)
        keep = [None if a is None else _f32(a) for a in (values, steps, fitness)]
        lib().sots_or_es_write_population(self._h, *[None if a is None else a.ctypes.data_as(C.c_void_p) for a in keep])

    def read_population(self):
        v = np.zeros((self.P, self.D), np.float32)
        s = np.zeros((self.P, self.D), np.float32)
        f = np.zeros(self.P, np.float32)
        lib().sots_or_es_read_population(self._h, v.ctypes.data_as(C.c_void_p),
                                         s.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p))
        return v, s, f

    def set_generation(self, g):
        lib().sots_or_es_set_generation(self._h, g)

    def recombine(self):
        lib().sots_or_es_recombine(self._h)

    def mutate(self):
        lib().sots_or_es_mutate(self._h)

    def evaluate(self):
        lib().sots_or_es_evaluate(self._h)

    def sort(self):
        lib().sots_or_es_sort(self._h)

    def generation(self):
        lib().sots_or_es_generation(self._h)

    def inject(self, rows):
        rows = _f32(rows)
        lib().sots_or_es_inject(self._h, rows, rows.shape[0])

    def pack_elites(self, n):
        rows = np.zeros((n, 2 * self.D + 1), np.float32)
        lib().sots_or_es_pack_elites(self._h, rows, n)
        return rows

    def audio(self):
        p = lib().sots_or_es_audio(self._h)
        return np.ctypeslib.as_array(p, shape=(self.P, self.N)).copy()

    def spectrum(self):
        p = lib().sots_or_es_spectrum(self._h)
        return np.ctypeslib.as_array(p, shape=(self.P, self.N // 2)).copy()
